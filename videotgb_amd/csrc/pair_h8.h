// pair_h8.h -- the operand format of RAFT's VTGB_F16C8 mode (gemm_h8.hip, raft_x3.hip): fp16 main value + two 8-bit correction operands.
//
// A product x.w is formed as
//     x . w  ~  xh . Wh                      fp16 x fp16 on v_mfma_f32_16x16x32_f16            (xh = fp16(x), Wh = fp16(w): 11 significant bits each)
//            +  2^-11/sw ( xl' . Wh8  +  xh8 . Wl' )    OCP fp8 on v_mfma_scale_f32_16x16x128_f8f6f4, twice the fp16 rate
// with  xl' = e5m2((x - xh) 2^11),  xh8 = e5m2(x),  Wh8 = e4m3(w sw),  Wl' = e4m3((w - Wh) sw 2^11),  sw = a power of two per layer (pack time).
// The two corrections are 2^-11 of the product, so the 3-4 significant bits of their operands leave an error of ~2^-16 of the product -- the
// level of the bf16x3 mode's pairs (tests/emul_f16c8.py: flows 4.4e-5 vs 1.3e-5 rel-RMS from fp32 RAFT with the update block in this form,
// 1.26e-4 vs 1.18e-4 with the encoders at bf16x3 in both), for 2/3 of its matrix-core time and 2/3 of its operand traffic.  e5m2 has fp16's
// exponent range, so the activations need NO data-dependent scale: |x| <= 57344 (values beyond saturate) down to 2^-14 keep their bits.
//
// Storage of a C-channel activation row (4 C bytes, the size of the bf16x3 pair row):
//     [ xh fp16 x C | per group of 4 channels: xl' x 4, xh8 x 4 ]
// i.e. the 8 correction bytes of channels 4g .. 4g+3 sit where the bf16 pair kept the lo values of those channels: every address of the pair
// layout (hi at column n, "lo" at column n + split_lo, in 2-byte units) stays valid.  A 64-channel chunk of the second half is one 128-byte
// k-tile row of the fp8 contraction; the weights are packed with the same byte order (Wh8 under xl', Wl' under xh8; ops.py h8_pack).
#pragma once
#include "common.h"

typedef unsigned h8_u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned))));
typedef _Float16 h8_f16x2 __attribute__((ext_vector_type(2)));
constexpr float H8_LO_SCALE = 2048.0f;       // 2^11: |x - fp16(x)| 2^11 <= |x|, so the scaled residual never leaves x's own range
constexpr float H8_MAX = 57344.0f;           // largest finite e5m2

#if defined(__HIPCC__)
// fp32 x 4 -> (hi: 4 fp16, lo: 4 e5m2 residuals | 4 e5m2 values)
__device__ __forceinline__ void h8_split4(f32x4 v, h8_u32x2& hi, h8_u32x2& lo) {
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = __builtin_amdgcn_fmed3f(v[e], -H8_MAX, H8_MAX);
    const h8_f16x2 h01 = {(_Float16)v[0], (_Float16)v[1]}, h23 = {(_Float16)v[2], (_Float16)v[3]};
    const float r0 = (v[0] - (float)h01[0]) * H8_LO_SCALE, r1 = (v[1] - (float)h01[1]) * H8_LO_SCALE;
    const float r2 = (v[2] - (float)h23[0]) * H8_LO_SCALE, r3 = (v[3] - (float)h23[1]) * H8_LO_SCALE;
    hi[0] = __builtin_bit_cast(unsigned, h01);
    hi[1] = __builtin_bit_cast(unsigned, h23);
    int l0 = __builtin_amdgcn_cvt_pk_bf8_f32(r0, r1, 0, false);
    l0 = __builtin_amdgcn_cvt_pk_bf8_f32(r2, r3, l0, true);
    int l1 = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
    l1 = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], l1, true);
    lo[0] = (unsigned)l0;
    lo[1] = (unsigned)l1;
}
// the value a pair stands for where it is read back element-wise (the GRU's h): xh + xl' 2^-11
__device__ __forceinline__ f32x4 h8_join4(const h8_u32x2 hi, const h8_u32x2 lo) {
    // (written on whole vectors: as four scalar fmaf's of (float)h01[0], .., (float)h23[1] hipcc (ROCm 7.2) packed the sums into two v_pk_fma_f32
    // and gave BOTH the first pair's fp16 values as addend -- elements 2, 3 came out as x0, x1 + their own residuals; tools/exp/h8_join_probe.hip)
    typedef _Float16 h8_f16x4 __attribute__((ext_vector_type(4)));
    const f32x4 hv = __builtin_convertvector(__builtin_bit_cast(h8_f16x4, hi), f32x4);
    const int l0 = (int)lo[0];
    const f32x4 lv = {__builtin_amdgcn_cvt_f32_bf8(l0, 0), __builtin_amdgcn_cvt_f32_bf8(l0, 1), __builtin_amdgcn_cvt_f32_bf8(l0, 2), __builtin_amdgcn_cvt_f32_bf8(l0, 3)};
    return lv * (1.0f / H8_LO_SCALE) + hv;
}
// one value: (fp16 bits, residual byte, value byte)
__device__ __forceinline__ void h8_split1(float v, unsigned short& hi, unsigned char& lo_r, unsigned char& lo_v) {
    v = __builtin_amdgcn_fmed3f(v, -H8_MAX, H8_MAX);
    const _Float16 h = (_Float16)v;
    hi = __builtin_bit_cast(unsigned short, h);
    const int l = __builtin_amdgcn_cvt_pk_bf8_f32((v - (float)h) * H8_LO_SCALE, v, 0, false);
    lo_r = (unsigned char)(l & 255);
    lo_v = (unsigned char)((l >> 8) & 255);
}
// byte offset of the correction bytes of channel c inside the "lo" half of a row: residual at h8_lo_off(c), value at h8_lo_off(c) + 4
__device__ __forceinline__ int h8_lo_off(int c) { return (c >> 2) * 8 + (c & 3); }
#endif
