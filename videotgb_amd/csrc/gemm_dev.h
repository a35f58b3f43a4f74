// gemm_dev.h -- device helpers shared by the MFMA GEMM / implicit-GEMM convolution kernels (gemm.hip, gemm_pp.hip).
#pragma once
#include "common.h"

enum { EPI_STORE = VTGB_EPI_STORE, EPI_GELU = VTGB_EPI_GELU, EPI_RESID_F32 = VTGB_EPI_RESID_F32,
       EPI_STORE_F32 = VTGB_EPI_STORE_F32, EPI_GRU = VTGB_EPI_GRU, EPI_SPLIT = VTGB_EPI_SPLIT, EPI_X3ZR = VTGB_EPI_X3ZR, EPI_X3Q = VTGB_EPI_X3Q };

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// GELU for bf16 outputs: x * Phi(x) with Phi(x) ~ sigmoid(x (a + b x^2 + c x^4)), a minimax fit on [-8, 8] (the argument is
// clamped there; Phi is 0 / 1 to 1e-12 outside): |error| <= 2.6e-5 absolute and <= 7e-4 of the result for |x| < 2.5, i.e. below
// the bf16 rounding of every result larger than 0.013 -- at 7 plain VALU instructions + v_exp + v_rcp (44 issue cycles per
// value).  The erf form it replaces (Abramowitz-Stegun 7.1.26: 15 plain + 2 transcendental, 76 cycles) made the fc1 epilogue
// 15 % of a tile's life: 128 values per lane, two waves per SIMD, nothing to hide under.  libm's erff is ~40 instructions.
// The fp32 exactness mode keeps the exact erf (gelu_erf).
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
    const float x2 = xc * xc;
    float q = fmaf(1.0145391570e-3f, x2, -1.0677742213e-1f);      // -(a + b x^2 + c x^4) log2(e), Horner in x^2
    q = fmaf(q, x2, -2.3011195660f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xc * q));
}
// two values at a time: the plain part on the packed fp32 pipe (v_pk_mul / v_pk_fma: r4 -- 128 values per lane in the fc1 epilogue), the
// clamp and the two transcendentals per value as before; bit-identical to gelu_erf_fast per element
typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f32x2 gelu_erf_fast2(gelu_f32x2 x) {
    const gelu_f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -8.0f, 8.0f), __builtin_amdgcn_fmed3f(x[1], -8.0f, 8.0f)};
    const gelu_f32x2 x2 = xc * xc;
    gelu_f32x2 q = __builtin_elementwise_fma(gelu_f32x2{1.0145391570e-3f, 1.0145391570e-3f}, x2, gelu_f32x2{-1.0677742213e-1f, -1.0677742213e-1f});
    q = __builtin_elementwise_fma(q, x2, gelu_f32x2{-2.3011195660f, -2.3011195660f});
    const gelu_f32x2 t = xc * q;
    const gelu_f32x2 d = gelu_f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
    return x * gelu_f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ void gelu_erf_fast4(f32x4& v) {
#ifdef VTGB_GELU_SCALAR      // (A/B builds)
    v = f32x4{gelu_erf_fast(v[0]), gelu_erf_fast(v[1]), gelu_erf_fast(v[2]), gelu_erf_fast(v[3])};
    return;
#endif
    const gelu_f32x2 a = gelu_erf_fast2(gelu_f32x2{v[0], v[1]}), b = gelu_erf_fast2(gelu_f32x2{v[2], v[3]});
    v = f32x4{a[0], a[1], b[0], b[1]};
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// tanh for the GRU candidate (bf16 mode): 1 - 2 / (exp(2x) + 1) with the hardware exp / rcp (~1e-6 relative, far below
// the bf16 rounding of the inputs) instead of libm tanhf's ~90 instructions per element, which made the GRU epilogue a
// VALU-bound tail as long as a third of the k-loop.  The exactness mode (conv_f32.hip) keeps tanhf.
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __expf(2.0f * x);                 // inf for large x -> 1 - 0 = 1; 0 for very negative x -> -1
    return 1.0f - 2.0f * __frcp_rn(e + 1.0f);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.0f / (1.0f + __expf(-v));
    return v;
}
// four values at a time: ONE scalar test per group (r4: per element, hipcc left a compare + branch per value in the unrolled epilogues --
// ~650 scalar instructions and taken branches per 256 x 256 tile of the ReLU convolutions)
__device__ __forceinline__ void apply_act4(f32x4& v, int act) {
    if (act == 1) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    } else if (act == 2) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = 1.0f / (1.0f + __expf(-v[e]));
    }
}

// Large-kernel epilogue split: bias (and the fp32 residual) are loaded INTO the accumulators before
// the k-loop -- 32 independent 16-byte loads per lane in flight while the first LDS-DMA tiles
// land, with the accumulator registers themselves as destination -- so that the tail of the tile
// is store-only.  (Measured: with the residual read in the tail, load -> add -> store chains at
// ~250 live VGPRs ran at ~6 B/clk/CU and the epilogue of the K=1408 projection took longer than its
// whole k-loop.)
// resid_late: the fp32 residual is added by the whole-row epilogue instead (large kernel, staged fp32 store)
template <int EPI>
__device__ __forceinline__ f32x4 acc_init4(const GemmDesc& p, int m, int n0, bool resid_late = false) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m >= p.M || n0 >= p.N) return v;
    const bool full = (n0 + 3 < p.N);
    if (EPI == EPI_RESID_F32 && !resid_late) {
        const float* r = p.resid + map_row(p.r_map, m) * p.ldr + n0;
        if (full && ((p.ldr & 3) == 0)) {
            v += *reinterpret_cast<const f32x4*>(r);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < p.N) v[i] += r[i];
        }
    }
    if (p.init_bf16 && !p.init_frag && full) {   // (launch_conv_gemm requires N % 4 == 0 and ldinit % 4 == 0 with init_bf16)
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.init_bf16) + (int64_t)m * p.ldinit + n0);
        v[0] += (float)t[0]; v[1] += (float)t[1]; v[2] += (float)t[2]; v[3] += (float)t[3];
    }
    return v;
}

template <int EPI>
__device__ __forceinline__ void store4(const GemmDesc& p, int m, int n0, f32x4 v) {
    if (m >= p.M || n0 >= p.N) return;
    const bool full = (n0 + 3 < p.N) && ((p.ldo & 3) == 0);
    const int64_t orow = map_row(p.o_map, m);
    if constexpr (EPI == EPI_STORE || EPI == EPI_STORE_F32) {
        if (p.act | (p.out_scale != 0.f)) {
            const float sc = p.out_scale != 0.f ? p.out_scale : 1.0f;
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = apply_act(v[i], p.act) * sc;
        }
    }
    if constexpr (EPI == EPI_GRU) {
        // h' = (1 - z) h + z tanh(acc + bias)
        const float* hp = p.resid + map_row(p.r_map, m) * p.ldr + n0;
        const bf16_t* zp = reinterpret_cast<const bf16_t*>(p.aux) + (int64_t)m * p.ldaux + n0;
        float* o = reinterpret_cast<float*>(p.out) + orow * p.ldo + n0;
        bf16_t* o2 = reinterpret_cast<bf16_t*>(p.out2) + (int64_t)m * p.ldo2 + n0;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (n0 + i < p.N) {
                const float z = (float)zp[i], h = hp[i], q = tanh_fast(v[i]);
                const float hn = (1.0f - z) * h + z * q;
                o[i] = hn;
                o2[i] = (bf16_t)hn;
            }
        return;
    }
    if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32) {
        float* o = reinterpret_cast<float*>(p.out) + orow * p.ldo + n0;
        if (full) {
            *reinterpret_cast<f32x4*>(o) = v;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < p.N) o[i] = v[i];
        }
    } else {
        if constexpr (EPI == EPI_GELU) {
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = gelu_erf_fast(v[i]);
        }
        bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + orow * p.ldo + n0;
        if (full) {
            const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *reinterpret_cast<bf16x4*>(o) = pk;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < p.N) o[i] = (bf16_t)v[i];
        }
    }
}


// ---- LayerNorm folded into the GEMMs around it (GemmDesc::ln_*): the SAME functions in every kernel that produces or consumes the
// statistics, so that a row's numbers do not depend on the kernel its batch size selects.
// sum and sum of squares of four consecutive values, fixed association
__device__ __forceinline__ void ln_part4(const f32x4 v, float& s, float& q) {
    s = (v[0] + v[1]) + (v[2] + v[3]);
    q = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
}
// rstd (acc - mean cs) + c, element-wise
__device__ __forceinline__ f32x4 ln_fold4(const f32x4 acc, float mean, float rstd, const f32x4 cs, const f32x4 c) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = fmaf(rstd, fmaf(-mean, cs[e], acc[e]), c[e]);
    return o;
}
