// raft.hip -- the recurrent part of RAFT on gfx950: per refinement iteration the
// correlation-pyramid lookup (the reference's missing `alt_cuda_corr`, raft_utils/corr.py:29-50,63-91),
// BasicMotionEncoder, SepConvGRU and FlowHead (raft_utils/update.py:39-144), then the mask head and the
// convex 8x upsample of the last iteration (xraft.py:88-99).  The two CNN encoders are raft_enc.hip, the all-pairs
// correlation + pyramid is raft_corr.hip.
//
// Layout: every activation is NHWC ("pixel-major") bf16, so each convolution is an implicit GEMM on the
// MFMA kernel of gemm.hip (LDS-DMA gathers the k-tile of the shifted pixel directly; out-of-image taps read
// a zero page; channel concatenations [h | inp | motion | flow] are virtual: two base pointers, no copies).
// The hidden state h and the flow / coordinates stay fp32, the correlation pyramid is fp16 (or fp32); GEMM operands are bf16
// with fp32 accumulation, gates (sigmoid / tanh) are evaluated in fp32 in the GEMM epilogue.
// VTGB_F32 (exactness mode): the same launch sequence with every bf16 buffer in fp32 and the convolutions on
// conv_f32.hip's FMA kernel (launch_conv_gemm dispatches on GemmDesc::dtype).
#include <math.h>
#include <string.h>

#include "common.h"
#include "pair_h8.h"

// ---------------------------------------------------------------------------------------
// state init: NCHW fp32 -> pixel-major buffers
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void raft_init_kernel(const float* __restrict__ net, const float* __restrict__ inp, const float* __restrict__ cnet_nhwc,
                                 float* __restrict__ h32, T* __restrict__ hb, T* __restrict__ hlo, T* __restrict__ X, float* __restrict__ flow,
                                 const float* __restrict__ flow_init, int64_t M, int HW) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * 128) return;
    const int64_t m = i >> 7;
    const int c = (int)(i & 127);
    const int64_t n = m / HW, p = m % HW;
    // net = tanh(cnet[:, :128]), inp = relu(cnet[:, 128:]) (xraft.py:126-127), either precomputed NCHW or straight
    // from the context encoder's pixel-major output
    const float hv = cnet_nhwc ? tanhf(cnet_nhwc[m * 256 + c]) : net[(n * 128 + c) * HW + p];
    const float iv = cnet_nhwc ? fmaxf(cnet_nhwc[m * 256 + 128 + c], 0.f) : inp[(n * 128 + c) * HW + p];
    if (hlo) {      // fused GRU (gru_fused.hip): the hidden state as a bf16 pair hi | lo instead of fp32 + bf16
        const float hi = bf16_round(hv);
        hb[i] = (T)hi;
        hlo[i] = (T)(hv - hi);
    } else {
        h32[i] = hv;
        hb[i] = (T)hv;
    }
    X[m * 256 + c] = (T)iv;
    if (c < 2) {   // flow = coords1 - coords0: zero, or flow_init [n, 2, H8, W8] (xraft.py:131-132)
        const float f0 = flow_init ? flow_init[(n * 2 + c) * HW + p] : 0.f;
        flow[m * 2 + c] = f0;
        X[m * 256 + 254 + c] = (T)f0;
    }
}

// The bench's form of the state init (bf16, context features pixel-major, fused GRU: hidden state as a bf16 hi | lo pair): four channels
// per thread with 16-byte loads, no per-thread 64-bit division (the image index is only needed for flow_init, by one thread per pixel),
// tanh through the hardware exp / rcp (1e-6 relative: below the hi | lo pair's 2^-17).  The element-wise kernel above -- libm tanhf, two
// 64-bit divisions and six 2- or 4-byte accesses per element -- was vector-instruction bound at 1.58 ms for 2.31 M pixels (its 4.7 GB at
// HBM rate: 0.9 ms); it still serves the fp32 mode and NCHW inputs.
__global__ __launch_bounds__(256) void raft_init_nhwc_bf16_kernel(const float* __restrict__ cnet, bf16_t* __restrict__ hb, bf16_t* __restrict__ hlo,
                                                                  bf16_t* __restrict__ X, float* __restrict__ flow, const float* __restrict__ flow_init,
                                                                  int64_t M, int HW) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (pixel, 4-channel group): 32 groups per pixel
    if (i >= M * 32) return;
    const int64_t m = i >> 5;
    const int c = (int)(i & 31) * 4;
    const f32x4 nv = *reinterpret_cast<const f32x4*>(cnet + m * 256 + c), iv = *reinterpret_cast<const f32x4*>(cnet + m * 256 + 128 + c);
    bf16x4 hi4, lo4, x4;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float ex = __expf(2.0f * nv[e]);                       // tanh = 1 - 2 / (exp(2 x) + 1); inf -> 1, 0 -> -1
        const float hv = 1.0f - 2.0f * __builtin_amdgcn_rcpf(ex + 1.0f);
        const float hi = bf16_round(hv);
        hi4[e] = (bf16_t)hi;
        lo4[e] = (bf16_t)(hv - hi);
        x4[e] = (bf16_t)fmaxf(iv[e], 0.f);
    }
    *reinterpret_cast<bf16x4*>(hb + m * 128 + c) = hi4;
    *reinterpret_cast<bf16x4*>(hlo + m * 128 + c) = lo4;
    *reinterpret_cast<bf16x4*>(X + m * 256 + c) = x4;
    if (c == 0) {                                                    // flow = coords1 - coords0: zero, or flow_init [n, 2, H8, W8] (xraft.py:131-132)
        float f0 = 0.f, f1 = 0.f;
        if (flow_init) {
            const int64_t n = m / HW, pp = m - n * HW;
            f0 = flow_init[(n * 2) * HW + pp];
            f1 = flow_init[(n * 2 + 1) * HW + pp];
        }
        flow[m * 2] = f0;
        flow[m * 2 + 1] = f1;
        X[m * 256 + 254] = (bf16_t)f0;
        X[m * 256 + 255] = (bf16_t)f1;
    }
}

// ---------------------------------------------------------------------------------------
// correlation lookup: one wave per pixel, 4 levels x 9 x 9 bilinear taps (zero outside), bf16 out,
// columns 324..383 zero (K padded to a multiple of 64 for the 1x1 convolution that follows).
// The 81 taps of a level share their fractional weights and read a 10 x 10 window of the pixel's own
// correlation map: the wave gathers the four windows into LDS once (2 loads per lane and level instead of
// 4 per tap) and forms the taps from there.
// NB the reference adds stack(meshgrid(dy, dx)) to (x, y): the x coordinate receives the ROW offset of
// the 9x9 window (corr.py:36-43) -- replicated as is.
// ---------------------------------------------------------------------------------------

constexpr int CL_PIX = 16;  // pixels per wave (the per-lane tables are built once and reused; 4 levels x 16 pixels = one lane each below)
// Software-pipelined over the wave's pixels: the 8 window loads of pixel i + 1 are in flight while pixel i is interpolated out of
// LDS; the 384 outputs of a pixel leave through an LDS staging row as ONE 16-byte store per lane (48 lanes; was six 2-byte stores
// per lane -- the store path is per-instruction bound).
// r4: the kernel was VALU-bound, not HBM-bound (218 vector instructions per pixel and wave = 250 cycles per pixel and CU = its
// whole 1.08 ms; it moves 4.4 GB).  Now (~90): the wave-uniform part of a pixel's work -- its coordinates, the four levels' window
// origins, byte offsets and fractions -- is computed ONCE for all 16 pixels x 4 levels, one (pixel, level) per lane, and read back
// with v_readlane (64 lanes of floorf / divisions per pixel and level before); the window loads go through a buffer descriptor of
// the pixel's own correlation map, whose range check returns the zeros of the padding (rows above / below fall outside by
// themselves, columns left / right get an out-of-range offset): no 64-bit address arithmetic, no validity bits carried to the
// deposit; the bilinear form is P + wy (Q - P) on the packed pairs the LDS reads deliver, then one more lerp along x; only the last
// 64 taps carry a guard.
// timing-only ablation builds of the standalone lookup (tools/exp/build_variant.sh -DCL_ABL=n): 1 no window loads, 2 no interpolation, 4 no
// output store.  Measured (0.90 ms whole): no loads 0.72, no interpolation 0.89, no store 0.73, no loads + no interpolation 0.35, none of the
// three 0.32 -- the memory side (gathers + tap store) and the issue side (the interpolation: 33 LDS instructions per pixel, ~116 LDS cycles
// per pixel and CU) bound it alternately; what is left with all three removed is addresses, deposit and the loop itself.
#ifndef CL_ABL
#define CL_ABL 0
#endif
typedef float lk_f32x2 __attribute__((ext_vector_type(2)));
template <typename CT>
__device__ __forceinline__ CT lk_load(__amdgpu_buffer_rsrc_t r, unsigned off);
template <>
__device__ __forceinline__ _Float16 lk_load<_Float16>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0));
}
template <>
__device__ __forceinline__ float lk_load<float>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
// PAIR (bf16x3 mode, OT = bf16): a pixel's row is [hi(384) | lo(384)], hi = bf16(v), lo = bf16(v - hi); PAIR == 2 (f16c8 mode): the f16c8
// pair row of pair_h8.h (fp16 values, then 8 correction bytes per 4 taps)
template <typename CT, typename OT, int PAIR = 0>
__global__ __launch_bounds__(256) void raft_corr_lookup_kernel(const CorrPyr pyr, const float* __restrict__ flow, OT* __restrict__ out,
                                                               int64_t M, int H8, int W8) {
    constexpr int OW = PAIR ? 768 : 384;   // elements per output row
    __shared__ float win[4][4][104];   // [wave][level][10 x 10 window | wx | wy | pad]
    __shared__ __attribute__((aligned(16))) OT stage[4][2][OW];
    // (readfirstlane: the wave index -- and with it the pixel index and the level bases -- is wave-uniform; told so, hipcc keeps that
    // arithmetic on the scalar unit: r3)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wy0 = lane / 10, wx0 = lane - wy0 * 10;                 // window element `lane`
    const int e1 = lane < 36 ? lane + 64 : 99;                        // window element `lane + 64` (lanes >= 36 repeat the last one)
    const int wy1 = e1 / 10, wx1 = e1 - wy1 * 10;
    int la[4], lb[4];                                                 // byte offsets of the lane's two window elements from the window origin
#pragma unroll
    for (int l = 0; l < 4; l++) {
        la[l] = (wy0 * pyr.w[l] + wx0) * (int)sizeof(CT);
        lb[l] = (wy1 * pyr.w[l] + wx1) * (int)sizeof(CT);
    }
    // output k = kk * 64 + lane = level * 81 + i * 9 + j  (i: x offset, j: y offset): LDS offsets, fixed per lane; k >= 324: zero
    int tap_off[6], frac_off[6];
#pragma unroll
    for (int kk = 0; kk < 6; kk++) {
        const int k = kk * 64 + lane;
        const int l = k / 81, t = k - l * 81, i = t / 9, j = t - i * 9;
        tap_off[kk] = k < 324 ? l * 104 + j * 10 + i : 0;
        frac_off[kk] = (k < 324 ? l : 0) * 104 + 100;
    }
    float* const wv = &win[wave][0][0];
    const int HW = H8 * W8;
    const int64_t m_first = ((int64_t)blockIdx.x * 4 + wave) * CL_PIX;
    if (m_first >= M) return;                                         // wave-uniform
    const int npx = (int)(M - m_first < CL_PIX ? M - m_first : CL_PIX);
    const int p_first = (int)(m_first % HW);                          // one 64-bit division per wave, not per pixel
    // lane = 4 * pixel + level: window origin (x, byte offset) and fractions of that pixel at that level (corr.py:36-43 with
    // bilinear_sampler's align_corners coordinates, utils.py:58-71)
    int sx0, sbase;
    float sqx, sqy;
    {
        const int spi = (lane >> 2) < npx ? (lane >> 2) : npx - 1, sl = lane & 3;
        int p = p_first + spi;                                        // pixel inside its image (CL_PIX <= HW)
        p = p >= HW ? p - HW : p;
        const float2 f = *reinterpret_cast<const float2*>(flow + (m_first + spi) * 2);
        const float cx = (float)(p % W8) + f.x, cy = (float)(p / W8) + f.y;
        const float sc = 1.0f / (float)(1 << sl);
        const float xs = cx * sc, ys = cy * sc, x0f = floorf(xs), y0f = floorf(ys);
        // (far outside the map the whole window is padding; clamped so that the byte offsets below stay far from 32-bit wrap-around)
        const int x0 = (int)fminf(fmaxf(x0f, -32768.f), 32768.f) - 4, y0 = (int)fminf(fmaxf(y0f, -32768.f), 32768.f) - 4;
        const int wl = sl == 0 ? pyr.w[0] : sl == 1 ? pyr.w[1] : sl == 2 ? pyr.w[2] : pyr.w[3];
        sx0 = x0;
        sbase = (y0 * wl + x0) * (int)sizeof(CT);
        sqx = xs - x0f; sqy = ys - y0f;
    }
    CT cur[8], nxt[8];
    // the window of pixel pi: 2 loads per level into r[2 l], r[2 l + 1]
    const CT* lvl0[4];                                                // the wave's first pixel's maps (64-bit arithmetic once per wave)
#pragma unroll
    for (int l = 0; l < 4; l++) lvl0[l] = reinterpret_cast<const CT*>(pyr.lvl[l]) + m_first * (int64_t)(pyr.h[l] * pyr.w[l]);
#define CL_FETCH(pi, r)                                                                                 \
    {                                                                                                   \
        _Pragma("unroll") for (int l = 0; l < 4; l++) {                                                 \
            const int hw = pyr.h[l] * pyr.w[l];                                                         \
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<CT*>(lvl0[l] + (unsigned)((pi) * hw)), 0, \
                                                              hw * (int)sizeof(CT), 0x00020000);        \
            const int x0 = __builtin_amdgcn_readlane(sx0, (pi) * 4 + l), b0 = __builtin_amdgcn_readlane(sbase, (pi) * 4 + l); \
            const unsigned oa = (unsigned)(x0 + wx0) < (unsigned)pyr.w[l] ? (unsigned)(b0 + la[l]) : 0xFFFFFFF0u; \
            const unsigned ob = (unsigned)(x0 + wx1) < (unsigned)pyr.w[l] ? (unsigned)(b0 + lb[l]) : 0xFFFFFFF0u; \
            if (CL_ABL & 1) { r[2 * l] = (CT)(float)(oa & 7); r[2 * l + 1] = (CT)(float)(ob & 7); }     \
            else { r[2 * l] = lk_load<CT>(rs, oa); r[2 * l + 1] = lk_load<CT>(rs, ob); }                \
        }                                                                                               \
    }
#define CL_DEPOSIT(pi, r)                                                                               \
    _Pragma("unroll") for (int l = 0; l < 4; l++) {                                                     \
        wv[l * 104 + lane] = (float)r[2 * l];                                                           \
        if (lane < 36) wv[l * 104 + lane + 64] = (float)r[2 * l + 1];                                   \
    }                                                                                                   \
    if ((lane >> 2) == (pi)) { wv[(lane & 3) * 104 + 100] = sqx; wv[(lane & 3) * 104 + 101] = sqy; }
    constexpr int CHUNKS = OW * (int)sizeof(OT) / 16;
#define CL_STORE(pi)                                                                                    \
    _Pragma("unroll") for (int c = lane; c < CHUNKS; c += 64)                                           \
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(out + (m_first + (pi)) * OW) + c * 16) =      \
            *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(&stage[wave][(pi) & 1][0]) + c * 16);
    CL_FETCH(0, cur)
    CL_DEPOSIT(0, cur)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // the windows are private to this wave
    // iteration pi: store pixel pi - 1 (staged last time) | issue the window loads of pixel pi + 1 | interpolate pixel pi out of
    // LDS into the other staging row | wait for the loads (the only vmcnt wait of the iteration: the store in front of them
    // has had the whole interpolation to complete) and deposit them
    for (int pi = 0; pi < npx; pi++) {
        if (pi > 0 && !(CL_ABL & 4)) { CL_STORE(pi - 1) }
        const int pn = pi + 1 < npx ? pi + 1 : pi;                    // (the last iteration re-fetches its own pixel: no tail branch)
        CL_FETCH(pn, nxt)
#pragma unroll
        for (int kk = 0; kk < ((CL_ABL & 2) ? 1 : 6); kk++) {
            const float wx = wv[frac_off[kk]], wy = wv[frac_off[kk] + 1];
            const float* q = wv + tap_off[kk];
            const lk_f32x2 top = {q[0], q[1]}, bot = {q[10], q[11]};
            const lk_f32x2 c = top + wy * (bot - top);                // both columns interpolated along y (packed)
            float v = c[0] + wx * (c[1] - c[0]);
            if (kk == 5) v = lane < 4 ? v : 0.f;                      // k >= 324: the zero padding of K
            if constexpr (PAIR == 2) {
                unsigned short h16; unsigned char lr, lv;
                h8_split1(v, h16, lr, lv);
                reinterpret_cast<unsigned short*>(&stage[wave][pi & 1][0])[kk * 64 + lane] = h16;
                unsigned char* lo8 = reinterpret_cast<unsigned char*>(&stage[wave][pi & 1][384]) + h8_lo_off(kk * 64 + lane);
                lo8[0] = lr; lo8[4] = lv;
            } else if constexpr (PAIR == 1) {
                const float hf = bf16_round(v);
                stage[wave][pi & 1][kk * 64 + lane] = (OT)hf;
                stage[wave][pi & 1][384 + kk * 64 + lane] = (OT)(v - hf);
            } else
            stage[wave][pi & 1][kk * 64 + lane] = (OT)v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // window reads done, staging row written
        CL_DEPOSIT(pn, nxt)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    CL_STORE(npx - 1)
#undef CL_STORE
#undef CL_FETCH
#undef CL_DEPOSIT
}

// ---------------------------------------------------------------------------------------
// Correlation lookup + convc1 (update.py:88-91: cor = relu(convc1(corr)), a 1x1 convolution 324 -> 256) in ONE launch (r4, bf16 mode).
// After the lookup's instruction diet (above) the pair was HBM-bound on a round trip: the lookup wrote 384 bf16 taps per pixel
// (1.77 GB per launch at the bench batch) that convc1 read straight back.  Here a workgroup takes 64 pixels: its eight waves look
// up 8 pixels each -- the same per-pixel code, four pixels' window loads in flight per wave since only 16 waves fit a CU -- and
// deposit the taps as rows of a [64][352] bf16 operand tile in LDS (row pitch 784 B: ds_read_b128 of 16 rows spreads over all
// banks); then wave w multiplies the tile with its 32 output channels' weights, streamed from L2 in MFMA fragment order (packed
// once per call, 176 KB shared by every workgroup; 11 k-steps of 32, K = 324 zero-padded to 352 instead of 384), adds the bias,
// applies the ReLU and the tile leaves through LDS as whole 512-byte rows.  Two workgroups per CU: one's lookup runs beside the
// other's MFMA phase.  Traffic per launch 2.7 GB of windows + 1.2 GB out (was 4.4 + 2.95 GB over two launches).
// ---------------------------------------------------------------------------------------
constexpr int LC_PX = 64, LC_LDA = 392, LC_KS = 11, LC_LDO = 264, LC_WAVES = 8, LC_WPX = LC_PX / LC_WAVES;
constexpr int LC_A_BYTES = LC_PX * LC_LDA * 2;
constexpr int LC_LDS = LC_A_BYTES + LC_WAVES * 4 * 104 * 4;
constexpr size_t LC_WPK_BYTES = (size_t)LC_WAVES * LC_KS * 2 * 1024;

// packed[((w * 11 + ks) * 2 + i) * 64 + lane][e] = W[w * 32 + i * 16 + (lane & 15)][ks * 32 + (lane >> 4) * 8 + e]   (W row-major [256][384])
__global__ __launch_bounds__(256) void raft_lkc1_pack_w_kernel(const bf16_t* __restrict__ w, bf16x8* __restrict__ packed) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= LC_WAVES * LC_KS * 2 * 64) return;
    const int lane = idx & 63, i = (idx >> 6) & 1, ks = (idx >> 7) % LC_KS, wv = (idx >> 7) / LC_KS;
    packed[idx] = *reinterpret_cast<const bf16x8*>(w + (int64_t)(wv * 32 + i * 16 + (lane & 15)) * 384 + ks * 32 + (lane >> 4) * 8);
}

template <typename CT>
__global__ __launch_bounds__(512, 2) void raft_lookup_convc1_kernel(const CorrPyr pyr, const float* __restrict__ flow, const bf16x8* __restrict__ wpk,
                                                                    const float* __restrict__ bias, bf16_t* __restrict__ c1, int64_t M, int H8, int W8) {
    extern __shared__ __attribute__((aligned(16))) char lc_smem[];
    bf16_t* const At = reinterpret_cast<bf16_t*>(lc_smem);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* const wv = reinterpret_cast<float*>(lc_smem + LC_A_BYTES) + wave * (4 * 104);
    const int fr = lane & 15, fg = lane >> 4;
    // this wave's weight fragments of the first three k-steps: in flight during the whole lookup phase
    const bf16x8* wp = wpk + (int64_t)wave * LC_KS * 2 * 64 + lane;
    bf16x8 wf[4][2];                                                  // ring of four k-steps: three in flight ahead of the one multiplied
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 2; i++) wf[k][i] = wp[(k * 2 + i) * 64];
    const int wy0 = lane / 10, wx0 = lane - wy0 * 10;
    const int e1 = lane < 36 ? lane + 64 : 99;
    const int wy1 = e1 / 10, wx1 = e1 - wy1 * 10;
    int la[4], lb[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
        la[l] = (wy0 * pyr.w[l] + wx0) * (int)sizeof(CT);
        lb[l] = (wy1 * pyr.w[l] + wx1) * (int)sizeof(CT);
    }
    int tap_off[6], frac_off[6];
#pragma unroll
    for (int kk = 0; kk < 6; kk++) {
        const int k = kk * 64 + lane;
        const int l = k / 81, t = k - l * 81, i = t / 9, j = t - i * 9;
        tap_off[kk] = k < 324 ? l * 104 + j * 10 + i : 0;
        frac_off[kk] = (k < 324 ? l : 0) * 104 + 100;
    }
    const int HW = H8 * W8;
    const int64_t m_tile = (int64_t)blockIdx.x * LC_PX, m_first = m_tile + wave * LC_WPX;
    const int npx = (int)(M - m_first < LC_WPX ? (M - m_first > 0 ? M - m_first : 0) : LC_WPX);   // (a trailing wave may have none: its rows are never stored)
#ifndef LC_ABL
#define LC_ABL 0        // timing-only ablation builds (tools/exp/build_variant.sh -DLC_ABL=n): 1 no lookup phase, 2 no MFMA loop, 4 no output
#endif                  // measured (1.38 ms whole): no lookup 0.44, no MFMA loop 0.99, no output 1.22, lookup alone 0.87 -- the phases ADD: the
                        // gathers (8 scattered 2-byte loads per pixel) and the weight stream (176 KB per tile) share the CU's vector-memory
                        // path, so a half-tile stagger of the CU's second workgroup changed nothing (1.37 vs 1.37 ms)
    if (npx > 0 && !(LC_ABL & 1)) {
        const int p_first = (int)(m_first % HW);
        int sx0, sbase;
        float sqx, sqy;
        {
            const int spi = (lane >> 2) < npx ? (lane >> 2) : npx - 1, sl = lane & 3;
            int p = p_first + spi;
            p = p >= HW ? p - HW : p;
            const float2 f = *reinterpret_cast<const float2*>(flow + (m_first + spi) * 2);
            const float cx = (float)(p % W8) + f.x, cy = (float)(p / W8) + f.y;
            const float sc = 1.0f / (float)(1 << sl);
            const float xs = cx * sc, ys = cy * sc, x0f = floorf(xs), y0f = floorf(ys);
            const int x0 = (int)fminf(fmaxf(x0f, -32768.f), 32768.f) - 4, y0 = (int)fminf(fmaxf(y0f, -32768.f), 32768.f) - 4;
            const int wl = sl == 0 ? pyr.w[0] : sl == 1 ? pyr.w[1] : sl == 2 ? pyr.w[2] : pyr.w[3];
            sx0 = x0;
            sbase = (y0 * wl + x0) * (int)sizeof(CT);
            sqx = xs - x0f; sqy = ys - y0f;
        }
        CT r[4][8];
        const CT* lvl0[4];                                            // the wave's first pixel's maps (64-bit arithmetic once per wave)
#pragma unroll
        for (int l = 0; l < 4; l++) lvl0[l] = reinterpret_cast<const CT*>(pyr.lvl[l]) + m_first * (int64_t)(pyr.h[l] * pyr.w[l]);
#define LC_FETCH(pi, d)                                                                                 \
    {                                                                                                   \
        _Pragma("unroll") for (int l = 0; l < 4; l++) {                                                 \
            const int hw = pyr.h[l] * pyr.w[l];                                                         \
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<CT*>(lvl0[l] + (unsigned)((pi) * hw)), 0, \
                                                              hw * (int)sizeof(CT), 0x00020000);        \
            const int x0 = __builtin_amdgcn_readlane(sx0, (pi) * 4 + l), b0 = __builtin_amdgcn_readlane(sbase, (pi) * 4 + l); \
            const unsigned oa = (unsigned)(x0 + wx0) < (unsigned)pyr.w[l] ? (unsigned)(b0 + la[l]) : 0xFFFFFFF0u; \
            const unsigned ob = (unsigned)(x0 + wx1) < (unsigned)pyr.w[l] ? (unsigned)(b0 + lb[l]) : 0xFFFFFFF0u; \
            r[d][2 * l] = lk_load<CT>(rs, oa);                                                          \
            r[d][2 * l + 1] = lk_load<CT>(rs, ob);                                                      \
        }                                                                                               \
    }
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int pf = d < npx ? d : npx - 1;
            LC_FETCH(pf, d)
        }
        // branch-free over all the wave's rows (hipcc's wait counting gives up across a conditional refill: it then waited out every load
        // right after issuing it); rows past the wave's last pixel -- only in the launch's last tile -- repeat that pixel and are never stored
        for (int pb = 0; pb < LC_WPX; pb += 4) {
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int pi = pb + d;
#pragma unroll
                for (int l = 0; l < 4; l++) {
                    wv[l * 104 + lane] = (float)r[d][2 * l];
                    if (lane < 36) wv[l * 104 + lane + 64] = (float)r[d][2 * l + 1];
                }
                if ((lane >> 2) == (pi < npx ? pi : npx - 1)) { wv[(lane & 3) * 104 + 100] = sqx; wv[(lane & 3) * 104 + 101] = sqy; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const int pf = pi + 4 < npx ? pi + 4 : npx - 1;
                LC_FETCH(pf, d)
                bf16_t* arow = At + (wave * LC_WPX + pi) * LC_LDA + lane;
#pragma unroll
                for (int kk = 0; kk < 6; kk++) {
                    const float wx = wv[frac_off[kk]], wy = wv[frac_off[kk] + 1];
                    const float* q = wv + tap_off[kk];
                    const lk_f32x2 top = {q[0], q[1]}, bot = {q[10], q[11]};
                    const lk_f32x2 c = top + wy * (bot - top);
                    float v = c[0] + wx * (c[1] - c[0]);
                    if (kk == 5) v = lane < 4 ? v : 0.f;
                    arow[kk * 64] = (bf16_t)v;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
#undef LC_FETCH
    }
    __syncthreads();
    // ---- [64 pixels][352] x this wave's [32 channels][352]^T
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + wave * 32 + i * 16 + fg * 4);
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = b;
    }
#pragma unroll
    for (int ks = 0; ks < ((LC_ABL & 2) ? 0 : LC_KS); ks++) {
        if (ks + 3 < LC_KS) {
#pragma unroll
            for (int i = 0; i < 2; i++) wf[(ks + 3) & 3][i] = wp[((ks + 3) * 2 + i) * 64];
        }
        bf16x8 af[4];
#pragma unroll
        for (int j = 0; j < 4; j++) af[j] = *reinterpret_cast<const bf16x8*>(At + (j * 16 + fr) * LC_LDA + ks * 32 + fg * 8);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 3][i], af[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                                  // every wave has read the operand tile: it becomes the output tile
    bf16_t* const Ot = At;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const f32x4 v = acc[i][j];
            const bf16x4 o = {(bf16_t)fmaxf(v[0], 0.f), (bf16_t)fmaxf(v[1], 0.f), (bf16_t)fmaxf(v[2], 0.f), (bf16_t)fmaxf(v[3], 0.f)};
            *reinterpret_cast<bf16x4*>(Ot + (j * 16 + fr) * LC_LDO + wave * 32 + i * 16 + fg * 4) = o;
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int idx = it * 512 + threadIdx.x, px = idx >> 5, c = idx & 31;
        if (m_tile + px < M && (!(LC_ABL & 4) || bias[0] == 12345.f))
            *reinterpret_cast<uint4*>(c1 + (m_tile + px) * 256 + c * 8) = *reinterpret_cast<const uint4*>(Ot + px * LC_LDO + c * 8);
    }
}

// ---------------------------------------------------------------------------------------
// The same fusion at VTGB_F16C8 (round 6; raft_x3.hip): lookup -> the operand tile as f16c8 pair rows in LDS ([64][392] fp16 values + [64][784 B]
// correction bytes: taps 4g .. 4g+3 as (xl' x 4, xh8 x 4), csrc/pair_h8.h) -> wave w multiplies it with its 32 channels' weights -- 11 fp16 k-steps of 32
// (v_mfma_f32_16x16x32_f16) and 6 fp8 k-tiles of 128 bytes (v_mfma_scale_f32_16x16x128_f8f6f4, K = 324 zero-padded to 384), both streamed from L2 in
// MFMA fragment order (packed once per call from the table's [256][fp16 x 384 | correction bytes x 768]) -> bias, ReLU -> the f16c8 pair rows of c1 leave
// through LDS as whole 1 KiB rows.  The tile is 113 KB: ONE workgroup per CU (the bf16 form runs two, one's lookup beside the other's MFMAs); what is saved
// is the 1.5 KB-per-pixel tap tensor's write + read and a launch.
// ---------------------------------------------------------------------------------------
constexpr int LH_KS16 = 11, LH_KT8 = 6, LH_LDA8 = 784, LH_LDO = 1040;      // fp16 k-steps, fp8 k-tiles, correction-row pitch (bytes), output-row pitch (bytes)
constexpr int LH_A16_BYTES = LC_PX * LC_LDA * 2, LH_A8_BYTES = LC_PX * LH_LDA8;
constexpr int LH_LDS = LH_A16_BYTES + LH_A8_BYTES + LC_WAVES * 4 * 104 * 4;
constexpr size_t LH_WPK_BYTES = (size_t)LC_WAVES * (LH_KS16 * 2 * 1024 + LH_KT8 * 2 * 2048);
static_assert(LC_PX * LH_LDO <= LH_A16_BYTES + LH_A8_BYTES, "the output tile reuses the operand tile");
typedef int lh_i32x4 __attribute__((ext_vector_type(4)));
typedef int lh_i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lh_f16x8 __attribute__((ext_vector_type(8)));

// w: [256][768 16-bit units] = per output channel fp16 values of the 384 taps, then 768 correction bytes (ops.h8_conv_pack of a 1x1 convolution).
// packed (16-byte units): wave w: [ks < 11][i < 2][lane] fp16 fragments (taps ks * 32 + (lane >> 4) * 8 ..), then [t < 6][i < 2][half < 2][lane] the
// two 16-byte pieces of the fp8 fragment (bytes t * 128 + half * 64 + (lane >> 4) * 16 ..) of channel w * 32 + i * 16 + (lane & 15)
__global__ __launch_bounds__(256) void raft_lkc1_h8_pack_w_kernel(const bf16_t* __restrict__ w, lh_i32x4* __restrict__ packed) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    constexpr int PER_WAVE = LH_KS16 * 2 * 64 + LH_KT8 * 2 * 2 * 64;
    if (idx >= LC_WAVES * PER_WAVE) return;
    const int wv = idx / PER_WAVE, r = idx - wv * PER_WAVE, lane = r & 63;
    const char* row;
    int off;
    if (r < LH_KS16 * 2 * 64) {
        const int i = (r >> 6) & 1, ks = r >> 7;
        row = reinterpret_cast<const char*>(w + (int64_t)(wv * 32 + i * 16 + (lane & 15)) * 768);
        off = (ks * 32 + (lane >> 4) * 8) * 2;
    } else {
        const int q = (r - LH_KS16 * 2 * 64) >> 6, half = q & 1, i = (q >> 1) & 1, t = q >> 2;
        row = reinterpret_cast<const char*>(w + (int64_t)(wv * 32 + i * 16 + (lane & 15)) * 768) + 768;
        off = t * 128 + half * 64 + (lane >> 4) * 16;
    }
    packed[idx] = *reinterpret_cast<const lh_i32x4*>(row + off);
}

template <typename CT>
__global__ __launch_bounds__(512, 2) void raft_lookup_convc1_h8_kernel(const CorrPyr pyr, const float* __restrict__ flow, const lh_i32x4* __restrict__ wpk,
                                                                       const int* __restrict__ scale, const float* __restrict__ bias, bf16_t* __restrict__ c1, int64_t M,
                                                                       int H8, int W8) {
    extern __shared__ __attribute__((aligned(16))) char lc_smem[];
    unsigned short* const At = reinterpret_cast<unsigned short*>(lc_smem);                 // fp16 values [64][LC_LDA]
    unsigned char* const A8 = reinterpret_cast<unsigned char*>(lc_smem + LH_A16_BYTES);   // correction bytes [64][LH_LDA8]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* const wv = reinterpret_cast<float*>(lc_smem + LH_A16_BYTES + LH_A8_BYTES) + wave * (4 * 104);
    const int fr = lane & 15, fg = lane >> 4;
    const int sc_w = *scale;
    // this wave's fp16 weight fragments of the first three k-steps: in flight during the whole lookup phase
    const lh_i32x4* wp = wpk + (int64_t)wave * (LH_KS16 * 2 * 64 + LH_KT8 * 2 * 2 * 64) + lane;
    lh_i32x4 wf[4][2];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 2; i++) wf[k][i] = wp[(k * 2 + i) * 64];
    const int wy0 = lane / 10, wx0 = lane - wy0 * 10;
    const int e1 = lane < 36 ? lane + 64 : 99;
    const int wy1 = e1 / 10, wx1 = e1 - wy1 * 10;
    int la[4], lb[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
        la[l] = (wy0 * pyr.w[l] + wx0) * (int)sizeof(CT);
        lb[l] = (wy1 * pyr.w[l] + wx1) * (int)sizeof(CT);
    }
    int tap_off[6], frac_off[6];
#pragma unroll
    for (int kk = 0; kk < 6; kk++) {
        const int k = kk * 64 + lane;
        const int l = k / 81, t = k - l * 81, i = t / 9, j = t - i * 9;
        tap_off[kk] = k < 324 ? l * 104 + j * 10 + i : 0;
        frac_off[kk] = (k < 324 ? l : 0) * 104 + 100;
    }
    const int HW = H8 * W8;
    const int64_t m_tile = (int64_t)blockIdx.x * LC_PX, m_first = m_tile + wave * LC_WPX;
    const int npx = (int)(M - m_first < LC_WPX ? (M - m_first > 0 ? M - m_first : 0) : LC_WPX);
#ifndef LH_ABL
#define LH_ABL 0        // timing-only ablation builds (tools/exp/build_variant.sh NAME raft.hip -DLH_ABL=n): 1 no lookup phase, 2 no MFMA loops, 4 no output.
#endif                  // Measured (2.62 ms whole): no lookup 0.99, no MFMA loops 1.84, no output 2.42, neither lookup nor MFMA 0.40 -- the phases ADD (one workgroup
                        // per CU: 113 KB of LDS); the lookup phase is 1.6 ms here against 0.87 in the bf16 kernel, whose second workgroup hides the gathers'
                        // latency.  A vector form of the deposit (fp32 staging row, one four-value split and two 8-byte writes per group) changed nothing
                        // (2.71 vs 2.62 ms): the deposit is not what the phase waits for.
    if (npx > 0 && !(LH_ABL & 1)) {
        const int p_first = (int)(m_first % HW);
        int sx0, sbase;
        float sqx, sqy;
        {
            const int spi = (lane >> 2) < npx ? (lane >> 2) : npx - 1, sl = lane & 3;
            int p = p_first + spi;
            p = p >= HW ? p - HW : p;
            const float2 f = *reinterpret_cast<const float2*>(flow + (m_first + spi) * 2);
            const float cx = (float)(p % W8) + f.x, cy = (float)(p / W8) + f.y;
            const float sc = 1.0f / (float)(1 << sl);
            const float xs = cx * sc, ys = cy * sc, x0f = floorf(xs), y0f = floorf(ys);
            const int x0 = (int)fminf(fmaxf(x0f, -32768.f), 32768.f) - 4, y0 = (int)fminf(fmaxf(y0f, -32768.f), 32768.f) - 4;
            const int wl = sl == 0 ? pyr.w[0] : sl == 1 ? pyr.w[1] : sl == 2 ? pyr.w[2] : pyr.w[3];
            sx0 = x0;
            sbase = (y0 * wl + x0) * (int)sizeof(CT);
            sqx = xs - x0f; sqy = ys - y0f;
        }
#ifndef LH_DEPTH
#define LH_DEPTH 4      // pixels whose window loads are in flight per wave (of LC_WPX = 8).  -DLH_DEPTH=8 (all of them up front) measured SLOWER: 2.83 vs 2.60 ms, same box
#endif
        CT r[LH_DEPTH][8];
        const CT* lvl0[4];
#pragma unroll
        for (int l = 0; l < 4; l++) lvl0[l] = reinterpret_cast<const CT*>(pyr.lvl[l]) + m_first * (int64_t)(pyr.h[l] * pyr.w[l]);
#define LH_FETCH(pi, d)                                                                                 \
    {                                                                                                   \
        _Pragma("unroll") for (int l = 0; l < 4; l++) {                                                 \
            const int hw = pyr.h[l] * pyr.w[l];                                                         \
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<CT*>(lvl0[l] + (unsigned)((pi) * hw)), 0, \
                                                              hw * (int)sizeof(CT), 0x00020000);        \
            const int x0 = __builtin_amdgcn_readlane(sx0, (pi) * 4 + l), b0 = __builtin_amdgcn_readlane(sbase, (pi) * 4 + l); \
            const unsigned oa = (unsigned)(x0 + wx0) < (unsigned)pyr.w[l] ? (unsigned)(b0 + la[l]) : 0xFFFFFFF0u; \
            const unsigned ob = (unsigned)(x0 + wx1) < (unsigned)pyr.w[l] ? (unsigned)(b0 + lb[l]) : 0xFFFFFFF0u; \
            r[d][2 * l] = lk_load<CT>(rs, oa);                                                          \
            r[d][2 * l + 1] = lk_load<CT>(rs, ob);                                                      \
        }                                                                                               \
    }
#pragma unroll
        for (int d = 0; d < LH_DEPTH; d++) {
            const int pf = d < npx ? d : npx - 1;
            LH_FETCH(pf, d)
        }
        for (int pb = 0; pb < LC_WPX; pb += LH_DEPTH) {
#pragma unroll
            for (int d = 0; d < LH_DEPTH; d++) {
                const int pi = pb + d;
#pragma unroll
                for (int l = 0; l < 4; l++) {
                    wv[l * 104 + lane] = (float)r[d][2 * l];
                    if (lane < 36) wv[l * 104 + lane + 64] = (float)r[d][2 * l + 1];
                }
                if ((lane >> 2) == (pi < npx ? pi : npx - 1)) { wv[(lane & 3) * 104 + 100] = sqx; wv[(lane & 3) * 104 + 101] = sqy; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (LH_DEPTH < LC_WPX) {
                    const int pf = pi + LH_DEPTH < npx ? pi + LH_DEPTH : npx - 1;
                    LH_FETCH(pf, d)
                }
                const int prow = wave * LC_WPX + pi;
#pragma unroll
                for (int kk = 0; kk < 6; kk++) {
                    const float wx = wv[frac_off[kk]], wy = wv[frac_off[kk] + 1];
                    const float* q = wv + tap_off[kk];
                    const lk_f32x2 top = {q[0], q[1]}, bot = {q[10], q[11]};
                    const lk_f32x2 c = top + wy * (bot - top);
                    float v = c[0] + wx * (c[1] - c[0]);
                    if (kk == 5) v = lane < 4 ? v : 0.f;
                    unsigned short h16; unsigned char lr, lv;
                    h8_split1(v, h16, lr, lv);
                    const int k = kk * 64 + lane;
                    At[prow * LC_LDA + k] = h16;
                    unsigned char* lo8 = A8 + prow * LH_LDA8 + h8_lo_off(k);
                    lo8[0] = lr; lo8[4] = lv;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
#undef LH_FETCH
    }
    __syncthreads();
    // ---- [64 pixels][384 taps] x this wave's [32 channels]^T: 11 fp16 k-steps, then 6 fp8 k-tiles (their fragments follow the fp16 ones in the stream)
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + wave * 32 + i * 16 + fg * 4);
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = b;
    }
    const lh_i32x4* wp8 = wp + LH_KS16 * 2 * 64;
    lh_i32x4 w8[2][2][2];      // ring of two fp8 k-tiles: [slot][i][half]
#pragma unroll
    for (int ks = 0; ks < ((LH_ABL & 2) ? 0 : LH_KS16); ks++) {
        if (ks + 3 < LH_KS16) {
#pragma unroll
            for (int i = 0; i < 2; i++) wf[(ks + 3) & 3][i] = wp[((ks + 3) * 2 + i) * 64];
        } else if (ks + 3 - LH_KS16 < 2) {      // the first two fp8 k-tiles, requested under the last fp16 k-steps
            const int t = ks + 3 - LH_KS16;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int h = 0; h < 2; h++) w8[t][i][h] = wp8[((t * 2 + i) * 2 + h) * 64];
        }
        lh_i32x4 af[4];
#pragma unroll
        for (int j = 0; j < 4; j++) af[j] = *reinterpret_cast<const lh_i32x4*>(At + (j * 16 + fr) * LC_LDA + ks * 32 + fg * 8);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lh_f16x8, wf[ks & 3][i]), __builtin_bit_cast(lh_f16x8, af[j]), acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < ((LH_ABL & 2) ? 0 : LH_KT8); t++) {
        lh_i32x8 a8[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned char* rp = A8 + (j * 16 + fr) * LH_LDA8 + t * 128 + fg * 16;
            const lh_i32x4 lo = *reinterpret_cast<const lh_i32x4*>(rp), hi = *reinterpret_cast<const lh_i32x4*>(rp + 64);
            a8[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        lh_i32x8 wv8[2];
#pragma unroll
        for (int i = 0; i < 2; i++) wv8[i] = __builtin_shufflevector(w8[t & 1][i][0], w8[t & 1][i][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv8[i], a8[j], acc[i][j], 0, 1, 0, sc_w, 0, 127);
        if (t + 2 < LH_KT8) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int h = 0; h < 2; h++) w8[t & 1][i][h] = wp8[(((t + 2) * 2 + i) * 2 + h) * 64];
        }
    }
    __syncthreads();                                                  // every wave has read the operand tile: it becomes the output tile
    char* const Ot = lc_smem;                                          // [64][LH_LDO]: fp16 x 256 | correction bytes x 512
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            f32x4 v = acc[i][j];
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
            h8_u32x2 hu, lu;
            h8_split4(v, hu, lu);
            const int ch = wave * 32 + i * 16 + fg * 4;
            char* o = Ot + (j * 16 + fr) * LH_LDO;
            *reinterpret_cast<h8_u32x2*>(o + ch * 2) = hu;
            *reinterpret_cast<h8_u32x2*>(o + 512 + ch * 2) = lu;
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int idx = it * 512 + threadIdx.x, px = idx >> 6, c = idx & 63;
        if (m_tile + px < M && (!(LH_ABL & 4) || bias[0] == 12345.f))
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(c1 + (m_tile + px) * 512) + c * 16) = *reinterpret_cast<const uint4*>(Ot + px * LH_LDO + c * 16);
    }
}

int raft_lkc1_h8_pack(const void* w, void* packed, hipStream_t s) {
    const int n = LC_WAVES * (LH_KS16 * 2 * 64 + LH_KT8 * 2 * 2 * 64);
    hipLaunchKernelGGL(raft_lkc1_h8_pack_w_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const bf16_t*)w, (lh_i32x4*)packed);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
size_t raft_lkc1_h8_pack_bytes() { return LH_WPK_BYTES; }
int raft_launch_lookup_convc1_h8(const CorrPyr& pyr, const float* flow, const void* wpk, const int* scale, const float* bias, void* c1, int64_t M, int H8, int W8,
                                 hipStream_t s) {
    static DeviceOnce attr;
    VTGB_FUNC_LDS_ONCE(attr, raft_lookup_convc1_h8_kernel<float>, LH_LDS);
    hipLaunchKernelGGL(raft_lookup_convc1_h8_kernel<float>, dim3((unsigned)((M + LC_PX - 1) / LC_PX)), dim3(512), LH_LDS, s, pyr, flow, (const lh_i32x4*)wpk, scale, bias,
                       (bf16_t*)c1, M, H8, W8);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---------------------------------------------------------------------------------------
// convf1: 7x7 convolution of the 2-channel flow -> 128 channels + ReLU (update.py:82,92) on the matrix cores.
// K = 49 taps x {x_hi, y_hi, x_lo, y_lo}: the fp32 flow enters as a bf16 head plus a bf16 remainder (exact to
// ~2^-17) against bf16 weights (each weight appears for the head and the remainder), 196 padded to 224 = 7
// MFMA k-steps.  Persistent workgroups: the packed weights [128][224] live in LDS, a wave takes 16 pixels at
// a time, builds its activation fragments straight from the flow field (two taps per lane and k-step) and
// sends the 16 x 128 outputs through a swizzled LDS tile so that whole 256-byte rows are stored.
// Also deposits the flow (bf16) in columns 254..255 of X (the motion features end with the flow, :97).
// ---------------------------------------------------------------------------------------
constexpr int CF1_K = 224, CF1_LD = 232;   // packed K and its LDS row pitch (elements)
__global__ __launch_bounds__(256) void raft_convf1_kernel(const float* __restrict__ flow, const bf16_t* __restrict__ wp, const float* __restrict__ b,
                                                          bf16_t* __restrict__ f1, bf16_t* __restrict__ X, int64_t M, int H8, int W8) {
    extern __shared__ __attribute__((aligned(16))) char cf1_smem[];
    bf16_t* const ws = reinterpret_cast<bf16_t*>(cf1_smem);              // [128][CF1_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const cst = cf1_smem + 128 * CF1_LD * 2 + wave * 4096;         // [16 px][128 ch] bf16, 16-byte chunks XOR px
    for (int i = tid; i < 128 * (CF1_K / 8); i += 256) {
        const int row = i / (CF1_K / 8), c = i - row * (CF1_K / 8);
        *reinterpret_cast<uint4*>(ws + row * CF1_LD + c * 8) = *reinterpret_cast<const uint4*>(wp + row * CF1_K + c * 8);
    }
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4, HW = H8 * W8;
    const int64_t n_groups = (M + 15) >> 4;
    // r4: the 14 tap loads of a group (2 per lane and k-step) used to sit behind bounds branches, each k-step waiting out its own loads
    // (~5800 cycles per group for 900 cycles of MFMA).  They now go through a buffer descriptor of the flow field with an out-of-range
    // offset for padding taps (the range check returns the zeros), are issued for the NEXT group before the current one is multiplied,
    // and the per-pixel 64-bit modulo became one per wave.
    const auto frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(flow), 0, (int)(M * 8), 0x00020000);
    int tdy[7][2], tdx[7][2];                                            // this lane's taps: offsets from the pixel, or a row far outside for k >= 49
#pragma unroll
    for (int ks = 0; ks < 7; ks++)
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
            const int tap = ks * 8 + fg * 2 + tt, ky = tap / 7, kx = tap - ky * 7;
            tdy[ks][tt] = tap < 49 ? ky - 3 : 1 << 20;
            tdx[ks][tt] = kx - 3;
        }
    const float inv_w = 1.0f / (float)W8;
    float2 cur[7][2], nxt[7][2];
#define CF1_FETCH(gg, r)                                                                                                  \
    {                                                                                                                      \
        const int64_t g0 = (gg) * 16;                                                                                      \
        const int p0 = __builtin_amdgcn_readfirstlane((int)(g0 % HW));                                                     \
        int pix = p0 + fr;                                                                                                 \
        pix = pix >= HW ? pix - HW : pix;                                                                                  \
        const int y = (int)(((float)pix + 0.5f) * inv_w), x = pix - y * W8;                                                \
        const unsigned mb = (unsigned)(g0 + fr);                                                                           \
        const bool valid = g0 + fr < M;                                                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 7; ks++) _Pragma("unroll") for (int tt = 0; tt < 2; tt++) {                \
            const int yy = y + tdy[ks][tt], xx = x + tdx[ks][tt];                                                          \
            const bool ok = valid && (unsigned)yy < (unsigned)H8 && (unsigned)xx < (unsigned)W8;                           \
            const unsigned off = ok ? (mb + (unsigned)(tdy[ks][tt] * W8 + tdx[ks][tt])) * 8u : 0xFFFFFFF0u;                \
            r[ks][tt] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(frs, off, 0, 0));                  \
        }                                                                                                                  \
    }
    const int64_t g_first = (int64_t)blockIdx.x * 4 + wave, g_step = (int64_t)gridDim.x * 4;
    if (g_first < n_groups) CF1_FETCH(g_first, cur)
    for (int64_t g = g_first; g < n_groups; g += g_step) {
        const int64_t m = g * 16 + fr;
        const bool valid = m < M;
        const int64_t gn = g + g_step < n_groups ? g + g_step : g;      // (the last pass re-fetches its own group: no tail branch)
        CF1_FETCH(gn, nxt)
        f32x4 acc[8];
#pragma unroll
        for (int ct = 0; ct < 8; ct++) acc[ct] = *reinterpret_cast<const f32x4*>(b + ct * 16 + fg * 4);
#pragma unroll
        for (int ks = 0; ks < 7; ks++) {
            bf16x8 xf;
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                const float2 f = cur[ks][tt];
                const float hxf = bf16_round(f.x), hyf = bf16_round(f.y);      // (hi | lo split on the bits: common.h)
                xf[tt * 4 + 0] = (bf16_t)hxf; xf[tt * 4 + 1] = (bf16_t)hyf;
                xf[tt * 4 + 2] = (bf16_t)(f.x - hxf); xf[tt * 4 + 3] = (bf16_t)(f.y - hyf);
            }
#pragma unroll
            for (int ct = 0; ct < 8; ct++) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(ws + (ct * 16 + fr) * CF1_LD + ks * 32 + fg * 8);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[ct], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 7; ks++) { cur[ks][0] = nxt[ks][0]; cur[ks][1] = nxt[ks][1]; }
        // D: column (lane & 15) = pixel, rows fg * 4 + reg = channel within the 16-channel tile
#pragma unroll
        for (int ct = 0; ct < 8; ct++) {
            const f32x4 v = acc[ct];
            const bf16x4 pk = {(bf16_t)fmaxf(v[0], 0.f), (bf16_t)fmaxf(v[1], 0.f), (bf16_t)fmaxf(v[2], 0.f), (bf16_t)fmaxf(v[3], 0.f)};
            const int chunk = (ct * 2 + (fg >> 1)) ^ fr;
            *reinterpret_cast<bf16x4*>(cst + fr * 256 + chunk * 16 + (fg & 1) * 8) = pk;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the tile is exchanged between lanes of this wave only
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int row = rr * 4 + fg, ch = fr;
            const uint4 v = *reinterpret_cast<const uint4*>(cst + row * 256 + ((ch ^ row) << 4));
            const int64_t mo = g * 16 + row;
            if (mo < M) *reinterpret_cast<uint4*>(f1 + mo * 128 + ch * 8) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (valid && fg == 0) {
            X[m * 256 + 254] = (bf16_t)flow[m * 2];
            X[m * 256 + 255] = (bf16_t)flow[m * 2 + 1];
        }
    }
#undef CF1_FETCH
}

// Exactness mode of convf1: the 7x7 convolution of the 2-channel flow as plain fp32 FMAs (98 taps per output); weights
// [98][128] (k = c * 49 + ky * 7 + kx, output channel minor).  Two pixels per workgroup, one thread per output channel.
__global__ __launch_bounds__(256) void raft_convf1_f32_kernel(const float* __restrict__ flow, const float* __restrict__ wt, const float* __restrict__ b,
                                                              float* __restrict__ f1, float* __restrict__ X, int64_t M, int H8, int W8) {
    __shared__ float win[2][100];
    const int tid = threadIdx.x, HW = H8 * W8;
    if (tid < 196) {
        const int px = tid / 98, k = tid - px * 98, c = k / 49, t = k - c * 49, ky = t / 7, kx = t - ky * 7;
        const int64_t m = (int64_t)blockIdx.x * 2 + px;
        float v = 0.f;
        if (m < M) {
            const int pix = (int)(m % HW), y = pix / W8 + ky - 3, x = pix % W8 + kx - 3;
            if ((unsigned)y < (unsigned)H8 && (unsigned)x < (unsigned)W8) v = flow[(m + (ky - 3) * W8 + (kx - 3)) * 2 + c];
        }
        win[px][k] = v;
    }
    __syncthreads();
    const int px = tid >> 7, co = tid & 127;
    const int64_t m = (int64_t)blockIdx.x * 2 + px;
    if (m >= M) return;
    float acc = b[co];
    for (int k = 0; k < 98; k++) acc = fmaf(win[px][k], wt[k * 128 + co], acc);
    f1[m * 128 + co] = fmaxf(acc, 0.f);
    if (co < 2) X[m * 256 + 254 + co] = flow[m * 2 + co];
}

// FlowHead.conv2 (3x3, 256 -> 2; update.py:14,18) + coords1 += delta (xraft.py:145).  The convolution is a
// GEMM with the taps moved to the OUTPUT side: P[m][tap*2 + o] = <FH[m], w[o][tap]> for every pixel (one pass
// over FH on the MFMA kernel, N = 18 padded to 32), then delta[m][o] = sum_tap P[m + offset(tap)][tap*2 + o]
// over the in-image neighbours -- 72 bytes per pixel instead of nine 512-byte rows.
// One workgroup per image (r3): the 18 used columns of the image's P rows are staged tap-major in LDS with coalesced 16-byte loads, then
// every pixel gathers its 9 taps from LDS (consecutive lanes = consecutive pixels = consecutive banks).  Rounds 1-2 gathered straight
// from global memory -- nine 8-byte loads per pixel, each touching 64 different 128-byte rows per wave instruction: 0.29 ms per launch
// for 0.33 GB.  Same taps in the same order: bit-identical flow.
__global__ __launch_bounds__(256) void raft_flow_head2_kernel(const float* __restrict__ P, const float* __restrict__ b, float* __restrict__ flow,
                                                              int64_t M, int H8, int W8) {
    extern __shared__ float fh2_ps[];                      // [18][HW]
    const int HW = H8 * W8, tid = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * HW;
    for (int i = tid; i < HW * 5; i += 256) {               // floats 0 .. 19 of every pixel's 32-float row
        const int px = i / 5, j = i - px * 5;
        const float4 v = *reinterpret_cast<const float4*>(P + (m0 + px) * 32 + j * 4);
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (j * 4 + q < 18) fh2_ps[(j * 4 + q) * HW + px] = e[q];
    }
    __syncthreads();
    const float b0 = b[0], b1 = b[1];
    for (int pix = tid; pix < HW; pix += 256) {
        const int y = pix / W8, x = pix - y * W8;
        float a0 = b0, a1 = b1;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if ((unsigned)(y + dy) < (unsigned)H8 && (unsigned)(x + dx) < (unsigned)W8) {
                const int q = pix + dy * W8 + dx;
                a0 += fh2_ps[(tap * 2) * HW + q];
                a1 += fh2_ps[(tap * 2 + 1) * HW + q];
            }
        }
        float2* f = reinterpret_cast<float2*>(flow + (m0 + pix) * 2);
        float2 o = *f;
        o.x += a0; o.y += a1;
        *f = o;
    }
}

// The same sum with the taps gathered straight from global memory (the form of rounds 1-2): images whose 18 x H8 x W8 floats do not fit
// the LDS image (> 2275 coarse pixels, e.g. 384 x 384 frames).  Same taps in the same order: the same flow bit for bit.
__global__ __launch_bounds__(256) void raft_flow_head2_gather_kernel(const float* __restrict__ P, const float* __restrict__ b, float* __restrict__ flow,
                                                                     int64_t M, int H8, int W8) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int HW = H8 * W8, pix = (int)(m % HW), y = pix / W8, x = pix - y * W8;
    float a0 = b[0], a1 = b[1];
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        if ((unsigned)(y + dy) < (unsigned)H8 && (unsigned)(x + dx) < (unsigned)W8) {
            const float2 v = *reinterpret_cast<const float2*>(P + (m + dy * W8 + dx) * 32 + tap * 2);
            a0 += v.x;
            a1 += v.y;
        }
    }
    float2* f = reinterpret_cast<float2*>(flow + m * 2);
    float2 o = *f;
    o.x += a0; o.y += a1;
    *f = o;
}

// upsample_flow (xraft.py:88-99): softmax over the 9 mask logits of each fine pixel, convex combination of
// the 3x3 neighbourhood of 8 * flow (zero padded unfold).  mask [M, 576] fp32 with channel = k*64 + sy*8 + sx.
__global__ void raft_upsample_kernel(const float* __restrict__ flow, const float* __restrict__ mask, float* __restrict__ up, int64_t n_img,
                                     int H8, int W8) {
    // a wave = one coarse pixel, lane = (sy, sx): the 9 mask reads of a wave are 256 contiguous bytes each (r3; a thread per fine pixel in
    // row order read 32-byte pieces of 8 different coarse pixels per wave instruction: 2.6 TB/s on the 5.3 GB of mask logits)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int HF = 8 * H8, WF = 8 * W8;
    if (i >= n_img * HF * WF) return;
    const int64_t m = i >> 6;
    const int sx = (int)(i & 7), sy = (int)((i >> 3) & 7);
    const int x = (int)(m % W8), y = (int)((m / W8) % H8);
    const int64_t n = m / ((int64_t)W8 * H8);
    const int X = x * 8 + sx, Y = y * 8 + sy;
    float lg[9], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        lg[k] = mask[m * 576 + k * 64 + sy * 8 + sx];
        mx = fmaxf(mx, lg[k]);
    }
    float sum = 0.f, f0 = 0.f, f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const float e = expf(lg[k] - mx);
        sum += e;
        const int dy = k / 3 - 1, dx = k % 3 - 1;
        if ((unsigned)(y + dy) < (unsigned)H8 && (unsigned)(x + dx) < (unsigned)W8) {
            const int64_t mm = m + dy * W8 + dx;
            f0 += e * 8.f * flow[mm * 2];
            f1 += e * 8.f * flow[mm * 2 + 1];
        }
    }
    up[((n * 2 + 0) * HF + Y) * WF + X] = f0 / sum;
    up[((n * 2 + 1) * HF + Y) * WF + X] = f1 / sum;
}

// ---------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------
static GemmDesc conv_desc(int dt, int M, int N, int H, int W, int KH, int KW, int Cin, int split, const void* A, int64_t lda, const void* A2,
                          int64_t lda2, const void* Wt, const float* bias, int epi, int act, void* out, int64_t ldo, const void* zero) {
    GemmDesc d;
    memset(&d, 0, sizeof(d));
    d.dtype = dt; d.M = M; d.N = N; d.K = KH * KW * Cin; d.epi = epi; d.act = act;
    d.A = A; d.lda = lda; d.A2 = A2; d.lda2 = lda2; d.W = Wt; d.ldw = d.K; d.bias = bias; d.out = out; d.ldo = ldo;
    d.conv_H = H; d.conv_W = W; d.conv_KH = KH; d.conv_KW = KW; d.conv_Cin = Cin; d.conv_split = split; d.zero_page = zero;
    return d;
}

// gru_fused.hip: one launch per SepConvGRU half-step (bf16 mode with the hoisted `inp` third)
bool gru_fused_supported(int n_img, int H, int W);
void gru_fused_startmap_bytes(int n_img, int H, int W, int vert, size_t* szr, size_t* sq);
size_t gru_fused_packed_w_bytes(int gate_blocks);
int launch_gru_pack_w(const void* w_rowmajor, void* packed, int gate_blocks, int rot, hipStream_t s);
int launch_gru_startmap(int n_img, int H, int W, int vert, const void* zr_rowmajor, const void* q_rowmajor, void* szr, void* sq, hipStream_t s);
int launch_gru_half(int n_img, int H, int W, int vert, void* hb, void* hlo, const void* X, const void* wzr_packed, const void* wq_packed, const void* szr,
                    const void* sq, hipStream_t s);
static int g_gru_fused = 1;   // (experiments: 0 keeps the two-launch half-step of rounds 1-3)
#ifndef VTGB_LK_FUSED
#define VTGB_LK_FUSED 1
#endif
static int g_lk_fused = VTGB_LK_FUSED;    // (experiments: 0 keeps the separate lookup and convc1 launches; tools/exp/build_variant.sh -DVTGB_LK_FUSED=0)
#ifdef VTGB_DEBUG_HOOKS
extern "C" void vtgb_debug_set_gru_fused(int v) { g_gru_fused = v; }
extern "C" void vtgb_debug_set_lk_fused(int v) { g_lk_fused = v; }
#endif

int raft_x3_impl(const vtgb_raft_update_args* a, Workspace& ws, hipStream_t s);   // raft_x3.hip: VTGB_BF16X3, VTGB_F16C8
static int raft_impl(const vtgb_raft_update_args* a, Workspace& ws, hipStream_t s) {
    VTGB_REQUIRE(a, VTGB_EINVAL, "raft_update: NULL args");
    if (a->dtype == VTGB_BF16X3 || a->dtype == VTGB_F16C8) return raft_x3_impl(a, ws, s);
    VTGB_REQUIRE(a->dtype == VTGB_BF16 || a->dtype == VTGB_F32, VTGB_EINVAL, "raft_update: bad dtype %d", a->dtype);
    VTGB_REQUIRE(a->n_pairs > 0 && a->H8 >= 8 && a->W8 >= 8 && a->iters > 0, VTGB_EINVAL, "raft_update: bad dims n=%d H8=%d W8=%d iters=%d",
                 a->n_pairs, a->H8, a->W8, a->iters);
    const int dt = a->dtype;
    const bool f32 = dt == VTGB_F32;
    const size_t es = dtype_size(dt);
    const int H8 = a->H8, W8 = a->W8, HW = H8 * W8;
    const int64_t M = (int64_t)a->n_pairs * HW;
    VTGB_REQUIRE(M < (1ll << 28), VTGB_EUNSUPPORTED, "raft_update: too many pixels (the flow field is addressed through one 32-bit buffer range)");
    // activations: bf16 (VTGB_BF16) or fp32 (VTGB_F32); typed access through char* + element size
    // bf16 mode with the hoisted `inp` third and whole lines that fit a tile: the fused half-step kernel (hidden state hi | lo in bf16)
    const bool fused = !f32 && g_gru_fused && a->weights && a->weights[26] != nullptr && gru_fused_supported(a->n_pairs, H8, W8);
    float* h32 = fused ? nullptr : (float*)ws.take(M * 128 * 4);
    char* hlo = fused ? (char*)ws.take(M * 128 * es) : nullptr;
    char* hb = (char*)ws.take(M * 128 * es);
    char* X = (char*)ws.take(M * 256 * es);
    const bool lk_fused = !f32 && g_lk_fused;                        // lookup + convc1 in one launch (bf16 mode)
    char* corrf = lk_fused ? nullptr : (char*)ws.take(M * 384 * es);
    void* w1pk = lk_fused ? ws.take(LC_WPK_BYTES) : nullptr;
    char* c1 = (char*)ws.take(M * 256 * es);
    char* CF = (char*)ws.take(M * 256 * es);
    char* f1 = (char*)ws.take(M * 128 * es);
    char* ZR = (char*)ws.take(M * 256 * es);
    char* RH = (char*)ws.take(M * 128 * es);
    char* FH = (char*)ws.take(M * 256 * es);
    float* flow = (float*)ws.take(M * 2 * 4);
    // bf16 mode: the GRU convolutions' contribution of `inp` (channels 128..255 of their 384 inputs: the context features,
    // constant over the refinement iterations) + bias, computed once per call and used as the accumulators' start value:
    // the 80 GRU launches then contract over 256 channels instead of 384 (-1/3 of their MFMA work)
    const bool hoist = !f32 && a->weights && a->weights[26] != nullptr;
    bf16_t* inp_zr[2] = {nullptr, nullptr};
    bf16_t* inp_q[2] = {nullptr, nullptr};
    void *gw_zr[2] = {nullptr, nullptr}, *gw_q[2] = {nullptr, nullptr};      // fused: weights in MFMA fragment order
    if (fused) {
        for (int half = 0; half < 2; half++) {
            size_t b_zr = 0, b_q = 0;
            gru_fused_startmap_bytes(a->n_pairs, H8, W8, half, &b_zr, &b_q);   // fragment order of the fused kernel's tiles (whole lines)
            inp_zr[half] = (bf16_t*)ws.take(b_zr);
            inp_q[half] = (bf16_t*)ws.take(b_q);
            gw_zr[half] = ws.take(gru_fused_packed_w_bytes(4));
            gw_q[half] = ws.take(gru_fused_packed_w_bytes(2));
        }
    } else if (!f32) {
        for (int half = 0; half < 2; half++) {
            const int64_t Mt = (M + 255) / 256 * 256;   // fragment order: whole 256-row tiles
            inp_zr[half] = (bf16_t*)ws.take(Mt * 256 * 2);
            inp_q[half] = (bf16_t*)ws.take(Mt * 128 * 2);
        }
    }
    float* mask = (float*)ws.take(M * 576 * 4);
    float* P2 = mask;   // [M, 32] per-tap partial products of FlowHead.conv2 (the mask buffer is idle until the last iteration)
    void* zero = ws.take(256);
    if (ws.dry) return VTGB_OK;
    VTGB_REQUIRE(ws.ok(), VTGB_EWORKSPACE, "raft_update: workspace %zu < %zu bytes", ws.size, ws.used);
    VTGB_REQUIRE(((a->net && a->inp) || a->cnet_nhwc) && a->weights && a->flow_up, VTGB_EINVAL, "raft_update: NULL operand");
    VTGB_REQUIRE(!(f32 && a->corr_f16), VTGB_EINVAL, "raft_update: the exactness mode takes an fp32 correlation pyramid");
    const void* const* w = a->weights;
    for (int i = 0; i < 26; i++) VTGB_REQUIRE(w[i], VTGB_EINVAL, "raft_update: weights[%d] is NULL", i);
    if (hoist)
        for (int i = 26; i < VTGB_RAFT_NW; i++) VTGB_REQUIRE(w[i], VTGB_EINVAL, "raft_update: weights[%d] is NULL", i);
    CorrPyr pyr;
    int hl = H8, wl = W8;
    for (int l = 0; l < 4; l++) {
        VTGB_REQUIRE(a->corr[l] && hl >= 1 && wl >= 1, VTGB_EINVAL, "raft_update: correlation level %d missing", l);
        pyr.lvl[l] = a->corr[l]; pyr.h[l] = hl; pyr.w[l] = wl;
        hl /= 2; wl /= 2;
    }
    VTGB_HIP(hipMemsetAsync(zero, 0, 256, s));
    const dim3 init_grid((unsigned)((M * 128 + 255) / 256));
    if (f32)
        hipLaunchKernelGGL(raft_init_kernel<float>, init_grid, dim3(256), 0, s, a->net, a->inp, a->cnet_nhwc, h32, (float*)hb, (float*)nullptr, (float*)X, flow, a->flow_init, M, HW);
    else if (a->cnet_nhwc && hlo && (((uintptr_t)a->cnet_nhwc) & 15) == 0)
        hipLaunchKernelGGL(raft_init_nhwc_bf16_kernel, dim3((unsigned)((M * 32 + 255) / 256)), dim3(256), 0, s, a->cnet_nhwc, (bf16_t*)hb, (bf16_t*)hlo, (bf16_t*)X, flow,
                           a->flow_init, M, HW);
    else
        hipLaunchKernelGGL(raft_init_kernel<bf16_t>, init_grid, dim3(256), 0, s, a->net, a->inp, a->cnet_nhwc, h32, (bf16_t*)hb, (bf16_t*)hlo, (bf16_t*)X, flow, a->flow_init, M, HW);
    const int Mi = (int)M;
    const size_t cf1_lds = 128 * CF1_LD * 2 + 4 * 4096;
    if (!f32)
        VTGB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(raft_convf1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cf1_lds));
    const int64_t cf1_groups = (M + 15) / 16, cf1_grid = cf1_groups < 4 * 512 ? (cf1_groups + 3) / 4 : 512;
    auto F = [](const void* p) { return (const float*)p; };
    auto E = [es](char* p, int64_t elems) { return (void*)(p + elems * (int64_t)es); };   // element offset into an activation buffer
    const dim3 lk_grid((unsigned)((M + 4 * CL_PIX - 1) / (4 * CL_PIX)));
    const size_t fh2_lds = (size_t)18 * HW * sizeof(float);
    const bool fh2_in_lds = fh2_lds <= 160 * 1024;      // else: the global-gather form (any image size, as in rounds 1-2)
    if (fh2_in_lds) {
        static DeviceOnce fh2_attr;
        VTGB_FUNC_LDS_ONCE(fh2_attr, raft_flow_head2_kernel, 160 * 1024);
    }
    if (lk_fused) {
        static DeviceOnce lc_attr_h, lc_attr_f;
        VTGB_FUNC_LDS_ONCE(lc_attr_h, raft_lookup_convc1_kernel<_Float16>, LC_LDS);
        VTGB_FUNC_LDS_ONCE(lc_attr_f, raft_lookup_convc1_kernel<float>, LC_LDS);
        hipLaunchKernelGGL(raft_lkc1_pack_w_kernel, dim3((LC_WAVES * LC_KS * 2 * 64 + 255) / 256), dim3(256), 0, s, (const bf16_t*)w[0], (bf16x8*)w1pk);
    }
    if (hoist) {
        // start maps = bias + conv(inp): X[:, 0:128] holds relu(cnet[:, 128:]) (raft_init_kernel); same taps as the GRU halves
        for (int half = 0; half < 2; half++) {
            const int kh = half == 0 ? 1 : 5, kw = half == 0 ? 5 : 1, wi = 10 + 4 * half;
            GemmDesc mz = conv_desc(dt, Mi, 256, H8, W8, kh, kw, 128, 128, X, 256, nullptr, 0, w[26 + 2 * half], F(w[wi + 1]), VTGB_EPI_STORE, 0,
                                    inp_zr[half], 256, zero);
            GemmDesc mq = conv_desc(dt, Mi, 128, H8, W8, kh, kw, 128, 128, X, 256, nullptr, 0, w[27 + 2 * half], F(w[wi + 3]), VTGB_EPI_STORE, 0,
                                    inp_q[half], 128, zero);
            mz.algo_flops = mq.algo_flops = -1.0;   // their work is credited to the 20 per-iteration launches (the reference's form)
            if (fused) {
                // row-major into the (still idle) z|r and r * h buffers, then re-ordered for the fused kernel's tiles; the GRU weights
                // into fragment order (q: chunk order x0, x1, rh0, rh1)
                mz.out = ZR; mz.ldo = 256; mq.out = RH; mq.ldo = 128;
                VTGB_TRY(launch_conv_gemm(mz, s));
                VTGB_TRY(launch_conv_gemm(mq, s));
                VTGB_TRY(launch_gru_startmap(a->n_pairs, H8, W8, half, ZR, RH, inp_zr[half], inp_q[half], s));
                VTGB_TRY(launch_gru_pack_w(w[wi], gw_zr[half], 4, 0, s));
                VTGB_TRY(launch_gru_pack_w(w[wi + 2], gw_q[half], 2, 2, s));
                continue;
            }
            mz.frag_out = mq.frag_out = 1;          // kept as the MFMA leaves them: the GRU launches read them back the same way
            VTGB_TRY(launch_conv_gemm(mz, s));
            VTGB_TRY(launch_conv_gemm(mq, s));
        }
    }
    for (int it = 0; it < a->iters; it++) {
        // ---- BasicMotionEncoder (update.py:88-97)
        if (lk_fused) {
            const dim3 g((unsigned)((M + LC_PX - 1) / LC_PX));
            if (a->corr_f16)
                hipLaunchKernelGGL(raft_lookup_convc1_kernel<_Float16>, g, dim3(512), LC_LDS, s, pyr, flow, (const bf16x8*)w1pk, F(w[1]), (bf16_t*)c1, M, H8, W8);
            else
                hipLaunchKernelGGL(raft_lookup_convc1_kernel<float>, g, dim3(512), LC_LDS, s, pyr, flow, (const bf16x8*)w1pk, F(w[1]), (bf16_t*)c1, M, H8, W8);
        } else {
            if (f32)
                hipLaunchKernelGGL((raft_corr_lookup_kernel<float, float>), lk_grid, dim3(256), 0, s, pyr, flow, (float*)corrf, M, H8, W8);
            else if (a->corr_f16)
                hipLaunchKernelGGL((raft_corr_lookup_kernel<_Float16, bf16_t>), lk_grid, dim3(256), 0, s, pyr, flow, (bf16_t*)corrf, M, H8, W8);
            else
                hipLaunchKernelGGL((raft_corr_lookup_kernel<float, bf16_t>), lk_grid, dim3(256), 0, s, pyr, flow, (bf16_t*)corrf, M, H8, W8);
            GemmDesc d = conv_desc(dt, Mi, 256, H8, W8, 0, 0, 0, 0, corrf, 384, nullptr, 0, w[0], F(w[1]), VTGB_EPI_STORE, 1, c1, 256, zero);
            d.K = 384; d.ldw = 384;
            VTGB_TRY(launch_conv_gemm(d, s));
        }
        VTGB_TRY(launch_conv_gemm(conv_desc(dt, Mi, 192, H8, W8, 3, 3, 256, 256, c1, 256, nullptr, 0, w[2], F(w[3]), VTGB_EPI_STORE, 1, CF, 256, zero), s));
        if (f32)
            hipLaunchKernelGGL(raft_convf1_f32_kernel, dim3((unsigned)((M + 1) / 2)), dim3(256), 0, s, flow, F(w[4]), F(w[5]), (float*)f1, (float*)X, M, H8, W8);
        else
            hipLaunchKernelGGL(raft_convf1_kernel, dim3((unsigned)cf1_grid), dim3(256), cf1_lds, s, flow, (const bf16_t*)w[4], F(w[5]), (bf16_t*)f1, (bf16_t*)X, M, H8, W8);
        VTGB_TRY(launch_conv_gemm(conv_desc(dt, Mi, 64, H8, W8, 3, 3, 128, 128, f1, 128, nullptr, 0, w[6], F(w[7]), VTGB_EPI_STORE, 1, E(CF, 192), 256, zero), s));
        VTGB_TRY(launch_conv_gemm(conv_desc(dt, Mi, 126, H8, W8, 3, 3, 256, 256, CF, 256, nullptr, 0, w[8], F(w[9]), VTGB_EPI_STORE, 1, E(X, 128), 256, zero), s));
        // ---- SepConvGRU (update.py:50-65): horizontal (1x5) then vertical (5x1)
        for (int half = 0; half < 2; half++) {
            const int kh = half == 0 ? 1 : 5, kw = half == 0 ? 5 : 1, wi = 10 + 4 * half;
            if (fused) {
                VTGB_TRY(launch_gru_half(a->n_pairs, H8, W8, half, hb, hlo, X, gw_zr[half], gw_q[half], inp_zr[half], inp_q[half], s));
                continue;
            }
            // z -> ZR[:, :128]; r is multiplied by h in the epilogue and lands in RH (update.py:55,62)
            // input channels [h(128) | inp(128) | motion(126) + flow(2)]; hoisted form: [h | motion + flow] = 256 channels, the
            // second operand starts at column 128 of X, bias and the inp term come from the start map
            GemmDesc zr = hoist ? conv_desc(dt, Mi, 256, H8, W8, kh, kw, 256, 128, hb, 128, E(X, 128), 256, w[wi], nullptr, VTGB_EPI_STORE, 2, ZR, 256, zero)
                                : conv_desc(dt, Mi, 256, H8, W8, kh, kw, 384, 128, hb, 128, X, 256, w[wi], F(w[wi + 1]), VTGB_EPI_STORE, 2, ZR, 256, zero);
            if (hoist) { zr.init_bf16 = inp_zr[half]; zr.ldinit = 256; zr.init_frag = 1; zr.algo_flops = 2.0 * Mi * 256.0 * (5 * 384); }
            zr.gate_from = 128; zr.aux = hb; zr.ldaux = 128; zr.out2 = RH; zr.ldo2 = 128;
            VTGB_TRY(launch_conv_gemm(zr, s));
            GemmDesc q = hoist ? conv_desc(dt, Mi, 128, H8, W8, kh, kw, 256, 128, RH, 128, E(X, 128), 256, w[wi + 2], nullptr, VTGB_EPI_GRU, 0, h32, 128, zero)
                               : conv_desc(dt, Mi, 128, H8, W8, kh, kw, 384, 128, RH, 128, X, 256, w[wi + 2], F(w[wi + 3]), VTGB_EPI_GRU, 0, h32, 128, zero);
            if (hoist) { q.init_bf16 = inp_q[half]; q.ldinit = 128; q.init_frag = 1; q.algo_flops = 2.0 * Mi * 128.0 * (5 * 384); }
            q.resid = h32; q.ldr = 128; q.aux = ZR; q.ldaux = 256; q.out2 = hb; q.ldo2 = 128;
            VTGB_TRY(launch_conv_gemm(q, s));
        }
        // ---- FlowHead (update.py:10-18) and coords1 += delta_flow (xraft.py:145)
        if (f32) {
            VTGB_TRY(launch_conv_gemm(conv_desc(dt, Mi, 256, H8, W8, 3, 3, 128, 128, hb, 128, nullptr, 0, w[18], F(w[19]), VTGB_EPI_STORE, 1, FH, 256, zero), s));
            GemmDesc d = conv_desc(dt, Mi, 32, H8, W8, 0, 0, 0, 0, FH, 256, nullptr, 0, w[20], nullptr, VTGB_EPI_STORE_F32, 0, P2, 32, zero);
            d.K = 256; d.ldw = 256;
            VTGB_TRY(launch_conv_gemm(d, s));
        } else {
            // relu(conv1) never leaves the CU: conv2's 18 per-tap partial products (32 padded columns, weights [32][256]) are
            // formed from the tile while it sits in LDS (GemmDesc::tail_w) -- one launch and a 2 x 1.2 GB round trip less
            GemmDesc d = conv_desc(dt, Mi, 256, H8, W8, 3, 3, 128, 128, hb, 128, nullptr, 0, w[18], F(w[19]), VTGB_EPI_STORE, 1, FH, 256, zero);
            d.tail_w = w[20]; d.tail_out = P2; d.ldtail = 32;
            d.algo_flops = 2.0 * Mi * 256.0 * (9 * 128) + 2.0 * Mi * 32.0 * 256.0;
            VTGB_TRY(launch_conv_gemm(d, s));
        }
        if (fh2_in_lds) hipLaunchKernelGGL(raft_flow_head2_kernel, dim3((unsigned)a->n_pairs), dim3(256), fh2_lds, s, P2, F(w[21]), flow, M, H8, W8);
        else hipLaunchKernelGGL(raft_flow_head2_gather_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, P2, F(w[21]), flow, M, H8, W8);
    }
    // ---- mask head of the last iteration (update.py:129-132,143) and convex upsample (xraft.py:88-99)
    VTGB_TRY(launch_conv_gemm(conv_desc(dt, Mi, 256, H8, W8, 3, 3, 128, 128, hb, 128, nullptr, 0, w[22], F(w[23]), VTGB_EPI_STORE, 1, FH, 256, zero), s));
    {
        GemmDesc d = conv_desc(dt, Mi, 576, H8, W8, 0, 0, 0, 0, FH, 256, nullptr, 0, w[24], F(w[25]), VTGB_EPI_STORE_F32, 0, mask, 576, zero);
        d.K = 256; d.ldw = 256; d.out_scale = 0.25f;
        VTGB_TRY(launch_conv_gemm(d, s));
    }
    const int64_t npx = (int64_t)a->n_pairs * 64 * HW;
    hipLaunchKernelGGL(raft_upsample_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, flow, mask, a->flow_up, (int64_t)a->n_pairs, H8, W8);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---- launchers shared with the bf16x3 orchestration (raft_x3.hip)
int raft_launch_lookup_pair(const CorrPyr& pyr, const float* flow, void* out_pair, int64_t M, int H8, int W8, int h8, hipStream_t s) {
    const dim3 lk_grid((unsigned)((M + 4 * CL_PIX - 1) / (4 * CL_PIX)));
    if (h8) hipLaunchKernelGGL((raft_corr_lookup_kernel<float, bf16_t, 2>), lk_grid, dim3(256), 0, s, pyr, flow, (bf16_t*)out_pair, M, H8, W8);
    else hipLaunchKernelGGL((raft_corr_lookup_kernel<float, bf16_t, 1>), lk_grid, dim3(256), 0, s, pyr, flow, (bf16_t*)out_pair, M, H8, W8);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
int raft_launch_flow_head2(const float* P2, const float* bias, float* flow, int n_pairs, int H8, int W8, hipStream_t s) {
    const int HW = H8 * W8;
    const int64_t M = (int64_t)n_pairs * HW;
    const size_t fh2_lds = (size_t)18 * HW * sizeof(float);
    if (fh2_lds <= 160 * 1024) {
        static DeviceOnce fh2_attr;
        VTGB_FUNC_LDS_ONCE(fh2_attr, raft_flow_head2_kernel, 160 * 1024);
        hipLaunchKernelGGL(raft_flow_head2_kernel, dim3((unsigned)n_pairs), dim3(256), fh2_lds, s, P2, bias, flow, M, H8, W8);
    } else {
        hipLaunchKernelGGL(raft_flow_head2_gather_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, P2, bias, flow, M, H8, W8);
    }
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
int raft_launch_upsample(const float* flow, const float* mask, float* flow_up, int n_pairs, int H8, int W8, hipStream_t s) {
    const int64_t npx = (int64_t)n_pairs * 64 * H8 * W8;
    hipLaunchKernelGGL(raft_upsample_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, flow, mask, flow_up, (int64_t)n_pairs, H8, W8);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
extern "C" size_t vtgb_raft_update_workspace_bytes(const vtgb_raft_update_args* a) {
    Workspace ws(nullptr, 0);
    if (raft_impl(a, ws, nullptr) != VTGB_OK) return 0;
    return align_up(ws.used, 256);
}
extern "C" int vtgb_raft_update(const vtgb_raft_update_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->workspace, VTGB_EWORKSPACE, "raft_update: workspace is NULL");
    Workspace ws(a->workspace, a->workspace_bytes);
    return raft_impl(a, ws, stream);
}
