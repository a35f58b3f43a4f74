// llm.hip -- building blocks of the KV-cached, clip-batched greedy decode step.
// The LLM weights and GEMMs stay third-party (HF Llama weights, hipBLASLt via PyTorch); what is
// fused here are the ~25 small elementwise/reduction launches per layer that otherwise dominate
// a graph-replayed decode step at batch <= 64 (the GEMVs themselves are HBM-bound).
// Semantics follow transformers' modeling_llama op by op, INCLUDING where it rounds to the
// activation dtype (bf16 in the benchmark, fp32 in the parity tests):
//   LlamaRMSNorm: variance in fp32; out = weight * (x_fp32 * rsqrt(var + eps)).to(dtype)
//   residual:     hidden = residual + hidden              (one rounding)
//   rotary:       q*cos + rotate_half(q)*sin              (each product and the sum rounded)
//   MLP:          down(act(gate) * up), act = SiLU        (act and product rounded)
// The current position is read from device memory, so one captured hipGraph serves every step.
#include "common.h"

#include <math.h>

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float to(float v) { return v; }
    static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
    static __device__ __forceinline__ bf16_t to(float v) { return (bf16_t)v; }
    static __device__ __forceinline__ float rnd(float v) { return (float)(bf16_t)v; }
};

// ---- x (+= delta) ; h = rmsnorm(x) * w.   One workgroup per row.
template <typename T>
__global__ __launch_bounds__(256) void llm_rmsnorm_kernel(T* __restrict__ x, const T* __restrict__ delta, const T* __restrict__ w,
                                                          T* __restrict__ h, int H, float eps) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    T* xr = x + row * H;
    const T* dr = delta ? delta + row * H : nullptr;
    float ss = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) {
        float v = (float)xr[i];
        if (dr) {
            v = Cvt<T>::rnd(v + (float)dr[i]);
            xr[i] = Cvt<T>::to(v);
        }
        ss += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float var = (red[0] + red[1] + red[2] + red[3]) / (float)H;
    const float rs = rsqrtf(var + eps);
    for (int i = threadIdx.x; i < H; i += 256) {
        const float n = Cvt<T>::rnd((float)xr[i] * rs);
        h[row * H + i] = Cvt<T>::to((float)w[i] * n);
    }
}

// ---- rotary on q and k at position *pos, k and v appended to the cache.  One workgroup per (batch, head).
template <typename T>
__global__ __launch_bounds__(128) void llm_rope_cache_kernel(const T* __restrict__ qkv, T* __restrict__ q_out, T* __restrict__ kc,
                                                             T* __restrict__ vc, const T* __restrict__ cos_t, const T* __restrict__ sin_t,
                                                             const int64_t* __restrict__ pos_p, int nq, int nkv, int hd, int tmax) {
    const int b = blockIdx.y, head = blockIdx.x;   // head in [0, nq + 2 nkv)
    const int64_t pos = *pos_p;
    const T* src = qkv + ((int64_t)b * (nq + 2 * nkv) + head) * hd;
    const int half = hd >> 1;
    for (int d = threadIdx.x; d < hd; d += blockDim.x) {
        const float v = (float)src[d];
        float outv = v;
        if (head < nq + nkv) {
            const float c = (float)cos_t[pos * hd + d], sn = (float)sin_t[pos * hd + d];
            const float rot = d < half ? -(float)src[d + half] : (float)src[d - half];
            outv = Cvt<T>::rnd(Cvt<T>::rnd(v * c) + Cvt<T>::rnd(rot * sn));
        }
        if (head < nq) q_out[((int64_t)b * nq + head) * hd + d] = Cvt<T>::to(outv);
        else if (head < nq + nkv) kc[(((int64_t)b * nkv + (head - nq)) * tmax + pos) * hd + d] = Cvt<T>::to(outv);
        else vc[(((int64_t)b * nkv + (head - nq - nkv)) * tmax + pos) * hd + d] = Cvt<T>::to(outv);
    }
}

// ---- single-query attention over the cache rows [0, *pos].  One wave per (batch, q head), 4 per workgroup.
constexpr int DEC_MAX_T = 2048;
template <typename T>
__global__ __launch_bounds__(256) void llm_decode_attn_kernel(const T* __restrict__ q, const T* __restrict__ kc, const T* __restrict__ vc,
                                                              T* __restrict__ out, const int64_t* __restrict__ pos_p, int B, int nq, int nkv,
                                                              int hd, int tmax, float scale) {
    extern __shared__ float dsm[];   // per wave: hd floats of q + (tmax) scores
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t idx = (int64_t)blockIdx.x * 4 + wave;
    const bool active = idx < (int64_t)B * nq;
    const int64_t id = active ? idx : (int64_t)B * nq - 1;
    const int b = id / nq, head = id % nq, kvh = head / (nq / nkv);
    const int n_keys = (int)(*pos_p) + 1;
    float* qs = dsm + wave * (hd + tmax);
    float* sc = qs + hd;
    const T* qr = q + ((int64_t)b * nq + head) * hd;
    const T* kr = kc + ((int64_t)b * nkv + kvh) * tmax * hd;
    const T* vr = vc + ((int64_t)b * nkv + kvh) * tmax * hd;
    for (int d = lane; d < hd; d += 64) qs[d] = (float)qr[d];
    __syncthreads();
    float mx = -INFINITY;
    for (int key = lane; key < n_keys; key += 64) {
        const T* k = kr + (int64_t)key * hd;
        float dot = 0.f;
        if constexpr (sizeof(T) == 2) {
            // a lane owns a key row: 16-byte loads (8 elements) instead of 2-byte ones, same summation order
            if ((hd & 7) == 0) {
                for (int d = 0; d < hd; d += 8) {
                    const bf16x8 kv = *reinterpret_cast<const bf16x8*>(k + d);
#pragma unroll
                    for (int e = 0; e < 8; e++) dot = fmaf(qs[d + e], (float)kv[e], dot);
                }
            } else {
                for (int d = 0; d < hd; d++) dot = fmaf(qs[d], (float)k[d], dot);
            }
        } else {
            for (int d = 0; d < hd; d++) dot = fmaf(qs[d], (float)k[d], dot);
        }
        dot *= scale;
        sc[key] = dot;
        mx = fmaxf(mx, dot);
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
    for (int key = lane; key < n_keys; key += 64) {
        const float e = expf(sc[key] - mx);
        sc[key] = e;
        sum += e;
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    __syncthreads();
    const float inv = 1.0f / sum;
    for (int d = lane; d < hd; d += 64) {
        float acc = 0.f;
        for (int key = 0; key < n_keys; key++) acc = fmaf(sc[key], (float)vr[(int64_t)key * hd + d], acc);
        if (active) out[((int64_t)b * nq + head) * hd + d] = Cvt<T>::to(acc * inv);
    }
}

// ---- act[b, i] = silu(gu[b, i]) * gu[b, I + i]
template <typename T>
__global__ void llm_silu_mul_kernel(const T* __restrict__ gu, T* __restrict__ act, int64_t rows, int I) {
    const int64_t n = rows * I;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / I, c = i - r * I;
        const float g = (float)gu[r * 2 * I + c], u = (float)gu[r * 2 * I + I + c];
        const float s = Cvt<T>::rnd(g / (1.0f + expf(-g)));
        act[i] = Cvt<T>::to(s * u);
    }
}

extern "C" int vtgb_llm_rmsnorm(int dtype, void* x, const void* delta, const void* w, void* h, int64_t rows, int32_t H, float eps,
                                vtgb_stream_t s) {
    VTGB_REQUIRE(x && w && h && rows > 0 && H > 0, VTGB_EINVAL, "llm_rmsnorm: bad argument");
    if (dtype == VTGB_BF16)
        hipLaunchKernelGGL(llm_rmsnorm_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)delta, (const bf16_t*)w, (bf16_t*)h, H, eps);
    else
        hipLaunchKernelGGL(llm_rmsnorm_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, (float*)x, (const float*)delta, (const float*)w, (float*)h, H, eps);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

extern "C" int vtgb_llm_rope_cache(int dtype, const void* qkv, void* q_out, void* kc, void* vc, const void* cos_t, const void* sin_t,
                                   const int64_t* pos, int32_t B, int32_t nq, int32_t nkv, int32_t hd, int32_t tmax, vtgb_stream_t s) {
    VTGB_REQUIRE(qkv && q_out && kc && vc && cos_t && sin_t && pos && B > 0 && nq > 0 && nkv > 0 && (hd % 2) == 0, VTGB_EINVAL, "llm_rope_cache: bad argument");
    const dim3 grid(nq + 2 * nkv, B);
    if (dtype == VTGB_BF16)
        hipLaunchKernelGGL(llm_rope_cache_kernel<bf16_t>, grid, dim3(128), 0, s, (const bf16_t*)qkv, (bf16_t*)q_out, (bf16_t*)kc, (bf16_t*)vc,
                           (const bf16_t*)cos_t, (const bf16_t*)sin_t, pos, nq, nkv, hd, tmax);
    else
        hipLaunchKernelGGL(llm_rope_cache_kernel<float>, grid, dim3(128), 0, s, (const float*)qkv, (float*)q_out, (float*)kc, (float*)vc,
                           (const float*)cos_t, (const float*)sin_t, pos, nq, nkv, hd, tmax);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

extern "C" int vtgb_llm_decode_attention(int dtype, const void* q, const void* kc, const void* vc, void* out, const int64_t* pos, int32_t B,
                                         int32_t nq, int32_t nkv, int32_t hd, int32_t tmax, float scale, vtgb_stream_t s) {
    VTGB_REQUIRE(q && kc && vc && out && pos && B > 0 && nq > 0 && nkv > 0 && nq % nkv == 0, VTGB_EINVAL, "llm_decode_attention: bad argument");
    VTGB_REQUIRE(tmax <= DEC_MAX_T && hd <= 256, VTGB_EUNSUPPORTED, "llm_decode_attention: tmax=%d hd=%d too large", tmax, hd);
    const size_t lds = 4 * (size_t)(hd + tmax) * sizeof(float);
    const dim3 grid((unsigned)(((int64_t)B * nq + 3) / 4));
    if (dtype == VTGB_BF16)
        hipLaunchKernelGGL(llm_decode_attn_kernel<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)kc, (const bf16_t*)vc, (bf16_t*)out,
                           pos, B, nq, nkv, hd, tmax, scale);
    else
        hipLaunchKernelGGL(llm_decode_attn_kernel<float>, grid, dim3(256), lds, s, (const float*)q, (const float*)kc, (const float*)vc, (float*)out, pos,
                           B, nq, nkv, hd, tmax, scale);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

extern "C" int vtgb_llm_silu_mul(int dtype, const void* gu, void* act, int64_t rows, int32_t I, vtgb_stream_t s) {
    VTGB_REQUIRE(gu && act && rows > 0 && I > 0, VTGB_EINVAL, "llm_silu_mul: bad argument");
    int64_t blocks = (rows * I + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == VTGB_BF16)
        hipLaunchKernelGGL(llm_silu_mul_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)gu, (bf16_t*)act, rows, I);
    else
        hipLaunchKernelGGL(llm_silu_mul_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)gu, (float*)act, rows, I);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---------------------------------------------------------------------------------------
// Skinny GEMM for the decode step: out[M, N] = x[M, K] . w[N, K]^T with M <= 128 rows (one token per clip) -- the weight
// matrix is read exactly once, so the kernel is an HBM stream of W with the matrix cores idling behind it.
//   work      one workgroup per (128-row weight tile, K split): K is split where N / 128 tiles alone would leave most of the 256
//             CUs without a stream (N = 4096: 32 tiles).  Every workgroup leaves an fp32 fragment of its tile; a second tiny
//             launch adds a tile's fragments in a FIXED order and rounds once (deterministic: no atomics).  (Equal runs of steps
//             per CU across tile boundaries -- "stream-K" -- were built and measured slower: two 63 KiB fragments per
//             workgroup at M = 124 cost more than the balance gains; round 2.)
//   tile      128 (all of M) x 128 weight rows, 4 waves of 64 x 64, v_mfma_f32_16x16x32_bf16
//   staging   buffer_load ... lds (LDS-DMA) into TWO rings: 7 weight slots (6 k-tiles = 96 KiB of weights in flight) and 3
//             activation slots (L2-resident, two ahead) = 160 KiB.  vmcnt completes in order PER WAVE, so a wave that mixed the
//             deep weight stream with the shallow activation stream would drain the weights every k-tile: waves 0-1 load only
//             weights, waves 2-3 only activations (each with its own counted wait); all four multiply.  One barrier per step.
// x rows beyond M are clamped duplicates (never stored); weight rows beyond N read as zeros (descriptor range / packed zeros).
// ---------------------------------------------------------------------------------------
constexpr int SK_BN = 128, SK_BK = 64, SK_WSLOTS = 7, SK_XSLOTS = 3, SK_TILE = 128 * 128;   // bytes of one operand tile (128 rows x 64 bf16)
typedef __attribute__((address_space(3))) void* sk_lptr_t;

__device__ __forceinline__ int sk_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// s_waitcnt vmcnt(8 n), n = 0..5 (vmcnt is 6 bits: [3:0] and [15:14]); the other counters are left alone
__device__ __forceinline__ void sk_wait_stages(int n) {
    if (n >= 5) __builtin_amdgcn_s_waitcnt(0x8F78);
    else if (n == 4) __builtin_amdgcn_s_waitcnt(0x8F70);
    else if (n == 3) __builtin_amdgcn_s_waitcnt(0x4F78);
    else if (n == 2) __builtin_amdgcn_s_waitcnt(0x4F70);
    else if (n == 1) __builtin_amdgcn_s_waitcnt(0x0F78);
    else __builtin_amdgcn_s_waitcnt(0x0F70);
}

// part: [gridDim.x tiles][gridDim.y splits][M][128] fp32
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const bf16_t* __restrict__ x, int M, int ldx, const bf16_t* __restrict__ w, int N, int ldw,
                                                          int nk, float* __restrict__ part, int w_tiled) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    char* const w_ring = sk_smem;                              // SK_WSLOTS tiles
    char* const x_ring = sk_smem + SK_WSLOTS * SK_TILE;        // SK_XSLOTS tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const bool w_loader = wave < 2;
    const int lw = wave & 1;                                   // which half (64 rows) of its operand tile this wave stages
    const int b = blockIdx.x, S = gridDim.y, sp = blockIdx.y;
    const int s0 = b * nk + (int)((int64_t)nk * sp / S), s1 = b * nk + (int)((int64_t)nk * (sp + 1) / S), ns = s1 - s0;   // steps = global k-tile index
    if (ns <= 0) return;
    // one descriptor per wave (its operand's base); per-lane byte offsets are loop invariant, everything else is scalar.
    // Row-major weights: range = the matrix, so the rows of the last tile beyond N read as zeros.
    const bool tiled = w_loader && w_tiled;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(w_loader ? w : x), 0,
                                                        w_loader && !w_tiled ? (int)(((int64_t)(N - 1) * ldw + nk * SK_BK) * 2) : 0x7FFFFF00, 0x00020000);
    unsigned voff[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int row = lw * 64 + i * 8 + (lane >> 3), slot = lane & 7, c = slot ^ ((row >> 1) & 7);
        const int r = w_loader ? row : (row < M ? row : M - 1);
        voff[i] = tiled ? (unsigned)((lw * 64 + i * 8) * 128 + lane * 16) : (unsigned)(r * (w_loader ? ldw : ldx) + c * 8) * 2u;
    }
    char* const ring = w_loader ? w_ring : x_ring;
    // scalar byte offset of step st: tiled weights -- the st-th 16 KiB block; row-major weights -- (tile * 128 rows, kt * 64);
    // activations -- kt * 64
#define SK_ISSUE(slot, st)                                                                                   \
    {                                                                                                        \
        const int b_ = (st) / nk, kt_ = (st) - b_ * nk;                                                      \
        const int so_ = tiled ? (st) * SK_TILE : w_loader ? (b_ * SK_BN * ldw + kt_ * SK_BK) * 2 : kt_ * (SK_BK * 2); \
        _Pragma("unroll") for (int i = 0; i < 8; i++)                                                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (sk_lptr_t)(ring + (slot) * SK_TILE + (lw * 64 + i * 8) * 128), 16, voff[i], so_, 0, 0); \
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int depth = w_loader ? SK_WSLOTS - 1 : SK_XSLOTS - 1, slots = w_loader ? SK_WSLOTS : SK_XSLOTS;   // steps ahead; ring size
    for (int t = 0; t < depth && t < ns; t++) { SK_ISSUE(t, s0 + t) }
    const int fr = lane & 15, fg = lane >> 4;
    int slot_w = 0, slot_x = 0, slot_in = depth % slots;       // slot_in: where this wave's next step goes
    for (int t = 0; t < ns; t++) {
        // my pieces of step t have landed when only the younger steps' (8 pieces each) are outstanding
        const int younger = ns - 1 - t < depth - 1 ? ns - 1 - t : depth - 1;
        sk_wait_stages(younger);
        __builtin_amdgcn_s_barrier();                          // both operands of step t are in LDS; everyone is done with step t - 1
        if (t + depth < ns) { SK_ISSUE(slot_in, s0 + t + depth) }
        slot_in = slot_in + 1 == slots ? 0 : slot_in + 1;
        const char* ws = w_ring + slot_w * SK_TILE;
        const char* xs = x_ring + slot_x * SK_TILE;
        slot_w = slot_w + 1 == SK_WSLOTS ? 0 : slot_w + 1;
        slot_x = slot_x + 1 == SK_XSLOTS ? 0 : slot_x + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                wf[i] = *reinterpret_cast<const bf16x8*>(ws + sk_swz(wn * 64 + i * 16 + fr, ks * 4 + fg));
                xf[i] = *reinterpret_cast<const bf16x8*>(xs + sk_swz(wm * 64 + i * 16 + fr, ks * 4 + fg));
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    }
    // this workgroup's fragment.  D layout: column (lane & 15) <- x row (m), rows (lane >> 4) * 4 + reg <- w row (n)
    float* const ps = part + (int64_t)(b * S + sp) * M * SK_BN;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int m = wm * 64 + j * 16 + fr;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (m < M) *reinterpret_cast<f32x4*>(ps + m * SK_BN + wn * 64 + i * 16 + fg * 4) = acc[i][j];
    }
#undef SK_ISSUE
#endif
}

// One-time weight preparation for the tiled layout: dst[tile b][k-tile kt][row r][16-byte slot q] = src[b * 128 + r][kt * 64 + 8 (q ^ ((r >> 1) & 7)) ...],
// rows beyond N zero.  One thread per 16-byte chunk.
__global__ __launch_bounds__(256) void skinny_pack_kernel(const bf16_t* __restrict__ src, int64_t ld, int N, int nk, bf16_t* __restrict__ dst, int64_t chunks) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= chunks) return;
    const int q = (int)(i & 7), r = (int)((i >> 3) & 127);
    const int64_t blk = i >> 10, b = blk / nk, kt = blk - b * nk;
    const int64_t n = b * 128 + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n < N) v = *reinterpret_cast<const uint4*>(src + n * ld + kt * 64 + 8 * (q ^ ((r >> 1) & 7)));
    *reinterpret_cast<uint4*>(dst + i * 8) = v;
}

extern "C" size_t vtgb_pack_skinny_weight_bytes(int32_t N, int32_t K) {
    if (N <= 0 || K <= 0 || (K % SK_BK) != 0) return 0;
    return (size_t)((N + SK_BN - 1) / SK_BN) * (K / SK_BK) * SK_TILE;
}

extern "C" int vtgb_pack_skinny_weight(const void* w, int64_t ldw, int32_t N, int32_t K, void* dst, vtgb_stream_t s) {
    VTGB_REQUIRE(w && dst && N > 0 && K > 0 && (K % SK_BK) == 0 && (ldw % 8) == 0 && ldw >= K, VTGB_EINVAL, "pack_skinny_weight: N=%d K=%d ldw=%lld", N, K,
                 (long long)ldw);
    const int64_t chunks = (int64_t)vtgb_pack_skinny_weight_bytes(N, K) / 16;
    hipLaunchKernelGGL(skinny_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, (const bf16_t*)w, ldw, N, K / SK_BK, (bf16_t*)dst, chunks);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// out[m][b * 128 + c] = sum over the splits, in split order, of tile b's fragments
template <typename T>
__global__ __launch_bounds__(256) void gemm_skinny_reduce_kernel(const float* __restrict__ part, int M, int N, int S, T* __restrict__ out, int64_t ldo) {
    const int b = blockIdx.x, c4 = (threadIdx.x & 31) * 4, n = b * SK_BN + c4;
    if (n >= N) return;
    for (int m = blockIdx.y * 8 + (threadIdx.x >> 5); m < M; m += gridDim.y * 8) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < S; sp++) v += *reinterpret_cast<const f32x4*>(part + ((int64_t)(b * S + sp) * M + m) * SK_BN + c4);
        T* o = out + m * ldo + n;
        if (n + 3 < N) {
            if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x4*>(o) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            else *reinterpret_cast<f32x4*>(o) = v;
        } else {
            for (int e = 0; e < 4 && n + e < N; e++) o[e] = (T)v[e];
        }
    }
}

// splits: every CU should hold a stream, but a split costs an fp32 fragment (M x 128 x 4 bytes written and read back) and a
// shorter pipeline: the smallest count in 1..8 that reaches 192 workgroups, never below 8 k-tiles per split
static int skinny_splits(const vtgb_gemm_skinny_args* a) {
    const int nk = a->K / SK_BK, n_tiles = (a->N + SK_BN - 1) / SK_BN;
    if (a->n_splits > 0) return a->n_splits < nk ? a->n_splits : nk;
    int S = 1;
    while (S < 8 && n_tiles * S < 192 && nk / (S + 1) >= 8) S++;
    return S;
}

static int skinny_check(const vtgb_gemm_skinny_args* a) {
    VTGB_REQUIRE(a, VTGB_EINVAL, "gemm_skinny: NULL args");
    VTGB_REQUIRE(a->M > 0 && a->M <= 128 && a->N > 0 && a->K > 0 && (a->K % SK_BK) == 0, VTGB_EUNSUPPORTED,
                 "gemm_skinny: M=%d (<= 128), N=%d, K=%d (multiple of 64)", a->M, a->N, a->K);
    VTGB_REQUIRE((a->ldx % 8) == 0 && a->ldx >= a->K && a->ldo >= a->N && (int64_t)128 * a->ldx * 2 < 0x7FFFFF00ll &&
                     (a->w_tiled ? (int64_t)((a->N + 127) / 128) * 128 * a->K * 2 < 0x7FFFFF00ll
                                 : ((a->ldw % 8) == 0 && a->ldw >= a->K && (int64_t)((a->N + 127) / 128) * 128 * a->ldw * 2 < 0x7FFFFF00ll)),
                 VTGB_EINVAL, "gemm_skinny: row pitches ldx=%lld ldw=%lld ldo=%lld (operands must stay below 2 GiB)", (long long)a->ldx, (long long)a->ldw,
                 (long long)a->ldo);
    VTGB_REQUIRE(a->out_dtype == VTGB_BF16 || a->out_dtype == VTGB_F32, VTGB_EINVAL, "gemm_skinny: bad out_dtype %d", a->out_dtype);
    VTGB_REQUIRE(a->n_splits >= 0 && a->n_splits <= 64, VTGB_EINVAL, "gemm_skinny: n_splits=%d", a->n_splits);
    return VTGB_OK;
}

extern "C" size_t vtgb_gemm_skinny_workspace_bytes(const vtgb_gemm_skinny_args* a) {
    if (skinny_check(a) != VTGB_OK) return 0;
    return (size_t)((a->N + SK_BN - 1) / SK_BN) * skinny_splits(a) * a->M * SK_BN * sizeof(float);
}

extern "C" int vtgb_gemm_skinny(const vtgb_gemm_skinny_args* a, vtgb_stream_t s) {
    VTGB_TRY(skinny_check(a));
    VTGB_REQUIRE(a->x && a->w && a->out && a->workspace, VTGB_EINVAL, "gemm_skinny: NULL operand");
    const int nk = a->K / SK_BK, n_tiles = (a->N + SK_BN - 1) / SK_BN, S = skinny_splits(a);
    const size_t need = (size_t)n_tiles * S * a->M * SK_BN * sizeof(float);
    VTGB_REQUIRE(a->workspace_bytes >= need, VTGB_EWORKSPACE, "gemm_skinny: workspace %zu < %zu bytes", a->workspace_bytes, need);
    constexpr int LDS = (SK_WSLOTS + SK_XSLOTS) * SK_TILE;
    static DeviceOnce attr;
    VTGB_FUNC_LDS_ONCE(attr, gemm_skinny_kernel, LDS);
    {
        ProfScope prof(VTGB_PROF_GEMM, 2.0 * a->M * a->N * a->K, s);
        hipLaunchKernelGGL(gemm_skinny_kernel, dim3(n_tiles, S), dim3(256), LDS, s, (const bf16_t*)a->x, a->M, (int)a->ldx, (const bf16_t*)a->w, a->N,
                           (int)a->ldw, nk, (float*)a->workspace, a->w_tiled);
    }
    const dim3 rgrid(n_tiles, (a->M + 7) / 8 < 4 ? (a->M + 7) / 8 : 4);
    if (a->out_dtype == VTGB_BF16)
        hipLaunchKernelGGL(gemm_skinny_reduce_kernel<bf16_t>, rgrid, dim3(256), 0, s, (const float*)a->workspace, a->M, a->N, S, (bf16_t*)a->out, a->ldo);
    else
        hipLaunchKernelGGL(gemm_skinny_reduce_kernel<float>, rgrid, dim3(256), 0, s, (const float*)a->workspace, a->M, a->N, S, (float*)a->out, a->ldo);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
