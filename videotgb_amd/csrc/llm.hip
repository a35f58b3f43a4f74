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
    // round-to-nearest-even to bf16 precision, by hand: hipcc may keep the excess precision of a float -> __bf16 -> float round trip (it
    // did in the rotary kernels: one product stayed unrounded and fused into an FMA, 1 ulp off HF's bf16 arithmetic in 7 % of the values)
    static __device__ __forceinline__ float rnd(float v) {
        unsigned u = __float_as_uint(v);
        if ((u & 0x7F800000u) != 0x7F800000u) u += 0x7FFFu + ((u >> 16) & 1u);      // (Inf / NaN pass through)
        return __uint_as_float(u & 0xFFFF0000u);
    }
};

// The decode step's split-K projections leave fp32 fragments part[(tile b * S + split) * M + row][128] (gemm_skinny_kernel); their consumers add a
// tile's fragments in split order and round once to the activation type themselves (r5: the separate reduce launch -- three per layer -- is gone:
// same sums, same rounding, 96 launches fewer per token).  V consecutive columns starting at column i (i % V == 0, V <= 8) of row `row`:
template <typename T, int V>
__device__ __forceinline__ void parts_sum(const float* __restrict__ part, int S, int M, int64_t row, int i, float (&out)[V]) {
    const int b = i >> 7, c = i & 127;
#pragma unroll
    for (int e = 0; e < V; e++) out[e] = 0.f;
    for (int sp = 0; sp < S; sp++) {
        const float* p = part + ((int64_t)(b * S + sp) * M + row) * 128 + c;
#pragma unroll
        for (int q = 0; q < V; q += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + q);
#pragma unroll
            for (int e = 0; e < 4; e++) out[q + e] += t[e];
        }
    }
#pragma unroll
    for (int e = 0; e < V; e++) out[e] = Cvt<T>::rnd(out[e]);
}

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

// The same with the row held in registers between the two passes (r4: the scalar loop above took 17 us per launch at 124 x 4096 -- sixteen
// dependent 2-byte round trips per pass; 65 launches per decoded token).  NV 16-byte vectors per thread: H == 256 * NV * (16 / sizeof(T)).
template <typename T, int NV>
__global__ __launch_bounds__(256) void llm_rmsnorm_vec_kernel(T* __restrict__ x, const T* __restrict__ delta, const T* __restrict__ w,
                                                              T* __restrict__ h, int H, float eps, const float* __restrict__ part = nullptr, int S = 0,
                                                              int M = 0) {
    constexpr int V = 16 / (int)sizeof(T);
    typedef T TV __attribute__((ext_vector_type(V)));
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    T* xr = x + row * H;
    float v[NV][V];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NV; c++) {
        const int i = (c * 256 + threadIdx.x) * V;
        const TV xv = *reinterpret_cast<const TV*>(xr + i);
#pragma unroll
        for (int e = 0; e < V; e++) v[c][e] = (float)xv[e];
        if (part) {            // delta = the sum of the K-split fragments, rounded once (what the reduce launch stored)
            float dv[V];
            parts_sum<T, V>(part, S, M, row, i, dv);
            TV o;
#pragma unroll
            for (int e = 0; e < V; e++) {
                v[c][e] = Cvt<T>::rnd(v[c][e] + dv[e]);
                o[e] = Cvt<T>::to(v[c][e]);
            }
            *reinterpret_cast<TV*>(xr + i) = o;
        } else if (delta) {
            const TV dv = *reinterpret_cast<const TV*>(delta + row * H + i);
            TV o;
#pragma unroll
            for (int e = 0; e < V; e++) {
                v[c][e] = Cvt<T>::rnd(v[c][e] + (float)dv[e]);
                o[e] = Cvt<T>::to(v[c][e]);
            }
            *reinterpret_cast<TV*>(xr + i) = o;
        }
#pragma unroll
        for (int e = 0; e < V; e++) ss += v[c][e] * v[c][e];
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float var = (red[0] + red[1] + red[2] + red[3]) / (float)H;
    const float rs = rsqrtf(var + eps);
#pragma unroll
    for (int c = 0; c < NV; c++) {
        const int i = (c * 256 + threadIdx.x) * V;
        const TV wv = *reinterpret_cast<const TV*>(w + i);
        TV o;
#pragma unroll
        for (int e = 0; e < V; e++) o[e] = Cvt<T>::to((float)wv[e] * Cvt<T>::rnd(v[c][e] * rs));
        *reinterpret_cast<TV*>(h + row * H + i) = o;
    }
}

// ---- rotary on q and k at position *pos, k and v appended to the cache.  One workgroup per (batch, head).
template <typename T>
__global__ __launch_bounds__(128) void llm_rope_cache_kernel(const T* __restrict__ qkv, T* __restrict__ q_out, T* __restrict__ kc,
                                                             T* __restrict__ vc, const T* __restrict__ cos_t, const T* __restrict__ sin_t,
                                                             const int64_t* __restrict__ pos_p, int nq, int nkv, int hd, int tmax,
                                                             const float* __restrict__ part = nullptr, int S = 0, int M = 0) {
#pragma clang fp contract(off)      // fp32: mul, mul, add are three roundings in the reference -- no FMA
    const int b = blockIdx.y, head = blockIdx.x;   // head in [0, nq + 2 nkv)
    const int64_t pos = *pos_p;
    const T* src = qkv + ((int64_t)b * (nq + 2 * nkv) + head) * hd;
    const int half = hd >> 1;
    // part != NULL: qkv is still the projection's K-split fragments -- element (b, col) = the fragments' sum in split order, rounded once
    auto at = [&](int d) -> float {
        if (!part) return (float)src[d];
        const int col = head * hd + d;
        float t = 0.f;
        for (int sp = 0; sp < S; sp++) t += part[((int64_t)((col >> 7) * S + sp) * M + b) * 128 + (col & 127)];
        return Cvt<T>::rnd(t);
    };
    for (int d = threadIdx.x; d < hd; d += blockDim.x) {
        const float v = at(d);
        float outv = v;
        if (head < nq + nkv && cos_t) {      // (cos_t == NULL: no rotary -- T5's decoder: q passed through, k / v appended)
            const float c = (float)cos_t[pos * hd + d], sn = (float)sin_t[pos * hd + d];
            const float rot = d < half ? -at(d + half) : at(d - half);
            const float pa = Cvt<T>::rnd(v * c), pb = Cvt<T>::rnd(rot * sn);      // (contract(off): the reference's three roundings, no FMA)
            outv = Cvt<T>::rnd(pa + pb);
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
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (uniform: batch / head / cache bases on the scalar unit)
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
    if constexpr (sizeof(T) == 2) {
        if ((hd & 1) == 0) {
            // a lane owns two adjacent channels: one 4-byte load per key (a key row = one coalesced 2 hd-byte read), sixteen keys in flight
            // (r4: four in flight made the ~68 keys of the bench seventeen dependent memory round trips: 38 us per launch), same summation order
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            for (int d = lane * 2; d < hd; d += 128) {
                float a0 = 0.f, a1 = 0.f;
                int key = 0;
                for (; key + 16 <= n_keys; key += 16) {
                    bf16x2_t v[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) v[u] = *reinterpret_cast<const bf16x2_t*>(vr + (int64_t)(key + u) * hd + d);
#pragma unroll
                    for (int u = 0; u < 16; u++) { a0 = fmaf(sc[key + u], (float)v[u][0], a0); a1 = fmaf(sc[key + u], (float)v[u][1], a1); }
                }
                for (; key + 4 <= n_keys; key += 4) {
                    bf16x2_t v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const bf16x2_t*>(vr + (int64_t)(key + u) * hd + d);
#pragma unroll
                    for (int u = 0; u < 4; u++) { a0 = fmaf(sc[key + u], (float)v[u][0], a0); a1 = fmaf(sc[key + u], (float)v[u][1], a1); }
                }
                for (; key < n_keys; key++) {
                    const bf16x2_t v = *reinterpret_cast<const bf16x2_t*>(vr + (int64_t)key * hd + d);
                    a0 = fmaf(sc[key], (float)v[0], a0); a1 = fmaf(sc[key], (float)v[1], a1);
                }
                if (active) *reinterpret_cast<bf16x2_t*>(out + ((int64_t)b * nq + head) * hd + d) = bf16x2_t{(bf16_t)(a0 * inv), (bf16_t)(a1 * inv)};
            }
            return;
        }
    }
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

// one 16-byte vector per thread, row = blockIdx.y (r4: the scalar form above divides a 64-bit index per element: 14 us for 124 x 11008)
template <typename T>
__global__ __launch_bounds__(256) void llm_silu_mul_vec_kernel(const T* __restrict__ gu, T* __restrict__ act, int I) {
    constexpr int V = 16 / (int)sizeof(T);
    typedef T TV __attribute__((ext_vector_type(V)));
    const int c = (blockIdx.x * 256 + threadIdx.x) * V;
    if (c >= I) return;
    const int64_t r = blockIdx.y;
    const TV g = *reinterpret_cast<const TV*>(gu + r * 2 * I + c), u = *reinterpret_cast<const TV*>(gu + r * 2 * I + I + c);
    TV o;
#pragma unroll
    for (int e = 0; e < V; e++) {
        const float gf = (float)g[e];
        o[e] = Cvt<T>::to(Cvt<T>::rnd(gf / (1.0f + expf(-gf))) * (float)u[e]);
    }
    *reinterpret_cast<TV*>(act + r * I + c) = o;
}

extern "C" int vtgb_llm_rmsnorm(int dtype, void* x, const void* delta, const void* w, void* h, int64_t rows, int32_t H, float eps,
                                vtgb_stream_t s) {
    VTGB_REQUIRE(x && w && h && rows > 0 && H > 0, VTGB_EINVAL, "llm_rmsnorm: bad argument");
    const bool al = (((uintptr_t)x | (uintptr_t)delta | (uintptr_t)w | (uintptr_t)h) & 15) == 0;
    if (dtype == VTGB_BF16 && al && H == 256 * 8 * 2)
        hipLaunchKernelGGL((llm_rmsnorm_vec_kernel<bf16_t, 2>), dim3((unsigned)rows), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)delta, (const bf16_t*)w, (bf16_t*)h, H, eps);
    else if (dtype == VTGB_BF16 && al && H == 256 * 8)
        hipLaunchKernelGGL((llm_rmsnorm_vec_kernel<bf16_t, 1>), dim3((unsigned)rows), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)delta, (const bf16_t*)w, (bf16_t*)h, H, eps);
    else if (dtype == VTGB_F32 && al && H == 256 * 4 * 4)
        hipLaunchKernelGGL((llm_rmsnorm_vec_kernel<float, 4>), dim3((unsigned)rows), dim3(256), 0, s, (float*)x, (const float*)delta, (const float*)w, (float*)h, H, eps);
    else if (dtype == VTGB_F32 && al && H == 256 * 4 * 2)
        hipLaunchKernelGGL((llm_rmsnorm_vec_kernel<float, 2>), dim3((unsigned)rows), dim3(256), 0, s, (float*)x, (const float*)delta, (const float*)w, (float*)h, H, eps);
    else if (dtype == VTGB_BF16)
        hipLaunchKernelGGL(llm_rmsnorm_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)delta, (const bf16_t*)w, (bf16_t*)h, H, eps);
    else
        hipLaunchKernelGGL(llm_rmsnorm_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, (float*)x, (const float*)delta, (const float*)w, (float*)h, H, eps);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// delta = the K-split fragments of a vtgb_gemm_skinny call with defer_reduce (bf16 activations; M = rows)
extern "C" int vtgb_llm_rmsnorm_parts(int dtype, void* x, const float* part, int32_t S, const void* w, void* h, int64_t rows, int32_t H, float eps,
                                      vtgb_stream_t s) {
    VTGB_REQUIRE(x && part && w && h && rows > 0 && rows <= 128 && S > 1 && dtype == VTGB_BF16, VTGB_EINVAL, "llm_rmsnorm_parts: bad argument");
    VTGB_REQUIRE((((uintptr_t)x | (uintptr_t)part | (uintptr_t)w | (uintptr_t)h) & 15) == 0 && (H == 4096 || H == 2048), VTGB_EUNSUPPORTED,
                 "llm_rmsnorm_parts: hidden size %d (4096 or 2048, 16-byte aligned operands)", H);
    if (H == 4096)
        hipLaunchKernelGGL((llm_rmsnorm_vec_kernel<bf16_t, 2>), dim3((unsigned)rows), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)nullptr, (const bf16_t*)w, (bf16_t*)h, H, eps,
                           part, S, (int)rows);
    else
        hipLaunchKernelGGL((llm_rmsnorm_vec_kernel<bf16_t, 1>), dim3((unsigned)rows), dim3(256), 0, s, (bf16_t*)x, (const bf16_t*)nullptr, (const bf16_t*)w, (bf16_t*)h, H, eps,
                           part, S, (int)rows);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

extern "C" int vtgb_llm_rope_cache_parts(int dtype, const float* part, int32_t S, void* q_out, void* kc, void* vc, const void* cos_t, const void* sin_t,
                                         const int64_t* pos, int32_t B, int32_t nq, int32_t nkv, int32_t hd, int32_t tmax, vtgb_stream_t s) {
    VTGB_REQUIRE(part && S > 1 && q_out && kc && vc && ((cos_t == nullptr) == (sin_t == nullptr)) && pos && B > 0 && B <= 128 && nq > 0 && nkv > 0 && (hd % 2) == 0 &&
                     dtype == VTGB_BF16,
                 VTGB_EINVAL, "llm_rope_cache_parts: bad argument");
    hipLaunchKernelGGL(llm_rope_cache_kernel<bf16_t>, dim3(nq + 2 * nkv, B), dim3(128), 0, s, (const bf16_t*)nullptr, (bf16_t*)q_out, (bf16_t*)kc, (bf16_t*)vc,
                       (const bf16_t*)cos_t, (const bf16_t*)sin_t, pos, nq, nkv, hd, tmax, part, S, B);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---- the prefill's counterpart: rotary on q and k of EVERY position, in place in qkv [B, S, (nq + 2 nkv) * hd] (the attention then reads q
// and k from there), k and v copied to the cache rows 0 .. S - 1.  One workgroup per (position, batch); bf16: a thread takes 8 channels of
// the first half of a head together with their partners in the second half (16-byte loads and stores); HF's roundings (each product, then
// the sum: LlamaRotaryEmbedding / apply_rotary_pos_emb in the model's dtype).
template <typename T>
__global__ __launch_bounds__(256) void llm_rope_cache_prefill_kernel(T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc, const T* __restrict__ cos_t,
                                                                     const T* __restrict__ sin_t, int S, int nq, int nkv, int hd, int tmax) {
#pragma clang fp contract(off)      // fp32: mul, mul, add are three roundings in the reference -- no FMA
    constexpr int V = 16 / (int)sizeof(T);                     // elements per 16-byte vector
    typedef T TV __attribute__((ext_vector_type(V)));
    const int spos = blockIdx.x, b = blockIdx.y, half = hd >> 1, cph = half / V;      // vectors per half head
    T* const row = qkv + ((int64_t)b * S + spos) * (nq + 2 * nkv) * hd;
    const T* const cr = cos_t + (int64_t)spos * hd;
    const T* const sr = sin_t + (int64_t)spos * hd;
    for (int it = threadIdx.x; it < (nq + nkv) * cph; it += blockDim.x) {
        const int head = it / cph, d = (it - head * cph) * V;
        T* const hp = row + head * hd;
        const TV x0 = *reinterpret_cast<const TV*>(hp + d), x1 = *reinterpret_cast<const TV*>(hp + d + half);
        const TV c0 = *reinterpret_cast<const TV*>(cr + d), c1 = *reinterpret_cast<const TV*>(cr + d + half);
        const TV s0 = *reinterpret_cast<const TV*>(sr + d), s1 = *reinterpret_cast<const TV*>(sr + d + half);
        TV o0, o1;
#pragma unroll
        for (int e = 0; e < V; e++) {
            // plain operators under `fp contract(off)` (the __fmul_rn / __fadd_rn of the HIP headers are inlined WITH their contract flags)
            const float p00 = Cvt<T>::rnd((float)x0[e] * (float)c0[e]), p01 = Cvt<T>::rnd(-(float)x1[e] * (float)s0[e]);
            const float p10 = Cvt<T>::rnd((float)x1[e] * (float)c1[e]), p11 = Cvt<T>::rnd((float)x0[e] * (float)s1[e]);
            o0[e] = Cvt<T>::to(Cvt<T>::rnd(p00 + p01));
            o1[e] = Cvt<T>::to(Cvt<T>::rnd(p10 + p11));
        }
        *reinterpret_cast<TV*>(hp + d) = o0;
        *reinterpret_cast<TV*>(hp + d + half) = o1;
        if (head >= nq) {
            T* const kd = kc + (((int64_t)b * nkv + (head - nq)) * tmax + spos) * hd;
            *reinterpret_cast<TV*>(kd + d) = o0;
            *reinterpret_cast<TV*>(kd + d + half) = o1;
        }
    }
    for (int it = threadIdx.x; it < nkv * (hd / V); it += blockDim.x) {
        const int head = it / (hd / V), d = (it - head * (hd / V)) * V;
        *reinterpret_cast<TV*>(vc + (((int64_t)b * nkv + head) * tmax + spos) * hd + d) = *reinterpret_cast<const TV*>(row + (nq + nkv + head) * hd + d);
    }
}

extern "C" int vtgb_llm_rope_cache(int dtype, const void* qkv, void* q_out, void* kc, void* vc, const void* cos_t, const void* sin_t,
                                   const int64_t* pos, int32_t B, int32_t nq, int32_t nkv, int32_t hd, int32_t tmax, vtgb_stream_t s) {
    VTGB_REQUIRE(qkv && q_out && kc && vc && ((cos_t == nullptr) == (sin_t == nullptr)) && pos && B > 0 && nq > 0 && nkv > 0 && (hd % 2) == 0, VTGB_EINVAL, "llm_rope_cache: bad argument");
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

extern "C" int vtgb_llm_rope_cache_prefill(int dtype, void* qkv, void* kc, void* vc, const void* cos_t, const void* sin_t, int32_t B, int32_t S, int32_t nq,
                                           int32_t nkv, int32_t hd, int32_t tmax, vtgb_stream_t s) {
    VTGB_REQUIRE(qkv && kc && vc && cos_t && sin_t && B > 0 && S > 0 && S <= tmax && nq > 0 && nkv > 0, VTGB_EINVAL, "llm_rope_cache_prefill: bad argument");
    VTGB_REQUIRE(dtype == VTGB_BF16 ? (hd % 16) == 0 : (hd % 8) == 0, VTGB_EUNSUPPORTED, "llm_rope_cache_prefill: head_dim=%d (16-byte vectors per half head)", hd);
    const dim3 grid(S, B);
    if (dtype == VTGB_BF16)
        hipLaunchKernelGGL(llm_rope_cache_prefill_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)qkv, (bf16_t*)kc, (bf16_t*)vc, (const bf16_t*)cos_t, (const bf16_t*)sin_t,
                           S, nq, nkv, hd, tmax);
    else
        hipLaunchKernelGGL(llm_rope_cache_prefill_kernel<float>, grid, dim3(256), 0, s, (float*)qkv, (float*)kc, (float*)vc, (const float*)cos_t, (const float*)sin_t, S,
                           nq, nkv, hd, tmax);
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
    const int V = dtype == VTGB_BF16 ? 8 : 4;
    if ((I % V) == 0 && rows <= 65535 && ((((uintptr_t)gu | (uintptr_t)act) & 15) == 0)) {
        const dim3 grid((unsigned)((I / V + 255) / 256), (unsigned)rows);
        if (dtype == VTGB_BF16) hipLaunchKernelGGL(llm_silu_mul_vec_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)gu, (bf16_t*)act, I);
        else hipLaunchKernelGGL(llm_silu_mul_vec_kernel<float>, grid, dim3(256), 0, s, (const float*)gu, (float*)act, I);
        VTGB_HIP(hipGetLastError());
        return VTGB_OK;
    }
    int64_t blocks = (rows * I + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == VTGB_BF16)
        hipLaunchKernelGGL(llm_silu_mul_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)gu, (bf16_t*)act, rows, I);
    else
        hipLaunchKernelGGL(llm_silu_mul_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)gu, (float*)act, rows, I);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---- attention of independent query rows over strided K / V with an additive per-(query position, head, key) bias: the T5 language
// model of the BLIP-2 flavours (transformers' modeling_t5: unscaled scores + bucketed relative position bias, softmax in fp32).  One
// wave per (row, head), 4 per workgroup.  Row r belongs to K/V batch r / rows_per_batch and query position r % rows_per_batch; it sees
// keys [0, n_keys) -- n_keys fixed, or *pos + 1 (decode step over a static cache, bias row *pos).  Covers the decoder's self-attention
// (rows_per_batch 1, cache [B, H, N, dk], bias), its cross-attention (fixed n_keys = encoder length, no bias) and the ENCODER's
// self-attention (rows = B x P straight out of the q|k|v projection: token-major strides, bias row = query position).
template <typename T>
__global__ __launch_bounds__(256) void llm_attn_rows_kernel(const vtgb_llm_attn_rows_args a) {
    extern __shared__ float dsm[];   // per wave: hd floats of q + t_pad scores
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hd = a.head_dim, nq = a.heads;
    const int64_t idx = (int64_t)blockIdx.x * 4 + wave;
    const int64_t total = (int64_t)a.rows * nq;
    const int64_t id = idx < total ? idx : total - 1;
    const int64_t r = id / nq;
    const int head = (int)(id - r * nq);
    const int64_t b = r / a.rows_per_batch;
    const int qpos = a.pos ? (int)(*a.pos) : (int)(r - b * a.rows_per_batch);
    const int n_keys = a.pos ? (int)(*a.pos) + 1 : a.n_keys;
    float* qs = dsm + wave * (hd + a.t_pad);
    float* sc = qs + hd;
    const T* qr = reinterpret_cast<const T*>(a.q) + r * a.q_row + (int64_t)head * hd;
    const T* kr = reinterpret_cast<const T*>(a.k) + b * a.kv_batch + (int64_t)head * a.kv_head;
    const T* vr = reinterpret_cast<const T*>(a.v) + b * a.kv_batch + (int64_t)head * a.kv_head;
    const T* br = a.bias ? reinterpret_cast<const T*>(a.bias) + (int64_t)qpos * a.bias_pos + (int64_t)head * a.bias_head : nullptr;
    for (int d = lane; d < hd; d += 64) qs[d] = (float)qr[d];
    __syncthreads();
    float mx = -INFINITY;
    for (int key = lane; key < n_keys; key += 64) {
        const T* k = kr + (int64_t)key * a.kv_tok;
        float dot = 0.f;
        for (int d = 0; d < hd; d++) dot = fmaf(qs[d], (float)k[d], dot);
        dot = Cvt<T>::rnd(dot * a.scale);                    // HF: scores in the model's dtype ...
        if (br) dot = Cvt<T>::rnd(dot + (float)br[key]);     // ... += position_bias, softmax in fp32
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
    if (idx < total) {
        T* o = reinterpret_cast<T*>(a.out) + r * a.o_row + (int64_t)head * hd;
        for (int d = lane; d < hd; d += 64) {
            float acc = 0.f;
            for (int key = 0; key < n_keys; key++) acc = fmaf(Cvt<T>::rnd(sc[key] * inv), (float)vr[(int64_t)key * a.kv_tok + d], acc);      // (weights as the model's dtype holds them)
            o[d] = Cvt<T>::to(acc);
        }
    }
}

extern "C" int vtgb_llm_attention_rows(const vtgb_llm_attn_rows_args* a, vtgb_stream_t s) {
    VTGB_REQUIRE(a && a->q && a->k && a->v && a->out && a->rows > 0 && a->heads > 0 && a->head_dim > 0 && a->rows_per_batch > 0, VTGB_EINVAL,
                 "llm_attention_rows: bad argument");
    VTGB_REQUIRE(a->pos || a->n_keys > 0, VTGB_EINVAL, "llm_attention_rows: n_keys or pos");
    VTGB_REQUIRE(a->t_pad >= (a->pos ? 1 : a->n_keys) && a->t_pad <= DEC_MAX_T && a->head_dim <= 256, VTGB_EUNSUPPORTED,
                 "llm_attention_rows: t_pad=%d head_dim=%d outside [n_keys .. %d], <= 256", a->t_pad, a->head_dim, DEC_MAX_T);
    const size_t lds = 4 * (size_t)(a->head_dim + a->t_pad) * sizeof(float);
    const dim3 grid((unsigned)(((int64_t)a->rows * a->heads + 3) / 4));
    if (a->dtype == VTGB_BF16) hipLaunchKernelGGL(llm_attn_rows_kernel<bf16_t>, grid, dim3(256), lds, s, *a);
    else hipLaunchKernelGGL(llm_attn_rows_kernel<float>, grid, dim3(256), lds, s, *a);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---- feed-forward activations of the language models: act[r, i] = f(gu[r, i]) (* gu[r, I + i] when gated).  kind 0 SiLU (Llama's SwiGLU = vtgb_llm_silu_mul),
// 1 gelu_new (T5 v1.1 / Flan-T5 "gated-gelu": 0.5 x (1 + tanh(sqrt(2 / pi) (x + 0.044715 x^3)))), 2 ReLU (original T5), 3 exact GELU.
template <typename T>
__global__ void llm_gated_act_kernel(const T* __restrict__ gu, T* __restrict__ act, int64_t rows, int I, int kind, int gated) {
    const int64_t n = rows * I;
    const int ld = gated ? 2 * I : I;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / I, c = i - r * I;
        const float g = (float)gu[r * ld + c];
        float f;
        if (kind == 0) f = g / (1.0f + expf(-g));
        else if (kind == 1) f = 0.5f * g * (1.0f + tanhf(0.7978845608028654f * (g + 0.044715f * g * g * g)));
        else if (kind == 2) f = fmaxf(g, 0.f);
        else f = 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f));
        f = Cvt<T>::rnd(f);
        act[i] = gated ? Cvt<T>::to(f * (float)gu[r * ld + I + c]) : Cvt<T>::to(f);
    }
}

extern "C" int vtgb_llm_gated_act(int dtype, const void* gu, void* act, int64_t rows, int32_t I, int32_t kind, int32_t gated, vtgb_stream_t s) {
    VTGB_REQUIRE(gu && act && rows > 0 && I > 0 && kind >= 0 && kind <= 3, VTGB_EINVAL, "llm_gated_act: bad argument");
    int64_t blocks = (rows * I + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == VTGB_BF16)
        hipLaunchKernelGGL(llm_gated_act_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)gu, (bf16_t*)act, rows, I, kind, gated);
    else
        hipLaunchKernelGGL(llm_gated_act_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)gu, (float*)act, rows, I, kind, gated);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---------------------------------------------------------------------------------------
// Skinny GEMM for the decode step: out[M, N] = x[M, K] . w[N, K]^T with M <= 128 rows (one token per clip) -- the weight
// matrix is read exactly once, so the kernel is an HBM stream of W with the matrix cores idling behind it, and what bounds it
// is the number of weight bytes a CU keeps IN FLIGHT (rate = bytes in flight / memory latency).
//   work      one workgroup per (128-row weight tile, K split): K is split where N / 128 tiles alone would leave most of the 256
//             CUs without a stream (N = 4096: 32 tiles).  Unsplit tiles round and store straight to `out`; split tiles leave an
//             fp32 fragment each and a second small launch adds a tile's fragments in split order and rounds once (deterministic:
//             no atomics.  Adding them in the last-arriving workgroup of the same launch was built in round 3 and measured 2-5x
//             SLOWER: the device-scope release / acquire fences it needs write back and invalidate the L2 under the stream).
//   tile      128 (all of M) x 128 weight rows; 4 multiplying waves of 128 x 32 (acc[2][8] of v_mfma_f32_16x16x32_bf16) + 2
//             loader waves.
//   weights   straight from global memory into the multiplying waves' REGISTERS in MFMA fragment layout (a wave owns its 32
//             weight rows, nobody else reads them: no LDS): SK_D k-tiles = SK_D x 4 KiB per wave ahead, 160 KiB per workgroup
//             (rounds 1-2 staged them through a 7-slot LDS ring: 96 KiB in flight, and LDS had to hold the activations too).
//             k-tiles past the end of the split are requested OUT OF RANGE of the descriptor (zeros, no memory traffic), so the
//             counted wait is the same constant on every step.
//   x         L2-resident; the two loader waves stage it with buffer_load ... lds (LDS-DMA) into a 9-slot ring, 7 k-tiles ahead.
//             They have their own vmcnt: a wave that mixed this shallow stream with the deep weight stream would drain the
//             weights every k-tile (vmcnt completes in order per wave).  One barrier per k-tile.
// x rows beyond M are clamped duplicates (never stored); weight rows beyond N read as zeros (descriptor range / packed zeros).
// ---------------------------------------------------------------------------------------
constexpr int SK_BN = 128, SK_BK = 64, SK_TILE = 128 * 128;   // SK_TILE: bytes of one operand tile (128 rows x 64 bf16)
constexpr int SK_D = 8;                                        // weight k-tiles in flight per multiplying wave (registers)
constexpr int SK_XSLOTS = 9, SK_XD = 8;                        // x ring (LDS) and how far ahead the loaders run
constexpr int SK_THREADS = 384;
typedef __attribute__((address_space(3))) void* sk_lptr_t;
typedef __attribute__((ext_vector_type(4))) int sk_i32x4;

__device__ __forceinline__ int sk_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// s_waitcnt vmcnt(8 n), n = 0..6 (vmcnt is 6 bits: [3:0] and [15:14]); the other counters are left alone
__device__ __forceinline__ void sk_wait_stages(int n) {
    if (n >= 6) __builtin_amdgcn_s_waitcnt(0xCF70);
    else if (n == 5) __builtin_amdgcn_s_waitcnt(0x8F78);
    else if (n == 4) __builtin_amdgcn_s_waitcnt(0x8F70);
    else if (n == 3) __builtin_amdgcn_s_waitcnt(0x4F78);
    else if (n == 2) __builtin_amdgcn_s_waitcnt(0x4F70);
    else if (n == 1) __builtin_amdgcn_s_waitcnt(0x0F78);
    else __builtin_amdgcn_s_waitcnt(0x0F70);
}

// part: [gridDim.x tiles][gridDim.y splits][M][128] fp32 (gridDim.y > 1 only)
__global__ __launch_bounds__(SK_THREADS) void gemm_skinny_kernel(const bf16_t* __restrict__ x, int M, int ldx, const bf16_t* __restrict__ w, int N, int ldw,
                                                                 int nk, float* __restrict__ part, int w_tiled, void* __restrict__ out, int64_t ldo, int out_f32) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];      // the x ring: SK_XSLOTS tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, S = gridDim.y, sp = blockIdx.y;
    const int k0 = (int)((int64_t)nk * sp / S), k1 = (int)((int64_t)nk * (sp + 1) / S), ns = k1 - k0;   // this split's k-tiles
    if (ns <= 0) return;
    if (wave >= 4) {
        // ---------------- loader waves: each stages 64 of the 128 x rows of every k-tile (8 pieces of 8 rows x 128 bytes)
        const int lw = wave - 4;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, 0x7FFFFF00, 0x00020000);
        unsigned voff[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int row = lw * 64 + i * 8 + (lane >> 3), slot = lane & 7, c = slot ^ ((row >> 1) & 7);
            voff[i] = (unsigned)((row < M ? row : M - 1) * ldx + c * 8) * 2u;
        }
#define SK_ISSUE_X(slot, kt)                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 8; i++)                                                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (sk_lptr_t)(sk_smem + (slot) * SK_TILE + (lw * 64 + i * 8) * 128), 16, voff[i], (kt) * (SK_BK * 2), 0, 0);
        for (int t = 0; t < SK_XD && t < ns; t++) { SK_ISSUE_X(t, k0 + t) }
        int slot_in = SK_XD % SK_XSLOTS;
        const int nsp = (ns + SK_D - 1) / SK_D * SK_D;        // the multiplying waves run whole groups of SK_D steps: same barrier count here
        for (int t = 0; t < nsp; t++) {
            // my pieces of k-tiles t AND t + 1 (the multiplying waves read one fragment group ahead, across the barrier) have landed
            // when only the k-tiles younger than t + 1 (8 pieces each) are outstanding
            const int younger = t >= ns - 2 ? 0 : ns - 2 - t < SK_XD - 2 ? ns - 2 - t : SK_XD - 2;
            sk_wait_stages(younger);
            __builtin_amdgcn_s_barrier();                      // x of k-tiles t, t + 1 is in LDS; everyone is done with k-tile t - 1
            if (t + SK_XD < ns) { SK_ISSUE_X(slot_in, k0 + t + SK_XD) }      // (slot of k-tile t + XD - XSLOTS = t - 1: free)
            slot_in = slot_in + 1 == SK_XSLOTS ? 0 : slot_in + 1;
        }
#undef SK_ISSUE_X
        return;
    }
    // ---------------- multiplying waves: 32 weight rows each, fragments straight from memory
    const int fr = lane & 15, fg = lane >> 4;
    // one descriptor for the tile; per-lane byte offsets of the two 16-row fragments' two 32-deep halves; k-tile step in bytes.
    // Row-major: range = the tile's rows inside the matrix (rows beyond N read as zeros).  Tiled (vtgb_pack_skinny_weight): the
    // tile's nk blocks of 16 KiB, rows in the LDS swizzle of rounds 1-2 (chunk q of row r holds k-chunk q ^ ((r >> 1) & 7)).
    const int rows_in = N - b * SK_BN < SK_BN ? N - b * SK_BN : SK_BN;
    const int64_t wbase = w_tiled ? (int64_t)b * nk * (SK_TILE / 2) : (int64_t)b * SK_BN * ldw;
    const unsigned wrange = w_tiled ? (unsigned)nk * SK_TILE : (unsigned)(((int64_t)(rows_in - 1) * ldw + nk * SK_BK) * 2);
    // (descriptor words by hand: the loads below are inline asm -- see SK_ISSUE_W)
    const uint64_t wptr = reinterpret_cast<uint64_t>(w + wbase);
    const sk_i32x4 wrsrc = {__builtin_amdgcn_readfirstlane((int)(unsigned)wptr), __builtin_amdgcn_readfirstlane((int)((wptr >> 32) & 0xFFFFu)),
                            __builtin_amdgcn_readfirstlane((int)wrange), 0x00020000};
    const int kstep = w_tiled ? SK_TILE : SK_BK * 2;
    unsigned wv[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int row = wave * 32 + i * 16 + fr, chunk = h * 4 + fg;
            wv[i][h] = w_tiled ? (unsigned)sk_swz(row, chunk) : (unsigned)(row * ldw + chunk * 8) * 2u;
        }
    sk_i32x4 wreg[SK_D][2][2];
    // The ring's loads are inline asm and its waits are written by hand: with the builtin, hipcc's own vmcnt bookkeeping drained
    // the whole ring (vmcnt(0)) at the top of every group of SK_D steps -- it cannot see that a register loaded in one trip of the
    // loop is consumed in the next -- which halves the bytes in flight.  SK_WAIT_W(u): everything but the SK_D - 1 younger k-tiles
    // (4 loads each) has landed, i.e. ring entry u; the empty asm ties the registers' next use to that point.
#define SK_ISSUE_W(u, t)                                                                                     \
    {                                                                                                        \
        const int so_ = __builtin_amdgcn_readfirstlane((t) < ns ? (k0 + (t)) * kstep : 0x7FFFFF00);   /* past the split: out of range, no traffic */ \
        _Pragma("unroll") for (int i = 0; i < 2; i++)                                                        \
            _Pragma("unroll") for (int h = 0; h < 2; h++)                                                    \
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(wreg[u][i][h]) : "v"(wv[i][h]), "s"(wrsrc), "s"(so_) : "memory"); \
    }
#define SK_WAIT_W(u)                                                                                         \
    {                                                                                                        \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (SK_D - 1)) : "memory");                                \
        _Pragma("unroll") for (int i = 0; i < 2; i++)                                                        \
            _Pragma("unroll") for (int h = 0; h < 2; h++) asm volatile("" : "+v"(wreg[u][i][h]));            \
    }
    f32x4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < SK_D; u++) { SK_ISSUE_W(u, u) }
    // x fragments: four at a time ("quad" q of a k-tile: 32-deep half q >> 1, x rows 64 (q & 1) ...), double buffered -- the next
    // quad's LDS reads are issued before the current quad's 8 MFMAs (one wave per SIMD: nobody else hides the LDS latency), and quad
    // 0 of the NEXT k-tile before the last quad of this one (the loaders guarantee k-tile t + 1 at barrier t).
    bf16x8 xf[2][4];
#define SK_READ_X(buf, xs_, q)                                                                               \
    _Pragma("unroll") for (int j = 0; j < 4; j++)                                                            \
        xf[buf][j] = *reinterpret_cast<const bf16x8*>((xs_) + sk_swz((((q) & 1) * 4 + j) * 16 + fr, ((q) >> 1) * 4 + fg));
    int slot_x = 0;
    const char* xs = sk_smem;
    bool primed = false;
    for (int t0 = 0; t0 < ns; t0 += SK_D) {
#pragma unroll
        for (int u = 0; u < SK_D; u++) {
            // Straight-line steps (a branch per step made hipcc drain the weight ring with vmcnt(0..3) at every join): the steps of
            // the last group past the split multiply ZERO weight fragments (out-of-range loads) with the last real x tile.
            const int t = t0 + u;
            __builtin_amdgcn_s_barrier();                      // x of k-tiles t, t + 1 is in LDS
            if (!primed) { SK_READ_X(0, xs, 0) primed = true; }
            if (t < ns - 1) slot_x = slot_x + 1 == SK_XSLOTS ? 0 : slot_x + 1;
            const char* const xs_next = sk_smem + slot_x * SK_TILE;
            SK_WAIT_W(u)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q < 3) { SK_READ_X((q + 1) & 1, xs, q + 1) } else { SK_READ_X(0, xs_next, 0) }
                __builtin_amdgcn_sched_barrier(0);             // (left alone, hipcc sinks the reads to just before their MFMAs)
                const bf16x8 wf0 = __builtin_bit_cast(bf16x8, wreg[u][0][q >> 1]), wf1 = __builtin_bit_cast(bf16x8, wreg[u][1][q >> 1]);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    acc[0][(q & 1) * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0, xf[q & 1][j], acc[0][(q & 1) * 4 + j], 0, 0, 0);
                    acc[1][(q & 1) * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1, xf[q & 1][j], acc[1][(q & 1) * 4 + j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            xs = xs_next;
            SK_ISSUE_W(u, t + SK_D)
        }
    }
#undef SK_READ_X
#undef SK_ISSUE_W
#undef SK_WAIT_W
    // D layout: column (lane & 15) <- x row (m), rows (lane >> 4) * 4 + reg <- w row (n)
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int m = j * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int c = wave * 32 + i * 16 + fg * 4, n = b * SK_BN + c;
            const f32x4 v = acc[i][j];
            if (S > 1) {
                *reinterpret_cast<f32x4*>(part + ((int64_t)(b * S + sp) * M + m) * SK_BN + c) = v;
            } else if (n < N) {                                // no K split: round once and store straight to `out`
                if (out_f32) {
                    float* o = reinterpret_cast<float*>(out) + m * ldo + n;
                    if (n + 3 < N) *reinterpret_cast<f32x4*>(o) = v;
                    else for (int e = 0; e < 4 && n + e < N; e++) o[e] = v[e];
                } else {
                    bf16_t* o = reinterpret_cast<bf16_t*>(out) + m * ldo + n;
                    if (n + 3 < N) *reinterpret_cast<bf16x4*>(o) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                    else for (int e = 0; e < 4 && n + e < N; e++) o[e] = (bf16_t)v[e];
                }
            }
        }
    }
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

// splits.  Measured (tools/exp/skinny_bench.py, hipGraph replay, M = 124; profiles/r03_skinny_experiments.md): a workgroup streams
// ~16 KiB of weights per 0.45 us whatever is in flight (ablations: without the x stream 0.47 us per k-tile, without the weight stream
// 0.43 us -- both go through the CU's one L1/TA path, and x is re-read by every tile), so the stream count decides: >= 160 tiles run
// unsplit (gate|up, lm_head); fewer tiles are split until ~128 workgroups stream, long K a little further (one split per 24
// k-tiles), never more than 8 splits (each costs an M x 128 fp32 fragment written and read back) or fewer than 8 k-tiles per split.
static int skinny_splits(const vtgb_gemm_skinny_args* a) {
    const int nk = a->K / SK_BK, n_tiles = (a->N + SK_BN - 1) / SK_BN;
    if (a->n_splits > 0) return a->n_splits < nk ? a->n_splits : nk;
    if (n_tiles >= 160) return 1;
    int S = (128 + n_tiles - 1) / n_tiles;
    if (S < nk / 24) S = nk / 24;
    if (S > 8) S = 8;
    while (S > 1 && nk / S < 8) S--;
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

extern "C" int32_t vtgb_gemm_skinny_splits(const vtgb_gemm_skinny_args* a) {
    if (skinny_check(a) != VTGB_OK) return 0;
    return skinny_splits(a);
}

extern "C" size_t vtgb_gemm_skinny_workspace_bytes(const vtgb_gemm_skinny_args* a) {
    if (skinny_check(a) != VTGB_OK) return 0;
    const int S = skinny_splits(a);
    return S == 1 ? 0 : (size_t)((a->N + SK_BN - 1) / SK_BN) * S * a->M * SK_BN * sizeof(float);
}

extern "C" int vtgb_gemm_skinny(const vtgb_gemm_skinny_args* a, vtgb_stream_t s) {
    VTGB_TRY(skinny_check(a));
    VTGB_REQUIRE(a->x && a->w && a->out, VTGB_EINVAL, "gemm_skinny: NULL operand");
    const int nk = a->K / SK_BK, n_tiles = (a->N + SK_BN - 1) / SK_BN, S = skinny_splits(a);
    const size_t need = S == 1 ? 0 : (size_t)n_tiles * S * a->M * SK_BN * sizeof(float);
    VTGB_REQUIRE(need == 0 || (a->workspace && a->workspace_bytes >= need), VTGB_EWORKSPACE, "gemm_skinny: workspace %zu < %zu bytes", a->workspace_bytes, need);
    constexpr int LDS = SK_XSLOTS * SK_TILE;
    static DeviceOnce attr;
    VTGB_FUNC_LDS_ONCE(attr, gemm_skinny_kernel, LDS);
    {
        ProfScope prof(VTGB_PROF_GEMM, 2.0 * a->M * a->N * a->K, s);
        hipLaunchKernelGGL(gemm_skinny_kernel, dim3(n_tiles, S), dim3(SK_THREADS), LDS, s, (const bf16_t*)a->x, a->M, (int)a->ldx, (const bf16_t*)a->w, a->N,
                           (int)a->ldw, nk, (float*)a->workspace, a->w_tiled, a->out, a->ldo, a->out_dtype == VTGB_F32 ? 1 : 0);
    }
    if (S > 1 && !a->defer_reduce) {      // (defer_reduce: the consumer adds the fragments -- vtgb_llm_rmsnorm_parts / vtgb_llm_rope_cache_parts)
        const dim3 rgrid(n_tiles, (a->M + 7) / 8 < 4 ? (a->M + 7) / 8 : 4);
        if (a->out_dtype == VTGB_BF16)
            hipLaunchKernelGGL(gemm_skinny_reduce_kernel<bf16_t>, rgrid, dim3(256), 0, s, (const float*)a->workspace, a->M, a->N, S, (bf16_t*)a->out, a->ldo);
        else
            hipLaunchKernelGGL(gemm_skinny_reduce_kernel<float>, rgrid, dim3(256), 0, s, (const float*)a->workspace, a->M, a->N, S, (float*)a->out, a->ldo);
    }
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
