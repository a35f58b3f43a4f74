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
