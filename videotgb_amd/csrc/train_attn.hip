// train_attn.hip -- softmax attention FORWARD + BACKWARD for the trainable stages (config C5 / the SF flavours: Q-Former self- and
// cross-attention, xinstructblip.py:611-694; the TGB's RoPE-BERT attention, xropebert.py:243-332), fp32, with the reference's
// attention-probability dropout as an injectable multiplicative mask.
//
//   A_ij  = softmax_j(q_i . k_j * scale + key_mask_j)         (the probabilities)
//   out_i = sum_j A_ij d_ij v_j                                (d_ij = 0 or 1 / (1 - p): dropout on the probabilities, :679 / :307)
//   backward:  dPd_ij = dout_i . v_j,   delta_i = dout_i . out_i,   dS_ij = A_ij (d_ij dPd_ij - delta_i)
//              dq_i = scale sum_j dS_ij k_j,   dk_j = scale sum_i dS_ij q_i,   dv_j = sum_i A_ij d_ij dout_i
//
// These stages are small (32 + Lt query rows against <= 257 keys per frame and head; <= 258 x 258 for the TGB): a few GFLOP per
// step next to the language model's TFLOPs -- the kernels are plain fp32 FMA code, correctness-shaped: FOUR lanes share a row
// (query row in the forward / dq pass, key row in the dk / dv pass), each holding a quarter of the head dimension in registers;
// the other operand streams through LDS in chunks of 64 rows (broadcast reads); dot products finish with two lane shuffles.
// The probabilities are never stored: the backward recomputes them from the saved log-sum-exp.
#include <math.h>

#include "common.h"

constexpr int TA_ROWS = 64;   // rows per LDS chunk, and rows per workgroup pass (256 threads / 4 lanes per row)

template <int EPL>
__device__ __forceinline__ float dot4(const float (&a)[EPL], const float* __restrict__ b) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; e++) s = fmaf(a[e], b[e], s);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    return s;
}

// stage `rows` rows [r0, r0 + rows) of a [*, heads * hd] token-major operand into LDS as [TA_ROWS][4 * EPL] (zero padded), times `mul`
template <int EPL>
__device__ __forceinline__ void stage_rows(float* lds, const float* __restrict__ src, int64_t tok_stride, int r0, int n_rows, int hd, float mul) {
    constexpr int W = 4 * EPL;
    for (int idx = threadIdx.x; idx < TA_ROWS * W; idx += blockDim.x) {
        const int r = idx / W, c = idx - r * W;
        float v = 0.f;
        if (r0 + r < n_rows && c < hd) v = src[(int64_t)(r0 + r) * tok_stride + c] * mul;
        lds[idx] = v;
    }
}

template <int EPL>
__global__ __launch_bounds__(256) void attn_train_fwd_kernel(const vtgb_attn_train_args a) {
    constexpr int W = 4 * EPL;
    __shared__ float Ks[TA_ROWS * W], Vs[TA_ROWS * W];
    const int bh = blockIdx.x, b = bh / a.heads, h = bh - b * a.heads, hd = a.head_dim;
    const int sub = threadIdx.x & 3, slot = threadIdx.x >> 2, c0 = sub * EPL;
    const float* qb = a.q + (int64_t)b * a.q_batch + h * hd;
    const float* kb = a.k + (int64_t)b * a.kv_batch + h * hd;
    const float* vb = a.v + (int64_t)b * a.kv_batch + h * hd;
    for (int i0 = 0; i0 < a.s_q; i0 += TA_ROWS) {
        const int i = i0 + slot;
        const bool live = i < a.s_q;
        float q[EPL], o[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            q[e] = (live && c0 + e < hd) ? qb[(int64_t)i * a.q_tok + c0 + e] * a.scale : 0.f;
            o[e] = 0.f;
        }
        float m = -INFINITY, l = 0.f;
        for (int j0 = 0; j0 < a.s_kv; j0 += TA_ROWS) {
            __syncthreads();
            stage_rows<EPL>(Ks, kb, a.kv_tok, j0, a.s_kv, hd, 1.0f);
            stage_rows<EPL>(Vs, vb, a.kv_tok, j0, a.s_kv, hd, 1.0f);
            __syncthreads();
            const int nj = min(TA_ROWS, a.s_kv - j0);
            for (int j = 0; j < nj; j++) {
                float s = dot4<EPL>(q, Ks + j * W + c0);
                if (a.key_mask) s += a.key_mask[(int64_t)b * a.s_kv + j0 + j];
                const float mn = fmaxf(m, s), alpha = __expf(m - mn), p = __expf(s - mn);     // (m = -inf: alpha = 0)
                l = l * alpha + p;
                const float d = (a.drop && live) ? a.drop[(((int64_t)b * a.heads + h) * a.s_q + i) * a.s_kv + j0 + j] : 1.0f;
                const float pd = p * d;
#pragma unroll
                for (int e = 0; e < EPL; e++) o[e] = fmaf(pd, Vs[j * W + c0 + e], o[e] * alpha);
                m = mn;
            }
        }
        if (live) {
            const float inv = 1.0f / l;
#pragma unroll
            for (int e = 0; e < EPL; e++)
                if (c0 + e < hd) a.out[(int64_t)b * a.o_batch + (int64_t)i * a.o_tok + h * hd + c0 + e] = o[e] * inv;
            if (sub == 0) a.lse[((int64_t)b * a.heads + h) * a.s_q + i] = m + __logf(l);
        }
    }
}

// dq (and delta_i = dout_i . out_i, parked for the dk / dv pass): four lanes per QUERY row, keys / values stream through LDS
template <int EPL>
__global__ __launch_bounds__(256) void attn_train_dq_kernel(const vtgb_attn_train_args a) {
    constexpr int W = 4 * EPL;
    __shared__ float Ks[TA_ROWS * W], Vs[TA_ROWS * W];
    const int bh = blockIdx.x, b = bh / a.heads, h = bh - b * a.heads, hd = a.head_dim;
    const int sub = threadIdx.x & 3, slot = threadIdx.x >> 2, c0 = sub * EPL;
    const float* qb = a.q + (int64_t)b * a.q_batch + h * hd;
    const float* kb = a.k + (int64_t)b * a.kv_batch + h * hd;
    const float* vb = a.v + (int64_t)b * a.kv_batch + h * hd;
    for (int i0 = 0; i0 < a.s_q; i0 += TA_ROWS) {
        const int i = i0 + slot;
        const bool live = i < a.s_q;
        float q[EPL], go[EPL], dq[EPL];
        float dl = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const bool ok = live && c0 + e < hd;
            q[e] = ok ? qb[(int64_t)i * a.q_tok + c0 + e] * a.scale : 0.f;
            go[e] = ok ? a.dout[(int64_t)b * a.o_batch + (int64_t)i * a.o_tok + h * hd + c0 + e] : 0.f;
            const float ov = ok ? a.out[(int64_t)b * a.o_batch + (int64_t)i * a.o_tok + h * hd + c0 + e] : 0.f;
            dl = fmaf(go[e], ov, dl);
            dq[e] = 0.f;
        }
        dl += __shfl_xor(dl, 1);
        dl += __shfl_xor(dl, 2);
        const float lse = live ? a.lse[((int64_t)b * a.heads + h) * a.s_q + i] : 0.f;
        if (live && sub == 0) a.delta[((int64_t)b * a.heads + h) * a.s_q + i] = dl;
        for (int j0 = 0; j0 < a.s_kv; j0 += TA_ROWS) {
            __syncthreads();
            stage_rows<EPL>(Ks, kb, a.kv_tok, j0, a.s_kv, hd, 1.0f);
            stage_rows<EPL>(Vs, vb, a.kv_tok, j0, a.s_kv, hd, 1.0f);
            __syncthreads();
            const int nj = min(TA_ROWS, a.s_kv - j0);
            for (int j = 0; j < nj; j++) {
                float s = dot4<EPL>(q, Ks + j * W + c0);
                if (a.key_mask) s += a.key_mask[(int64_t)b * a.s_kv + j0 + j];
                const float A = __expf(s - lse);
                const float dpd = dot4<EPL>(go, Vs + j * W + c0);
                const float d = (a.drop && live) ? a.drop[(((int64_t)b * a.heads + h) * a.s_q + i) * a.s_kv + j0 + j] : 1.0f;
                const float ds = A * (d * dpd - dl);
#pragma unroll
                for (int e = 0; e < EPL; e++) dq[e] = fmaf(ds, Ks[j * W + c0 + e], dq[e]);
            }
        }
        if (live) {
#pragma unroll
            for (int e = 0; e < EPL; e++)
                if (c0 + e < hd) a.dq[(int64_t)b * a.q_batch + (int64_t)i * a.q_tok + h * hd + c0 + e] = dq[e] * a.scale;
        }
    }
}

// dk, dv: four lanes per KEY row; (scaled) queries and dout stream through LDS together with the rows' lse / delta
template <int EPL>
__global__ __launch_bounds__(256) void attn_train_dkv_kernel(const vtgb_attn_train_args a) {
    constexpr int W = 4 * EPL;
    __shared__ float Qs[TA_ROWS * W], Gs[TA_ROWS * W], Ls[TA_ROWS], Ds[TA_ROWS];
    const int bh = blockIdx.x, b = bh / a.heads, h = bh - b * a.heads, hd = a.head_dim;
    const int sub = threadIdx.x & 3, slot = threadIdx.x >> 2, c0 = sub * EPL;
    const float* qb = a.q + (int64_t)b * a.q_batch + h * hd;
    const float* gb = a.dout + (int64_t)b * a.o_batch + h * hd;
    const float* kb = a.k + (int64_t)b * a.kv_batch + h * hd;
    const float* vb = a.v + (int64_t)b * a.kv_batch + h * hd;
    for (int j0 = 0; j0 < a.s_kv; j0 += TA_ROWS) {
        const int j = j0 + slot;
        const bool live = j < a.s_kv;
        float k[EPL], v[EPL], dk[EPL], dv[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const bool ok = live && c0 + e < hd;
            k[e] = ok ? kb[(int64_t)j * a.kv_tok + c0 + e] : 0.f;
            v[e] = ok ? vb[(int64_t)j * a.kv_tok + c0 + e] : 0.f;
            dk[e] = 0.f; dv[e] = 0.f;
        }
        const float mask = (a.key_mask && live) ? a.key_mask[(int64_t)b * a.s_kv + j] : 0.f;
        for (int i0 = 0; i0 < a.s_q; i0 += TA_ROWS) {
            __syncthreads();
            stage_rows<EPL>(Qs, qb, a.q_tok, i0, a.s_q, hd, a.scale);
            stage_rows<EPL>(Gs, gb, a.o_tok, i0, a.s_q, hd, 1.0f);
            if (threadIdx.x < TA_ROWS) {
                const int i = i0 + threadIdx.x;
                Ls[threadIdx.x] = i < a.s_q ? a.lse[((int64_t)b * a.heads + h) * a.s_q + i] : 0.f;
                Ds[threadIdx.x] = i < a.s_q ? a.delta[((int64_t)b * a.heads + h) * a.s_q + i] : 0.f;
            }
            __syncthreads();
            const int ni = min(TA_ROWS, a.s_q - i0);
            for (int i = 0; i < ni; i++) {
                const float s = dot4<EPL>(k, Qs + i * W + c0) + mask;
                const float A = __expf(s - Ls[i]);
                const float dpd = dot4<EPL>(v, Gs + i * W + c0);
                const float d = (a.drop && live) ? a.drop[(((int64_t)b * a.heads + h) * a.s_q + i0 + i) * a.s_kv + j] : 1.0f;
                const float pd = A * d, ds = A * (d * dpd - Ds[i]);
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    dv[e] = fmaf(pd, Gs[i * W + c0 + e], dv[e]);
                    dk[e] = fmaf(ds, Qs[i * W + c0 + e], dk[e]);      // (Qs holds q * scale)
                }
            }
        }
        if (live) {
#pragma unroll
            for (int e = 0; e < EPL; e++)
                if (c0 + e < hd) {
                    a.dk[(int64_t)b * a.kv_batch + (int64_t)j * a.kv_tok + h * hd + c0 + e] = dk[e];
                    a.dv[(int64_t)b * a.kv_batch + (int64_t)j * a.kv_tok + h * hd + c0 + e] = dv[e];
                }
        }
    }
}

static int check_args(const vtgb_attn_train_args* a, bool bwd) {
    VTGB_REQUIRE(a, VTGB_EINVAL, "attn_train: NULL args");
    VTGB_REQUIRE(a->batch > 0 && a->heads > 0 && a->head_dim > 0 && a->head_dim <= 128 && a->s_q > 0 && a->s_kv > 0, VTGB_EINVAL,
                 "attn_train: bad dims B=%d H=%d hd=%d Sq=%d Skv=%d", a->batch, a->heads, a->head_dim, a->s_q, a->s_kv);
    VTGB_REQUIRE(a->q && a->k && a->v && a->out && a->lse, VTGB_EINVAL, "attn_train: NULL operand");
    if (bwd) VTGB_REQUIRE(a->dout && a->dq && a->dk && a->dv && a->delta, VTGB_EINVAL, "attn_train backward: NULL gradient operand");
    return VTGB_OK;
}

#define TA_DISPATCH(KERNEL)                                                                                      \
    do {                                                                                                         \
        const dim3 grid((unsigned)(a->batch * a->heads));                                                        \
        if (a->head_dim <= 32) hipLaunchKernelGGL(KERNEL<8>, grid, dim3(256), 0, stream, *a);                    \
        else if (a->head_dim <= 64) hipLaunchKernelGGL(KERNEL<16>, grid, dim3(256), 0, stream, *a);              \
        else hipLaunchKernelGGL(KERNEL<32>, grid, dim3(256), 0, stream, *a);                                     \
        VTGB_HIP(hipGetLastError());                                                                             \
    } while (0)

extern "C" int vtgb_attn_train_forward(const vtgb_attn_train_args* a, vtgb_stream_t stream) {
    VTGB_TRY(check_args(a, false));
    TA_DISPATCH(attn_train_fwd_kernel);
    return VTGB_OK;
}

extern "C" int vtgb_attn_train_backward(const vtgb_attn_train_args* a, vtgb_stream_t stream) {
    VTGB_TRY(check_args(a, true));
    TA_DISPATCH(attn_train_dq_kernel);       // also writes delta
    TA_DISPATCH(attn_train_dkv_kernel);
    return VTGB_OK;
}
