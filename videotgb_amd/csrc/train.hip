// train.hip -- the loss side of the LoRA training step:
//   concat_text_input_output + label masking (src/models/LSTP_Vicuna_IVT_module.py:692-718, :284-291)
//   shifted cross-entropy with ignore_index -100, mean reduction (:297-299, :325-326) and its gradient.
// The language model, LoRA adapters and AdamW stay PyTorch (third-party peft / torch.optim in the reference);
// what lives here is the integer token shuffling (a per-row Python loop with one host sync per row in the
// reference) and the [B, S, V] logits pass, which the reference materialises three times (a contiguous shifted
// copy, log-softmax, gradient): here the shift is an index, the logits are read once forward and once backward.
#include "common.h"

#include <math.h>

// ---- one workgroup per batch row: n = sum(input_atts[b]); llm = [input[:n] | output[1:] | input[n:]];
// labels = [-100 x prefix_len | llm_ids with pad -> -100 and the first n positions -> -100]
__global__ __launch_bounds__(256) void concat_text_io_kernel(const vtgb_concat_text_io_args a) {
    __shared__ int s_n;
    __shared__ int red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t* ia = a.input_atts + (int64_t)b * a.Li;
    int part = 0;
    for (int i = tid; i < a.Li; i += 256) part += (int)ia[i];
    red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        s_n = red[0];
        if (a.input_len) a.input_len[b] = red[0];
    }
    __syncthreads();
    const int n = s_n, L = a.Li + a.Lo - 1;
    const int64_t* ii = a.input_ids + (int64_t)b * a.Li;
    const int64_t* oi = a.output_ids + (int64_t)b * a.Lo;
    const int64_t* oa = a.output_atts + (int64_t)b * a.Lo;
    for (int j = tid; j < L; j += 256) {
        int64_t id, at;
        if (j < n) { id = ii[j]; at = ia[j]; }
        else if (j < n + a.Lo - 1) { id = oi[j - n + 1]; at = oa[j - n + 1]; }
        else { id = ii[j - (a.Lo - 1)]; at = ia[j - (a.Lo - 1)]; }
        a.llm_ids[(int64_t)b * L + j] = id;
        a.llm_atts[(int64_t)b * L + j] = at;
        if (a.labels) a.labels[(int64_t)b * (a.prefix_len + L) + a.prefix_len + j] = (id == a.pad_id || j < n) ? -100 : id;
    }
    if (a.labels)
        for (int j = tid; j < a.prefix_len; j += 256) a.labels[(int64_t)b * (a.prefix_len + L) + j] = -100;
}

extern "C" int vtgb_concat_text_io(const vtgb_concat_text_io_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->input_ids && a->input_atts && a->output_ids && a->output_atts && a->llm_ids && a->llm_atts, VTGB_EINVAL,
                 "concat_text_io: NULL argument");
    VTGB_REQUIRE(a->B > 0 && a->Li > 0 && a->Lo > 0 && a->prefix_len >= 0, VTGB_EINVAL, "concat_text_io: B=%d Li=%d Lo=%d prefix=%d", a->B, a->Li,
                 a->Lo, a->prefix_len);
    hipLaunchKernelGGL(concat_text_io_kernel, dim3(a->B), dim3(256), 0, stream, *a);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---- shifted cross-entropy.  Row (b, t), t < S - 1, scores logits[b, t, :] against labels[b, t + 1].
// forward: lse[b, t] = logsumexp(logits[b, t, :]) (fp32, two-pass max / sum over the row kept in registers when
// V <= 256 * 128), row_loss = lse - logit[label] or 0 for ignored rows; then one workgroup reduces the rows in a
// fixed order (deterministic) into loss = sum / count.
template <typename T>
__global__ __launch_bounds__(256) void ce_rows_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels, float* __restrict__ lse,
                                                      float* __restrict__ row_loss, int S, int V, int64_t ld_b, int64_t ld_t) {
    __shared__ float red[256];
    const int64_t r = blockIdx.x;                       // r = b * (S - 1) + t
    const int64_t b = r / (S - 1), t = r - b * (S - 1);
    const int tid = threadIdx.x;
    const int64_t lab = labels[b * S + t + 1];
    if (lab == -100) {                                   // ignored row: no pass over the logits
        if (tid == 0) { lse[r] = 0.f; row_loss[r] = 0.f; }
        return;
    }
    const T* x = logits + b * ld_b + t * ld_t;
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 256) mx = fmaxf(mx, (float)x[i]);
    red[tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < V; i += 256) sum += expf((float)x[i] - mx);
    red[tid] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        const float l = mx + logf(red[0]);
        lse[r] = l;
        row_loss[r] = l - (float)x[lab];
    }
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ labels, float* __restrict__ out,
                                                        int64_t rows, int S) {
    __shared__ double rs[256];
    __shared__ int64_t rc[256];
    const int tid = threadIdx.x;
    double s = 0.0;
    int64_t c = 0;
    for (int64_t r = tid; r < rows; r += 256) {
        const int64_t b = r / (S - 1), t = r - b * (S - 1);
        if (labels[b * S + t + 1] != -100) { s += (double)row_loss[r]; c++; }
    }
    rs[tid] = s; rc[tid] = c;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) { rs[tid] += rs[tid + k]; rc[tid] += rc[tid + k]; }
        __syncthreads();
    }
    if (tid == 0) {
        out[0] = rc[0] ? (float)(rs[0] / (double)rc[0]) : NAN;   // CrossEntropyLoss(mean) over zero targets is nan
        out[1] = (float)rc[0];
    }
}

// backward: dlogits[b, t, v] = g / count * (softmax(logits[b, t])[v] - [v == label]); rows that are ignored and
// the last position of every sequence (it predicts nothing) get zeros.
template <typename T>
__global__ __launch_bounds__(256) void ce_backward_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ lse,
                                                          const float* __restrict__ loss_count, const float* __restrict__ grad_out, T* __restrict__ dlogits,
                                                          int S, int V, int64_t ld_b, int64_t ld_t) {
    const int64_t r = blockIdx.x;                        // r = b * S + t over ALL positions
    const int64_t b = r / S, t = r - b * S;
    const int tid = threadIdx.x;
    T* d = dlogits + b * ld_b + t * ld_t;
    const int64_t lab = t + 1 < S ? labels[b * S + t + 1] : -100;
    if (lab == -100) {
        for (int i = tid; i < V; i += 256) d[i] = (T)0.f;
        return;
    }
    const float g = grad_out[0] / loss_count[1];
    const float l = lse[b * (S - 1) + t];
    const T* x = logits + b * ld_b + t * ld_t;
    for (int i = tid; i < V; i += 256) {
        float p = expf((float)x[i] - l);
        if (i == lab) p -= 1.f;
        d[i] = (T)(g * p);
    }
}

extern "C" int vtgb_shifted_ce_forward(const vtgb_shifted_ce_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->logits && a->labels && a->lse && a->row_loss && a->loss, VTGB_EINVAL, "shifted_ce: NULL argument");
    VTGB_REQUIRE(a->B > 0 && a->S > 1 && a->V > 0 && (a->dtype == VTGB_F32 || a->dtype == VTGB_BF16), VTGB_EINVAL, "shifted_ce: B=%d S=%d V=%d dtype=%d",
                 a->B, a->S, a->V, a->dtype);
    const int64_t rows = (int64_t)a->B * (a->S - 1);
    if (a->dtype == VTGB_F32)
        hipLaunchKernelGGL(ce_rows_kernel<float>, dim3((unsigned)rows), dim3(256), 0, stream, (const float*)a->logits, a->labels, a->lse, a->row_loss, a->S, a->V,
                           (int64_t)a->S * a->V, (int64_t)a->V);
    else
        hipLaunchKernelGGL(ce_rows_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, stream, (const bf16_t*)a->logits, a->labels, a->lse, a->row_loss, a->S,
                           a->V, (int64_t)a->S * a->V, (int64_t)a->V);
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, stream, a->row_loss, a->labels, a->loss, rows, a->S);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

extern "C" int vtgb_shifted_ce_backward(const vtgb_shifted_ce_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->logits && a->labels && a->lse && a->loss && a->grad_out && a->dlogits, VTGB_EINVAL, "shifted_ce backward: NULL argument");
    VTGB_REQUIRE(a->B > 0 && a->S > 1 && a->V > 0 && (a->dtype == VTGB_F32 || a->dtype == VTGB_BF16), VTGB_EINVAL, "shifted_ce: B=%d S=%d V=%d dtype=%d",
                 a->B, a->S, a->V, a->dtype);
    const int64_t rows = (int64_t)a->B * a->S;
    if (a->dtype == VTGB_F32)
        hipLaunchKernelGGL(ce_backward_kernel<float>, dim3((unsigned)rows), dim3(256), 0, stream, (const float*)a->logits, a->labels, a->lse, a->loss, a->grad_out,
                           (float*)a->dlogits, a->S, a->V, (int64_t)a->S * a->V, (int64_t)a->V);
    else
        hipLaunchKernelGGL(ce_backward_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, stream, (const bf16_t*)a->logits, a->labels, a->lse, a->loss, a->grad_out,
                           (bf16_t*)a->dlogits, a->S, a->V, (int64_t)a->S * a->V, (int64_t)a->V);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
