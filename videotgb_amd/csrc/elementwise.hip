// elementwise.hip -- HBM-bound row kernels of the path: LayerNorm, im2col, embedding gathers,
// the optical-flow patch reduction of the TGB embedding, frame mean-pool, mask conversion.
// All of them move each byte once with 16-byte accesses per lane where the layout allows.
#include "common.h"

// ---------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row kept in registers (D <= 2048, D % 4 == 0), two-pass
// mean / biased variance in fp32 like torch.nn.functional.layer_norm.
// ---------------------------------------------------------------------------------------
template <typename TAct>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnDesc p) {
    const int lane = threadIdx.x & 63;
    const int nchunk = p.D >> 2;
    // grid-stride over rows: a bounded grid of resident workgroups walks the rows (one workgroup per 4 rows exits after
    // ~2 us and the dispatcher, not HBM, then sets the pace: 2.9 TB/s on the ViT's 255 k x 1408 rows)
    // gamma / beta are staged once per workgroup in LDS (r3; rounds 1-2 kept them in 64 registers per lane, which capped the kernel at 4
    // waves per SIMD: a wave walks its rows serially -- load, two reductions, store -- so the bytes in flight are waves x one row, and
    // the launch ran at 3.6 TB/s)
    __shared__ float4 gms[512], bts[512];
    for (int ch = threadIdx.x; ch < nchunk; ch += 256) {
        gms[ch] = *reinterpret_cast<const float4*>(p.gamma + ch * 4);
        bts[ch] = *reinterpret_cast<const float4*>(p.beta + ch * 4);
    }
    __syncthreads();
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < p.M; row += (int64_t)gridDim.x * 4) {
    const float* x = p.x + map_row(p.x_map, row) * p.ldx;
    float4 v[8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int ch = c * 64 + lane;
        v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ch < nchunk) {
            v[c] = *reinterpret_cast<const float4*>(x + ch * 4);
            sum += (v[c].x + v[c].y) + (v[c].z + v[c].w);
        }
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)p.D;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int ch = c * 64 + lane;
        if (ch < nchunk) {
            const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
            sq += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = rsqrtf(sq / (float)p.D + p.eps);
    const int64_t orow = map_row(p.o_map, row);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int ch = c * 64 + lane;
        if (ch < nchunk) {
            const float4 g = gms[ch], b = bts[ch];
            float4 y;
            y.x = (v[c].x - mean) * rstd * g.x + b.x;
            y.y = (v[c].y - mean) * rstd * g.y + b.y;
            y.z = (v[c].z - mean) * rstd * g.z + b.z;
            y.w = (v[c].w - mean) * rstd * g.w + b.w;
            if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + orow * p.ldo + ch * 4) = y;
            if (p.out_act) {
                TAct* o = reinterpret_cast<TAct*>(p.out_act) + orow * p.ldo + ch * 4;
                if constexpr (sizeof(TAct) == 2) {
                    bf16x4 pk = {(bf16_t)y.x, (bf16_t)y.y, (bf16_t)y.z, (bf16_t)y.w};
                    *reinterpret_cast<bf16x4*>(o) = pk;
                } else {
                    *reinterpret_cast<float4*>(o) = y;
                }
            }
        }
    }
    }
}

int launch_layernorm(const LnDesc& d, hipStream_t s) {
    VTGB_REQUIRE(d.x && d.gamma && d.beta && (d.out_f32 || d.out_act), VTGB_EINVAL, "layernorm: NULL operand");
    VTGB_REQUIRE(d.M > 0 && d.D > 0 && (d.D % 4) == 0 && d.D <= 2048 && (d.ldx % 4) == 0 && (d.ldo % 4) == 0, VTGB_EUNSUPPORTED,
                 "layernorm: D=%d must be a multiple of 4, <= 2048 (ldx=%lld ldo=%lld)", d.D, (long long)d.ldx, (long long)d.ldo);
    const int64_t blocks = (d.M + 3) / 4;
    dim3 grid((unsigned)(blocks < 256 * 10 ? blocks : 256 * 10));      // 10 resident workgroups per CU (16 KB of LDS each), each walking rows
    if (d.dtype == VTGB_BF16)
        hipLaunchKernelGGL(layernorm_kernel<bf16_t>, grid, dim3(256), 0, s, d);
    else
        hipLaunchKernelGGL(layernorm_kernel<float>, grid, dim3(256), 0, s, d);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---------------------------------------------------------------------------------------
// im2col for a kernel==stride convolution: out[(img, py, px)][c*P*P + ky*P + kx], zero padded
// to kpad columns.  (ViT patch embedding, xinstructblip.py:113-117.)
// ---------------------------------------------------------------------------------------
template <typename TAct>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ pix, TAct* __restrict__ out, int n_img, int ch,
                                                     int image, int patch, int kpad) {
    const int grid_sz = image / patch;
    const int64_t row = blockIdx.x;  // (img, py, px)
    const int px = row % grid_sz, py = (row / grid_sz) % grid_sz;
    const int64_t img = row / (grid_sz * grid_sz);
    const int kreal = ch * patch * patch;
    for (int k = threadIdx.x; k < kpad; k += blockDim.x) {
        float v = 0.f;
        if (k < kreal) {
            const int c = k / (patch * patch), rem = k - c * patch * patch, ky = rem / patch, kx = rem - ky * patch;
            v = pix[((img * ch + c) * image + (py * patch + ky)) * image + px * patch + kx];
        }
        out[row * kpad + k] = (TAct)v;
    }
}

int launch_im2col(int dtype, const float* pix, void* out, int n_img, int ch, int image, int patch, int kpad, hipStream_t s) {
    const int g = image / patch;
    dim3 grid((unsigned)((int64_t)n_img * g * g));
    if (dtype == VTGB_BF16)
        hipLaunchKernelGGL(im2col_kernel<bf16_t>, grid, dim3(256), 0, s, pix, (bf16_t*)out, n_img, ch, image, patch, kpad);
    else
        hipLaunchKernelGGL(im2col_kernel<float>, grid, dim3(256), 0, s, pix, (float*)out, n_img, ch, image, patch, kpad);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// x[f, 0, :] = class_embedding + position_embedding[0]   (xinstructblip.py:119-121)
__global__ void vit_cls_kernel(const float* cls, const float* pos, float* x, int n_frames, int tokens, int hidden) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_frames * hidden) return;
    const int f = i / hidden, d = i % hidden;
    x[(int64_t)f * tokens * hidden + d] = cls[d] + pos[d];
}
int launch_vit_cls_rows(const float* cls, const float* pos, float* x, int n_frames, int tokens, int hidden, hipStream_t s) {
    const int64_t n = (int64_t)n_frames * hidden;
    hipLaunchKernelGGL(vit_cls_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cls, pos, x, n_frames, tokens, hidden);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

template <typename TAct>
__global__ void cast_kernel(const float* __restrict__ src, TAct* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (TAct)src[i];
}
int launch_cast_act(int dtype, const float* src, void* dst, int64_t n, hipStream_t s) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == VTGB_BF16)
        hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, src, (bf16_t*)dst, n);
    else
        hipLaunchKernelGGL(cast_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, src, (float*)dst, n);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// additive mask: out = (1 - mask) * neg      (xropebert.py:1044-1045 / HF invert_attention_mask)
__global__ void mask_kernel(const int64_t* mask, float* out, int64_t n, float neg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (1.0f - (float)mask[i]) * neg;
}
int launch_mask_to_additive(const int64_t* mask, float* out, int64_t n, float neg, hipStream_t s) {
    hipLaunchKernelGGL(mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mask, out, n, neg);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

__global__ void fill_kernel(float* dst, float v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}
int launch_fill_f32(float* dst, float v, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, v, n);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// Q-Former input rows before LayerNorm (xinstructblip.py:1033-1042):
// x[f, r<nq] = query_tokens[r];  x[f, nq+t] = word_emb[ids[f,t]] + pos_emb[t]
__global__ void qformer_embed_kernel(const float* query, const int64_t* ids, const float* wemb, const float* pemb, float* x,
                                     int n_frames, int n_query, int n_text, int hidden) {
    const int S = n_query + n_text;
    const int64_t row = blockIdx.x;
    const int f = row / S, r = row % S;
    const float* src;
    const float* add = nullptr;
    if (r < n_query) {
        src = query + (int64_t)r * hidden;
    } else {
        const int t = r - n_query;
        src = wemb + ids[(int64_t)f * n_text + t] * hidden;
        add = pemb + (int64_t)t * hidden;
    }
    for (int d = threadIdx.x; d < hidden; d += blockDim.x) x[row * hidden + d] = src[d] + (add ? add[d] : 0.f);
}
int launch_qformer_embed(const float* query, const int64_t* ids, const float* wemb, const float* pemb, float* x, int n_frames,
                         int n_query, int n_text, int hidden, hipStream_t s) {
    hipLaunchKernelGGL(qformer_embed_kernel, dim3((unsigned)((int64_t)n_frames * (n_query + n_text))), dim3(256), 0, s, query, ids,
                       wemb, pemb, x, n_frames, n_query, n_text, hidden);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// TGB question rows before LayerNorm (xropebert.py:197-204): word_emb[id] + token_type_emb[0]
__global__ void tgb_text_embed_kernel(const int64_t* ids, const float* wemb, const float* temb, float* x, int hidden) {
    const int64_t row = blockIdx.x;
    const float* src = wemb + ids[row] * hidden;
    for (int d = threadIdx.x; d < hidden; d += blockDim.x) x[row * hidden + d] = src[d] + temb[d];
}
int launch_tgb_text_embed(const int64_t* ids, const float* wemb, const float* temb, float* x, int64_t rows, int hidden, hipStream_t s) {
    hipLaunchKernelGGL(tgb_text_embed_kernel, dim3((unsigned)rows), dim3(256), 0, s, ids, wemb, temb, x, hidden);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---------------------------------------------------------------------------------------
// TGB flow embedding, step 1 (xropebert.py:106-114).  Conv2d(k=stride=P) followed by
// Linear(n_patches -> 1) over the patch axis is linear in the flow, so the patch axis is
// reduced FIRST:  red[img][c, ky, kx] = sum_p fc_w[p] * of[img, c, py*P+ky, px*P+kx]
// and the convolution becomes a [n_img, 2*P*P] x [2*P*P, hidden] GEMM (196x fewer FLOPs; the
// flow, the only large operand, is read exactly once: the kernel is HBM-bound).
// One workgroup per (img, channel, ky): each lane accumulates 4 contiguous pixels of the
// 14 image rows py*P+ky; reduction over px within the row is done through LDS.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void flow_reduce_kernel(const float* __restrict__ of, const float* __restrict__ fcw,
                                                         float* __restrict__ red, int image, int patch) {
    __shared__ float part[64 * 4];
    const int g = image / patch;               // patches per side
    const int ky = blockIdx.x % patch;
    const int64_t plane = blockIdx.x / patch;  // img * 2 + c
    const int lane = threadIdx.x;              // image/4 float4 columns per row (56 for 224)
    const int ncol4 = image >> 2;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < ncol4) {
        const int px = (lane * 4) / patch;
        for (int py = 0; py < g; py++) {
            const float w = fcw[py * g + px];
            const float4 v = *reinterpret_cast<const float4*>(of + (plane * image + (py * patch + ky)) * image + lane * 4);
            acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
        }
    }
    part[lane * 4 + 0] = acc.x; part[lane * 4 + 1] = acc.y; part[lane * 4 + 2] = acc.z; part[lane * 4 + 3] = acc.w;
    __syncthreads();
    if (lane < patch) {   // kx = lane: sum over px of column px*P + kx
        float s = 0.f;
        for (int px = 0; px < g; px++) s += part[px * patch + lane];
        red[(plane * patch + ky) * patch + lane] = s;
    }
}
int launch_flow_reduce(const float* of, const float* fcw, float* red, int n_img, int image, int patch, hipStream_t s) {
    VTGB_REQUIRE((image % patch) == 0 && (patch % 4) == 0 && image / 4 <= 64 && patch <= 64, VTGB_EUNSUPPORTED,
                 "flow embed: image=%d patch=%d unsupported", image, patch);
    hipLaunchKernelGGL(flow_reduce_kernel, dim3((unsigned)((int64_t)n_img * 2 * patch)), dim3(64), 0, s, of, fcw, red, image, patch);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// step 2 (xropebert.py:115-125): rows 1..L get conv + proj_b*sum(fc_w) + fc_b; row 0 = bos; row L+1 = 0;
// row ends[b] = eos (ends = of_mask.sum - 1); then += frame_pos_embed[row].  LayerNorm follows.
__global__ void flow_assemble_kernel(const float* conv, const float* proj_b, const float* fcw, const float* fcb,
                                     const float* bos, const float* eos, const float* pos, const int64_t* of_mask, float* x,
                                     int B, int L, int hidden, int n_patches) {
    const int S = L + 2;
    const int64_t row = blockIdx.x;
    const int b = row / S, r = row % S;
    __shared__ int ends_s;
    __shared__ float sfc_s;
    if (threadIdx.x == 0) {
        int64_t e = 0;
        for (int i = 0; i < S; i++) e += of_mask[(int64_t)b * S + i];
        ends_s = (int)e - 1;
        float sf = 0.f;
        for (int i = 0; i < n_patches; i++) sf += fcw[i];
        sfc_s = sf;
    }
    __syncthreads();
    int ends = ends_s;
    if (ends < 0) ends += S;   // python negative index
    for (int d = threadIdx.x; d < hidden; d += blockDim.x) {
        float v;
        if (r == ends) v = eos[d];
        else if (r == 0) v = bos[d];
        else if (r == S - 1) v = 0.f;
        else v = conv[((int64_t)b * L + (r - 1)) * hidden + d] + proj_b[d] * sfc_s + fcb[0];
        x[row * hidden + d] = v + pos[(int64_t)r * hidden + d];
    }
}
int launch_flow_assemble(const float* conv, const float* proj_b, const float* fcw, const float* fcb, const float* bos,
                         const float* eos, const float* pos, const int64_t* of_mask, float* x, int B, int L, int hidden,
                         int n_patches, hipStream_t s) {
    hipLaunchKernelGGL(flow_assemble_kernel, dim3((unsigned)((int64_t)B * (L + 2))), dim3(256), 0, s, conv, proj_b, fcw, fcb, bos, eos,
                       pos, of_mask, x, B, L, hidden, n_patches);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// mean over `width` consecutive frames of [*, row_elems] rows; width 0 -> zeros
__global__ void mean_pool_kernel(const float* __restrict__ q, float* __restrict__ out, int width, int64_t row_elems) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_elems) return;
    const int64_t clip = blockIdx.y;
    const float* src = q + clip * width * row_elems + i;
    float acc = 0.f;
    for (int f = 0; f < width; f++) acc += src[(int64_t)f * row_elems];
    out[clip * row_elems + i] = width > 0 ? acc / (float)width : 0.f;
}
int launch_mean_pool_uniform(const float* q, float* out, int n_clips, int width, int64_t row_elems, hipStream_t s) {
    hipLaunchKernelGGL(mean_pool_kernel, dim3((unsigned)((row_elems + 255) / 256), n_clips), dim3(256), 0, s, q, out, width, row_elems);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// logits[b, l, j] = x[b, 1 + l, :] . w[j, :] + bias[j]     (xropebert.py:1164): one wave per (b, l)
__global__ __launch_bounds__(256) void mrc_head_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ logits, int B, int L,
                                                       int hidden) {
    const int lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= (int64_t)B * L) return;
    const int b = idx / L, l = idx % L;
    const float* xr = x + ((int64_t)b * (L + 2) + 1 + l) * hidden;
    float a0 = 0.f, a1 = 0.f;
    for (int d = lane; d < hidden; d += 64) {
        a0 = fmaf(xr[d], w[d], a0);
        a1 = fmaf(xr[d], w[hidden + d], a1);
    }
    for (int off = 32; off > 0; off >>= 1) {
        a0 += __shfl_xor(a0, off);
        a1 += __shfl_xor(a1, off);
    }
    if (lane == 0) {
        logits[idx * 2 + 0] = a0 + bias[0];
        logits[idx * 2 + 1] = a1 + bias[1];
    }
}
int launch_mrc_head(const float* x, const float* w, const float* b, float* logits, int B, int L, int hidden, hipStream_t s) {
    hipLaunchKernelGGL(mrc_head_kernel, dim3((unsigned)(((int64_t)B * L + 3) / 4)), dim3(256), 0, s, x, w, b, logits, B, L, hidden);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// fp32 -> bf16 weight packing with zero column padding
__global__ void pack_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows, int64_t cols, int64_t cols_pad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols_pad; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols_pad, c = i - r * cols_pad;
        dst[i] = c < cols ? (bf16_t)src[r * cols + c] : (bf16_t)0.f;
    }
}
int launch_pack_bf16(const float* src, void* dst, int64_t rows, int64_t cols, int64_t cols_pad, hipStream_t s) {
    int64_t blocks = (rows * cols_pad + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, (bf16_t*)dst, rows, cols, cols_pad);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}


// ---------------------------------------------------------------------------------------
// LayerNorm folded into the GEMMs around it (GemmDesc::ln_*, r5)
// ---------------------------------------------------------------------------------------
// (mean, rstd) of every row from the (sum, sum of squares) of its 64-column blocks, added in block order
__global__ void ln_fold_stats_kernel(const float* __restrict__ part, int nblk, float inv_d, float eps, float* __restrict__ stats, int64_t M) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float s = 0.f, q = 0.f;
    for (int b = 0; b < nblk; b++) {
        const float2 v = *reinterpret_cast<const float2*>(part + (m * nblk + b) * 2);
        s += v.x; q += v.y;
    }
    const float mean = s * inv_d;
    const float var = fmaxf(fmaf(-mean, mean, q * inv_d), 0.f);
    *reinterpret_cast<float2*>(stats + m * 2) = make_float2(mean, rsqrtf(var + eps));
}
int launch_ln_fold_stats(const float* part, int nblk, int D, float eps, float* stats, int64_t M, hipStream_t s) {
    hipLaunchKernelGGL(ln_fold_stats_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, part, nblk, 1.0f / (float)D, eps, stats, M);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// the first LayerNorm of a stack (its input comes from the embedding, not from a residual GEMM): bf16 copy of x and the rows' block moments, in the
// SAME association as the GEMM epilogues produce them (four columns per lane, a 16-lane butterfly per 64-column block: gemm_dev.h ln_part4)
__global__ __launch_bounds__(256) void ln_fold_prepare_kernel(const float* __restrict__ x, int64_t ldx, int D, bf16_t* __restrict__ xb, float* __restrict__ part, int64_t M) {
    const int lane = threadIdx.x & 63, cl = lane & 15, sub = lane >> 4;
    const int64_t m = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub;      // a wave: four rows, sixteen lanes each
    const int nblk = (D + 63) >> 6;
    for (int b = 0; b < nblk; b++) {
        const int n = b * 64 + cl * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < M && n < D) {
            v = *reinterpret_cast<const f32x4*>(x + m * ldx + n);
            *reinterpret_cast<bf16x4*>(xb + m * D + n) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
        float s_ = (v[0] + v[1]) + (v[2] + v[3]);
        float q_ = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { s_ += __shfl_xor(s_, off); q_ += __shfl_xor(q_, off); }
        if (m < M && cl == 0) *reinterpret_cast<float2*>(part + (m * nblk + b) * 2) = make_float2(s_, q_);
    }
}
int launch_ln_fold_prepare(const float* x, int64_t ldx, int D, void* xb, float* part, int64_t M, hipStream_t s) {
    VTGB_REQUIRE((D & 3) == 0 && (ldx & 3) == 0, VTGB_EINVAL, "ln_fold_prepare: 4-aligned rows");
    hipLaunchKernelGGL(ln_fold_prepare_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, s, x, ldx, D, (bf16_t*)xb, part, M);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
