// raft_enc.hip -- RAFT's BasicEncoder (raft_utils/extractor.py:116-189: 7x7/2 stem, six ResidualBlocks
// :6-56 at 1/2, 1/4, 1/8 resolution, 1x1 head) on gfx950, the remaining piece of SURVEY.md row f1.
// NHWC bf16 activations (96-channel stages padded to 128), every 3x3 / 1x1 convolution an implicit GEMM on
// the MFMA kernel of gemm.hip (stride 1 and 2) with fp32 outputs; the norm that follows each convolution
// (InstanceNorm2d for fnet, eval-mode BatchNorm2d for cnet -- folded into the packed weights by the
// caller) is applied by a separate HBM-bound pass that also does ReLU, the residual add and the bf16 cast.
// The stem (3 input channels) runs on the same kernel after a space-to-depth repack (see below).
#include <math.h>
#include <string.h>

#include "common.h"

// ---- stem: conv 7x7 stride 2 pad 3, 3 -> 64, on 2*(x/255)-1 (xraft.py:105-106), as an implicit GEMM.
// Space-to-depth: the image is repacked (NHWC at half resolution) so that pixel (Y, X) carries the 4 x 2 x 2 x 3
// values img[c][2Y + py][2(X + dX - 2) + px], dX = 0..3 -- 48 channels padded to 64 -- and the stride-2 7x7
// stencil becomes a 4 x 1 stride-1 convolution (vertical taps dY = -2..1; 147 of the 4 * 48 products non-zero).
//
// fp32 mode packs the reference's own normalised value 2 * (x / 255) - 1 (same operations, same roundings).
//
// bf16 mode packs v = x - 127.5 as a bf16 PAIR: hi = bf16(v) in channel chunk 0, lo = bf16(v - hi) in chunk 1 (K = 4 * 128;
// the packed weights w * 2/255 appear in both chunks, the bias is untouched).  hi + lo carries 16 significant bits of v.
// For integer frames 0..255 (RAFT's own convention) lo = 0 and hi is exact.  The eval path, however, hands RAFT
// CLIP-normalised floats (eval/inference.py:68 -> eval/utils/model.py:79), |x| < 3: there v = -127.5 + x sits where bf16
// has a step of 0.5 .. 1, and hi alone (rounds 1 and 2) kept one or two bits of x -- InstanceNorm then removes the constant and
// amplifies what is left: 7 % feature error in fnet (round-2 VERDICT).  With the pair the input error is <= 2^-9 of |v - hi| <= 0.5,
// i.e. <= 1e-3 absolute on x -- below one bf16 rounding of any later activation.
// An out-of-image tap is 0 in either encoding (= normalised 0) -- horizontally in the packed tensor, vertically through the
// convolution's zero padding (which the implicit-GEMM kernel gets from the buffer descriptor's range check, not from a page).
template <typename T>
__global__ __launch_bounds__(256) void raft_stem_pack_kernel(const float* __restrict__ img, T* __restrict__ out, T* __restrict__ pad_page,
                                                             int64_t n_px, int H, int W) {
    constexpr bool PAIR = sizeof(T) == 2;          // bf16: hi | lo chunks
    constexpr int CP = PAIR ? 128 : 64;            // packed channels per half-resolution pixel
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 64) pad_page[i] = (T)0.f;   // (the fp32 exactness kernel still reads its padding from a page)
    if (i >= n_px * 4) return;
    const int64_t px_i = i >> 2;
    const int dX = (int)(i & 3), W2 = W >> 1, H2 = H >> 1;
    const int X = (int)(px_i % W2), Y = (int)((px_i / W2) % H2);
    const int64_t n = px_i / ((int64_t)W2 * H2);
    T v[12], l[12];
#pragma unroll
    for (int py = 0; py < 2; py++)
#pragma unroll
        for (int pxx = 0; pxx < 2; pxx++) {
            const int col = 2 * (X + dX - 2) + pxx, row = 2 * Y + py;
            const bool ok = (unsigned)col < (unsigned)W;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int e = py * 6 + pxx * 3 + c;
                const float raw = ok ? img[((n * 3 + c) * H + row) * W + col] : 0.f;
                if constexpr (PAIR) {
                    const float d = ok ? raw - 127.5f : 0.f;
                    const float hf = bf16_round(d);      // (on the bits: see common.h)
                    v[e] = (T)hf;
                    l[e] = (T)(d - hf);
                } else {
                    v[e] = (T)(ok ? 2.0f * (raw / 255.0f) - 1.0f : 0.f);
                }
            }
        }
    typedef T T4 __attribute__((ext_vector_type(4)));
    T* o = out + px_i * CP + dX * 12;
#pragma unroll
    for (int q = 0; q < 3; q++) *reinterpret_cast<T4*>(o + q * 4) = T4{v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]};
    if constexpr (PAIR) {
#pragma unroll
        for (int q = 0; q < 3; q++) *reinterpret_cast<T4*>(o + 64 + q * 4) = T4{l[q * 4], l[q * 4 + 1], l[q * 4 + 2], l[q * 4 + 3]};
    }
    if (dX == 3) {
        const T4 z = {(T)0.f, (T)0.f, (T)0.f, (T)0.f};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            *reinterpret_cast<T4*>(out + px_i * CP + 48 + q * 4) = z;
            if constexpr (PAIR) *reinterpret_cast<T4*>(out + px_i * CP + 112 + q * 4) = z;
        }
    }
}

// ---- InstanceNorm2d statistics: per (image, channel) sum and sum of squares over the HW pixels.  r5: DETERMINISTIC -- every producer
// (the implicit-GEMM epilogues: one slot per 256-row tile; conv64.hip's kernels: one slot per run of rows; the separate pass below: one
// slot per slice) STORES its partial moments and a second pass adds an image's slots in a fixed order.  Rounds 1-4 accumulated them with
// atomics: the order, and with it the last bits of every normalised feature, changed from run to run, which the bf16 TGB downstream could
// turn into a different frame selection.
// This pass serves the stem of the fp32 mode and images of fewer than 256 pixels: grid (n, splits), each workgroup reduces a slice of the
// image into part[((n * splits + split) * C + c) * 2 + {0, 1}].
__global__ __launch_bounds__(256) void inorm_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int HW, int C, int ldx) {
    __shared__ float s1[256], s2[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int c4 = C >> 2, cq = tid % c4, lane_p = tid / c4, np = 256 / c4;   // a thread owns 4 channels of every np-th pixel
    const int per = (HW + gridDim.y - 1) / gridDim.y, p0 = blockIdx.y * per, p1 = min(HW, p0 + per);
    float a[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (lane_p < np) {
        const float* base = x + (int64_t)n * HW * ldx + cq * 4;
        for (int p = p0 + lane_p; p < p1; p += np) {
            const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)p * ldx);
            a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
            q[0] = fmaf(v.x, v.x, q[0]); q[1] = fmaf(v.y, v.y, q[1]); q[2] = fmaf(v.z, v.z, q[2]); q[3] = fmaf(v.w, v.w, q[3]);
        }
    }
    for (int e = 0; e < 4; e++) {
        __syncthreads();
        s1[tid] = a[e]; s2[tid] = q[e];
        __syncthreads();
        if (tid < c4) {
            float sa = 0.f, sq = 0.f;
            for (int l = 0; l < np; l++) { sa += s1[l * c4 + tid]; sq += s2[l * c4 + tid]; }
            float* st = part + (((int64_t)n * gridDim.y + blockIdx.y) * C + tid * 4 + e) * 2;
            st[0] = sa;
            st[1] = sq;
        }
    }
}

// stats[(img * C + c) * 2 + w] = sum over the image's `parts` slots, in slot order
__global__ void stats_finish_parts_kernel(const float* __restrict__ part, float* __restrict__ stats, int64_t total, int parts, int C2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (img, c, w)
    if (i >= total) return;
    const int64_t img = i / C2;
    const int cw = (int)(i - img * C2);
    float t = 0.f;
    for (int pt = 0; pt < parts; pt++) t += part[(img * parts + pt) * C2 + cw];
    stats[i] = t;
}
int launch_stats_finish_parts(const float* part, float* stats, int n_img, int parts, int C, hipStream_t s) {
    const int64_t total = (int64_t)n_img * C * 2;
    hipLaunchKernelGGL(stats_finish_parts_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, part, stats, total, parts, C * 2);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
// the GEMM epilogues' slots part[(m_tile * N + n) * 4 + {sum a, sq a, sum b, sq b}] (256-row tiles; a = the image of the tile's first
// row, b = the next one): stats[(img * N + n) * 2 + w] = the sum over the tiles that touch the image, in tile order
__global__ void stats_finish_tiles_kernel(const float* __restrict__ part, float* __restrict__ stats, int n_img, int HW, int N, int64_t M) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (img, n)
    if (i >= (int64_t)n_img * N) return;
    const int img = (int)(i / N), n = (int)(i - (int64_t)img * N);
    const int64_t r0 = (int64_t)img * HW, r1 = r0 + HW;                    // the image's rows
    const int t0 = (int)(r0 >> 8), t1 = (int)((r1 - 1) >> 8);
    float sa = 0.f, sq = 0.f;
    for (int t = t0; t <= t1; t++) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)t * N + n) * 4);
        const int img_a = (int)(((int64_t)t << 8) / HW);
        if (img_a == img) { sa += v.x; sq += v.y; }
        else { sa += v.z; sq += v.w; }                                      // (stats_rows >= 256: a tile touches at most two images)
    }
    stats[i * 2] = sa;
    stats[i * 2 + 1] = sq;
}
int launch_stats_finish_tiles(const float* part, float* stats, int n_img, int HW, int N, int64_t M, hipStream_t s) {
    VTGB_REQUIRE(HW >= 256, VTGB_EINVAL, "stats_finish_tiles: images of >= 256 rows");
    const int64_t total = (int64_t)n_img * N;
    hipLaunchKernelGGL(stats_finish_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, part, stats, n_img, HW, N, M);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---- y = [relu]((x - mean) * rstd); out = [relu](resid + y); bf16 NHWC with Cpad channels (pad = 0).
// mean = sum / HW, rstd = 1 / sqrt(biased var + 1e-5) from the accumulated moments; 4 channels per thread.
// r4: a workgroup belongs to ONE image (blockIdx.y) and a thread keeps its four channels over NA_PASS pixels: the statistics are
// turned into mean / rstd once per thread instead of once per 16 bytes, the loads of all its pixels are issued before the first is
// used, and the two 64-bit divisions per thread (pixel and image from a flat index: ~100 of the old form's ~290 vector
// instructions per 16 bytes of input) are gone.  Same arithmetic per element: results are bit-identical.
constexpr int NA_PASS = 4;
template <typename T>
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, const T* __restrict__ resid,
                                                         T* __restrict__ out, int HW, int C, int Cpad, int ldx, int relu_inner, int relu_outer) {
    typedef T T4 __attribute__((ext_vector_type(4)));
    const int sh = Cpad == 64 ? 4 : 5, cp4 = 1 << sh, ppb = 256 >> sh;   // Cpad / 4 threads per pixel; pixels per pass (Cpad = 64: 16, 128: 8)
    const int tid = threadIdx.x, c = (tid & (cp4 - 1)) * 4, pl = tid >> sh;
    const int n = blockIdx.y, p0 = blockIdx.x * (NA_PASS * ppb) + pl;
    const int64_t m0 = (int64_t)n * HW;
    float mean[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4] = {1.f, 1.f, 1.f, 1.f};
    const bool live = c < C;
    if (live && stats) {
        const float inv = 1.0f / (float)HW;
        const float4 s01 = *reinterpret_cast<const float4*>(stats + ((int64_t)n * C + c) * 2);
        const float4 s23 = *reinterpret_cast<const float4*>(stats + ((int64_t)n * C + c) * 2 + 4);
        const float sm[4] = {s01.x, s01.z, s23.x, s23.z}, sq[4] = {s01.y, s01.w, s23.y, s23.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            mean[e] = sm[e] * inv;
            rstd[e] = rsqrtf(fmaxf(sq[e] * inv - mean[e] * mean[e], 0.f) + 1e-5f);
        }
    }
    float4 xv[NA_PASS];
    T4 rv[NA_PASS];
#pragma unroll
    for (int j = 0; j < NA_PASS; j++) {
        const int px = p0 + j * ppb;
        const int pc = px < HW ? px : HW - 1;                           // (clamped: the loads stay unconditional)
        xv[j] = live ? *reinterpret_cast<const float4*>(x + (m0 + pc) * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (resid && live) rv[j] = *reinterpret_cast<const T4*>(resid + (m0 + pc) * Cpad + c);
    }
#pragma unroll
    for (int j = 0; j < NA_PASS; j++) {
        const int px = p0 + j * ppb;
        float v[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
        if (live) {
            if (stats) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = (v[e] - mean[e]) * rstd[e];
            }
            if (relu_inner) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
            }
            if (resid) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] += (float)rv[j][e];
            }
            if (relu_outer) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
            }
        }
        const T4 o = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
        if (px < HW) *reinterpret_cast<T4*>(out + (m0 + px) * Cpad + c) = o;
    }
}

// VTGB_BF16X3: the input is a pair row [hi(Cin) | lo(Cin)] contracted as [hi | lo | hi] against weights packed [Wh | Wh | Wl] (raft_x3.hip)
static GemmDesc enc_conv(int dt, int Mo, int N, int Ho, int Wo, int K, int Cin, int stride, int Hi, int Wi, const void* A, const void* Wt, const float* bias,
                         float* out, int ldo, const void* zero, float* col_stats) {
    GemmDesc d;
    memset(&d, 0, sizeof(d));
    const bool x3 = dt == VTGB_BF16X3;
    const int Ce = x3 ? 3 * Cin : Cin;
    d.dtype = x3 ? VTGB_BF16 : dt; d.M = Mo; d.N = N; d.K = K * K * Ce; d.epi = VTGB_EPI_STORE_F32;
    d.A = A; d.lda = x3 ? 2 * Cin : Cin; d.W = Wt; d.ldw = d.K; d.bias = bias; d.out = out; d.ldo = ldo;
    d.conv_H = Ho; d.conv_W = Wo; d.conv_KH = K; d.conv_KW = K; d.conv_Cin = Ce; d.conv_split = Ce; d.conv_wrap = x3 ? 2 * Cin : 0;
    d.conv_stride = stride; d.conv_Hi = Hi; d.conv_Wi = Wi; d.zero_page = zero;
    d.col_stats = col_stats; d.stats_rows = Ho * Wo;
    if (x3) d.algo_flops = 2.0 * Mo * (double)N * (K * K * Cin);
    return d;
}
int launch_x3_pair_pass(const float* x, int64_t ldx, const float* stats, int HW, const void* resid, int64_t ldr, int r_lo, void* out, int64_t ldo, int o_lo,
                        int C, int Cpad, int relu_in, int relu_out, int64_t M, hipStream_t s, int h8 = 0, int resid_h8 = 0);   // raft_x3.hip

static int enc_impl(const vtgb_raft_encoder_args* a, Workspace& ws, hipStream_t s) {
    VTGB_REQUIRE(a, VTGB_EINVAL, "raft_encoder: NULL args");
    VTGB_REQUIRE(a->n_images > 0 && a->H >= 64 && a->W >= 64 && (a->H % 8) == 0 && (a->W % 8) == 0 && (a->norm == 0 || a->norm == 1), VTGB_EINVAL,
                 "raft_encoder: bad dims n=%d H=%d W=%d", a->n_images, a->H, a->W);
    VTGB_REQUIRE(a->dtype == VTGB_BF16 || a->dtype == VTGB_F32 || a->dtype == VTGB_BF16X3 || a->dtype == VTGB_F16C8, VTGB_EINVAL, "raft_encoder: bad dtype %d", a->dtype);
    // VTGB_F16C8 (round 6): the bf16x3 encoder with the stride-1 3x3 convolutions of its residual blocks -- 10 of the 12, 87 % of the encoder's FLOPs -- on
    // f16c8 operands (gemm_h8.hip: 2 k-tiles per channel chunk and tap instead of 3; tests/emul_f16c8.py: flows 1.35e-4 vs 1.26e-4 rel-RMS from fp32 with
    // ALL of layer1-3 of both encoders in this form).  The stem, the stride-2 convolutions, the 1x1 downsamples and the 1x1 head -- fnet's head feeds the
    // correlation directly -- stay bf16x3.  weights[40] = device int32 [12]: the scale bytes of block b's conv1 / conv2 at [2 b] / [2 b + 1]
    const bool h8l1 = a->dtype == VTGB_F16C8;
    const int n = a->n_images, dt = h8l1 ? VTGB_BF16X3 : a->dtype;
    const bool x3 = dt == VTGB_BF16X3;                 // activations as pairs: 4 bytes per channel
    const size_t es = x3 ? 4 : dtype_size(dt);
    const int H2 = a->H / 2, W2 = a->W / 2, H4 = a->H / 4, W4 = a->W / 4, H8 = a->H / 8, W8 = a->W / 8;
    const int64_t M2 = (int64_t)n * H2 * W2, M4 = (int64_t)n * H4 * W4, M8 = (int64_t)n * H8 * W8;
    VTGB_REQUIRE(M2 < (1ll << 31), VTGB_EUNSUPPORTED, "raft_encoder: too many pixels per call (chunk the images)");
    float* cf = (float*)ws.take(M2 * 64 * 4);          // fp32 conv output (largest stage; later stages reuse it)
    float* cf2 = (float*)ws.take(M4 * 128 * 4);        // fp32 output of the downsample branch
    void* act0 = ws.take(M2 * 64 * es);   // activations: bf16 (VTGB_BF16) or fp32 (VTGB_F32)
    char* act12 = (char*)ws.take(2 * M2 * 64 * es);   // act1 | act2, contiguous: the bf16 stem's 128-channel (hi | lo) input spans both
    void* act1 = act12;
    void* act2 = act12 + (ws.dry ? 0 : M2 * 64 * es);
    float* stats = (float*)ws.take((int64_t)n * 128 * 2 * 4);
    float* stats2 = (float*)ws.take((int64_t)n * 128 * 2 * 4);   // the downsample branch's moments
    // partial moments (one producer at a time): a slot per 256-row GEMM tile / per conv64 run of rows / per slice of the separate pass
    size_t part_floats = (size_t)(M2 / 256 + 2) * 128 * 4;
    if (conv64_stats_part_floats(n) > part_floats) part_floats = conv64_stats_part_floats(n);
    if ((size_t)n * 16 * 128 * 2 > part_floats) part_floats = (size_t)n * 16 * 128 * 2;
    float* spart = (float*)ws.take(part_floats * 4);
    void* zero = ws.take(256);
    void* pad_page = ws.take(256);   // the stem's out-of-image value (0 = normalised 0 in the raw - 127.5 encoding)
    if (ws.dry) return VTGB_OK;
    VTGB_REQUIRE(ws.ok(), VTGB_EWORKSPACE, "raft_encoder: workspace %zu < %zu bytes", ws.size, ws.used);
    VTGB_REQUIRE(a->images && a->weights && a->out, VTGB_EINVAL, "raft_encoder: NULL operand");
    const void* const* w = a->weights;
    auto F = [](const void* p) { return (const float*)p; };
    const bool inorm = a->norm == 0;
    VTGB_HIP(hipMemsetAsync(zero, 0, 256, s));

    // `fused`: the convolution that produced x has already accumulated the moments (stats zeroed before it)
    auto norm = [&](const float* x, int64_t M, int HW, int C, int Cpad, int ldx, const void* resid, void* out, int relu_in, int relu_out,
                    float* st, bool fused, int out_h8 = 0, int resid_h8 = 0) -> int {
        if (inorm && !fused) {
            const int splits = HW >= 4096 ? 16 : HW >= 1024 ? 4 : 1;
            hipLaunchKernelGGL(inorm_stats_kernel, dim3(n, splits), dim3(256), 0, s, x, spart, HW, C, ldx);
            VTGB_TRY(launch_stats_finish_parts(spart, st, n, splits, C, s));
        }
        if (x3)
            return launch_x3_pair_pass(x, ldx, inorm ? st : nullptr, HW, resid, 2 * Cpad, Cpad, out, 2 * Cpad, Cpad, C, Cpad, relu_in, relu_out, M, s, out_h8, resid_h8);
        VTGB_REQUIRE(Cpad == 64 || Cpad == 128, VTGB_EUNSUPPORTED, "raft_encoder: normalisation pass over %d padded channels", Cpad);
        const int ppb = 256 / (Cpad / 4);
        VTGB_REQUIRE(M / HW <= 65535, VTGB_EUNSUPPORTED, "raft_encoder: %lld images per call exceed the normalisation pass's grid (chunk the images)", (long long)(M / HW));
        const dim3 grid((unsigned)((HW + NA_PASS * ppb - 1) / (NA_PASS * ppb)), (unsigned)(M / HW));
        if (dt == VTGB_BF16)
            hipLaunchKernelGGL(norm_apply_kernel<bf16_t>, grid, dim3(256), 0, s, x, inorm ? st : nullptr, (const bf16_t*)resid, (bf16_t*)out, HW, C, Cpad, ldx,
                               relu_in, relu_out);
        else
            hipLaunchKernelGGL(norm_apply_kernel<float>, grid, dim3(256), 0, s, x, inorm ? st : nullptr, (const float*)resid, (float*)out, HW, C, Cpad, ldx,
                               relu_in, relu_out);
        VTGB_HIP(hipGetLastError());
        return VTGB_OK;
    };
    // moments for a convolution output: the GEMM epilogue writes per-tile partial moments into `spart` when the image is large enough for the
    // two-images-per-tile bookkeeping (conv_stats then adds them into st in tile order), else they are left to the separate pass
    auto stats_for = [&](float* st, int HW, int C) -> float* {
        if (!inorm || HW < 256 || dt == VTGB_F32) return nullptr;   // (the fp32 kernel leaves the moments to the separate pass)
        return spart;
    };
    auto conv_stats = [&](const GemmDesc& d, float* st) -> int {    // the convolution, then its moments (if it produced any) into st
        VTGB_TRY(launch_conv_gemm(d, s));
        if (d.col_stats) VTGB_TRY(launch_stats_finish_tiles(d.col_stats, st, n, d.stats_rows, d.N, d.M, s));
        return VTGB_OK;
    };
    // ---- stem: repack, then a 4x1 implicit-GEMM convolution whose epilogue also yields the InstanceNorm moments
#ifdef VTGB_DEBUG_HOOKS
    static const bool stem_on = !(getenv("VTGB_STEM64") && getenv("VTGB_STEM64")[0] == '0');      // (VTGB_STEM64=0: pack + implicit GEMM, for A/B runs; debug-hook builds only)
#else
    constexpr bool stem_on = true;
#endif
    if (stem_on && dt == VTGB_BF16 && stem7x7_supported(a->H, a->W)) {
        // the stem on the raw frames (conv64.hip: the packed rows are built in LDS, no HBM round trip)
        if (inorm) {
            VTGB_TRY(launch_stem7x7(a->images, w[0], F(w[1]), cf, stats, spart, nullptr, n, a->H, a->W, 0, s));
            VTGB_TRY(norm(cf, M2, H2 * W2, 64, 64, 64, nullptr, act0, 1, 0, stats, true));
        } else {
            VTGB_TRY(launch_stem7x7(a->images, w[0], F(w[1]), nullptr, nullptr, nullptr, act0, n, a->H, a->W, 1, s));     // relu(bn1(conv1(x))) straight to bf16
        }
    } else {
        if (dt != VTGB_F32)
            hipLaunchKernelGGL(raft_stem_pack_kernel<bf16_t>, dim3((unsigned)((M2 * 4 + 255) / 256)), dim3(256), 0, s, a->images, (bf16_t*)act1, (bf16_t*)pad_page, M2,
                               a->H, a->W);
        else
            hipLaunchKernelGGL(raft_stem_pack_kernel<float>, dim3((unsigned)((M2 * 4 + 255) / 256)), dim3(256), 0, s, a->images, (float*)act1, (float*)pad_page, M2,
                               a->H, a->W);
        {
            float* sf = stats_for(stats, H2 * W2, 64);
            const int cp = dt == VTGB_BF16 ? 128 : 64;            // bf16: hi | lo chunks (raft_stem_pack_kernel); bf16x3: the same image read as a pair of 64
            GemmDesc d = enc_conv(dt, (int)M2, 64, H2, W2, 1, cp, 1, H2, W2, act1, w[0], F(w[1]), cf, 64, pad_page, sf);
            d.conv_KH = 4; d.K = 4 * d.conv_Cin; d.ldw = d.K;
            if (x3) d.algo_flops = 2.0 * (double)M2 * 64 * 147;
            if (!inorm && !x3) { d.epi = VTGB_EPI_STORE; d.act = 1; d.out = act0; }        // relu(bn1(conv1(x))) straight to bf16
            VTGB_TRY(conv_stats(d, stats));
            if (inorm || x3) VTGB_TRY(norm(cf, M2, H2 * W2, 64, 64, 64, nullptr, act0, 1, 0, stats, sf != nullptr, h8l1));      // (f16c8: layer1 reads f16c8 pairs)
        }
    }
    // ---- six residual blocks
    struct Stage { int C, Cpad, stride, Ho, Wo; };
    const Stage st[6] = {{64, 64, 1, H2, W2}, {64, 64, 1, H2, W2}, {96, 128, 2, H4, W4}, {96, 128, 1, H4, W4}, {128, 128, 2, H8, W8}, {128, 128, 1, H8, W8}};
    void* x = act0;
    void* t1 = act1;
    void* t2 = act2;
    int Cin_pad = 64, Hi = H2, Wi = W2;
    // cnet (norm == 1, BatchNorm folded): no statistics are needed, so ReLU, the skip connection and the bf16 cast
    // live in the convolution's epilogue and the fp32 round trip + norm pass disappear (the packed weights carry
    // C_pad output rows, the padded ones zero, so the padded channels come out as zeros)
    auto conv_bn = [&](int64_t Mo, const Stage& g, int K, int Cin, int stride, int Hi_, int Wi_, const void* A, const void* Wt, const float* bias,
                       int relu, const void* resid, int post_relu, void* out) -> int {
        GemmDesc d = enc_conv(dt, (int)Mo, g.Cpad, g.Ho, g.Wo, K, Cin, stride, Hi_, Wi_, A, Wt, bias, nullptr, g.Cpad, zero, nullptr);
        d.epi = VTGB_EPI_STORE; d.act = relu; d.out = out;
        d.resid_bf16 = resid; d.ldrb = g.Cpad; d.post_relu = post_relu;
        return launch_conv_gemm(d, s);
    };
    for (int b = 0; b < 6; b++) {
        const Stage& g = st[b];
        const void* const* bw = w + 2 + 6 * b;
        const int64_t Mo = (int64_t)n * g.Ho * g.Wo;
        const int HWo = g.Ho * g.Wo;
        if (g.stride != 1) VTGB_REQUIRE(bw[4] && bw[5], VTGB_EINVAL, "raft_encoder: block %d lacks its downsample weights", b);
        void* outb;
        // layer1 (64 -> 64 channels, stride 1) at bf16: the LDS-resident-rows kernel (conv64.hip); VTGB_CONV64=0 keeps the general kernel (A/B runs)
#ifdef VTGB_DEBUG_HOOKS
        static const bool c64_on = !(getenv("VTGB_CONV64") && getenv("VTGB_CONV64")[0] == '0');      // (debug-hook builds only)
#else
        constexpr bool c64_on = true;
#endif
        const bool c64 = c64_on && dt == VTGB_BF16 && g.C == 64 && g.Cpad == 64 && Cin_pad == 64 && g.stride == 1 && conv3x3_c64_supported(g.Ho, g.Wo);
        if (c64 && !inorm) {
            VTGB_TRY(launch_conv3x3_c64(x, bw[0], F(bw[1]), nullptr, nullptr, nullptr, t1, nullptr, n, g.Ho, g.Wo, 1, 0, s));      // y = relu(bn1(conv1(x)))
            outb = t2;
            VTGB_TRY(launch_conv3x3_c64(t1, bw[2], F(bw[3]), nullptr, nullptr, nullptr, outb, x, n, g.Ho, g.Wo, 1, 1, s));        // relu(x + relu(bn2(conv2(y))))
        } else if (c64) {
            VTGB_TRY(launch_conv3x3_c64(x, bw[0], F(bw[1]), cf, stats, spart, nullptr, nullptr, n, g.Ho, g.Wo, 0, 0, s));
            VTGB_TRY(norm(cf, Mo, HWo, g.C, g.Cpad, g.Cpad, nullptr, t1, 1, 0, stats, true));                            // y = relu(norm1(conv1(x)))
            VTGB_TRY(launch_conv3x3_c64(t1, bw[2], F(bw[3]), cf, stats, spart, nullptr, nullptr, n, g.Ho, g.Wo, 0, 0, s));
            outb = t1;
            VTGB_TRY(norm(cf, Mo, HWo, g.C, g.Cpad, g.Cpad, x, outb, 1, 1, stats, true));                                // relu(x + relu(norm2(conv2(y))))
        } else if (!inorm && !x3) {
            VTGB_TRY(conv_bn(Mo, g, 3, Cin_pad, g.stride, Hi, Wi, x, bw[0], F(bw[1]), 1, nullptr, 0, t1));        // y = relu(bn1(conv1(x)))
            const void* res = x;
            outb = t2;                                                                                          // conv2 reads t1: it cannot be the output
            if (g.stride != 1) {                                                                                // x = bn3(downsample(x))
                VTGB_TRY(conv_bn(Mo, g, 1, Cin_pad, g.stride, Hi, Wi, x, bw[4], F(bw[5]), 0, nullptr, 0, t2));
                res = t2;
                outb = x;                                                                                       // the block input is dead from here on
            }
            VTGB_TRY(conv_bn(Mo, g, 3, g.Cpad, 1, g.Ho, g.Wo, t1, bw[2], F(bw[3]), 1, res, 1, outb));             // relu(x + relu(bn2(conv2(y))))
        } else if (h8l1) {
            // VTGB_F16C8: every stride-1 3x3 convolution of the residual blocks reads f16c8 pairs (gemm_h8.hip: 2 k-tiles per 64-channel chunk and tap
            // instead of 3); outputs are fp32 (+ the InstanceNorm moments) for fnet, pairs straight from the epilogue for cnet.
            // The stride-2 blocks' conv1 and 1x1 downsample stay bf16x3 launches over the block input as a bf16 pair: they share that input and a
            // 1x1's K (2 k-tiles) is below the f16c8 k-loop's minimum.  So the pair format alternates: the first block of a layer hands an f16c8 pair to
            // the stride-1 block behind it, the second a bf16 pair to the bf16x3 launches that follow (the next layer's stride-2 block; the head).
            VTGB_REQUIRE(w[40], VTGB_EINVAL, "raft_encoder: weights[40] (the f16c8 convolutions' scale bytes) is NULL at VTGB_F16C8");
            const int C2 = 2 * g.Cpad;      // 16-bit units per pair row
            const bool s2 = g.stride != 1;
            // 3x3, stride 1, g.Cpad -> g.C channels over an f16c8 pair -> cf fp32 [Mo, ld Cpad] (+ per-tile moments)
            auto conv_h8 = [&](const void* A, const void* Wt, const float* bias, float* col_stats, int si) -> int {
                GemmDesc d;
                memset(&d, 0, sizeof(d));
                d.dtype = VTGB_BF16; d.M = (int)Mo; d.N = g.C; d.K = 9 * C2; d.epi = VTGB_EPI_STORE_F32;
                d.A = A; d.lda = C2; d.W = Wt; d.ldw = d.K; d.bias = bias; d.out = cf; d.ldo = g.Cpad;
                d.conv_H = g.Ho; d.conv_W = g.Wo; d.conv_KH = 3; d.conv_KW = 3; d.conv_Cin = C2; d.conv_split = C2; d.zero_page = zero;
                d.col_stats = col_stats; d.stats_rows = HWo;
                d.h8_run = 9 * (g.Cpad / 64); d.h8_scale = (const int*)w[40] + si;
                d.algo_flops = 2.0 * Mo * (double)g.C * (9 * g.C);
                VTGB_TRY(launch_conv_h8(d, s));
                if (col_stats) VTGB_TRY(launch_stats_finish_tiles(col_stats, stats, n, HWo, g.C, Mo, s));
                return VTGB_OK;
            };
            // the same convolution over g.Cpad output rows with ReLU (+ the block's tail relu(x + .), x an f16c8 pair) and the pair store in its epilogue (cnet)
            auto conv_pair = [&](const void* A, const void* Wt, const float* bias, int si, const void* resid, void* out, int out_bf16) -> int {
                GemmDesc d;
                memset(&d, 0, sizeof(d));
                d.dtype = VTGB_BF16; d.M = (int)Mo; d.N = g.Cpad; d.K = 9 * C2; d.epi = VTGB_EPI_SPLIT; d.act = 1;
                d.A = A; d.lda = C2; d.W = Wt; d.ldw = d.K; d.bias = bias; d.out = out; d.ldo = C2; d.split_lo = g.Cpad;
                d.conv_H = g.Ho; d.conv_W = g.Wo; d.conv_KH = 3; d.conv_KW = 3; d.conv_Cin = C2; d.conv_split = C2; d.zero_page = zero;
                d.resid_bf16 = resid; d.ldrb = C2; d.post_relu = resid != nullptr;
                d.h8_run = 9 * (g.Cpad / 64); d.h8_scale = (const int*)w[40] + si; d.h8_out_bf16 = out_bf16;
                d.algo_flops = 2.0 * Mo * (double)g.C * (9 * g.C);
                return launch_conv_h8(d, s);
            };
            const int out_h8 = (b & 1) == 0;      // the block's output pair: f16c8 for the stride-1 block behind it, bf16 for a bf16x3 consumer
            if (!inorm) {
                // cnet: no statistics, so ReLU, the skip connection and the pair store live in the convolutions' epilogues -- no fp32 round trip, no pair pass.
                // Stride-2 blocks: conv1 and the downsample branch are bf16x3 launches that write f16c8 pairs (GemmDesc::split_f16c8)
                auto split_conv = [&](int K, const void* Wt, const float* bias, int relu, void* out) -> int {
                    GemmDesc d = enc_conv(dt, (int)Mo, g.Cpad, g.Ho, g.Wo, K, Cin_pad, g.stride, Hi, Wi, x, Wt, bias, nullptr, C2, zero, nullptr);
                    d.epi = VTGB_EPI_SPLIT; d.act = relu; d.out = out; d.split_lo = g.Cpad; d.split_f16c8 = 1;
                    d.algo_flops = 2.0 * Mo * (double)g.C * (K * K * Cin_pad);
                    return launch_conv_gemm(d, s);
                };
                if (s2) VTGB_TRY(split_conv(3, bw[0], F(bw[1]), 1, t1));                                                    // y = relu(bn1(conv1(x)))
                else VTGB_TRY(conv_pair(x, bw[0], F(bw[1]), 2 * b, nullptr, t1, 0));
                const void* res = x;
                outb = t2;                                                                                                  // conv2 reads t1 and the skip operand: a third buffer
                if (s2) {                                                                                                   // x = bn3(downsample(x))
                    VTGB_TRY(split_conv(1, bw[4], F(bw[5]), 0, t2));
                    res = t2;
                    outb = x;                                                                                               // the block input is dead from here on
                }
                VTGB_TRY(conv_pair(t1, bw[2], F(bw[3]), 2 * b + 1, res, outb, !out_h8));                                    // relu(x + relu(bn2(conv2(y))))
            } else {
                float* sf = stats_for(stats, HWo, g.C);
                if (s2) VTGB_TRY(conv_stats(enc_conv(dt, (int)Mo, g.C, g.Ho, g.Wo, 3, Cin_pad, g.stride, Hi, Wi, x, bw[0], F(bw[1]), cf, g.Cpad, zero, sf), stats));
                else VTGB_TRY(conv_h8(x, bw[0], F(bw[1]), sf, 2 * b));
                VTGB_TRY(norm(cf, Mo, HWo, g.C, g.Cpad, g.Cpad, nullptr, t1, 1, 0, stats, sf != nullptr, 1));               // y = relu(norm1(conv1(x))), f16c8 pair
                sf = stats_for(stats, HWo, g.C);
                VTGB_TRY(conv_h8(t1, bw[2], F(bw[3]), sf, 2 * b + 1));
                const void* res = x;
                if (s2) {                                                                                                   // x = norm3(downsample(x)): a bf16 pair
                    float* sf2 = stats_for(stats2, HWo, g.C);
                    VTGB_TRY(conv_stats(enc_conv(dt, (int)Mo, g.C, g.Ho, g.Wo, 1, Cin_pad, g.stride, Hi, Wi, x, bw[4], F(bw[5]), cf2, g.Cpad, zero, sf2), stats2));
                    VTGB_TRY(norm(cf2, Mo, HWo, g.C, g.Cpad, g.Cpad, nullptr, t2, 0, 0, stats2, sf2 != nullptr));
                    res = t2;
                }
                outb = t1;
                VTGB_TRY(norm(cf, Mo, HWo, g.C, g.Cpad, g.Cpad, res, outb, 1, 1, stats, sf != nullptr, out_h8, !s2));       // relu(x + relu(norm2(conv2(y))))
            }
        } else if (x3 && !inorm && g.Cpad == 128) {
            // cnet at bf16x3, 128-channel stages: no statistics are needed, so conv1 (+ ReLU) and the downsample branch leave the convolution as pairs
            // (EPI_SPLIT; the packed weights carry C_pad output rows, the padded ones zero); conv2 needs the skip operand: fp32 + the pair pass
            auto split_conv = [&](int K, int Cin, int stride, int Hi_, int Wi_, const void* A, const void* Wt, const float* bias, int relu, void* out) -> int {
                GemmDesc d = enc_conv(dt, (int)Mo, g.Cpad, g.Ho, g.Wo, K, Cin, stride, Hi_, Wi_, A, Wt, bias, nullptr, 2 * g.Cpad, zero, nullptr);
                d.epi = VTGB_EPI_SPLIT; d.act = relu; d.out = out; d.split_lo = g.Cpad;
                d.algo_flops = 2.0 * Mo * (double)g.C * (K * K * Cin);
                return launch_conv_gemm(d, s);
            };
            VTGB_TRY(split_conv(3, Cin_pad, g.stride, Hi, Wi, x, bw[0], F(bw[1]), 1, t1));                                 // y = relu(bn1(conv1(x)))
            VTGB_TRY(launch_conv_gemm(enc_conv(dt, (int)Mo, g.C, g.Ho, g.Wo, 3, g.Cpad, 1, g.Ho, g.Wo, t1, bw[2], F(bw[3]), cf, g.Cpad, zero, nullptr), s));
            const void* res = x;
            outb = t1;
            if (g.stride != 1) {                                                                                          // x = bn3(downsample(x))
                VTGB_TRY(split_conv(1, Cin_pad, g.stride, Hi, Wi, x, bw[4], F(bw[5]), 0, t2));
                res = t2;
            }
            VTGB_TRY(norm(cf, Mo, HWo, g.C, g.Cpad, g.Cpad, res, outb, 1, 1, stats, false));                                // relu(x + relu(bn2(conv2(y))))
        } else {
            float* sf = stats_for(stats, HWo, g.C);
            VTGB_TRY(conv_stats(enc_conv(dt, (int)Mo, g.C, g.Ho, g.Wo, 3, Cin_pad, g.stride, Hi, Wi, x, bw[0], F(bw[1]), cf, g.Cpad, zero, sf), stats));
            VTGB_TRY(norm(cf, Mo, HWo, g.C, g.Cpad, g.Cpad, nullptr, t1, 1, 0, stats, sf != nullptr));  // y = relu(norm1(conv1(x)))
            sf = stats_for(stats, HWo, g.C);
            VTGB_TRY(conv_stats(enc_conv(dt, (int)Mo, g.C, g.Ho, g.Wo, 3, g.Cpad, 1, g.Ho, g.Wo, t1, bw[2], F(bw[3]), cf, g.Cpad, zero, sf), stats));
            const void* res = x;
            if (g.stride != 1) {                                                                      // x = norm3(downsample(x))
                float* sf2 = stats_for(stats2, HWo, g.C);
                VTGB_TRY(conv_stats(enc_conv(dt, (int)Mo, g.C, g.Ho, g.Wo, 1, Cin_pad, g.stride, Hi, Wi, x, bw[4], F(bw[5]), cf2, g.Cpad, zero, sf2), stats2));
                VTGB_TRY(norm(cf2, Mo, HWo, g.C, g.Cpad, g.Cpad, nullptr, t2, 0, 0, stats2, sf2 != nullptr));
                res = t2;
            }
            outb = t1;
            // out = relu(x + relu(norm2(conv2(y))))   -- t1 is free once conv2 has consumed it (stream order)
            VTGB_TRY(norm(cf, Mo, HWo, g.C, g.Cpad, g.Cpad, res, outb, 1, 1, stats, sf != nullptr));
        }
        // rotate buffers: the block output becomes the next input, the other two are scratch
        void* all3[3] = {x, t1, t2};
        void* rest[2];
        int k = 0;
        for (int i = 0; i < 3; i++) if (all3[i] != outb) rest[k++] = all3[i];
        x = outb; t1 = rest[0]; t2 = rest[1];
        Cin_pad = g.Cpad; Hi = g.Ho; Wi = g.Wo;
    }
    // ---- head: 1x1, 128 -> 256 (no norm)
    {
        GemmDesc d;
        memset(&d, 0, sizeof(d));
        d.dtype = dt; d.M = (int)M8; d.N = 256; d.K = 128; d.epi = VTGB_EPI_STORE_F32;
        d.A = x; d.lda = 128; d.W = w[38]; d.ldw = 128; d.bias = F(w[39]); d.out = a->out; d.ldo = 256;
        if (x3) d = enc_conv(dt, (int)M8, 256, H8, W8, 1, 128, 1, H8, W8, x, w[38], F(w[39]), a->out, 256, zero, nullptr);
        VTGB_TRY(launch_conv_gemm(d, s));
    }
    return VTGB_OK;
}

extern "C" size_t vtgb_raft_encoder_workspace_bytes(const vtgb_raft_encoder_args* a) {
    Workspace ws(nullptr, 0);
    if (enc_impl(a, ws, nullptr) != VTGB_OK) return 0;
    return align_up(ws.used, 256);
}
extern "C" int vtgb_raft_encoder(const vtgb_raft_encoder_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->workspace, VTGB_EWORKSPACE, "raft_encoder: workspace is NULL");
    Workspace ws(a->workspace, a->workspace_bytes);
    return enc_impl(a, ws, stream);
}
