// raft_x3.hip -- RAFT's refinement loop (xraft.py:135-156, raft_utils/update.py:39-144) at the reference's fp32 ACCURACY on the bf16
// matrix cores: VTGB_BF16X3.  The reference runs RAFT in fp32 whatever the Lightning precision is (xraft.py:58,113-118); the bf16 mode of
// raft.hip is 1.4e-2 off its flows under input-sensitive weights, the fp32 FMA mode (conv_f32.hip) is exact but runs on the vector ALUs.
// Here every convolution operand is a bf16 PAIR -- hi = bf16(x), lo = bf16(x - hi), 16 significant bits -- and every product is formed as
//     x . w  ~  hi . Wh + lo . Wh + hi . Wl            (three bf16 MFMA products, fp32 accumulation; the dropped lo . Wl is 2^-16 of it)
// by the SAME implicit-GEMM kernels as the bf16 mode: an activation of C channels is stored as the row [hi(C) | lo(C)] and enters the
// contraction as the 3C channels [hi | lo | hi] (GemmDesc::conv_wrap: the third block's LDS-DMA reads the hi half again) against weights
// packed once as [Wh | Wh | Wl] (ops.py); the epilogue writes act(acc + bias) as a pair again (EPI_SPLIT).  The SepConvGRU's gates are
// evaluated in fp32 in the epilogues of its two convolutions (EPI_X3ZR: z, r * h; EPI_X3Q: h' = (1 - z) h + z tanh(q), in place on the h pair);
// the loop-invariant `inp` third of the GRU convolutions is computed once per call into fp32 start maps.  The flow, the correlation pyramid
// (split-bf16 products, raft_corr.hip) and its lookup stay fp32.  Cost: 3 x the bf16 mode's MFMA work.
#include <math.h>
#include <string.h>

#include "common.h"
#include "pair_h8.h"

// VTGB_F16C8 (round 6): the same orchestration with the update block's large convolutions over f16c8 pairs (pair_h8.h, gemm_h8.hip: fp16 main
// product + two fp8 correction products at twice the rate -- 2/3 of the bf16x3 form's matrix work at the same accuracy class); the small ones
// (convf2, FlowHead.conv2, mask.2, the once-per-call start maps) and the encoders / correlation volume stay bf16x3.  `h8` below selects the format of
// the buffers h, r h, motion | flow, corr taps, c1, [cor | flo], convf1's output; inp and the flow / mask heads' hidden maps stay bf16 pairs.
int raft_launch_lookup_pair(const CorrPyr& pyr, const float* flow, void* out_pair, int64_t M, int H8, int W8, int h8, hipStream_t s);
// f16c8: lookup + convc1 as one launch (raft.hip: the taps of a pixel never leave the CU)
int raft_lkc1_h8_pack(const void* w, void* packed, hipStream_t s);
size_t raft_lkc1_h8_pack_bytes();
int raft_launch_lookup_convc1_h8(const CorrPyr& pyr, const float* flow, const void* wpk, const int* scale, const float* bias, void* c1, int64_t M, int H8, int W8,
                                 hipStream_t s);
int raft_launch_flow_head2(const float* P2, const float* bias, float* flow, int n_pairs, int H8, int W8, hipStream_t s);
int raft_launch_upsample(const float* flow, const float* mask, float* flow_up, int n_pairs, int H8, int W8, hipStream_t s);

__device__ __forceinline__ void pair_split4(const f32x4 v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float hf = bf16_round(v[e]);
        hi[e] = (bf16_t)hf;
        lo[e] = (bf16_t)(v[e] - hf);
    }
}
__device__ __forceinline__ f32x4 pair_join4(const bf16x4 hi, const bf16x4 lo) {
    return f32x4{(float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]};
}
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f * __frcp_rn(__expf(2.0f * x) + 1.0f); }   // (~1e-6 relative: far below the pair's 2^-17)

// ---- fp32 rows -> pair rows with the element-wise tails of the encoders and of the 64-channel convolution:
// y = [relu]((x - mean) * rstd) (InstanceNorm moments optional); out = [relu](resid + y); channels [C, Cpad) are written as zeros.
// One thread per (pixel, 4 channels).
struct PairPass {
    const float* x; int64_t ldx;
    const float* stats; int HW;                 // moments [(image * C + c) * 2 + {sum, sum of squares}] or NULL
    const bf16_t* resid; int64_t ldr; int r_lo;
    bf16_t* out; int64_t ldo; int o_lo;
    int C, Cpad, relu_in, relu_out;
    int64_t M;
    int h8;                                     // out as an f16c8 pair (pair_h8.h) instead of a bf16 pair
    int resid_h8;                               // resid is an f16c8 pair
};
__global__ __launch_bounds__(256) void x3_pair_pass_kernel(const PairPass p) {
    const int g = p.Cpad >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.M * g) return;
    const int64_t m = i / g;
    const int c = (int)(i - m * g) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < p.C) {
        v = *reinterpret_cast<const f32x4*>(p.x + m * p.ldx + c);
        if (p.stats) {
            const int64_t n = m / p.HW;
            const float inv = 1.0f / (float)p.HW;
            const f32x4 s01 = *reinterpret_cast<const f32x4*>(p.stats + (n * p.C + c) * 2), s23 = *reinterpret_cast<const f32x4*>(p.stats + (n * p.C + c) * 2 + 4);
            const float sm[4] = {s01[0], s01[2], s23[0], s23[2]}, sq[4] = {s01[1], s01[3], s23[1], s23[3]};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float mean = sm[e] * inv;
                const float rstd = rsqrtf(fmaxf(sq[e] * inv - mean * mean, 0.f) + 1e-5f);
                v[e] = (v[e] - mean) * rstd;
            }
        }
        if (p.relu_in) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.resid) {
            const bf16_t* r = p.resid + m * p.ldr + c;
            if (p.resid_h8) v += h8_join4(*reinterpret_cast<const h8_u32x2*>(r), *reinterpret_cast<const h8_u32x2*>(r + p.r_lo));
            else v += pair_join4(*reinterpret_cast<const bf16x4*>(r), *reinterpret_cast<const bf16x4*>(r + p.r_lo));
        }
        if (p.relu_out) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
        }
    }
    bf16_t* o = p.out + m * p.ldo + c;
    if (p.h8) {
        h8_u32x2 hu, lu;
        h8_split4(v, hu, lu);
        *reinterpret_cast<h8_u32x2*>(o) = hu;
        *reinterpret_cast<h8_u32x2*>(o + p.o_lo) = lu;
        return;
    }
    bf16x4 hi, lo;
    pair_split4(v, hi, lo);
    *reinterpret_cast<bf16x4*>(o) = hi;
    *reinterpret_cast<bf16x4*>(o + p.o_lo) = lo;
}
int launch_x3_pair_pass(const float* x, int64_t ldx, const float* stats, int HW, const void* resid, int64_t ldr, int r_lo, void* out, int64_t ldo, int o_lo,
                        int C, int Cpad, int relu_in, int relu_out, int64_t M, hipStream_t s, int h8, int resid_h8) {
    VTGB_REQUIRE((C & 3) == 0 && (Cpad & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0 && (o_lo & 3) == 0 && (ldr & 3) == 0 && (r_lo & 3) == 0, VTGB_EINVAL,
                 "pair pass: 4-aligned rows");
    PairPass p{x, ldx, stats, HW, (const bf16_t*)resid, ldr, r_lo, (bf16_t*)out, ldo, o_lo, C, Cpad, relu_in, relu_out, M, h8, resid_h8};
    const int64_t n = M * (Cpad >> 2);
    hipLaunchKernelGGL(x3_pair_pass_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// the flow as columns 126, 127 of the motion buffer X [M, 256 16-bit units] (update.py:97), in the buffer's pair format
__device__ __forceinline__ void x3_put_flow(bf16_t* __restrict__ X, int64_t m, int i, float f, int h8) {
    if (h8) {
        unsigned short hi; unsigned char lr, lv;
        h8_split1(f, hi, lr, lv);
        reinterpret_cast<unsigned short*>(X)[m * 256 + 126 + i] = hi;
        unsigned char* lo = reinterpret_cast<unsigned char*>(X) + m * 512 + 256 + h8_lo_off(126 + i);
        lo[0] = lr; lo[4] = lv;
    } else {
        const float fh = bf16_round(f);
        X[m * 256 + 126 + i] = (bf16_t)fh;
        X[m * 256 + 254 + i] = (bf16_t)(f - fh);
    }
}

// ---- state init (xraft.py:126-132): h = tanh(cnet[:, :128]) -> hb pair [M, 256]; inp = relu(cnet[:, 128:]) -> INP pair [M, 256];
// flow = flow_init or 0 -> flow fp32 and columns 126, 127 (| + 128) of the motion buffer X [M, 256].  4 channels per thread.
__global__ __launch_bounds__(256) void x3_init_kernel(const float* __restrict__ net, const float* __restrict__ inp, const float* __restrict__ cnet,
                                                      bf16_t* __restrict__ hb, bf16_t* __restrict__ INP, bf16_t* __restrict__ X, float* __restrict__ flow,
                                                      const float* __restrict__ flow_init, int64_t M, int HW, int h8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * 32) return;
    const int64_t m = i >> 5;
    const int c = (int)(i & 31) * 4;
    f32x4 hv, iv;
    if (cnet) {
        const f32x4 nv = *reinterpret_cast<const f32x4*>(cnet + m * 256 + c);
        iv = *reinterpret_cast<const f32x4*>(cnet + m * 256 + 128 + c);
#pragma unroll
        for (int e = 0; e < 4; e++) { hv[e] = tanh_f(nv[e]); iv[e] = fmaxf(iv[e], 0.f); }
    } else {      // NCHW, tanh / relu already applied by the caller
        const int64_t n = m / HW, pp = m - n * HW;
#pragma unroll
        for (int e = 0; e < 4; e++) { hv[e] = net[(n * 128 + c + e) * HW + pp]; iv[e] = inp[(n * 128 + c + e) * HW + pp]; }
    }
    bf16x4 hi, lo;
    if (h8) {
        h8_u32x2 hu, lu;
        h8_split4(hv, hu, lu);
        *reinterpret_cast<h8_u32x2*>(hb + m * 256 + c) = hu;
        *reinterpret_cast<h8_u32x2*>(hb + m * 256 + 128 + c) = lu;
    } else {
        pair_split4(hv, hi, lo);
        *reinterpret_cast<bf16x4*>(hb + m * 256 + c) = hi;
        *reinterpret_cast<bf16x4*>(hb + m * 256 + 128 + c) = lo;
    }
    pair_split4(iv, hi, lo);
    *reinterpret_cast<bf16x4*>(INP + m * 256 + c) = hi;
    *reinterpret_cast<bf16x4*>(INP + m * 256 + 128 + c) = lo;
    if (c == 0) {
        float f0 = 0.f, f1 = 0.f;
        if (flow_init) {
            const int64_t n = m / HW, pp = m - n * HW;
            f0 = flow_init[(n * 2) * HW + pp];
            f1 = flow_init[(n * 2 + 1) * HW + pp];
        }
        flow[m * 2] = f0;
        flow[m * 2 + 1] = f1;
        x3_put_flow(X, m, 0, f0, h8);
        x3_put_flow(X, m, 1, f1, h8);
    }
}

// ---- convf1 (update.py:81,92: 7x7, 2 -> 128, ReLU) in exact fp32 on the fp32-input matrix instruction (weights [98][128], k = c * 49 + ky * 7 + kx),
// pair out; the flow itself goes to columns 126, 127 of the motion buffer X (update.py:97).  A workgroup takes 64 consecutive pixels: their 98-tap
// windows are staged tap-major in LDS (zero padded; rows padded to 65 floats so that the four k rows of a fragment read fall into different banks),
// then D[channel][pixel] = W^T[128][100] . win[100][64] (K = 98 padded to 100) on v_mfma_f32_16x16x4_f32 -- bit for bit a k-ordered fmaf chain
// (cdna guide: 'FP32-input MFMA'), i.e. the arithmetic of the vector-FMA form it replaces (round 5: 49 register-pair weights per thread, packed
// FMAs with op_sel broadcasts, 1.06 ms per launch = 55 TFLOP/s; the first form of all, two pixels per workgroup, took 3.8 ms).  Wave w holds the
// weights of channels 32 w .. 32 w + 31 as 50 registers (one float per k-step and 16-channel block), reads each window value once per 16-channel
// pair of blocks (100 ds_read_b32 for 200 MFMAs), and stores four consecutive channels of a pixel per lane as 8-byte pair halves.
constexpr int CF1_PX = 64, CF1_LDW = 65, CF1_KS = 25;
__global__ __launch_bounds__(256) void x3_convf1_kernel(const float* __restrict__ flow, const float* __restrict__ wt, const float* __restrict__ b,
                                                        bf16_t* __restrict__ f1, bf16_t* __restrict__ X, int64_t M, int H8, int W8, int h8) {
    __shared__ float win[4 * CF1_KS][CF1_LDW];
    const int tid = threadIdx.x, HW = H8 * W8;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int64_t n_tiles = (M + CF1_PX - 1) / CF1_PX;
    // this wave's weights, A[row = channel][k]: lane holds k = 4 s + fg of channel 32 wave + 16 i + fr (k >= 98: zero) -- loaded ONCE per workgroup:
    // the workgroups are persistent (as one workgroup per 64 pixels each re-read its 50 KB of weights in front of 3 us of MFMAs)
    float wa[CF1_KS][2];
#pragma unroll
    for (int s_ = 0; s_ < CF1_KS; s_++) {
        const int k = 4 * s_ + fg;
#pragma unroll
        for (int i = 0; i < 2; i++) wa[s_][i] = k < 98 ? wt[k * 128 + wave * 32 + i * 16 + fr] : 0.f;
    }
    f32x4 bias4[2];
#pragma unroll
    for (int i = 0; i < 2; i++) bias4[i] = *reinterpret_cast<const f32x4*>(b + wave * 32 + i * 16 + fg * 4);      // (the lane's four rows of block i)
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t m0 = tile * CF1_PX;
    {
        // thread -> (pixel, 13 taps): pixel = tid & 63 (consecutive lanes = consecutive pixels: coalesced-ish 8-byte gathers).  All 13 loads are issued
        // before the first LDS write (through a buffer descriptor of the flow field: a tap outside the image gets an out-of-range offset and reads
        // zeros -- as a loop with conditional loads hipcc waited out every gather on its own); one 64-bit modulo per workgroup, not per thread
        const int px = tid & 63, part = tid >> 6;
        const int64_t m = m0 + px;
        const bool live = m < M;
        const int p0 = __builtin_amdgcn_readfirstlane((int)(m0 % HW));
        int pix = p0 + px;
        pix = pix >= HW ? pix - HW : pix;                                  // (CF1_PX <= HW)
        const int y0 = pix / W8, x0 = pix - y0 * W8;
        const auto frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(flow), 0, (int)(M * 8), 0x00020000);
        typedef unsigned cf1_u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned))));
        cf1_u32x2 v[13];
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int t = part + 4 * i, tt = t < 49 ? t : 48;
            const int ky = tt / 7, kx = tt - ky * 7, y = y0 + ky - 3, x = x0 + kx - 3;
            const bool ok = live && t < 49 && (unsigned)y < (unsigned)H8 && (unsigned)x < (unsigned)W8;
            const unsigned off = ok ? (unsigned)((int)m + (ky - 3) * W8 + (kx - 3)) * 8u : 0xFFFFFFF0u;
            v[i] = __builtin_amdgcn_raw_buffer_load_b64(frs, off, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const int t = part + 4 * i;
            if (t < 49) {
                win[t][px] = __uint_as_float(v[i][0]);
                win[49 + t][px] = __uint_as_float(v[i][1]);
            }
        }
        if (tid < 2 * CF1_PX) win[98 + (tid >> 6)][tid & 63] = 0.f;      // the two padding rows of K
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = bias4[i];
    __syncthreads();
#pragma unroll
    for (int s_ = 0; s_ < CF1_KS; s_++) {
        float xb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) xb[j] = win[4 * s_ + fg][j * 16 + fr];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s_][i], xb[j], acc[i][j], 0, 0, 0);
    }
    // D: column = lane & 15 = pixel j * 16 + fr, rows = 4 fg + e = channels 32 wave + 16 i + 4 fg + e
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int64_t m = m0 + j * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            f32x4 v = acc[i][j];
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
            bf16_t* o = f1 + m * 256 + wave * 32 + i * 16 + fg * 4;
            if (h8) {      // (f16c8: convf2 is an h8 launch)
                h8_u32x2 hu, lu;
                h8_split4(v, hu, lu);
                *reinterpret_cast<h8_u32x2*>(o) = hu;
                *reinterpret_cast<h8_u32x2*>(o + 128) = lu;
            } else {
                bf16x4 hi, lo;
                pair_split4(v, hi, lo);
                *reinterpret_cast<bf16x4*>(o) = hi;
                *reinterpret_cast<bf16x4*>(o + 128) = lo;
            }
        }
    }
    if (tid < 2 * CF1_PX) {
        const int64_t m = m0 + (tid >> 1);
        if (m < M) x3_put_flow(X, m, tid & 1, flow[m * 2 + (tid & 1)], h8);
    }
    __syncthreads();      // the windows are re-filled by the next tile
    }
}

// a convolution over pair operands: C1 channels from A (row [hi(C1) | lo(C1)]), optionally C2 more from A2; K = taps * 3 (C1 + C2)
static GemmDesc x3_conv(int M, int N, int H, int W, int KH, int KW, const void* A, int C1, const void* A2, int C2, const void* Wt, const float* bias, int epi,
                        int act, void* out, int64_t ldo, int split_lo, const void* zero) {
    GemmDesc d;
    memset(&d, 0, sizeof(d));
    const int Cin = 3 * (C1 + C2);
    d.dtype = VTGB_BF16; d.M = M; d.N = N; d.K = KH * KW * Cin; d.epi = epi; d.act = act;
    d.A = A; d.lda = 2 * C1; d.A2 = A2; d.lda2 = 2 * C2; d.W = Wt; d.ldw = d.K; d.bias = bias; d.out = out; d.ldo = ldo; d.split_lo = split_lo;
    d.conv_H = H; d.conv_W = W; d.conv_KH = KH; d.conv_KW = KW; d.conv_Cin = Cin; d.conv_split = 3 * C1; d.conv_wrap = 2 * C1; d.conv_wrap2 = 2 * C2;
    d.zero_page = zero;
    d.algo_flops = 2.0 * M * (double)N * (KH * KW * (C1 + C2));      // the fp32 convolution this launch stands for (executed: 3 x)
    return d;
}
// the same over f16c8 pairs (pair_h8.h; gemm_h8.hip): K = taps * 2 (C1 + C2) 16-bit units, C2 == 0 or C2 == C1; `scale`: device pointer to the layer's
// E8M0 scale byte (weights[30][...])
static GemmDesc h8_conv(int M, int N, int H, int W, int KH, int KW, const void* A, int C1, const void* A2, int C2, const void* Wt, const float* bias, int epi,
                        int act, void* out, int64_t ldo, int split_lo, const void* zero, const int* scale, int out_bf16 = 0) {
    GemmDesc d;
    memset(&d, 0, sizeof(d));
    const int Cin = 2 * (C1 + C2);
    d.dtype = VTGB_BF16; d.M = M; d.N = N; d.K = KH * KW * Cin; d.epi = epi; d.act = act;
    d.A = A; d.lda = 2 * C1; d.A2 = A2; d.lda2 = 2 * C2; d.W = Wt; d.ldw = d.K; d.bias = bias; d.out = out; d.ldo = ldo; d.split_lo = split_lo;
    d.conv_H = H; d.conv_W = W; d.conv_KH = KH; d.conv_KW = KW; d.conv_Cin = Cin; d.conv_split = 2 * C1;
    d.zero_page = zero;
    d.h8_run = KH * KW * (C1 / 64); d.h8_scale = scale; d.h8_out_bf16 = out_bf16;
    d.algo_flops = 2.0 * M * (double)N * (KH * KW * (C1 + C2));      // the fp32 convolution this launch stands for (executed: 2 x, in fp16-MFMA units)
    return d;
}

int raft_x3_impl(const vtgb_raft_update_args* a, Workspace& ws, hipStream_t s) {
    VTGB_REQUIRE(a->n_pairs > 0 && a->H8 >= 8 && a->W8 >= 8 && a->iters > 0, VTGB_EINVAL, "raft_update: bad dims n=%d H8=%d W8=%d iters=%d", a->n_pairs, a->H8,
                 a->W8, a->iters);
    const int H8 = a->H8, W8 = a->W8, HW = H8 * W8;
    const int64_t M = (int64_t)a->n_pairs * HW;
    VTGB_REQUIRE(M < (1ll << 28), VTGB_EUNSUPPORTED, "raft_update: too many pixels (the flow field is addressed through one 32-bit buffer range)");
    // pair buffers (bf16 elements per pixel = 2 x channels)
    bf16_t* hb = (bf16_t*)ws.take(M * 256 * 2);        // h
    bf16_t* X = (bf16_t*)ws.take(M * 256 * 2);         // [motion(126) | flow(2)]
    bf16_t* INP = (bf16_t*)ws.take(M * 256 * 2);       // inp (128): read by the start-map convolutions only
    float* ZRI[2] = {(float*)ws.take(M * 256 * 4), (float*)ws.take(M * 256 * 4)};   // per GRU half: bias + the convolution of `inp` (z | r), loop invariant
    float* QI[2] = {(float*)ws.take(M * 128 * 4), (float*)ws.take(M * 128 * 4)};    // ... (q)
    const int h8 = a->dtype == VTGB_F16C8;
    bf16_t* corrf = h8 ? nullptr : (bf16_t*)ws.take(M * 768 * 2);     // 324 taps, zero-padded to 384 (f16c8: the taps never leave the fused lookup kernel)
    bf16_t* c1 = (bf16_t*)ws.take(M * 512 * 2);
    bf16_t* CF = (bf16_t*)ws.take(M * 512 * 2);        // [cor(192) | flo(64)]
    bf16_t* f1 = (bf16_t*)ws.take(M * 256 * 2);
    bf16_t* RH = (bf16_t*)ws.take(M * 256 * 2);
    bf16_t* FH = (bf16_t*)ws.take(M * 512 * 2);
    float* ZR = (float*)ws.take(M * 256 * 4);          // z (fp32 [M, 128]); the unfused form: z | r pre-activations [M, 256], then z
    float* Q = (float*)ws.take(M * 128 * 4);           // q pre-activation; also convf2's fp32 output [M, 64]
    float* flow = (float*)ws.take(M * 2 * 4);
    float* mask = (float*)ws.take(M * 576 * 4);
    float* P2 = mask;   // [M, 32] per-tap partial products of FlowHead.conv2 (the mask buffer is idle until the last iteration)
    void* zero = ws.take(256);
    void* w1pk = ws.take(raft_lkc1_h8_pack_bytes());      // f16c8: convc1's weights in the fused lookup kernel's fragment order
    if (ws.dry) return VTGB_OK;
    VTGB_REQUIRE(ws.ok(), VTGB_EWORKSPACE, "raft_update: workspace %zu < %zu bytes", ws.size, ws.used);
    VTGB_REQUIRE(((a->net && a->inp) || a->cnet_nhwc) && a->weights && a->flow_up, VTGB_EINVAL, "raft_update: NULL operand");
    VTGB_REQUIRE(!a->corr_f16, VTGB_EINVAL, "raft_update: the bf16x3 / f16c8 modes take an fp32 correlation pyramid");
    const void* const* w = a->weights;
    for (int i = 0; i < VTGB_RAFT_NW + h8; i++) VTGB_REQUIRE(w[i], VTGB_EINVAL, "raft_update: weights[%d] is NULL (the bf16x3 / f16c8 tables always carry the inp split)", i);
    const int* hs = h8 ? (const int*)w[VTGB_RAFT_NW] : nullptr;      // f16c8: the ten f16c8 convolutions' weight-scale bytes (include/vtgb.h)
    // one large convolution in the mode's operand format
    auto conv = [&](int N, int KH, int KW, const void* A, int C1, const void* A2, int C2, int wi, int si, const float* bias, int epi, int act, void* out, int64_t ldo,
                    int split_lo, int out_bf16 = 0) {
        return h8 ? h8_conv((int)M, N, H8, W8, KH, KW, A, C1, A2, C2, w[wi], bias, epi, act, out, ldo, split_lo, zero, hs + si, out_bf16)
                  : x3_conv((int)M, N, H8, W8, KH, KW, A, C1, A2, C2, w[wi], bias, epi, act, out, ldo, split_lo, zero);
    };
    auto run = [&](const GemmDesc& d) { return h8 ? launch_conv_h8(d, s) : launch_conv_gemm(d, s); };
    CorrPyr pyr;
    int hl = H8, wl = W8;
    for (int l = 0; l < 4; l++) {
        VTGB_REQUIRE(a->corr[l] && hl >= 1 && wl >= 1, VTGB_EINVAL, "raft_update: correlation level %d missing", l);
        pyr.lvl[l] = a->corr[l]; pyr.h[l] = hl; pyr.w[l] = wl;
        hl /= 2; wl /= 2;
    }
    VTGB_HIP(hipMemsetAsync(zero, 0, 256, s));
    const int Mi = (int)M;
    const dim3 g32((unsigned)((M * 32 + 255) / 256));
    auto F = [](const void* p) { return (const float*)p; };
    hipLaunchKernelGGL(x3_init_kernel, g32, dim3(256), 0, s, a->net, a->inp, a->cnet_nhwc, hb, INP, X, flow, a->flow_init, M, HW, h8);
    // the GRU convolutions' contribution of `inp` (input channels 128..255: constant over the refinement iterations) + bias, once per call:
    // the 80 GRU launches contract over [h | motion | flow] = 256 channels instead of 384 (as in the bf16 mode, here as fp32 maps
    // that the gate kernels add)
    for (int half = 0; half < 2; half++) {
        const int kh = half == 0 ? 1 : 5, kw = half == 0 ? 5 : 1, wi = 10 + 4 * half;
        GemmDesc mz = x3_conv(Mi, 256, H8, W8, kh, kw, INP, 128, nullptr, 0, w[26 + 2 * half], F(w[wi + 1]), VTGB_EPI_STORE_F32, 0, ZRI[half], 256, 0, zero);
        GemmDesc mq = x3_conv(Mi, 128, H8, W8, kh, kw, INP, 128, nullptr, 0, w[27 + 2 * half], F(w[wi + 3]), VTGB_EPI_STORE_F32, 0, QI[half], 128, 0, zero);
        mz.algo_flops = mq.algo_flops = -1.0;   // credited to the 20 per-iteration launches (the reference's form)
        VTGB_TRY(launch_conv_gemm(mz, s));
        VTGB_TRY(launch_conv_gemm(mq, s));
    }
    if (h8) VTGB_TRY(raft_lkc1_h8_pack(w[0], w1pk, s));
    for (int it = 0; it < a->iters; it++) {
        // ---- BasicMotionEncoder (update.py:88-97)
        if (h8) {      // lookup + convc1 (1x1, 324 -> 256, ReLU) as one launch: the 1.5 KB-per-pixel tap tensor is never written
            VTGB_TRY(raft_launch_lookup_convc1_h8(pyr, flow, w1pk, hs, F(w[1]), c1, M, H8, W8, s));
        } else {
            VTGB_TRY(raft_launch_lookup_pair(pyr, flow, corrf, M, H8, W8, h8, s));
            VTGB_TRY(run(conv(256, 1, 1, corrf, 384, nullptr, 0, 0, 0, F(w[1]), VTGB_EPI_SPLIT, 1, c1, 512, 256)));
        }
        VTGB_TRY(run(conv(192, 3, 3, c1, 256, nullptr, 0, 2, 1, F(w[3]), VTGB_EPI_SPLIT, 1, CF, 512, 256)));
        {
            const int64_t nt = (M + CF1_PX - 1) / CF1_PX, cap = (int64_t)cu_count() * 6;      // persistent: six workgroups per CU (26 KB of LDS each)
            hipLaunchKernelGGL(x3_convf1_kernel, dim3((unsigned)(nt < cap ? nt : cap)), dim3(256), 0, s, flow, F(w[4]), F(w[5]), f1, X, M, H8, W8, h8);
        }
        if (h8) {      // convf2 (3x3, 128 -> 64, ReLU) straight into the [cor | flo] pair: the 64-wide tile's pair store, no fp32 round trip
            VTGB_TRY(run(conv(64, 3, 3, f1, 128, nullptr, 0, 6, 9, F(w[7]), VTGB_EPI_SPLIT, 1, CF + 192, 512, 256)));
        } else {
            VTGB_TRY(launch_conv_gemm(x3_conv(Mi, 64, H8, W8, 3, 3, f1, 128, nullptr, 0, w[6], F(w[7]), VTGB_EPI_STORE_F32, 0, Q, 64, 0, zero), s));
            VTGB_TRY(launch_x3_pair_pass(Q, 64, nullptr, HW, nullptr, 0, 0, CF + 192, 512, 256, 64, 64, 1, 0, M, s, 0, 0));
        }
        VTGB_TRY(run(conv(126, 3, 3, CF, 256, nullptr, 0, 8, 2, F(w[9]), VTGB_EPI_SPLIT, 1, X, 256, 128)));
        // ---- SepConvGRU (update.py:50-65): horizontal (1x5) then vertical (5x1); input channels [h(128) | motion(126) | flow(2)], the inp third comes from the start maps
        for (int half = 0; half < 2; half++) {
            const int kh = half == 0 ? 1 : 5, kw = half == 0 ? 5 : 1, wi = 10 + 4 * half;
            // z | r convolution with the gates in its epilogue: z = sigmoid(. + start map) -> ZR [M, 128] fp32, r * h -> RH pair; then the q
            // convolution over [r h | motion | flow] with the update h' = (1 - z) h + z tanh(. + start map) in ITS epilogue, in place on the h pair
            GemmDesc zr = conv(256, kh, kw, hb, 128, X, 128, wi, 3 + 2 * half, nullptr, VTGB_EPI_X3ZR, 0, ZR, 128, 128);
            zr.resid = ZRI[half]; zr.ldr = 256; zr.aux = hb; zr.ldaux = 256; zr.out2 = RH; zr.ldo2 = 256;
            zr.algo_flops = 2.0 * Mi * 256.0 * (5 * 384);
            VTGB_TRY(run(zr));
            GemmDesc q = conv(128, kh, kw, RH, 128, X, 128, wi + 2, 4 + 2 * half, nullptr, VTGB_EPI_X3Q, 0, hb, 256, 128);
            q.resid = QI[half]; q.ldr = 128; q.aux = ZR; q.ldaux = 128;
            q.algo_flops = 2.0 * Mi * 128.0 * (5 * 384);
            VTGB_TRY(run(q));
        }
        // ---- FlowHead (update.py:10-18) and coords1 += delta_flow (xraft.py:145); the hidden map leaves as a bf16 pair in both modes (conv2 is a bf16x3 launch)
        if (h8) {      // conv1 with conv2's 18 per-tap products formed in its epilogue (gemm_h8.hip EPI_FTAIL: exact fp32, the hidden map is never stored)
            GemmDesc fh = conv(256, 3, 3, hb, 128, nullptr, 0, 18, 7, F(w[19]), VTGB_EPI_SPLIT, 1, FH, 512, 256, 1);
            fh.tail_w = w[20]; fh.tail_out = P2; fh.ldtail = 32;
            VTGB_TRY(run(fh));
        } else {
            VTGB_TRY(run(conv(256, 3, 3, hb, 128, nullptr, 0, 18, 7, F(w[19]), VTGB_EPI_SPLIT, 1, FH, 512, 256, 1)));
            VTGB_TRY(launch_conv_gemm(x3_conv(Mi, 32, H8, W8, 1, 1, FH, 256, nullptr, 0, w[20], nullptr, VTGB_EPI_STORE_F32, 0, P2, 32, 0, zero), s));
        }
        VTGB_TRY(raft_launch_flow_head2(P2, F(w[21]), flow, a->n_pairs, H8, W8, s));
    }
    // ---- mask head of the last iteration (update.py:129-132,143; the 0.25 is folded into [24] / [25]) and convex upsample (xraft.py:88-99)
    VTGB_TRY(run(conv(256, 3, 3, hb, 128, nullptr, 0, 22, 8, F(w[23]), VTGB_EPI_SPLIT, 1, FH, 512, 256, 1)));
    VTGB_TRY(launch_conv_gemm(x3_conv(Mi, 576, H8, W8, 1, 1, FH, 256, nullptr, 0, w[24], F(w[25]), VTGB_EPI_STORE_F32, 0, mask, 576, 0, zero), s));
    VTGB_TRY(raft_launch_upsample(flow, mask, a->flow_up, a->n_pairs, H8, W8, s));
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---- unit-level entry points of the pair formats (include/vtgb.h: vtgb_pair_pack, vtgb_pair_conv)
extern "C" int vtgb_pair_pack(int32_t fmt, const float* x, void* out, int64_t M, int32_t C, int32_t ld_pair, vtgb_stream_t stream) {
    VTGB_REQUIRE(x && out && M > 0 && C > 0 && (C & 3) == 0 && ld_pair >= C && (ld_pair & 3) == 0 && (fmt == VTGB_F16C8 || fmt == VTGB_BF16X3), VTGB_EINVAL,
                 "pair_pack: bad argument (C=%d ld_pair=%d fmt=%d)", C, ld_pair, fmt);
    return launch_x3_pair_pass(x, C, nullptr, 1, nullptr, 0, 0, out, 2 * (int64_t)ld_pair, ld_pair, C, ld_pair, 0, 0, M, (hipStream_t)stream, fmt == VTGB_F16C8, 0);
}
static int pair_conv_impl(const vtgb_pair_conv_args* a, const vtgb_pair_conv_ex_args* x, vtgb_stream_t stream);
extern "C" int vtgb_pair_conv(const vtgb_pair_conv_args* a, vtgb_stream_t stream) { return pair_conv_impl(a, nullptr, stream); }
extern "C" int vtgb_pair_conv_ex(const vtgb_pair_conv_ex_args* x, vtgb_stream_t stream) {
    VTGB_REQUIRE(x, VTGB_EINVAL, "pair_conv_ex: NULL args");
    VTGB_REQUIRE((x->resid != nullptr) + (x->tail_w != nullptr) + (x->out_f32 != nullptr) <= 1, VTGB_EINVAL, "pair_conv_ex: at most one of resid / tail_w / out_f32");
    return pair_conv_impl(&x->conv, x, stream);
}
static int pair_conv_impl(const vtgb_pair_conv_args* a, const vtgb_pair_conv_ex_args* x, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->a && a->weights && a->scale && a->out && a->M > 0 && a->N > 0 && (a->N & 1) == 0 && a->C1 > 0 && (a->C1 % 64) == 0 && a->ld_out >= a->N &&
                     (a->ld_out & 3) == 0 && (a->out_fmt == VTGB_F16C8 || a->out_fmt == VTGB_BF16X3) && (a->act == 0 || a->act == 1),
                 VTGB_EINVAL, "pair_conv: bad argument");
    static void* zero = nullptr;      // 256 bytes of zeros for the out-of-image taps (allocated once per process)
    if (!zero) {
        VTGB_HIP(hipMalloc(&zero, 256));
        VTGB_HIP(hipMemset(zero, 0, 256));
    }
    GemmDesc d = h8_conv(a->M, a->N, a->H, a->W, a->KH, a->KW, a->a, a->C1, a->a2, a->a2 ? a->C1 : 0, a->weights, a->bias, VTGB_EPI_SPLIT, a->act, a->out, 2 * (int64_t)a->ld_out,
                         a->ld_out, zero, a->scale, a->out_fmt == VTGB_BF16X3);
    if (x && x->resid) {
        VTGB_REQUIRE(x->ld_resid >= a->N && (x->ld_resid & 3) == 0 && x->ld_resid == a->ld_out, VTGB_EINVAL, "pair_conv_ex: the skip operand's pair rows have the output's layout");
        d.resid_bf16 = x->resid; d.ldrb = 2 * (int64_t)x->ld_resid; d.post_relu = 1;
    }
    if (x && x->tail_w) {
        VTGB_REQUIRE(x->tail_out, VTGB_EINVAL, "pair_conv_ex: tail_out is NULL");
        d.tail_w = x->tail_w; d.tail_out = x->tail_out; d.ldtail = 32;
    }
    if (x && x->out_f32) {
        VTGB_REQUIRE(x->ld_f32 >= a->N && (x->ld_f32 & 3) == 0, VTGB_EINVAL, "pair_conv_ex: fp32 rows need ld_f32 >= N, %% 4 == 0");
        d.epi = VTGB_EPI_STORE_F32; d.out = x->out_f32; d.ldo = x->ld_f32; d.split_lo = 0;
    }
    return launch_conv_h8(d, (hipStream_t)stream);
}
