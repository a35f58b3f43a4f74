// raft_corr.hip -- CorrBlock.__init__ of RAFT (raft_utils/corr.py:12-27 with the all-pairs product :52-60) in ONE
// kernel: for a 16-pixel tile of image 1 the dot products with every pixel of image 2 (K = 256 features), the
// 1/sqrt(dim) scale and the three 2x2 average pools, straight into the four pyramid levels.  The correlation
// volume never exists in any other form: the reference's matmul output (233 MB per T=96 clip in fp32) plus the
// three pooled copies are written once, as the lookup kernel reads them.
//
// A workgroup (4 waves) owns 16 rows p of one pair.  Its slice of the volume S[16][HW] lives in LDS as fp32:
//   VTGB_BF16: v_mfma_f32_16x16x32_f16 on an IEEE-half copy of the features (11 significant bits on the inputs,
//              fp32 accumulation; the scale is applied to the fp32 accumulator, so there is no half-precision
//              overflow however large the raw dot product is); image 2's rows are the MFMA "A" operand read
//              straight from L2 (400 KB per image, shared by the 49 workgroups of a pair), image 1's tile is the
//              "B" operand in LDS; levels are stored as half.
//   VTGB_F32 : fp32 FMAs, features summed in order; levels are stored as fp32 (the exactness mode).
// Level l+1 is the 2x2 mean of the fp32 level l kept in LDS (avg_pool2d floors odd sizes: 7 -> 3).
#include "common.h"

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;

// bf16x3 mode: the features as a bf16 pair, hi = bf16(x) in y[0 .. n), lo = bf16(x - hi) in y[n .. 2n)
__global__ void split_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    const float e[4] = {v.x, v.y, v.z, v.w};
    bf16x4 hi, lo;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float hf = bf16_round(e[k]);
        hi[k] = (bf16_t)hf;
        lo[k] = (bf16_t)(e[k] - hf);
    }
    *reinterpret_cast<bf16x4*>(y + i * 4) = hi;
    *reinterpret_cast<bf16x4*>(y + n4 * 4 + i * 4) = lo;
}

__global__ void cast_f16_kernel(const float* __restrict__ x, half_t* __restrict__ y, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    *reinterpret_cast<half4*>(y + i * 4) = half4{(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
}

#ifndef CORR_ABL
#define CORR_ABL 0      // (timing experiments, WRONG RESULTS: 1 = no level-0 store, 2 = no products and no image-2 stream, 4 = no pooled levels, 8 = the stream without the MFMAs, 16 = the MFMAs without the stream; tools/exp/build_variant.sh)
#endif
constexpr int CORR_D = 256;
constexpr int CORR_F1H_LD = CORR_D + 8;    // half elements per LDS row of the image-1 tile (528 B: 16 rows -> 16 bank groups)
constexpr int CORR_F1F_LD = CORR_D + 4;    // fp32 variant

struct CorrLds { int ldS, n1, n2, n3, tp; size_t bytes; };
// tp: image-1 rows per workgroup.  bf16 mode: 32 when the slice fits LDS (every workgroup streams ALL of image 2 from L2 -- 400 KB at
// 28 x 28 -- so 32 rows halve the launch's L2 traffic: 58 GB per 2 945 pairs at 16 rows was what bound it), else 16; fp32 mode: 16.
static inline CorrLds corr_lds_tp(int H8, int W8, bool f32, int tp, bool split = false) {
    CorrLds c;
    const int HW = H8 * W8;
    c.ldS = (HW + 15) / 16 * 16 + 4;
    c.n1 = (H8 / 2) * (W8 / 2);
    c.n2 = (H8 / 4) * (W8 / 4);
    c.n3 = (H8 / 8) * (W8 / 8);
    c.tp = tp;
    // the pooled levels 1, 2 live where the image-1 tile was (its rows are in registers / consumed when the products are done): at bf16x3 that
    // takes the 16-row slice from 83 KB to 67 KB -- TWO workgroups per CU (round 6: 15.4 -> see DESIGN.md section 4)
    const size_t f1_bytes = f32 ? (size_t)tp * CORR_F1F_LD * 4 : (size_t)tp * CORR_F1H_LD * 2 * (split ? 2 : 1);
    const size_t pool_bytes = (size_t)tp * (c.n1 + c.n2) * 4;
    c.bytes = (size_t)tp * c.ldS * 4 + (f1_bytes > pool_bytes ? f1_bytes : pool_bytes);
    return c;
}
static inline CorrLds corr_lds(int H8, int W8, bool f32) {
    if (!f32) {
        const CorrLds c32 = corr_lds_tp(H8, W8, false, 32);
        if (c32.bytes <= 160 * 1024) return c32;
    }
    return corr_lds_tp(H8, W8, f32, 16);
}

// bf16x3: 32 image-1 rows per workgroup when the slice fits (135 KB at 28 x 28: one workgroup of 8 waves per CU) -- the launch is bound by the image-2
// stream out of L2 (ablations, round 6: the stream alone 13.7 ms of 13.7, the MFMAs alone 4.9), which 32 rows halve; else 16 rows (67 KB, two per CU)
static inline CorrLds corr_lds_x3(int H8, int W8) {
    const CorrLds c32 = corr_lds_tp(H8, W8, false, 32, true);
    return c32.bytes <= 160 * 1024 ? c32 : corr_lds_tp(H8, W8, false, 16, true);
}

// SPLIT (VTGB_BF16X3): the features are bf16 pairs (fh = hi plane, then lo plane), S = f1h.f2h + f1h.f2l + f1l.f2h on the bf16 MFMA with
// fp32 accumulation (the dropped lo.lo term is 2^-16 of a product), fp32 levels: the fp32 mode's accuracy at 3 x the bf16 mode's MFMA work
// instead of fp32 FMAs (65 ms -> per 2 945 pairs)
template <bool F32, typename OT, int TP, bool SPLIT = false>
__global__ __launch_bounds__(TP * 16) void raft_corr_kernel(const vtgb_raft_corr_args a, const half_t* __restrict__ fh, const int ldS, const int n1,
                                                        const int n2) {
    extern __shared__ __attribute__((aligned(16))) char corr_sm[];
    const int H = a.H8, W = a.W8, HW = H * W;
    float* const S = reinterpret_cast<float*>(corr_sm);
    constexpr int NT = TP * 16;      // threads: 4 waves per 16 image-1 rows
    char* const f1s = reinterpret_cast<char*>(S + TP * ldS);
    float* const S1 = reinterpret_cast<float*>(f1s);      // (levels 1, 2 reuse the image-1 tile's bytes: written after the barrier behind the products)
    float* const S2 = S1 + TP * n1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (pair, 16-row tile), XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own 4 MB L2, and every
    // workgroup of a pair streams ALL of image 2 (0.8 MB as a bf16 pair at 28 x 28).  With the pair in blockIdx.y a pair's 49 workgroups sat on all 8 XCDs
    // and each L2 held slices of ~10 pairs at a time (8 MB): the stream missed L2 (116 GB per 2 945 pairs out of the Infinity Cache / HBM).  Now id = 8 k + x
    // works on pair 8 (k / tiles) + x: a pair stays on ONE XCD, whose L2 holds the one or two images in flight.
    const int tiles = (HW + TP - 1) / TP;
    const int xk = (int)(blockIdx.x >> 3);
    const int64_t n = (int64_t)(xk / tiles) * 8 + (blockIdx.x & 7);
    if (n >= a.n_pairs) return;
    const int p0 = (xk % tiles) * TP;
    const int64_t img = (n / a.pairs_per_clip) * a.frames_per_clip + n % a.pairs_per_clip;
    const int64_t i1 = img + a.first_off, i2 = img + a.second_off;
    // ---- image-1 tile -> LDS (rows past the image: the last row again; their outputs are never stored)
    if constexpr (F32) {
        const float* f1 = a.fmap + i1 * HW * CORR_D;
        float* dst = reinterpret_cast<float*>(f1s);
        for (int i = tid; i < TP * (CORR_D / 4); i += NT) {
            const int r = i / (CORR_D / 4), c = i - r * (CORR_D / 4);
            const int p = min(p0 + r, HW - 1);
            *reinterpret_cast<float4*>(dst + r * CORR_F1F_LD + c * 4) = *reinterpret_cast<const float4*>(f1 + (int64_t)p * CORR_D + c * 4);
        }
    } else {
        const half_t* f1 = fh + i1 * HW * CORR_D;
        half_t* dst = reinterpret_cast<half_t*>(f1s);
        const int64_t plane = (int64_t)a.n_images * HW * CORR_D;      // SPLIT: the lo plane follows the hi plane (2-byte elements either way)
        for (int i = tid; i < TP * (CORR_D / 8); i += NT) {
            const int r = i / (CORR_D / 8), c = i - r * (CORR_D / 8);
            const int p = min(p0 + r, HW - 1);
            *reinterpret_cast<half8*>(dst + r * CORR_F1H_LD + c * 8) = *reinterpret_cast<const half8*>(f1 + (int64_t)p * CORR_D + c * 8);
            if constexpr (SPLIT)
                *reinterpret_cast<half8*>(dst + (TP + r) * CORR_F1H_LD + c * 8) = *reinterpret_cast<const half8*>(f1 + plane + (int64_t)p * CORR_D + c * 8);
        }
    }
    __syncthreads();
    // ---- S[p][q] = scale * <f1[p], f2[q]>
    if constexpr (F32) {
        const float* f2 = a.fmap + i2 * HW * CORR_D;
        const float* f1l = reinterpret_cast<const float*>(f1s);
        static_assert(!F32 || TP == 16, "the fp32 path keeps 16 accumulators per thread");
        for (int q = tid; q < HW; q += NT) {
            const float* row = f2 + (int64_t)q * CORR_D;
            float acc[16];
#pragma unroll
            for (int p = 0; p < 16; p++) acc[p] = 0.f;
            for (int d = 0; d < CORR_D; d += 4) {
                const float4 b = *reinterpret_cast<const float4*>(row + d);
#pragma unroll
                for (int p = 0; p < 16; p++) {
                    const float4 x = *reinterpret_cast<const float4*>(f1l + p * CORR_F1F_LD + d);   // same address in every lane: broadcast
                    acc[p] = fmaf(x.x, b.x, acc[p]);
                    acc[p] = fmaf(x.y, b.y, acc[p]);
                    acc[p] = fmaf(x.z, b.z, acc[p]);
                    acc[p] = fmaf(x.w, b.w, acc[p]);
                }
            }
#pragma unroll
            for (int p = 0; p < 16; p++) S[p * ldS + q] = acc[p] * a.scale;
        }
    } else if constexpr (SPLIT) {
        constexpr int NH = TP / 16;      // 16-row halves of the tile (32 rows: every wave multiplies each q-tile with both -- half the image-2 stream per output)
        const bf16_t* f2 = reinterpret_cast<const bf16_t*>(fh) + i2 * HW * CORR_D;
        const int64_t plane = (int64_t)a.n_images * HW * CORR_D;
        const bf16_t* f1l = reinterpret_cast<const bf16_t*>(f1s);
        const int fr = lane & 15, fg = lane >> 4;
        constexpr int NW = NT / 64, KS = CORR_D / 32;
        // k order of the contraction (any order serves, both operands use it): k-steps 2 s and 2 s + 1 take, for lane group fg, the two 8-element halves of
        // the 16 CONTIGUOUS features 64 s + 16 fg .. + 16 -- a lane's two loads of image 2 are adjacent and the four lanes of a row cover one whole
        // 128-byte line per pair of steps (with 32 s + 8 fg each step touched half of 16 lines and the other halves came from L2 again one step later:
        // the launch moved ~230 GB through L2 for 12 GB of HBM traffic and was bound by that)
        auto koff = [&](int ks) { return (ks >> 1) * 64 + fg * 16 + (ks & 1) * 8; };
        const int n_qt = (HW + 15) >> 4;
        bf16x8 ah[2][KS], al[2][KS];
        auto load_a = [&](int qt, bf16x8 (&dh)[KS], bf16x8 (&dl)[KS]) {
            if ((CORR_ABL & 16) != 0 && qt != wave) return;      // (no image-2 stream: the first q-tile's rows again)
            const int q = min(qt * 16 + fr, HW - 1);
            const bf16_t* row = f2 + (int64_t)q * CORR_D;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                dh[ks] = *reinterpret_cast<const bf16x8*>(row + koff(ks));
                dl[ks] = *reinterpret_cast<const bf16x8*>(row + plane + koff(ks));
            }
        };
        auto compute = [&](int qt, const bf16x8 (&sh)[KS], const bf16x8 (&sl)[KS]) {
            if constexpr ((CORR_ABL & 8) != 0) {      // (no products: the loads are kept alive)
#pragma unroll
                for (int ks = 0; ks < KS; ks++) asm volatile("" ::"v"(sh[ks]), "v"(sl[ks]));
                return;
            }
#pragma unroll
            for (int h = 0; h < NH; h++) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;      // the two small terms on their own accumulator: added last
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {      // image 1's fragments come from LDS each time (64 registers per half otherwise)
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(f1l + (h * 16 + fr) * CORR_F1H_LD + koff(ks));
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(f1l + (TP + h * 16 + fr) * CORR_F1H_LD + koff(ks));
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sl[ks], bh, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sh[ks], bl, acc2, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sh[ks], bh, acc, 0, 0, 0);
                }
                if (qt < n_qt) *reinterpret_cast<f32x4*>(S + (h * 16 + fr) * ldS + qt * 16 + fg * 4) = (acc + acc2) * a.scale;      // (a q-tile past the end: computed, not kept)
            }
        };
        // every load is UNCONDITIONAL (a q-tile past the end reads the last row again: load_a clamps): with `if (next tile exists) load` hipcc's
        // s_waitcnt at the joins was vmcnt(0) -- each q-tile waited out its own L2 round trip (2.9 us per q-tile and wave; the launch: 12.6 ms)
        int qt = (CORR_ABL & 2) ? n_qt : wave;
        load_a(qt, ah[0], al[0]);
        for (; qt < n_qt; qt += 2 * NW) {
            // (the scheduling barriers keep each group of 16 loads in front of the OTHER buffer's products: left alone, hipcc sinks the loads to their uses)
            load_a(qt + NW, ah[1], al[1]);
            __builtin_amdgcn_sched_barrier(0);
            compute(qt, ah[0], al[0]);
            __builtin_amdgcn_sched_barrier(0);
            load_a(qt + 2 * NW, ah[0], al[0]);
            __builtin_amdgcn_sched_barrier(0);
            compute(qt + NW, ah[1], al[1]);      // (unconditional as well: a branch here and hipcc unswitches the loop, losing the two-buffer order)
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const half_t* f2 = fh + i2 * HW * CORR_D;
        const half_t* f1l = reinterpret_cast<const half_t*>(f1s);
        const int fr = lane & 15, fg = lane >> 4;
        constexpr int NH = TP / 16, NW = NT / 64;      // 16-row halves of the tile; waves
        half8 bfrag[NH][CORR_D / 32];
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
            for (int ks = 0; ks < CORR_D / 32; ks++) bfrag[h][ks] = *reinterpret_cast<const half8*>(f1l + (h * 16 + fr) * CORR_F1H_LD + ks * 32 + fg * 8);
        const int n_qt = (HW + 15) >> 4;
        // image-2 fragments come straight from L2: the loads of q-tile t+1 are in flight while tile t's MFMAs run (without the
        // prefetch every q-tile waited out a full L2 round trip: the kernel was latency-bound at 0.6 TB/s of output)
        half8 afrag[2][CORR_D / 32];
        auto load_a = [&](int qt, half8 (&dst)[CORR_D / 32]) {
            const int q = min(qt * 16 + fr, HW - 1);
            const half_t* row = f2 + (int64_t)q * CORR_D + fg * 8;
#pragma unroll
            for (int ks = 0; ks < CORR_D / 32; ks++) dst[ks] = *reinterpret_cast<const half8*>(row + ks * 32);
        };
        auto compute = [&](int qt, const half8 (&src)[CORR_D / 32]) {
#pragma unroll
            for (int h = 0; h < NH; h++) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < CORR_D / 32; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(src[ks], bfrag[h][ks], acc, 0, 0, 0);
                // D: rows fg * 4 + r = q within the tile, column fr = p
                if (qt < n_qt) *reinterpret_cast<f32x4*>(S + (h * 16 + fr) * ldS + qt * 16 + fg * 4) = acc * a.scale;
            }
        };
        int qt = wave;      // (unconditional, clamped loads: see the split path)
        load_a(qt, afrag[0]);
        for (; qt < n_qt; qt += 2 * NW) {
            load_a(qt + NW, afrag[1]);
            __builtin_amdgcn_sched_barrier(0);
            compute(qt, afrag[0]);
            __builtin_amdgcn_sched_barrier(0);
            load_a(qt + 2 * NW, afrag[0]);
            __builtin_amdgcn_sched_barrier(0);
            compute(qt + NW, afrag[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    // ---- the four levels.  Row p of level l is [n * HW + p0 + p][n_l]: the tile's outputs are one contiguous span.
    const int rows = min(TP, HW - p0);
    OT* const o0 = reinterpret_cast<OT*>(a.levels[0]) + (n * HW + p0) * (int64_t)HW;
    if (CORR_ABL & 1) {
    } else if ((HW & 7) == 0) {
        // 8 consecutive q per lane: 16-byte (half) / 2 x 16-byte (fp32) stores -- element-wise 2-byte stores made this
        // write-out (3/4 of the kernel's bytes) the longest phase of the kernel
        typedef OT OT8 __attribute__((ext_vector_type(8)));
        for (int i = tid * 8; i < rows * HW; i += NT * 8) {
            const int p = i / HW, q = i - p * HW;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(S + p * ldS + q), hi = *reinterpret_cast<const f32x4*>(S + p * ldS + q + 4);
            const OT8 v = {(OT)lo[0], (OT)lo[1], (OT)lo[2], (OT)lo[3], (OT)hi[0], (OT)hi[1], (OT)hi[2], (OT)hi[3]};
            *reinterpret_cast<OT8*>(o0 + i) = v;
        }
    } else {
        for (int i = tid; i < rows * HW; i += NT) {
            const int p = i / HW, q = i - p * HW;
            o0[i] = (OT)S[p * ldS + q];
        }
    }
    if (CORR_ABL & 4) return;
    const int w1 = W / 2, w2 = W / 4, w3 = W / 8, n3 = (H / 8) * w3;
    OT* const o1 = reinterpret_cast<OT*>(a.levels[1]) + (n * HW + p0) * (int64_t)n1;
    for (int i = tid; i < rows * n1; i += NT) {
        const int p = i / n1, r = i - p * n1, y = r / w1, x = r - y * w1;
        const float* s = S + p * ldS + (2 * y) * W + 2 * x;
        const float v = (((s[0] + s[1]) + s[W]) + s[W + 1]) * 0.25f;
        S1[i] = v;
        o1[i] = (OT)v;
    }
    __syncthreads();
    OT* const o2 = reinterpret_cast<OT*>(a.levels[2]) + (n * HW + p0) * (int64_t)n2;
    for (int i = tid; i < rows * n2; i += NT) {
        const int p = i / n2, r = i - p * n2, y = r / w2, x = r - y * w2;
        const float* s = S1 + p * n1 + (2 * y) * w1 + 2 * x;
        const float v = (((s[0] + s[1]) + s[w1]) + s[w1 + 1]) * 0.25f;
        S2[i] = v;
        o2[i] = (OT)v;
    }
    __syncthreads();
    OT* const o3 = reinterpret_cast<OT*>(a.levels[3]) + (n * HW + p0) * (int64_t)n3;
    for (int i = tid; i < rows * n3; i += NT) {
        const int p = i / n3, r = i - p * n3, y = r / w3, x = r - y * w3;
        const float* s = S2 + p * n2 + (2 * y) * w2 + 2 * x;
        o3[i] = (OT)((((s[0] + s[1]) + s[w2]) + s[w2 + 1]) * 0.25f);
    }
}

static int corr_check(const vtgb_raft_corr_args* a) {
    VTGB_REQUIRE(a, VTGB_EINVAL, "raft_corr: NULL args");
    VTGB_REQUIRE(a->dtype == VTGB_BF16 || a->dtype == VTGB_F32 || a->dtype == VTGB_BF16X3, VTGB_EINVAL, "raft_corr: bad dtype %d", a->dtype);
    VTGB_REQUIRE(a->n_pairs <= 65535, VTGB_EUNSUPPORTED, "raft_corr: at most 65535 pairs per call (got %d)", a->n_pairs);
    VTGB_REQUIRE(a->n_pairs > 0 && a->H8 >= 8 && a->W8 >= 8 && a->dim == CORR_D && a->pairs_per_clip > 0 && a->frames_per_clip > 0 && a->n_images > 0,
                 VTGB_EINVAL, "raft_corr: bad dims n_pairs=%d H8=%d W8=%d dim=%d", a->n_pairs, a->H8, a->W8, a->dim);
    const int64_t last = ((int64_t)(a->n_pairs - 1) / a->pairs_per_clip) * a->frames_per_clip + (a->n_pairs - 1) % a->pairs_per_clip;
    VTGB_REQUIRE(a->first_off >= 0 && a->second_off >= 0 && last + a->first_off < a->n_images && last + a->second_off < a->n_images, VTGB_EINVAL,
                 "raft_corr: pair -> image map leaves the %d feature maps", a->n_images);
    const CorrLds c = a->dtype == VTGB_BF16X3 ? corr_lds_x3(a->H8, a->W8) : corr_lds(a->H8, a->W8, a->dtype == VTGB_F32);
    VTGB_REQUIRE(c.bytes <= 160 * 1024, VTGB_EUNSUPPORTED, "raft_corr: %d x %d maps exceed the LDS tile", a->H8, a->W8);
    return VTGB_OK;
}

extern "C" size_t vtgb_raft_corr_workspace_bytes(const vtgb_raft_corr_args* a) {
    if (corr_check(a) != VTGB_OK) return 0;
    if (a->dtype == VTGB_F32) return 256;
    return align_up((size_t)a->n_images * a->H8 * a->W8 * CORR_D * sizeof(half_t) * (a->dtype == VTGB_BF16X3 ? 2 : 1), 256);
}

extern "C" int vtgb_raft_corr(const vtgb_raft_corr_args* a, vtgb_stream_t stream) {
    VTGB_TRY(corr_check(a));
    VTGB_REQUIRE(a->fmap && a->levels[0] && a->levels[1] && a->levels[2] && a->levels[3], VTGB_EINVAL, "raft_corr: NULL operand");
    const bool f32 = a->dtype == VTGB_F32, x3 = a->dtype == VTGB_BF16X3;
    const CorrLds c = x3 ? corr_lds_x3(a->H8, a->W8) : corr_lds(a->H8, a->W8, f32);
    const int HW = a->H8 * a->W8;
    const dim3 grid((unsigned)(((HW + c.tp - 1) / c.tp) * ((a->n_pairs + 7) / 8) * 8));      // (id -> (pair, tile): see the kernel)
    if (x3) {
        const size_t need = vtgb_raft_corr_workspace_bytes(a);
        VTGB_REQUIRE(a->workspace && a->workspace_bytes >= need, VTGB_EWORKSPACE, "raft_corr: workspace %zu < %zu bytes", a->workspace_bytes, need);
        half_t* fh = reinterpret_cast<half_t*>(a->workspace);      // bf16 hi plane | lo plane
        const int64_t n4 = (int64_t)a->n_images * HW * CORR_D / 4;
        hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, a->fmap, reinterpret_cast<bf16_t*>(fh), n4);
        if (c.tp == 32) {
            VTGB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(raft_corr_kernel<false, float, 32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.bytes));
            hipLaunchKernelGGL((raft_corr_kernel<false, float, 32, true>), grid, dim3(512), c.bytes, stream, *a, fh, c.ldS, c.n1, c.n2);
        } else {
            VTGB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(raft_corr_kernel<false, float, 16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.bytes));
            hipLaunchKernelGGL((raft_corr_kernel<false, float, 16, true>), grid, dim3(256), c.bytes, stream, *a, fh, c.ldS, c.n1, c.n2);
        }
    } else if (f32) {
        VTGB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(raft_corr_kernel<true, float, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.bytes));
        hipLaunchKernelGGL((raft_corr_kernel<true, float, 16>), grid, dim3(256), c.bytes, stream, *a, nullptr, c.ldS, c.n1, c.n2);
    } else {
        const size_t need = vtgb_raft_corr_workspace_bytes(a);
        VTGB_REQUIRE(a->workspace && a->workspace_bytes >= need, VTGB_EWORKSPACE, "raft_corr: workspace %zu < %zu bytes", a->workspace_bytes, need);
        half_t* fh = reinterpret_cast<half_t*>(a->workspace);
        const int64_t n4 = (int64_t)a->n_images * HW * CORR_D / 4;
        hipLaunchKernelGGL(cast_f16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, a->fmap, fh, n4);
        if (c.tp == 32) {
            VTGB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(raft_corr_kernel<false, half_t, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.bytes));
            hipLaunchKernelGGL((raft_corr_kernel<false, half_t, 32>), grid, dim3(512), c.bytes, stream, *a, fh, c.ldS, c.n1, c.n2);
        } else {
            VTGB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(raft_corr_kernel<false, half_t, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.bytes));
            hipLaunchKernelGGL((raft_corr_kernel<false, half_t, 16>), grid, dim3(256), c.bytes, stream, *a, fh, c.ldS, c.n1, c.n2);
        }
    }
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
