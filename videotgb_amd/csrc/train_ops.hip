// train_ops.hip -- the training graph's own kernels beside the attention (train_attn.hip) and the loss (train.hip):
//
//   vtgb_gemm_train     out[M, N] fp32 = op(a) . op(b) (+ bias) with either operand read WHERE AUTOGRAD LEFT IT: contraction index
//                       contiguous (x [M, K], W [N, K]: the forward's layout) or contraction index major (k-major: element (row, k) at
//                       p + k * ld + row), stored in fp32 or bf16.  y = x W^T is (k-contiguous, k-contiguous); dX = dY W reads W
//                       k-major; dW = dY^T X reads both k-major -- no `.t().contiguous()` and no `.to(bfloat16)` copy of a weight, an
//                       activation or a gradient (rounds 1-3: three full copies per backward GEMM).  compute = bf16: operands are
//                       rounded to bf16 on their way into LDS (k-major operands are transposed in registers: a thread loads 8 k-rows of 4
//                       (fp32) or 8 (bf16) neighbouring rows and writes 4 / 8 k-contiguous 16-byte pieces), v_mfma_f32_16x16x32_bf16,
//                       fp32 accumulation; compute = fp32: FMA kernel, contraction summed in index order (the exactness mode).
//   vtgb_col_sum_f32    bias gradients (column sums of dY), fixed summation order.
//   vtgb_layernorm_train_forward / _backward   y = LayerNorm(x * mask + resid): the post-LN residual sites of the Q-Former
//                       (xinstructblip.py:707, :788 -> BertSelfOutput / BertOutput) and of the TGB (xropebert.py:542-582) with their dropout
//                       mask folded in; backward returns d(x), d(resid) and deterministic dgamma / dbeta (per-workgroup partials in row
//                       order, then one pass over the partials).
//   vtgb_gelu_forward / _backward   exact (erf) GELU and its derivative Phi(x) + x phi(x).
#include "common.h"

#include <math.h>

namespace {

// ============================================================================ GEMM, bf16 MFMA
constexpr int TG_BM = 128, TG_BN = 128, TG_BK = 64;
constexpr int TG_TILE = TG_BM * TG_BK * 2;       // 16 KiB per operand per buffer

struct TgOperand {
    const void* p;
    int64_t ld;
    int rows;        // valid rows (M for a, N for b)
    int vec;         // 16-byte loads allowed (base and ld aligned)
};
struct TgParams {
    TgOperand a, b;
    int M, N, K;
    const float* bias;
    float* out;
    int64_t ldo;
    int kt_per_split;     // k-tiles per blockIdx.z (split contraction: slice z leaves its fp32 tile in out + z * M * ldo, bias added by the reduction)
};

// LDS image of an operand tile: row-major [128 rows][8 pieces of 8 k] with the piece index XOR-swizzled by the row pair (ds_read_b128 of 16
// rows at one piece position then spreads over all banks)
__device__ __forceinline__ int tg_swz(int row, int piece) { return row * 128 + ((piece ^ ((row >> 1) & 7)) << 4); }

template <bool F32>
struct TgElem;
template <>
struct TgElem<true> {
    typedef float T;
};
template <>
struct TgElem<false> {
    typedef bf16_t T;
};

// One operand's share of a k-tile for one thread: NP 16-byte pieces (8 k of one row), produced from either layout.
template <bool KM, bool F32>
struct TgStage {
    typedef typename TgElem<F32>::T T;
    static constexpr int NP = KM ? (F32 ? 4 : 8) : 4;
    bf16x8 pc[NP];

    __device__ __forceinline__ void load(const TgOperand& op, int r0, int k0, int K, int tid) {
        const T* base = reinterpret_cast<const T*>(op.p);
        if constexpr (!KM) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int q = tid + 256 * i, row = q >> 3, k = k0 + (q & 7) * 8;
                int gr = r0 + row;
                gr = gr < op.rows ? gr : op.rows - 1;            // rows past the edge: any valid row (their outputs are never stored)
                const T* src = base + (int64_t)gr * op.ld + k;
                if (op.vec && k + 8 <= K) {
                    if constexpr (F32) {
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
                        pc[i] = bf16x8{(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
                    } else {
                        pc[i] = *reinterpret_cast<const bf16x8*>(src);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) pc[i][e] = (k + e < K) ? (bf16_t)(float)src[e] : (bf16_t)0.f;
                }
            }
        } else {
            constexpr int RW = F32 ? 4 : 8;                       // neighbouring rows per 16-byte load
            const int rg = F32 ? (tid & 31) : (tid & 15), kp = F32 ? (tid >> 5) : (tid >> 4);
            if (!F32 && tid >= 128) return;
            const int r = r0 + rg * RW, k = k0 + kp * 8;
            T v[8][RW];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const T* src = base + (int64_t)(k + j) * op.ld + r;
                if (k + j < K && op.vec && r + RW <= op.rows) {
                    if constexpr (F32) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                        for (int e = 0; e < 4; e++) v[j][e] = t[e];
                    } else {
                        const bf16x8 t = *reinterpret_cast<const bf16x8*>(src);
#pragma unroll
                        for (int e = 0; e < 8; e++) v[j][e] = t[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < RW; e++) v[j][e] = (k + j < K && r + e < op.rows) ? src[e] : (T)0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < RW; e++)
#pragma unroll
                for (int j = 0; j < 8; j++) pc[e][j] = (bf16_t)(float)v[j][e];
        }
    }

    __device__ __forceinline__ void write(char* tile, int tid) const {
        if constexpr (!KM) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int q = tid + 256 * i;
                *reinterpret_cast<bf16x8*>(tile + tg_swz(q >> 3, q & 7)) = pc[i];
            }
        } else {
            constexpr int RW = F32 ? 4 : 8;
            const int rg = F32 ? (tid & 31) : (tid & 15), kp = F32 ? (tid >> 5) : (tid >> 4);
            if (!F32 && tid >= 128) return;
#pragma unroll
            for (int e = 0; e < RW; e++) *reinterpret_cast<bf16x8*>(tile + tg_swz(rg * RW + e, kp)) = pc[e];
        }
    }
};

template <bool AKM, bool AF32, bool BKM, bool BF32>
__global__ __launch_bounds__(256, 2) void tg_mfma_kernel(const TgParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + 2 * TG_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.y * TG_BM, n0 = blockIdx.x * TG_BN;
    TgStage<AKM, AF32> sa;
    TgStage<BKM, BF32> sb;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kt0 = blockIdx.z * p.kt_per_split, nk = min((p.K + TG_BK - 1) / TG_BK, kt0 + p.kt_per_split);
    sa.load(p.a, m0, kt0 * TG_BK, p.K, tid);
    sb.load(p.b, n0, kt0 * TG_BK, p.K, tid);
    sa.write(As, tid);
    sb.write(Bs, tid);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = kt0; kt < nk; kt++) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < nk) {
            sa.load(p.a, m0, (kt + 1) * TG_BK, p.K, tid);
            sb.load(p.b, n0, (kt + 1) * TG_BK, p.K, tid);
        }
        const char* as = As + buf * TG_TILE;
        const char* bs = Bs + buf * TG_TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 bf[4], af[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                bf[i] = *reinterpret_cast<const bf16x8*>(bs + tg_swz(wn * 64 + i * 16 + fr, ks * 4 + fg));
                af[i] = *reinterpret_cast<const bf16x8*>(as + tg_swz(wm * 64 + i * 16 + fr, ks * 4 + fg));
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[i], af[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            sa.write(As + (buf ^ 1) * TG_TILE, tid);
            sb.write(Bs + (buf ^ 1) * TG_TILE, tid);
        }
        __syncthreads();
    }
    // D: column (lane & 15) <- a row (m); rows (lane >> 4) * 4 + reg <- b row (n): four consecutive n per lane
    const bool vst = (p.ldo & 3) == 0;
    float* outz = p.out + (int64_t)blockIdx.z * p.M * p.ldo;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int m = m0 + wm * 64 + j * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int n = n0 + wn * 64 + i * 16 + fg * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            float* o = outz + (int64_t)m * p.ldo + n;
            if (n + 3 < p.N) {
                if (p.bias) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] += p.bias[n + e];
                }
                if (vst) {
                    *reinterpret_cast<f32x4*>(o) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = v[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (n + e < p.N) o[e] = v[e] + (p.bias ? p.bias[n + e] : 0.f);
            }
        }
    }
}

// split contraction: out[m, n] = sum_z part[z][m][n] (+ bias[n]) in slice order
__global__ __launch_bounds__(256) void tg_split_reduce_kernel(const float* __restrict__ part, int splits, int M, int N, const float* __restrict__ bias,
                                                              float* __restrict__ out, int64_t ldo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
    float s = part[i];
    for (int z = 1; z < splits; z++) s += part[(int64_t)z * M * N + i];
    out[(int64_t)m * ldo + n] = s + (bias ? bias[n] : 0.f);
}

// ============================================================================ GEMM, fp32 FMA (exactness mode)
// 64 x 64 tile, 16 k per step, 4 x 4 outputs per thread; element (row, k) of an operand at p + row * rs + k * ks, so both layouts
// are strides.  The contraction is summed in index order with one fmaf per term, as gemm_f32_kernel (gemm.hip) does.
struct TgF32Params {
    const void* a;
    const void* b;
    int64_t a_rs, a_ks, b_rs, b_ks;
    int a_km, b_km;
    int M, N, K;
    const float* bias;
    float* out;
    int64_t ldo;
};
template <typename TA, typename TB>
__global__ __launch_bounds__(256) void tg_f32_kernel(const TgF32Params p) {
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const TA* __restrict__ A = reinterpret_cast<const TA*>(p.a);
    const TB* __restrict__ B = reinterpret_cast<const TB*>(p.b);
    // staging: the thread index runs along the contiguous direction of the operand
    const int arow = p.a_km ? (tid & 63) : (tid >> 2), ak = p.a_km ? (tid >> 6) * 4 : (tid & 3) * 4;
    const int brow = p.b_km ? (tid & 63) : (tid >> 2), bk = p.b_km ? (tid >> 6) * 4 : (tid & 3) * 4;
    int am = m0 + arow; am = am < p.M ? am : p.M - 1;
    int bn = n0 + brow; bn = bn < p.N ? bn : p.N - 1;
    const TA* ap = A + (int64_t)am * p.a_rs;
    const TB* bp = B + (int64_t)bn * p.b_rs;
    // r5: the products run on v_mfma_f32_32x32x2_f32 -- bit for bit the k-ordered fmaf chain of the scalar loop it replaces (cdna_hip_programming.md,
    // "FP32-input MFMA").  Four waves of 32 x 32; the b slab is the A operand (rows = n), so a lane holds runs of four consecutive n of one row m.
    typedef float tg_f32x16 __attribute__((ext_vector_type(16)));
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    tg_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ka = k0 + ak + i, kb = k0 + bk + i;
            As[ak + i][arow] = ka < p.K ? (float)ap[(int64_t)ka * p.a_ks] : 0.f;
            Bs[bk + i][brow] = kb < p.K ? (float)bp[(int64_t)kb * p.b_ks] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 8; kk++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Bs[2 * kk + kh][wn * 32 + l31], As[2 * kk + kh][wm * 32 + l31], acc, 0, 0, 0);
        __syncthreads();
    }
    const int m = m0 + wm * 32 + l31;
    if (m < p.M) {
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int n = n0 + wn * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            if (n < p.N) p.out[(int64_t)m * p.ldo + n] = acc[reg] + (p.bias ? p.bias[n] : 0.f);
        }
    }
}

// ============================================================================ column sums
// out[n] = sum_m x[m, n]: the rows are cut into `parts` runs; a workgroup sums one run of 64 columns (four row phases in row order, combined in
// phase order) into part[run][n], a second launch adds the runs in order.  (One launch over whole columns was 12 workgroups for a 768-wide
// gradient: ~100 us per bias gradient, as long as the weight-gradient GEMM beside it.)
constexpr int CS_ROWS = 128;        // rows per run
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ x, int64_t ldx, int M, int N, float* __restrict__ part) {
    __shared__ float ph[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, n = blockIdx.x * 64 + tx;
    const int m0 = blockIdx.y * CS_ROWS, m1 = min(M, m0 + CS_ROWS);
    float s = 0.f;
    if (n < N)
        for (int m = m0 + ty; m < m1; m += 4) s += x[(int64_t)m * ldx + n];
    ph[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < N) part[(int64_t)blockIdx.y * N + n] = ((ph[0][tx] + ph[1][tx]) + ph[2][tx]) + ph[3][tx];
}
__global__ __launch_bounds__(256) void col_sum_reduce_kernel(const float* __restrict__ part, int parts, int N, float* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int i = 0; i < parts; i++) s += part[(int64_t)i * N + n];
    out[n] = s;
}

// ============================================================================ LayerNorm (training)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int LN_ROWS_PER_WG = 32;      // backward: rows per workgroup (8 per wave) = one partial row of dgamma / dbeta
constexpr int LN_MAX_V4 = 8;            // float4 per lane: D <= 2048

// one wave per row: s = x * mask + resid; mean; variance about the mean (two passes over the cache-hot row); y
__global__ __launch_bounds__(256) void ln_train_fwd_kernel(const vtgb_layernorm_train_args a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int D = a.D;
    const float* x = a.x + (int64_t)row * D;
    const float* mk = a.mask ? a.mask + (int64_t)row * D : nullptr;
    const float* rs = a.resid ? a.resid + (int64_t)row * D : nullptr;
    float* sp = a.sum ? a.sum + (int64_t)row * D : nullptr;
    float* y = a.y + (int64_t)row * D;
    float acc = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
        if (mk) v *= *reinterpret_cast<const f32x4*>(mk + c);
        if (rs) v += *reinterpret_cast<const f32x4*>(rs + c);
        if (sp) *reinterpret_cast<f32x4*>(sp + c) = v;
        *reinterpret_cast<f32x4*>(y + c) = v;                    // parked in y: the next passes read it back (this lane's own elements)
        acc += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = wave_sum(acc) / (float)D;
    float var = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + c) - mean;
        var += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const float rstd = rsqrtf(wave_sum(var) / (float)D + a.eps);
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = (*reinterpret_cast<const f32x4*>(y + c) - mean) * rstd;
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + c), b = *reinterpret_cast<const f32x4*>(a.beta + c);
        *reinterpret_cast<f32x4*>(y + c) = v * g + b;
    }
    if (lane == 0) {
        a.mean[row] = mean;
        a.rstd[row] = rstd;
    }
}

// one wave per row, LN_ROWS_PER_WG rows per workgroup: ds = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma; the lanes keep their
// columns' dgamma / dbeta sums over the wave's rows in registers, the four waves are combined in wave order -> partial[wg]
template <int NV>
__global__ __launch_bounds__(256) void ln_train_bwd_kernel(const vtgb_layernorm_train_args a) {
    __shared__ float red[3][2][NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = a.D;
    f32x4 dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) dg[i] = db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r_begin = blockIdx.x * LN_ROWS_PER_WG + wave * (LN_ROWS_PER_WG / 4);
    for (int rr = 0; rr < LN_ROWS_PER_WG / 4; rr++) {
        const int row = r_begin + rr;
        if (row >= a.rows) break;
        const float* s = a.sum + (int64_t)row * D;
        const float* dy = a.dy + (int64_t)row * D;
        const float mean = a.mean[row], rstd = a.rstd[row];
        f32x4 g[NV], xh[NV];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int c = lane * 4 + i * 256;
            g[i] = xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < D) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + c);
                xh[i] = (*reinterpret_cast<const f32x4*>(s + c) - mean) * rstd;
                g[i] = d * *reinterpret_cast<const f32x4*>(a.gamma + c);
                dg[i] += d * xh[i];
                db[i] += d;
                c1 += (g[i][0] + g[i][1]) + (g[i][2] + g[i][3]);
                const f32x4 gx = g[i] * xh[i];
                c2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
            }
        }
        c1 = wave_sum(c1) / (float)D;
        c2 = wave_sum(c2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int c = lane * 4 + i * 256;
            if (c < D) {
                const f32x4 d = (g[i] - c1 - xh[i] * c2) * rstd;
                *reinterpret_cast<f32x4*>(a.ds + (int64_t)row * D + c) = d;
                if (a.dx) *reinterpret_cast<f32x4*>(a.dx + (int64_t)row * D + c) = d * *reinterpret_cast<const f32x4*>(a.mask + (int64_t)row * D + c);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            *reinterpret_cast<f32x4*>(&red[wave - 1][0][i * 256 + lane * 4]) = dg[i];
            *reinterpret_cast<f32x4*>(&red[wave - 1][1][i * 256 + lane * 4]) = db[i];
        }
    }
    __syncthreads();
    if (wave == 0) {
        float* pg = a.partial + (int64_t)blockIdx.x * 2 * D;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int c = lane * 4 + i * 256;
            if (c < D) {
                f32x4 sg = dg[i], sb = db[i];
#pragma unroll
                for (int w = 0; w < 3; w++) {
                    sg += *reinterpret_cast<const f32x4*>(&red[w][0][i * 256 + lane * 4]);
                    sb += *reinterpret_cast<const f32x4*>(&red[w][1][i * 256 + lane * 4]);
                }
                *reinterpret_cast<f32x4*>(pg + c) = sg;
                *reinterpret_cast<f32x4*>(pg + D + c) = sb;
            }
        }
    }
}

__global__ __launch_bounds__(256) void ln_train_bwd_reduce_kernel(const float* __restrict__ partial, int parts, int D, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= 2 * D) return;
    float s = 0.f;
    for (int i = 0; i < parts; i++) s += partial[(int64_t)i * 2 * D + c];
    if (c < D) dgamma[c] = s;
    else dbeta[c - D] = s;
}

// ============================================================================ GELU (erf)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
        *reinterpret_cast<f32x4*>(y + i) = o;
    } else {
        for (int64_t j = i; j < n; j++) y[j] = 0.5f * x[j] * (1.0f + erff(x[j] * 0.70710678118654752440f));
    }
}
__device__ __forceinline__ float gelu_grad(float v) {
    return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.39894228040143267794f * __expf(-0.5f * v * v);
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i), d = *reinterpret_cast<const f32x4*>(dy + i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = d[e] * gelu_grad(v[e]);
        *reinterpret_cast<f32x4*>(dx + i) = o;
    } else {
        for (int64_t j = i; j < n; j++) dx[j] = dy[j] * gelu_grad(x[j]);
    }
}

// ============================================================================ launch helpers
// Few output tiles and a long contraction (a weight gradient: dW [768, 1408] over 8224 tokens is 66 tiles of 129 k-tiles): the contraction is cut
// into slices so that ~2 workgroups per CU exist; every slice leaves an fp32 tile in the workspace and a second launch adds them in slice order.
int tg_splits(int M, int N, int K) {
    const int tiles = ((M + TG_BM - 1) / TG_BM) * ((N + TG_BN - 1) / TG_BN), nk = (K + TG_BK - 1) / TG_BK;
    if (tiles > 128 || nk < 8) return 1;
    const int want = (512 + tiles - 1) / tiles, most = nk / 4;
    const int sp = want < most ? want : most;
    return sp < 2 ? 1 : sp;
}

template <bool AKM, bool AF32, bool BKM, bool BF32>
int launch_tg(const TgParams& p0, float* workspace, size_t workspace_bytes, hipStream_t s) {
    static DeviceOnce attr;
    VTGB_FUNC_LDS_ONCE(attr, (tg_mfma_kernel<AKM, AF32, BKM, BF32>), 4 * TG_TILE);
    TgParams p = p0;
    const int nk = (p.K + TG_BK - 1) / TG_BK;
    int splits = tg_splits(p.M, p.N, p.K);
    if (splits > 1 && (!workspace || workspace_bytes < (size_t)splits * p.M * p.N * sizeof(float))) splits = 1;      // no workspace: unsplit, same result class
    p.kt_per_split = (nk + splits - 1) / splits;
    splits = (nk + p.kt_per_split - 1) / p.kt_per_split;
    if (splits > 1) {
        p.out = workspace;
        p.ldo = p.N;
        p.bias = nullptr;
    }
    hipLaunchKernelGGL((tg_mfma_kernel<AKM, AF32, BKM, BF32>), dim3((p.N + TG_BN - 1) / TG_BN, (p.M + TG_BM - 1) / TG_BM, splits), dim3(256), 4 * TG_TILE, s, p);
    VTGB_HIP(hipGetLastError());
    if (splits > 1) {
        hipLaunchKernelGGL(tg_split_reduce_kernel, dim3((unsigned)(((int64_t)p.M * p.N + 255) / 256)), dim3(256), 0, s, workspace, splits, p.M, p.N, p0.bias,
                           p0.out, p0.ldo);
        VTGB_HIP(hipGetLastError());
    }
    return VTGB_OK;
}
template <bool AKM, bool AF32>
int launch_tg_b(const TgParams& p, bool bkm, bool bf32, float* ws, size_t wsb, hipStream_t s) {
    if (bkm) return bf32 ? launch_tg<AKM, AF32, true, true>(p, ws, wsb, s) : launch_tg<AKM, AF32, true, false>(p, ws, wsb, s);
    return bf32 ? launch_tg<AKM, AF32, false, true>(p, ws, wsb, s) : launch_tg<AKM, AF32, false, false>(p, ws, wsb, s);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int vtgb_gemm_train(const vtgb_gemm_train_args* g, vtgb_stream_t stream) {
    VTGB_REQUIRE(g && g->a && g->b && g->out, VTGB_EINVAL, "gemm_train: NULL argument");
    VTGB_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0, VTGB_EINVAL, "gemm_train: M=%d N=%d K=%d", g->M, g->N, g->K);
    VTGB_REQUIRE(g->compute == VTGB_BF16 || g->compute == VTGB_F32, VTGB_EINVAL, "gemm_train: compute type %d", g->compute);
    VTGB_REQUIRE((g->a_dtype == VTGB_BF16 || g->a_dtype == VTGB_F32) && (g->b_dtype == VTGB_BF16 || g->b_dtype == VTGB_F32), VTGB_EINVAL,
                 "gemm_train: operand storage types %d, %d", g->a_dtype, g->b_dtype);
    VTGB_REQUIRE(g->lda >= (g->a_kmajor ? g->M : g->K) && g->ldb >= (g->b_kmajor ? g->N : g->K) && g->ldo >= g->N, VTGB_EINVAL,
                 "gemm_train: leading dimensions %lld, %lld, %lld too small for M=%d N=%d K=%d", (long long)g->lda, (long long)g->ldb, (long long)g->ldo,
                 g->M, g->N, g->K);
    hipStream_t s = (hipStream_t)stream;
    const bool af32 = g->a_dtype == VTGB_F32, bf32 = g->b_dtype == VTGB_F32;
    if (g->compute == VTGB_F32) {
        TgF32Params p{g->a, g->b, g->a_kmajor ? 1 : g->lda, g->a_kmajor ? g->lda : 1, g->b_kmajor ? 1 : g->ldb, g->b_kmajor ? g->ldb : 1,
                      g->a_kmajor != 0, g->b_kmajor != 0, g->M, g->N, g->K, g->bias, g->out, g->ldo};
        const dim3 grid((g->N + 63) / 64, (g->M + 63) / 64);
        if (af32 && bf32) hipLaunchKernelGGL((tg_f32_kernel<float, float>), grid, dim3(256), 0, s, p);
        else if (af32) hipLaunchKernelGGL((tg_f32_kernel<float, bf16_t>), grid, dim3(256), 0, s, p);
        else if (bf32) hipLaunchKernelGGL((tg_f32_kernel<bf16_t, float>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((tg_f32_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, p);
        VTGB_HIP(hipGetLastError());
        return VTGB_OK;
    }
    TgParams p;
    const int64_t a_el = af32 ? 4 : 8, b_el = bf32 ? 4 : 8;          // elements per 16-byte load
    p.a = TgOperand{g->a, g->lda, g->M, aligned16(g->a) && (g->lda % a_el) == 0};
    p.b = TgOperand{g->b, g->ldb, g->N, aligned16(g->b) && (g->ldb % b_el) == 0};
    p.M = g->M; p.N = g->N; p.K = g->K;
    p.bias = g->bias; p.out = g->out; p.ldo = g->ldo;
    p.kt_per_split = 0;
    float* ws = reinterpret_cast<float*>(g->workspace);
    const size_t wsb = g->workspace_bytes;
    if (g->a_kmajor) return af32 ? launch_tg_b<true, true>(p, g->b_kmajor != 0, bf32, ws, wsb, s) : launch_tg_b<true, false>(p, g->b_kmajor != 0, bf32, ws, wsb, s);
    return af32 ? launch_tg_b<false, true>(p, g->b_kmajor != 0, bf32, ws, wsb, s) : launch_tg_b<false, false>(p, g->b_kmajor != 0, bf32, ws, wsb, s);
}

extern "C" size_t vtgb_gemm_train_workspace_bytes(const vtgb_gemm_train_args* g) {
    if (!g || g->compute != VTGB_BF16 || g->M <= 0 || g->N <= 0 || g->K <= 0) return 0;
    const int sp = tg_splits(g->M, g->N, g->K);
    return sp > 1 ? (size_t)sp * g->M * g->N * sizeof(float) : 0;
}

extern "C" int32_t vtgb_col_sum_parts(int32_t M) { return M > 0 ? (M + CS_ROWS - 1) / CS_ROWS : 0; }

extern "C" int vtgb_col_sum_f32(const float* x, int64_t ldx, int32_t M, int32_t N, float* out, float* partial, vtgb_stream_t stream) {
    VTGB_REQUIRE(x && out && partial && M > 0 && N > 0 && ldx >= N, VTGB_EINVAL, "col_sum_f32: x=%p out=%p partial=%p M=%d N=%d ldx=%lld", (const void*)x,
                 (void*)out, (void*)partial, M, N, (long long)ldx);
    const int parts = vtgb_col_sum_parts(M);
    hipLaunchKernelGGL(col_sum_kernel, dim3((N + 63) / 64, parts), dim3(256), 0, (hipStream_t)stream, x, ldx, M, N, partial);
    VTGB_HIP(hipGetLastError());
    hipLaunchKernelGGL(col_sum_reduce_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial, parts, N, out);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

extern "C" int32_t vtgb_layernorm_train_partials(int32_t rows) { return rows > 0 ? (rows + LN_ROWS_PER_WG - 1) / LN_ROWS_PER_WG : 0; }

static int ln_train_check(const vtgb_layernorm_train_args* a, const char* what) {
    VTGB_REQUIRE(a && a->gamma && a->mean && a->rstd, VTGB_EINVAL, "%s: NULL argument", what);
    VTGB_REQUIRE(a->rows > 0 && a->D > 0 && (a->D & 3) == 0 && a->D <= LN_MAX_V4 * 256, VTGB_EUNSUPPORTED,
                 "%s: rows=%d D=%d (D must be a multiple of 4, at most %d)", what, a->rows, a->D, LN_MAX_V4 * 256);
    return VTGB_OK;
}

extern "C" int vtgb_layernorm_train_forward(const vtgb_layernorm_train_args* a, vtgb_stream_t stream) {
    VTGB_TRY(ln_train_check(a, "layernorm_train_forward"));
    VTGB_REQUIRE(a->x && a->beta && a->y, VTGB_EINVAL, "layernorm_train_forward: NULL x / beta / y");
    VTGB_REQUIRE(a->sum || (!a->mask && !a->resid), VTGB_EINVAL, "layernorm_train_forward: `sum` is required with a mask or a residual");
    hipLaunchKernelGGL(ln_train_fwd_kernel, dim3((a->rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

extern "C" int vtgb_layernorm_train_backward(const vtgb_layernorm_train_args* a, vtgb_stream_t stream) {
    VTGB_TRY(ln_train_check(a, "layernorm_train_backward"));
    VTGB_REQUIRE(a->sum && a->dy && a->ds && a->dgamma && a->dbeta && a->partial, VTGB_EINVAL, "layernorm_train_backward: NULL sum / dy / ds / dgamma / dbeta / partial");
    VTGB_REQUIRE(!a->dx || a->mask, VTGB_EINVAL, "layernorm_train_backward: dx without a mask (dx = ds)");
    hipStream_t s = (hipStream_t)stream;
    const int parts = vtgb_layernorm_train_partials(a->rows), nv = (a->D + 255) / 256;
    const dim3 grid(parts), block(256);
    if (nv <= 1) hipLaunchKernelGGL(ln_train_bwd_kernel<1>, grid, block, 0, s, *a);
    else if (nv <= 2) hipLaunchKernelGGL(ln_train_bwd_kernel<2>, grid, block, 0, s, *a);
    else if (nv <= 3) hipLaunchKernelGGL(ln_train_bwd_kernel<3>, grid, block, 0, s, *a);
    else if (nv <= 4) hipLaunchKernelGGL(ln_train_bwd_kernel<4>, grid, block, 0, s, *a);
    else if (nv <= 6) hipLaunchKernelGGL(ln_train_bwd_kernel<6>, grid, block, 0, s, *a);
    else hipLaunchKernelGGL(ln_train_bwd_kernel<8>, grid, block, 0, s, *a);
    VTGB_HIP(hipGetLastError());
    hipLaunchKernelGGL(ln_train_bwd_reduce_kernel, dim3((2 * a->D + 255) / 256), block, 0, s, a->partial, parts, a->D, a->dgamma, a->dbeta);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

extern "C" int vtgb_gelu_forward(const float* x, float* y, int64_t n, vtgb_stream_t stream) {
    VTGB_REQUIRE(x && y && n > 0, VTGB_EINVAL, "gelu_forward: x=%p y=%p n=%lld", (const void*)x, (void*)y, (long long)n);
    VTGB_REQUIRE(aligned16(x) && aligned16(y), VTGB_EINVAL, "gelu_forward: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

extern "C" int vtgb_gelu_backward(const float* x, const float* dy, float* dx, int64_t n, vtgb_stream_t stream) {
    VTGB_REQUIRE(x && dy && dx && n > 0, VTGB_EINVAL, "gelu_backward: x=%p dy=%p dx=%p n=%lld", (const void*)x, (const void*)dy, (void*)dx, (long long)n);
    VTGB_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx), VTGB_EINVAL, "gelu_backward: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
