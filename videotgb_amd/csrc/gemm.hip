// gemm.hip -- out[M,N] = epilogue(A[M,K] . W[N,K]^T + bias) for gfx950.
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights [out,in]), which
// is the natural MFMA layout: every lane's fragment is 16 contiguous bytes along K.
//
// bf16 path: 128x128x64 tile, 256 threads = 4 waves in 2(M) x 2(N), each wave 64x64 as 4x4
// v_mfma_f32_16x16x32_bf16 tiles (64 accumulator VGPRs).  W rows are the MFMA "A" operand
// and X rows the "B" operand, so a lane ends up with 4 consecutive n for one m: bias, residual
// and stores are 8/16-byte vectors.  Global -> register -> LDS staging with the next k-tile's
// loads issued before the current tile's MFMAs (one barrier per k-tile, two LDS buffers);
// LDS rows are 128 B with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7 so the 16
// rows of a fragment read hit 16 distinct 16-byte slots of the 256-byte bank row.
//
// fp32 path (exactness mode): plain 64x64x16 register-tiled FMA kernel, k summed in order.
#include <stdlib.h>

#include "common.h"
#include "gemm_dev.h"

// Store 4 consecutive n (n0..n0+3) of logical row m.  TAct is the activation type of
// EPI_STORE / EPI_GELU outputs.
template <int EPI, typename TAct, bool FAST_GELU = false, bool ADD_BIAS = true>
__device__ __forceinline__ void epilogue4(const GemmDesc& p, int m, int n0, float v0, float v1, float v2, float v3) {
    if (m >= p.M || n0 >= p.N) return;
    float v[4] = {v0, v1, v2, v3};
    const bool full = (n0 + 3 < p.N);
    if (ADD_BIAS && p.bias) {
        if (full) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + n0);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < p.N) v[i] += p.bias[n0 + i];
        }
    }
    const int64_t orow = map_row(p.o_map, m);
    if (EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32) {
        float* o = reinterpret_cast<float*>(p.out) + orow * p.ldo + n0;
        if (EPI == EPI_RESID_F32) {
            const float* r = p.resid + map_row(p.r_map, m) * p.ldr + n0;
            if (full && ((p.ldr & 3) == 0)) {
                const float4 rv = *reinterpret_cast<const float4*>(r);
                v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
            } else {
    #pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < p.N) v[i] += r[i];
            }
        }
        if (full && ((p.ldo & 3) == 0)) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < p.N) o[i] = v[i];
        }
    } else {
        if (EPI == EPI_GELU) {
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = FAST_GELU ? gelu_erf_fast(v[i]) : gelu_erf(v[i]);
        }
        TAct* o = reinterpret_cast<TAct*>(p.out) + orow * p.ldo + n0;
        if (full && ((p.ldo & 3) == 0)) {
            if constexpr (sizeof(TAct) == 2) {
                bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(o) = pk;
            } else {
                *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < p.N) o[i] = (TAct)v[i];
        }
    }
}

// ---------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ---------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per buffer


template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmDesc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                   // X tile (m rows)   [2][TILE_BYTES]
    char* Ws = smem + 2 * TILE_BYTES;  // W tile (n rows)   [2][TILE_BYTES]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(p.W);

    // staging assignment: 1024 16-byte chunks per operand tile, 4 per thread (named, so nothing
    // is runtime-indexed and everything stays in registers)
#define STAGE_SETUP(i)                                                              \
    const int q##i = tid + 256 * i, row##i = q##i >> 3, c##i = q##i & 7;            \
    const int am##i = (m0 + row##i) < p.M ? (m0 + row##i) : p.M - 1;                \
    const int wr##i = (n0 + row##i) < p.N ? (n0 + row##i) : p.N - 1;                \
    const bf16_t* a_ptr##i = A + map_row(p.a_map, am##i) * p.lda + c##i * 8;        \
    const bf16_t* w_ptr##i = W + (int64_t)wr##i * p.ldw + c##i * 8;                 \
    const int lds_off##i = swz(row##i, c##i);                                       \
    const int kc##i = c##i * 8;
    STAGE_SETUP(0) STAGE_SETUP(1) STAGE_SETUP(2) STAGE_SETUP(3)
#undef STAGE_SETUP
    uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
#define STAGE_LOAD1(i, RA, RW, k0)                                                        \
    {                                                                                      \
        const bool ok = ((k0) + kc##i) < p.K;   /* out-of-range chunk: read k=0, then zero */ \
        const int kk = ok ? (k0) : -kc##i;                                                 \
        RA = *reinterpret_cast<const uint4*>(a_ptr##i + kk);                               \
        RW = *reinterpret_cast<const uint4*>(w_ptr##i + kk);                               \
        if (!ok) { RA = zero4; RW = zero4; }                                               \
    }
#define STAGE_LOAD(k0) STAGE_LOAD1(0, ra0, rw0, k0) STAGE_LOAD1(1, ra1, rw1, k0) STAGE_LOAD1(2, ra2, rw2, k0) STAGE_LOAD1(3, ra3, rw3, k0)
#define STAGE_WRITE1(i, RA, RW, buf)                                                      \
    *reinterpret_cast<uint4*>(As + (buf) * TILE_BYTES + lds_off##i) = RA;                  \
    *reinterpret_cast<uint4*>(Ws + (buf) * TILE_BYTES + lds_off##i) = RW;
#define STAGE_WRITE(buf) STAGE_WRITE1(0, ra0, rw0, buf) STAGE_WRITE1(1, ra1, rw1, buf) STAGE_WRITE1(2, ra2, rw2, buf) STAGE_WRITE1(3, ra3, rw3, buf)

    // the accumulators START at the bias, as in the large kernels (r5): bias + sum_k, then the residual -- the same association whichever
    // kernel the problem size selects, so a row's result does not depend on how many rows the call holds (tests/test_gpu_scale.py)
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int nb = n0 + (wave >> 1) * 64 + i * 16 + (lane >> 4) * 4;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int e = 0; e < 4; e++) if (nb + e < p.N) b4[e] = p.bias[nb + e];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = b4;
    }

    const int nk = (p.K + BK - 1) / BK;
    STAGE_LOAD(0)
    STAGE_WRITE(0)
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) { STAGE_LOAD((kt + 1) * BK) }
        const char* as = As + buf * TILE_BYTES;
        const char* ws = Ws + buf * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int rw_ = wn * 64 + i * 16 + fr;
                wf[i] = *reinterpret_cast<const bf16x8*>(ws + swz(rw_, ks * 4 + fg));
                const int rx = wm * 64 + i * 16 + fr;
                xf[i] = *reinterpret_cast<const bf16x8*>(as + swz(rx, ks * 4 + fg));
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) { STAGE_WRITE(buf ^ 1) }
        __syncthreads();
    }
#undef STAGE_LOAD
#undef STAGE_LOAD1
#undef STAGE_WRITE
#undef STAGE_WRITE1
    // D layout: column (lane & 15) <- X row (m), rows (lane >> 4) * 4 + reg <- W row (n)
    // LayerNorm folded into the GEMM (GemmDesc::ln_*; gemm_dev.h): the same arithmetic as the persistent kernel's epilogues
    if constexpr (EPI == EPI_STORE || EPI == EPI_GELU) {
        if (p.ln_stats) {
            f32x4 cs4[4], c4[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int nc = min(n0 + wn * 64 + i * 16 + fg * 4, p.N - 4);
                cs4[i] = *reinterpret_cast<const f32x4*>(p.ln_cs + nc);
                c4[i] = *reinterpret_cast<const f32x4*>(p.ln_c + nc);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int mr = min(m0 + wm * 64 + j * 16 + fr, p.M - 1);
                const float2 st = *reinterpret_cast<const float2*>(p.ln_stats + (int64_t)mr * 2);
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i][j] = ln_fold4(acc[i][j], st.x, st.y, cs4[i], c4[i]);
            }
        }
    }
    if constexpr (EPI == EPI_RESID_F32) {
        if (p.ln_xb) {      // producer: x = acc + resid -> fp32 out, bf16 copy, (sum, sum of squares) of the row's 64-column block
            const int nblk = (p.N + 63) >> 6, blk = (n0 + wn * 64) >> 6;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int m = m0 + wm * 64 + j * 16 + fr;
                const bool mok = m < p.M;
                float s4[4], q4[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int n = n0 + wn * 64 + i * 16 + fg * 4;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (mok && n < p.N) {
                        v = acc[i][j] + *reinterpret_cast<const f32x4*>(p.resid + (int64_t)m * p.ldr + n);
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n) = v;
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.ln_xb) + (int64_t)m * p.ldxb + n) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                    }
                    ln_part4(v, s4[i], q4[i]);
                    s4[i] += __shfl_xor(s4[i], 16); q4[i] += __shfl_xor(q4[i], 16);      // column groups cl = 4 i + fg: cl ^ 1, cl ^ 2 are lanes, cl ^ 4, cl ^ 8 registers
                    s4[i] += __shfl_xor(s4[i], 32); q4[i] += __shfl_xor(q4[i], 32);
                }
                const float S = (s4[0] + s4[1]) + (s4[2] + s4[3]), Q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
                if (mok && fg == 0 && blk < nblk) *reinterpret_cast<float2*>(p.ln_part + ((int64_t)m * nblk + blk) * 2) = make_float2(S, Q);
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int m = m0 + wm * 64 + j * 16 + fr;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int n = n0 + wn * 64 + i * 16 + fg * 4;
            epilogue4<EPI, bf16_t, true, false>(p, m, n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// bf16 MFMA kernel, large-problem variant: 256x256x64 tile, 512 threads = 8 waves in
// 2(M) x 4(N), each wave 128(m) x 64(n) = 8 x 4 v_mfma_f32_16x16x32_bf16 tiles (128 accumulator
// VGPRs).  Operands are staged by LDS-DMA (buffer_load_dwordx4 ... lds through per-tile descriptors:
// no VGPR round trip, no ds_write, no 64-bit per-lane address arithmetic in the k-loop; round 1 used
// global_load_lds_dwordx4).  The whole 160 KiB of LDS is the staging ring: THREE 32 KiB slots for the
// activation tile (streamed from HBM: two k-tiles of lookahead) and TWO for the weight tile
// (L2 / Infinity-Cache resident: one k-tile of lookahead).  The k-loop is rotated so that a
// wave's LDS fragment reads (two 48-register sets) always run under its own MFMAs; once per
// k-tile there is a COUNTED s_waitcnt vmcnt(4) -- everything but the four youngest DMAs has
// landed -- and one barrier, after which the freed slots are re-armed, the DMA pieces spread between
// the MFMA groups.  The prologue issues the first k-tile's DMAs before anything else is loaded
// (start values, later k-tiles); every epilogue finishes its LDS reads and operand loads before its
// first store.  (Evidence for each of these choices: DESIGN.md section 4.)
// The LDS image is lane-linear per DMA instruction (8 rows x 128 B per wave-instruction), so
// the bank swizzle is applied to the per-lane SOURCE chunk and to the fragment reads (same
// involution: slot = chunk ^ ((row >> 1) & 7); SQ_LDS_BANK_CONFLICT = 0 measured).
// Workgroup -> tile map is XCD-aware: the 8 XCDs (private 4 MiB L2 each) take interleaved
// m-tiles, and the ~32 workgroups resident on one XCD form a 4 (m) x 8 (n) super-tile so
// that every A k-slice fetched into that L2 is used by 8 workgroups and every W k-slice by 4.
// Requires K % 64 == 0 (every GEMM of the full-size path); anything else uses the kernel above.
// ABL != 0: timing-only ablation builds for tools/gemm_ablate.py (results are wrong).
// ---------------------------------------------------------------------------------------
constexpr int L_BM = 256, L_BN = 256, L_BK = 64;
__device__ __forceinline__ bool gru_staged(const GemmDesc& p) {
    return ((p.N | p.ldo | p.ldr | p.ldaux | p.ldo2) & 3) == 0;
}
constexpr int L_OP_BYTES = 256 * L_BK * 2;   // 32 KiB per operand tile
constexpr int L_A_SLOTS = 3, L_W_SLOTS = 2;
constexpr int L_LDS = (L_A_SLOTS + L_W_SLOTS) * L_OP_BYTES;   // 160 KiB

// NWN (waves along N): 4 = the 256 x 256 tile above; 2 / 1 = 256 x 128 / 256 x 64 tiles for narrow outputs
// (RAFT's 64..128-channel convolutions): waves 4(M) x 2(N) of 64 x 64 or 8(M) x 1(N) of 32 x 64, so that all
// eight waves have work and only the output channels that exist are staged (weight slots of 16 / 8 KiB).
// debug: shader-clock cycles and 100 MHz reference ticks spent inside the large kernel (summed over
// workgroups), so that tools/gemm_ablate.py can report the clock the chip actually holds under each variant
__device__ unsigned long long g_clk[2];
#ifdef VTGB_DEBUG_HOOKS
__device__ unsigned long long g_stamp[8];   // exp 10 / 11: prologue phase sums (10 ns ticks) + tile count; per-wave landing times
extern "C" void vtgb_debug_read_stamps(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(g_stamp));
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z));
    }
}
__device__ int g_exp_dev = 0;   // experiment selector read by the large kernel (debug-hook builds only)
extern "C" void vtgb_debug_set_exp(int v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_exp_dev), &v, sizeof(v)); }
#endif
#ifdef VTGB_DEBUG_HOOKS
extern "C" void vtgb_debug_read_clk(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk), sizeof(g_clk));
    if (reset) {
        const unsigned long long z[2] = {0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z));
    }
}
#endif
struct ClkScope {
    unsigned long long c0, r0;
    bool on;
    __device__ ClkScope(bool enable) : on(enable) {
        if (on) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    }
    __device__ void stop() {
        if (on && threadIdx.x == 0) {
            atomicAdd(&g_clk[0], __builtin_amdgcn_s_memtime() - c0);
            atomicAdd(&g_clk[1], __builtin_amdgcn_s_memrealtime() - r0);
        }
    }
};

// NXF (activation fragments per wave, default 2 NWN): NWN = 2 with NXF = 8 is a 512 x 128 tile -- waves 4(M) x 2(N) of
// 128 x 64, the SAME wave tile (and LDS-read : MFMA ratio, 12 fragment reads per 32 MFMAs) as the 256 x 256 tile, for
// 128-channel outputs with many rows (RAFT's GRU q convolution, the 126-channel motion-encoder output, the stage 2 / 3
// encoder convolutions); its activation slot is 64 KiB, so the ring is 2 activation + 2 weight slots = 160 KiB.

// Column-statistics fold (EPI_STORE_F32 + col_stats): 64 NWN threads add up the MW waves' partial sums of a column (parked in the
// waves' staging regions) and store them to the tile's slot of the partial-moment buffer (r5: was one atomic per (image, column, moment)).  Run by waves 0 .. NWN-1 WHETHER OR NOT their own
// sub-tile holds valid rows: with a last tile of <= WROWS valid rows those waves are row-inactive, and (rounds 1-2) returned
// before the fold -- the last image's moments then missed that tile for the columns they should have folded (found in round 3:
// fnet 3-10 % off on the last image whenever n_images * H/8 * W/8 mod 256 <= 64).
#define L_STATS_FOLD(PR_)                                                                                           \
    if (tid < 64 * NWN) {                                                                                           \
        const int wn_ = tid >> 6, c = tid & 63, n = n0 + wn_ * 64 + c;                                              \
        if (n < p.N) {                                                                                              \
            const int img_a_ = m0 / p.stats_rows, m_b_ = (img_a_ + 1) * p.stats_rows;                               \
            float t[4] = {0.f, 0.f, 0.f, 0.f};                                                                      \
            for (int wm_ = 0; wm_ < MW; wm_++) {                                                                    \
                if (m0 + wm_ * WROWS >= p.M) break;                                                                 \
                const float* pr = reinterpret_cast<const float*>(smem + (wn_ * MW + wm_) * ((PR_) * 256));          \
                _Pragma("unroll") for (int e = 0; e < 4; e++) t[e] += pr[e * 64 + c];                               \
            }                                                                                                       \
            /* r5: the tile's partial moments are STORED to its own slot [m-tile][column][image a: sum, sum sq | image b: ...]; a second */ \
            /* pass adds the slots in tile order (stats_finish_tiles): no atomics, the same bits on every run */              \
            *reinterpret_cast<float4*>(p.col_stats + ((int64_t)mt * p.N + n) * 4) = make_float4(t[0], t[1], t[2], t[3]);    \
        }                                                                                                           \
    }

template <int EPI, int ABL = 0, bool CONV = false, int NWN = 4, int NXF = 2 * NWN>
__global__ __launch_bounds__(512, 2) void gemm_bf16_large_kernel(const GemmDesc p, const int m_tiles, const int n_tiles, const int G) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass needs only the launch stub (and cannot type the buffer-descriptor builtins)
    ClkScope clk(ABL != 0);
#ifdef VTGB_DEBUG_HOOKS
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int NX = NXF;              // activation fragments per wave: wave tile = (16 NX) x 64
    constexpr int WROWS = 16 * NX;       // rows of the wave tile
    constexpr int MW = 8 / NWN;          // waves along M
    constexpr int T_BN = 64 * NWN;       // tile width
    constexpr int W_OP = T_BN * 128;     // bytes per weight slot
    constexpr int WI = NWN;              // weight DMA instructions per wave and k-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // activation ring: 3 slots; 2 for the 64-wide tile, whose 80 KiB then let TWO workgroups share a CU (these tiles
    // have few k-tiles -- K = 576 for RAFT's 64-channel 3x3 convolutions -- so prologue and epilogue are a third of
    // a tile's life and overlap with the other workgroup's k-loop instead of idling the CU)
    constexpr int T_BM = MW * WROWS;     // tile height: 256, or 512 (NWN = 2, NXF = 8)
    constexpr int A_OP = T_BM * 128;     // bytes per activation slot
    constexpr int AI = T_BM / 64;        // activation DMA instructions per wave and k-tile (8 rows each)
    constexpr int A_SLOTS = (NWN == 1 || T_BM > 256) ? 2 : 3;
    char* const smem_w = smem + A_SLOTS * A_OP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- XCD-aware tile assignment
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int group = idx / (G * n_tiles), r = idx - group * (G * n_tiles);
    const int nt = r / G, ml = group * G + (r - nt * G);
    // plain GEMM: the XCDs take interleaved m-tiles; convolution: each XCD takes a contiguous run of m-tiles,
    // so that the halo rows two neighbouring tiles both gather are fetched into ONE private L2, not two
    if (CONV && ml >= ((m_tiles + 7) >> 3)) return;
    const int mt = CONV ? xcd * ((m_tiles + 7) >> 3) + ml : ml * 8 + xcd;
    if (mt >= m_tiles) return;
    const int m0 = mt * T_BM, n0 = nt * T_BN;
    const int wm = wave % MW, wn = wave / MW;
    // a wave whose 128 x 64 sub-tile lies wholly outside the matrix (N = 1408 is 5.5 tiles wide) still
    // stages its share of the operands and joins every barrier, but issues no LDS reads and no MFMAs:
    // its SIMD partner then has the matrix pipe to itself and the edge tile finishes in half the time
    const bool wave_active = (n0 + wn * 64 < p.N) && (m0 + wm * WROWS < p.M);
    const bool staged_store = ((p.N & 7) == 0) && ((p.ldo & 7) == 0);   // bf16 outputs leave through LDS as whole rows
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(p.W);

    // ---- LDS-DMA assignment: wave w stages rows [32w, 32w+32) of both operands, 4 instructions of 8 rows each; lane l of an
    // instruction fills slot (l & 7) of row r0 + (l >> 3).  Addressing is `buffer_load_dwordx4 ... lds`: a descriptor on the
    // TILE's first row (wave-uniform: SGPRs), a loop-invariant 32-bit byte offset per lane and piece, and the k offset in the
    // scalar offset -- no per-piece 64-bit VALU arithmetic in the k-loop (plain GEMM: none at all).  Convolution: the per-lane
    // offset is (centre pixel + tap shift) * row pitch, and a tap that falls outside the image gets an offset beyond the
    // descriptor's range: the hardware range check then writes ZEROS to LDS for that lane (measured, tools/exp/buf_lds_oob.hip),
    // which is the padding -- no zero page, no pointer select.
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr unsigned OOB = 0x80000000u;   // + the scalar offset (< 2^31) it neither wraps nor re-enters the range
    constexpr int RANGE = 0x7FFFFF00;
    unsigned w_voff[WI], a_voff[AI];
    int a_bits[AI];   // CONV: bit ky = tap row ky lands inside the image, bit 8 + kx = tap column kx does
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W + (int64_t)n0 * p.ldw), 0, RANGE, 0x00020000);
#pragma unroll
    for (int i = 0; i < WI; i++) {
        const int row = wave * (8 * WI) + i * 8 + (lane >> 3), slot = lane & 7, c = slot ^ ((row >> 1) & 7);
        const int wr = (n0 + row) < p.N ? row : p.N - 1 - n0;
        w_voff[i] = (unsigned)(wr * (int)p.ldw + c * 8) * 2u;
    }
    const int cv_hw = CONV ? p.conv_H * p.conv_W : 1, cv_Hi = CONV ? (p.conv_Hi ? p.conv_Hi : p.conv_H) : 1,
              cv_Wi = CONV ? (p.conv_Wi ? p.conv_Wi : p.conv_W) : 1, cv_st = CONV ? (p.conv_stride ? p.conv_stride : 1) : 1;
    const int cv_img0 = CONV ? m0 / cv_hw : 0;
    const int64_t a_row0 = CONV ? (int64_t)cv_img0 * (cv_Hi * cv_Wi) : map_row(p.a_map, m0);   // first source row of the tile
    const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A + a_row0 * p.lda), 0, RANGE, 0x00020000);
    const auto a2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(CONV && p.A2 ? p.A2 : p.A) + a_row0 * (CONV && p.A2 ? p.lda2 : p.lda)), 0, RANGE, 0x00020000);
#pragma unroll
    for (int i = 0; i < AI; i++) {
        const int row = wave * (8 * AI) + i * 8 + (lane >> 3), slot = lane & 7, c = slot ^ ((row >> 1) & 7);
        const int am = (m0 + row) < p.M ? (m0 + row) : p.M - 1;
        if constexpr (CONV) {
            // (two magic-number divisions and closed-form tap ranges: with `/`, `%` and per-tap loops this setup was 1.6 us per tile)
            const int img = (int)((__umulhi((unsigned)am, p.div_hw_mul) + (unsigned)am) >> p.div_hw_sh), rem = am - img * cv_hw;
            const int oy = (int)((__umulhi((unsigned)rem, p.div_w_mul) + (unsigned)rem) >> p.div_w_sh), y = oy * cv_st, x = (rem - oy * p.conv_W) * cv_st;
            a_voff[i] = (unsigned)((img - cv_img0) * (cv_Hi * cv_Wi) + y * cv_Wi + x) | ((unsigned)c << 28);   // pixel (24 bits) | chunk
            // taps k with 0 <= y + k - pad < Hi: k in [max(0, pad - y), min(KH - 1, Hi - 1 - y + pad)]
            const int py = p.conv_KH >> 1, px = p.conv_KW >> 1;
            const int ylo = max(0, py - y), yhi = min(p.conv_KH - 1, cv_Hi - 1 - y + py), xlo = max(0, px - x), xhi = min(p.conv_KW - 1, cv_Wi - 1 - x + px);
            const int yb = yhi >= ylo ? ((2 << yhi) - 1) & ~((1 << ylo) - 1) : 0, xb = xhi >= xlo ? ((2 << xhi) - 1) & ~((1 << xlo) - 1) : 0;
            a_bits[i] = yb | (xb << 8);
        } else {
            a_voff[i] = (unsigned)((int)(map_row(p.a_map, am) - a_row0) * (int)p.lda + c * 8) * 2u;
            a_bits[i] = 0;
        }
    }
    // CONV: running (channel chunk, tap) of the next A k-tile to stage; A tiles are issued in k order, and K runs
    // chunk-major / tap-minor: the KH*KW shifted reads of one 64-channel slab follow each other directly
    // (32 KiB per workgroup, L2 / L1 hits), instead of sweeping the whole tile footprint once per tap (which
    // overflowed the XCD's 4 MiB L2 and sent every tap's re-read to the fabric: 5.4 TB/s of FETCH on the GRU convs)
    int cv_ky = 0, cv_kx = 0, cv_c0 = 0;
#define L_ISSUE_A(slot, k0)                                                                             \
    if constexpr (CONV) {                                                                               \
        const bool first = cv_c0 < p.conv_split;                                                        \
        const unsigned ldb = (unsigned)(first ? p.lda : p.lda2) * 2u;                                   \
        const int cs_ = first ? cv_c0 : cv_c0 - p.conv_split, wr_ = first ? p.conv_wrap : p.conv_wrap2;   /* (bf16x3 pairs: the third block is hi again) */ \
        const int cc2 = ((wr_ > 0 && cs_ >= wr_) ? cs_ - wr_ : cs_) * 2;                                \
        const int dpix = (cv_ky - (p.conv_KH >> 1)) * cv_Wi + (cv_kx - (p.conv_KW >> 1));               \
        const int need = (1 << cv_ky) | (256 << cv_kx);                                                 \
        const auto rs_ = first ? a_rsrc : a2_rsrc;      /* (as in gemm_pp.hip: one select per k-tile, 24-bit multiply-add + select per piece) */ \
        _Pragma("unroll") for (int i = 0; i < AI; i++) {                                                \
            const unsigned pix = (a_voff[i] & 0x00FFFFFFu) + (unsigned)dpix;                            \
            const unsigned vin = __umul24(pix, ldb) + (a_voff[i] >> 28) * 16u;                          \
            const unsigned v = ((a_bits[i] & need) == need) ? vin : OOB;                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lptr_t)(smem + (slot) * A_OP + (wave * (8 * AI) + i * 8) * 128), 16, v, cc2, 0, 0); \
        }                                                                                               \
        if (++cv_kx == p.conv_KW) { cv_kx = 0; if (++cv_ky == p.conv_KH) { cv_ky = 0; cv_c0 += L_BK; } } \
    } else {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < AI; i++)                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)(smem + (slot) * A_OP + (wave * (8 * AI) + i * 8) * 128), 16, a_voff[i], (k0) * 2, 0, 0); \
    }
#define L_ISSUE_W(slot, k0)                                                                             \
    _Pragma("unroll") for (int i = 0; i < WI; i++)                                                      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lptr_t)(smem_w + (slot) * W_OP + (wave * (8 * WI) + i * 8) * 128), 16, w_voff[i], (k0) * 2, 0, 0);
    const int nk = p.K / L_BK;
    const int fr = lane & 15, fg = lane >> 4;
#ifdef VTGB_DEBUG_HOOKS
    const unsigned long long t_setup = __builtin_amdgcn_s_memrealtime();
#endif
    // The first k-tile's DMAs go out FIRST; the accumulator start values (bias: one 16-byte load per weight-fragment column group,
    // not one per accumulator block; + fp32 residual / bf16 start map) are requested behind them, so the two latencies overlap.
    // (In-kernel stamps, tools/exp/prologue_probe.py: with the start values loaded first -- and the adds that consume them
    // waiting for every one -- that phase alone was 4.2 us of a 46 us qkv tile and 20 us of a 60 us projection tile.)
#ifdef VTGB_DEBUG_HOOKS
    const unsigned long long t_init = __builtin_amdgcn_s_memrealtime();
#endif
    L_ISSUE_A(0, 0)
    L_ISSUE_W(0, 0)
#ifdef VTGB_DEBUG_HOOKS
    const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();
#endif
    // EPI_RESID_F32 through the staged fp32 store: the residual tile (256 KB) is NOT preloaded into the accumulators (16 rows x
    // 64 B per load instruction, all of it waited for before the first MFMA: 20 us of a 60 us projection tile) but added in the
    // epilogue from whole 256-byte row segments requested one pass ahead
    const bool resid_late = EPI == EPI_RESID_F32 && ((p.N & 3) == 0) && ((p.ldo & 3) == 0) && ((p.ldr & 3) == 0) && p.act == 0 && p.out_scale == 0.f;
    f32x4 acc[4][NX];
    {
        f32x4 b4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int n = n0 + wn * 64 + i * 16 + fg * 4;
            b4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && n < p.N) {
                if (n + 3 < p.N) b4[i] = *reinterpret_cast<const f32x4*>(p.bias + n);
                else
                    for (int e = 0; e < 4; e++) if (n + e < p.N) b4[i][e] = p.bias[n + e];
            }
        }
        if (p.init_frag) {
            const bf16x4* fsrc = reinterpret_cast<const bf16x4*>(p.init_bf16) + ((int64_t)(mt * n_tiles + nt) * 8 + wave) * (NX * 4 * 64) + lane;
#pragma unroll
            for (int j = 0; j < NX; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const bf16x4 t = fsrc[(j * 4 + i) * 64];
                    acc[i][j] = b4[i] + f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
                }
        } else {
#pragma unroll
        for (int j = 0; j < NX; j++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                acc[i][j] = b4[i] + acc_init4<EPI>(p, m0 + wm * WROWS + j * 16 + fr, n0 + wn * 64 + i * 16 + fg * 4, resid_late);
        }
    }
    // the later k-tiles queue behind EVERY wave's first one (and behind the start values): the first barrier waits for the
    // slowest wave's A(0) / W(0), which must not sit behind another wave's A(2)
    if (nk > 1) {
        L_ISSUE_A(1, L_BK)
        L_ISSUE_W(1, L_BK)
    }
    if (A_SLOTS == 3 && nk > 2) L_ISSUE_A(2, 2 * L_BK)
    if (A_SLOTS == 3 && nk > 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * AI + WI));   // vmcnt(2 AI + WI): A(0), W(0) landed
    else if (nk >= 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (AI + WI));                 // vmcnt(AI + WI)
    else __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0)
#ifdef VTGB_DEBUG_HOOKS
    const unsigned long long t_landed = __builtin_amdgcn_s_memrealtime();
#endif
    __builtin_amdgcn_s_barrier();
#ifdef VTGB_DEBUG_HOOKS
    if ((g_exp_dev & 0xff) == 10 && tid == 0) {   // prologue phases of wave 0 (10 ns ticks): setup | acc init issue | DMA issue | wait | barrier; tile count
        const unsigned long long t_bar = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&g_stamp[0], t_setup - t_entry); atomicAdd(&g_stamp[1], t_init - t_setup); atomicAdd(&g_stamp[2], t_issued - t_init);
        atomicAdd(&g_stamp[3], t_landed - t_issued); atomicAdd(&g_stamp[4], t_bar - t_landed); atomicAdd(&g_stamp[5], 1ull);
        atomicAdd(&g_clk[0], t_bar - t_entry); atomicAdd(&g_clk[1], 1ull);
    }
    if ((g_exp_dev & 0xff) == 11 && lane == 0) atomicAdd(&g_stamp[wave], t_landed - t_entry);   // per wave: entry -> its own first operands landed
#endif

    // fragment byte offsets inside an operand tile for the two 32-deep halves of a k-tile
    int w_off[2][4], x_off[2][NX];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
        for (int i = 0; i < 4; i++) w_off[ks][i] = swz(wn * 64 + i * 16 + fr, ks * 4 + fg);
#pragma unroll
        for (int j = 0; j < NX; j++) x_off[ks][j] = swz(wm * WROWS + j * 16 + fr, ks * 4 + fg);
    }
    bf16x8 wf0[4], xf0[NX], wf1[4], xf1[NX];
#define L_READ(WF, XF, as_, ws_, ks)                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; i++) WF[i] = *reinterpret_cast<const bf16x8*>((ws_) + w_off[ks][i]); \
    _Pragma("unroll") for (int j = 0; j < NX; j++) XF[j] = *reinterpret_cast<const bf16x8*>((as_) + x_off[ks][j]);
#define L_MFMA(WF, XF)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; i++)                                                        \
        _Pragma("unroll") for (int j = 0; j < NX; j++)                                                   \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[i], XF[j], acc[i][j], 0, 0, 0);
    // Rotated k-loop: a wave's LDS fragment reads always run under its own MFMAs.
    //   read half 1 of tile t | MFMAs of half 0 | counted wait + barrier (tile t fully read by everyone;
    //   A(t+1), W(t+1) visible) | re-arm the freed slots with W(t+2), A(t+3) | read half 0 of tile t+1 |
    //   MFMAs of half 1
#ifdef VTGB_DEBUG_HOOKS
    if (g_exp_dev == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if (g_exp_dev == 2 && wave < 4) __builtin_amdgcn_s_setprio(1);
    if (g_exp_dev == 3 && (wave & 1)) __builtin_amdgcn_s_setprio(1);
    if ((g_exp_dev & 0xff) == 4 && bid < 256) {   // phase-stagger the first wave of workgroups: CU slot c of 32 waits c/32 of (g_exp_dev >> 8) us
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), ticks = (unsigned long long)((bid >> 3) & 31) * (g_exp_dev >> 8) * 100 / 32;
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    }
#endif
    int a_slot = 0;   // A(t) lives in slot t % 3, W(t) in slot t % 2
    if (!wave_active) {
        // same DMA issues, waits and barriers as the active waves, nothing else
        for (int kt = 0; kt + 1 < nk; kt++) {
            if (A_SLOTS == 3 && kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0F74);   // vmcnt(4)
            else __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0)
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) { L_ISSUE_W(kt & 1, (kt + 2) * L_BK) }
            if (kt + A_SLOTS < nk) { L_ISSUE_A(a_slot, (kt + A_SLOTS) * L_BK) }
            a_slot = a_slot == A_SLOTS - 1 ? 0 : a_slot + 1;
        }
        if constexpr (EPI == EPI_STORE || EPI == EPI_GELU) {
            if (EPI == EPI_STORE && p.frag_out) return;        // fragment-order stores use no LDS: no barrier to join
            if (staged_store) __builtin_amdgcn_s_barrier();   // the barrier in front of the LDS-staged stores
            if constexpr (EPI == EPI_STORE && NWN == 4 && NXF == 8) {
                if (staged_store && p.tail_w) __builtin_amdgcn_s_barrier();   // ... and the one in front of the fused 1x1 tail
            }
        } else if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32) {
            if (((p.N & 3) == 0) && ((p.ldo & 3) == 0) && p.act == 0 && p.out_scale == 0.f) {
                __builtin_amdgcn_s_barrier();
                if constexpr (EPI == EPI_STORE_F32) {
                    if (p.col_stats) {
                        __builtin_amdgcn_s_barrier();   // the partial-sum exchange
                        constexpr int PR_I = WROWS < 64 ? WROWS : 64;
                        L_STATS_FOLD(PR_I)              // a row-inactive wave still folds its share of the columns
                    }
                }
            }
        } else if constexpr (EPI == EPI_GRU) {
            if (gru_staged(p)) __builtin_amdgcn_s_barrier();
        }
        return;
    }
    L_READ(wf0, xf0, smem, smem_w, 0)
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): see the note at the bottom of the loop
    // One k-loop iteration.  The eight LDS-DMA pieces of an iteration are not issued as one burst behind the barrier (a piece
    // costs its wave 100-185 issue cycles inside a burst, ~60 between MFMAs; both waves of a SIMD burst at the same moment):
    // the W pieces go between the MFMA groups of the second half, and -- with three activation slots -- the A pieces of the
    // tile three ahead are DEFERRED into the first half of the next iteration (they have two k-tiles of slack; program order
    // stays W(t+2), A(t+3), next counted wait, so the vmcnt bookkeeping is unchanged).  DEFER / WCOND / ACOND are literal
    // `true` in the steady-state loop, which keeps each half one basic block for the sched_group_barrier pipeline below.
#define L_SCHED_IL(PIECES)                                                                             \
    if constexpr ((PIECES) > 0 && (4 * NX) % (PIECES) == 0) {                         \
        _Pragma("unroll") for (int g_ = 0; g_ < (PIECES); g_++) {                                      \
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);               /* one LDS-DMA piece */   \
            __builtin_amdgcn_sched_group_barrier(0x008, (4 * NX) / (PIECES), 0);   /* its share of the half's MFMAs */ \
        }                                                                                              \
    }
#define L_ITER(DEFER, WCOND, ACOND, WAIT4)                                                             \
    {                                                                                                  \
        const char* as = smem + a_slot * A_OP;                                                         \
        const char* ws = smem_w + (kt & 1) * W_OP;                                                     \
        const int a_nxt = a_slot == A_SLOTS - 1 ? 0 : a_slot + 1;                                      \
        const int a_prv = a_slot == 0 ? A_SLOTS - 1 : a_slot - 1;                                      \
        if constexpr (!(ABL & 1)) { if (DEFER) { L_ISSUE_A(a_prv, (kt - 1 + A_SLOTS) * L_BK) } }       \
        L_READ(wf1, xf1, as, ws, 1)                                                                    \
        L_MFMA(wf0, xf0)                                                                               \
        if constexpr (!(ABL & 1) && A_SLOTS == 3) { L_SCHED_IL(AI) }                                   \
        if constexpr (!(ABL & 8)) {                                                                    \
            if (WAIT4) __builtin_amdgcn_s_waitcnt(0x0074);   /* vmcnt(4) lgkmcnt(0): all but A(t+2) landed */ \
            else __builtin_amdgcn_s_waitcnt(0x0070);         /* vmcnt(0) lgkmcnt(0) */                 \
            __builtin_amdgcn_s_barrier();                                                              \
        }                                                                                              \
        if constexpr (!(ABL & 1)) {                                                                    \
            if (WCOND) { L_ISSUE_W(kt & 1, (kt + 2) * L_BK) }                                          \
            if (ACOND) { L_ISSUE_A(a_slot, (kt + A_SLOTS) * L_BK) }                                    \
        }                                                                                              \
        L_READ(wf0, xf0, smem + a_nxt * A_OP, smem_w + ((kt + 1) & 1) * W_OP, 0)                       \
        L_MFMA(wf1, xf1)                                                                               \
        if constexpr (!(ABL & 1)) { L_SCHED_IL(A_SLOTS == 3 ? WI : WI + AI) }                          \
        /* the half-0 fragments of the next tile were requested 32 MFMAs ago: this wait is free, and it lets hipcc's      \
           waitcnt pass see (at the loop-header join) that set 0 is complete */                        \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                            \
        a_slot = a_nxt;                                                                                \
    }
    {
        constexpr bool A3 = A_SLOTS == 3, A2 = A_SLOTS != 3;
        int kt = 0;
        if (nk > 1) { L_ITER(false, kt + 2 < nk, A2 && kt + 2 < nk, A3 && kt + 2 < nk) kt = 1; }
        for (; kt + 2 < nk; kt++) L_ITER(A3, true, A2, A3)                      // steady state: no conditions
        for (; kt + 1 < nk; kt++) L_ITER(A3 && kt + 2 < nk, kt + 2 < nk, A2 && kt + 2 < nk, A3 && kt + 2 < nk)
    }
#undef L_ITER
#undef L_SCHED_IL
    {   // last k-tile
        const char* as = smem + a_slot * A_OP;
        const char* ws = smem_w + ((nk - 1) & 1) * W_OP;
        L_READ(wf1, xf1, as, ws, 1)
        L_MFMA(wf0, xf0)
        L_MFMA(wf1, xf1)
    }
    clk.stop();
#undef L_READ
#undef L_MFMA
#undef L_ISSUE_A
#undef L_ISSUE_W
    // D layout: column (lane & 15) <- X row (m), rows (lane >> 4) * 4 + reg <- W row (n)
    if constexpr ((ABL & 16) != 0) {   // timing-only: no epilogue (keep the accumulators live)
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < NX; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 123.456f) reinterpret_cast<float*>(p.out)[0] = t;
        return;
    }
    if constexpr (EPI == EPI_STORE) {
        if (p.frag_out) {   // fragment order: straight from the accumulators, 512 contiguous bytes per wave instruction, no LDS
            bf16x4* fdst = reinterpret_cast<bf16x4*>(p.out) + ((int64_t)(mt * n_tiles + nt) * 8 + wave) * (NX * 4 * 64) + lane;
#pragma unroll
            for (int j = 0; j < NX; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    f32x4 v = acc[i][j];
                    if (p.act) apply_act4(v, p.act);
                    fdst[(j * 4 + i) * 64] = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                }
            return;
        }
    }
    if constexpr (EPI == EPI_STORE || EPI == EPI_GELU) {
        if (staged_store) {
            // bf16 outputs: each wave transposes its 128 x 64 tile through a private 16 KiB LDS region
            // (16-byte chunks XOR-swizzled by row & 7) and stores 16 bytes per lane, 8 whole 128-byte
            // rows per instruction: half the store instructions of the 8-byte fragment-shaped stores
            // (the tail is store-issue bound), and full lines.
            __builtin_amdgcn_s_waitcnt(0xC07F);   // my fragment reads are done
            __builtin_amdgcn_s_barrier();         // ... and everyone else's: the ring can be overwritten
            char* const cst = smem + wave * (WROWS * 128);
#pragma unroll
            for (int j = 0; j < NX; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    f32x4 v = acc[i][j];
                    if constexpr (EPI == EPI_GELU) {
gelu_erf_fast4(v);
                    } else if (p.act) apply_act4(v, p.act);
                    const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                    const int row = j * 16 + fr, c16 = (i * 2 + (fg >> 1)) ^ (row & 7);
                    *reinterpret_cast<bf16x4*>(cst + row * 128 + c16 * 16 + (fg & 1) * 8) = pk;
                }
            if constexpr (EPI == EPI_STORE && NWN == 4 && NXF == 8) {
                if (p.tail_w) {
                    // Fused 1x1 tail: T[256 px][32] = tile[256 px][256 ch] . tail_w[32][256]^T.  The activated bf16 tile is in LDS
                    // (wave (wm, wn) owns rows wm * 128 .., columns wn * 64 .. as 128-byte rows, 16-byte chunks XOR row & 7); the
                    // four waves of a row half split its 8 pixel blocks, two each: 2 blocks x 2 output blocks x 8 k-steps = 32 MFMAs.
                    // tail_w fragments come straight from global (16 KiB, L2 resident) into registers the accumulators no longer need.
                    const bf16_t* const tw = reinterpret_cast<const bf16_t*>(p.tail_w);
                    bf16x8 twf[2][8];
#pragma unroll
                    for (int nb = 0; nb < 2; nb++)
#pragma unroll
                        for (int ks = 0; ks < 8; ks++) twf[nb][ks] = *reinterpret_cast<const bf16x8*>(tw + (nb * 16 + fr) * 256 + ks * 32 + fg * 8);
                    __builtin_amdgcn_s_waitcnt(0xC07F);   // my staging writes
                    __builtin_amdgcn_s_barrier();         // everyone's: the tile is complete
#pragma unroll
                    for (int pb = 0; pb < 2; pb++) {
                        const int blk = wn * 2 + pb, row = blk * 16 + fr;          // pixel block inside this wave's row half
                        f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
#pragma unroll
                        for (int ks = 0; ks < 8; ks++) {
                            const int k = ks * 32 + fg * 8, reg = k >> 6, c16 = (k & 63) >> 3;   // column k lives in wave (wm, k / 64)'s region
                            const bf16x8 xfr = *reinterpret_cast<const bf16x8*>(smem + (wm + MW * reg) * (WROWS * 128) + row * 128 + ((c16 ^ (row & 7)) << 4));
                            t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twf[0][ks], xfr, t0, 0, 0, 0);
                            t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twf[1][ks], xfr, t1, 0, 0, 0);
                        }
                        const int m = m0 + wm * WROWS + row;                       // D: column (lane & 15) <- pixel, rows 4 fg + r <- output
                        if (m < p.M) {
                            float* o = p.tail_out + (int64_t)m * p.ldtail + fg * 4;
                            *reinterpret_cast<f32x4*>(o) = t0;
                            *reinterpret_cast<f32x4*>(o + 16) = t1;
                        }
                    }
                    return;
                }
            }
            bf16_t* const outp = reinterpret_cast<bf16_t*>(p.out);
#ifdef VTGB_DEBUG_HOOKS
            const bool skip_store = (g_exp_dev & 0xff) == 5;   // timing only: the whole epilogue but the global stores
#else
            constexpr bool skip_store = false;
#endif
            // operands of the fused elementwise tails (r * h gate, ResidualBlock skip): all rows' loads in flight before the
            // first store, not one HBM round trip per row
            const bool gated = EPI == EPI_STORE && p.gate_from > 0, resd = EPI == EPI_STORE && !gated && p.resid_bf16 != nullptr;
            const int ch0 = lane & 7, nn = n0 + wn * 64 + ch0 * 8;
            bf16x8 opnd[WROWS / 8];
            if (gated || resd) {
#pragma unroll
                for (int rr = 0; rr < WROWS / 8; rr++) {
                    const int m = m0 + wm * WROWS + rr * 8 + (lane >> 3);
                    opnd[rr] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    if (m < p.M && nn < p.N) {
                        if (gated) {
                            if (nn >= p.gate_from)
                                opnd[rr] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.aux) + (int64_t)m * p.ldaux + (nn - p.gate_from));
                        } else {
                            opnd[rr] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.resid_bf16) + (int64_t)m * p.ldrb + nn);
                        }
                    }
                }
            }
#ifdef VTGB_DEBUG_HOOKS
            unsigned long long ts0 = 0;
            if ((g_exp_dev & 0xff) == 6) ts0 = __builtin_amdgcn_s_memrealtime();
#endif
            // Every LDS read of the wave's tile is issued before its first global store, and the common case (identity row
            // map, no fused tail) runs a branch-free store loop: stores then leave back to back.  (Interleaved
            // read -> address -> store through the general path cost 8.2 us per wave for 16 stores -- 15 GB/s per CU against
            // the 95 GB/s the same store shape reaches in isolation, tools/exp/store_rate.hip: a 64-bit division per row for
            // the row map, ~10 uniform branches per row, and vmcnt(0) waits the compiler places in front of LDS reads that
            // follow a store.)
            uint4 vv[WROWS / 8];
#pragma unroll
            for (int rr = 0; rr < WROWS / 8; rr++) {
                const int row = rr * 8 + (lane >> 3);
                vv[rr] = *reinterpret_cast<const uint4*>(cst + row * 128 + ((ch0 ^ (row & 7)) << 4));
            }
            // fused tails, in place, for every row before the first store (a conditional load-use + store per row makes hipcc's
            // waitcnt pass fall back to vmcnt(0) in front of each row: every row then waits for the previous row's store)
            const bool to_out2 = gated && nn >= p.gate_from;   // per lane: this lane's 8 columns are r (-> r * h, out2) or z (-> out)
            if (gated || resd) {
                __builtin_amdgcn_s_waitcnt(0x0F70);   // the operand loads, once and outside the per-row conditionals (see the GRU path)
#pragma unroll
                for (int rr = 0; rr < WROWS / 8; rr++) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, vv[rr]);
                    const bf16x8 g = opnd[rr];
                    bf16x8 o = a;
                    if (to_out2) {
#pragma unroll
                        for (int e = 0; e < 8; e++) o[e] = (bf16_t)((float)a[e] * (float)g[e]);
                    } else if (resd) {
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const float t = (float)a[e] + (float)g[e];
                            o[e] = (bf16_t)(p.post_relu ? fmaxf(t, 0.f) : t);
                        }
                    }
                    vv[rr] = __builtin_bit_cast(uint4, o);
                }
            }
            const int mrow = m0 + wm * WROWS + (lane >> 3);
            const bool nok = nn < p.N && !skip_store;
            if (to_out2 || p.o_map.seg_rows == 0) {
                bf16_t* const o = to_out2 ? reinterpret_cast<bf16_t*>(p.out2) + (int64_t)mrow * p.ldo2 + (nn - p.gate_from) : outp + (int64_t)mrow * p.ldo + nn;
                const int64_t step = (int64_t)8 * (to_out2 ? p.ldo2 : p.ldo);
#pragma unroll
                for (int rr = 0; rr < WROWS / 8; rr++)
                    if (nok && mrow + rr * 8 < p.M) *reinterpret_cast<uint4*>(o + rr * step) = vv[rr];
            } else {
#pragma unroll
                for (int rr = 0; rr < WROWS / 8; rr++)
                    if (nok && mrow + rr * 8 < p.M) *reinterpret_cast<uint4*>(outp + map_row(p.o_map, mrow + rr * 8) * p.ldo + nn) = vv[rr];
            }
#ifdef VTGB_DEBUG_HOOKS
            if ((g_exp_dev & 0xff) == 6) {   // per wave: 10 ns ticks from the first store's issue to the last one's, and to their completion
                const unsigned long long ts1 = __builtin_amdgcn_s_memrealtime();
                __builtin_amdgcn_s_waitcnt(0x0F70);
                const unsigned long long ts2 = __builtin_amdgcn_s_memrealtime();
                if (lane == 0) { atomicAdd(&g_clk[0], ts1 - ts0); atomicAdd(&g_clk[1], ts2 - ts0); }
            }
#endif
            return;
        }
    }
    if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32) {
        if (((p.N & 3) == 0) && ((p.ldo & 3) == 0) && p.act == 0 && p.out_scale == 0.f) {
            // fp32 outputs: same idea, 64 rows per pass (the wave's region is 16 KiB): 16-byte chunks of a
            // 256-byte row XOR-swizzled by row & 15; each store instruction then covers 4 whole 256-byte row
            // segments.  (Fragment-shaped fp32 stores -- 16 rows x 64 B per instruction, half lines --
            // measured 1.6 TB/s against 6.5 TB/s for whole-line stores.)
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            constexpr int PR = EPI == EPI_RESID_F32 ? 32 : (WROWS < 64 ? WROWS : 64);   // rows per pass (the residual rows ride along: half the registers)
            char* const cst = smem + wave * (PR * 256);
            float* const outp = reinterpret_cast<float*>(p.out);
            const int rl = lane >> 4, cl = lane & 15;
            // column statistics (EPI_STORE_F32 + col_stats): the 256 rows of a tile touch at most two images
            // (stats_rows >= 256): set a = the image of row m0, set b = the next one
            const bool do_stats = (EPI == EPI_STORE_F32) && p.col_stats != nullptr;
            const int img_a = do_stats ? m0 / p.stats_rows : 0;
            const int m_b = do_stats ? (img_a + 1) * p.stats_rows : 0x7fffffff;   // first row of image b
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, qa = sa, sb = sa, qb = sa;
            const int n = n0 + wn * 64 + cl * 4;
            // late residual: the rows of pass `h` in the store layout (4 rows x 256 B per instruction)
            f32x4 rq[PR / 4];
#define L_RESID_LOAD(dst, h)                                                                                        \
    _Pragma("unroll") for (int rr = 0; rr < PR / 4; rr++) {                                                          \
        const int m_ = m0 + wm * WROWS + (h) * PR + rr * 4 + rl;                                                    \
        dst[rr] = f32x4{0.f, 0.f, 0.f, 0.f};                                                                        \
        if (m_ < p.M && n < p.N) dst[rr] = *reinterpret_cast<const f32x4*>(p.resid + (p.r_map.seg_rows == 0 ? (int64_t)m_ : map_row(p.r_map, m_)) * p.ldr + n); \
    }
#pragma unroll
            for (int half = 0; half < WROWS / PR; half++) {
#pragma unroll
                for (int jj = 0; jj < PR / 16; jj++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int row = jj * 16 + fr, chunk = i * 4 + fg;
                        *reinterpret_cast<f32x4*>(cst + row * 256 + ((chunk ^ (row & 15)) << 4)) = acc[i][half * (PR / 16) + jj];
                    }
                if constexpr (EPI == EPI_RESID_F32) {
                    if (half == 0) { L_RESID_LOAD(rq, 0) }   // (behind the first LDS write: that half of the accumulators is dead, the registers are there)
                    __builtin_amdgcn_s_waitcnt(0x0F70);      // this pass's residual rows (one wait, outside the per-row conditionals)
                }
                // all of the pass's LDS reads before its first store (see the bf16 path)
                f32x4 vv[PR / 4];
#pragma unroll
                for (int rr = 0; rr < PR / 4; rr++) {
                    const int row = rr * 4 + rl;
                    vv[rr] = *reinterpret_cast<const f32x4*>(cst + row * 256 + ((cl ^ (row & 15)) << 4));
                    if constexpr (EPI == EPI_RESID_F32) vv[rr] += rq[rr];
                }
                if constexpr (EPI == EPI_RESID_F32) {
                    if (half + 1 < WROWS / PR) { L_RESID_LOAD(rq, half + 1) }   // the next pass's rows, requested before this pass's stores
                }
                const bool ident = p.o_map.seg_rows == 0;
#pragma unroll
                for (int rr = 0; rr < PR / 4; rr++) {
                    const int row = rr * 4 + rl;
                    const f32x4 v = vv[rr];
                    const int m = m0 + wm * WROWS + half * PR + row;
                    if (m < p.M && n < p.N) {
                        *reinterpret_cast<f32x4*>(outp + (ident ? (int64_t)m : map_row(p.o_map, m)) * p.ldo + n) = v;
                        if constexpr (EPI == EPI_STORE_F32) {
                            if (do_stats) {
                                if (m < m_b) { sa += v; qa += v * v; }
                                else { sb += v; qb += v * v; }
                            }
                        }
                    }
                }
            }
#undef L_RESID_LOAD
            if constexpr (EPI == EPI_STORE_F32) {
                if (do_stats) {
                    // lanes with equal cl hold the same four columns: fold the four row groups, park the
                    // wave's partials in its own staging region, then 64 NWN threads fold the MW waves of a
                    // column and issue one atomic per (image, column, moment)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        sa[e] += __shfl_xor(sa[e], 16); sa[e] += __shfl_xor(sa[e], 32);
                        qa[e] += __shfl_xor(qa[e], 16); qa[e] += __shfl_xor(qa[e], 32);
                        sb[e] += __shfl_xor(sb[e], 16); sb[e] += __shfl_xor(sb[e], 32);
                        qb[e] += __shfl_xor(qb[e], 16); qb[e] += __shfl_xor(qb[e], 32);
                    }
                    if (rl == 0) {
                        f32x4* const pr = reinterpret_cast<f32x4*>(cst);
                        pr[cl] = sa; pr[16 + cl] = qa; pr[32 + cl] = sb; pr[48 + cl] = qb;
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_s_barrier();
                    L_STATS_FOLD(PR)
                }
            }
            return;
        }
    }
    if constexpr (EPI == EPI_GRU) {
        if (gru_staged(p)) {
            // h' = (1 - z) h + z tanh(acc): the accumulators go through LDS as in the fp32 path so that h, z
            // and both outputs are touched as whole row segments (16 B of fp32 / 8 B of bf16 per lane).
            // ALL of a pass's h / z loads are issued before the pass's LDS transpose and its first store (16 rows x 24 B per
            // lane in flight): interleaved load -> tanh -> store made every row wait out a full HBM round trip plus the
            // previous row's store acknowledgement (vmcnt counts stores), a serial chain as long as half the k-loop.
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            constexpr int PR = WROWS < 64 ? WROWS : 64;
            char* const cst = smem + wave * (PR * 256);
            const int rl = lane >> 4, cl = lane & 15;
            const int n = n0 + wn * 64 + cl * 4;
            const bool ncol = n < p.N;
#pragma unroll
            for (int half = 0; half < WROWS / PR; half++) {
                f32x4 hreg[PR / 4];
                bf16x4 zreg[PR / 4];
#pragma unroll
                for (int rr = 0; rr < PR / 4; rr++) {
                    const int m = m0 + wm * WROWS + half * PR + rr * 4 + rl;
                    hreg[rr] = f32x4{0.f, 0.f, 0.f, 0.f};
                    zreg[rr] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                    if (m < p.M && ncol) {
                        hreg[rr] = *reinterpret_cast<const f32x4*>(p.resid + (p.r_map.seg_rows == 0 ? (int64_t)m : map_row(p.r_map, m)) * p.ldr + n);
                        zreg[rr] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.aux) + (int64_t)m * p.ldaux + n);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < PR / 16; jj++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int row = jj * 16 + fr, chunk = i * 4 + fg;
                        *reinterpret_cast<f32x4*>(cst + row * 256 + ((chunk ^ (row & 15)) << 4)) = acc[i][half * (PR / 16) + jj];
                    }
                // ONE explicit vmcnt(0) here, outside every conditional: the h / z loads above sit in per-row conditionals, so
                // hipcc's waitcnt pass cannot count them and, left alone, puts `s_waitcnt vmcnt(0)` in front of every row's
                // first use -- which also waits for the PREVIOUS row's stores to be acknowledged (vmcnt counts stores): a
                // memory round trip per row, 16 rows per pass.  After this wait no later row needs one.
                __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
                for (int rr = 0; rr < PR / 4; rr++) {
                    const int row = rr * 4 + rl;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(cst + row * 256 + ((cl ^ (row & 15)) << 4));
                    f32x4 hn;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float z = (float)zreg[rr][e];
                        hn[e] = (1.0f - z) * hreg[rr][e] + z * tanh_fast(v[e]);
                    }
                    hreg[rr] = hn;
                    zreg[rr] = bf16x4{(bf16_t)hn[0], (bf16_t)hn[1], (bf16_t)hn[2], (bf16_t)hn[3]};
                }
                const bool o_ident = p.o_map.seg_rows == 0;
#pragma unroll
                for (int rr = 0; rr < PR / 4; rr++) {
                    const int m = m0 + wm * WROWS + half * PR + rr * 4 + rl;
                    if (m < p.M && ncol) {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (o_ident ? (int64_t)m : map_row(p.o_map, m)) * p.ldo + n) = hreg[rr];
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.out2) + (int64_t)m * p.ldo2 + n) = zreg[rr];
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < NX; j++) {
        const int m = m0 + wm * WROWS + j * 16 + fr;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int n = n0 + wn * 64 + i * 16 + fg * 4;
            store4<EPI>(p, m, n, acc[i][j]);
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------
// fp32 kernel (exactness mode; also the on-GPU cross-check for the MFMA kernel)
// ---------------------------------------------------------------------------------------
// r5: the products run on v_mfma_f32_32x32x2_f32, bit for bit a k-ordered fmaf chain (cdna_hip_programming.md, "FP32-input MFMA"): the same
// numbers as the scalar-FMA loop of rounds 1-4 at twice its rate.  128 x 64 tile, 16-deep k-slabs through LDS (k-major), four waves of
// 64 x 32; the weight slab is the A operand, so a lane holds four runs of four consecutive columns of one row; the next slab's loads fly
// under the current slab's products.
typedef float gf_f32x16 __attribute__((ext_vector_type(16)));
constexpr int GF_BM = 128, GF_BN = 64;
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmDesc p) {
    __shared__ float As[16][GF_BM + 4];
    __shared__ float Ws[16][GF_BN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.y * GF_BM, n0 = blockIdx.x * GF_BN;
    const float* __restrict__ A = reinterpret_cast<const float*>(p.A);
    const float* __restrict__ W = reinterpret_cast<const float*>(p.W);
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    int wr = n0 + lrow; wr = wr < p.N ? wr : p.N - 1;
    const float* wp = W + (int64_t)wr * p.ldw;
    const float* ap[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        int am = m0 + lrow + h * 64; am = am < p.M ? am : p.M - 1;
        ap[h] = A + map_row(p.a_map, am) * p.lda;
    }
    // 16-byte loads when every row start and the contraction length allow them (wave-uniform)
    const bool vec = ((p.K | (int)p.lda | (int)p.ldw) & 3) == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) == 0;
    float av[2][4], wv[4];
    auto fetch = [&](int k0) {
        const int k = k0 + lk;
        if (vec) {
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, w4 = a0;
            if (k < p.K) { a0 = *reinterpret_cast<const float4*>(ap[0] + k); a1 = *reinterpret_cast<const float4*>(ap[1] + k); w4 = *reinterpret_cast<const float4*>(wp + k); }
            av[0][0] = a0.x; av[0][1] = a0.y; av[0][2] = a0.z; av[0][3] = a0.w;
            av[1][0] = a1.x; av[1][1] = a1.y; av[1][2] = a1.z; av[1][3] = a1.w;
            wv[0] = w4.x; wv[1] = w4.y; wv[2] = w4.z; wv[3] = w4.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                av[0][i] = k + i < p.K ? ap[0][k + i] : 0.f;
                av[1][i] = k + i < p.K ? ap[1][k + i] : 0.f;
                wv[i] = k + i < p.K ? wp[k + i] : 0.f;
            }
        }
    };
    gf_f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
    fetch(0);
    const int l31 = lane & 31, kh = lane >> 5;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            As[lk + i][lrow] = av[0][i];
            As[lk + i][lrow + 64] = av[1][i];
            Ws[lk + i][lrow] = wv[i];
        }
        __syncthreads();
        if (k0 + 16 < p.K) fetch(k0 + 16);
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            const float wf = Ws[2 * kk + kh][wn * 32 + l31];
#pragma unroll
            for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf, As[2 * kk + kh][wm * 64 + j * 32 + l31], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[i][j]: column j = lane & 31 -> row m, rows i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> column n: four consecutive n per reg >> 2
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int m = m0 + wm * 64 + j * 32 + l31;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++)
            epilogue4<EPI, float>(p, m, n0 + wn * 32 + 8 * g4 + 4 * kh, acc[j][4 * g4], acc[j][4 * g4 + 1], acc[j][4 * g4 + 2], acc[j][4 * g4 + 3]);
    }
}

// round 3: the persistent ping-pong form of the large kernel (gemm_pp.hip); VTGB_GEMM_OLD=1 keeps the one-tile-per-workgroup kernel
template <int EPI, bool CONV, int NWN, int WF = 4>
int launch_large_pp(const GemmDesc& d, hipStream_t s);
bool pp_supported(const GemmDesc& d);
// (A/B switches -- VTGB_GEMM_OLD=1, VTGB_PP_MASK=<bits> -- exist only in builds with -DVTGB_DEBUG_HOOKS: the production library reads no
// environment variable)
static bool use_old_large() {
#ifdef VTGB_DEBUG_HOOKS
    static int v = -1;
    if (v < 0) { const char* e = getenv("VTGB_GEMM_OLD"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
#else
    return false;
#endif
}
// bisecting aid: VTGB_PP_MASK = bit set of launch classes that may use the persistent kernel (default: all)
//   1 plain GEMM   2 conv 256-wide bf16 store   4 conv + fused 1x1 tail   8 conv 128-wide bf16 store   16 GRU   32 conv fp32 store
static bool pp_class_enabled(int epi, bool conv, int nwn, bool tail) {
#ifdef VTGB_DEBUG_HOOKS
    static int mask = -1;
    if (mask < 0) { const char* e = getenv("VTGB_PP_MASK"); mask = e ? atoi(e) : 63; }
#else
    const int mask = 63;
#endif
    int c;
    if (!conv) c = 1;
    else if (epi == EPI_GRU) c = 16;
    else if (epi == EPI_STORE_F32) c = 32;
    else if (tail) c = 4;
    else c = nwn == 4 ? 2 : 8;
    return (mask & c) != 0;
}

// tile-count threshold above which the 256x256 LDS-DMA kernel is used (tunable for experiments)
// Experiment knobs.  The setters (and the timing-only ablation instantiations, whose results are WRONG by construction) exist
// only in builds with -DVTGB_DEBUG_HOOKS (VTGB_DEBUG_HOOKS=1 python -m videotgb_amd.build; tools/gemm_ablate.py): the
// production library exports none of them.
static int g_large_min_tiles = 200;
static int g_large_variant = 0;   // > 0 forces the m-tiles per XCD super-tile (G); 0 = heuristic
static int g_ablate = 0;          // 1 no DMA, 2 one LDS stage, 4 no LDS reads, 8 no barrier (timing only)
#ifdef VTGB_DEBUG_HOOKS
extern "C" void vtgb_debug_set_gemm_large_min_tiles(int v) { g_large_min_tiles = v; }
extern "C" void vtgb_debug_set_gemm_large_variant(int v) { g_large_variant = v; }
extern "C" void vtgb_debug_set_gemm_ablate(int v) { g_ablate = v; }
#endif

// The large kernel addresses its operands as (tile base descriptor) + 32-bit byte offset: true when every offset inside one
// tile fits (row maps must be monotone: a 256-row tile then spans 256 rows plus the gaps of the segments it crosses).
static bool large_kernel_addressable(const GemmDesc& d) {
    int64_t span = 512;                                        // rows of the widest tile
    if (d.a_map.seg_rows != 0) {
        if (d.a_map.seg_stride < d.a_map.seg_rows) return false;
        span += (512 / d.a_map.seg_rows + 2) * (d.a_map.seg_stride - d.a_map.seg_rows);
    }
    if (d.conv_KH > 0) {
        if (d.conv_KH > 7 || d.conv_KW > 7) return false;      // tap-validity bits: 7 + 7 per staged row
        const int64_t hw = (int64_t)d.conv_H * d.conv_W, hiwi = (int64_t)(d.conv_Hi ? d.conv_Hi : d.conv_H) * (d.conv_Wi ? d.conv_Wi : d.conv_W);
        span = (512 / hw + 2) * hiwi;                          // input pixels of the images one tile touches
        if (span >= (1 << 24)) return false;                   // pixel index field of the per-lane offset word
    }
    const int64_t ld = d.lda > d.lda2 ? d.lda : d.lda2;
    return span * ld * 2 < 0x7FFFFF00ll && (int64_t)256 * d.ldw * 2 < 0x7FFFFF00ll;
}

template <int EPI>
static int launch_epi(const GemmDesc& d, hipStream_t s) {
    if (d.dtype == VTGB_BF16) {
        const int m_tiles = (d.M + L_BM - 1) / L_BM, n_tiles = (d.N + L_BN - 1) / L_BN;
        if ((d.K % L_BK) == 0 && (int64_t)m_tiles * n_tiles >= g_large_min_tiles && large_kernel_addressable(d)) {
            // m-tiles per XCD super-tile: measured best 2 for narrow outputs (<= 8 n-tiles), 8 for wide ones
            const int G = g_large_variant > 0 ? g_large_variant : (n_tiles <= 8 ? 2 : 8);
            const int mx = (m_tiles + 7) / 8, groups = (mx + G - 1) / G;
            const dim3 grid(8 * groups * G * n_tiles);
            if (!use_old_large() && !g_ablate && pp_supported(d) && pp_class_enabled(EPI, false, 4, false)) return launch_large_pp<EPI, false, 4>(d, s);
            VTGB_REQUIRE(!d.ln_xb && !d.ln_stats, VTGB_EUNSUPPORTED, "gemm: the folded LayerNorm needs the persistent or the 128 x 128 kernel (row maps / alignment)");
            ProfScope prof(VTGB_PROF_GEMM, 2.0 * d.M * d.N * d.K, s);
#ifdef VTGB_DEBUG_HOOKS
            if (g_ablate && (EPI == EPI_STORE || EPI == EPI_RESID_F32)) {
#define ABL_CASE(v)                                                                                              \
    case v:                                                                                                       \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_large_kernel<EPI, v>),                 \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, L_LDS);                            \
        hipLaunchKernelGGL((gemm_bf16_large_kernel<EPI, v>), grid, dim3(512), L_LDS, s, d, m_tiles, n_tiles, G);    \
        break;
                switch (g_ablate) { ABL_CASE(1) ABL_CASE(8) ABL_CASE(16) ABL_CASE(17) ABL_CASE(25) ABL_CASE(32) default: break; }
#undef ABL_CASE
            } else
#endif
            {
                static DeviceOnce attr0;
                VTGB_FUNC_LDS_ONCE(attr0, (gemm_bf16_large_kernel<EPI, 0>), L_LDS);
                hipLaunchKernelGGL((gemm_bf16_large_kernel<EPI, 0>), grid, dim3(512), L_LDS, s, d, m_tiles, n_tiles, G);
            }
            VTGB_HIP(hipGetLastError());
            return VTGB_OK;
        }
        dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM);
        const size_t lds = 4 * TILE_BYTES;
        static DeviceOnce attr_set;
        VTGB_FUNC_LDS_ONCE(attr_set, gemm_bf16_kernel<EPI>, lds);
        ProfScope prof(VTGB_PROF_GEMM, 2.0 * d.M * d.N * d.K, s);
        hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, grid, dim3(256), lds, s, d);
    } else {
        dim3 grid((d.N + GF_BN - 1) / GF_BN, (d.M + GF_BM - 1) / GF_BM);
        hipLaunchKernelGGL(gemm_f32_kernel<EPI>, grid, dim3(256), 0, s, d);
    }
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// Implicit-GEMM convolution (and plain GEMMs that need the activation / GRU epilogues) on the large
// kernel.  d.conv_KH == 0: plain GEMM through the same kernel (RAFT's 1x1 convolutions).
template <int EPI, bool CONV, int NWN, int NXF = 2 * NWN>
static int launch_large_nwn(const GemmDesc& d, hipStream_t s) {
    constexpr int T_BM = (8 / NWN) * 16 * NXF, T_BN = 64 * NWN;
    constexpr int LDS = ((NWN == 1 || T_BM > 256) ? 2 : 3) * T_BM * 128 + L_W_SLOTS * T_BN * 128;
    const int m_tiles = (d.M + T_BM - 1) / T_BM, n_tiles = (d.N + T_BN - 1) / T_BN;
    const int G = n_tiles <= 8 ? 2 : 8;
    const int mx = (m_tiles + 7) / 8, groups = (mx + G - 1) / G;
    if constexpr (NXF == 2 * NWN) {
        if constexpr (!(EPI == EPI_GRU && NWN == 4) && NWN != 1) {      // (64-wide tiles: two workgroups per CU already overlap prologues)
            if (!use_old_large() && pp_supported(d) && pp_class_enabled(EPI, CONV, NWN, d.tail_w != nullptr)) return launch_large_pp<EPI, CONV, NWN>(d, s);
        }
    }
    static DeviceOnce attr;
    VTGB_FUNC_LDS_ONCE(attr, (gemm_bf16_large_kernel<EPI, 0, CONV, NWN, NXF>), LDS);
    const double exec_flops = 2.0 * d.M * d.N * d.K;
    ProfScope prof(CONV ? VTGB_PROF_CONV : VTGB_PROF_GEMM, d.algo_flops > 0 ? d.algo_flops : d.algo_flops < 0 ? 0.0 : exec_flops, s, exec_flops);
    hipLaunchKernelGGL((gemm_bf16_large_kernel<EPI, 0, CONV, NWN, NXF>), dim3(8 * groups * G * n_tiles), dim3(512), LDS, s, d, m_tiles, n_tiles, G);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

static int g_conv_nwn = 0;   // > 0 forces the tile (experiments: 1, 2, 4 = waves along N; 5 = the 512 x 128 tile for 128-wide outputs); 0 = by shape
#ifdef VTGB_DEBUG_HOOKS
extern "C" void vtgb_debug_set_conv_nwn(int v) { g_conv_nwn = v; }
#endif

template <int EPI, bool CONV>
static int launch_large_forced(const GemmDesc& d, hipStream_t s) {
    const int nwn = (g_conv_nwn > 0 && g_conv_nwn != 5) ? g_conv_nwn : (d.N <= 64 ? 1 : d.N <= 128 ? 2 : 4);
    const int nwn_shape = d.N <= 64 ? 1 : d.N <= 128 ? 2 : 4;
    // the fused 1x1 tail exists only in the 256-wide tile, and fragment-order maps are only meaningful between launches of ONE tile
    // shape (the one the shape rule picks): a forced tile (debug hook) would otherwise succeed and leave the output unwritten / scrambled
    VTGB_REQUIRE(!d.tail_w || nwn == 4, VTGB_EUNSUPPORTED, "conv gemm: the fused 1x1 tail needs the 256-wide tile (forced tile %d)", nwn);
    VTGB_REQUIRE(!(d.frag_out || d.init_frag) || nwn == nwn_shape, VTGB_EUNSUPPORTED,
                 "conv gemm: fragment-order start maps need the producer's and the consumer's tile shape to agree (forced tile %d, shape rule %d)", nwn, nwn_shape);
    if constexpr (EPI == EPI_STORE && CONV) {
        // 128 < N <= 192 (RAFT's convc2: 192 channels): the 256 x 192 tile whose eight waves all multiply (gemm_pp.hip, WF = 3)
        if (g_conv_nwn == 0 && d.N > 128 && d.N <= 192 && (d.N & 7) == 0 && !use_old_large() && pp_supported(d) && !d.frag_out && !d.init_bf16 && d.gate_from == 0 &&
            !d.resid_bf16 && !d.tail_w && pp_class_enabled(EPI, CONV, 4, false))
            return launch_large_pp<EPI, CONV, 4, 3>(d, s);
    }
    if (nwn == 1) return launch_large_nwn<EPI, CONV, 1>(d, s);
    if (nwn == 2) {
        // The 512 x 128 tile (same wave tile as 256 x 256) measured NO faster than 256 x 128 on RAFT's 128-channel
        // convolutions (GRU q: 1.69 vs 1.71 ms before the epilogue change, 1.76 vs 1.61 ms after it; round 2): every tile
        // shape sits at the same ~30 GB/s per CU of operand fill, which is what bounds them, not the fragment-read ratio.
        // It stays selectable for experiments only.
#ifdef VTGB_DEBUG_HOOKS
        if (g_conv_nwn == 5 && d.M >= 512 * 1024 && !d.col_stats) return launch_large_nwn<EPI, CONV, 2, 8>(d, s);
#endif
        return launch_large_nwn<EPI, CONV, 2>(d, s);
    }
    return launch_large_nwn<EPI, CONV, 4>(d, s);
}

int launch_conv_f32(const GemmDesc& d, hipStream_t s);   // conv_f32.hip: the exactness mode of this entry

// bf16x3 pair store (EPI_SPLIT): persistent kernel only -- 256 x 128 tile up to 128 channels, 256 x 192 (48-column wave tiles) up to 192,
// else 256 x 256
static int launch_split(const GemmDesc& d, hipStream_t s) {
    VTGB_REQUIRE(d.conv_KH > 0 && pp_supported(d), VTGB_EUNSUPPORTED, "conv gemm: pair store needs a convolution with N %% 2 == 0, 4-aligned rows and split_lo (N=%d ldo=%lld split_lo=%d)",
                 d.N, (long long)d.ldo, d.split_lo);
    VTGB_REQUIRE(!d.split_f16c8 || (d.N & 3) == 0, VTGB_EUNSUPPORTED, "conv gemm: an f16c8 pair store needs N %% 4 == 0 (N=%d)", d.N);
    if (d.N <= 128) return launch_large_pp<EPI_SPLIT, true, 2>(d, s);
    if (d.N <= 192 && (d.N & 7) == 0) return launch_large_pp<EPI_SPLIT, true, 4, 3>(d, s);
    return launch_large_pp<EPI_SPLIT, true, 4>(d, s);
}

// round-up magic for n / d, 0 <= n < 2^31, 1 <= d < 2^31: q = (umulhi(n, mul) + n) >> sh
static void magic_div(uint32_t d, uint32_t* mul, uint32_t* sh) {
    uint32_t s = 0;
    while ((1ull << s) < d) s++;
    *mul = (uint32_t)((((1ull << s) - d) << 32) / d + 1);
    *sh = s;
}

int launch_conv_gemm(const GemmDesc& d_in, hipStream_t s) {
    GemmDesc d = d_in;
    if (d.conv_KH > 0 && d.conv_H > 0 && d.conv_W > 0) {
        magic_div((uint32_t)(d.conv_H * d.conv_W), &d.div_hw_mul, &d.div_hw_sh);
        magic_div((uint32_t)d.conv_W, &d.div_w_mul, &d.div_w_sh);
    }
    VTGB_REQUIRE((d.dtype == VTGB_BF16 || d.dtype == VTGB_F32) && d.A && d.W && d.out && d.M > 0 && d.N > 0, VTGB_EINVAL, "conv gemm: bad argument");
    const bool conv = d.conv_KH > 0;
    if (conv) {
        VTGB_REQUIRE(d.zero_page && (d.conv_Cin % L_BK) == 0 && (d.conv_split % L_BK) == 0 && d.K == d.conv_KH * d.conv_KW * d.conv_Cin &&
                         (d.M % (d.conv_H * d.conv_W)) == 0 && (d.conv_split == d.conv_Cin || d.A2),
                     VTGB_EINVAL, "conv gemm: inconsistent convolution geometry");
    }
    if (d.dtype == VTGB_F32) return launch_conv_f32(d, s);
    VTGB_REQUIRE(d.conv_wrap >= 0 && d.conv_wrap2 >= 0 && (d.conv_wrap % L_BK) == 0 && (d.conv_wrap2 % L_BK) == 0 && (conv || (d.conv_wrap | d.conv_wrap2) == 0), VTGB_EINVAL,
                 "conv gemm: pair wrap must be a multiple of 64 channels on a convolution");
    VTGB_REQUIRE((d.K % L_BK) == 0 && (d.lda % 8) == 0 && (d.ldw % 8) == 0, VTGB_EUNSUPPORTED, "conv gemm: K=%d must be a multiple of 64", d.K);
    VTGB_REQUIRE(large_kernel_addressable(d), VTGB_EUNSUPPORTED, "conv gemm: tile footprint beyond the 32-bit offsets of the LDS-DMA descriptors (%dx%d taps)", d.conv_KH, d.conv_KW);
    if (d.gate_from > 0)
        VTGB_REQUIRE(d.epi == EPI_STORE && (d.gate_from % 8) == 0 && (d.N % 8) == 0 && (d.ldo % 8) == 0 && d.aux && d.out2 && (d.ldaux % 8) == 0 &&
                         (d.ldo2 % 8) == 0,
                     VTGB_EINVAL, "conv gemm: gated store needs 8-aligned bf16 rows, aux and out2");
    if (d.resid_bf16)
        VTGB_REQUIRE(d.epi == EPI_STORE && d.gate_from == 0 && (d.N % 8) == 0 && (d.ldo % 8) == 0 && (d.ldrb % 8) == 0, VTGB_EINVAL,
                     "conv gemm: bf16 residual needs 8-aligned bf16 rows");
    if (d.init_bf16)
        VTGB_REQUIRE(d.dtype == VTGB_BF16 && (d.N % 4) == 0 && (d.ldinit % 4) == 0, VTGB_EINVAL, "conv gemm: accumulator start map needs 4-aligned bf16 rows");
    if (d.frag_out)
        VTGB_REQUIRE(d.dtype == VTGB_BF16 && d.epi == EPI_STORE && d.gate_from == 0 && !d.resid_bf16 && !d.tail_w && d.out_scale == 0.f, VTGB_EINVAL,
                     "conv gemm: fragment-order output needs a plain bf16 EPI_STORE launch");
    if (d.init_frag) VTGB_REQUIRE(d.dtype == VTGB_BF16 && d.init_bf16, VTGB_EINVAL, "conv gemm: init_frag without a start map");
    if (d.tail_w)
        VTGB_REQUIRE(d.dtype == VTGB_BF16 && d.epi == EPI_STORE && d.N == 256 && d.gate_from == 0 && !d.resid_bf16 && d.tail_out && (d.ldtail % 4) == 0 &&
                         d.ldtail >= 32 && (d.ldo % 8) == 0,
                     VTGB_EINVAL, "conv gemm: the fused 1x1 tail needs a 256-channel bf16 EPI_STORE convolution and a 4-aligned fp32 output of >= 32 columns");
    if (d.col_stats)
        VTGB_REQUIRE(d.epi == EPI_STORE_F32 && d.stats_rows >= L_BM && (d.N % 4) == 0 && (d.ldo % 4) == 0 && d.act == 0 && d.out_scale == 0.f,
                     VTGB_EINVAL, "conv gemm: column statistics need fp32 whole-row stores and images of >= 256 rows");
    VTGB_REQUIRE(!d.split_f16c8 || d.epi == EPI_SPLIT, VTGB_EINVAL, "conv gemm: split_f16c8 belongs to the pair store");
    switch (d.epi) {
        case EPI_STORE: return conv ? launch_large_forced<EPI_STORE, true>(d, s) : launch_large_forced<EPI_STORE, false>(d, s);
        case EPI_STORE_F32: return conv ? launch_large_forced<EPI_STORE_F32, true>(d, s) : launch_large_forced<EPI_STORE_F32, false>(d, s);
        case EPI_GRU:
            VTGB_REQUIRE(conv && d.resid && d.aux && d.out2, VTGB_EINVAL, "conv gemm: GRU epilogue needs h, z and both outputs");
            return launch_large_forced<EPI_GRU, true>(d, s);
        case EPI_SPLIT: return launch_split(d, s);
        case EPI_X3ZR:
            VTGB_REQUIRE(conv && pp_supported(d), VTGB_EUNSUPPORTED, "conv gemm: the bf16x3 z | r gate epilogue needs a 256-channel convolution with its start map, h and both outputs");
            return launch_large_pp<EPI_X3ZR, true, 4>(d, s);
        case EPI_X3Q:
            VTGB_REQUIRE(conv && pp_supported(d), VTGB_EUNSUPPORTED, "conv gemm: the bf16x3 GRU update epilogue needs a 128-channel convolution with its start map and z");
            return launch_large_pp<EPI_X3Q, true, 2>(d, s);
    }
    vtgb_set_error("conv gemm: unsupported epilogue %d", d.epi);
    return VTGB_EINVAL;
}

int launch_gemm(const GemmDesc& d, hipStream_t s) {
    VTGB_REQUIRE(d.A && d.W && d.out, VTGB_EINVAL, "gemm: NULL operand");
    VTGB_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, VTGB_EINVAL, "gemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
    VTGB_REQUIRE(d.dtype == VTGB_BF16 || d.dtype == VTGB_F32, VTGB_EINVAL, "gemm: bad dtype %d", d.dtype);
    if (d.dtype == VTGB_BF16) {
        VTGB_REQUIRE((d.K % 8) == 0 && (d.lda % 8) == 0 && (d.ldw % 8) == 0, VTGB_EUNSUPPORTED,
                     "gemm bf16: K=%d lda=%lld ldw=%lld must be multiples of 8", d.K, (long long)d.lda, (long long)d.ldw);
        VTGB_REQUIRE(((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.W % 16) == 0, VTGB_EINVAL, "gemm bf16: operands must be 16-byte aligned");
    }
    if (d.epi == EPI_RESID_F32) VTGB_REQUIRE(d.resid != nullptr, VTGB_EINVAL, "gemm: residual epilogue without resid");
    if (d.ln_xb || d.ln_stats)
        VTGB_REQUIRE(d.dtype == VTGB_BF16 && (d.N & 3) == 0 && (d.ldo & 3) == 0 && d.o_map.seg_rows == 0 && d.r_map.seg_rows == 0 && d.a_map.seg_rows == 0 &&
                         (d.ln_xb ? (d.epi == EPI_RESID_F32 && d.ln_part && (d.ldxb & 3) == 0 && (d.ldr & 3) == 0)
                                  : ((d.epi == EPI_STORE || d.epi == EPI_GELU) && !d.bias && d.ln_cs && d.ln_c)),
                     VTGB_EINVAL, "gemm: folded LayerNorm needs bf16, identity row maps, 4-aligned rows and (producer) the partial buffer / (consumer) no bias");
    switch (d.epi) {
        case EPI_STORE: return launch_epi<EPI_STORE>(d, s);
        case EPI_GELU: return launch_epi<EPI_GELU>(d, s);
        case EPI_RESID_F32: return launch_epi<EPI_RESID_F32>(d, s);
        case EPI_STORE_F32: return launch_epi<EPI_STORE_F32>(d, s);
    }
    vtgb_set_error("gemm: bad epilogue %d", d.epi);
    return VTGB_EINVAL;
}
