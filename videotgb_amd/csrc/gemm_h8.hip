// gemm_h8.hip -- the implicit-GEMM convolution of RAFT's update block in the VTGB_F16C8 operand format (pair_h8.h; xraft.py:135-152,
// raft_utils/update.py:39-144): the persistent 256-row tile kernel of gemm_pp.hip (same LDS image, same LDS-DMA staging, same tile walk,
// same accumulator layout, same epilogue structure) with a k-loop that alternates RUNS of two kinds of k-tile:
//   * fp16 k-tiles  -- 64 channels of xh against Wh: 2 x v_mfma_f32_16x16x32_f16 per 16 x 16 block (as the bf16 kernels);
//   * fp8 k-tiles   -- the 128 correction bytes of the same 64 channels ([xl' x 4 | xh8 x 4] per group of four channels) against
//                      [Wh8 x 4 | Wl' x 4]: ONE v_mfma_scale_f32_16x16x128_f8f6f4 per block (weights e4m3, activations e5m2, the power-of-two
//                      scale 2^-11 / sw in the weight operand's E8M0 scale) -- the same 128 bytes per row and the same matrix-pipe cycles as an
//                      fp16 k-tile, for BOTH corrections (tools/exp/f8_mfma_probe.hip: operand bytes pair up position by position whatever
//                      the instruction's k order is, so the fragment reads of the fp16 tiles serve as they are).
// A C-channel source therefore costs 2 C / 64 k-tiles per tap where the bf16x3 form ([hi | lo | hi] against [Wh | Wh | Wl]) costs 3 C / 64.
// K order (weights packed by ops.py h8_conv_pack): per source, `run` = taps x C / 64 fp16 k-tiles, then `run` fp8 k-tiles (chunk major, tap
// minor inside each); all sources of a launch have the same C.
//
// The fp8 instruction needs BOTH 64-byte halves of its operands at once, so the fp16 loop's "multiply one half while the other is being
// read" does not carry over.  An fp8 k-tile is multiplied in four phases over quarters of the (weight fragment, activation fragment) grid,
// ordered so that fragments die early and are re-read from the NEXT k-tile under the remaining phases:
//     1a (w lo, x lo)   1b (w hi, x lo)   -- barrier: next k-tile landed --   2a (w lo, x hi) + read x lo'   2b (w hi, x hi) + read w lo'
//     tail: read w hi', x hi' (they land under the next tile's phase 1a)
// with no more registers than the two fragment sets of the fp16 loop.
#include <type_traits>
#include "common.h"
#include "gemm_dev.h"
#include "pair_h8.h"
#ifndef H8_VAR
#define H8_VAR 0      // (timing experiments: tools/exp/build_variant.sh NAME gemm_h8.hip -DH8_VAR=n.  1 / 2 / 4 / 8: schedule variants of the k-loop, all within +-2 %
                      // of the default; 16: WRONG RESULTS -- the GRU epilogues without their start-map loads: z | r 3.26 -> 3.06 ms, q 1.94 -> 1.81: the upper
                      // bound of what fragment-order start maps loaded into the accumulators at tile start could save, ~0.35 ms per iteration if half of it;
                      // 32: NO RESULTS -- the k-loops alone, every epilogue removed)
#endif

#ifndef H8_DELAY
#define H8_DELAY 6    // (H8_VAR & 64: s_sleep(127) units, ~3.4 us each, per start phase)
#endif

constexpr int EPI_FTAIL = 8;      // (this file only) EPI_SPLIT's arithmetic with GemmDesc::tail_w: relu(conv) is multiplied by a 256 x 32 fp32 matrix in the epilogue and never stored
constexpr int H_BK = 64;
constexpr int H_AOP = 256 * H_BK * 2;   // 32 KiB activation slot (256 rows x 128 bytes)

typedef int h8_i32x4 __attribute__((ext_vector_type(4)));
typedef int h8_i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8_f16x8 __attribute__((ext_vector_type(8)));

#define H_STORE128(data, rs, lane_off, soff) __builtin_amdgcn_raw_buffer_store_b128(data, rs, (lane_off) + (unsigned)(soff), 0, 0)
#define H_STORE64(data, rs, lane_off, soff) __builtin_amdgcn_raw_buffer_store_b64(data, rs, (lane_off) + (unsigned)(soff), 0, 0)
// s_waitcnt immediate for vmcnt(n) alone (gfx9: vmcnt = bits 3:0 and 15:14; expcnt / lgkmcnt left at their maxima)
constexpr int h8_vmcnt(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }
#define H_PHASE_BARRIER() __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);

// OFMT: what the pair-writing epilogues store -- 1 the f16c8 pair (pair_h8.h), 0 the bf16 pair [hi | lo] of the bf16x3 kernels (FlowHead.conv1 /
// mask.0, whose 1x1 successors stay on the bf16x3 kernels)
// WV: waves per workgroup.  8 (the 256-row tile, ONE workgroup per CU) or 4 (a 128-row tile of the same 64 x 64 wave tiles in 80 KB of LDS: TWO workgroups
// per CU, so that one's epilogue -- memory latency, transcendentals -- runs under the other's k-loop; built for the GRU's q convolution, measured SLOWER
// there (-DH8_Q4=1, see launch_conv_h8) and not dispatched)
template <int EPI, int NWN, int WF, int OFMT, int WV = 8>
__global__ __launch_bounds__(64 * WV, 2) void conv_h8_kernel(const GemmDesc p, const int m_tiles, const int n_tiles, const int G, const int total_blocks) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int T_BM = 32 * WV;
    constexpr int MW = WV / NWN;
    constexpr int WROWS = T_BM / MW;
    constexpr int NX = WROWS / 16;       // activation fragments per wave: wave tile = (16 NX) x (16 WF)  (= 2 NWN)
    constexpr int WC = 16 * WF;
    constexpr int T_BN = WC * NWN;
    constexpr int A_OP = T_BM * H_BK * 2, W_OP = T_BN * 128;
    constexpr int AI = 4, WI = T_BN / (8 * WV);
    constexpr int A_SLOTS = 3;
    constexpr int SB = (A_SLOTS - 1) * A_OP / WV;   // epilogue staging bytes per wave (A slots 1, 2)
    static_assert(WV == 8 || (WV == 4 && NWN == 2 && WF == 4 && (EPI == EPI_X3Q || EPI == EPI_X3ZR || EPI == EPI_SPLIT)), "4-wave workgroups: the 128 x 128 tile");
    constexpr int NPRE = AI + WI;
    constexpr int WH = WF / 2, XH = NX / 2;        // the quarters of the fragment grid (WF = 3: 1 + 2 weight fragments)
    static_assert(WF == 4 || (WF == 3 && NWN == 4 && EPI == EPI_SPLIT), "48-column wave tiles: the pair-store convolution (convc2) only");
    static_assert(EPI != EPI_FTAIL || (NWN == 4 && WF == 4), "the flow-head tail lives on the 256-wide tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const smem_w = smem + A_SLOTS * A_OP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % MW, wn = wave / MW;
    const int nk = p.K / H_BK;
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);      // (16-bit units: the fp8 half of a row is addressed as 64 "channels" per 128 bytes)
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(p.W);
    char* const stage = smem + A_OP + wave * SB;
    const int sc_w = *p.h8_scale;        // E8M0 byte of 2^-11 / sw (the activations' own scale is 1: byte 127)

    const int Gn = G * n_tiles;
    auto decode = [&](int q, int& mt_, int& nt_) {
        const int g = q / Gn, r = q - g * Gn;
        const int Gc = min(G, m_tiles - g * G);
        nt_ = r / Gc;
        mt_ = g * G + (r - nt_ * Gc);
    };
    auto next_valid = [&](int q, int& mt_, int& nt_) -> int {
        if (q >= total_blocks) return -1;
        decode(q, mt_, nt_);
        return q;
    };
    const int grid_ = (int)gridDim.x;
    const int first_q = (grid_ & 7) == 0 ? (grid_ >> 3) * ((int)blockIdx.x & 7) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;

    // ---- per-tile LDS-DMA state (gemm_pp.hip: tile descriptors + loop-invariant lane offsets)
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int RANGE = 0x7FFFFF00;
    unsigned w_voff, a_voff[AI];
    int a_bits[AI];
    int m0 = 0, n0 = 0, mt = 0, nt = 0;
    bool wave_active = false;
    auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W), 0, RANGE, 0x00020000);
    auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, RANGE, 0x00020000);
    auto a2_rsrc = a_rsrc;
    const int cv_hw = p.conv_H * p.conv_W, cv_Wi = p.conv_W;
    int cv_ky = 0, cv_kx = 0, cv_c0 = 0;
#define H_TILE_SETUP()                                                                                                       \
    {                                                                                                                        \
        m0 = mt * T_BM; n0 = nt * T_BN;                                                                                      \
        wave_active = (n0 + wn * WC < p.N) && (m0 + wm * WROWS < p.M);                                                       \
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W + (int64_t)n0 * p.ldw), 0, (int)(min(T_BN, p.N - n0) * p.ldw * 2), 0x00020000); \
        {                                                                                                                    \
            const int row = wave * (8 * WI) + (lane >> 3), slot = lane & 7, c = slot ^ ((row >> 1) & 7);                     \
            w_voff = (unsigned)(row * (int)p.ldw + c * 8) * 2u;                                                              \
        }                                                                                                                    \
        const int cv_img0 = m0 / cv_hw;                                                                                      \
        const int64_t a_row0 = (int64_t)cv_img0 * cv_hw;                                                                     \
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A + a_row0 * p.lda), 0, RANGE, 0x00020000);           \
        a2_rsrc = __builtin_amdgcn_make_buffer_rsrc(                                                                         \
            const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(p.A2 ? p.A2 : p.A) + a_row0 * (p.A2 ? p.lda2 : p.lda)), 0, RANGE, 0x00020000); \
        _Pragma("unroll") for (int i = 0; i < AI; i++) {                                                                     \
            const int row = wave * (8 * AI) + i * 8 + (lane >> 3), slot = lane & 7, c = slot ^ ((row >> 1) & 7);             \
            const int am = (m0 + row) < p.M ? (m0 + row) : p.M - 1;                                                          \
            const int img = (int)((__umulhi((unsigned)am, p.div_hw_mul) + (unsigned)am) >> p.div_hw_sh), rem = am - img * cv_hw; \
            const int y = (int)((__umulhi((unsigned)rem, p.div_w_mul) + (unsigned)rem) >> p.div_w_sh), x = rem - y * p.conv_W; \
            a_voff[i] = (unsigned)((img - cv_img0) * cv_hw + y * cv_Wi + x) | ((unsigned)c << 28);                           \
            const int py = p.conv_KH >> 1, px = p.conv_KW >> 1;                                                              \
            const int ylo = max(0, py - y), yhi = min(p.conv_KH - 1, p.conv_H - 1 - y + py), xlo = max(0, px - x), xhi = min(p.conv_KW - 1, cv_Wi - 1 - x + px); \
            const int yb = yhi >= ylo ? ((2 << yhi) - 1) & ~((1 << ylo) - 1) : 0, xb = xhi >= xlo ? ((2 << xhi) - 1) & ~((1 << xlo) - 1) : 0; \
            a_bits[i] = yb | (xb << 8);                                                                                      \
        }                                                                                                                    \
        cv_ky = 0; cv_kx = 0; cv_c0 = 0;                                                                                     \
    }
#define H_ISSUE_A(slot)                                                                                 \
    {                                                                                                   \
        const bool first = cv_c0 < p.conv_split;                                                        \
        const unsigned ldb = (unsigned)(first ? p.lda : p.lda2) * 2u;                                   \
        const int cc2 = (first ? cv_c0 : cv_c0 - p.conv_split) * 2;                                     \
        const int dpix = (cv_ky - (p.conv_KH >> 1)) * cv_Wi + (cv_kx - (p.conv_KW >> 1));               \
        const int need = (1 << cv_ky) | (256 << cv_kx);                                                 \
        const auto rs_ = first ? a_rsrc : a2_rsrc;                                                      \
        _Pragma("unroll") for (int i = 0; i < AI; i++) {                                                \
            const unsigned pix = (a_voff[i] & 0x00FFFFFFu) + (unsigned)dpix;                            \
            const unsigned vin = __umul24(pix, ldb) + (a_voff[i] >> 28) * 16u;                          \
            const unsigned v = ((a_bits[i] & need) == need) ? vin : OOB;                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lptr_t)(smem + (slot) * A_OP + (wave * (8 * AI) + i * 8) * 128), 16, v, cc2, 0, 0); \
        }                                                                                               \
        if (++cv_kx == p.conv_KW) { cv_kx = 0; if (++cv_ky == p.conv_KH) { cv_ky = 0; cv_c0 += H_BK; } } \
    }
#define H_ISSUE_W(slot, k0)                                                                             \
    _Pragma("unroll") for (int i = 0; i < WI; i++)                                                      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lptr_t)(smem_w + (slot) * W_OP + (wave * (8 * WI) + i * 8) * 128), 16, w_voff ^ ((i & 1) * 64), (k0) * 2 + i * 8 * (int)p.ldw * 2, 0, 0);

    // fragments: 32 bytes per (16-row block, lane) = the two 64-byte halves' 16-byte pieces, as ONE register tuple (the fp8 instruction's operand);
    // the fp16 instruction takes either half of it
    h8_i32x8 wfr[WF], xfr[NX];
#define H_LW(i, H, ws_)                                                                                                   \
    {                                                                                                                     \
        const h8_i32x4 t_ = *reinterpret_cast<const h8_i32x4*>((ws_) + (w_off0 ^ ((H) * 64)) + (i) * 2048);               \
        if ((H) == 0) wfr[i].lo = t_; else wfr[i].hi = t_;                                                                \
    }
#define H_LX(j, H, as_)                                                                                                   \
    {                                                                                                                     \
        const h8_i32x4 t_ = *reinterpret_cast<const h8_i32x4*>((as_) + (x_off0 ^ ((H) * 64)) + (j) * 2048);               \
        if ((H) == 0) xfr[j].lo = t_; else xfr[j].hi = t_;                                                                \
    }
#define H_MF16(i, j, H)                                                                                                   \
    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_f16x8, (H) == 0 ? wfr[i].lo : wfr[i].hi),    \
                                                       __builtin_bit_cast(h8_f16x8, (H) == 0 ? xfr[j].lo : xfr[j].hi), acc[i][j], 0, 0, 0);
#define H_MF8(i, j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wfr[i], xfr[j], acc[i][j], 0, 1, 0, sc_w, 0, 127);
#define H_MF8_BLOCK(i0, i1, j0, j1)                                                                      \
    _Pragma("unroll") for (int i = (i0); i < (i1); i++)                                                  \
        _Pragma("unroll") for (int j = (j0); j < (j1); j++) { H_MF8(i, j) }
#define H_SCHED_IL(PIECES, NMF)                                                                          \
    if constexpr ((H8_VAR & 2) == 0 && (PIECES) > 0 && (NMF) % (PIECES) == 0) {                                               \
        _Pragma("unroll") for (int g_ = 0; g_ < (PIECES); g_++) {                                        \
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                                           \
            __builtin_amdgcn_sched_group_barrier(0x008, (NMF) / (PIECES), 0);                            \
        }                                                                                                \
    }

    int b = next_valid(first_q, mt, nt);
    if (b < 0) return;
#if (H8_VAR & 64)
    {   // (timing experiment: four start phases a quarter of a tile apart, so that the CUs' epilogues -- HBM bursts -- do not coincide)
        const int steps = (((int)blockIdx.x >> 3) & 3) * H8_DELAY;
        for (int i = 0; i < steps; i++) __builtin_amdgcn_s_sleep(127);
    }
#endif
    H_TILE_SETUP()
    H_ISSUE_A(0)
    H_ISSUE_W(0, 0)
    while (true) {
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        const int w_off0 = swz(wn * WC + (lane_k & 15), lane_k >> 4), x_off0 = swz(wm * WROWS + (lane_k & 15), lane_k >> 4);
        const int fg = lane_k >> 4;
        f32x4 acc[WF][NX];
        {
            f32x4 b4[WF];
#pragma unroll
            for (int i = 0; i < WF; i++) {
                const int n = n0 + wn * WC + i * 16 + fg * 4;
                b4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias && n + 3 < p.N) b4[i] = *reinterpret_cast<const f32x4*>(p.bias + n);
                else if (p.bias && n < p.N) {
                    for (int e = 0; e < 4 && n + e < p.N; e++) b4[i][e] = p.bias[n + e];
                }
            }
#pragma unroll
            for (int j = 0; j < NX; j++)
#pragma unroll
                for (int i = 0; i < WF; i++) acc[i][j] = b4[i];
        }
        // ---------- k-loop (nk >= 4, runs of >= 2 k-tiles; the first run is fp16, the last fp8).  A(g) in slot g % 3, W(g) in slot g % 2
        H_ISSUE_A(1) H_ISSUE_W(1, H_BK)
        H_ISSUE_A(2)
        __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * AI + WI));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        int a_slot = 0;
        if (!wave_active) {
            for (int kt = 0; kt + 1 < nk; kt++) {
                if (kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | AI);
                else __builtin_amdgcn_s_waitcnt(0x0F70);
                __builtin_amdgcn_s_barrier();
                if (kt + 2 < nk) { H_ISSUE_W(kt & 1, (kt + 2) * H_BK) }
                if (kt + 3 < nk) { H_ISSUE_A(a_slot) }
                a_slot = a_slot == 2 ? 0 : a_slot + 1;
            }
        } else {
            // One k-tile of either kind, multiplied in four phases over the quarters of the fragment grid (file header); KIND 0: fp16 (two
            // 32-deep MFMAs per block, the eight first halves of a quarter before its eight second halves), KIND 1: fp8 (one 128-deep MFMA).
            // The fragments of the whole k-tile are in registers (or on their way) at the top; the next k-tile's are read under phases 2a / 2b
            // and behind them.  Every LDS read of a k-tile is complete at the NEXT k-tile's barrier (lgkmcnt(0) in front of it), after which its
            // slots are re-armed (W(t+2) right there, A(t+3) at the top of the k-tile after).
#define H_MF_BLOCK(KIND, i0, i1, j0, j1)                                                                 \
    if constexpr ((KIND) == 1) {                                                                         \
        H_MF8_BLOCK(i0, i1, j0, j1)                                                                      \
    } else if constexpr ((H8_VAR & 1) != 0) {                                                            \
        _Pragma("unroll") for (int i = (i0); i < (i1); i++)                                              \
            _Pragma("unroll") for (int j = (j0); j < (j1); j++) { H_MF16(i, j, 0) H_MF16(i, j, 1) }      \
    } else {                                                                                             \
        _Pragma("unroll") for (int i = (i0); i < (i1); i++)                                              \
            _Pragma("unroll") for (int j = (j0); j < (j1); j++) { H_MF16(i, j, 0) }                      \
        _Pragma("unroll") for (int i = (i0); i < (i1); i++)                                              \
            _Pragma("unroll") for (int j = (j0); j < (j1); j++) { H_MF16(i, j, 1) }                      \
    }
#define H_ITER(KIND, DEFER, WCOND, WAIT4)                                                                \
    {                                                                                                    \
        constexpr int MPB = (KIND) == 1 ? 1 : 2;      /* MFMAs per block */                              \
        const int a_nxt = a_slot == 2 ? 0 : a_slot + 1;                                                  \
        const int a_prv = a_slot == 0 ? 2 : a_slot - 1;                                                  \
        if (DEFER) { H_ISSUE_A(a_prv) }                                                                  \
        if constexpr ((H8_VAR & 4) != 0) __builtin_amdgcn_s_setprio(1);                                  \
        H_MF_BLOCK(KIND, 0, WH, 0, XH)                                                                   \
        H_MF_BLOCK(KIND, WH, WF, 0, XH)                                                                  \
        H_SCHED_IL(AI, WF * XH * MPB)                                                                    \
        if constexpr ((H8_VAR & 4) != 0) __builtin_amdgcn_s_setprio(0);                                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (WAIT4) __builtin_amdgcn_s_waitcnt(0x0070 | AI);                                              \
        else __builtin_amdgcn_s_waitcnt(0x0070);                                                         \
        __builtin_amdgcn_s_barrier();                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (WCOND) { H_ISSUE_W(kt & 1, (kt + 2) * H_BK) }                                                \
        {                                                                                                \
            const char* an = smem + a_nxt * A_OP;                                                        \
            const char* wn_ = smem_w + ((kt + 1) & 1) * W_OP;                                            \
            if constexpr ((H8_VAR & 8) != 0) {                                                           \
                H_MF_BLOCK(KIND, 0, WH, XH, NX)                                                          \
                _Pragma("unroll") for (int j = 0; j < XH; j++) { H_LX(j, 0, an) H_LX(j, 1, an) }         \
                __builtin_amdgcn_sched_barrier(0);                                                       \
                H_MF_BLOCK(KIND, WH, WF, XH, NX)                                                         \
                _Pragma("unroll") for (int i = 0; i < WH; i++) { H_LW(i, 0, wn_) H_LW(i, 1, wn_) }       \
                __builtin_amdgcn_sched_barrier(0);                                                       \
            } else {                                                                                     \
            _Pragma("unroll") for (int j = 0; j < XH; j++) { H_LX(j, 0, an) H_LX(j, 1, an) }             \
            H_MF_BLOCK(KIND, 0, WH, XH, NX)                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            _Pragma("unroll") for (int i = 0; i < WH; i++) { H_LW(i, 0, wn_) H_LW(i, 1, wn_) }           \
            H_MF_BLOCK(KIND, WH, WF, XH, NX)                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            }                                                                                            \
            _Pragma("unroll") for (int i = WH; i < WF; i++) { H_LW(i, 0, wn_) H_LW(i, 1, wn_) }          \
            _Pragma("unroll") for (int j = XH; j < NX; j++) { H_LX(j, 0, an) H_LX(j, 1, an) }            \
        }                                                                                                \
        a_slot = a_nxt;                                                                                  \
    }
            {
                _Pragma("unroll") for (int i = 0; i < WF; i++) { H_LW(i, 0, smem_w) H_LW(i, 1, smem_w) }
                _Pragma("unroll") for (int j = 0; j < NX; j++) { H_LX(j, 0, smem) H_LX(j, 1, smem) }
            }
            int kt = 0;
            { H_ITER(0, false, true, true) kt = 1; }
            // runs: [fp16 x run | fp8 x run] per source, as straight-line inner loops (a two-way branch per k-tile between two bodies made hipcc
            // spill ~800 registers); the last two k-tiles (fp8) are peeled: nothing is left to issue there
            const int run = p.h8_run, npair = nk / (2 * run);
            for (int pr = 0; pr < npair; pr++) {
                for (int r = pr == 0 ? 1 : 0; r < run; r++, kt++) { H_ITER(0, true, true, true) }
                const int n8 = pr == npair - 1 ? run - 2 : run;
                for (int r = 0; r < n8; r++, kt++) { H_ITER(1, true, true, true) }
            }
            { H_ITER(1, false, false, false) kt++; }      // k-tile nk - 2 (fp8): nothing left to issue
            {   // last k-tile (fp8; its fragments were read by the tile before)
                H_MF8_BLOCK(0, WF, 0, NX)
            }
#undef H_ITER
#undef H_MF_BLOCK
        }
        H_PHASE_BARRIER()                      // every wave's LDS reads and DMAs of this tile are done: the ring is free

        // ================= the next tile: its setup replaces this tile's DMA state, its first k-tile is issued by the epilogue
        const int em0 = m0, en0 = n0, emt = mt;
        (void)emt;
        const bool eactive = wave_active;
        (void)eactive;
        const int b2 = next_valid(b + (int)gridDim.x, mt, nt);
        const bool has_next = b2 >= 0;
        if (has_next) H_TILE_SETUP()
        constexpr bool PARK = NWN == 4;
        uint4* const park = reinterpret_cast<uint4*>(smem_w + W_OP) + tid;
        auto prefetch = [&]() {
            if (has_next) {
                H_ISSUE_A(0)
                H_ISSUE_W(0, 0)
                if constexpr (PARK) {
                    park[0] = make_uint4(a_voff[0], a_voff[1], a_voff[2], a_voff[3]);
                    park[512] = make_uint4((unsigned)a_bits[0], (unsigned)a_bits[1], (unsigned)a_bits[2], (unsigned)a_bits[3]);
                    park[1024] = make_uint4(w_voff, 0u, 0u, 0u);
                }
            }
        };
#define H_WAIT_OPERANDS() if (has_next) __builtin_amdgcn_s_waitcnt(0x0F70 | NPRE); else __builtin_amdgcn_s_waitcnt(0x0F70);

        // ================= epilogue of tile (em0, en0): gemm_pp.hip's pair epilogues with the pair written / read in the f16c8 format
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int rows_here = min(T_BM, p.M - em0);
        auto tile_rsrc = [&](const void* base, int64_t ld, int es) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + (int64_t)em0 * ld * es, 0, (int)(rows_here * ld * es), 0x00020000);
        };
        typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
        typedef h8_u32x2 u32x2_t;
        // v -> the pair's two 8-byte words
        auto encode = [&](const f32x4 v, u32x2_t& hu, u32x2_t& lu) {
            if constexpr (OFMT == 1) {
                h8_split4(v, hu, lu);
            } else {
                const f32x4 hf = {bf16_round(v[0]), bf16_round(v[1]), bf16_round(v[2]), bf16_round(v[3])};
                const bf16x4 hi = {(bf16_t)hf[0], (bf16_t)hf[1], (bf16_t)hf[2], (bf16_t)hf[3]};
                const bf16x4 lo = {(bf16_t)(v[0] - hf[0]), (bf16_t)(v[1] - hf[1]), (bf16_t)(v[2] - hf[2]), (bf16_t)(v[3] - hf[3])};
                hu = __builtin_bit_cast(u32x2_t, hi); lu = __builtin_bit_cast(u32x2_t, lo);
            }
        };
        constexpr bool NO_EPI = (H8_VAR & 32) != 0;
        if constexpr (NO_EPI) {      // (timing experiment: the k-loops alone -- the accumulators are kept alive, nothing is stored)
            prefetch();
#pragma unroll
            for (int i = 0; i < WF; i++)
#pragma unroll
                for (int j = 0; j < NX; j++) asm volatile("" ::"v"(acc[i][j]));
        }
        if constexpr (EPI == EPI_SPLIT && !NO_EPI) {
            constexpr int PR = SB / 256 < WROWS ? SB / 256 : WROWS, NP = WROWS / PR;
            const int rl = lane_e >> 4, cl = lane_e & 15;
            const int n = en0 + wn * WC + cl * 4;
            const bool inw = cl * 4 < WC;
            const auto o_rs = tile_rsrc(p.out, p.ldo, 2);
            const unsigned o_off = (unsigned)((wm * WROWS + rl) * (int)p.ldo + n) * 2u;
            const unsigned o_lane = (inw && n + 3 < p.N) ? o_off : OOB;
            const unsigned o_part = (inw && n < p.N && n + 3 >= p.N) ? o_off : OOB;      // (N % 4 == 2: two columns)
            const bool cut = (p.N & 3) != 0;
            const unsigned lo_b = (unsigned)p.split_lo * 2u;
            // 64- and 128-wide tiles (the context encoder's residual blocks, BatchNorm folded): the ResidualBlock's tail relu(x + relu(conv)) in the
            // epilogue -- resid_bf16 = the skip operand as an f16c8 pair (row stride ldrb, second half at + split_lo), post_relu; all its rows are requested
            // up front around the next tile's first pieces (the k-loop's fragment registers are dead here).  One straight-line instance per case: a
            // run-time `if (residual)` around the loads made hipcc's own waits at the joins drain the stores of the pass before.
            constexpr bool RES = NWN <= 2;
            auto split_epilogue = [&](auto with_resid) {
                constexpr bool RD = decltype(with_resid)::value;
                u32x2_t rh[RD ? NP : 1][RD ? PR / 4 : 1], rl8[RD ? NP : 1][RD ? PR / 4 : 1];
                if constexpr (RD) {
                    const auto r_rs = tile_rsrc(p.resid_bf16, p.ldrb, 2);
                    const unsigned r_lane = (inw && n + 3 < p.N) ? (unsigned)((wm * WROWS + rl) * (int)p.ldrb + n) * 2u : OOB;
#pragma unroll
                    for (int ps = 0; ps < NP; ps++) {
#pragma unroll
                        for (int rr = 0; rr < PR / 4; rr++) {
                            rh[ps][rr] = __builtin_amdgcn_raw_buffer_load_b64(r_rs, r_lane, (ps * PR + rr * 4) * (int)p.ldrb * 2, 0);
                            rl8[ps][rr] = __builtin_amdgcn_raw_buffer_load_b64(r_rs, r_lane + lo_b, (ps * PR + rr * 4) * (int)p.ldrb * 2, 0);
                        }
                        if (ps == 0) prefetch();
                    }
                } else {
                    prefetch();
                }
#pragma unroll
                for (int ps = 0; ps < NP; ps++) {
#pragma unroll
                    for (int jj = 0; jj < PR / 16; jj++)
#pragma unroll
                        for (int i = 0; i < WF; i++) {
                            const int row = jj * 16 + (lane_e & 15), chunk = i * 4 + (lane_e >> 4);
                            *reinterpret_cast<f32x4*>(stage + row * 256 + ((chunk ^ (row & 15)) << 4)) = acc[i][ps * (PR / 16) + jj];
                        }
                    if constexpr (RD) {      // this pass's skip rows have landed (behind them: the later passes' rows, the next tile's pieces, earlier stores)
                        constexpr int LATER = (NP - 1) * (PR / 2);
                        if (ps == 0) {
                            if (has_next) __builtin_amdgcn_s_waitcnt(h8_vmcnt(NPRE + LATER)); else __builtin_amdgcn_s_waitcnt(h8_vmcnt(LATER));
                        } else __builtin_amdgcn_s_waitcnt(h8_vmcnt(PR / 2));      // (the last pass: only the previous pass's stores may still fly)
                    }
#pragma unroll
                    for (int rr = 0; rr < PR / 4; rr++) {
                        const int row = rr * 4 + rl;
                        f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 256 + ((cl ^ (row & 15)) << 4));
                        if (p.act) apply_act4(v, p.act);
                        if constexpr (RD) {
                            v += h8_join4(rh[ps][rr], rl8[ps][rr]);
                            if (p.post_relu) apply_act4(v, 1);
                        }
                        u32x2_t hu, lu;
                        encode(v, hu, lu);
                        const unsigned so = (unsigned)((ps * PR + rr * 4) * (int)p.ldo * 2);
                        H_STORE64(hu, o_rs, o_lane, so);
                        H_STORE64(lu, o_rs, o_lane + lo_b, so);
                        if (cut) {      // the group that straddles N: its first two columns
                            __builtin_amdgcn_raw_buffer_store_b32(hu[0], o_rs, o_part + so, 0, 0);
                            if constexpr (OFMT == 1) {
                                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(lu[0] & 0xFFFFu), o_rs, o_part + lo_b + so, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(lu[1] & 0xFFFFu), o_rs, o_part + lo_b + 4u + so, 0, 0);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b32(lu[0], o_rs, o_part + lo_b + so, 0, 0);
                            }
                        }
                    }
                }
            };
            if constexpr (RES) {
                if (p.resid_bf16 != nullptr) split_epilogue(std::true_type{});
                else split_epilogue(std::false_type{});
            } else {
                split_epilogue(std::false_type{});
            }
        }
        if constexpr (EPI == EPI_FTAIL && !NO_EPI) {
            // FlowHead (update.py:10-18): hidden = relu(conv1(h) + b) [256 channels] is followed by a 3x3 convolution to 2 channels = per pixel the 18 per-tap
            // products hidden . W2[tap][c] (summed over the taps' neighbours by raft_flow_head2_kernel).  Those products are formed HERE, from the
            // accumulators, on the fp32-input matrix instruction -- D2[out][pixel] += W2t[out][k] v[k][pixel] with the k of a step = the four channels
            // 16 i + 4 fg + e (fg = 0..3) that the lanes of a 16 x 16 accumulator block hold in element e: a lane's own register IS its B operand -- so the
            // 1 KB-per-pixel hidden map is neither stored nor re-read, and the product is exact fp32 (the bf16x3 1x1 launch it replaces rounded the map
            // to a bf16 pair).  tail_w: fp32, packed per (wn, i, lane) as [e][ob] (ops.py flow_tail_pack); the four column waves' partial sums meet in the
            // staging region, in wave order (deterministic), half a tile at a time (64 KB).
            const int fr = lane_e & 15, fg4 = lane_e >> 4;
            f32x4 wa4[4][2];
            {
                const f32x4* wp = reinterpret_cast<const f32x4*>(p.tail_w) + ((int64_t)(wn * 4) * 64 + lane_e) * 2;
#pragma unroll
                for (int i = 0; i < 4; i++) { wa4[i][0] = wp[i * 128]; wa4[i][1] = wp[i * 128 + 1]; }
            }
            prefetch();
            float* const red = reinterpret_cast<float*>(smem + A_OP);
#pragma unroll
            for (int half = 0; half < 2; half++) {
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int j = half * 4 + jj;
                    f32x4 d2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float bv = p.act ? fmaxf(acc[i][j][e], 0.f) : acc[i][j][e];
                            d2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa4[i][e >> 1][(e & 1) * 2], bv, d2[0], 0, 0, 0);
                            d2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa4[i][e >> 1][(e & 1) * 2 + 1], bv, d2[1], 0, 0, 0);
                        }
                    const int rowh = wm * 64 + jj * 16 + fr;
#pragma unroll
                    for (int ob = 0; ob < 2; ob++) *reinterpret_cast<f32x4*>(red + ((wn * 128 + rowh) * 32 + 16 * ob + 4 * fg4)) = d2[ob];
                }
                H_PHASE_BARRIER()
#pragma unroll
                for (int rep = 0; rep < 2; rep++) {
                    const int idx = tid + rep * 512, rowh = idx >> 3, oc = idx & 7;
                    f32x4 sum = *reinterpret_cast<const f32x4*>(red + (rowh * 32 + oc * 4));
#pragma unroll
                    for (int w_ = 1; w_ < 4; w_++) sum += *reinterpret_cast<const f32x4*>(red + ((w_ * 128 + rowh) * 32 + oc * 4));
                    const int64_t grow = (int64_t)em0 + (rowh >> 6) * 128 + half * 64 + (rowh & 63);
                    if (grow < p.M) *reinterpret_cast<f32x4*>(p.tail_out + grow * p.ldtail + oc * 4) = sum;
                }
                if (half == 0) { H_PHASE_BARRIER() }      // (the second half's partial sums overwrite the region; behind it: the loop's own barrier)
            }
        }
        if constexpr (EPI == EPI_STORE_F32 && !NO_EPI) {
            // fp32 outputs (the encoders' layer1: the InstanceNorm / skip pass behind it needs the unrounded sums) through the staging region as whole
            // 256-byte row segments, with gemm_pp.hip's per-tile column moments (GemmDesc::col_stats: stored per m-tile, added in tile order by
            // launch_stats_finish_tiles -- no atomics)
            constexpr int PR = SB / 256 < WROWS ? SB / 256 : WROWS, NP = WROWS / PR;
            const int rl = lane_e >> 4, cl = lane_e & 15;
            const bool do_stats = p.col_stats != nullptr;
            const int img_a = do_stats ? em0 / p.stats_rows : 0;
            const int m_b = do_stats ? (img_a + 1) * p.stats_rows : 0x7fffffff;
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, qa = sa, sb = sa, qb = sa;
            const int n = en0 + wn * 64 + cl * 4;
            const auto o_rs = tile_rsrc(p.out, p.ldo, 4);
            const unsigned o_lane = n < p.N ? (unsigned)((wm * WROWS + rl) * (int)p.ldo + n) * 4u : OOB;
            prefetch();
#pragma unroll
            for (int ps = 0; ps < NP; ps++) {
#pragma unroll
                for (int jj = 0; jj < PR / 16; jj++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int row = jj * 16 + (lane_e & 15), chunk = i * 4 + (lane_e >> 4);
                        *reinterpret_cast<f32x4*>(stage + row * 256 + ((chunk ^ (row & 15)) << 4)) = acc[i][ps * (PR / 16) + jj];
                    }
#pragma unroll
                for (int rr = 0; rr < PR / 4; rr++) {
                    const int row = rr * 4 + rl;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 256 + ((cl ^ (row & 15)) << 4));
                    H_STORE128(__builtin_bit_cast(u32x4_t, v), o_rs, o_lane, (ps * PR + rr * 4) * (int)p.ldo * 4);
                    if (do_stats) {
                        const int m = em0 + wm * WROWS + ps * PR + rr * 4 + rl;
                        if (m < p.M && n < p.N && eactive) {
                            if (m < m_b) { sa += v; qa += v * v; }
                            else { sb += v; qb += v * v; }
                        }
                    }
                }
            }
            if (do_stats) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    sa[e] += __shfl_xor(sa[e], 16); sa[e] += __shfl_xor(sa[e], 32);
                    qa[e] += __shfl_xor(qa[e], 16); qa[e] += __shfl_xor(qa[e], 32);
                    sb[e] += __shfl_xor(sb[e], 16); sb[e] += __shfl_xor(sb[e], 32);
                    qb[e] += __shfl_xor(qb[e], 16); qb[e] += __shfl_xor(qb[e], 32);
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);      // (my staging reads are done before the region is reused)
                if (rl == 0) {
                    f32x4* const pr = reinterpret_cast<f32x4*>(stage);
                    pr[cl] = sa; pr[16 + cl] = qa; pr[32 + cl] = sb; pr[48 + cl] = qb;    // (a row-inactive wave parks zeros)
                }
                H_PHASE_BARRIER()
                if (tid < 64 * NWN) {
                    const int wn_ = tid >> 6, c = tid & 63, nc = en0 + wn_ * 64 + c;
                    if (nc < p.N) {
                        float t[4] = {0.f, 0.f, 0.f, 0.f};
                        for (int wm_ = 0; wm_ < MW; wm_++) {
                            if (em0 + wm_ * WROWS >= p.M) break;
                            const float* pr = reinterpret_cast<const float*>(smem + A_OP + (wn_ * MW + wm_) * SB);
#pragma unroll
                            for (int e = 0; e < 4; e++) t[e] += pr[e * 64 + c];
                        }
                        *reinterpret_cast<float4*>(p.col_stats + ((int64_t)emt * p.N + nc) * 4) = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
            }
        }
        if constexpr ((EPI == EPI_X3ZR || EPI == EPI_X3Q) && !NO_EPI) {
            // SepConvGRU gates in the convolution's epilogue (common.h: VTGB_EPI_X3ZR / X3Q), h / r h / h' as f16c8 pairs
            static_assert(EPI == EPI_SPLIT || OFMT == 1, "the GRU's pairs are f16c8 pairs");
            constexpr bool ZR = EPI == EPI_X3ZR;
            constexpr int PR = SB / 256 < WROWS ? SB / 256 : WROWS, NP = WROWS / PR, NS = NP * (PR / 16);
            static_assert(EPI == EPI_SPLIT || PR == 32, "two sub-passes of 16 rows per staged pass");
            const int rl = lane_e >> 4, cl = lane_e & 15;
            const int n = en0 + wn * 64 + cl * 4;
            const bool is_r = ZR && en0 + wn * 64 >= 128;          // wave-uniform, and from scalar values only: as `n >= 128` (n holds the lane's column) hipcc predicated
                                                                   // the r waves' loads under exec masks and its own waits then drained the previous sub-pass's stores
            const int c = is_r ? n - 128 : n;
            const int rowl = wm * WROWS + rl;
            const unsigned lo_b = (unsigned)p.split_lo * 2u;
            const auto m_rs = tile_rsrc(p.resid, p.ldr, 4);                                      // fp32 start map
            const auto h_rs = ZR ? tile_rsrc(p.aux, p.ldaux, 2) : tile_rsrc(p.out, p.ldo, 2);     // h pair
            const auto z_rs = ZR ? tile_rsrc(p.out, p.ldo, 4) : tile_rsrc(p.aux, p.ldaux, 4);     // z fp32 (X3ZR: written; X3Q: read)
            const auto o_rs = ZR ? tile_rsrc(p.out2, p.ldo2, 2) : tile_rsrc(p.out, p.ldo, 2);     // pair output (r h | h')
            const int ld_h = ZR ? (int)p.ldaux : (int)p.ldo, ld_z = ZR ? (int)p.ldo : (int)p.ldaux, ld_o = ZR ? (int)p.ldo2 : (int)p.ldo;
            const bool ok = n < p.N;
            const unsigned m_lane = ok ? (unsigned)(rowl * (int)p.ldr + n) * 4u : OOB, h_lane = ok ? (unsigned)(rowl * ld_h + c) * 2u : OOB;
            const unsigned z_lane = ok ? (unsigned)(rowl * ld_z + c) * 4u : OOB, o_lane = ok ? (unsigned)(rowl * ld_o + c) * 2u : OOB;
            // One straight-line instance per wave role (IS_R: the r columns of the z | r launch; the q launch has one role): with the role as a run-time
            // condition inside the loops hipcc's own s_waitcnt at every join assumed the conditional loads away and drained the previous sub-pass's
            // stores before each sub-pass (vmcnt(3) .. vmcnt(0) in front of the four rows).  Operands are requested TWO sub-passes ahead into
            // alternating register sets (the k-loop's fragment registers are dead here).
            auto gru_epilogue = [&](auto role) {
                constexpr bool IS_R = decltype(role)::value;
                constexpr bool HL = !ZR || IS_R;           // this role reads h
                constexpr int NLD = (ZR ? 4 : 8) + (HL ? 8 : 0), NST = (ZR && !IS_R) ? 4 : 8;      // vector-memory operations per sub-pass: loads, stores
                u32x4_t mq[2][4], zq[2][4];
                u32x2_t hh[2][4], hl[2][4];
#define H_X3_LOAD(sp)                                                                                                  \
    _Pragma("unroll") for (int rr = 0; rr < 4; rr++) {                                                                 \
        const int r0_ = (sp) * 16 + rr * 4;                                                                            \
        const int S_ = (sp) & 1;                                                                                       \
        if constexpr ((H8_VAR & 16) == 0) mq[S_][rr] = __builtin_amdgcn_raw_buffer_load_b128(m_rs, m_lane, r0_ * (int)p.ldr * 4, 0); else mq[S_][rr] = u32x4_t{0u, 0u, 0u, 0u};   \
        if constexpr (!ZR) zq[S_][rr] = __builtin_amdgcn_raw_buffer_load_b128(z_rs, z_lane, r0_ * ld_z * 4, 0);         \
        if constexpr (HL) {                                                                                            \
            hh[S_][rr] = __builtin_amdgcn_raw_buffer_load_b64(h_rs, h_lane, r0_ * ld_h * 2, 0);                         \
            hl[S_][rr] = __builtin_amdgcn_raw_buffer_load_b64(h_rs, h_lane + lo_b, r0_ * ld_h * 2, 0);                  \
        }                                                                                                              \
    }
                H_X3_LOAD(0)
                prefetch();
                H_X3_LOAD(1)
#pragma unroll
                for (int ps = 0; ps < NP; ps++) {
#pragma unroll
                    for (int jj = 0; jj < PR / 16; jj++)
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int row = jj * 16 + (lane_e & 15), chunk = i * 4 + (lane_e >> 4);
                            *reinterpret_cast<f32x4*>(stage + row * 256 + ((chunk ^ (row & 15)) << 4)) = acc[i][ps * (PR / 16) + jj];
                        }
#pragma unroll
                    for (int sub = 0; sub < 2; sub++) {
                        const int sp = ps * 2 + sub;
                        constexpr int AHEAD = NLD + NST;      // in flight behind this sub-pass's operands: the next sub-pass's loads, the previous one's stores
                        if (sp == 0) {
                            if (has_next) __builtin_amdgcn_s_waitcnt(h8_vmcnt(NPRE + NLD)); else __builtin_amdgcn_s_waitcnt(h8_vmcnt(NLD));
                        } else if (sp + 1 < NS) __builtin_amdgcn_s_waitcnt(h8_vmcnt(AHEAD));
                        else __builtin_amdgcn_s_waitcnt(h8_vmcnt(NST));
                        const int S = sp & 1;
                        f32x4 res[4];
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) {
                            const int row = sub * 16 + rr * 4 + rl;
                            f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 256 + ((cl ^ (row & 15)) << 4));
                            v += __builtin_bit_cast(f32x4, mq[S][rr]);
                            if constexpr (ZR) {
#pragma unroll
                                for (int e = 0; e < 4; e++) v[e] = __builtin_amdgcn_rcpf(1.0f + __expf(-v[e]));      // (v_rcp_f32: 1 ulp; __frcp_rn expands to the 10-instruction IEEE division)
                                if constexpr (IS_R) v *= h8_join4(hh[S][rr], hl[S][rr]);
                            } else {
                                const f32x4 z = __builtin_bit_cast(f32x4, zq[S][rr]);
                                const f32x4 h = h8_join4(hh[S][rr], hl[S][rr]);
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const float q = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v[e]) + 1.0f);
                                    v[e] = (1.0f - z[e]) * h[e] + z[e] * q;
                                }
                            }
                            res[rr] = v;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (sp + 2 < NS) { H_X3_LOAD(sp + 2) }       // into the set this sub-pass has just consumed, before this sub-pass's stores
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) {
                            const unsigned r0 = (unsigned)(sp * 16 + rr * 4);
                            if constexpr (ZR && !IS_R) {
                                H_STORE128(__builtin_bit_cast(u32x4_t, res[rr]), z_rs, z_lane, r0 * (unsigned)ld_z * 4u);
                            } else {
                                u32x2_t hu, lu;
                                encode(res[rr], hu, lu);
                                H_STORE64(hu, o_rs, o_lane, r0 * (unsigned)ld_o * 2u);
                                H_STORE64(lu, o_rs, o_lane + lo_b, r0 * (unsigned)ld_o * 2u);
                            }
                        }
                    }
                }
#undef H_X3_LOAD
            };
            if (is_r) gru_epilogue(std::true_type{});
            else gru_epilogue(std::false_type{});
        }
        if (!has_next) break;
        b = b2;
        if constexpr (PARK) {
            const uint4 t0 = park[0], t1 = park[512], t2 = park[1024];
            a_voff[0] = t0.x; a_voff[1] = t0.y; a_voff[2] = t0.z; a_voff[3] = t0.w;
            a_bits[0] = (int)t1.x; a_bits[1] = (int)t1.y; a_bits[2] = (int)t1.z; a_bits[3] = (int)t1.w;
            w_voff = t2.x;
        }
        H_PHASE_BARRIER()                 // every wave is done with its staging region (and its parked state): A'(1) / W'(1) may land
    }
#endif
}

// ---------------------------------------------------------------------------------------
template <int EPI, int NWN, int WF, int OFMT, int WV = 8>
static int launch_h8(const GemmDesc& d, hipStream_t s) {
    constexpr int T_BM = 32 * WV, T_BN = 16 * WF * NWN;
    constexpr int LDS = 3 * T_BM * 128 + 2 * T_BN * 128;
    static DeviceOnce attr;
    VTGB_FUNC_LDS_ONCE(attr, (conv_h8_kernel<EPI, NWN, WF, OFMT, WV>), LDS);
    const int m_tiles = (d.M + T_BM - 1) / T_BM, n_tiles = (d.N + T_BN - 1) / T_BN, total = m_tiles * n_tiles;
    const int slots = cu_count() * (WV == 4 ? 2 : 1);      // persistent workgroups: what fits the CUs at once
    const int grid = total < slots ? total : slots;
    const double exec_flops = 2.0 * d.M * d.N * d.K;      // in fp16-MFMA units: an fp8 k-tile takes the matrix-pipe cycles of an fp16 one
    ProfScope prof(VTGB_PROF_CONV, d.algo_flops > 0 ? d.algo_flops : d.algo_flops < 0 ? 0.0 : exec_flops, s, exec_flops);
    hipLaunchKernelGGL((conv_h8_kernel<EPI, NWN, WF, OFMT, WV>), dim3(grid), dim3(64 * WV), LDS, s, d, m_tiles, n_tiles, 8, total);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// Convolution over f16c8 operands (GemmDesc::h8_run > 0).  epi: EPI_SPLIT (pair store; d.h8_out_bf16: as a bf16 pair), EPI_X3ZR, EPI_X3Q, EPI_STORE_F32
// (N <= 128: fp32 rows + optional per-tile column moments).
int launch_conv_h8(const GemmDesc& d_in, hipStream_t s) {
    GemmDesc d = d_in;
    VTGB_REQUIRE(d.conv_KH > 0 && d.conv_H > 0 && d.conv_W > 0 && d.A && d.W && d.out && d.M > 0 && d.N > 0 && d.zero_page, VTGB_EINVAL, "conv h8: bad argument");
    {
        auto magic = [](uint32_t dv, uint32_t* mul, uint32_t* sh) {
            uint32_t q = 0;
            while ((1ull << q) < dv) q++;
            *mul = (uint32_t)((((1ull << q) - dv) << 32) / dv + 1);
            *sh = q;
        };
        magic((uint32_t)(d.conv_H * d.conv_W), &d.div_hw_mul, &d.div_hw_sh);
        magic((uint32_t)d.conv_W, &d.div_w_mul, &d.div_w_sh);
    }
    const int nk = d.K / H_BK;
    VTGB_REQUIRE(d.dtype == VTGB_BF16 && (d.K % H_BK) == 0 && d.K == d.conv_KH * d.conv_KW * d.conv_Cin && (d.conv_Cin % 128) == 0 && (d.conv_split % 128) == 0 &&
                     (d.conv_split == d.conv_Cin || (d.A2 && d.conv_split * 2 == d.conv_Cin)) && (d.M % (d.conv_H * d.conv_W)) == 0 && d.conv_stride <= 1 && d.conv_Hi == 0 &&
                     d.conv_Wi == 0 && d.conv_wrap == 0 && d.conv_wrap2 == 0,
                 VTGB_EINVAL, "conv h8: inconsistent geometry (sources of equal width, stride 1)");
    const int src_c = d.conv_split / 2;      // channels per source; its fp16 half = src_c / 64 chunks
    VTGB_REQUIRE(d.h8_run == d.conv_KH * d.conv_KW * (src_c / 64) && d.h8_run >= 3 && nk >= 6 && nk % (2 * d.h8_run) == 0, VTGB_EINVAL,
                 "conv h8: run %d does not match taps x channels / 64 = %d (K = %d)", d.h8_run, d.conv_KH * d.conv_KW * (src_c / 64), d.K);
    VTGB_REQUIRE(d.h8_scale != nullptr, VTGB_EINVAL, "conv h8: no weight scale");
    VTGB_REQUIRE((d.lda % 8) == 0 && (d.lda2 % 8) == 0 && (d.ldw % 64) == 0 && d.conv_KH <= 7 && d.conv_KW <= 7, VTGB_EUNSUPPORTED, "conv h8: operand alignment");
    {
        const int64_t hw = (int64_t)d.conv_H * d.conv_W, span = (512 / hw + 2) * hw, ld = d.lda > d.lda2 ? d.lda : d.lda2;
        VTGB_REQUIRE(span < (1 << 24) && span * ld * 2 < 0x7FFFFF00ll && (int64_t)256 * d.ldw * 2 < 0x7FFFFF00ll, VTGB_EUNSUPPORTED, "conv h8: tile footprint beyond the 32-bit descriptor offsets");
    }
    VTGB_REQUIRE(d.o_map.seg_rows == 0 && d.r_map.seg_rows == 0 && d.out_scale == 0.f, VTGB_EUNSUPPORTED, "conv h8: identity row maps only");
    // 256-wide outputs on FEW m-tiles (a single clip: 291 m-tiles = 1.14 rounds of 256 CUs, paid as 2 on the 256-wide tile): two 128-wide n-tiles per
    // m-tile instead -- 582 tiles = 2.27 rounds of half the work each (paid as 1.5).  Cost model in units of one 256 x 128 tile's work; the 128-wide
    // tile is ~10 % dearer per column at the bench batch (z | r 3.57 vs 3.21 ms, flow head 2.84 vs 2.55: same-box A/B, round 6), so large batches keep the
    // 256-wide tile.
    const int cus = cu_count(), m_tiles = (d.M + 255) / 256;
    const bool narrow = ((2 * m_tiles + cus - 1) / cus) * 1.1 < ((m_tiles + cus - 1) / cus) * 2.0;
    switch (d.epi) {
        case EPI_SPLIT:
            VTGB_REQUIRE((d.N & 1) == 0 && (d.ldo & 3) == 0 && (d.split_lo & 3) == 0 && d.split_lo > 0, VTGB_EINVAL, "conv h8: pair store needs 4-aligned rows and split_lo");
            VTGB_REQUIRE(!d.resid_bf16 || (d.N <= 128 && (d.N & 3) == 0 && (d.ldrb & 3) == 0), VTGB_EUNSUPPORTED, "conv h8: the residual tail exists on the 64- and 128-wide tiles only");
            if (d.tail_w) {      // FlowHead: the 1x1 tail on the accumulators, nothing else stored
                VTGB_REQUIRE(d.N == 256 && d.tail_out && (d.ldtail & 3) == 0 && d.ldtail >= 32 && !d.resid_bf16, VTGB_EINVAL,
                             "conv h8: the fused 1x1 tail needs a 256-channel convolution and a 4-aligned fp32 output of >= 32 columns");
                return launch_h8<EPI_FTAIL, 4, 4, 1>(d, s);
            }
            if (d.h8_out_bf16) {
                VTGB_REQUIRE((d.N & 3) == 0, VTGB_EUNSUPPORTED, "conv h8: bf16-pair output needs N %% 4 == 0");
                if (d.N <= 64) return launch_h8<EPI_SPLIT, 1, 4, 0>(d, s);
                if (d.N <= 128 || (narrow && d.N > 192)) return launch_h8<EPI_SPLIT, 2, 4, 0>(d, s);
                return launch_h8<EPI_SPLIT, 4, 4, 0>(d, s);
            }
            if (d.N <= 64) return launch_h8<EPI_SPLIT, 1, 4, 1>(d, s);      // 256 x 64 tile: eight waves of 32 rows
            if (d.N <= 128) return launch_h8<EPI_SPLIT, 2, 4, 1>(d, s);
            if (d.N <= 192 && (d.N & 7) == 0) return launch_h8<EPI_SPLIT, 4, 3, 1>(d, s);
            if (narrow) return launch_h8<EPI_SPLIT, 2, 4, 1>(d, s);
            return launch_h8<EPI_SPLIT, 4, 4, 1>(d, s);
        case EPI_STORE_F32:
            VTGB_REQUIRE(d.N <= 128 && (d.N & 3) == 0 && (d.ldo & 3) == 0 && d.act == 0 && (!d.col_stats || d.stats_rows >= 256), VTGB_EUNSUPPORTED,
                         "conv h8: fp32 outputs on the 64- and 128-wide tiles only (the encoders' residual blocks)");
            if (d.N <= 64) return launch_h8<EPI_STORE_F32, 1, 4, 1>(d, s);
            return launch_h8<EPI_STORE_F32, 2, 4, 1>(d, s);
        case EPI_X3ZR:
            VTGB_REQUIRE(d.N == 256 && d.resid && d.aux && d.out2 && ((d.ldr | d.ldaux | d.ldo | d.ldo2 | d.split_lo) & 3) == 0 && d.act == 0 && !d.bias, VTGB_EINVAL,
                         "conv h8: the z | r gate epilogue needs a 256-channel convolution with its start map, h and both outputs");
            if (narrow) return launch_h8<EPI_X3ZR, 2, 4, 1>(d, s);      // (n-tile 0 = the z columns, n-tile 1 = the r columns)
            return launch_h8<EPI_X3ZR, 4, 4, 1>(d, s);
        case EPI_X3Q:
            VTGB_REQUIRE(d.N == 128 && d.resid && d.aux && ((d.ldr | d.ldaux | d.ldo | d.split_lo) & 3) == 0 && d.act == 0 && !d.bias, VTGB_EINVAL,
                         "conv h8: the GRU update epilogue needs a 128-channel convolution with its start map and z");
#ifndef H8_Q4
#define H8_Q4 0      // (measured, round 6: 128 x 128 tiles with two workgroups per CU -- 2.13 ms against 1.94 for the 256 x 128 tile, same box: the second
                     // workgroup hides the epilogue, but half-height tiles read every weight k-tile twice as often and synchronise twice as often)
#endif
            if (H8_Q4) return launch_h8<EPI_X3Q, 2, 4, 1, 4>(d, s);
            return launch_h8<EPI_X3Q, 2, 4, 1>(d, s);
    }
    vtgb_set_error("conv h8: unsupported epilogue %d", d.epi);
    return VTGB_EINVAL;
}
