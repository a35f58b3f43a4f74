// gemm_pp.hip -- the round-3 form of the large bf16 MFMA kernel (plain GEMM and implicit-GEMM convolution): PERSISTENT workgroups
// with a PING-PONG k-loop.  Same tile (256 rows x 64 NWN columns x 64 deep), same LDS image (LDS-DMA staging, XOR-swizzled 128-byte
// rows), same accumulator layout and the same epilogue arithmetic as gemm_bf16_large_kernel in gemm.hip (which stays for A/B runs and
// for the timing ablations: VTGB_GEMM_OLD=1); what changed is the structure around them:
//
//  * Ping-pong k-loop.  The rounds-1/2 loop ran both waves of a SIMD in lockstep (one barrier per k-tile, two fragment sets): both
//    issued LDS-DMA pieces, both waited on LDS reads, both multiplied at the same moments.  Here waves 4-7 run ONE BARRIER BEHIND
//    waves 0-3 and every wave alternates a COMPUTE segment (the 4 NX MFMAs of one 32-deep half, nothing else) with a LOAD segment
//    (the next half's 4 + NX ds_read_b128 and this wave's share of the LDS-DMA pieces), so that on every SIMD one wave's MFMAs run
//    beside its partner's LDS / DMA traffic.  One fragment set (48 VGPRs fewer at NWN = 4), four barriers per k-tile.
//    A(t+2) is issued in the segment after half 0 of tile t (its slot held tile t-1), W(t+2) in the segment after half 1 (the slot of
//    tile t); one counted wait per k-tile (all but the youngest activation pieces), placed where BOTH groups pass it before the first
//    read of tile t+1.  Standalone (tools/exp/pp_gemm.hip, same box, random data): +3 ... +6.5 % over the lockstep loop.
//  * Persistent workgroups.  Rounds 1-2 launched one workgroup per tile: every tile paid its own prologue -- first operands' HBM
//    latency plus the skew until the slowest wave's pieces land: 5-7 us of a 43-54 us ViT tile, 15 us of the average 27 us RAFT
//    convolution tile (profiles/r02_exp_prologue_probe_after.log) -- with the matrix pipe idle.  Here a workgroup walks the tile list
//    (stride = grid size, all its tiles on one XCD) and issues the NEXT tile's first k-tile (A'(0), W'(0)) at the start of the current
//    tile's epilogue -- behind the epilogue's operand loads, in front of its stores; every epilogue's first wait is a counted one that
//    leaves exactly those pieces in flight -- so they land while the accumulators are stored.  The epilogues stage through A slots
//    1.. only (8 KiB per wave, more passes) so that slot 0 of both rings is free for them.
//
// Requires K % 64 == 0 and descriptor-addressable operands (launch_* in gemm.hip check).
#include "common.h"
#include "gemm_dev.h"
#include "pair_h8.h"

constexpr int P_BK = 64;
constexpr int P_AOP = 256 * P_BK * 2;   // 32 KiB activation slot (256 rows)


// Buffer STORES keep the row offset in the per-lane offset register, not in the scalar offset: with a scalar-register soffset hipcc
// (ROCm 7.2) places NO wait state between a > 8-byte store and the next instruction that overwrites its data registers (its hazard
// table exempts MUBUF stores with an SGPR soffset), and on gfx950 that exemption does not hold: single components of the 16-byte
// h' stores of the GRU epilogue came out as the NEXT value written to the register (0.0), different elements on every run
// (tools/exp/conv_unit.hip; round 3).  Loads are unaffected (their registers are scoreboarded).
#define P_STORE128(data, rs, lane_off, soff) __builtin_amdgcn_raw_buffer_store_b128(data, rs, (lane_off) + (unsigned)(soff), 0, 0)
#define P_STORE64(data, rs, lane_off, soff) __builtin_amdgcn_raw_buffer_store_b64(data, rs, (lane_off) + (unsigned)(soff), 0, 0)

// WF = 16-column weight fragments per wave: 4 (64-column wave tiles: every instantiation but one) or 3 -- the 256 x 192 tile of RAFT's convc2
// (3x3, 256 -> 192 channels; update.py:79): on the 256-wide tile two of its eight waves held only padding, and since waves w and w + 4
// share a SIMD two SIMDs carried twice the MFMA work of the other two; with 48-column wave tiles all eight waves multiply (round 4).
// LNF: the LayerNorm-folded forms of the plain GEMMs (GemmDesc::ln_*) are their own instantiations -- the extra epilogue state (column sums, row
// statistics / the bf16 copy and the row moments) pushed the shared ones past 256 registers
template <int EPI, bool CONV, int NWN, bool TAIL = false, bool PING = !CONV, int WF = 4, bool LNF = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp_kernel(const GemmDesc p, const int m_tiles, const int n_tiles, const int G, const int total_blocks) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NX = 2 * NWN;          // activation fragments per wave: wave tile = (16 NX) x 64
    constexpr int WROWS = 16 * NX;
    constexpr int MW = 8 / NWN;          // waves along M
    constexpr int WC = 16 * WF;          // columns per wave
    constexpr int T_BM = 256, T_BN = WC * NWN;
    constexpr int A_OP = P_AOP, W_OP = T_BN * 128;
    constexpr int AI = 4, WI = T_BN / 64;      // LDS-DMA instructions per wave and k-tile
    static_assert(WF == 4 || (WF == 3 && NWN == 4 && (EPI == EPI_STORE || EPI == EPI_SPLIT) && CONV && !TAIL && !PING), "48-column wave tiles: the plain bf16-store / pair-store convolution only");
    constexpr int A_SLOTS = 3;
    constexpr bool WHOLE = PING && NWN < 4;   // ping-pong on narrow tiles: one compute segment per k-tile (both halves), three weight slots
    constexpr int W_SLOTS = WHOLE ? 3 : 2;
    constexpr int SB = (A_SLOTS - 1) * A_OP / 8;   // epilogue staging bytes per wave (A slots 1, 2): 8 KiB
    constexpr int NPRE = AI + WI;        // pieces of the next tile's first k-tile, in flight under the epilogue
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const smem_w = smem + A_SLOTS * A_OP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % MW, wn = wave / MW;
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = p.K / P_BK;
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(p.W);
    char* const stage = smem + A_OP + wave * SB;   // this wave's private epilogue staging region

    // ---- workgroup -> tile list.  The tiles are numbered q = 0 .. m_tiles * n_tiles - 1 in grouped order (G consecutive m-tiles x all
    // n-tiles, m fastest: neighbours in q share a weight tile and, for convolutions, halo rows).  With the grid a multiple of 8,
    // workgroup w (hardware XCD w & 7) takes q = j * grid + (grid / 8) * (w & 7) + (w >> 3), j = 0, 1, ...: in every round an XCD works
    // on grid / 8 CONSECUTIVE q's (8 m-tiles x 4 n-tiles at 256 workgroups: 12 operand tiles per 32 output tiles through its L2), and
    // every workgroup gets the same number of tiles +- 1 whatever m_tiles is.  (Rounds 1-2 strided a sparse logical grid whose
    // invalid slots fell on the same workgroups every time: at 26 m-tiles -- the LLM prefill -- 104 of the 256 workgroups did all
    // the work, 12 tiles each instead of 5.)
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

    // ---- per-tile LDS-DMA state (see gemm.hip for the addressing scheme: tile descriptors + loop-invariant lane offsets)
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
    const int cv_hw = CONV ? p.conv_H * p.conv_W : 1, cv_Hi = CONV ? (p.conv_Hi ? p.conv_Hi : p.conv_H) : 1,
              cv_Wi = CONV ? (p.conv_Wi ? p.conv_Wi : p.conv_W) : 1, cv_st = CONV ? (p.conv_stride ? p.conv_stride : 1) : 1;
    int cv_ky = 0, cv_kx = 0, cv_c0 = 0;   // CONV: running (channel chunk, tap) of the next A k-tile to stage (K runs chunk-major, tap-minor)
#define P_TILE_SETUP()                                                                                                       \
    {                                                                                                                        \
        m0 = mt * T_BM; n0 = nt * T_BN;                                                                                      \
        wave_active = (n0 + wn * WC < p.N) && (m0 + wm * WROWS < p.M);                                                       \
        /* weight rows beyond N lie outside the descriptor's range: their LDS rows read as zeros (columns that are never stored) */ \
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W + (int64_t)n0 * p.ldw), 0, (int)(min(T_BN, p.N - n0) * p.ldw * 2), 0x00020000); \
        {                                                                                                                    \
            const int row = wave * (8 * WI) + (lane >> 3), slot = lane & 7, c = slot ^ ((row >> 1) & 7);   /* (piece i: row + 8 i, same swizzle term only if ... see P_ISSUE_W) */ \
            w_voff = (unsigned)(row * (int)p.ldw + c * 8) * 2u;                                                              \
        }                                                                                                                    \
        const int cv_img0 = CONV ? m0 / cv_hw : 0;                                                                           \
        const int64_t a_row0 = CONV ? (int64_t)cv_img0 * (cv_Hi * cv_Wi) : map_row(p.a_map, m0);                             \
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A + a_row0 * p.lda), 0, RANGE, 0x00020000);           \
        a2_rsrc = __builtin_amdgcn_make_buffer_rsrc(                                                                         \
            const_cast<bf16_t*>(reinterpret_cast<const bf16_t*>(CONV && p.A2 ? p.A2 : p.A) + a_row0 * (CONV && p.A2 ? p.lda2 : p.lda)), 0, RANGE, 0x00020000); \
        _Pragma("unroll") for (int i = 0; i < AI; i++) {                                                                     \
            const int row = wave * (8 * AI) + i * 8 + (lane >> 3), slot = lane & 7, c = slot ^ ((row >> 1) & 7);             \
            const int am = (m0 + row) < p.M ? (m0 + row) : p.M - 1;                                                          \
            if constexpr (CONV) {                                                                                            \
                const int img = (int)((__umulhi((unsigned)am, p.div_hw_mul) + (unsigned)am) >> p.div_hw_sh), rem = am - img * cv_hw; \
                const int oy = (int)((__umulhi((unsigned)rem, p.div_w_mul) + (unsigned)rem) >> p.div_w_sh), y = oy * cv_st, x = (rem - oy * p.conv_W) * cv_st; \
                a_voff[i] = (unsigned)((img - cv_img0) * (cv_Hi * cv_Wi) + y * cv_Wi + x) | ((unsigned)c << 28);             \
                const int py = p.conv_KH >> 1, px = p.conv_KW >> 1;                                                          \
                const int ylo = max(0, py - y), yhi = min(p.conv_KH - 1, cv_Hi - 1 - y + py), xlo = max(0, px - x), xhi = min(p.conv_KW - 1, cv_Wi - 1 - x + px); \
                const int yb = yhi >= ylo ? ((2 << yhi) - 1) & ~((1 << ylo) - 1) : 0, xb = xhi >= xlo ? ((2 << xhi) - 1) & ~((1 << xlo) - 1) : 0; \
                a_bits[i] = yb | (xb << 8);                                                                                  \
            } else {                                                                                                         \
                a_voff[i] = (unsigned)((int)(map_row(p.a_map, am) - a_row0) * (int)p.lda + c * 8) * 2u;                      \
                a_bits[i] = 0;                                                                                               \
            }                                                                                                                \
        }                                                                                                                    \
        cv_ky = 0; cv_kx = 0; cv_c0 = 0;                                                                                     \
    }
#define P_ISSUE_A(slot, k0)                                                                             \
    if constexpr (CONV) {                                                                               \
        const bool first = cv_c0 < p.conv_split;                                                        \
        const unsigned ldb = (unsigned)(first ? p.lda : p.lda2) * 2u;                                   \
        const int cs_ = first ? cv_c0 : cv_c0 - p.conv_split, wr_ = first ? p.conv_wrap : p.conv_wrap2;   /* (bf16x3 pairs: the third block is hi again) */ \
        const int cc2 = ((wr_ > 0 && cs_ >= wr_) ? cs_ - wr_ : cs_) * 2;                                \
        const int dpix = (cv_ky - (p.conv_KH >> 1)) * cv_Wi + (cv_kx - (p.conv_KW >> 1));               \
        const int need = (1 << cv_ky) | (256 << cv_kx);                                                 \
        /* r4: one descriptor select per k-tile (was a branch + four s_cselect per piece), the offset as a 24-bit multiply-add with a   \
           select (hipcc had made the ?: a divergent branch around a quarter-rate v_mad_u64_u32 per piece) */                           \
        const auto rs_ = first ? a_rsrc : a2_rsrc;                                                      \
        _Pragma("unroll") for (int i = 0; i < AI; i++) {                                                \
            const unsigned pix = (a_voff[i] & 0x00FFFFFFu) + (unsigned)dpix;                            \
            const unsigned vin = __umul24(pix, ldb) + (a_voff[i] >> 28) * 16u;                          \
            const unsigned v = ((a_bits[i] & need) == need) ? vin : OOB;                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lptr_t)(smem + (slot) * A_OP + (wave * (8 * AI) + i * 8) * 128), 16, v, cc2, 0, 0); \
        }                                                                                               \
        if (++cv_kx == p.conv_KW) { cv_kx = 0; if (++cv_ky == p.conv_KH) { cv_ky = 0; cv_c0 += P_BK; } } \
    } else {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < AI; i++)                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)(smem + (slot) * A_OP + (wave * (8 * AI) + i * 8) * 128), 16, a_voff[i], (k0) * 2, 0, 0); \
    }
/* piece i stages rows r + 8 i: (row >> 1) & 7 advances by 4 i, so its swizzled chunk is c ^ 4 for odd i: byte offset ^ 64 (16-byte chunks) */ \
#define P_ISSUE_W(slot, k0)                                                                             \
    _Pragma("unroll") for (int i = 0; i < WI; i++)                                                      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lptr_t)(smem_w + (slot) * W_OP + (wave * (8 * WI) + i * 8) * 128), 16, w_voff ^ ((i & 1) * 64), (k0) * 2 + i * 8 * (int)p.ldw * 2, 0, 0);

    // fragment byte offsets inside an operand tile: fragment i / j of a wave sits 16 rows = 2048 bytes below fragment 0 with the SAME
    // swizzle term ((row >> 1) & 7 does not see multiples of 16), and the second 32-deep half is chunk index ^ 4 = byte offset ^ 64:
    // two registers per operand instead of 8 + 2 NX, the rest are ds_read immediates
    constexpr int NH = (WHOLE || !PING) ? 2 : 1;   // fragment sets: one 32-deep half (ping-pong at NWN = 4), else two
    bf16x8 wf[NH][WF], xf[NH][NX];
#define P_READ(H, as_, ws_, ks)                                                                          \
    if (wave_active) {                                                                                   \
        const char* const wp_ = (ws_) + (w_off0 ^ ((ks) * 64));                                          \
        const char* const xp_ = (as_) + (x_off0 ^ ((ks) * 64));                                          \
        _Pragma("unroll") for (int i = 0; i < WF; i++) wf[H][i] = *reinterpret_cast<const bf16x8*>(wp_ + i * 2048); \
        _Pragma("unroll") for (int j = 0; j < NX; j++) xf[H][j] = *reinterpret_cast<const bf16x8*>(xp_ + j * 2048); \
    }
#define P_MFMAS(H)                                                                                       \
        _Pragma("unroll") for (int i = 0; i < WF; i++)                                                   \
            _Pragma("unroll") for (int j = 0; j < NX; j++)                                               \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[H][i], xf[H][j], acc[i][j], 0, 0, 0);
#define P_COMPUTE()                                                                                      \
    if (wave_active) {                                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                   \
        P_MFMAS(0)                                                                                       \
        if constexpr (WHOLE) { P_MFMAS(1) }                                                              \
        __builtin_amdgcn_s_setprio(0);                                                                   \
    }
#define P_SEG_END() __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
// phase barrier: hipcc may otherwise move LDS accesses across a bare s_barrier (it did: the epilogue's staging writes of a fast wave
// landed in the slot a slower wave was still reading its last fragments from -- run-to-run different GRU outputs, found in round 3)
#define P_PHASE_BARRIER() __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);

    int b = next_valid(first_q, mt, nt);
    if (b < 0) return;
    P_TILE_SETUP()
    P_ISSUE_A(0, 0)
    P_ISSUE_W(0, 0)
    while (true) {
        // (lane-derived constants are re-derived per phase from a laundered lane id: carried across the whole loop they -- not the
        // accumulators -- were what hipcc pushed into scratch, and every reload is a serial memory round trip)
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        const int w_off0 = swz(wn * WC + (lane_k & 15), lane_k >> 4), x_off0 = swz(wm * WROWS + (lane_k & 15), lane_k >> 4);
        // ================= accumulator start values, behind the first k-tile's DMAs (bias once per column group; start maps)
        f32x4 acc[WF][NX];
        {
            f32x4 b4[WF];
#pragma unroll
            for (int i = 0; i < WF; i++) {
                const int n = n0 + wn * WC + i * 16 + fg * 4;
                b4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias && n + 3 < p.N) b4[i] = *reinterpret_cast<const f32x4*>(p.bias + n);
                else if (p.bias && n < p.N) {                                                    // (N % 4 != 0: plain bf16 stores only, pp_supported)
                    for (int e = 0; e < 4 && n + e < p.N; e++) b4[i][e] = p.bias[n + e];
                }
            }
            if (p.init_frag) {
                // all NX * 4 loads (512 contiguous bytes per wave instruction) in flight BEFORE the first add: left to itself hipcc
                // consumed each load right behind its issue with `s_waitcnt vmcnt(0)` (a DMA is in flight: no counted waits), i.e.
                // 32 serial memory round trips per tile -- 12 us of a 63 us GRU z|r tile
                const bf16x4* fsrc = reinterpret_cast<const bf16x4*>(p.init_bf16) + ((int64_t)(mt * n_tiles + nt) * 8 + wave) * (NX * 4 * 64) + lane;
                bf16x4 tq[NX * 4];
#pragma unroll
                for (int q = 0; q < NX * 4; q++) tq[q] = fsrc[q * 64];
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NX; j++)
#pragma unroll
                    for (int i = 0; i < WF; i++) {
                        const bf16x4 t = tq[j * 4 + i];
                        acc[i][j] = b4[i] + f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
                    }
            } else {   // (row-major start maps and early residuals stay with the one-tile-per-workgroup kernel: pp_supported)
#pragma unroll
                for (int j = 0; j < NX; j++)
#pragma unroll
                    for (int i = 0; i < WF; i++) acc[i][j] = b4[i];
            }
        }
        if constexpr (!PING) {
            // ---------- LOCKSTEP k-loop (the rounds-1/2 loop of gemm.hip inside the persistent frame; used for the convolutions): both
            // waves of a SIMD multiply at the same time, one barrier per k-tile, two fragment sets; the LDS-DMA pieces sit BETWEEN the
            // MFMAs (sched_group_barrier), where the partner wave's MFMAs cover their issue time.  The ping-pong loop below puts the
            // pieces (plus, for a convolution, ~6 VALU per piece of tap arithmetic) into the load segment, which then outlasts the
            // partner's compute segment: measured on RAFT's GRU convolutions +23 ... +28 % per launch, so they keep this loop.
            if (nk > 1) { P_ISSUE_A(1, P_BK) P_ISSUE_W(1, P_BK) }
            if (nk > 2) { P_ISSUE_A(2, 2 * P_BK) }
            if (nk > 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * AI + WI));
            else if (nk > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | NPRE);
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            int a_slot = 0;
            if (!wave_active) {
                // same DMA issues, waits and barriers as the active waves, nothing else
                for (int kt = 0; kt + 1 < nk; kt++) {
                    if (kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | AI);
                    else __builtin_amdgcn_s_waitcnt(0x0F70);
                    __builtin_amdgcn_s_barrier();
                    if (kt + 2 < nk) { P_ISSUE_W(kt & 1, (kt + 2) * P_BK) }
                    if (kt + 3 < nk) { P_ISSUE_A(a_slot, (kt + 3) * P_BK) }
                    a_slot = a_slot == 2 ? 0 : a_slot + 1;
                }
            } else {
#define P_LREAD(H, as_, ws_, ks)                                                                         \
    {                                                                                                    \
        const char* const wp_ = (ws_) + (w_off0 ^ ((ks) * 64));                                          \
        const char* const xp_ = (as_) + (x_off0 ^ ((ks) * 64));                                          \
        _Pragma("unroll") for (int i = 0; i < WF; i++) wf[H][i] = *reinterpret_cast<const bf16x8*>(wp_ + i * 2048); \
        _Pragma("unroll") for (int j = 0; j < NX; j++) xf[H][j] = *reinterpret_cast<const bf16x8*>(xp_ + j * 2048); \
    }
#define P_SCHED_IL(PIECES)                                                                               \
    if constexpr ((PIECES) > 0 && (WF * NX) % (PIECES) == 0) {                                           \
        _Pragma("unroll") for (int g_ = 0; g_ < (PIECES); g_++) {                                        \
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);               /* one LDS-DMA piece */     \
            __builtin_amdgcn_sched_group_barrier(0x008, (WF * NX) / (PIECES), 0);   /* its share of the half's MFMAs */ \
        }                                                                                                \
    }
#define P_LITER(DEFER, WCOND, WAIT4)                                                                     \
    {                                                                                                    \
        const char* as = smem + a_slot * A_OP;                                                           \
        const char* ws = smem_w + (kt & 1) * W_OP;                                                       \
        const int a_nxt = a_slot == 2 ? 0 : a_slot + 1;                                                  \
        const int a_prv = a_slot == 0 ? 2 : a_slot - 1;                                                  \
        if (DEFER) { P_ISSUE_A(a_prv, (kt + 2) * P_BK) }                                                 \
        P_LREAD(1, as, ws, 1)                                                                            \
        P_MFMAS(0)                                                                                       \
        P_SCHED_IL(AI)                                                                                   \
        if (WAIT4) __builtin_amdgcn_s_waitcnt(0x0070 | AI);   /* all but A(t+2) landed; lgkmcnt(0) */    \
        else __builtin_amdgcn_s_waitcnt(0x0070);                                                         \
        __builtin_amdgcn_s_barrier();                                                                    \
        if (WCOND) { P_ISSUE_W(kt & 1, (kt + 2) * P_BK) }                                                \
        P_LREAD(0, smem + a_nxt * A_OP, smem_w + ((kt + 1) & 1) * W_OP, 0)                               \
        P_MFMAS(1)                                                                                       \
        P_SCHED_IL(WI)                                                                                   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                              \
        a_slot = a_nxt;                                                                                  \
    }
                P_LREAD(0, smem, smem_w, 0)
                __builtin_amdgcn_s_waitcnt(0xC07F);
                int kt = 0;
                if (nk > 1) { P_LITER(false, kt + 2 < nk, kt + 2 < nk) kt = 1; }
                for (; kt + 2 < nk; kt++) P_LITER(true, true, true)                     // steady state: no conditions
                for (; kt + 1 < nk; kt++) P_LITER(kt + 2 < nk, kt + 2 < nk, kt + 2 < nk)
                {   // last k-tile
                    const char* as = smem + a_slot * A_OP;
                    const char* ws = smem_w + ((nk - 1) & 1) * W_OP;
                    P_LREAD(1, as, ws, 1)
                    P_MFMAS(0)
                    P_MFMAS(1)
                }
#undef P_LITER
#undef P_SCHED_IL
#undef P_LREAD
            }
            P_PHASE_BARRIER()                      // every wave's LDS reads and DMAs of this tile are done: the ring is free
        } else
        if constexpr (!WHOLE) {
            // ---------- 256-wide tile: half-k-tile segments.  A(g) in slot g % 3, W(g) in slot g % 2
            if (nk > 1) {
                P_ISSUE_A(1, P_BK)
                P_ISSUE_W(1, P_BK)
                __builtin_amdgcn_s_waitcnt(0x0F70 | NPRE);   // vmcnt(AI + WI): k-tile 0 (and everything older: the previous tile's stores) done
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (wave >= 4) __builtin_amdgcn_s_barrier();     // the second group runs one barrier behind the first
            __builtin_amdgcn_sched_barrier(0);
            P_READ(0, smem, smem_w, 0)
            __builtin_amdgcn_s_waitcnt(0xC07F);
            P_SEG_END()
            int a_slot = 0;
            for (int kt = 0; kt < nk; kt++) {
                const char* as = smem + a_slot * A_OP;
                const char* ws = smem_w + (kt & 1) * W_OP;
                const int a_nxt = a_slot == A_SLOTS - 1 ? 0 : a_slot + 1;
                const int a_prv = a_slot == 0 ? A_SLOTS - 1 : a_slot - 1;
                // C(t, 0)
                P_COMPUTE()
                P_SEG_END()
                // L_a(t): A(t+2) into the slot of tile t-1; fragments of half 1; counted wait: tile t+1 has landed
                if (kt + 2 < nk) { P_ISSUE_A(a_prv, (kt + 2) * P_BK) }
                P_READ(0, as, ws, 1)
                if (kt + 2 < nk) __builtin_amdgcn_s_waitcnt(0x0070 | AI);   // vmcnt(AI) lgkmcnt(0)
                else __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) lgkmcnt(0)
                P_SEG_END()
                // C(t, 1)
                P_COMPUTE()
                P_SEG_END()
                // L_b(t): W(t+2) into the slot of tile t; fragments of half 0 of tile t+1
                if (kt + 2 < nk) { P_ISSUE_W(kt & 1, (kt + 2) * P_BK) }
                if (kt + 1 < nk) { P_READ(0, smem + a_nxt * A_OP, smem_w + ((kt + 1) & 1) * W_OP, 0) }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                P_SEG_END()
                a_slot = a_nxt;
            }
        } else {
            // ---------- narrow tiles (16 MFMAs per half would leave the barriers as long as the segments): ONE compute segment per
            // k-tile (both halves, two fragment sets) and one load segment that re-arms the slots of the tile just multiplied with
            // tile t+3 and reads tile t+1.  A(g), W(g) in slot g % 3: three k-tiles of lookahead, two barriers per k-tile.
            if (nk > 1) { P_ISSUE_A(1, P_BK) P_ISSUE_W(1, P_BK) }
            if (nk > 2) { P_ISSUE_A(2, 2 * P_BK) P_ISSUE_W(2, 2 * P_BK) }
            if (nk > 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * NPRE));
            else if (nk > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | NPRE);
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (wave >= 4) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            P_READ(0, smem, smem_w, 0)
            P_READ(1, smem, smem_w, 1)
            if (nk > 2) __builtin_amdgcn_s_waitcnt(0x0070 | NPRE);   // k-tile 1 has landed (k-tile 2 may fly); lgkmcnt(0)
            else __builtin_amdgcn_s_waitcnt(0x0070);
            P_SEG_END()
            int a_slot = 0;
            for (int kt = 0; kt < nk; kt++) {
                const int a_nxt = a_slot == 2 ? 0 : a_slot + 1;
                P_COMPUTE()
                P_SEG_END()
                if (kt + 3 < nk) {
                    P_ISSUE_A(a_slot, (kt + 3) * P_BK)
                    P_ISSUE_W(a_slot, (kt + 3) * P_BK)
                }
                if (kt + 1 < nk) {
                    P_READ(0, smem + a_nxt * A_OP, smem_w + a_nxt * W_OP, 0)
                    P_READ(1, smem + a_nxt * A_OP, smem_w + a_nxt * W_OP, 1)
                }
                if (kt + 3 < nk) __builtin_amdgcn_s_waitcnt(0x0070 | NPRE);   // k-tile t+2 has landed (t+3 may fly); lgkmcnt(0)
                else __builtin_amdgcn_s_waitcnt(0x0070);
                P_SEG_END()
                a_slot = a_nxt;
            }
        }
        if constexpr (PING) {
            __builtin_amdgcn_sched_barrier(0);
            if (wave < 4) __builtin_amdgcn_s_barrier();      // every wave's LDS reads and DMAs of this tile are done: the ring is free
            __builtin_amdgcn_sched_barrier(0);
        }

        // ================= the next tile: its setup replaces this tile's DMA state (dead), its first k-tile is issued by the epilogue
        const int em0 = m0, en0 = n0, emt = mt, ent = nt;
        const bool eactive = wave_active;
        const int b2 = next_valid(b + (int)gridDim.x, mt, nt);
        const bool has_next = b2 >= 0;
        if (has_next) P_TILE_SETUP()
        // NWN = 4 (128 accumulator registers): once the next tile's first k-tile is issued, its per-lane DMA state (AI offsets, AI tap
        // masks, WI weight offsets) is parked in W slot 1 -- free during every epilogue -- and read back below: 12 registers the
        // epilogue's staging arrays would otherwise push into scratch
        constexpr bool PARK = NWN == 4;
        uint4* const park = reinterpret_cast<uint4*>(smem_w + W_OP) + tid;
        auto prefetch = [&]() {
            if (has_next) {
                P_ISSUE_A(0, 0)
                P_ISSUE_W(0, 0)
                if constexpr (PARK) {
                    park[0] = make_uint4(a_voff[0], a_voff[1], a_voff[2], a_voff[3]);
                    park[512] = make_uint4((unsigned)a_bits[0], (unsigned)a_bits[1], (unsigned)a_bits[2], (unsigned)a_bits[3]);
                    park[1024] = make_uint4(w_voff, 0u, 0u, 0u);
                }
            }
        };
#define P_WAIT_OPERANDS()   /* the epilogue's first operand loads are done; the NPRE pieces issued behind them may still fly */ \
    if (has_next) __builtin_amdgcn_s_waitcnt(0x0F70 | NPRE); else __builtin_amdgcn_s_waitcnt(0x0F70);

        // ================= epilogue of tile (em0, en0)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        // Every global access of the epilogues goes through a buffer descriptor on the TILE's rows of the matrix (rows beyond M fall
        // outside its range: loads return 0, stores are dropped by the hardware) with ONE per-lane_e byte offset per matrix (columns
        // beyond N: an out-of-range offset) and the row inside the tile in the scalar offset: no per-row predicates, branches or 64-bit
        // per-lane_e address arithmetic -- the round-2 pointer form of these loops kept ~2 address registers per row alive and, inside
        // the persistent loop, spilled 100-350 registers per tile.  (launch_large_pp takes only identity output / residual row maps.)
        const int rows_here = min(T_BM, p.M - em0);
        auto tile_rsrc = [&](const void* base, int64_t ld, int es) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + (int64_t)em0 * ld * es, 0, (int)(rows_here * ld * es), 0x00020000);
        };
        typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
        typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2_t;
        typedef __attribute__((__vector_size__(3 * sizeof(unsigned)))) unsigned u32x3_t;
        if constexpr (EPI == EPI_STORE && !TAIL) {
            if (p.frag_out) {   // fragment order: straight from the accumulators, 512 contiguous bytes per wave instruction, no LDS
                prefetch();
                if (eactive) {
                    bf16x4* fdst = reinterpret_cast<bf16x4*>(p.out) + ((int64_t)(emt * n_tiles + ent) * 8 + wave) * (NX * 4 * 64) + lane_e;
#pragma unroll
                    for (int j = 0; j < NX; j++)
#pragma unroll
                        for (int i = 0; i < WF; i++) {
                            f32x4 v = acc[i][j];
                            if (p.act) apply_act4(v, p.act);
                            fdst[(j * 4 + i) * 64] = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                        }
                }
                goto tile_done;
            }
        }
        if constexpr (EPI == EPI_STORE && NWN == 4 && TAIL) {
            {
                // Fused 1x1 tail (RAFT's FlowHead): T[256 px][32] = relu(tile)[256 px][256 ch] . tail_w[32][256]^T.  The activated tile
                // takes the whole ring (16 KiB per wave from offset 0), so the next tile's first k-tile is issued AFTER the tail's
                // LDS reads (no overlap here; one launch of eleven per refinement iteration).
                const bf16_t* const tw = reinterpret_cast<const bf16_t*>(p.tail_w);
                char* const cst = smem + wave * (WROWS * 128);
#pragma unroll
                for (int j = 0; j < NX; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        f32x4 v = acc[i][j];
                        if (p.act) apply_act4(v, p.act);
                        const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                        const int row = j * 16 + (lane_e & 15), c16 = (i * 2 + ((lane_e >> 4) >> 1)) ^ (row & 7);
                        *reinterpret_cast<bf16x4*>(cst + row * 128 + c16 * 16 + ((lane_e >> 4) & 1) * 8) = pk;
                    }
                bf16x8 twf[2][8];
#pragma unroll
                for (int nb = 0; nb < 2; nb++)
#pragma unroll
                    for (int ks = 0; ks < 8; ks++) twf[nb][ks] = *reinterpret_cast<const bf16x8*>(tw + (nb * 16 + (lane_e & 15)) * 256 + ks * 32 + (lane_e >> 4) * 8);
                P_PHASE_BARRIER()
                const auto t_rs = tile_rsrc(p.tail_out, p.ldtail, 4);
#pragma unroll
                for (int pb = 0; pb < 2; pb++) {
                    const int blk = wn * 2 + pb, row = blk * 16 + (lane_e & 15);
                    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
#pragma unroll
                    for (int ks = 0; ks < 8; ks++) {
                        const int k = ks * 32 + (lane_e >> 4) * 8, reg = k >> 6, c16 = (k & 63) >> 3;
                        const bf16x8 xfr = *reinterpret_cast<const bf16x8*>(smem + (wm + MW * reg) * (WROWS * 128) + row * 128 + ((c16 ^ (row & 7)) << 4));
                        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twf[0][ks], xfr, t0, 0, 0, 0);
                        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(twf[1][ks], xfr, t1, 0, 0, 0);
                    }
                    const unsigned vo = (unsigned)((wm * WROWS + row) * (int)p.ldtail + (lane_e >> 4) * 4) * 4u;      // (rows beyond M: out of range)
                    P_STORE128(__builtin_bit_cast(u32x4_t, t0), t_rs, vo, 0);
                    P_STORE128(__builtin_bit_cast(u32x4_t, t1), t_rs, vo + 64, 0);
                }
                P_PHASE_BARRIER()                      // every wave's reads of the staged tile are done: the ring is free again
                prefetch();
                goto tile_done;
            }
        }
        if constexpr ((EPI == EPI_STORE || EPI == EPI_GELU) && !TAIL) {
            // bf16 outputs through the wave's staging region in passes of RP rows (16-byte chunks XOR-swizzled by row & 7): 16 bytes
            // per lane_e, 8 whole 128-byte row segments per store instruction.  The operands of the fused elementwise tails (r * h gate,
            // ResidualBlock skip) of ALL passes are requested first, then the next tile's first k-tile; one counted wait.
            constexpr int RP = SB / 128 < WROWS ? SB / 128 : WROWS, NP = WROWS / RP, JB = RP / 16;
            const bool gated = !LNF && EPI == EPI_STORE && p.gate_from > 0, resd = !LNF && EPI == EPI_STORE && !gated && p.resid_bf16 != nullptr;
            const int ch0 = lane_e & 7, nn = ch0 * 8 < WC ? en0 + wn * WC + ch0 * 8 : 0x40000000;      // (48-column wave tiles: chunks 6, 7 of a staged row do not exist)
            const bool to_out2 = gated && (en0 + wn * WC) >= p.gate_from;     // wave-uniform (gate_from % 64 == 0): this wave's columns are r -> r * h
            const int col = to_out2 ? nn - p.gate_from : nn;
            const int64_t ld_o = to_out2 ? p.ldo2 : p.ldo;
            const auto o_rs = tile_rsrc(to_out2 ? p.out2 : p.out, ld_o, 2);
            // N % 8 != 0 (RAFT's 126-channel motion convolution, whose last two output columns hold the flow written by another kernel):
            // the chunk that straddles N is stored as its N % 8 valid columns (1-3 dwords) by a second, wave-uniformly guarded store
            const int ncut = p.N & 7;
            const bool whole = nn + 8 <= p.N || (ncut == 0 && nn < p.N);
            const unsigned o_lane = whole ? (unsigned)((wm * WROWS + (lane_e >> 3)) * (int)ld_o + col) * 2u : OOB;
            const unsigned o_part = (ncut != 0 && nn < p.N && nn + 8 > p.N) ? (unsigned)((wm * WROWS + (lane_e >> 3)) * (int)ld_o + col) * 2u : OOB;
            u32x4_t opnd[RP / 8];
            const int64_t ld_g = to_out2 ? p.ldaux : p.ldrb;
            const bool has_opnd = to_out2 || resd;
            const auto g_rs = tile_rsrc(to_out2 ? p.aux : (resd ? p.resid_bf16 : p.out), has_opnd ? ld_g : p.ldo, 2);
            const unsigned g_lane = nn < p.N ? (unsigned)((wm * WROWS + (lane_e >> 3)) * (int)ld_g + col) * 2u : OOB;
#define P_OPND_LOAD(h) _Pragma("unroll") for (int rr = 0; rr < RP / 8; rr++) opnd[rr] = __builtin_amdgcn_raw_buffer_load_b128(g_rs, g_lane, ((h) * RP + rr * 8) * (int)ld_g * 2, 0);
            if (has_opnd) { P_OPND_LOAD(0) }
            // LayerNorm folded into this GEMM (GemmDesc::ln_stats; plain GEMMs): the lane's column sums / constants and its rows' (mean, rstd),
            // requested in front of the next tile's pieces like every other epilogue operand
            constexpr bool lnf = LNF && !CONV;
            f32x4 ln_cs4[WF], ln_c4[WF];
            float ln_m[NX], ln_r[NX];
            if (lnf) {
#pragma unroll
                for (int i = 0; i < WF; i++) {
                    const int nc = min(en0 + wn * WC + i * 16 + (lane_e >> 4) * 4, p.N - 4);      // (columns beyond N are never stored)
                    ln_cs4[i] = *reinterpret_cast<const f32x4*>(p.ln_cs + nc);
                    ln_c4[i] = *reinterpret_cast<const f32x4*>(p.ln_c + nc);
                }
#pragma unroll
                for (int j = 0; j < NX; j++) {
                    const int mr = min(em0 + wm * WROWS + j * 16 + (lane_e & 15), p.M - 1);
                    const float2 st_ = *reinterpret_cast<const float2*>(p.ln_stats + (int64_t)mr * 2);
                    ln_m[j] = st_.x; ln_r[j] = st_.y;
                }
            }
            prefetch();
            if (lnf) { P_WAIT_OPERANDS() }
#pragma unroll
            for (int ps = 0; ps < NP; ps++) {
#pragma unroll
                for (int jj = 0; jj < JB; jj++)
#pragma unroll
                    for (int i = 0; i < WF; i++) {
                        f32x4 v = acc[i][ps * JB + jj];
                        if (lnf) v = ln_fold4(v, ln_m[ps * JB + jj], ln_r[ps * JB + jj], ln_cs4[i], ln_c4[i]);
                        if constexpr (EPI == EPI_GELU) {
gelu_erf_fast4(v);
                        } else if (p.act) apply_act4(v, p.act);
                        const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                        const int row = jj * 16 + (lane_e & 15), c16 = (i * 2 + ((lane_e >> 4) >> 1)) ^ (row & 7);
                        *reinterpret_cast<bf16x4*>(stage + row * 128 + c16 * 16 + ((lane_e >> 4) & 1) * 8) = pk;
                    }
                if (gated || resd) {      // (wave-uniform per launch; waves of a gated launch that hold z columns wait for nothing they issued)
                    if (ps == 0) { P_WAIT_OPERANDS() }
                    else __builtin_amdgcn_s_waitcnt(0x0F70 | (RP / 8));      // this pass's operands; the previous pass's RP / 8 stores may still fly
                }
                u32x4_t vv[RP / 8];
#pragma unroll
                for (int rr = 0; rr < RP / 8; rr++) {
                    const int row = rr * 8 + (lane_e >> 3);
                    vv[rr] = *reinterpret_cast<const u32x4_t*>(stage + row * 128 + ((ch0 ^ (row & 7)) << 4));
                }
                if (has_opnd) {
#pragma unroll
                    for (int rr = 0; rr < RP / 8; rr++) {
                        const bf16x8 a = __builtin_bit_cast(bf16x8, vv[rr]);
                        const bf16x8 g = __builtin_bit_cast(bf16x8, opnd[rr]);
                        bf16x8 o;
                        if (to_out2) {
#pragma unroll
                            for (int e = 0; e < 8; e++) o[e] = (bf16_t)((float)a[e] * (float)g[e]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; e++) {
                                const float t = (float)a[e] + (float)g[e];
                                o[e] = (bf16_t)(p.post_relu ? fmaxf(t, 0.f) : t);
                            }
                        }
                        vv[rr] = __builtin_bit_cast(u32x4_t, o);
                    }
                    if (ps + 1 < NP) { P_OPND_LOAD(ps + 1) }     // the next pass's operands (same registers), requested before this pass's stores
                }
#pragma unroll
                for (int rr = 0; rr < RP / 8; rr++) P_STORE128(vv[rr], o_rs, o_lane, (ps * RP + rr * 8) * (int)ld_o * 2);
                if (ncut != 0 && !has_opnd) {
#pragma unroll
                    for (int rr = 0; rr < RP / 8; rr++) {
                        const unsigned off = o_part + (unsigned)((ps * RP + rr * 8) * (int)ld_o * 2);
                        if (ncut == 6) __builtin_amdgcn_raw_buffer_store_b96(u32x3_t{vv[rr][0], vv[rr][1], vv[rr][2]}, o_rs, off, 0, 0);
                        else if (ncut == 4) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{vv[rr][0], vv[rr][1]}, o_rs, off, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b32(vv[rr][0], o_rs, off, 0, 0);
                    }
                }
            }
#undef P_OPND_LOAD
        }
        if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32) {
            // fp32 outputs: PR rows per pass through the staging region (256-byte rows, 16-byte chunks XOR-swizzled by row & 15), 4
            // whole 256-byte row segments per store instruction.  The fp32 residual rows of pass h+1 are requested before pass h's
            // stores; every wait is counted (the next tile's pieces / the previous pass's stores stay in flight).
            constexpr int PR = SB / 256 < WROWS ? SB / 256 : WROWS, NP = WROWS / PR;
            const int rl = lane_e >> 4, cl = lane_e & 15;
            const bool do_stats = (EPI == EPI_STORE_F32) && p.col_stats != nullptr;
            const int img_a = do_stats ? em0 / p.stats_rows : 0;
            const int m_b = do_stats ? (img_a + 1) * p.stats_rows : 0x7fffffff;
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, qa = sa, sb = sa, qb = sa;
            const int n = en0 + wn * 64 + cl * 4;
            const auto o_rs = tile_rsrc(p.out, p.ldo, 4);
            const unsigned o_lane = n < p.N ? (unsigned)((wm * WROWS + rl) * (int)p.ldo + n) * 4u : OOB;
            const auto r_rs = tile_rsrc(EPI == EPI_RESID_F32 ? (const void*)p.resid : (const void*)p.out, EPI == EPI_RESID_F32 ? p.ldr : p.ldo, 4);
            const unsigned r_lane = n < p.N ? (unsigned)((wm * WROWS + rl) * (int)(EPI == EPI_RESID_F32 ? p.ldr : p.ldo) + n) * 4u : OOB;
            // LayerNorm fold, producer side (GemmDesc::ln_xb): bf16 copy of the rows + per (row, 64-column block) moments
            constexpr bool lnp = LNF && EPI == EPI_RESID_F32 && !CONV;
            const int ln_nblk = (p.N + 63) >> 6;
            const auto xb_rs = tile_rsrc(lnp ? p.ln_xb : p.out, lnp ? p.ldxb : p.ldo, 2);
            const auto pt_rs = tile_rsrc(lnp ? (const void*)p.ln_part : (const void*)p.out, lnp ? ln_nblk * 2 : p.ldo, 4);
            const unsigned xb_lane = (lnp && n < p.N) ? (unsigned)((wm * WROWS + rl) * (int)p.ldxb + n) * 2u : OOB;
            const unsigned pt_lane = (lnp && cl == 0 && n < p.N) ? (unsigned)((wm * WROWS + rl) * ln_nblk * 2 + (n >> 6) * 2) * 4u : OOB;
            u32x4_t rq[PR / 4];
#define P_RESID_LOAD(h)                                                                                             \
    _Pragma("unroll") for (int rr = 0; rr < PR / 4; rr++) rq[rr] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, r_lane, ((h) * PR + rr * 4) * (int)p.ldr * 4, 0);
            if constexpr (EPI == EPI_RESID_F32) { P_RESID_LOAD(0) }
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
                if constexpr (EPI == EPI_RESID_F32) {
                    if (ps == 0) { P_WAIT_OPERANDS() }
                    else if (lnp) __builtin_amdgcn_s_waitcnt(0x4F78);     // (LayerNorm fold: 3 stores per row group -- vmcnt(24))
                    else __builtin_amdgcn_s_waitcnt(0x0F70 | (PR / 4));   // this pass's rows; the previous pass's PR / 4 stores may still fly
                }
                f32x4 vv[PR / 4];
#pragma unroll
                for (int rr = 0; rr < PR / 4; rr++) {
                    const int row = rr * 4 + rl;
                    vv[rr] = *reinterpret_cast<const f32x4*>(stage + row * 256 + ((cl ^ (row & 15)) << 4));
                    if constexpr (EPI == EPI_RESID_F32) vv[rr] += __builtin_bit_cast(f32x4, rq[rr]);
                }
                if constexpr (EPI == EPI_RESID_F32) {
                    if (ps + 1 < NP) { P_RESID_LOAD(ps + 1) }   // the next pass's rows, requested before this pass's stores
                }
#pragma unroll
                for (int rr = 0; rr < PR / 4; rr++) {
                    P_STORE128(__builtin_bit_cast(u32x4_t, vv[rr]), o_rs, o_lane, (ps * PR + rr * 4) * (int)p.ldo * 4);
                    if constexpr (EPI == EPI_RESID_F32) {
                        if (lnp) {
                            const f32x4 v = vv[rr];
                            const bf16x4 xb4 = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                            P_STORE64(__builtin_bit_cast(u32x2_t, xb4), xb_rs, xb_lane, (ps * PR + rr * 4) * (int)p.ldxb * 2);
                            float s_, q_;
                            ln_part4(v, s_, q_);
#pragma unroll
                            for (int off = 1; off < 16; off <<= 1) { s_ += __shfl_xor(s_, off); q_ += __shfl_xor(q_, off); }
                            const u32x2_t sq = {__float_as_uint(s_), __float_as_uint(q_)};
                            P_STORE64(sq, pt_rs, pt_lane, (ps * PR + rr * 4) * ln_nblk * 2 * 4);
                        }
                    }
                    if constexpr (EPI == EPI_STORE_F32) {
                        if (do_stats) {
                            const int m = em0 + wm * WROWS + ps * PR + rr * 4 + rl;
                            if (m < p.M && n < p.N && eactive) {
                                if (m < m_b) { sa += vv[rr]; qa += vv[rr] * vv[rr]; }
                                else { sb += vv[rr]; qb += vv[rr] * vv[rr]; }
                            }
                        }
                    }
                }
            }
#undef P_RESID_LOAD
            if constexpr (EPI == EPI_STORE_F32) {
                if (do_stats) {
                    // lanes with equal cl hold the same four columns: fold the four row groups, park the wave's partials in its staging
                    // region, then 64 NWN threads fold the MW waves of a column and issue one atomic per (image, column, moment)
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
                    P_PHASE_BARRIER()
                    if (tid < 64 * NWN) {                    // (threads of waves 0 .. NWN-1, whether or not their own rows are valid)
                        const int wn_ = tid >> 6, c = tid & 63, nc = en0 + wn_ * 64 + c;
                        if (nc < p.N) {
                            float t[4] = {0.f, 0.f, 0.f, 0.f};
                            for (int wm_ = 0; wm_ < MW; wm_++) {
                                if (em0 + wm_ * WROWS >= p.M) break;
                                const float* pr = reinterpret_cast<const float*>(smem + A_OP + (wn_ * MW + wm_) * SB);
#pragma unroll
                                for (int e = 0; e < 4; e++) t[e] += pr[e * 64 + c];
                            }
                            // r5: stored to the tile's own slot [m-tile][column][4]; stats_finish_tiles adds the slots in tile order (no atomics)
                            *reinterpret_cast<float4*>(p.col_stats + ((int64_t)emt * p.N + nc) * 4) = make_float4(t[0], t[1], t[2], t[3]);
                        }
                    }
                }
            }
        }
        if constexpr (EPI == EPI_SPLIT) {
            // bf16x3 pair store (GemmDesc::split_lo): v = act(acc) goes through the staging region as fp32 (as in the fp32 path), is read back
            // as whole row segments and leaves as hi = bf16(v) at column n and lo = bf16(v - hi) at column n + split_lo: 8 bytes per lane,
            // 128 contiguous bytes per row and half, four rows per store instruction.  N % 4 == 2 (RAFT's 126-channel motion convolution, whose
            // last two columns hold the flow written by another kernel): the group that straddles N is stored as 4 bytes.
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
            prefetch();
#pragma unroll
            for (int ps = 0; ps < NP; ps++) {
#pragma unroll
                for (int jj = 0; jj < PR / 16; jj++)
#pragma unroll
                    for (int i = 0; i < WF; i++) {
                        const int row = jj * 16 + (lane_e & 15), chunk = i * 4 + (lane_e >> 4);
                        *reinterpret_cast<f32x4*>(stage + row * 256 + ((chunk ^ (row & 15)) << 4)) = acc[i][ps * (PR / 16) + jj];
                    }
#pragma unroll
                for (int rr = 0; rr < PR / 4; rr++) {
                    const int row = rr * 4 + rl;
                    f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 256 + ((cl ^ (row & 15)) << 4));
                    if (p.act) apply_act4(v, p.act);
                    const unsigned so = (unsigned)((ps * PR + rr * 4) * (int)p.ldo * 2);
                    u32x2_t hu, lu;
                    if (p.split_f16c8) {      // (wave-uniform: the consumer is an f16c8 launch -- pair_h8.h)
                        h8_split4(v, hu, lu);
                    } else {
                        const f32x4 hf = {bf16_round(v[0]), bf16_round(v[1]), bf16_round(v[2]), bf16_round(v[3])};
                        const bf16x4 hi = {(bf16_t)hf[0], (bf16_t)hf[1], (bf16_t)hf[2], (bf16_t)hf[3]};
                        const bf16x4 lo = {(bf16_t)(v[0] - hf[0]), (bf16_t)(v[1] - hf[1]), (bf16_t)(v[2] - hf[2]), (bf16_t)(v[3] - hf[3])};
                        hu = __builtin_bit_cast(u32x2_t, hi); lu = __builtin_bit_cast(u32x2_t, lo);
                    }
                    P_STORE64(hu, o_rs, o_lane, so);
                    P_STORE64(lu, o_rs, o_lane + lo_b, so);
                    if (cut) {
                        __builtin_amdgcn_raw_buffer_store_b32(hu[0], o_rs, o_part + so, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(lu[0], o_rs, o_part + lo_b + so, 0, 0);
                    }
                }
            }
        }
        if constexpr (EPI == EPI_X3ZR || EPI == EPI_X3Q) {
            // bf16x3 SepConvGRU gates in the convolution's epilogue (common.h): the accumulators go through the staging region as fp32, passes of 32
            // rows in two sub-passes of 16; a lane holds four columns of four rows per sub-pass.  The operands (start map, h pair, z) of sub-pass
            // s + 1 are requested after sub-pass s has consumed its own and before its stores; every wait is counted (the next tile's pieces /
            // the previous sub-pass's stores stay in flight).  X3ZR: waves 0-3 hold z columns, waves 4-7 r columns (wave-uniform roles).
            constexpr bool ZR = EPI == EPI_X3ZR;
            constexpr int PR = SB / 256 < WROWS ? SB / 256 : WROWS, NP = WROWS / PR, NS = NP * (PR / 16);
            static_assert(PR == 32, "two sub-passes of 16 rows per staged pass");
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
            u32x4_t mq[4], zq[4];
            u32x2_t hh[4], hl[4];
#define P_X3_LOAD(sp)                                                                                                  \
    _Pragma("unroll") for (int rr = 0; rr < 4; rr++) {                                                                 \
        const int r0_ = (sp) * 16 + rr * 4;                                                                            \
        mq[rr] = __builtin_amdgcn_raw_buffer_load_b128(m_rs, m_lane, r0_ * (int)p.ldr * 4, 0);                          \
        if (!ZR) zq[rr] = __builtin_amdgcn_raw_buffer_load_b128(z_rs, z_lane, r0_ * ld_z * 4, 0);                        \
        if (!ZR || is_r) {                                                                                             \
            hh[rr] = __builtin_amdgcn_raw_buffer_load_b64(h_rs, h_lane, r0_ * ld_h * 2, 0);                             \
            hl[rr] = __builtin_amdgcn_raw_buffer_load_b64(h_rs, h_lane + lo_b, r0_ * ld_h * 2, 0);                      \
        }                                                                                                              \
    }
            P_X3_LOAD(0)
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
                for (int sub = 0; sub < 2; sub++) {
                    const int sp = ps * 2 + sub;
                    if (sp == 0) { P_WAIT_OPERANDS() }
                    else if (!ZR || is_r) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);     // the previous sub-pass's 8 stores may still fly
                    else __builtin_amdgcn_s_waitcnt(0x0F70 | 4);                     // (z waves: 4 stores)
                    f32x4 res[4];
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const int row = sub * 16 + rr * 4 + rl;
                        f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 256 + ((cl ^ (row & 15)) << 4));
                        v += __builtin_bit_cast(f32x4, mq[rr]);
                        if constexpr (ZR) {
#pragma unroll
                            for (int e = 0; e < 4; e++) v[e] = __frcp_rn(1.0f + __expf(-v[e]));
                            if (is_r) {
                                const bf16x4 a = __builtin_bit_cast(bf16x4, hh[rr]), b = __builtin_bit_cast(bf16x4, hl[rr]);
#pragma unroll
                                for (int e = 0; e < 4; e++) v[e] *= (float)a[e] + (float)b[e];
                            }
                        } else {
                            const f32x4 z = __builtin_bit_cast(f32x4, zq[rr]);
                            const bf16x4 a = __builtin_bit_cast(bf16x4, hh[rr]), b = __builtin_bit_cast(bf16x4, hl[rr]);
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const float q = 1.0f - 2.0f * __frcp_rn(__expf(2.0f * v[e]) + 1.0f);
                                v[e] = (1.0f - z[e]) * ((float)a[e] + (float)b[e]) + z[e] * q;
                            }
                        }
                        res[rr] = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (sp + 1 < NS) { P_X3_LOAD(sp + 1) }       // the next sub-pass's operands (same registers), before this sub-pass's stores
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const unsigned r0 = (unsigned)(sp * 16 + rr * 4);
                        if (ZR && !is_r) {
                            P_STORE128(__builtin_bit_cast(u32x4_t, res[rr]), z_rs, z_lane, r0 * (unsigned)ld_z * 4u);
                        } else {
                            const f32x4 v = res[rr];
                            const f32x4 hf = {bf16_round(v[0]), bf16_round(v[1]), bf16_round(v[2]), bf16_round(v[3])};
                            const bf16x4 hi = {(bf16_t)hf[0], (bf16_t)hf[1], (bf16_t)hf[2], (bf16_t)hf[3]};
                            const bf16x4 lo = {(bf16_t)(v[0] - hf[0]), (bf16_t)(v[1] - hf[1]), (bf16_t)(v[2] - hf[2]), (bf16_t)(v[3] - hf[3])};
                            P_STORE64(__builtin_bit_cast(u32x2_t, hi), o_rs, o_lane, r0 * (unsigned)ld_o * 2u);
                            P_STORE64(__builtin_bit_cast(u32x2_t, lo), o_rs, o_lane + lo_b, r0 * (unsigned)ld_o * 2u);
                        }
                    }
                }
            }
#undef P_X3_LOAD
        }
        if constexpr (EPI == EPI_GRU) {
            // h' = (1 - z) h + z tanh(acc): the accumulators go through the staging region as in the fp32 path so that h, z and both
            // outputs are touched as whole row segments.  h / z of pass h+1 are requested before pass h's stores (two register sets),
            // every wait is counted.
            constexpr int PR = SB / 256 < WROWS ? SB / 256 : WROWS, NP = WROWS / PR;
            const int rl = lane_e >> 4, cl = lane_e & 15;
            const int n = en0 + wn * 64 + cl * 4;
            const auto h_rs = tile_rsrc(p.resid, p.ldr, 4), z_rs = tile_rsrc(p.aux, p.ldaux, 2), o_rs = tile_rsrc(p.out, p.ldo, 4), o2_rs = tile_rsrc(p.out2, p.ldo2, 2);
            const int rowl = wm * WROWS + rl;
            const unsigned h_lane = n < p.N ? (unsigned)(rowl * (int)p.ldr + n) * 4u : OOB, z_lane = n < p.N ? (unsigned)(rowl * (int)p.ldaux + n) * 2u : OOB;
            const unsigned o_lane = n < p.N ? (unsigned)(rowl * (int)p.ldo + n) * 4u : OOB, o2_lane = n < p.N ? (unsigned)(rowl * (int)p.ldo2 + n) * 2u : OOB;
            u32x4_t hreg[2][PR / 4];
            u32x2_t zreg[2][PR / 4];
#define P_GRU_LOAD(S, h)                                                                                            \
    _Pragma("unroll") for (int rr = 0; rr < PR / 4; rr++) {                                                          \
        hreg[S][rr] = __builtin_amdgcn_raw_buffer_load_b128(h_rs, h_lane, ((h) * PR + rr * 4) * (int)p.ldr * 4, 0);   \
        zreg[S][rr] = __builtin_amdgcn_raw_buffer_load_b64(z_rs, z_lane, ((h) * PR + rr * 4) * (int)p.ldaux * 2, 0);  \
    }
            P_GRU_LOAD(0, 0)
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
                if (ps == 0) { P_WAIT_OPERANDS() }
                else __builtin_amdgcn_s_waitcnt(0x0F70 | ((PR / 2) & 15) | (((PR / 2) >> 4) << 14));      // the previous pass's 2 PR / 4 stores may still fly
                if (ps + 1 < NP) {
                    if ((ps & 1) == 0) { P_GRU_LOAD(1, ps + 1) } else { P_GRU_LOAD(0, ps + 1) }
                }
#pragma unroll
                for (int rr = 0; rr < PR / 4; rr++) {
                    const int row = rr * 4 + rl;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 256 + ((cl ^ (row & 15)) << 4));
                    const f32x4 hv = __builtin_bit_cast(f32x4, hreg[ps & 1][rr]);
                    const bf16x4 zv = __builtin_bit_cast(bf16x4, zreg[ps & 1][rr]);
                    f32x4 hn;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float z = (float)zv[e];
                        hn[e] = (1.0f - z) * hv[e] + z * tanh_fast(v[e]);
                    }
                    const bf16x4 hb = {(bf16_t)hn[0], (bf16_t)hn[1], (bf16_t)hn[2], (bf16_t)hn[3]};
                    P_STORE128(__builtin_bit_cast(u32x4_t, hn), o_rs, o_lane, (ps * PR + rr * 4) * (int)p.ldo * 4);
                    P_STORE64(__builtin_bit_cast(u32x2_t, hb), o2_rs, o2_lane, (ps * PR + rr * 4) * (int)p.ldo2 * 2);
                }
            }
#undef P_GRU_LOAD
        }
    tile_done:
        if (!has_next) break;
        b = b2;
        if constexpr (PARK) {
            const uint4 t0 = park[0], t1 = park[512], t2 = park[1024];
            a_voff[0] = t0.x; a_voff[1] = t0.y; a_voff[2] = t0.z; a_voff[3] = t0.w;
            a_bits[0] = (int)t1.x; a_bits[1] = (int)t1.y; a_bits[2] = (int)t1.z; a_bits[3] = (int)t1.w;
            w_voff = t2.x;
        }
        P_PHASE_BARRIER()                 // every wave is done with its staging region (and its parked state): A'(1) / W'(1) may land
    }
#endif
}

// ---------------------------------------------------------------------------------------
static int g_cu_count = 0;
int cu_count() {
    if (g_cu_count == 0) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_cu_count = n;
    }
    return g_cu_count;
}

// What the persistent kernel takes (everything else stays with gemm_bf16_large_kernel): identity output / residual row maps,
// whole-vector rows for its staged epilogues, fragment-order start maps, wave-aligned gates.
bool pp_supported(const GemmDesc& d) {
    if (d.o_map.seg_rows != 0 || d.r_map.seg_rows != 0) return false;
    if (d.ln_xb && (d.epi != EPI_RESID_F32 || d.conv_KH > 0 || (d.ldxb & 3) != 0 || !d.ln_part || (d.N & 3) != 0)) return false;
    if (d.ln_stats && (!(d.epi == EPI_STORE || d.epi == EPI_GELU) || d.conv_KH > 0 || d.bias || !d.ln_cs || !d.ln_c || (d.N & 3) != 0 || d.frag_out || d.gate_from > 0 ||
                       d.resid_bf16 || d.tail_w))
        return false;
    if (d.init_bf16 && !d.init_frag) return false;
    if ((d.ldw & 63) != 0) return false;      // (the weight pieces' swizzle is applied as offset ^ 64: rows of whole 128 bytes)
    if ((d.N & 3) != 0 && !((d.epi == EPI_STORE || d.epi == EPI_SPLIT) && !d.frag_out && (d.N & 1) == 0)) return false;
    switch (d.epi) {
        case EPI_STORE:
        case EPI_GELU:
            if (d.frag_out) return true;
            if ((d.ldo & 7) != 0 || d.out_scale != 0.f) return false;
            if ((d.N & 7) != 0 && ((d.N & 1) != 0 || d.gate_from > 0 || d.resid_bf16 || d.tail_w)) return false;      // (plain stores only: see the N % 8 store)
            if (d.gate_from > 0 && ((d.gate_from & 63) != 0 || (d.ldaux & 7) != 0 || (d.ldo2 & 7) != 0)) return false;
            if (d.resid_bf16 && (d.ldrb & 7) != 0) return false;
            if (d.tail_w && (d.N != 256 || (d.ldtail & 3) != 0 || d.conv_KH == 0)) return false;
            return true;
        case EPI_RESID_F32:
            return (d.ldo & 3) == 0 && (d.ldr & 3) == 0 && d.act == 0 && d.out_scale == 0.f && d.resid != nullptr;
        case EPI_STORE_F32:
            return (d.ldo & 3) == 0 && d.act == 0 && d.out_scale == 0.f;
        case EPI_GRU:
            return ((d.ldo | d.ldr | d.ldaux | d.ldo2) & 3) == 0 && d.N <= 128;      // (RAFT's q convolutions: 128 channels)
        case EPI_X3ZR:
            return d.N == 256 && d.conv_KH > 0 && d.resid && d.aux && d.out2 && ((d.ldr | d.ldaux | d.ldo | d.ldo2 | d.split_lo) & 3) == 0 && d.act == 0 && d.out_scale == 0.f && !d.bias;
        case EPI_X3Q:
            return d.N == 128 && d.conv_KH > 0 && d.resid && d.aux && ((d.ldr | d.ldaux | d.ldo | d.split_lo) & 3) == 0 && d.act == 0 && d.out_scale == 0.f && !d.bias;
        case EPI_SPLIT:
            return (d.N & 1) == 0 && (d.ldo & 3) == 0 && (d.split_lo & 3) == 0 && d.split_lo > 0 && d.out_scale == 0.f && d.conv_KH > 0;
    }
    return false;
}

template <int EPI, bool CONV, int NWN, int WF>
int launch_large_pp(const GemmDesc& d, hipStream_t s) {
    constexpr int T_BM = 256, T_BN = 16 * WF * NWN;
    static_assert(NWN == 4 || NWN == 2, "the 64-wide tile (two workgroups per CU) stays with gemm_bf16_large_kernel");
    constexpr bool PING = !CONV;
    constexpr int LDS = 3 * P_AOP + ((PING && NWN < 4) ? 3 : 2) * T_BN * 128;
    if constexpr (WF == 3) {      // RAFT's convc2: N = 192 as ONE 256 x 192 tile per m-tile, eight waves of 128 x 48
        static DeviceOnce attr3;
        VTGB_FUNC_LDS_ONCE(attr3, (gemm_bf16_pp_kernel<EPI, CONV, NWN, false, PING, 3>), LDS);
        const int m_tiles3 = (d.M + T_BM - 1) / T_BM, n_tiles3 = (d.N + T_BN - 1) / T_BN, total3 = m_tiles3 * n_tiles3;
        const int grid3 = total3 < cu_count() ? total3 : cu_count();
        const double ef = 2.0 * d.M * d.N * d.K;
        ProfScope prof(VTGB_PROF_CONV, d.algo_flops > 0 ? d.algo_flops : ef, s, ef);
        hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, CONV, NWN, false, PING, 3>), dim3(grid3), dim3(512), LDS, s, d, m_tiles3, n_tiles3, 8, total3);
        VTGB_HIP(hipGetLastError());
        return VTGB_OK;
    } else {
    const int m_tiles = (d.M + T_BM - 1) / T_BM, n_tiles = (d.N + T_BN - 1) / T_BN;
    const int G = 8;                                                   // m-tiles per group of the tile order (see the kernel)
    const int total = m_tiles * n_tiles;
    const int resident = cu_count();                                   // one workgroup per CU (LDS-limited)
    const int grid = total < resident ? total : resident;
    const double exec_flops = 2.0 * d.M * d.N * d.K;
    ProfScope prof(CONV ? VTGB_PROF_CONV : VTGB_PROF_GEMM, d.algo_flops > 0 ? d.algo_flops : d.algo_flops < 0 ? 0.0 : exec_flops, s, exec_flops);
    if constexpr (EPI == EPI_STORE && CONV && NWN == 4) {
        if (d.tail_w) {      // the fused 1x1 tail is its own instantiation (its 64 weight-fragment registers are not every tile's problem)
            static DeviceOnce attr_t;
            VTGB_FUNC_LDS_ONCE(attr_t, (gemm_bf16_pp_kernel<EPI, CONV, NWN, true>), LDS);
            hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, CONV, NWN, true>), dim3(grid), dim3(512), LDS, s, d, m_tiles, n_tiles, G, total);
            VTGB_HIP(hipGetLastError());
            return VTGB_OK;
        }
    }
    if constexpr (!CONV && NWN == 4 && (EPI == EPI_STORE || EPI == EPI_GELU || EPI == EPI_RESID_F32)) {
        if (d.ln_stats || d.ln_xb) {      // LayerNorm folded into the GEMM: its own instantiation
            static DeviceOnce attr_ln;
            VTGB_FUNC_LDS_ONCE(attr_ln, (gemm_bf16_pp_kernel<EPI, CONV, NWN, false, PING, 4, true>), LDS);
            hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, CONV, NWN, false, PING, 4, true>), dim3(grid), dim3(512), LDS, s, d, m_tiles, n_tiles, G, total);
            VTGB_HIP(hipGetLastError());
            return VTGB_OK;
        }
    }
    static DeviceOnce attr;
    VTGB_FUNC_LDS_ONCE(attr, (gemm_bf16_pp_kernel<EPI, CONV, NWN>), LDS);
    hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, CONV, NWN>), dim3(grid), dim3(512), LDS, s, d, m_tiles, n_tiles, G, total);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
    }
}

// explicit entry points used by gemm.hip (one per instantiation it dispatches to)
#define PP_INST(EPI, CONV, NWN) template int launch_large_pp<EPI, CONV, NWN, 4>(const GemmDesc&, hipStream_t);
template int launch_large_pp<EPI_STORE, true, 4, 3>(const GemmDesc&, hipStream_t);
template int launch_large_pp<EPI_SPLIT, true, 4, 3>(const GemmDesc&, hipStream_t);
PP_INST(EPI_STORE, false, 4) PP_INST(EPI_GELU, false, 4) PP_INST(EPI_RESID_F32, false, 4) PP_INST(EPI_STORE_F32, false, 4)
PP_INST(EPI_STORE, false, 2) PP_INST(EPI_STORE_F32, false, 2)
PP_INST(EPI_STORE, true, 4) PP_INST(EPI_STORE, true, 2)
PP_INST(EPI_STORE_F32, true, 4) PP_INST(EPI_STORE_F32, true, 2)
PP_INST(EPI_GRU, true, 2)
PP_INST(EPI_SPLIT, true, 4) PP_INST(EPI_SPLIT, true, 2)
PP_INST(EPI_X3ZR, true, 4) PP_INST(EPI_X3Q, true, 2)
