// gru_fused.hip -- one SepConvGRU half-step (raft_utils/update.py:50-65) as ONE launch on gfx950 (round 4):
//
//     z, r = sigmoid(convz([h | x])), sigmoid(convr([h | x]));  q = tanh(convq([r * h | x]));  h' = (1 - z) h + z q
//
// with convz / convr / convq the 1x5 (horizontal half) or 5x1 (vertical half) convolutions over 384 channels, of which the
// loop-invariant `inp` third enters through start maps (raft.hip, `hoist`), leaving [h | motion + flow] = 256 channels.
//
// Rounds 1-3 ran two implicit-GEMM launches per half-step (z|r with the r * h gate, then q with the GRU update): every tap of
// every 64-channel chunk was its own LDS-DMA of the activation tile (5 x the bytes through the DMA path, 5 x the per-piece
// address arithmetic), z, r * h and the fp32 hidden state made a round trip through HBM in between (the q launch moved 5.7 GB
// at 4.65 TB/s: HBM-bound), and each launch paid its own prologue / epilogue per tile.  Here:
//
//  * Whole LINES per tile.  A 1x5 convolution never crosses an image row, a 5x1 never an image column: a tile is NL whole
//    lines (rows or columns, of any images) = up to 112 pixels = 7 MFMA row blocks, so r * h of the tile's own pixels is all
//    the q convolution needs: no halo, no recomputation.
//  * Tap reuse in LDS.  The tile's activations are staged ONCE (LDS-DMA): four 64-channel chunk images [slot][64 ch] (h0, h1, x0,
//    x1), a line's pixels in consecutive 128-byte slots with two zero slots between lines (and in front of the first), so that
//    tap d of pixel s is slot s + d -- zero padding is the image (out-of-range DMA lanes).  The fragments of all five taps are
//    read from shifted addresses (16-byte chunks rotated by 2 (slot >> 1): conflict-free for any window of 16 slots); the 35 (tap, row block) addresses of a lane
//    live in a small LDS table built once per workgroup, a tap's eight entries are two ds_read_b128 a tap ahead.
//  * Weights never touch LDS.  They are re-packed once per call in MFMA fragment order per (k-step, wave column, block): a
//    wave loads its own fragments with fully coalesced 1 KiB buffer loads straight into a four-k-step register ring
//    (inline asm, hand-counted vmcnt, three k-steps of lead), so the k-loops have NO barrier and no LDS-DMA: free-running waves,
//    each alternating one fragment read with 4 (z|r) or 2 (q) MFMAs, reads six pairs ahead of their MFMAs.
//  * The half-step stays on the CU: after the z|r k-loop a wave holds z and r of its 112 pixels x 32 channels; z is kept
//    (packed bf16, as rounds 1-3 stored it), r * h overwrites h IN PLACE in the h chunk images, the q k-loop reads
//    [x0, x1, rh0, rh1] from the same images, and the epilogue applies the GRU update.  The output-channel order of the packed
//    weights is chosen so that a lane's two accumulator blocks are 8 CONSECUTIVE channels of one pixel: every global access of
//    the epilogues is a 16-byte access, 64 contiguous bytes per pixel and wave instruction, no staging.
//  * The hidden state is kept as a bf16 pair hi | lo (hi = bf16(h) is what every convolution consumes anyway, lo =
//    bf16(h - hi): 16 significant bits) instead of an fp32 copy next to the bf16 one.
//  * Two persistent 4-wave workgroups per CU (74-78 KiB of LDS each): one's loads, gates and stores run beside the other's
//    k-loops; the next tile's images are requested as soon as the last fragment of this tile has been read, in front of the
//    GRU update's arithmetic and stores.
//
// HBM per pixel and half-step: 512 (h hi, x) + 768 (start maps) + 512 (hi, lo) + 512 (hi', lo') = 2.25 KiB, against 3.8 KiB for the
// two launches; MFMA work unchanged (2 x 5 x 256 x 384 FLOP per pixel executed, 2 x 5 x 384 x 384 algorithmic).
//
// Measured (bench batch, 2.31 M pixels; profiles/r04_gru_*.log; tools/exp/gru_abl.sh builds timing-only ablations and in-kernel
// phase stamps): 2.99 ms (two launches) -> 2.27 ms per half-step.  What the stamps showed on the way: a wave alone on its SIMD
// is INSTRUCTION-ISSUE bound in this loop (~11 instructions per 4 MFMAs; 30 k cycles for 18 k cycles of MFMA), two workgroups
// per CU recover 1.56 x in cycles but the chip then clocks 1.63 instead of 2.15 GHz; the W stream (8.6 KB per pixel out of L2)
// and the per-CU fill rate of ~30 GB/s are the same as the implicit-GEMM kernel's (5 taps of A through the DMA path there,
// weights through registers here); hipcc chaining an accumulator's two updates back to back cost 2 x on the k-loop; spilled
// registers reloaded behind the image DMAs serialised the update phase; priorities and a half-tile stagger were zero-sum.
#include <string.h>

#include "common.h"
#include "gemm_dev.h"

namespace {

constexpr int GF_JF = 7;                       // MFMA row blocks per wave = per tile: 112 rows (four waves along the channels)
constexpr int GF_ROWS = GF_JF * 16;
constexpr int GF_WAVES = 4;
constexpr int GF_MAX_IMG = 17 * 1024;          // bytes per chunk image (<= 152 slots): two workgroups (4 images each) share a CU's 160 KiB
constexpr unsigned GF_OOB = 0x80000000u;

typedef __attribute__((__vector_size__(4 * sizeof(int)))) int gf_i32x4;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned gf_u32x4;
typedef __attribute__((address_space(3))) void* gf_lptr_t;

struct GruHalfParams {
    int H, W, vert;                 // coarse grid; 0: lines are image rows (1x5), 1: lines are image columns (5x1)
    int L, NL, n_lines, n_tiles;    // line length, lines per tile, lines in the batch, tiles
    int slots, img_bytes;           // slots per chunk image (incl. the dummy rows' slots), bytes per image (slots rounded up to 8)
    uint32_t divW_mul, divW_sh, divL_mul, divL_sh, divL2_mul, divL2_sh;     // n / W, n / L, n / (L + 2)
    const bf16_t* hb;               // [M][128] bf16(h): operand and (in place) output
    bf16_t* hlo;                    // [M][128] bf16(h - bf16(h)): in place
    const bf16_t* X;                // [M][256]: motion features + flow in columns 128..255
    const uint4* wzr;               // packed z|r weights: [40 k-steps][4 wave columns][4 blocks][64 lanes] x 16 B
    const uint4* wq;                // packed q weights:   [40 k-steps][4 wave columns][2 blocks][64 lanes] x 16 B  (chunk order x0, x1, rh0, rh1)
    const uint4* szr;               // start maps in fragment order: [tile][wave][7][z, r][64 lanes] x 16 B (8 bf16 channels)
    const uint4* sq;                //                                [tile][wave][7][64 lanes]
    unsigned long long* dbg;        // GF_ABL & 16 builds: per (tile, phase) clock stamps of wave 0
};

__device__ __forceinline__ int gf_div(uint32_t n, uint32_t mul, uint32_t sh) { return (int)((__umulhi(n, mul) + n) >> sh); }

// slot -> byte offset of 16-byte chunk `chunk` (0..7) inside a chunk image
// (16-byte chunk c of slot s sits at position (c + 2 (s >> 1)) & 7 of the slot's 128 bytes -- a ROTATION by twice the slot pair, not the
// XOR by (s >> 1) & 7 of the GEMM kernels' images: fragment rows here start at ANY slot (taps shift them by -2 .. 2), and the XOR form is
// conflict-free only for windows that start at a multiple of 4 slots -- PMC: 162 M LDS bank-conflict cycles per launch, 17 x the
// implicit-GEMM kernel's.  With the rotation the two lane groups of a ds_read_b128 (chunk c: rows 0-3, 12-15; chunk c + 1: rows 4-11)
// land on even / odd positions for every window of 16 consecutive slots, and k-half 1 (chunk + 4) is still the address ^ 64.)
__device__ __forceinline__ int gf_swz(int slot, int chunk) { return slot * 128 + (((chunk + 2 * (slot >> 1)) & 7) << 4); }

// a, b rounded (nearest even) to bf16: one v_cvt_pk_bf16_f32; .x / .y are the rounded values as floats, .z the packed pair
typedef __attribute__((ext_vector_type(2))) __bf16 gf_bf16x2;
typedef __attribute__((ext_vector_type(2))) float gf_f32x2;
__device__ __forceinline__ unsigned gf_pack2(float a, float b) {
    const gf_f32x2 v = {a, b};
    const gf_bf16x2 t = __builtin_convertvector(v, gf_bf16x2);      // (one v_cvt_pk_bf16_f32; the unpacking below works on the bits, so no excess precision survives)
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ float gf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float gf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

__device__ __forceinline__ float gf_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// tanh for the candidate state: 1 - 2 / (exp(2x) + 1) on the hardware exp / rcp (gemm_dev.h's tanh_fast asks for the correctly rounded
// reciprocal: a 10-instruction division sequence per value, a quarter of this kernel's update phase)
__device__ __forceinline__ float gf_tanh(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v) + 1.0f); }

// pixel index of (line gl of the batch, position pos in the line)
__device__ __forceinline__ int gf_pixel(const GruHalfParams& p, int gl, int pos) {
    if (!p.vert) return gl * p.L + pos;
    const int img = gf_div((uint32_t)gl, p.divW_mul, p.divW_sh), x = gl - img * p.W;
    return img * (p.H * p.W) + pos * p.W + x;
}

#define GF_PHASE_BARRIER() __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);

__global__ __launch_bounds__(256, 2) void gru_half_kernel(const GruHalfParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave;                            // wave column: z / r / q channels [32 wn, 32 wn + 32)
    const int fr = lane & 15, fg = lane >> 4;
    const int IMG = p.img_bytes, L2 = p.L + 2;
    const int rows_valid = p.NL * p.L;

    // ---- fragment address table in LDS, built once per workgroup: tbl[lane][tap t][row block j] = byte offset, inside a chunk image, of
    // k-half 0 of this lane's fragment row of block j shifted by tap t - 2 (slot * 128 + swizzled 16-byte chunk of lane >> 4); k-half 1 is
    // that offset ^ 64.  Rows beyond the tile's lines use the dummy slot (zeros; five zero slots around it).  The k-loops read a tap's
    // eight entries with two ds_read_b128 (lane stride 160 B: conflict-free) instead of forming every address on the vector unit --
    // a wave alone on its SIMD is issue-bound: 3.5 of the ~11 instructions per (row block, k-half) pair were address arithmetic.
    char* const tbl = smem + 4 * IMG;
    {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = j * 16 + fr;
            const int l = gf_div((uint32_t)r, p.divL_mul, p.divL_sh), pos = r - l * p.L;
            const int slot = (j < GF_JF && r < rows_valid) ? 2 + l * L2 + pos : 2 + p.NL * L2 + 2;
#pragma unroll
            for (int t = 0; t < 5; t++) {
                const int sp = slot + t - 2;
                if (wave == 0) *reinterpret_cast<int*>(tbl + lane * 160 + t * 32 + j * 4) = gf_swz(sp, fg);
            }
        }
    }
    const int tbl_lane = 4 * IMG + lane * 160;      // (byte offset of this lane's table rows)
    // ---- LDS-DMA pieces (8 slots x 128 B each): this wave stages pieces wave + 8 i of every image
    const int n_pieces = IMG >> 10;
    const gf_i32x4 wzr_rs = {__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)p.wzr), __builtin_amdgcn_readfirstlane((int)(((uint64_t)p.wzr >> 32) & 0xFFFFu)),
                             40 * 16 * 1024, 0x00020000};
    const gf_i32x4 wq_rs = {__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)p.wq), __builtin_amdgcn_readfirstlane((int)(((uint64_t)p.wq >> 32) & 0xFFFFu)),
                            40 * 8 * 1024, 0x00020000};
    const unsigned w1_voff = (unsigned)(wn * 4096 + lane * 16), w2_voff = (unsigned)(wn * 2048 + lane * 16);

    // the tile's first pixel's image start: base of the per-tile descriptors (every offset inside a tile is then small)
    auto tile_base = [&](int tile) -> int {
        const int gl0 = tile * p.NL;
        if (!p.vert) return gl0 * p.L;
        return gf_div((uint32_t)gl0, p.divW_mul, p.divW_sh) * (p.H * p.W);
    };
    auto issue_images = [&](int tile) {
        const int mb = tile_base(tile);
        const auto h_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.hb) + (int64_t)mb * 128, 0, 0x7FFFFF00, 0x00020000);
        const auto x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X) + (int64_t)mb * 256, 0, 0x7FFFFF00, 0x00020000);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int pi = wave + GF_WAVES * i;
            if (pi < n_pieces) {                                        // (wave-uniform)
                int lane_p = lane;
                asm volatile("" : "+v"(lane_p));                        // (re-derived per tile: nothing of this is worth a register across the k-loops)
                const int s = pi * 8 + (lane_p >> 3), s2 = s - 2;
                const int l = s2 >= 0 ? gf_div((uint32_t)s2, p.divL2_mul, p.divL2_sh) : 0, pos = s2 - l * L2, gl = tile * p.NL + l;
                const bool ok = s2 >= 0 && pos < p.L && l < p.NL && gl < p.n_lines;
                const int dc = (((lane_p & 7) - 2 * (s >> 1)) & 7) * 16;    // the data chunk that lands in position lane & 7 of slot s (gf_swz)
                const int mrel = ok ? gf_pixel(p, gl, pos) - mb : 0;
                const unsigned vh = ok ? (unsigned)(mrel * 256 + dc) : GF_OOB, vx = ok ? (unsigned)(mrel * 512 + 256 + dc) : GF_OOB;
                char* const dst = smem + pi * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(h_rs, (gf_lptr_t)(dst), 16, vh, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(h_rs, (gf_lptr_t)(dst + IMG), 16, vh, 128, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (gf_lptr_t)(dst + 2 * IMG), 16, vx, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (gf_lptr_t)(dst + 3 * IMG), 16, vx, 128, 0, 0);
            }
        }
    };

    // ---- weight ring: two taps (2 x 2 k-steps) of fragments in registers; loads are inline asm (hipcc would drain a ring it
    // cannot see through), waits are counted by hand: each tap is NB * 2 loads, issued in tap order
#ifndef GF_ABL
#define GF_ABL 0      /* timing-only variants (tools/exp/gru_abl.sh; results are WRONG by construction): 1 no weight loads, 2 no fragment reads, 4 no address arithmetic, 8 no MFMAs, 16 phase stamps, 32 no state stores, 64 no image DMA after the first tile */
#endif
#define GF_W_ISSUE_H(REGH, NB, RS, VOFF, STEPBYTES, step)      /* the NB fragments of k-step `step` (= 2 tap + k-half) */      \
    if constexpr (!(GF_ABL & 1)) {                                                                                             \
        const int so_ = __builtin_amdgcn_readfirstlane((step) * (STEPBYTES));                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < NB; i_++) {                                                                    \
            const int so_i_ = so_ + i_ * 1024;                                                                                 \
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(REGH[i_]) : "v"(VOFF), "s"(RS), "s"(so_i_) : "memory"); \
        }                                                                                                                      \
    }
#define GF_W_ISSUE(REG, NB, RS, VOFF, STEPBYTES, tap)                                                                          \
    {                                                                                                                          \
        GF_W_ISSUE_H(REG[0], NB, RS, VOFF, STEPBYTES, (tap) * 2)                                                               \
        GF_W_ISSUE_H(REG[1], NB, RS, VOFF, STEPBYTES, (tap) * 2 + 1)                                                           \
    }
#define GF_W_DRAIN(REG, NB)   /* the ring's last (out-of-range) requests have returned: the registers may be reused */        \
    {                                                                                                                          \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                       \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ks_++)                                                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < NB; i_++) asm volatile("" : "+v"(REG[ks_][i_]));                           \
    }
#define GF_W_WAIT_H(REGH, NB)     /* everything but the three younger k-steps' fragments has landed */                         \
    {                                                                                                                          \
        if constexpr (!(GF_ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NB) : "memory");                             \
        _Pragma("unroll") for (int i_ = 0; i_ < NB; i_++) asm volatile("" : "+v"(REGH[i_]));                                   \
    }
    // ---- k-loop: the 20 taps as 280 (row block, k-half) pairs, each one fragment read + NB MFMAs.  Pair p of a tap: k-half p / 7, row
    // block p % 7 -- the two updates of an accumulator block are 7 pairs apart (hipcc, left to itself, put them back to back with the
    // accumulator renamed in between: every MFMA then waited out its predecessor's write-back, 32 cycles per MFMA measured); the order is
    // pinned with sched_barrier.  The reads run D - 1 pairs ahead of their MFMAs through a ring of D fragments (LDS latency under the
    // wave's own MFMAs; a tap boundary is not a bubble); a pair's address is one v_add / v_xad on its table entry (image base, ^ 64 for
    // k-half 1); the table rows of a tap (two ds_read_b128) are requested a tap ahead.
#define GF_TBL_LOAD(AT, tap_)                                                                                                  \
    {                                                                                                                          \
        const int t_ = (tap_) % 5;                                                                                             \
        const char* const r_ = smem + tbl_lane + t_ * 32;                                                                      \
        AT[0] = *reinterpret_cast<const gf_i32x4*>(r_);                                                                        \
        AT[1] = *reinterpret_cast<const gf_i32x4*>(r_ + 16);                                                                   \
    }
#define GF_IMG_OFF(tap_, ROT) ((((((tap_) / 5) > 3 ? 3 : ((tap_) / 5)) + ROT) & 3) * IMG)
#define GF_KLOOP(ACC, RA, RB, NB, D, RS, VOFF, STEPBYTES, ROT)                                                                 \
    {                                                                                                                          \
        bf16x8 xr_[D];                                                                                                         \
        gf_i32x4 ata_[2], atb_[2];               /* table rows of the even / odd tap in flight: entries [j >> 2][j & 3] */      \
        GF_TBL_LOAD(ata_, 0)                                                                                                   \
        GF_TBL_LOAD(atb_, 1)                                                                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < D - 1; q_++) {                                                                 \
            const int a_ = (GF_ABL & 4) ? 0 : (ata_[(q_ % 7) >> 2][(q_ % 7) & 3] ^ ((q_ / 7) * 64)) + GF_IMG_OFF(0, ROT);      \
            if constexpr (GF_ABL & 2) asm volatile("" : "=v"(xr_[q_]) : "v"(a_));                                              \
            else xr_[q_] = *reinterpret_cast<const bf16x8*>(smem + a_);                                                        \
        }                                                                                                                      \
        for (int tp = 0; tp < 10; tp++) {           /* taps 2 tp (ring entry RA), 2 tp + 1 (RB): chunk-major, tap-minor = the weights' K order */ \
            const int ioa_ = GF_IMG_OFF(2 * tp, ROT), iob_ = GF_IMG_OFF(2 * tp + 1, ROT), ioc_ = GF_IMG_OFF(2 * tp + 2, ROT);  \
            _Pragma("unroll") for (int P_ = 0; P_ < 28; P_++) {                                                                \
                if (P_ == 0) GF_W_WAIT_H(RA[0], NB)                                                                            \
                if (P_ == 7) GF_W_WAIT_H(RA[1], NB)                                                                            \
                if (P_ == 14) GF_W_WAIT_H(RB[0], NB)                                                                           \
                if (P_ == 21) GF_W_WAIT_H(RB[1], NB)                                                                           \
                /* the even tap's rows are last used by the prefetch of pair 13 (P_ = 13 - (D - 1)): re-request them for tap 2 tp + 2; the odd tap's after pair 27 */ \
                if (P_ == 14 - (D - 1)) GF_TBL_LOAD(ata_, 2 * tp + 2)                                                          \
                if (P_ == 28 - (D - 1)) GF_TBL_LOAD(atb_, 2 * tp + 3)                                                          \
                {                                                                                                              \
                    const int Q_ = P_ + D - 1, pq_ = Q_ % 14;                                                                  \
                    const int e_ = Q_ < 14 ? ata_[(pq_ % 7) >> 2][(pq_ % 7) & 3] : Q_ < 28 ? atb_[(pq_ % 7) >> 2][(pq_ % 7) & 3] : ata_[(pq_ % 7) >> 2][(pq_ % 7) & 3]; \
                    const int a_ = (GF_ABL & 4) ? 0 : (e_ ^ ((pq_ / 7) * 64)) + (Q_ < 14 ? ioa_ : Q_ < 28 ? iob_ : ioc_);      \
                    if constexpr (GF_ABL & 2) asm volatile("" : "+v"(xr_[Q_ % D]) : "v"(a_));                                  \
                    else xr_[Q_ % D] = *reinterpret_cast<const bf16x8*>(smem + a_);                                            \
                }                                                                                                              \
                _Pragma("unroll") for (int i = 0; i < NB; i++) {                                                               \
                    const bf16x8 wf_ = __builtin_bit_cast(bf16x8, P_ < 14 ? RA[(P_ % 14) / 7][i] : RB[(P_ % 14) / 7][i]);      \
                    if constexpr (GF_ABL & 8) asm volatile("" : "+v"(ACC[i][P_ % 7]) : "v"(wf_), "v"(xr_[P_ % D]));           \
                    else ACC[i][P_ % 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf_, xr_[P_ % D], ACC[i][P_ % 7], 0, 0, 0);  \
                }                                                                                                              \
                /* a k-step's fragments are re-requested for two taps on as soon as its 7 pairs are done: 3 k-steps (21 pairs) of lead */ \
                if (P_ == 6) GF_W_ISSUE_H(RA[0], NB, RS, VOFF, STEPBYTES, 4 * tp + 4)  /* (taps 20, 21: beyond the descriptor's range -- no traffic) */ \
                if (P_ == 13) GF_W_ISSUE_H(RA[1], NB, RS, VOFF, STEPBYTES, 4 * tp + 5)                                         \
                if (P_ == 20) GF_W_ISSUE_H(RB[0], NB, RS, VOFF, STEPBYTES, 4 * tp + 6)                                         \
                if (P_ == 27) GF_W_ISSUE_H(RB[1], NB, RS, VOFF, STEPBYTES, 4 * tp + 7)                                         \
                __builtin_amdgcn_sched_barrier(0);                                                                             \
            }                                                                                                                  \
        }                                                                                                                      \
    }
#ifndef GF_PRIO
#define GF_PRIO 0      /* experiments: 1 = the k-loops at s_setprio 1 (gates / update at 0), 2 = the reverse */
#endif
#define GF_PRIO_K(on_) { if constexpr (GF_PRIO == 1) __builtin_amdgcn_s_setprio((on_) ? 1 : 0); if constexpr (GF_PRIO == 2) __builtin_amdgcn_s_setprio((on_) ? 0 : 1); }
#define GF_STAMP(k_)                                                                                                           \
    if constexpr (GF_ABL & 16) {                                                                                               \
        if (p.dbg && wave == 0 && lane == 0)                                                                                   \
            p.dbg[(int64_t)tile * 8 + (k_)] = ((k_) == 0 ? ((unsigned long long)(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) & 0xFFFF) << 48) | ((unsigned long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF) << 44) : 0ull) | (__builtin_amdgcn_s_memtime() & 0xFFFFFFFFFFFull); \
    }

    int tile = (int)blockIdx.x;
    if (tile >= p.n_tiles) return;
    issue_images(tile);
    while (true) {
        GF_STAMP(0)
        gf_i32x4 wa[2][4], wb[2][4];               // the two ring entries (taps of even / odd index)
        // ================= z|r: start maps -> accumulators, behind the first two taps' weight loads
        GF_W_ISSUE(wa, 4, wzr_rs, w1_voff, 16384, 0)
        GF_W_ISSUE(wb, 4, wzr_rs, w1_voff, 16384, 1)
        f32x4 acc[4][GF_JF];
        {
            const uint4* src = p.szr + ((int64_t)(tile * GF_WAVES + wave) * GF_JF * 2) * 64 + lane;
            uint4 t[GF_JF][2];
#pragma unroll
            for (int j = 0; j < GF_JF; j++) { t[j][0] = src[(j * 2) * 64]; t[j][1] = src[(j * 2 + 1) * 64]; }
#pragma unroll
            for (int j = 0; j < GF_JF; j++)
#pragma unroll
                for (int pr = 0; pr < 2; pr++) {
                    const gf_u32x4 v = __builtin_bit_cast(gf_u32x4, t[j][pr]);
                    acc[pr * 2][j] = f32x4{gf_lo(v[0]), gf_hi(v[0]), gf_lo(v[1]), gf_hi(v[1])};
                    acc[pr * 2 + 1][j] = f32x4{gf_lo(v[2]), gf_hi(v[2]), gf_lo(v[3]), gf_hi(v[3])};
                }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0): this wave's image pieces have landed (and the ring's first taps)
        GF_PHASE_BARRIER()                          // ... and everybody else's
        GF_STAMP(1)
        GF_PRIO_K(1)
        GF_KLOOP(acc, wa, wb, 4, 7, wzr_rs, w1_voff, 16384, 0)
        GF_PRIO_K(0)
        GF_STAMP(2)
        GF_W_DRAIN(wa, 4)
        GF_W_DRAIN(wb, 4)
        // ================= r * h in place, z and h to registers; q's start map and first weights requested first
        gf_i32x4 qa[2][2], qb[2][2];
        GF_W_ISSUE(qa, 2, wq_rs, w2_voff, 8192, 0)
        GF_W_ISSUE(qb, 2, wq_rs, w2_voff, 8192, 1)
        GF_PHASE_BARRIER()                          // every wave has read its last h fragment
        GF_STAMP(3)
        gf_u32x4 zpk[GF_JF];                        // 8 bf16: z of this lane's 8 channels of row block j (h is re-read in the update phase: 28 registers
                                                    // across the q k-loop cost spills whose reloads waited behind the next tile's image DMAs)
        f32x4 acc2[2][GF_JF];
        {
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));        // (nothing of the epilogues' address arithmetic is worth a register across the k-loops)
            char* const himg = smem + (wn >> 1) * IMG;
#pragma unroll
            for (int j = 0; j < GF_JF; j++) {
                // (LDS is accessed as bf16x8 everywhere: a store through another type would not alias the k-loops' fragment reads for hipcc)
                bf16x8* const hp = reinterpret_cast<bf16x8*>(himg + (*reinterpret_cast<const int*>(smem + 4 * IMG + lane_e * 160 + 2 * 32 + j * 4) ^ ((wn & 1) * 64)));
                const gf_u32x4 h8 = __builtin_bit_cast(gf_u32x4, *hp);
                gf_u32x4 z8, rh8;
#pragma unroll
                for (int e = 0; e < 4; e++) {       // dword e: channels 2 e, 2 e + 1 -> accumulator block e >> 1, registers 2 (e & 1), 2 (e & 1) + 1
                    const int bi = e >> 1, r0 = 2 * (e & 1);
                    z8[e] = gf_pack2(gf_sigmoid(acc[bi][j][r0]), gf_sigmoid(acc[bi][j][r0 + 1]));
                    const unsigned rr = gf_pack2(gf_sigmoid(acc[2 + bi][j][r0]), gf_sigmoid(acc[2 + bi][j][r0 + 1]));      // r as rounds 1-3 stored it (bf16)
                    rh8[e] = gf_pack2(gf_lo(rr) * gf_lo(h8[e]), gf_hi(rr) * gf_hi(h8[e]));
                }
                *hp = __builtin_bit_cast(bf16x8, rh8);
                zpk[j] = z8;
                acc2[0][j] = f32x4{0.f, 0.f, 0.f, 0.f};      // (q's start map -- bias + the hoisted `inp` term -- is added in the update phase: 28 registers
                acc2[1][j] = f32x4{0.f, 0.f, 0.f, 0.f};      //  fewer across the gates, where z, h and the accumulators of both stages overlap)
            }
        }
        GF_PHASE_BARRIER()                          // r * h of every wave is in the images
        GF_STAMP(4)
        // ================= q: chunk order x0, x1, rh0, rh1 (images 2, 3, 0, 1)
        GF_PRIO_K(1)
        GF_KLOOP(acc2, qa, qb, 2, 7, wq_rs, w2_voff, 8192, 2)
        GF_PRIO_K(0)
        GF_STAMP(5)
        // ================= GRU update: h = hi + lo; h' = (1 - z) h + z tanh(q); hi' | lo' stored in place
        const int mb = tile_base(tile);
        const auto lo_rs = __builtin_amdgcn_make_buffer_rsrc(p.hlo + (int64_t)mb * 128, 0, 0x7FFFFF00, 0x00020000);
        const auto hi_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.hb) + (int64_t)mb * 128, 0, 0x7FFFFF00, 0x00020000);
        unsigned off_j[GF_JF];
        gf_u32x4 lo_t[GF_JF], hpk[GF_JF];
        gf_u32x4 sq_t[GF_JF];
        {
            const gf_u32x4* src = reinterpret_cast<const gf_u32x4*>(p.sq) + ((int64_t)(tile * GF_WAVES + wave) * GF_JF) * 64 + lane;
#pragma unroll
            for (int j = 0; j < GF_JF; j++) sq_t[j] = src[j * 64];
        }
        {
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
#pragma unroll
            for (int j = 0; j < GF_JF; j++) {
                const int r = j * 16 + (lane_e & 15);
                const int l = gf_div((uint32_t)r, p.divL_mul, p.divL_sh), pos = r - l * p.L, gl = tile * p.NL + l;
                const bool ok = r < rows_valid && gl < p.n_lines;
                off_j[j] = ok ? (unsigned)((gf_pixel(p, gl, pos) - mb) * 256 + wn * 64 + (lane_e >> 4) * 16) : GF_OOB;
                lo_t[j] = __builtin_amdgcn_raw_buffer_load_b128(lo_rs, off_j[j], 0, 0);
                hpk[j] = __builtin_amdgcn_raw_buffer_load_b128(hi_rs, off_j[j], 0, 0);
            }
        }
        GF_PHASE_BARRIER()                          // every wave has read its last fragment: the images are free
        GF_STAMP(6)
        const int next = tile + (int)gridDim.x;
        GF_W_DRAIN(qa, 2)                           // (vmcnt(0): the lo rows are here too -- the DMA pieces below must not sit in front of their wait)
        GF_W_DRAIN(qb, 2)
#pragma unroll
        for (int j = 0; j < GF_JF; j++) { asm volatile("" : "+v"(lo_t[j])); asm volatile("" : "+v"(sq_t[j])); asm volatile("" : "+v"(hpk[j])); asm volatile("" : "+v"(zpk[j])); }
        // (z too: three of its seven entries are spilled across the q k-loop; un-pinned, hipcc reloaded them in the MIDDLE of the update phase with
        //  vmcnt(0) -- i.e. behind the next tile's image DMA issued just above)
        if (next < p.n_tiles && !(GF_ABL & 64)) issue_images(next);
#pragma unroll
        for (int j = 0; j < GF_JF; j++) {
            gf_u32x4 hi_o, lo_o;
#pragma unroll
            for (int e = 0; e < 4; e++) {           // dword e: channels 2 e, 2 e + 1
                const int bi = e >> 1, r0 = 2 * (e & 1);
                const unsigned zz = zpk[j][e], hh = hpk[j][e], ll = lo_t[j][e], ss = sq_t[j][e];
                const float za = gf_lo(zz), zb = gf_hi(zz);
                const float ha = gf_lo(hh) + gf_lo(ll), hb_ = gf_hi(hh) + gf_hi(ll);
                const float na = (1.0f - za) * ha + za * gf_tanh(acc2[bi][j][r0] + gf_lo(ss)), nb = (1.0f - zb) * hb_ + zb * gf_tanh(acc2[bi][j][r0 + 1] + gf_hi(ss));
                const unsigned hi2 = gf_pack2(na, nb);
                hi_o[e] = hi2;
                lo_o[e] = gf_pack2(na - gf_lo(hi2), nb - gf_hi(hi2));
            }
            if constexpr (GF_ABL & 32) { asm volatile("" :: "v"(hi_o), "v"(lo_o)); }
            else {
                __builtin_amdgcn_raw_buffer_store_b128(hi_o, hi_rs, off_j[j], 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(lo_o, lo_rs, off_j[j], 0, 0);
            }
        }
        GF_STAMP(7)
        if (next >= p.n_tiles) break;
        tile = next;
    }
#endif
}

// ---- one-time (per call) re-packing of the GRU weights into fragment order.  src: [rows][K] bf16, K = 20 x 64 in the implicit-GEMM K
// order (64-channel chunk major, tap minor; chunks h0, h1, x0, x1).  dst step s = (cc * 5 + t) * 2 + ks with cc the PROCESSING order of
// the chunks (z|r: 0 1 2 3; q: 2 3 0 1).  Block i of wave column wn, MFMA row a (= lane & 15): output channel wn * 32 + (a >> 2) * 8 +
// (i & 1) * 4 + (a & 3) -- a lane's two blocks of a gate are then 8 consecutive channels -- of gate i >> 1 (z|r: rows 0..127 z, 128..255 r).
__global__ void gru_pack_w_kernel(const bf16_t* __restrict__ src, uint4* __restrict__ dst, int NB, int rot) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 40 * 4 * NB * 64) return;
    const int lane = idx & 63, i = (idx >> 6) % NB, wn = (idx / (64 * NB)) & 3, s = idx / (64 * NB * 4);
    const int ks = s & 1, ct = s >> 1, cc = ct / 5, t = ct - cc * 5, kc = (cc + rot) & 3;
    const int a = lane & 15, fg = lane >> 4;
    const int row = (i >> 1) * 128 + wn * 32 + (a >> 2) * 8 + (i & 1) * 4 + (a & 3);
    const int k0 = (kc * 5 + t) * 64 + ks * 32 + fg * 8;
    dst[idx] = *reinterpret_cast<const uint4*>(src + (int64_t)row * 1280 + k0);
}

// ---- start maps (bias + the loop-invariant `inp` convolution), row-major [M][256] (z | r) and [M][128] (q) -> the fused kernel's
// fragment order, once per call and half
__global__ void gru_startmap_kernel(const GruHalfParams p, const bf16_t* __restrict__ zr, const bf16_t* __restrict__ q, uint4* __restrict__ szr,
                                    uint4* __restrict__ sq) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)p.n_tiles * GF_WAVES * GF_JF * 64) return;
    const int lane = (int)(idx & 63), j = (int)((idx >> 6) % GF_JF), wave = (int)((idx / (64 * GF_JF)) % GF_WAVES), tile = (int)(idx / (64 * GF_JF * GF_WAVES));
    const int wn = wave, fr = lane & 15, fg = lane >> 4;
    const int r = j * 16 + fr, l = r / p.L, pos = r - l * p.L, gl = tile * p.NL + l;
    uint4 vz = make_uint4(0, 0, 0, 0), vr = vz, vq = vz;
    if (r < p.NL * p.L && gl < p.n_lines) {
        const int64_t m = gf_pixel(p, gl, pos);
        vz = *reinterpret_cast<const uint4*>(zr + m * 256 + wn * 32 + fg * 8);
        vr = *reinterpret_cast<const uint4*>(zr + m * 256 + 128 + wn * 32 + fg * 8);
        vq = *reinterpret_cast<const uint4*>(q + m * 128 + wn * 32 + fg * 8);
    }
    const int64_t o = ((int64_t)(tile * GF_WAVES + wave) * GF_JF + j);
    szr[(o * 2) * 64 + lane] = vz;
    szr[(o * 2 + 1) * 64 + lane] = vr;
    sq[o * 64 + lane] = vq;
}

void magic_div_u32(uint32_t d, uint32_t* mul, uint32_t* sh) {
    uint32_t s = 0;
    while ((1ull << s) < d) s++;
    *mul = (uint32_t)((((1ull << s) - d) << 32) / d + 1);
    *sh = s;
}

GruHalfParams gru_geometry(int n_img, int H, int W, int vert) {
    GruHalfParams p;
    memset(&p, 0, sizeof(p));
    p.H = H; p.W = W; p.vert = vert;
    p.L = vert ? H : W;
    p.NL = GF_ROWS / p.L;
    while (p.NL > 1 && 2 + p.NL * (p.L + 2) + 5 > GF_MAX_IMG / 128) p.NL--;      // (short lines: the pad slots count)
    p.n_lines = n_img * (vert ? W : H);
    p.n_tiles = p.NL > 0 ? (p.n_lines + p.NL - 1) / p.NL : 0;
    p.slots = 2 + p.NL * (p.L + 2) + 5;             // + the dummy rows' slot and two zero slots either side of it
    p.img_bytes = (p.slots + 7) / 8 * 8 * 128;
    magic_div_u32((uint32_t)W, &p.divW_mul, &p.divW_sh);
    magic_div_u32((uint32_t)p.L, &p.divL_mul, &p.divL_sh);
    magic_div_u32((uint32_t)(p.L + 2), &p.divL2_mul, &p.divL2_sh);
    return p;
}

}  // namespace

// ---------------------------------------------------------------------------------------
// host side (called by raft.hip)
// ---------------------------------------------------------------------------------------
bool gru_fused_supported(int n_img, int H, int W) {
    for (int vert = 0; vert < 2; vert++) {
        const GruHalfParams p = gru_geometry(n_img, H, W, vert);
        if (p.NL < 1 || p.img_bytes > GF_MAX_IMG || p.n_tiles < 1) return false;
        if ((int64_t)p.n_tiles * GF_WAVES * GF_JF * 2 * 64 >= (1ll << 31)) return false;
        // the per-tile descriptors address a tile's pixels relative to its first image: offsets stay far below 2^31
        const int64_t span = ((int64_t)p.NL / (vert ? W : 1) + 2) * H * W;
        if (span * 512 >= 0x7FFFFF00ll) return false;
    }
    return true;
}

// bytes of the fragment-order start maps of one half (z|r, q)
void gru_fused_startmap_bytes(int n_img, int H, int W, int vert, size_t* szr, size_t* sq) {
    const GruHalfParams p = gru_geometry(n_img, H, W, vert);
    *sq = (size_t)p.n_tiles * GF_WAVES * GF_JF * 64 * 16;
    *szr = 2 * *sq;
}
size_t gru_fused_packed_w_bytes(int gate_blocks) { return (size_t)40 * 4 * gate_blocks * 1024; }

int launch_gru_pack_w(const void* w_rowmajor, void* packed, int gate_blocks, int rot, hipStream_t s) {
    const int n = 40 * 4 * gate_blocks * 64;
    hipLaunchKernelGGL(gru_pack_w_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const bf16_t*)w_rowmajor, (uint4*)packed, gate_blocks, rot);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

int launch_gru_startmap(int n_img, int H, int W, int vert, const void* zr_rowmajor, const void* q_rowmajor, void* szr, void* sq, hipStream_t s) {
    const GruHalfParams p = gru_geometry(n_img, H, W, vert);
    const int64_t n = (int64_t)p.n_tiles * GF_WAVES * GF_JF * 64;
    hipLaunchKernelGGL(gru_startmap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, (const bf16_t*)zr_rowmajor, (const bf16_t*)q_rowmajor, (uint4*)szr,
                       (uint4*)sq);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

#if (GF_ABL & 16)
static unsigned long long* g_gru_dbg = nullptr;
extern "C" void vtgb_debug_set_gru_dbg(void* p) { g_gru_dbg = (unsigned long long*)p; }
#endif

int launch_gru_half(int n_img, int H, int W, int vert, void* hb, void* hlo, const void* X, const void* wzr_packed, const void* wq_packed, const void* szr,
                    const void* sq, hipStream_t s) {
    GruHalfParams p = gru_geometry(n_img, H, W, vert);
    p.hb = (const bf16_t*)hb; p.hlo = (bf16_t*)hlo; p.X = (const bf16_t*)X;
    p.wzr = (const uint4*)wzr_packed; p.wq = (const uint4*)wq_packed; p.szr = (const uint4*)szr; p.sq = (const uint4*)sq;
#if (GF_ABL & 16)
    p.dbg = g_gru_dbg;
#endif
    const int lds = 4 * p.img_bytes + 64 * 160;      // four chunk images + the fragment address table
    static DeviceOnce attr;
    VTGB_FUNC_LDS_ONCE(attr, gru_half_kernel, 4 * GF_MAX_IMG + 64 * 160);
#ifndef GF_WG_PER_CU
#define GF_WG_PER_CU 2
#endif
    const int grid = p.n_tiles < GF_WG_PER_CU * cu_count() ? p.n_tiles : GF_WG_PER_CU * cu_count();      // two workgroups per CU: one's memory phases under the other's k-loops
    const double M = (double)n_img * H * W;
    ProfScope prof(VTGB_PROF_CONV, 2.0 * M * 384.0 * (5 * 384), s, 2.0 * M * 384.0 * (5 * 256));
    hipLaunchKernelGGL(gru_half_kernel, dim3(grid), dim3(256), lds, s, p);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
