// conv64.hip -- 3x3, stride 1, 64 -> 64 channel convolution with the input rows RESIDENT in LDS (gfx950).
//
// RAFT's feature / context encoders spend their first residual stage (112 x 112 x 64 at 224 x 224 input; extractor.py:136-143,
// layer1) in four of these per image.  As an implicit GEMM on the general kernel (gemm.hip, 64-wide tile) every k-tile re-stages a
// 256-pixel x 64-channel operand block for each of the nine taps: 64 FLOP per operand byte, and the launches ran at 460 TFLOP/s
// with the L2 -> LDS path, not the matrix cores, setting the pace (a plain [M, 576] x [64, 576]^T GEMM on the same tile: 274 TFLOP/s
// at 4.3 TB/s of operand stream; profiles/r03_conv64.md).  Here an input pixel is staged ONCE and read from LDS by all nine taps:
//   * a persistent workgroup (8 waves) walks runs of image rows two output rows per step (tile = 2 rows x 128 pixel slots x 64
//     channels; W <= 126); an image is cut into as many runs as balance the grid;
//   * LDS holds a ring of 8 input rows x 128 pixel slots x 128 bytes (slot 0 and W + 1 = the zero padding, written by the LDS-DMA's
//     out-of-range reads; rows -1 and H likewise), filled two rows ahead with buffer_load ... lds; 16-byte chunks XOR-swizzled by
//     (slot >> 1) & 7 like every other operand image of this library;
//   * the WEIGHTS live in registers: waves form a 4 (pixels) x 2 (channels) grid, a wave keeps its 32 output channels' 9 x 64
//     weights as 36 MFMA fragments (144 VGPRs) for the whole launch -- LDS bandwidth goes to the pixels only (2 fragment reads per 4
//     MFMAs: 128 B/clk per CU at full MFMA rate, i.e. LDS and matrix cores are balanced);
//   * the four resident input rows of a step feed both output rows: 48 fragment reads for 144 MFMAs;
//   * one barrier per step; the step's 8 stores are ALWAYS issued (invalid pixels store out of the descriptor's range), so the
//     counted wait in front of the next barrier (`vmcnt(8)`: the rows requested one step ago have landed) is exact.
// Epilogues: fp32 output + per-(image, channel) sum / sum of squares for InstanceNorm (fnet: reduced in registers / LDS per run, one
// atomic per channel and run -- stored without atomics when a workgroup owns the image), or bf16 output with ReLU / skip + ReLU
// (cnet, BatchNorm folded into the weights).
#include "common.h"

namespace {

constexpr int C64_R = 8, C64_RS = 128;                 // ring rows, pixel slots per row
constexpr int C64_ROW_BYTES = C64_RS * 128;            // 16 KiB
constexpr int C64_LDS = C64_R * C64_ROW_BYTES + 8 * 32 * 2 * 4 + 64 * 4;   // + the moments exchange (8 waves x 32 channels x 2) + the bias
typedef __attribute__((address_space(3))) void* c64_lptr_t;

struct Conv64Params {
    const bf16_t* in;      // [n_img, H, W, 64]
    const bf16_t* w;       // [64, 9 * 64]: k = tap * 64 + channel
    const float* bias;     // [64]
    float* out_f32;        // MODE 0: [n_img * H * W, 64]
    float* stats;          // MODE 0: [n_img, 64, 2] (sum, sum of squares)
    float* stats_part;     // parts > 1: [n_img * parts, 64, 2] per-run partial moments
    bf16_t* out_bf16;      // MODE 1
    const bf16_t* resid;   // MODE 1, nullable
    int n_img, H, W, relu, post_relu;
    int parts, rows_per_part;   // an image is cut into `parts` runs of `rows_per_part` (even) output rows: one workgroup walks a run
};

template <int MODE>
__global__ __launch_bounds__(512) void conv3x3_c64_kernel(const Conv64Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char ring[];
    float* const xch = reinterpret_cast<float*>(ring + C64_R * C64_ROW_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int H = p.H, W = p.W;

    // ---- this wave's weights: fragments [tap][32-deep half][16-channel block] and bias
    bf16x8 wf[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < 2; i++)
                wf[t][ks][i] = *reinterpret_cast<const bf16x8*>(p.w + (wn * 32 + i * 16 + fr) * 576 + t * 64 + ks * 32 + fg * 8);
    float* const bias_s = xch + 8 * 32 * 2;                // the bias, read back from LDS in every epilogue (8 registers the weight fragments need)
    if (tid < 64) bias_s[tid] = p.bias[tid];

    // ---- staging: wave w issues pieces 2 w, 2 w + 1 of a row (8 pixel slots x 128 bytes each).  Lane l of a piece: slot 8 piece + (l >> 3),
    // LDS chunk position l & 7 <- global chunk (l & 7) ^ ((slot >> 1) & 7) of pixel slot - 1; slots outside [1, W] read out of range (zeros)
    unsigned dma_off[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int slot = (wave * 2 + q) * 8 + (lane >> 3), x = slot - 1, c = (lane & 7) ^ ((slot >> 1) & 7);
        dma_off[q] = (x >= 0 && x < W) ? (unsigned)(x * 64 + c * 8) * 2u : 0x80000000u;
    }
    // ---- fragment reads: pixel px = 32 wm + 16 j + fr of the output row reads slot px + kx (kx = 0..2) of ring row y + ky - 1
    // (fragment j = 1 sits 16 slots = 2048 bytes further with the same swizzle term; the second 32-deep half is chunk ^ 4 = byte ^ 64:
    // three registers instead of twelve.  Pixels beyond the row read whatever slot that lands on: their results are never stored.)
    unsigned x_off[3];
    int px[2];
#pragma unroll
    for (int j = 0; j < 2; j++) px[j] = wm * 32 + j * 16 + fr;
#pragma unroll
    for (int kx = 0; kx < 3; kx++) x_off[kx] = (unsigned)((px[0] + kx) * 128 + ((fg ^ (((px[0] + kx) >> 1) & 7)) << 4));

    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
    typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2_t;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int ES = MODE == 0 ? 4 : 2;
    // output lane offsets inside an image row (channel block i, pixel fragment j); pixels beyond the row store out of range
    unsigned o_off[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) o_off[i][j] = px[j] < W ? (unsigned)(px[j] * 64 + wn * 32 + i * 16 + fg * 4) * ES : OOB;

    const int n_units = p.n_img * p.parts;
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int img = unit / p.parts, y0 = (unit - img * p.parts) * p.rows_per_part;
        if (y0 >= H) continue;
        const int y1 = y0 + p.rows_per_part < H ? y0 + p.rows_per_part : H;
        const bf16_t* const ibase = p.in + (int64_t)img * H * W * 64;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ibase), 0, H * W * 128, 0x00020000);
        const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(MODE == 0 ? (void*)(p.out_f32 + (int64_t)img * H * W * 64) : (void*)(p.out_bf16 + (int64_t)img * H * W * 64), 0,
                                                             H * W * 64 * ES, 0x00020000);
        const auto rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.resid ? p.resid + (int64_t)img * H * W * 64 : ibase), 0, H * W * 128, 0x00020000);
#define C64_ISSUE(yy)                                                                                                          \
        {                                                                                                                      \
            const int so_ = ((yy) >= 0 && (yy) < H) ? (yy) * W * 128 : 0x7FFFFF00;      /* rows -1 and H: out of range, zeros */  \
            char* const dst_ = ring + ((yy) & (C64_R - 1)) * C64_ROW_BYTES + (wave * 2) * 1024;                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (c64_lptr_t)dst_, 16, dma_off[0], so_, 0, 0);                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (c64_lptr_t)(dst_ + 1024), 16, dma_off[1], so_, 0, 0);              \
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);               // (my rows of the previous unit have landed / been stored: their ring rows are reused)
        __builtin_amdgcn_s_barrier();                     // the previous unit's last rows have been consumed by every wave
        C64_ISSUE(y0 - 1)
        C64_ISSUE(y0)
        C64_ISSUE(y0 + 1)
        C64_ISSUE(y0 + 2)
        f32x4 s1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, s2[2] = {s1[0], s1[0]};
        // Two output rows per step: input rows y - 1 .. y + 2 are resident, each x fragment feeds the taps of both output rows that touch it
        // (48 fragment reads for 144 MFMAs: the matrix cores, not LDS, bound the step).
        for (int y = y0; y < y1; y += 2) {
            // rows y + 1, y + 2 (4 pieces, issued one step ago) have landed when only what was issued behind them is outstanding: the 8
            // stores of the previous step (always issued: invalid pixels / rows store out of range).  First step: everything.
            if (y == y0) __builtin_amdgcn_s_waitcnt(0x0F70);
            else __builtin_amdgcn_s_waitcnt(0x0F78);
            __builtin_amdgcn_s_barrier();                 // everybody's pieces are in LDS; everybody is done with the previous step's rows
            bf16x4 rres[2][2][2];
            if constexpr (MODE == 1) {
                if (p.resid) {                            // (requested BEFORE this step's DMA: vmcnt completes in order)
#pragma unroll
                    for (int o = 0; o < 2; o++)
#pragma unroll
                        for (int i = 0; i < 2; i++)
#pragma unroll
                            for (int j = 0; j < 2; j++) {
                                const unsigned off = (y + o < y1 && o_off[i][j] != OOB) ? o_off[i][j] + (unsigned)((y + o) * W * 64 * 2) : OOB;
                                rres[o][i][j] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rrsrc, off, 0, 0));
                            }
                }
            }
            C64_ISSUE(y + 3)
            C64_ISSUE(y + 4)
            f32x4 acc[2][2][2];
#pragma unroll
            for (int o = 0; o < 2; o++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[o][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const char* const rowp = ring + ((y + rr - 1) & (C64_R - 1)) * C64_ROW_BYTES;
#pragma unroll
                for (int kx = 0; kx < 3; kx++)
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) {
                        const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(rowp + (x_off[kx] ^ (ks * 64)));
                        const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(rowp + (x_off[kx] ^ (ks * 64)) + 2048);
#pragma unroll
                        for (int o = 0; o < 2; o++) {
                            const int ky = rr - o;
                            if (ky < 0 || ky > 2) continue;
#pragma unroll
                            for (int i = 0; i < 2; i++) {
                                acc[o][i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky * 3 + kx][ks][i], x0, acc[o][i][0], 0, 0, 0);
                                acc[o][i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky * 3 + kx][ks][i], x1, acc[o][i][1], 0, 0, 0);
                            }
                        }
                    }
            }
            // ---- epilogue: 8 stores, always issued
            f32x4 b4[2];
#pragma unroll
            for (int i = 0; i < 2; i++) b4[i] = *reinterpret_cast<const f32x4*>(bias_s + wn * 32 + i * 16 + fg * 4);
#pragma unroll
            for (int o = 0; o < 2; o++) {
                const bool row_ok = y + o < y1;
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const bool ok = row_ok && o_off[i][j] != OOB;
                        const unsigned off = ok ? o_off[i][j] + (unsigned)((y + o) * W * 64 * ES) : OOB;
                        f32x4 v = acc[o][i][j] + b4[i];
                        if constexpr (MODE == 0) {
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), orsrc, off, 0, 0);
                            // (selects, not a branch -- a branch here makes hipcc duplicate the store into both arms -- and not a multiply
                            // by 0 / 1: pixel slots beyond the row multiply whatever LDS holds there, possibly NaN)
                            const f32x4 vz = {ok ? v[0] : 0.f, ok ? v[1] : 0.f, ok ? v[2] : 0.f, ok ? v[3] : 0.f};
                            s1[i] += vz;
                            s2[i] += vz * vz;
                        } else {
                            if (p.relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                            if (p.resid) {
                                const bf16x4 r = rres[o][i][j];
                                v += f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
                            }
                            if (p.post_relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                            const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, pk), orsrc, off, 0, 0);
                        }
                    }
            }
        }
#undef C64_ISSUE
        if constexpr (MODE == 0) {
            // moments of this unit's rows: over the 16 pixel lanes of a fragment, then over the four pixel waves (LDS), added to the image's
            // (sum, sum of squares) -- stored per unit, added in unit order by a second pass (r5: was one atomic per channel and unit)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float a_ = s1[i][e], b_ = s2[i][e];
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) { a_ += __shfl_xor(a_, off); b_ += __shfl_xor(b_, off); }
                    s1[i][e] = a_; s2[i][e] = b_;
                }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();                 // (the exchange area is free: the previous unit's reads are done)
            if (fr == 0) {
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        xch[(wave * 32 + i * 16 + fg * 4 + e) * 2] = s1[i][e];
                        xch[(wave * 32 + i * 16 + fg * 4 + e) * 2 + 1] = s2[i][e];
                    }
            }
            __syncthreads();
            if (tid < 128) {                              // 64 channels x (sum, sum of squares)
                const int ch = tid >> 1, which = tid & 1, wn_ = ch >> 5, c32 = ch & 31;
                float t = 0.f;
#pragma unroll
                for (int m_ = 0; m_ < 4; m_++) t += xch[((wn_ * 4 + m_) * 32 + c32) * 2 + which];
                // the workgroup owns the image: stored; else stored to this run's slot, added in run order by launch_stats_finish_parts (r5: no atomics)
                if (p.parts == 1) p.stats[((int64_t)img * 64 + ch) * 2 + which] = t;
                else p.stats_part[((int64_t)unit * 64 + ch) * 2 + which] = t;      // (unit = img * parts + run)
            }
        }
    }
#endif
}


// ---------------------------------------------------------------------------------------
// The stem (7x7, stride 2, 3 -> 64 channels; extractor.py:128-130 with the input scaling of xraft.py:105-106) without its HBM
// round trip.  Rounds 1-3a ran it as raft_stem_pack_kernel (a 128-channel "space to depth" image per half-resolution pixel: bf16
// hi | lo pairs of x - 127.5 for a 4 x 2 input-column window, 3.2 MB per frame written) + a 4 x 1 implicit-GEMM convolution that read
// that image back four times: 8.5 + 8.7 ms per 31-clip RAFT batch.  Here a persistent workgroup walks runs of output rows and builds
// the SAME packed rows (same K order, same packed weights: vtgb.h) straight into an LDS ring of 5 rows x W/2 pixels x 256 bytes, one
// row ahead of the row whose 4 vertical taps it multiplies; the wave's 32 x 512 weights live in registers (128 VGPRs) as in
// conv3x3_c64_kernel.  16-byte chunks of a pixel's 256 bytes are XOR-swizzled by (X & 15): the 16 pixels of a fragment read hit 16
// different chunk positions.
// ---------------------------------------------------------------------------------------
constexpr int ST_R = 5;
struct StemParams {
    const float* img;      // [n_img, 3, H, W] raw 0..255 (or CLIP-normalised floats: see raft_enc.hip)
    const bf16_t* w;       // [64, 512]: k = chunk (hi | lo) * 256 + tY * 64 + (dX * 12 + py * 6 + px * 3 + c)
    const float* bias;
    float* out_f32;        // MODE 0: [n_img * H2 * W2, 64] + moments
    float* stats;
    float* stats_part;
    bf16_t* out_bf16;      // MODE 1: relu -> bf16
    int n_img, H, W, relu;
    int parts, rows_per_part;
};

template <int MODE>
__global__ __launch_bounds__(512) void stem7x7_kernel(const StemParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int H = p.H, W = p.W, H2 = H >> 1, W2 = W >> 1;
    const int row_bytes = W2 * 256;
    float* const xch = reinterpret_cast<float*>(ring + ST_R * row_bytes);
    float* const bias_s = xch + 8 * 32 * 2;
    if (tid < 64) bias_s[tid] = p.bias[tid];
    // the 16 zero columns of every 64 (48 real values per tap): chunks 6, 7 (hi) and 14, 15 (lo) of every pixel, written once
    for (int i = tid; i < ST_R * W2 * 4; i += 512) {
        const int pxl = i >> 2, X = pxl % W2, c = 6 + (i & 1) + ((i & 2) << 2);      // 6, 7, 14, 15
        *reinterpret_cast<uint4*>(ring + pxl * 256 + ((c ^ (X & 15)) << 4)) = make_uint4(0, 0, 0, 0);
    }
    // ---- weights: 16 k-steps of 32 (s = chunk * 8 + tY * 2 + half) x two 16-channel blocks
    bf16x8 wf[16][2];
#pragma unroll
    for (int s_ = 0; s_ < 16; s_++)
#pragma unroll
        for (int i = 0; i < 2; i++) wf[s_][i] = *reinterpret_cast<const bf16x8*>(p.w + (wn * 32 + i * 16 + fr) * 512 + s_ * 32 + fg * 8);
    // ---- building a packed row: thread = (pixel X, column pair dX): 12 values (2 rows x 2 columns x 3 channels) as hi / lo bf16
    const bool builder = tid < W2 * 4;
    const int bX = tid >> 2, bdX = tid & 3, bcol = 2 * (bX + bdX - 2);        // first of the two input columns
    const bool bcol_ok = builder && bcol >= 0 && bcol + 1 < W;                // (W even, bcol even: the pair is inside or outside together)
    const int bcol_c = bcol < 0 ? 0 : bcol + 1 < W ? bcol : W - 2;            // clamped: the loads are unconditional, ST_WRITE zeroes what is outside
    unsigned b_off[3];                                                         // LDS byte offsets of the three 8-byte pieces (hi; lo = chunk + 8)
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int off = bdX * 24 + q * 8;
        b_off[q] = (unsigned)(bX * 256 + (((off >> 4) ^ (bX & 15)) << 4) + (off & 8));
    }
    // ---- fragment reads: pixel X = 32 wm + 16 j + fr, chunk (hl * 8 + half * 4 + fg) ^ (X & 15)
    int px[2];
    unsigned x_off[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        px[j] = wm * 32 + j * 16 + fr;
        const int pc = px[j] < W2 ? px[j] : 0;
        x_off[j] = (unsigned)(pc * 256 + ((fg ^ (pc & 15)) << 4));
    }
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
    typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2_t;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int ES = MODE == 0 ? 4 : 2;
    unsigned o_off[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) o_off[i][j] = px[j] < W2 ? (unsigned)(px[j] * 64 + wn * 32 + i * 16 + fg * 4) * ES : OOB;

    const int n_units = p.n_img * p.parts;
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const int img = unit / p.parts, y0 = (unit - img * p.parts) * p.rows_per_part;
        if (y0 >= H2) continue;
        const int y1 = y0 + p.rows_per_part < H2 ? y0 + p.rows_per_part : H2;
        const float* const ibase = p.img + (int64_t)img * 3 * H * W;
        const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(MODE == 0 ? (void*)(p.out_f32 + (int64_t)img * H2 * W2 * 64) : (void*)(p.out_bf16 + (int64_t)img * H2 * W2 * 64),
                                                             0, H2 * W2 * 64 * ES, 0x00020000);
        float2 ld[6];
        // packed row Yp <- input rows 2 Yp, 2 Yp + 1 (zeros outside the image)
#define ST_LOAD(Yp)      /* unconditional loads from clamped addresses (conditional ones are serialised by hipcc: one vmcnt(0) each) */ \
        {                                                                                                    \
            const int yc_ = (Yp) < 0 ? 0 : (Yp) >= H2 ? H2 - 1 : (Yp);                                       \
            _Pragma("unroll") for (int py = 0; py < 2; py++)                                                 \
                _Pragma("unroll") for (int c = 0; c < 3; c++)                                                \
                    ld[py * 3 + c] = *reinterpret_cast<const float2*>(ibase + ((int64_t)c * H + 2 * yc_ + py) * W + bcol_c); \
        }
#define ST_WRITE(Yp)                                                                                         \
        if (builder) {                                                                                       \
            const bool in_ = bcol_ok && (Yp) >= 0 && (Yp) < H2;                                              \
            bf16_t hi[12], lo[12];                                                                           \
            _Pragma("unroll") for (int py = 0; py < 2; py++)                                                 \
                _Pragma("unroll") for (int c = 0; c < 3; c++) {                                              \
                    const float d0 = in_ ? ld[py * 3 + c].x - 127.5f : 0.f, d1 = in_ ? ld[py * 3 + c].y - 127.5f : 0.f; \
                    const float h0_ = bf16_round(d0), h1_ = bf16_round(d1);                                  \
                    hi[py * 6 + c] = (bf16_t)h0_; lo[py * 6 + c] = (bf16_t)(d0 - h0_);                       \
                    hi[py * 6 + 3 + c] = (bf16_t)h1_; lo[py * 6 + 3 + c] = (bf16_t)(d1 - h1_);               \
                }                                                                                            \
            char* const rb_ = ring + ((((Yp) % ST_R) + ST_R) % ST_R) * row_bytes;                            \
            _Pragma("unroll") for (int q = 0; q < 3; q++) {                                                  \
                *reinterpret_cast<bf16x4*>(rb_ + b_off[q]) = bf16x4{hi[q * 4], hi[q * 4 + 1], hi[q * 4 + 2], hi[q * 4 + 3]}; \
                *reinterpret_cast<bf16x4*>(rb_ + (b_off[q] ^ 128)) = bf16x4{lo[q * 4], lo[q * 4 + 1], lo[q * 4 + 2], lo[q * 4 + 3]}; \
            }                                                                                                \
        }
        __syncthreads();                                       // the previous unit's rows have been consumed (and the one-time zero / bias writes are done)
        for (int yy = y0 - 2; yy <= y0 + 1; yy++) {            // the four rows of the first step (one load latency each: once per unit)
            ST_LOAD(yy)
            ST_WRITE(yy)
        }
        f32x4 s1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, s2[2] = {s1[0], s1[0]};
        for (int y = y0; y < y1; y++) {
            // rows y - 2 .. y + 1 are complete (LDS writes only: __syncthreads would also wait for the previous row's output stores,
            // a memory round trip per row); everybody is done with row y - 3's slot
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            ST_LOAD(y + 2)                                     // in flight under this row's MFMAs
            f32x4 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tY = 0; tY < 4; tY++) {
                const char* const rowp = ring + ((y + tY - 2 + ST_R) % ST_R) * row_bytes;
#pragma unroll
                for (int hl = 0; hl < 2; hl++)
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        const int s_ = hl * 8 + tY * 2 + half;
                        const unsigned cx = (unsigned)((hl * 8 + half * 4) << 4);      // chunk index bits above fg: XORed in (16-byte units)
                        const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(rowp + (x_off[0] ^ cx));
                        const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(rowp + (x_off[1] ^ cx));
#pragma unroll
                        for (int i = 0; i < 2; i++) {
                            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s_][i], x0, acc[i][0], 0, 0, 0);
                            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s_][i], x1, acc[i][1], 0, 0, 0);
                        }
                    }
            }
            f32x4 b4[2];
#pragma unroll
            for (int i = 0; i < 2; i++) b4[i] = *reinterpret_cast<const f32x4*>(bias_s + wn * 32 + i * 16 + fg * 4);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const bool ok = o_off[i][j] != OOB;
                    const unsigned off = ok ? o_off[i][j] + (unsigned)(y * W2 * 64 * ES) : OOB;
                    f32x4 v = acc[i][j] + b4[i];
                    if constexpr (MODE == 0) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), orsrc, off, 0, 0);
                        const f32x4 vz = {ok ? v[0] : 0.f, ok ? v[1] : 0.f, ok ? v[2] : 0.f, ok ? v[3] : 0.f};
                        s1[i] += vz;
                        s2[i] += vz * vz;
                    } else {
                        if (p.relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                        const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, pk), orsrc, off, 0, 0);
                    }
                }
            ST_WRITE(y + 2)                                    // slot of row y - 3
        }
#undef ST_LOAD
#undef ST_WRITE
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float a_ = s1[i][e], b_ = s2[i][e];
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) { a_ += __shfl_xor(a_, off); b_ += __shfl_xor(b_, off); }
                    s1[i][e] = a_; s2[i][e] = b_;
                }
            __syncthreads();
            if (fr == 0) {
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        xch[(wave * 32 + i * 16 + fg * 4 + e) * 2] = s1[i][e];
                        xch[(wave * 32 + i * 16 + fg * 4 + e) * 2 + 1] = s2[i][e];
                    }
            }
            __syncthreads();
            if (tid < 128) {
                const int ch = tid >> 1, which = tid & 1, wn_ = ch >> 5, c32 = ch & 31;
                float t = 0.f;
#pragma unroll
                for (int m_ = 0; m_ < 4; m_++) t += xch[((wn_ * 4 + m_) * 32 + c32) * 2 + which];
                if (p.parts == 1) p.stats[((int64_t)img * 64 + ch) * 2 + which] = t;
                else p.stats_part[((int64_t)unit * 64 + ch) * 2 + which] = t;      // (unit = img * parts + run)
            }
        }
    }
#endif
}

}   // namespace

// x [n_img, H, W, 64] bf16 -> 3x3 / stride 1 / pad 1 convolution with w [64, 576] (+ bias): fp32 [.., 64] + moments (out_f32 != NULL) or bf16
bool conv3x3_c64_supported(int H, int W) { return W >= 16 && W <= C64_RS - 2 && H >= 1; }

size_t conv64_stats_part_floats(int n_img) { return ((size_t)n_img + 4 * (size_t)cu_count()) * 2 * 64 * 2; }   // n_img * parts < 2 * (n_img + 4 CUs)

int launch_conv3x3_c64(const void* in, const void* w, const float* bias, float* out_f32, float* stats, float* stats_part, void* out_bf16, const void* resid,
                       int n_img, int H, int W, int relu, int post_relu, hipStream_t s) {
    VTGB_REQUIRE(conv3x3_c64_supported(H, W), VTGB_EUNSUPPORTED, "conv3x3_c64: W=%d outside [16, %d]", W, C64_RS - 2);
    VTGB_REQUIRE((int64_t)H * W * 128 < 0x7FFFFF00ll, VTGB_EUNSUPPORTED, "conv3x3_c64: image too large");
    Conv64Params p;
    p.in = (const bf16_t*)in; p.w = (const bf16_t*)w; p.bias = bias; p.out_f32 = out_f32; p.stats = stats; p.stats_part = stats_part; p.out_bf16 = (bf16_t*)out_bf16;
    p.resid = (const bf16_t*)resid; p.n_img = n_img; p.H = H; p.W = W; p.relu = relu; p.post_relu = post_relu;
    // runs of rows: enough of them that the persistent grid is balanced (>= 4 per workgroup), each an even number of rows
    int parts = 1;
    while ((int64_t)n_img * parts < 4 * cu_count() && H / (parts * 2) >= 8) parts *= 2;
    p.rows_per_part = 2 * ((H + 2 * parts - 1) / (2 * parts));
    p.parts = (H + p.rows_per_part - 1) / p.rows_per_part;
    const int64_t units = (int64_t)n_img * p.parts;
    const int grid = units < cu_count() ? (int)units : cu_count();
    if (out_f32 && p.parts > 1) VTGB_REQUIRE(stats_part && (size_t)units * 128 <= conv64_stats_part_floats(n_img), VTGB_EINVAL, "conv3x3_c64: partial-moment scratch missing");
    const double flops = 2.0 * n_img * H * W * 64.0 * 576.0;
    ProfScope prof(VTGB_PROF_CONV, flops, s, flops);
    static DeviceOnce a0, a1;
    if (out_f32) {
        VTGB_REQUIRE(stats, VTGB_EINVAL, "conv3x3_c64: fp32 output needs the moments buffer");
        VTGB_FUNC_LDS_ONCE(a0, conv3x3_c64_kernel<0>, C64_LDS);
        hipLaunchKernelGGL(conv3x3_c64_kernel<0>, dim3(grid), dim3(512), C64_LDS, s, p);
        if (p.parts > 1) VTGB_TRY(launch_stats_finish_parts(stats_part, stats, n_img, p.parts, 64, s));
    } else {
        VTGB_FUNC_LDS_ONCE(a1, conv3x3_c64_kernel<1>, C64_LDS);
        hipLaunchKernelGGL(conv3x3_c64_kernel<1>, dim3(grid), dim3(512), C64_LDS, s, p);
    }
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// the stem on raw frames: img [n_img, 3, H, W] fp32 -> [n_img, H/2, W/2, 64]; packed weights [64, 512] (vtgb.h: vtgb_raft_encoder)
bool stem7x7_supported(int H, int W) { return (H % 2) == 0 && (W % 2) == 0 && W / 2 >= 16 && W / 2 <= 112 && H / 2 >= 4; }

int launch_stem7x7(const float* img, const void* w, const float* bias, float* out_f32, float* stats, float* stats_part, void* out_bf16, int n_img, int H, int W,
                   int relu, hipStream_t s) {
    VTGB_REQUIRE(stem7x7_supported(H, W), VTGB_EUNSUPPORTED, "stem7x7: %d x %d unsupported", H, W);
    const int H2 = H / 2, W2 = W / 2;
    StemParams p;
    p.img = img; p.w = (const bf16_t*)w; p.bias = bias; p.out_f32 = out_f32; p.stats = stats; p.stats_part = stats_part; p.out_bf16 = (bf16_t*)out_bf16;
    p.n_img = n_img; p.H = H; p.W = W; p.relu = relu;
    int parts = 1;
    while ((int64_t)n_img * parts < 4 * cu_count() && H2 / (parts * 2) >= 14) parts *= 2;
    p.rows_per_part = (H2 + parts - 1) / parts;
    p.parts = (H2 + p.rows_per_part - 1) / p.rows_per_part;
    const int64_t units = (int64_t)n_img * p.parts;
    const int grid = units < cu_count() ? (int)units : cu_count();
    if (out_f32 && p.parts > 1) VTGB_REQUIRE(stats_part && (size_t)units * 128 <= conv64_stats_part_floats(n_img), VTGB_EINVAL, "stem7x7: partial-moment scratch missing");
    const int lds = ST_R * W2 * 256 + 8 * 32 * 2 * 4 + 64 * 4;
    const double flops = 2.0 * n_img * H2 * W2 * 64.0 * 147.0, exec = 2.0 * n_img * H2 * W2 * 64.0 * 512.0;
    ProfScope prof(VTGB_PROF_CONV, flops, s, exec);
    static DeviceOnce a0, a1;
    if (out_f32) {
        VTGB_REQUIRE(stats, VTGB_EINVAL, "stem7x7: fp32 output needs the moments buffer");
        VTGB_FUNC_LDS_ONCE(a0, stem7x7_kernel<0>, ST_R * 112 * 256 + 8 * 32 * 2 * 4 + 64 * 4);
        hipLaunchKernelGGL(stem7x7_kernel<0>, dim3(grid), dim3(512), lds, s, p);
        if (p.parts > 1) VTGB_TRY(launch_stats_finish_parts(stats_part, stats, n_img, p.parts, 64, s));
    } else {
        VTGB_FUNC_LDS_ONCE(a1, stem7x7_kernel<1>, ST_R * 112 * 256 + 8 * 32 * 2 * 4 + 64 * 4);
        hipLaunchKernelGGL(stem7x7_kernel<1>, dim3(grid), dim3(512), lds, s, p);
    }
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
