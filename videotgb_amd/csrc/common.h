// common.h -- internal declarations shared by the libvtgb.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "vtgb.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// v rounded (to nearest even) to bf16 precision, as a float -- on the bits: hipcc may keep the excess precision of a float -> __bf16 ->
// float cast round trip, which would turn a hi | lo split (lo = v - hi) into lo = 0
#if defined(__HIPCC__)
__device__ __forceinline__ float bf16_round(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7F800000u) != 0x7F800000u) u += 0x7FFFu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xFFFF0000u);
}
#endif

// ---------------------------------------------------------------------------------------
// error plumbing (thread-local message, negative return codes; include/vtgb.h)
// ---------------------------------------------------------------------------------------
void vtgb_set_error(const char* fmt, ...);

#define VTGB_REQUIRE(cond, code, ...)  \
    do {                               \
        if (!(cond)) {                 \
            vtgb_set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

#define VTGB_HIP(expr)                                                              \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            vtgb_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return VTGB_EHIP;                                                       \
        }                                                                           \
    } while (0)

#define VTGB_TRY(expr)        \
    do {                      \
        int _r = (expr);      \
        if (_r != VTGB_OK) return _r; \
    } while (0)

// ---------------------------------------------------------------------------------------
// logical row -> physical row.  seg_rows == 0: identity.  Otherwise the logical rows are the
// concatenation of segments of seg_rows rows; segment s starts at physical row
// s * seg_stride + seg_off.  (Q-Former query rows / text rows of each frame; ViT patch rows
// interleaved with the cls row; position-embedding rows broadcast over frames.)
// ---------------------------------------------------------------------------------------
struct RowMap {
    int32_t seg_rows;
    int64_t seg_stride;
    int64_t seg_off;
};
static inline RowMap rowmap_identity() { return RowMap{0, 0, 0}; }
static inline RowMap rowmap(int seg_rows, int64_t seg_stride, int64_t seg_off) {
    return RowMap{seg_rows, seg_stride, seg_off};
}
__host__ __device__ static inline int64_t map_row(const RowMap& m, int64_t r) {
    if (m.seg_rows == 0) return r;
    const int32_t r32 = (int32_t)r, seg = r32 / m.seg_rows;   // GemmDesc::M is an int32: 32-bit division
    return (int64_t)seg * m.seg_stride + m.seg_off + (r32 - seg * m.seg_rows);
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// "once per device" flag for per-function attributes (hipFuncSetAttribute is per device): need() is true until mark() has
// been called while that device was current.  Thread-safe: a second thread may repeat the idempotent attribute call, it can
// never skip it before it has completed.
#include <atomic>
struct DeviceOnce {
    std::atomic<uint64_t> done{0};
    static uint64_t bit() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        return 1ull << (dev & 63);
    }
    bool need() const { return !(done.load(std::memory_order_acquire) & bit()); }
    void mark() { done.fetch_or(bit(), std::memory_order_acq_rel); }
};
#define VTGB_FUNC_LDS_ONCE(flag, kernel, bytes)                                                                              \
    do {                                                                                                                     \
        if ((flag).need()) {                                                                                                 \
            VTGB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            (flag).mark();                                                                                                   \
        }                                                                                                                    \
    } while (0)
static inline size_t dtype_size(int dtype) { return dtype == VTGB_BF16 ? 2 : 4; }

// bump allocator over the caller's workspace
struct Workspace {
    char* base;
    size_t size, used;
    bool dry;  // size query: only count
    Workspace(void* p, size_t n) : base((char*)p), size(n), used(0), dry(p == nullptr) {}
    void* take(size_t bytes) {
        size_t off = align_up(used, 256);
        used = off + bytes;
        return dry ? nullptr : (void*)(base + off);
    }
    bool ok() const { return dry || used <= size; }
};

// ---------------------------------------------------------------------------------------
// kernel launchers (each returns VTGB_OK or a negative code with the message set)
// ---------------------------------------------------------------------------------------
struct GemmDesc {
    int dtype, M, N, K, epi;
    const void* A;
    int64_t lda;
    RowMap a_map;
    const void* W;
    int64_t ldw;
    const float* bias;
    const float* resid;
    int64_t ldr;
    RowMap r_map;
    void* out;
    int64_t ldo;
    RowMap o_map;
    // ---- extensions used by the RAFT update block (zero-initialised = plain GEMM)
    int act;            // applied to acc + bias before the store: 0 none, 1 relu, 2 sigmoid
    float out_scale;    // 0 -> 1
    // implicit-GEMM convolution (stride 1, "same" zero padding) over NHWC activations: A rows are
    // pixels of conv_H x conv_W images, K = KH*KW*Cin ordered (64-channel chunk, tap, channel in chunk), Cin % 64 == 0.
    // Channels [0, split_c) come from A (row stride lda), channels [split_c, Cin) from A2 (lda2):
    // a virtual concat.  conv_KH == 0 -> not a convolution.
    int conv_H, conv_W, conv_KH, conv_KW, conv_Cin, conv_split;
    int conv_stride, conv_Hi, conv_Wi;   // 0 -> stride 1, input grid == output grid; padding is always KH/2, KW/2
    const void* A2;
    int64_t lda2;
    const void* zero_page;   // >= 256 B of zeros (out-of-image taps read it)
    // EPI_GRU: h' = (1 - z) * h + z * tanh(acc + bias): h fp32 in `resid` (ldr), z bf16 in `aux` (ldaux);
    // h' written fp32 to `out` (ldo) and bf16 to `out2` (ldo2)
    const void* aux;
    int64_t ldaux;
    void* out2;
    int64_t ldo2;
    // EPI_STORE_F32 (N % 4 == 0): per column the sum and sum of squares of the stored values of every 256-row tile are
    // written to col_stats[(m_tile * N + n) * 4 + {sum a, sq a, sum b, sq b}] (a = the image of the tile's first row, row / stats_rows;
    // b = the next image, if the tile reaches into it) -- InstanceNorm statistics without a second pass over the output.
    // stats_rows >= 256; launch_stats_finish_tiles adds the slots of an image in tile order (r5: deterministic; rounds 1-4 accumulated
    // with atomics, whose order -- and with it the last bits of every normalised feature -- changed from run to run).
    float* col_stats;
    int stats_rows;
    // EPI_STORE (bf16, whole-row stores): columns n >= gate_from are not stored to `out` but multiplied by
    // aux[m][n - gate_from] (bf16, ldaux) and stored to out2[m][n - gate_from] (bf16, ldo2) -- SepConvGRU's
    // r * h straight from the z|r convolution.  gate_from % 8 == 0; 0 = off.
    int gate_from;
    // EPI_STORE (bf16, whole-row stores): out = [relu](act(acc + bias) + resid_bf16[m][n]) -- a ResidualBlock's
    // tail (x + relu(conv)) -> relu with the skip operand in bf16 (row stride ldrb, 8-aligned)
    const void* resid_bf16;
    int64_t ldrb;
    int post_relu;
    // bf16 large kernel: the accumulators start at init_bf16[m][n] (bf16, row stride ldinit, 4-aligned) in addition to the
    // bias -- a per-(row, column) constant folded out of the k-loop (RAFT: the loop-invariant `inp` third of the GRU
    // convolutions, computed once per pair instead of once per refinement iteration)
    const void* init_bf16;
    int64_t ldinit;
    // fragment order (bf16 large kernel, EPI_STORE without tails / start maps): instead of rows, a tile's accumulators are kept
    // as the MFMA leaves them -- element ((((tile * 8 + wave) * NX + j) * 4 + i) * 64 + lane) is that lane's 4 values (8 bytes)
    // of accumulator block (i, j) -- so one wave instruction moves 512 contiguous bytes.  frag_out: `out` is written in this
    // order (activated, bf16; size m_tiles * n_tiles * 256 * tile width * 2 bytes).  init_frag: init_bf16 is read in this order
    // (it must come from a frag_out launch with the same M, N and tile shape).  RAFT's loop-invariant start maps: the row-major
    // form cost 32 loads of 16 rows x 32 bytes per lane in front of every GRU tile's first MFMA.
    int frag_out, init_frag;
    // filled by launch_conv_gemm: round-up magic numbers for n / (conv_H * conv_W) and n / conv_W, n < 2^31 (the per-lane
    // pixel decomposition of every tile's prologue: q = (umulhi(n, mul) + n) >> sh)
    uint32_t div_hw_mul, div_hw_sh, div_w_mul, div_w_sh;
    // launch timing (vtgb_prof_*): ALGORITHMIC FLOPs of this launch when they differ from the executed 2 M N K -- the GRU
    // convolutions with the hoisted `inp` third are credited with the full 384-channel convolution the reference computes in
    // every iteration; the once-per-call start-map convolutions that carry the hoisted part are credited with 0 (< 0 here).
    // 0 = executed.  The executed FLOPs are accumulated separately (vtgb_prof_executed_flops).
    double algo_flops;
    // EPI_STORE, 256-wide tile, N == 256 (one n-tile), bf16 staged store: instead of storing the activated tile, multiply it
    // by tail_w [32][256] (bf16, a following 1x1 convolution with <= 32 outputs, no bias) while it sits in LDS and store
    // tail_out[m][0..32) (fp32, row stride ldtail) -- RAFT's FlowHead: relu(conv1) never leaves the CU, only the 18 per-tap
    // partial products of conv2 do.  `out` is not written.
    const void* tail_w;
    float* tail_out;
    int64_t ldtail;
    // ---- split-bf16 ("bf16x3") operands: an activation of C channels is stored as a bf16 PAIR row [hi(C) | lo(C)] (hi = bf16(x),
    // lo = bf16(x - hi): 16 significant bits) and enters the contraction as the 3C channels [hi | lo | hi] against weights packed as
    // [Wh | Wh | Wl] -- x.w ~ hi.Wh + lo.Wh + hi.Wl, fp32 accumulation (the lo.Wl term, 2^-16 of the product, is dropped).
    // conv_wrap / conv_wrap2 (convolutions only; 0 = off): channel offsets >= wrap of the first / second source wrap around to
    // offset - wrap, i.e. the third block reads the hi half again (wrap = 2C of that source; conv_Cin / conv_split count the 3C form).
    int conv_wrap, conv_wrap2;
    // EPI_SPLIT: out (bf16, row stride ldo) receives act(acc + bias) as such a pair: hi at column n, lo at column n + split_lo.
    // N % 2 == 0, ldo % 4 == 0, split_lo % 4 == 0; persistent kernel only.
    int split_lo;
    // ---- LayerNorm folded into the GEMMs around it (bf16 ViT path, r5: the 79 stand-alone LayerNorm passes of EVA-ViT-g are gone).
    // y = LN(x) W^T + b = rstd_m (x_m . W'_n - mean_m cs_n) + c_n   with W' = W diag(gamma), cs_n = sum_k W'_nk, c_n = sum_k beta_k W_nk + b_n.
    // PRODUCER (EPI_RESID_F32: the GEMM that writes the residual stream x): ln_xb (bf16 [M, ldxb]) also receives bf16(x) -- the next GEMM's
    // operand -- and ln_part [M][ceil(N / 64)][2] the (sum, sum of squares) of every 64-column block of the row (launch_ln_fold_stats turns
    // them into (mean, rstd) per row, in block order).
    void* ln_xb;
    int64_t ldxb;
    float* ln_part;
    // CONSUMER (EPI_STORE / EPI_GELU, bias == NULL, N % 4 == 0): out = act(rstd_m (acc - mean_m cs_n) + c_n); ln_stats [M][2] = (mean, rstd).
    const float* ln_stats;
    const float* ln_cs;
    const float* ln_c;
    // ---- VTGB_F16C8 operands (pair_h8.h; launch_conv_h8 only): the activation row of a C-channel source is [xh fp16 x C | 8 correction bytes per 4
    // channels] = 2 C 16-bit units (conv_Cin / conv_split / lda count those), K = taps x conv_Cin runs per source `h8_run` fp16 k-tiles then
    // `h8_run` fp8 k-tiles (h8_run = taps x C / 64).  h8_scale: DEVICE pointer to the E8M0 byte (as an int) of 2^-11 / sw, sw = the layer's
    // power-of-two weight scale (ops.py h8_pack).  h8_out_bf16 (EPI_SPLIT): the output pair is written as a bf16 pair [hi | lo] instead.
    // split_f16c8 (the bf16 kernels' EPI_SPLIT, gemm_pp.hip): a bf16x3 convolution writes its output as an f16c8 pair (its consumer is an h8 launch).
    int h8_run;
    const int* h8_scale;
    int h8_out_bf16;
    int split_f16c8;
};
int launch_conv_h8(const GemmDesc& d, hipStream_t s);   // gemm_h8.hip
int launch_ln_fold_stats(const float* part, int nblk, int D, float eps, float* stats, int64_t M, hipStream_t s);
int launch_ln_fold_prepare(const float* x, int64_t ldx, int D, void* xb, float* part, int64_t M, hipStream_t s);   // bf16(x) -> xb [M, D], block moments -> part
#define VTGB_EPI_GRU 4
#define VTGB_EPI_SPLIT 5
// bf16x3 SepConvGRU epilogues (raft_x3.hip; persistent kernel only):
// X3ZR (N = 256: z | r): v = acc + resid[m][n] (fp32 start map: the loop-invariant `inp` third + bias, ldr); columns [0, 128): out[m][n] (fp32, ldo) =
//   sigmoid(v); columns [128, 256): out2[m][n - 128] (bf16 pair, ldo2, lo at + split_lo) = sigmoid(v) * h, h = aux[m][n - 128] + aux[m][n - 128 + split_lo]
//   (bf16 pair, ldaux)
// X3Q (N = 128): q = tanh(acc + resid[m][n]); z = aux[m][n] (fp32, ldaux); h = out[m][n] + out[m][n + split_lo] (bf16 pair, ldo);
//   out <- (1 - z) h + z q as a pair, in place
#define VTGB_EPI_X3ZR 6
#define VTGB_EPI_X3Q 7
int launch_gemm(const GemmDesc& d, hipStream_t s);
int launch_conv_gemm(const GemmDesc& d, hipStream_t s);   // large kernel forced: implicit conv / activations / GRU

struct AttnDesc {
    int dtype, batch, heads, head_dim, s_q, s_kv;
    const void *q, *k, *v;
    int64_t q_tok, kv_tok, q_batch, kv_batch;  // element strides
    const float* key_mask;                     // additive fp32 [batch, s_kv] or null
    const float* rope_q;
    const float* rope_k;
    float scale;
    void* out;
    int64_t o_tok, o_batch;
    int causal = 0;                            // query q sees keys <= q + (s_kv - s_q)
};
int launch_attention(const AttnDesc& d, hipStream_t s);

bool stem7x7_supported(int H, int W);
// stats_part: scratch for the per-run partial moments (stats_part_floats(n_img)); the moments land in `stats` [n_img, 64, 2] in a fixed order
int launch_stem7x7(const float* img, const void* w, const float* bias, float* out_f32, float* stats, float* stats_part, void* out_bf16, int n_img, int H, int W,
                   int relu, hipStream_t s);
size_t conv64_stats_part_floats(int n_img);
// InstanceNorm moments from per-tile / per-part partial sums, added in a fixed order (raft_enc.hip)
int launch_stats_finish_tiles(const float* part, float* stats, int n_img, int HW, int N, int64_t M, hipStream_t s);
int launch_stats_finish_parts(const float* part, float* stats, int n_img, int parts, int C, hipStream_t s);
int cu_count();   // compute units of the current device (gemm_pp.hip)

// conv64.hip: 3x3 / stride 1 / 64 -> 64 channels over NHWC bf16 with the input rows resident in LDS (RAFT encoders, layer1)
bool conv3x3_c64_supported(int H, int W);
int launch_conv3x3_c64(const void* in, const void* w, const float* bias, float* out_f32, float* stats, float* stats_part, void* out_bf16, const void* resid,
                       int n_img, int H, int W, int relu, int post_relu, hipStream_t s);

struct LnDesc {
    int dtype, M, D;
    float eps;
    const float* x;
    int64_t ldx;
    RowMap x_map;
    const float* gamma;
    const float* beta;
    float* out_f32;  // nullable
    void* out_act;   // nullable, `dtype`
    int64_t ldo;
    RowMap o_map;
};
int launch_layernorm(const LnDesc& d, hipStream_t s);

// elementwise helpers (elementwise.hip)
int launch_im2col(int dtype, const float* pix, void* out, int n_img, int ch, int image, int patch, int kpad, hipStream_t s);
int launch_vit_cls_rows(const float* cls, const float* pos, float* x, int n_frames, int tokens, int hidden, hipStream_t s);
int launch_cast_act(int dtype, const float* src, void* dst, int64_t n, hipStream_t s);
int launch_mask_to_additive(const int64_t* mask, float* out, int64_t n, float neg, hipStream_t s);
int launch_fill_f32(float* dst, float v, int64_t n, hipStream_t s);
int launch_qformer_embed(const float* query, const int64_t* ids, const float* wemb, const float* pemb, float* x,
                         int n_frames, int n_query, int n_text, int hidden, hipStream_t s);
int launch_tgb_text_embed(const int64_t* ids, const float* wemb, const float* temb, float* x, int64_t rows, int hidden,
                          hipStream_t s);
int launch_flow_reduce(const float* of, const float* fcw, float* red, int n_img, int image, int patch, hipStream_t s);
int launch_flow_assemble(const float* conv, const float* proj_b, const float* fcw, const float* fcb, const float* bos,
                         const float* eos, const float* pos, const int64_t* of_mask, float* x, int B, int L,
                         int hidden, int n_patches, hipStream_t s);
int launch_mean_pool_uniform(const float* q, float* out, int n_clips, int width, int64_t row_elems, hipStream_t s);
int launch_pack_bf16(const float* src, void* dst, int64_t rows, int64_t cols, int64_t cols_pad, hipStream_t s);
int launch_mrc_head(const float* x, const float* w, const float* b, float* logits, int B, int L, int hidden, hipStream_t s);

// RAFT's correlation pyramid as the lookup kernels take it (raft.hip, raft_x3.hip)
struct CorrPyr { const void* lvl[4]; int h[4], w[4]; };

// launch timing (forward.hip): bracket a launch with events when profiling is enabled
struct ProfScope {
    int slot;
    hipStream_t s;
    ProfScope(int kind, double flops, hipStream_t stream, double executed_flops = -1.0);
    ~ProfScope();
};
