// forward.hip -- host side of libvtgb.so: the C-ABI entry points of include/vtgb.h and the
// launch sequences of the ViT, Q-Former, pooling/projection and TGB stages.  No allocation, no
// synchronisation: every buffer is carved from the caller's workspace, every launch goes to
// the caller's stream (so a whole stage can be captured into a hipGraph).
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "common.h"

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void vtgb_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* vtgb_last_error(void) { return g_err; }
extern "C" int vtgb_version(void) { return VTGB_VERSION; }

extern "C" int vtgb_pack_bf16(const float* src, void* dst, int64_t rows, int64_t cols, int64_t cols_pad, vtgb_stream_t stream) {
    VTGB_REQUIRE(src && dst && rows > 0 && cols > 0 && cols_pad >= cols, VTGB_EINVAL, "pack_bf16: bad argument");
    return launch_pack_bf16(src, dst, rows, cols, cols_pad, stream);
}

// ---------------------------------------------------------------------------------------
// launch timing
// ---------------------------------------------------------------------------------------
#include <vector>
namespace {
struct ProfRec { hipEvent_t a, b; int kind; double flops, executed; };
struct Prof {
    bool on = false;
    std::vector<ProfRec> recs;   // pool: events are created once and reused after reset
    size_t used = 0;
} g_prof;
}  // namespace
ProfScope::ProfScope(int kind, double flops, hipStream_t stream, double executed_flops) : slot(-1), s(stream) {
    if (!g_prof.on) return;
    if (g_prof.used == g_prof.recs.size()) {
        ProfRec r;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        g_prof.recs.push_back(r);
    }
    slot = (int)g_prof.used++;
    g_prof.recs[slot].kind = kind;
    g_prof.recs[slot].flops = flops;
    g_prof.recs[slot].executed = executed_flops < 0 ? flops : executed_flops;
    (void)hipEventRecord(g_prof.recs[slot].a, s);
}
ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof.recs[slot].b, s);
}
extern "C" void vtgb_prof_enable(int on) { g_prof.on = on != 0; }
extern "C" void vtgb_prof_reset(void) { g_prof.used = 0; }
extern "C" int vtgb_prof_summary(int kind, int64_t* launches, double* ms, double* flops) {
    int64_t n = 0;
    double t = 0, f = 0;
    for (size_t i = 0; i < g_prof.used; i++) {
        const ProfRec& r = g_prof.recs[i];
        if (r.kind != kind) continue;
        VTGB_HIP(hipEventSynchronize(r.b));
        float e = 0.f;
        VTGB_HIP(hipEventElapsedTime(&e, r.a, r.b));
        n++; t += e; f += r.flops;
    }
    if (launches) *launches = n;
    if (ms) *ms = t;
    if (flops) *flops = f;
    return VTGB_OK;
}

extern "C" int vtgb_prof_executed_flops(int kind, double* flops) {
    double f = 0;
    for (size_t i = 0; i < g_prof.used; i++)
        if (g_prof.recs[i].kind == kind) f += g_prof.recs[i].executed;
    if (flops) *flops = f;
    return VTGB_OK;
}

static inline const void* off(const void* p, int64_t elems, int dtype) { return (const char*)p + elems * (int64_t)dtype_size(dtype); }
static inline void* off(void* p, int64_t elems, int dtype) { return (char*)p + elems * (int64_t)dtype_size(dtype); }

static GemmDesc gemm(int dtype, int M, int N, int K, int epi, const void* A, int64_t lda, const void* W, int64_t ldw,
                     const float* bias, void* out, int64_t ldo) {
    GemmDesc d;
    memset(&d, 0, sizeof(d));
    d.dtype = dtype; d.M = M; d.N = N; d.K = K; d.epi = epi;
    d.A = A; d.lda = lda; d.W = W; d.ldw = ldw; d.bias = bias; d.out = out; d.ldo = ldo;
    return d;
}
static GemmDesc with_resid(GemmDesc d, const float* resid, int64_t ldr, RowMap r_map) {
    d.resid = resid; d.ldr = ldr; d.r_map = r_map;
    return d;
}
static LnDesc ln(int dtype, int M, int D, float eps, const float* x, const float* g, const float* b, float* of32, void* oact) {
    LnDesc d;
    memset(&d, 0, sizeof(d));
    d.dtype = dtype; d.M = M; d.D = D; d.eps = eps; d.x = x; d.ldx = D; d.gamma = g; d.beta = b;
    d.out_f32 = of32; d.out_act = oact; d.ldo = D;
    return d;
}

// =======================================================================================
// ViT
// =======================================================================================
extern "C" int32_t vtgb_vit_patch_kpad(int32_t dtype, int32_t patch) {
    const int k = 3 * patch * patch;
    return dtype == VTGB_BF16 ? (k + 63) / 64 * 64 : k;
}

static int vit_impl(const vtgb_vit_args* a, Workspace& ws, hipStream_t s) {
    VTGB_REQUIRE(a, VTGB_EINVAL, "vit: NULL args");
    VTGB_REQUIRE(a->dtype == VTGB_BF16 || a->dtype == VTGB_F32, VTGB_EINVAL, "vit: bad dtype %d", a->dtype);
    VTGB_REQUIRE(a->n_frames > 0 && a->patch > 0 && a->image % a->patch == 0 && a->heads > 0 && a->hidden % a->heads == 0 &&
                     a->layers >= 0 && a->mlp > 0,
                 VTGB_EINVAL, "vit: bad dims (n_frames=%d image=%d patch=%d hidden=%d heads=%d)", a->n_frames, a->image,
                 a->patch, a->hidden, a->heads);
    const int dt = a->dtype;
    const size_t es = dtype_size(dt);
    const int g = a->image / a->patch, tokens = g * g + 1, D = a->hidden, n = a->n_frames, hd = D / a->heads;
    const int64_t M = (int64_t)n * tokens, Mp = (int64_t)n * g * g;
    const int kpad = vtgb_vit_patch_kpad(dt, a->patch);
    void* col = ws.take(Mp * kpad * es);
    float* x = (float*)ws.take(M * D * 4);
    void* h = ws.take(M * D * es);
    void* qkv = ws.take(M * 3 * D * es);
    void* ctx = ws.take(M * D * es);
    void* mlp = ws.take(M * a->mlp * es);
    const int ln_nblk = (D + 63) / 64;
    float* ln_part = (float*)ws.take(M * ln_nblk * 2 * 4);      // folded LayerNorms: per (row, 64-column block) moments; (mean, rstd) per row
    float* ln_stats = (float*)ws.take(M * 2 * 4);
    if (ws.dry) return VTGB_OK;
    VTGB_REQUIRE(ws.ok(), VTGB_EWORKSPACE, "vit: workspace %zu < %zu bytes", ws.size, ws.used);
    VTGB_REQUIRE(a->pixel_values && a->weights && (a->out_f32 || a->out_act), VTGB_EINVAL, "vit: You have to specify pixel_values");
    const void* const* w = a->weights;
    // bf16: the LayerNorms folded into the qkv / fc1 GEMMs when the table carries the folded weights (+12 .. +17 of every layer)
    // (all six folded slots of every layer or none -- a partially filled block is an error, not a silent fall back; with them present the
    // unfolded qkv / fc1 weights at +2 / +8 are not read and may be NULL: 1.1 GB of HBM at EVA-ViT-g)
    int n_fold = 0;
    for (int l = 0; l < a->layers; l++)
        for (int k = 12; k < VTGB_VIT_NW_LAYER; k++) n_fold += w[VTGB_VIT_NW_GLOBAL + VTGB_VIT_NW_LAYER * l + k] != nullptr;
    VTGB_REQUIRE(n_fold == 0 || n_fold == 6 * a->layers, VTGB_EINVAL, "vit: %d of %d folded-LayerNorm table entries are set (all six per layer, or none)", n_fold, 6 * a->layers);
    const bool fold = dt == VTGB_BF16 && a->layers > 0 && (D % 4) == 0 && n_fold > 0;
    for (int i = 0; i < VTGB_VIT_NW_GLOBAL + VTGB_VIT_NW_LAYER * a->layers; i++) {
        const int li = i < VTGB_VIT_NW_GLOBAL ? -1 : (i - VTGB_VIT_NW_GLOBAL) % VTGB_VIT_NW_LAYER;
        if (li >= 12 || (fold && (li == 2 || li == 8))) continue;
        VTGB_REQUIRE(w[i], VTGB_EINVAL, "vit: weights[%d] is NULL", i);
    }

    // embeddings (xinstructblip.py:113-122): patch GEMM writes rows 1.. with + position_embedding fused
    VTGB_TRY(launch_im2col(dt, a->pixel_values, col, n, 3, a->image, a->patch, kpad, s));
    {
        GemmDesc d = gemm(dt, (int)Mp, D, kpad, VTGB_EPI_RESID_F32, col, kpad, w[0], kpad, (const float*)w[1], x, D);
        d = with_resid(d, (const float*)w[3], D, rowmap(g * g, 0, 1));
        d.o_map = rowmap(g * g, tokens, 1);
        VTGB_TRY(launch_gemm(d, s));
    }
    VTGB_TRY(launch_vit_cls_rows((const float*)w[2], (const float*)w[3], x, n, tokens, D, s));
    const float scale = (float)pow((double)hd, -0.5);   // :140
    // folded LayerNorms (GemmDesc::ln_*): h holds bf16(x), written with the rows' block moments by whoever produced x -- the embedding (one
    // pass below) or the projection / fc2 epilogue -- and the GEMM that follows applies rstd (acc - mean cs) + c in ITS epilogue
    auto ln_consumer = [&](GemmDesc d, const void* wf, const void* cs, const void* c) {
        d.W = wf; d.bias = nullptr; d.ln_stats = ln_stats; d.ln_cs = (const float*)cs; d.ln_c = (const float*)c;
        return d;
    };
    auto ln_producer = [&](GemmDesc d, bool on) {
        if (on) { d.ln_xb = h; d.ldxb = D; d.ln_part = ln_part; }
        return d;
    };
    if (fold) {
        VTGB_TRY(launch_ln_fold_prepare(x, D, D, h, ln_part, M, s));
        VTGB_TRY(launch_ln_fold_stats(ln_part, ln_nblk, D, a->eps, ln_stats, M, s));
    }
    for (int l = 0; l < a->layers; l++) {
        const void* const* lw = w + VTGB_VIT_NW_GLOBAL + VTGB_VIT_NW_LAYER * l;
        if (fold) {
            VTGB_TRY(launch_gemm(ln_consumer(gemm(dt, (int)M, 3 * D, D, VTGB_EPI_STORE, h, D, lw[12], D, nullptr, qkv, 3 * D), lw[12], lw[13], lw[14]), s));
        } else {
        VTGB_TRY(launch_layernorm(ln(dt, (int)M, D, a->eps, x, (const float*)lw[0], (const float*)lw[1], nullptr, h), s));
        VTGB_TRY(launch_gemm(gemm(dt, (int)M, 3 * D, D, VTGB_EPI_STORE, h, D, lw[2], D, (const float*)lw[3], qkv, 3 * D), s));
        }
        AttnDesc at;
        memset(&at, 0, sizeof(at));
        at.dtype = dt; at.batch = n; at.heads = a->heads; at.head_dim = hd; at.s_q = tokens; at.s_kv = tokens;
        at.q = qkv; at.k = off(qkv, D, dt); at.v = off(qkv, 2 * D, dt);
        at.q_tok = at.kv_tok = 3 * D; at.q_batch = at.kv_batch = (int64_t)tokens * 3 * D;
        at.scale = scale; at.out = ctx; at.o_tok = D; at.o_batch = (int64_t)tokens * D;
        VTGB_TRY(launch_attention(at, s));
        VTGB_TRY(launch_gemm(ln_producer(with_resid(gemm(dt, (int)M, D, D, VTGB_EPI_RESID_F32, ctx, D, lw[4], D, (const float*)lw[5], x, D), x, D,
                                                    rowmap_identity()), fold), s));
        if (fold) {
            VTGB_TRY(launch_ln_fold_stats(ln_part, ln_nblk, D, a->eps, ln_stats, M, s));
            VTGB_TRY(launch_gemm(ln_consumer(gemm(dt, (int)M, a->mlp, D, VTGB_EPI_GELU, h, D, lw[15], D, nullptr, mlp, a->mlp), lw[15], lw[16], lw[17]), s));
        } else {
        VTGB_TRY(launch_layernorm(ln(dt, (int)M, D, a->eps, x, (const float*)lw[6], (const float*)lw[7], nullptr, h), s));
        VTGB_TRY(launch_gemm(gemm(dt, (int)M, a->mlp, D, VTGB_EPI_GELU, h, D, lw[8], D, (const float*)lw[9], mlp, a->mlp), s));
        }
        const bool more = fold && l + 1 < a->layers;      // (the last layer's output goes to post_layernorm, a pass of its own: it has two outputs)
        VTGB_TRY(launch_gemm(ln_producer(with_resid(gemm(dt, (int)M, D, a->mlp, VTGB_EPI_RESID_F32, mlp, a->mlp, lw[10], a->mlp, (const float*)lw[11], x, D),
                                                    x, D, rowmap_identity()), more), s));
        if (more) VTGB_TRY(launch_ln_fold_stats(ln_part, ln_nblk, D, a->eps, ln_stats, M, s));
    }
    VTGB_TRY(launch_layernorm(ln(dt, (int)M, D, a->eps, x, (const float*)w[4], (const float*)w[5], a->out_f32, a->out_act), s));
    return VTGB_OK;
}

extern "C" size_t vtgb_vit_workspace_bytes(const vtgb_vit_args* a) {
    Workspace ws(nullptr, 0);
    if (vit_impl(a, ws, nullptr) != VTGB_OK) return 0;
    return align_up(ws.used, 256);
}
extern "C" int vtgb_vit_forward(const vtgb_vit_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->workspace, VTGB_EWORKSPACE, "vit: workspace is NULL");
    Workspace ws(a->workspace, a->workspace_bytes);
    return vit_impl(a, ws, stream);
}

// =======================================================================================
// Q-Former
// =======================================================================================
// additive self mask [n, nq + nt]: 0 for query columns, (1 - m) * -10000 for text (xinstructblip.py:1118-1119)
__global__ void qformer_self_mask_kernel(const int64_t* text_mask, float* out, int n, int nq, int nt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int S = nq + nt;
    if (i >= (int64_t)n * S) return;
    const int f = i / S, r = i % S;
    out[i] = r < nq ? 0.f : (1.0f - (float)text_mask[(int64_t)f * nt + (r - nq)]) * -10000.0f;
}

static int bert_self_attention(int dt, const void* const* lw, int wq, int64_t Ms, int D, int heads, int batch, int S,
                               const void* Hb, void* qkv, void* ctx, const float* key_mask, const float* rope, hipStream_t s) {
    // three projections into one [Ms, 3D] buffer, then attention over it
    for (int i = 0; i < 3; i++)
        VTGB_TRY(launch_gemm(gemm(dt, (int)Ms, D, D, VTGB_EPI_STORE, Hb, D, lw[wq + 2 * i], D, (const float*)lw[wq + 2 * i + 1],
                                  off(qkv, (int64_t)i * D, dt), 3 * D), s));
    AttnDesc at;
    memset(&at, 0, sizeof(at));
    const int hd = D / heads;
    at.dtype = dt; at.batch = batch; at.heads = heads; at.head_dim = hd; at.s_q = S; at.s_kv = S;
    at.q = qkv; at.k = off(qkv, D, dt); at.v = off(qkv, 2 * D, dt);
    at.q_tok = at.kv_tok = 3 * D; at.q_batch = at.kv_batch = (int64_t)S * 3 * D;
    at.key_mask = key_mask; at.rope_q = rope; at.rope_k = rope;
    at.scale = (float)(1.0 / sqrt((double)hd));
    at.out = ctx; at.o_tok = D; at.o_batch = (int64_t)S * D;
    return launch_attention(at, s);
}

static int qformer_impl(const vtgb_qformer_args* a, Workspace& ws, hipStream_t s) {
    VTGB_REQUIRE(a, VTGB_EINVAL, "qformer: NULL args");
    VTGB_REQUIRE(a->dtype == VTGB_BF16 || a->dtype == VTGB_F32, VTGB_EINVAL, "qformer: bad dtype %d", a->dtype);
    VTGB_REQUIRE(a->n_frames > 0 && a->n_query > 0 && a->n_text >= 0 && a->heads > 0 && a->hidden % a->heads == 0 && a->cross_freq > 0 &&
                     a->enc_tokens > 0 && a->enc_hidden > 0,
                 VTGB_EINVAL, "qformer: bad dims");
    const int dt = a->dtype;
    const size_t es = dtype_size(dt);
    const int n = a->n_frames, nq = a->n_query, nt = a->has_text ? a->n_text : 0, S = nq + nt, D = a->hidden, E = a->enc_hidden;
    const int64_t Ms = (int64_t)n * S, Mq = (int64_t)n * nq, Mt = (int64_t)n * nt, Mi = (int64_t)n * a->enc_tokens;
    float* H = (float*)ws.take(Ms * D * 4);
    void* Hb = ws.take(Ms * D * es);
    float* tmp = (float*)ws.take(Ms * D * 4);
    void* qkv = ws.take(Ms * 3 * D * es);
    void* ctx = ws.take(Ms * D * es);
    float* A1 = (float*)ws.take(Ms * D * 4);
    void* A1b = ws.take(Ms * D * es);
    void* qc = ws.take(Mq * D * es);
    void* kvc = ws.take(Mi * 2 * D * es);
    void* ctx2 = ws.take(Mq * D * es);
    void* inter = ws.take((Mq > Mt ? Mq : Mt) * a->ffn * es);
    float* self_mask = (float*)ws.take(Ms * 4);
    float* cross_mask = (float*)ws.take(Mi * 4);
    if (ws.dry) return VTGB_OK;
    VTGB_REQUIRE(ws.ok(), VTGB_EWORKSPACE, "qformer: workspace %zu < %zu bytes", ws.size, ws.used);
    VTGB_REQUIRE(a->image_embeds && a->query_tokens && a->weights && a->out_f32, VTGB_EINVAL,
                 "qformer: You have to specify query_embeds when input_ids is None");
    VTGB_REQUIRE(!(a->has_text && nt > 0) || a->text_ids, VTGB_EINVAL, "qformer: has_text without text_ids");
    const void* const* w = a->weights;
    const bool use_self_mask = nt > 0 && a->text_mask;
    if (use_self_mask)
        hipLaunchKernelGGL(qformer_self_mask_kernel, dim3((unsigned)((Ms + 255) / 256)), dim3(256), 0, s, a->text_mask, self_mask, n, nq, nt);
    if (a->image_mask) VTGB_TRY(launch_mask_to_additive(a->image_mask, cross_mask, Mi, -3.4028234663852886e38f, s));
    // embeddings -> LayerNorm (xinstructblip.py:1018-1046 / xblip2.py:1108)
    VTGB_TRY(launch_qformer_embed(a->query_tokens, a->text_ids, (const float*)w[0], (const float*)w[1], tmp, n, nq, nt, D, s));
    VTGB_TRY(launch_layernorm(ln(dt, (int)Ms, D, a->eps, tmp, (const float*)w[2], (const float*)w[3], H, Hb), s));
    const RowMap qrows = rowmap(nq, S, 0), trows = rowmap(nt > 0 ? nt : 1, S, nq);
    for (int l = 0; l < a->layers; l++) {
        const void* const* lw = w + VTGB_QF_NW_GLOBAL + VTGB_QF_NW_LAYER * l;
        // self attention over queries + text (:827-834)
        VTGB_TRY(bert_self_attention(dt, lw, 0, Ms, D, a->heads, n, S, Hb, qkv, ctx, use_self_mask ? self_mask : nullptr, nullptr, s));
        VTGB_TRY(launch_gemm(with_resid(gemm(dt, (int)Ms, D, D, VTGB_EPI_RESID_F32, ctx, D, lw[6], D, (const float*)lw[7], tmp, D), H, D,
                                        rowmap_identity()), s));
        VTGB_TRY(launch_layernorm(ln(dt, (int)Ms, D, a->eps, tmp, (const float*)lw[8], (const float*)lw[9], A1, A1b), s));
        if (l % a->cross_freq == 0) {   // cross attention of the query rows to the image tokens (:842-855)
            VTGB_REQUIRE(lw[10] && lw[12] && lw[14] && lw[16], VTGB_EINVAL, "qformer: layer %d lacks cross-attention weights", l);
            GemmDesc gq = gemm(dt, (int)Mq, D, D, VTGB_EPI_STORE, A1b, D, lw[10], D, (const float*)lw[11], qc, D);
            gq.a_map = qrows;
            VTGB_TRY(launch_gemm(gq, s));
            VTGB_TRY(launch_gemm(gemm(dt, (int)Mi, D, E, VTGB_EPI_STORE, a->image_embeds, E, lw[12], E, (const float*)lw[13], kvc, 2 * D), s));
            VTGB_TRY(launch_gemm(gemm(dt, (int)Mi, D, E, VTGB_EPI_STORE, a->image_embeds, E, lw[14], E, (const float*)lw[15], off(kvc, D, dt), 2 * D), s));
            AttnDesc at;
            memset(&at, 0, sizeof(at));
            const int hd = D / a->heads;
            at.dtype = dt; at.batch = n; at.heads = a->heads; at.head_dim = hd; at.s_q = nq; at.s_kv = a->enc_tokens;
            at.q = qc; at.q_tok = D; at.q_batch = (int64_t)nq * D;
            at.k = kvc; at.v = off(kvc, D, dt); at.kv_tok = 2 * D; at.kv_batch = (int64_t)a->enc_tokens * 2 * D;
            at.key_mask = a->image_mask ? cross_mask : nullptr;
            at.scale = (float)(1.0 / sqrt((double)hd));
            at.out = ctx2; at.o_tok = D; at.o_batch = (int64_t)nq * D;
            VTGB_TRY(launch_attention(at, s));
            VTGB_TRY(launch_gemm(with_resid(gemm(dt, (int)Mq, D, D, VTGB_EPI_RESID_F32, ctx2, D, lw[16], D, (const float*)lw[17], tmp, D), A1, D, qrows), s));
            LnDesc d = ln(dt, (int)Mq, D, a->eps, tmp, (const float*)lw[18], (const float*)lw[19], A1, A1b);
            d.o_map = qrows;
            VTGB_TRY(launch_layernorm(d, s));
        }
        // query FFN (:857-862) and text FFN (:864-871), each on its own row range
        for (int part = 0; part < 2; part++) {
            const int64_t Mr = part == 0 ? Mq : Mt;
            if (Mr == 0) continue;
            const RowMap rows = part == 0 ? qrows : trows;
            const int wi = part == 0 ? 20 : 26;
            VTGB_REQUIRE(lw[wi] && lw[wi + 2] && lw[wi + 4], VTGB_EINVAL, "qformer: layer %d lacks FFN weights (part %d)", l, part);
            GemmDesc g1 = gemm(dt, (int)Mr, a->ffn, D, VTGB_EPI_GELU, A1b, D, lw[wi], D, (const float*)lw[wi + 1], inter, a->ffn);
            g1.a_map = rows;
            VTGB_TRY(launch_gemm(g1, s));
            VTGB_TRY(launch_gemm(with_resid(gemm(dt, (int)Mr, D, a->ffn, VTGB_EPI_RESID_F32, inter, a->ffn, lw[wi + 2], a->ffn,
                                                 (const float*)lw[wi + 3], tmp, D), A1, D, rows), s));
            LnDesc d = ln(dt, (int)Mr, D, a->eps, tmp, (const float*)lw[wi + 4], (const float*)lw[wi + 5], H, Hb);
            d.o_map = rows;
            VTGB_TRY(launch_layernorm(d, s));
        }
    }
    VTGB_HIP(hipMemcpy2DAsync(a->out_f32, (size_t)nq * D * 4, H, (size_t)S * D * 4, (size_t)nq * D * 4, n, hipMemcpyDeviceToDevice, s));
    return VTGB_OK;
}

extern "C" size_t vtgb_qformer_workspace_bytes(const vtgb_qformer_args* a) {
    Workspace ws(nullptr, 0);
    if (qformer_impl(a, ws, nullptr) != VTGB_OK) return 0;
    return align_up(ws.used, 256);
}
extern "C" int vtgb_qformer_forward(const vtgb_qformer_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->workspace, VTGB_EWORKSPACE, "qformer: workspace is NULL");
    Workspace ws(a->workspace, a->workspace_bytes);
    return qformer_impl(a, ws, stream);
}

// =======================================================================================
// frame pooling + language_projection
// =======================================================================================
static int pool_impl(const vtgb_pool_project_args* a, Workspace& ws, hipStream_t s) {
    VTGB_REQUIRE(a, VTGB_EINVAL, "pool_project: NULL args");
    VTGB_REQUIRE(a->dtype == VTGB_BF16 || a->dtype == VTGB_F32, VTGB_EINVAL, "pool_project: bad dtype %d", a->dtype);
    VTGB_REQUIRE(a->mode == VTGB_POOL_MEAN || a->mode == VTGB_POOL_CONCAT, VTGB_EINVAL, "INVALID POOL MODE: %d", a->mode);
    VTGB_REQUIRE(a->n_clips > 0 && a->n_query > 0 && a->hidden > 0 && a->out_dim > 0 && a->widths, VTGB_EINVAL, "pool_project: bad dims");
    const int dt = a->dtype;
    int64_t total = 0;
    for (int i = 0; i < a->n_clips; i++) {
        VTGB_REQUIRE(a->widths[i] >= 0, VTGB_EINVAL, "pool_project: negative width");
        total += a->widths[i];
    }
    const int64_t rows = a->mode == VTGB_POOL_MEAN ? (int64_t)a->n_clips * a->n_query : total * a->n_query;
    float* pooled = (float*)ws.take((int64_t)a->n_clips * a->n_query * a->hidden * 4);
    void* act = ws.take(rows * a->hidden * dtype_size(dt));
    if (ws.dry) return VTGB_OK;
    VTGB_REQUIRE(ws.ok(), VTGB_EWORKSPACE, "pool_project: workspace %zu < %zu bytes", ws.size, ws.used);
    VTGB_REQUIRE(a->query_out && a->proj_w && a->out, VTGB_EINVAL, "pool_project: NULL operand");
    VTGB_REQUIRE(rows > 0, VTGB_EINVAL, "pool_project: no rows");
    const int64_t re = (int64_t)a->n_query * a->hidden;
    const float* src = a->query_out;
    if (a->mode == VTGB_POOL_MEAN) {
        bool uniform = true;
        for (int i = 1; i < a->n_clips; i++) uniform &= a->widths[i] == a->widths[0];
        if (uniform) {
            VTGB_TRY(launch_mean_pool_uniform(a->query_out, pooled, a->n_clips, a->widths[0], re, s));
        } else {
            int64_t o = 0;
            for (int i = 0; i < a->n_clips; i++) {
                VTGB_TRY(launch_mean_pool_uniform(a->query_out + o * re, pooled + (int64_t)i * re, 1, a->widths[i], re, s));
                o += a->widths[i];
            }
        }
        src = pooled;
    }
    const void* A = src;
    if (dt == VTGB_BF16) {
        VTGB_TRY(launch_cast_act(dt, src, act, rows * a->hidden, s));
        A = act;
    }
    return launch_gemm(gemm(dt, (int)rows, a->out_dim, a->hidden, VTGB_EPI_STORE_F32, A, a->hidden, a->proj_w, a->hidden, a->proj_b, a->out,
                            a->out_dim), s);
}
extern "C" size_t vtgb_pool_project_workspace_bytes(const vtgb_pool_project_args* a) {
    Workspace ws(nullptr, 0);
    if (pool_impl(a, ws, nullptr) != VTGB_OK) return 0;
    return align_up(ws.used, 256);
}
extern "C" int vtgb_pool_project(const vtgb_pool_project_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->workspace, VTGB_EWORKSPACE, "pool_project: workspace is NULL");
    Workspace ws(a->workspace, a->workspace_bytes);
    return pool_impl(a, ws, stream);
}

// =======================================================================================
// Temporal Grounding Bridge
// =======================================================================================
static int tgb_impl(const vtgb_tgb_args* a, Workspace& ws, hipStream_t s) {
    VTGB_REQUIRE(a, VTGB_EINVAL, "tgb: NULL args");
    VTGB_REQUIRE(a->dtype == VTGB_BF16 || a->dtype == VTGB_F32, VTGB_EINVAL, "tgb: bad dtype %d", a->dtype);
    VTGB_REQUIRE(a->mode == VTGB_TGB_MODE_TEXT || a->mode == VTGB_TGB_MODE_FUSION || a->mode == VTGB_TGB_MODE_MULTIMODAL, VTGB_EINVAL,
                 "INVALID MODE: %d", a->mode);
    VTGB_REQUIRE(a->B > 0 && a->L > 0 && a->n_text > 0 && a->heads > 0 && a->hidden % a->heads == 0 && a->fusion_layer >= 0 &&
                     a->fusion_layer <= a->layers && a->patch > 0 && a->image % a->patch == 0,
                 VTGB_EINVAL, "tgb: bad dims");
    const int dt = a->dtype;
    const size_t es = dtype_size(dt);
    const int B = a->B, L = a->L, S = L + 2, nt = a->n_text, D = a->hidden, P = a->patch, g = a->image / P;
    const int64_t Ms = (int64_t)B * S, Mt = (int64_t)B * nt, Mf = (int64_t)B * L;
    const int kf = 2 * P * P;
    float* red = (float*)ws.take(Mf * kf * 4);
    float* conv = (float*)ws.take(Mf * D * 4);
    float* H = (float*)ws.take(Ms * D * 4);
    void* Hb = ws.take(Ms * D * es);
    float* tmp = (float*)ws.take((Ms > Mt ? Ms : Mt) * D * 4);
    void* Tb = ws.take(Mt * D * es);
    void* qkv = ws.take(Ms * 3 * D * es);
    void* ctx = ws.take(Ms * D * es);
    float* A1 = (float*)ws.take(Ms * D * 4);
    void* A1b = ws.take(Ms * D * es);
    void* kvc = ws.take(Mt * 2 * D * es);
    void* inter = ws.take(Ms * a->ffn * es);
    float* self_mask = (float*)ws.take(Ms * 4);
    float* cross_mask = (float*)ws.take(Mt * 4);
    if (ws.dry) return VTGB_OK;
    VTGB_REQUIRE(ws.ok(), VTGB_EWORKSPACE, "tgb: workspace %zu < %zu bytes", ws.size, ws.used);
    VTGB_REQUIRE(a->of && a->of_mask && a->text_ids && a->text_mask && a->weights && a->logits, VTGB_EINVAL,
                 "tgb: You have to specify either input_ids or inputs_embeds or encoder_embeds");
    const void* const* w = a->weights;
    for (int i = 0; i < VTGB_TGB_NW_GLOBAL; i++) VTGB_REQUIRE(w[i], VTGB_EINVAL, "tgb: weights[%d] is NULL", i);
    // ---- TemporalOFEmbedding (xropebert.py:103-129), patch axis reduced first (elementwise.hip)
    VTGB_TRY(launch_flow_reduce(a->of, (const float*)w[8], red, (int)Mf, a->image, P, s));
    VTGB_TRY(launch_gemm(gemm(VTGB_F32, (int)Mf, D, kf, VTGB_EPI_STORE_F32, red, kf, w[6], kf, nullptr, conv, D), s));
    VTGB_TRY(launch_flow_assemble(conv, (const float*)w[7], (const float*)w[8], (const float*)w[9], (const float*)w[4], (const float*)w[5],
                                  (const float*)w[10], a->of_mask, tmp, B, L, D, g * g, s));
    VTGB_TRY(launch_layernorm(ln(dt, (int)Ms, D, 1e-5f, tmp, (const float*)w[11], (const float*)w[12], H, Hb), s));
    // ---- RopeBertEmbeddings on the question (:190-208)
    VTGB_TRY(launch_tgb_text_embed(a->text_ids, (const float*)w[0], (const float*)w[1], tmp, Mt, D, s));
    VTGB_TRY(launch_layernorm(ln(dt, (int)Mt, D, a->eps, tmp, (const float*)w[2], (const float*)w[3], nullptr, Tb), s));
    VTGB_TRY(launch_mask_to_additive(a->of_mask, self_mask, Ms, -10000.0f, s));                 // :1044-1045
    VTGB_TRY(launch_mask_to_additive(a->text_mask, cross_mask, Mt, -3.4028234663852886e38f, s)); // :1127 invert_attention_mask
    const float* rope = (const float*)w[13];
    const float* c_rope = (const float*)w[14];
    int lo = 0, hi = a->layers;                                                                  // :621-634
    if (a->mode == VTGB_TGB_MODE_TEXT) hi = a->fusion_layer;
    if (a->mode == VTGB_TGB_MODE_FUSION) lo = a->fusion_layer;
    const int hd = D / a->heads;
    for (int l = lo; l < hi; l++) {
        const void* const* lw = w + VTGB_TGB_NW_GLOBAL + VTGB_TGB_NW_LAYER * l;
        VTGB_TRY(bert_self_attention(dt, lw, 0, Ms, D, a->heads, B, S, Hb, qkv, ctx, self_mask, rope, s));
        VTGB_TRY(launch_gemm(with_resid(gemm(dt, (int)Ms, D, D, VTGB_EPI_RESID_F32, ctx, D, lw[6], D, (const float*)lw[7], tmp, D), H, D,
                                        rowmap_identity()), s));
        VTGB_TRY(launch_layernorm(ln(dt, (int)Ms, D, a->eps, tmp, (const float*)lw[8], (const float*)lw[9], A1, A1b), s));
        if (l >= a->fusion_layer) {   // cross attention to the question (:466-510)
            VTGB_REQUIRE(lw[10] && lw[12] && lw[14] && lw[16], VTGB_EINVAL, "tgb: layer %d lacks cross-attention weights", l);
            VTGB_TRY(launch_gemm(gemm(dt, (int)Ms, D, D, VTGB_EPI_STORE, A1b, D, lw[10], D, (const float*)lw[11], qkv, D), s));
            VTGB_TRY(launch_gemm(gemm(dt, (int)Mt, D, D, VTGB_EPI_STORE, Tb, D, lw[12], D, (const float*)lw[13], kvc, 2 * D), s));
            VTGB_TRY(launch_gemm(gemm(dt, (int)Mt, D, D, VTGB_EPI_STORE, Tb, D, lw[14], D, (const float*)lw[15], off(kvc, D, dt), 2 * D), s));
            AttnDesc at;
            memset(&at, 0, sizeof(at));
            at.dtype = dt; at.batch = B; at.heads = a->heads; at.head_dim = hd; at.s_q = S; at.s_kv = nt;
            at.q = qkv; at.q_tok = D; at.q_batch = (int64_t)S * D;
            at.k = kvc; at.v = off(kvc, D, dt); at.kv_tok = 2 * D; at.kv_batch = (int64_t)nt * 2 * D;
            at.key_mask = cross_mask; at.rope_q = rope; at.rope_k = c_rope;
            at.scale = (float)(1.0 / sqrt((double)hd));
            at.out = ctx; at.o_tok = D; at.o_batch = (int64_t)S * D;
            VTGB_TRY(launch_attention(at, s));
            VTGB_TRY(launch_gemm(with_resid(gemm(dt, (int)Ms, D, D, VTGB_EPI_RESID_F32, ctx, D, lw[16], D, (const float*)lw[17], tmp, D), A1, D,
                                            rowmap_identity()), s));
            VTGB_TRY(launch_layernorm(ln(dt, (int)Ms, D, a->eps, tmp, (const float*)lw[18], (const float*)lw[19], A1, A1b), s));
        }
        VTGB_TRY(launch_gemm(gemm(dt, (int)Ms, a->ffn, D, VTGB_EPI_GELU, A1b, D, lw[20], D, (const float*)lw[21], inter, a->ffn), s));
        VTGB_TRY(launch_gemm(with_resid(gemm(dt, (int)Ms, D, a->ffn, VTGB_EPI_RESID_F32, inter, a->ffn, lw[22], a->ffn, (const float*)lw[23], tmp, D),
                                        A1, D, rowmap_identity()), s));
        VTGB_TRY(launch_layernorm(ln(dt, (int)Ms, D, a->eps, tmp, (const float*)lw[24], (const float*)lw[25], H, Hb), s));
    }
    if (a->seq_out) VTGB_HIP(hipMemcpyAsync(a->seq_out, H, Ms * D * 4, hipMemcpyDeviceToDevice, s));
    return launch_mrc_head(H, (const float*)w[15], (const float*)w[16], a->logits, B, L, D, s);   // :1164
}
extern "C" size_t vtgb_tgb_workspace_bytes(const vtgb_tgb_args* a) {
    Workspace ws(nullptr, 0);
    if (tgb_impl(a, ws, nullptr) != VTGB_OK) return 0;
    return align_up(ws.used, 256);
}
extern "C" int vtgb_tgb_forward(const vtgb_tgb_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->workspace, VTGB_EWORKSPACE, "tgb: workspace is NULL");
    Workspace ws(a->workspace, a->workspace_bytes);
    return tgb_impl(a, ws, stream);
}

// =======================================================================================
// building blocks exported for per-kernel tests / roofline bench
// =======================================================================================
extern "C" int vtgb_gemm(const vtgb_gemm_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a, VTGB_EINVAL, "gemm: NULL args");
    GemmDesc d = gemm(a->dtype, a->M, a->N, a->K, a->epilogue, a->A, a->lda, a->W, a->ldw, a->bias, a->out, a->ldo);
    if (a->resid) d = with_resid(d, a->resid, a->ldo, rowmap_identity());
    return launch_gemm(d, stream);
}
extern "C" int vtgb_attention(const vtgb_attention_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a, VTGB_EINVAL, "attention: NULL args");
    AttnDesc d;
    memset(&d, 0, sizeof(d));
    d.dtype = a->dtype; d.batch = a->batch; d.heads = a->heads; d.head_dim = a->head_dim; d.s_q = a->s_q; d.s_kv = a->s_kv;
    d.q = a->q; d.k = a->k; d.v = a->v; d.q_tok = a->q_tok_stride; d.kv_tok = a->kv_tok_stride; d.q_batch = a->q_batch_stride;
    d.kv_batch = a->kv_batch_stride; d.key_mask = a->key_mask; d.rope_q = a->rope_q; d.rope_k = a->rope_k; d.scale = a->scale;
    d.out = a->out; d.o_tok = a->out_tok_stride; d.o_batch = a->out_batch_stride; d.causal = a->causal;
    return launch_attention(d, stream);
}
extern "C" int vtgb_layernorm(const vtgb_layernorm_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a, VTGB_EINVAL, "layernorm: NULL args");
    return launch_layernorm(ln(a->dtype, a->M, a->D, a->eps, a->x, a->gamma, a->beta, a->out_f32, a->out_act), stream);
}
