/* vtgb.h -- C ABI of libvtgb.so: the MI355X (gfx950) kernels behind the VideoTGB
 * video -> LLM-prefix hot path.
 *
 * The reference (bigai-nlco/VideoTGB) is pure Python with no FFI of its own; each entry
 * point below replaces the torch op sequence of one reference function (cited as
 * file:line under /root/reference) and is what a ctypes stub on the reference side binds
 * (INTEGRATION.md).  Conventions, common to every call:
 *   - int return: 0 (VTGB_OK) or a negative VTGB_E* code; vtgb_last_error() returns a
 *     thread-local message for the last failing call on the calling thread;
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless the field
 *     comment says "host";  the library never allocates, frees or synchronises: the
 *     caller passes a workspace whose size comes from vtgb_<op>_workspace_bytes();
 *   - asynchronous on the given hipStream_t, re-entrant across streams, capturable
 *     into a hipGraph;
 *   - `dtype` selects the arithmetic of the GEMM/attention operands: VTGB_BF16 (bf16
 *     MFMA, fp32 accumulate, fp32 residual stream / LayerNorm / softmax) or VTGB_F32
 *     (fp32 everywhere: the exactness mode).  GEMM weights are passed in that dtype in
 *     the reference's nn.Linear layout [out_features, in_features] row-major; for
 *     VTGB_F32 these are the state_dict tensors themselves, for VTGB_BF16 a one-time
 *     vtgb_pack_bf16() of them.  Biases, LayerNorm parameters, embeddings: always fp32.
 */
#ifndef VTGB_H
#define VTGB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* vtgb_stream_t; /* == hipStream_t */

#define VTGB_VERSION 601 /* 601: vtgb_pair_conv_ex; vtgb_raft_encoder at VTGB_F16C8: every stride-1 3x3 of the residual blocks on f16c8 operands (weights[40] = int32 [12]); vtgb_raft_update at VTGB_F16C8: convf2 on f16c8 operands (weights[30] = int32 [10]), [20] = FlowHead.conv2 as the fp32 tail of conv1.  600: VTGB_F16C8 (vtgb_raft_update: the update block over fp16 + fp8-correction operands, weights[30] = scale bytes).  500: VTGB_BF16X3 (RAFT at fp32 accuracy on the bf16 MFMA), deterministic InstanceNorm moments, vtgb_gemm_skinny defer_reduce +
                            vtgb_llm_rmsnorm_parts / vtgb_llm_rope_cache_parts.  401: vtgb_gemm_train / vtgb_col_sum_f32 / vtgb_layernorm_train_* / vtgb_gelu_* (the training graph without operand copies or torch
                            elementwise passes); 400 = round 4: vtgb_llm_attention_rows / vtgb_llm_gated_act (the T5 language model of the BLIP-2 flavours on own kernels:
                            eval/utils/model.py:427-437), vtgb_llm_rope_cache with NULL tables = plain cache append; 300 = round 3: stem weight hi|lo layout; vtgb_attention_args.causal; vtgb_gemm_skinny without workspace when unsplit;
                            vtgb_llm_rope_cache_prefill; vtgb_attn_train_forward / backward; vtgb_comm_* / vtgb_allreduce_f32 */

#define VTGB_OK 0
#define VTGB_EINVAL (-1)       /* bad argument (NULL pointer, unsupported size, bad mode) */
#define VTGB_EWORKSPACE (-2)   /* workspace missing or too small                         */
#define VTGB_EHIP (-3)         /* a HIP launch failed                                    */
#define VTGB_EUNSUPPORTED (-4) /* shape outside what the kernels are built for           */

#define VTGB_F32 0
#define VTGB_BF16 1
#define VTGB_BF16X3 2 /* RAFT entry points only: split-bf16 operands (hi | lo pairs, three bf16 MFMA products per fp32 product) */
#define VTGB_F16C8 3  /* vtgb_raft_update / vtgb_raft_encoder only: fp16 main product + two OCP-fp8 correction products (fp32 accuracy class at 2/3 of VTGB_BF16X3's matrix work) */

int vtgb_version(void);
const char* vtgb_last_error(void);

/* One-time weight packing: dst[rows, cols_pad] bf16 <- src[rows, cols] fp32 (zero pad). */
int vtgb_pack_bf16(const float* src, void* dst, int64_t rows, int64_t cols, int64_t cols_pad,
                   vtgb_stream_t stream);

/* ---- K16: Gumbel top-k span selection -----------------------------------------------
 * Replaces eval/utils/model.py:101-113 (LSTP.generate), :315-327 (LSTP_blip2.generate),
 * src/models/LSTP_module.py:403-418.  idx[d][r] = first argmax over L of
 * (logit[r] + noise[d][r]) / tau, rows r < B from logits[..., 0] (start), rows >= B from
 * logits[..., 1] (end).  The caller supplies the Gumbel noise. */
typedef struct {
    const float* logits; /* [B, L, 2]      */
    const float* noise;  /* [draws, 2B, L] */
    int64_t* idx;        /* [draws, 2B]    */
    int32_t B, L, draws;
    float tau;
} vtgb_span_select_args;
int vtgb_span_select(const vtgb_span_select_args* a, vtgb_stream_t stream);

/* ---- K17: span -> candidate-frame index map + subsample -----------------------------
 * Replaces eval/utils/model.py:124-150 (variant A, :135) and :337-366 (variant B, :350);
 * src/models/LSTP_module.py:423-447, src/models/LSTP_SF_module.py:508-533.
 * Rounding rules: SURVEY.md Appendix B.  On an empty union -> range(N); duplicate-double
 * while shorter than nframe; float64 linspace midpoint subsample. */
#define VTGB_MAP_A 0 /* int(s / V * N)             (float32 divide, float32 multiply) */
#define VTGB_MAP_B 1 /* int(s * (N-1) / (V-1))     (int64 multiply, float32 divide)   */
typedef struct {
    const int64_t* sel;   /* [draws, 2B] from vtgb_span_select                        */
    const int32_t* V;     /* [B] per-clip video length, or NULL -> V_all              */
    int64_t* frame_idx;   /* [B, nframe]                                              */
    int32_t B, draws, V_all, N, nframe, variant;
} vtgb_span_to_frames_args;
int vtgb_span_to_frames(const vtgb_span_to_frames_args* a, vtgb_stream_t stream);

/* ---- K18: frame gather ---------------------------------------------------------------
 * Replaces eval/utils/model.py:122,151: out[b, i] = pixel_values[b, frame_idx[b, i]],
 * fp32, every output element written (the reference's zero-init buffer is fully
 * overwritten). frame_elems = C*H*W, must be a multiple of 4. */
typedef struct {
    const float* pixel_values; /* [B, N, frame_elems]      */
    const int64_t* frame_idx;  /* [B, nframe]              */
    float* out;                /* [B, nframe, frame_elems] */
    int32_t B, N, nframe;
    int64_t frame_elems;
} vtgb_gather_frames_args;
int vtgb_gather_frames(const vtgb_gather_frames_args* a, vtgb_stream_t stream);

/* ---- f3: frame preprocessing on the device ----------------------------------------------
 * Replaces the transform chain of get_frames (eval/utils/builder_utils.py:117-128: ResizeVideo ->
 * ToUint8 -> ToTHWC -> ToTensorVideo -> NormalizeVideo, i.e. src/gadgets/functional_video.py:33-41
 * resize (F.interpolate bilinear, align_corners=False, no antialias), truncation to uint8, /255
 * (:76-90) and (x - mean) / std (:93-110)) for decoded frames that are already in HBM.
 * frame_idx (optional): the 32-frame pick of builder_utils.py:133-141 applied on the fly -- output frame i is
 * source frame frame_idx[i]; NULL = identity (the flow_frames tensor). */
typedef struct {
    const uint8_t* raw;        /* [T, H0, W0, 3] decoded RGB frames                          */
    const int64_t* frame_idx;  /* [n_out] or NULL                                            */
    float* out;                /* [n_out, 3, size, size] fp32                                */
    int32_t T, H0, W0, n_out, size;
    float mean[3], std[3];
} vtgb_preprocess_args;
int vtgb_preprocess_frames(const vtgb_preprocess_args* a, vtgb_stream_t stream);

/* ---- a14: loss side of the LoRA training step (config C5) --------------------------------
 * vtgb_concat_text_io replaces LSTPModule.concat_text_input_output and the label construction around it
 * (src/models/LSTP_Vicuna_IVT_module.py:692-718 and :284-291): per row n = sum(input_atts),
 * llm = [input[:n] | output[1:] | input[n:]] (ids and attention mask), labels = -100 for the prefix_len visual
 * positions, for the first n text positions and for pad tokens, else the token id. */
typedef struct {
    const int64_t* input_ids;    /* [B, Li]  question tokens (right-padded)                  */
    const int64_t* input_atts;   /* [B, Li]                                                  */
    const int64_t* output_ids;   /* [B, Lo]  answer tokens (first one, BOS, is dropped)      */
    const int64_t* output_atts;  /* [B, Lo]                                                  */
    int64_t* llm_ids;            /* [B, Li + Lo - 1]                                         */
    int64_t* llm_atts;           /* [B, Li + Lo - 1]                                         */
    int64_t* input_len;          /* [B] or NULL: the n of each row (input_part_targets_len)  */
    int64_t* labels;             /* [B, prefix_len + Li + Lo - 1] or NULL                    */
    int64_t pad_id;
    int32_t B, Li, Lo, prefix_len;
} vtgb_concat_text_io_args;
int vtgb_concat_text_io(const vtgb_concat_text_io_args* a, vtgb_stream_t stream);

/* Shifted cross-entropy (LSTP_Vicuna_IVT_module.py:297-299, :325-326: logits[..., :-1, :] against labels[..., 1:],
 * CrossEntropyLoss(reduction="mean"), ignore_index -100) without the shifted copies: forward reads the logits once
 * (rows whose target is ignored are skipped), backward writes d loss / d logits for all S positions. */
typedef struct {
    int32_t dtype, B, S, V;      /* logits dtype VTGB_F32 / VTGB_BF16                        */
    const void* logits;          /* [B, S, V]                                                */
    const int64_t* labels;       /* [B, S]                                                   */
    float* lse;                  /* [B, S-1] log-sum-exp per scored row (kept for backward)  */
    float* row_loss;             /* [B, S-1] scratch                                         */
    float* loss;                 /* [2]: mean loss, number of scored rows                    */
    const float* grad_out;       /* [1] upstream gradient (backward)                         */
    void* dlogits;               /* [B, S, V] same dtype as logits (backward)                */
} vtgb_shifted_ce_args;
int vtgb_shifted_ce_forward(const vtgb_shifted_ce_args* a, vtgb_stream_t stream);
int vtgb_shifted_ce_backward(const vtgb_shifted_ce_args* a, vtgb_stream_t stream);

/* ---- K1-K6: EVA-ViT-g vision tower ----------------------------------------------------
 * Replaces InstructBlipVisionModel.forward, src/models/components/xinstructblip.py:515-558
 * (embeddings :113-122, 39 x encoder layer :233-269 with attention :162-204 and MLP
 * :216-220, post_layernorm :545) and the identical Blip2VisionModel (xblip2.py:500).
 * weights (host array of device pointers), in this order:
 *   [0] patch_embedding.weight  `dtype` [hidden, kpad]   (kpad: see vtgb_vit_patch_kpad)
 *   [1] patch_embedding.bias  [2] class_embedding  [3] position_embedding [tokens, hidden]
 *   [4] post_layernorm.weight [5] post_layernorm.bias
 *   then per layer l, at 6 + 18*l:
 *   +0 layer_norm1.weight +1 layer_norm1.bias +2 self_attn.qkv.weight `dtype` [3h, h]
 *   +3 self_attn.qkv.bias +4 self_attn.projection.weight `dtype` +5 .bias
 *   +6 layer_norm2.weight +7 layer_norm2.bias +8 mlp.fc1.weight `dtype` +9 .bias
 *   +10 mlp.fc2.weight `dtype` +11 .bias
 *   +12 .. +17 optional, VTGB_BF16 only (all six of EVERY layer or none -- a partially filled block is VTGB_EINVAL since version 600; none = the
 *   LayerNorms run as their own passes; with them present +2 and +8 are not read and may be NULL): the two LayerNorms FOLDED into the
 *   GEMMs that follow them -- LN(x) W^T + b = rstd (x W'^T - mean cs) + c with
 *   +12 W'_qkv = bf16(qkv.weight * layer_norm1.weight[None, :]) [3h, h]   +13 cs = row sums of W'_qkv (of the bf16 values), fp32 [3h]
 *   +14 c = qkv.weight @ layer_norm1.bias + qkv.bias, fp32 [3h]           +15 / +16 / +17 the same for mlp.fc1 with layer_norm2 [mlp]:
 *   the producer of the residual stream (projection / fc2 epilogue) then also writes bf16(x) and the rows' moments, and no stand-alone
 *   LayerNorm pass is left between the GEMMs of a layer (version 500) */
#define VTGB_VIT_NW_GLOBAL 6
#define VTGB_VIT_NW_LAYER 18
typedef struct {
    int32_t dtype, n_frames, image, patch, hidden, heads, mlp, layers;
    float eps;
    const float* pixel_values;  /* [n_frames, 3, image, image] fp32                    */
    const void* const* weights; /* host array                                          */
    float* out_f32;             /* [n_frames, tokens, hidden] fp32 or NULL             */
    void* out_act;              /* same, in `dtype`, or NULL (feeds vtgb_qformer)      */
    void* workspace;
    size_t workspace_bytes;
} vtgb_vit_args;
int32_t vtgb_vit_patch_kpad(int32_t dtype, int32_t patch);
size_t vtgb_vit_workspace_bytes(const vtgb_vit_args* a);
int vtgb_vit_forward(const vtgb_vit_args* a, vtgb_stream_t stream);

/* ---- K7-K10: Q-Former -----------------------------------------------------------------
 * Replaces InstructBlipQFormerModel.forward xinstructblip.py:1122-1242 (has_text = 1:
 * embeddings :1018-1046, layers :814-883, attention :611-694) and Blip2QFormerModel.forward
 * xblip2.py:1063-1174 (has_text = 0: layernorm(query_embeds) :1108).  Output: the query
 * rows [:, :n_query] of the last layer.
 * weights: has_text: [0] embeddings.word_embeddings.weight [1] embeddings.position_embeddings.weight
 *                    [2] embeddings.layernorm.weight [3] .bias
 *          else:     [0] NULL [1] NULL [2] layernorm.weight [3] layernorm.bias
 *   then per layer l at 4 + 32*l (entries of absent sub-blocks are NULL):
 *   +0..+5  attention.attention.{query,key,value}.{weight `dtype`, bias}
 *   +6 attention.output.dense.weight `dtype` +7 .bias +8 attention.output.LayerNorm.weight +9 .bias
 *   +10..+15 crossattention.attention.{query,key,value}.{weight,bias}  (key/value: [h, enc_hidden])
 *   +16 crossattention.output.dense.weight +17 .bias +18 crossattention.output.LayerNorm.weight +19 .bias
 *   +20 intermediate_query.dense.weight +21 .bias +22 output_query.dense.weight +23 .bias
 *   +24 output_query.LayerNorm.weight +25 .bias
 *   +26 intermediate.dense.weight +27 .bias +28 output.dense.weight +29 .bias
 *   +30 output.LayerNorm.weight +31 .bias                                              */
#define VTGB_QF_NW_GLOBAL 4
#define VTGB_QF_NW_LAYER 32
typedef struct {
    int32_t dtype, n_frames, n_query, n_text, hidden, heads, ffn, layers, cross_freq;
    int32_t enc_tokens, enc_hidden, has_text;
    float eps;
    const void* image_embeds;   /* [n_frames, enc_tokens, enc_hidden] in `dtype`        */
    const float* query_tokens;  /* [n_query, hidden] fp32                               */
    const int64_t* text_ids;    /* [n_frames, n_text] or NULL                           */
    const int64_t* text_mask;   /* [n_frames, n_text] (1 = attend) or NULL = all ones   */
    const int64_t* image_mask;  /* [n_frames, enc_tokens] or NULL = all ones            */
    const void* const* weights; /* host array                                           */
    float* out_f32;             /* [n_frames, n_query, hidden]                          */
    void* workspace;
    size_t workspace_bytes;
} vtgb_qformer_args;
size_t vtgb_qformer_workspace_bytes(const vtgb_qformer_args* a);
int vtgb_qformer_forward(const vtgb_qformer_args* a, vtgb_stream_t stream);

/* ---- K11: frame pooling + language_projection -----------------------------------------
 * mean: eval/utils/model.py:186-195 and the ragged `widths` form of
 * src/models/LSTP_Vicuna_IVT_module.py:244-249 (width 0 -> zero row, i.e. bias only);
 * concat: src/models/LSTP_module.py:477-481.
 * out: mean [n_clips, n_query, out_dim]; concat [sum(widths) * n_query, out_dim].      */
#define VTGB_POOL_MEAN 0
#define VTGB_POOL_CONCAT 1
typedef struct {
    int32_t dtype, n_clips, n_query, hidden, out_dim, mode;
    const float* query_out;  /* [sum(widths), n_query, hidden] fp32 */
    const int32_t* widths;   /* host array [n_clips]                */
    const void* proj_w;      /* `dtype` [out_dim, hidden]           */
    const float* proj_b;     /* [out_dim]                           */
    float* out;
    void* workspace;
    size_t workspace_bytes;
} vtgb_pool_project_args;
size_t vtgb_pool_project_workspace_bytes(const vtgb_pool_project_args* a);
int vtgb_pool_project(const vtgb_pool_project_args* a, vtgb_stream_t stream);

/* ---- K12-K15: Temporal Grounding Bridge encoder ---------------------------------------
 * Replaces RopeBertModel.forward src/models/components/xropebert.py:1048-1169 called with
 * encoder_embeds=of (TemporalOFEmbedding :103-129, RopeBertEmbeddings :190-208, rotary
 * attention :243-377, layers :450-533, mode switch :621-634, mrc_head :1164).
 * weights: [0] embeddings.word_embeddings.weight [1] embeddings.token_type_embeddings.weight
 *   [2] embeddings.LayerNorm.weight [3] .bias [4] temporal_embeddings.bos [5] .eos
 *   [6] temporal_embeddings.projection.weight fp32 [hidden, 2*patch*patch] [7] .bias
 *   [8] temporal_embeddings.fc.weight [9] .fc.bias [10] temporal_embeddings.frame_pos_embed.weight
 *   [11] temporal_embeddings.ln.weight [12] .ln.bias [13] encoder.embed_positions.weight
 *   [14] encoder.c_embed_positions.weight [15] mrc_head.weight fp32 [2, hidden] [16] mrc_head.bias
 *   then per layer l at 17 + 26*l (crossattention entries NULL below fusion_layer):
 *   +0..+5 attention.self.{query,key,value}.{weight `dtype`, bias}
 *   +6 attention.output.dense.weight `dtype` +7 .bias +8 attention.output.LayerNorm.weight +9 .bias
 *   +10..+15 crossattention.self.{query,key,value}.{weight,bias}
 *   +16 crossattention.output.dense.weight +17 .bias +18 crossattention.output.LayerNorm.weight +19 .bias
 *   +20 intermediate.dense.weight +21 .bias +22 output.dense.weight +23 .bias
 *   +24 output.LayerNorm.weight +25 .bias                                              */
#define VTGB_TGB_NW_GLOBAL 17
#define VTGB_TGB_NW_LAYER 26
#define VTGB_TGB_MODE_TEXT 0       /* layers [0, fusion_layer)  ("text" / "vision") */
#define VTGB_TGB_MODE_FUSION 1     /* layers [fusion_layer, layers)                 */
#define VTGB_TGB_MODE_MULTIMODAL 2 /* all layers                                    */
typedef struct {
    int32_t dtype, B, L, n_text, hidden, heads, ffn, layers, fusion_layer, mode, image, patch;
    float eps;
    const float* of;            /* [B, L, 2, image, image] fp32 */
    const int64_t* of_mask;     /* [B, L+2]                     */
    const int64_t* text_ids;    /* [B, n_text]                  */
    const int64_t* text_mask;   /* [B, n_text]                  */
    const void* const* weights; /* host array                   */
    float* seq_out;             /* [B, L+2, hidden] or NULL     */
    float* logits;              /* [B, L, 2]                    */
    void* workspace;
    size_t workspace_bytes;
} vtgb_tgb_args;
size_t vtgb_tgb_workspace_bytes(const vtgb_tgb_args* a);
int vtgb_tgb_forward(const vtgb_tgb_args* a, vtgb_stream_t stream);

/* ---- building blocks, exported for the per-kernel parity tests and the roofline bench ----
 * out[M, N] = epilogue(A[M, K] . W[N, K]^T + bias).  A, W in `dtype`; K % 8 == 0.       */
#define VTGB_EPI_STORE 0      /* out `dtype`  = acc + bias                      */
#define VTGB_EPI_GELU 1       /* out `dtype`  = gelu_erf(acc + bias)            */
#define VTGB_EPI_RESID_F32 2  /* out fp32     = acc + bias + resid (may alias)  */
#define VTGB_EPI_STORE_F32 3  /* out fp32     = acc + bias                      */
typedef struct {
    int32_t dtype, M, N, K, epilogue;
    const void* A;
    int64_t lda;
    const void* W;
    int64_t ldw;
    const float* bias;  /* [N] or NULL */
    const float* resid; /* [M, ldo] fp32 (VTGB_EPI_RESID_F32) */
    void* out;
    int64_t ldo;
} vtgb_gemm_args;
int vtgb_gemm(const vtgb_gemm_args* a, vtgb_stream_t stream);

/* softmax(scale * Q K^T + key_mask) V over [batch, heads]; Q/K/V/out token-major in `dtype`. */
typedef struct {
    int32_t dtype, batch, heads, head_dim, s_q, s_kv;
    const void* q;
    const void* k;
    const void* v;
    int64_t q_tok_stride, kv_tok_stride; /* elements between consecutive tokens        */
    int64_t q_batch_stride, kv_batch_stride;
    const float* key_mask;               /* additive [batch, s_kv] fp32 or NULL        */
    const float* rope_q;                 /* [>= s_q, head_dim] sin|cos table or NULL   */
    const float* rope_k;                 /* [>= s_kv, head_dim] or NULL                */
    float scale;
    void* out;                           /* [batch, s_q, heads*head_dim] in `dtype`    */
    int64_t out_tok_stride, out_batch_stride;
    int32_t causal;                      /* != 0: query q attends keys <= q + (s_kv - s_q) (the LLM's prefill) */
} vtgb_attention_args;
int vtgb_attention(const vtgb_attention_args* a, vtgb_stream_t stream);

/* LayerNorm over the last dim of fp32 rows; writes fp32 and/or `dtype` copies. */
typedef struct {
    int32_t dtype, M, D;
    float eps;
    const float* x;
    const float* gamma;
    const float* beta;
    float* out_f32; /* or NULL */
    void* out_act;  /* or NULL */
} vtgb_layernorm_args;
int vtgb_layernorm(const vtgb_layernorm_args* a, vtgb_stream_t stream);

/* ---- a2 / f1: RAFT (src/models/components/xraft.py:102-156) --------------------------------------
 * Three entry points cover RAFT.forward: vtgb_raft_encoder (fnet / cnet), vtgb_raft_corr (CorrBlock.__init__) and
 * vtgb_raft_update (the refinement loop, mask head and convex upsample).  `dtype` selects the arithmetic of all
 * three: VTGB_BF16 = bf16 MFMA implicit-GEMM convolutions over NHWC bf16 activations, fp32 accumulation, fp32
 * hidden state / flow / norms, IEEE-half correlation pyramid (a reduced-precision mode the reference does not have);
 * VTGB_F32 = fp32 operands and FMAs everywhere, k summed in order -- the exactness mode, which is how the reference
 * runs RAFT (xraft.py:118-119; since version 500 on the fp32-input matrix instruction, the same fmaf chain bit for bit).
 * VTGB_BF16X3 (version 500) = the reference's fp32 ACCURACY on the bf16 matrix cores: every convolution operand is a bf16 pair
 * (hi = bf16(x), lo = bf16(x - hi): 16 significant bits), x . w ~ hi . Wh + lo . Wh + hi . Wl with fp32 accumulation; activations
 * are stored as rows [hi(C) | lo(C)], gates / flow / correlation pyramid / lookup stay fp32 (the pyramid from split-bf16 products).
 * Flows within 1e-4 relative RMS of the reference's under input-sensitive weights (tests/test_gpu_raft.py), 3 x the MFMA work of VTGB_BF16.
 * Weight tables at VTGB_BF16X3: every MFMA convolution [C_out, taps, 3 C_in] in the K order below with the channel blocks
 * [Wh | Wh | Wl] per source of the (virtual) input concatenation (bf16; built by ops.split3); vtgb_raft_update: [4] convf1 as at VTGB_F32,
 * [10] .. [17] the GRU convolutions over [h(128) | motion(126) | flow(2)] and [26] .. [29] their `inp` parts (always present: the loop-invariant
 * third is computed once per call into fp32 start maps), [24] / [25] mask.2 with its 0.25 folded in; vtgb_raft_encoder: [0] the stem
 * [64, 4 (tY), 3 x 64] scaled by 2/255, the 1x1 head [256, 3 x 128].
 * VTGB_F16C8 (version 600; vtgb_raft_update and vtgb_raft_encoder -- the correlation volume of that mode runs at VTGB_BF16X3; vtgb_raft_encoder at
 * VTGB_F16C8 is the VTGB_BF16X3 encoder with the ten stride-1 3x3 convolutions of its residual blocks [both convolutions of layer1.0, layer1.1,
 * layer2.1, layer3.1 and conv2 of layer2.0, layer3.0: entries 2, 4, 8, 10, 16, 20, 22, 28, 32, 34] on these operands and weights[40] = DEVICE
 * int32 [12], the scale byte of block b's conv1 / conv2 at [2 b] / [2 b + 1]; the stem, the stride-2 convolutions, the 1x1 downsamples and the 1x1
 * head stay split-bf16): the nine large
 * convolutions of the update block ([0] convc1, [2] convc2, [8] conv, [10] / [12] / [14] / [16] the GRU's, [18] flow_head.conv1, [22] mask.0) take
 * their operands as  x . w ~ xh . Wh  (fp16 x fp16)  +  2^-11/sw (xl' . Wh8 + xh8 . Wl')  (OCP fp8 on the block-scaled matrix instruction, twice the
 * fp16 rate), xh = fp16(x), xl' = e5m2((x - xh) 2^11), xh8 = e5m2(x), Wh = fp16(w), Wh8 = e4m3(w sw), Wl' = e4m3((w - Wh) sw 2^11), sw a power of two
 * per layer: the corrections are 2^-11 of the product, so 3-4 significant bits on each side leave ~2^-16 -- the bf16 pair's level (tests/
 * test_gpu_raft.py holds this mode to the bf16x3 mode's bounds).  Their weights are [C_out, K] 16-bit units with K = per source (fp16 [taps, C] |
 * correction bytes [taps, C / 4 groups of (Wh8 x 4, Wl' x 4)]), each half in the 64-channel-chunk-major K order below (ops.h8_conv_pack);
 * weights[30] = DEVICE int32 [10]: the E8M0 byte of 2^-11 / sw of those nine convolutions in the order above, then of [6] convf2 (3x3, 128 -> 64: an
 * f16c8 convolution as well, its input -- convf1's output -- an f16c8 pair).  Everything else in the table is as at VTGB_BF16X3.  Activations beyond +-57344 saturate (e5m2's range; RAFT's are normalised features, gates and correlations of unit-scale features).
 * Convolution weights are packed [C_out, K] in `dtype` with K running 64-channel chunk
 * major, tap minor, channel-in-chunk innermost (written [C_out, KH,KW,C_in] below for the shapes only).
 *
 * vtgb_raft_update replaces the refinement loop xraft.py:135-156: per iteration CorrBlock.__call__
 * (raft_utils/corr.py:29-50; the reference's optional `alt_cuda_corr`, corr.py:63-91, is the CUDA counterpart of the
 * lookup kernel), BasicUpdateBlock (raft_utils/update.py:123-144: BasicMotionEncoder :75-97, SepConvGRU :39-65,
 * FlowHead :6-18), then the mask head and upsample_flow (xraft.py:88-99) of the last iteration.  At VTGB_BF16 the lookup and
 * convc1 are one launch (the 324 taps of a pixel never leave the CU).  n_pairs * H8 * W8 < 2^28 coarse pixels per call.
 * weights (host array of device pointers):
 *   [0] encoder.convc1.weight [256, 384] (324 input channels zero-padded to 384) [1] .bias
 *   [2] encoder.convc2.weight [192, 3,3,256] [3] .bias
 *   [4] encoder.convf1.weight -- VTGB_BF16: bf16 [128, 56 taps, {x,y,x,y}] (49 taps zero-padded; the flow enters as a
 *       bf16 head + bf16 remainder); VTGB_F32: fp32 [98, 128] (k = c*49 + ky*7 + kx, output channel minor)   [5] .bias
 *   [6] encoder.convf2.weight [64, 3,3,128]  [7] .bias   [8] encoder.conv.weight [126, 3,3,256]     [9] .bias
 *   [10] gru.convz1|convr1.weight [256, 1,5,384] [11] bias [256]  [12] gru.convq1.weight [128, 1,5,384] [13] .bias
 *   [14] gru.convz2|convr2.weight [256, 5,1,384] [15] bias [256]  [16] gru.convq2.weight [128, 5,1,384] [17] .bias
 *   [18] flow_head.conv1.weight [256, 3,3,128] [19] .bias  [20] flow_head.conv2.weight [32, 256], row tap*2+o (18 rows zero-padded) [21] .bias
 *   [22] mask.0.weight [256, 3,3,128] [23] .bias           [24] mask.2.weight [576, 256] [25] .bias
 *   [26..29] optional, VTGB_BF16 only (all four or none; NULL = the plain form above): the `inp` split of the GRU
 *       convolutions.  `inp` (input channels 128..255) does not change over the refinement iterations, so its contribution
 *       is computed once per call and the 80 GRU launches contract over the other 256 channels only.  With the split,
 *       [10] / [12] / [14] / [16] hold the [h(128) | motion(126) | flow(2)] channels: [256 or 128, taps, 256], and
 *       [26] gru.convz1|convr1 inp part [256, 1,5,128]   [27] gru.convq1 inp part [128, 1,5,128]
 *       [28] gru.convz2|convr2 inp part [256, 5,1,128]   [29] gru.convq2 inp part [128, 5,1,128]
 * Biases fp32.  GRU input channels are [h(128) | inp(128) | motion(126) | flow(2)] as in the reference. */
#define VTGB_RAFT_NW 30
typedef struct {
    int32_t dtype, n_pairs, H8, W8, iters;
    const float* net;           /* [n_pairs, 128, H8, W8] tanh(cnet[:, :128])   (xraft.py:126-127) */
    const float* inp;           /* [n_pairs, 128, H8, W8] relu(cnet[:, 128:])                      */
    const void* corr[4];        /* pyramid level l: [n_pairs*H8*W8, H8>>l, W8>>l] (corr.py:19-27), fp32 or fp16 (corr_f16) */
    const void* const* weights; /* host array                                                      */
    float* flow_up;             /* [n_pairs, 2, 8*H8, 8*W8]                                        */
    void* workspace;
    size_t workspace_bytes;
    int32_t corr_f16;           /* 0: fp32 pyramid; 1: IEEE half (what vtgb_raft_corr writes in VTGB_BF16 mode)  */
    const float* cnet_nhwc;     /* optional: the context encoder's raw output [n_pairs*H8*W8, 256] (vtgb_raft_encoder layout);
                                   net = tanh(first 128), inp = relu(last 128) are then taken from it and net/inp may be NULL */
    const float* flow_init;     /* optional [n_pairs, 2, H8, W8]: coords1 = coords0 + flow_init (xraft.py:131-132)    */
} vtgb_raft_update_args;
size_t vtgb_raft_update_workspace_bytes(const vtgb_raft_update_args* a);
int vtgb_raft_update(const vtgb_raft_update_args* a, vtgb_stream_t stream);

/* Unit-level surface of the VTGB_F16C8 operand format (version 600; what tests/test_gpu_h8.py checks piece by piece -- vtgb_raft_update is built from
 * these): vtgb_pair_pack writes fp32 rows [M, C] (C % 4 == 0) as pair rows [M, 2 * ld_pair 16-bit units] -- fmt VTGB_F16C8: fp16 values at unit c, the
 * eight correction bytes of channels 4g .. 4g+3 at unit ld_pair + 4g (csrc/pair_h8.h); fmt VTGB_BF16X3: bf16 hi at unit c, bf16 lo at unit ld_pair + c.
 * Channels [C, ld_pair) are written as zeros.  vtgb_pair_conv: one stride-1 "same" convolution (RAFT update block, raft_utils/update.py:75-97) over
 * f16c8 pair rows: out = act(conv(x) + bias) as a pair row again (out_fmt VTGB_F16C8 or VTGB_BF16X3), x = C1 channels from `a` (+ C1 more from `a2`:
 * a virtual concatenation), weights / scale as vtgb_raft_update's table holds them (ops.h8_conv_pack; scale = device int32: the E8M0 byte). */
int vtgb_pair_pack(int32_t fmt, const float* x, void* out, int64_t M, int32_t C, int32_t ld_pair, vtgb_stream_t stream);
typedef struct {
    int32_t M, N, H, W, KH, KW, C1;   /* M = images * H * W output pixels; N output channels (even); C1 % 64 == 0 input channels per source */
    const void* a;                    /* pair rows [M, 2 * C1 units] */
    const void* a2;                   /* optional second source, same width */
    const void* weights;              /* [N, taps * 2 * C1 * sources] 16-bit units */
    const int32_t* scale;             /* device: E8M0 byte of 2^-11 / sw */
    const float* bias;                /* optional [N] */
    int32_t act;                      /* 0 none, 1 relu */
    int32_t out_fmt;                  /* VTGB_F16C8 or VTGB_BF16X3 */
    void* out;                        /* pair rows [M, 2 * ld_out units], second half at unit ld_out */
    int32_t ld_out;                   /* >= N, % 4 == 0 */
} vtgb_pair_conv_args;
int vtgb_pair_conv(const vtgb_pair_conv_args* a, vtgb_stream_t stream);
/* The same convolution with one of the three other epilogues the RAFT launches use (version 601; unit tests of those epilogues): exactly one of
 *   resid    -- the ResidualBlock tail (extractor.py:56-60): out = relu(resid + act(conv(x) + bias)) as a pair row; resid = f16c8 pair rows [M, 2 * ld_resid
 *               units]; N <= 128, N % 4 == 0 (cnet's blocks at VTGB_F16C8);
 *   tail_w   -- FlowHead (update.py:10-18): N == 256; tail_out [M, 32] fp32 = act(conv(x) + bias) . W2, the 18 per-tap products of the following 3x3
 *               convolution (columns 18 .. 31: zero weights), formed from the accumulators in exact fp32; tail_w = W2 [256, 32] fp32 in the epilogue's lane
 *               order (ops.flow_tail_pack); `out` is not written;
 *   out_f32  -- fp32 rows [M, ld_f32] instead of a pair (N <= 128, N % 4 == 0, act == 0: fnet's convolutions in front of an InstanceNorm)
 * may be non-NULL (all NULL: vtgb_pair_conv). */
typedef struct {
    vtgb_pair_conv_args conv;
    const void* resid;
    int32_t ld_resid;
    const float* tail_w;
    float* tail_out;
    float* out_f32;
    int32_t ld_f32;
} vtgb_pair_conv_ex_args;
int vtgb_pair_conv_ex(const vtgb_pair_conv_ex_args* a, vtgb_stream_t stream);

/* CorrBlock.__init__ (raft_utils/corr.py:12-27; the all-pairs product :52-60): for every pair the correlation of each
 * pixel of image 1 with every pixel of image 2 over the `dim` = 256 features, divided by sqrt(dim), and its three
 * avg_pool2d(2, stride 2) -- one kernel, the four levels are its only output.  Pair n uses the feature maps of images
 * (n / pairs_per_clip) * frames_per_clip + n % pairs_per_clip + {first_off, second_off}: consecutive frames of whole
 * clips are (T-1, T, 0, 1); separate image1 / image2 batches of N encoded as cat([image1, image2]) are (N, N, 0, N).
 * VTGB_BF16: half-precision MFMA on an fp16 copy of the features (made in the workspace), fp32 accumulate and scale,
 * levels IEEE half; VTGB_F32: fp32 FMAs, levels fp32; VTGB_BF16X3: the features as bf16 pairs (hi | lo planes in the workspace),
 * three bf16 MFMA products per fp32 product, fp32 accumulate, levels fp32. */
typedef struct {
    int32_t dtype, n_pairs, H8, W8, dim;
    int32_t pairs_per_clip, frames_per_clip, first_off, second_off, n_images;
    float scale;                /* 1 / sqrt(dim) = 1/16                                                            */
    const float* fmap;          /* [n_images, H8*W8, dim] fp32 (vtgb_raft_encoder output)                           */
    void* levels[4];            /* level l: [n_pairs*H8*W8, H8>>l, W8>>l] half (VTGB_BF16) or fp32 (VTGB_F32, VTGB_BF16X3) */
    void* workspace;
    size_t workspace_bytes;
} vtgb_raft_corr_args;
size_t vtgb_raft_corr_workspace_bytes(const vtgb_raft_corr_args* a);
int vtgb_raft_corr(const vtgb_raft_corr_args* a, vtgb_stream_t stream);

/* RAFT BasicEncoder (raft_utils/extractor.py:116-189): stem 7x7/2, six ResidualBlocks, 1x1 head; the input
 * scaling 2*(x/255)-1 of RAFT.forward (xraft.py:105-106) is applied inside.  norm = 0: InstanceNorm2d (fnet);
 * norm = 1: the caller has folded the eval-mode BatchNorm2d that follows every convolution into the packed
 * weights and biases (cnet).  Output: NHWC features [n_images * H/8 * W/8, 256] fp32.
 * weights (`dtype`): [0] conv1.weight: the stem as a 4x1 convolution over the 2x2 space-to-depth image, channel =
 *   dX*12 + py*6 + px*3 + c (48, zero-padded to 64), ky = 2 tY + py - 1, kx = 2 dX + px - 1.  VTGB_F32: [64, 4 (tY), 64], unscaled
 *   (the kernel packs 2*(x/255)-1 itself).  VTGB_BF16: [64, 2 (chunk), 4 (tY), 64] = the same table twice, scaled by 2/255: the
 *   kernel feeds x - 127.5 to the GEMM as a bf16 pair, chunk 0 = hi = bf16(v), chunk 1 = lo = bf16(v - hi), so float-valued
 *   (CLIP-normalised) frames keep 16 significant bits (since version 300; 210 fed hi alone) [1] conv1.bias; per block b (layer1.0, 1.1, 2.0, 2.1, 3.0, 3.1) at
 *   2 + 6 b: conv1.weight [C, 3,3,Cin_pad] , conv1.bias, conv2.weight [C, 3,3,C_pad], conv2.bias,
 *   downsample.0.weight [C, Cin_pad] or NULL, downsample.0.bias or NULL  (96-channel stages: C_pad = 128, zero-filled; with norm = 1 the
 *   OUTPUT rows and biases of those stages are padded to C_pad as well: the GEMM stores the activations directly);
 *   [38] conv2.weight [256, 128] [39] conv2.bias */
#define VTGB_RAFT_ENC_NW 40
typedef struct {
    int32_t dtype, n_images, H, W, norm;
    const float* images;        /* [n_images, 3, H, W] fp32 */
    const void* const* weights; /* host array              */
    float* out;                 /* [n_images * H/8 * W/8, 256] fp32 */
    void* workspace;
    size_t workspace_bytes;
} vtgb_raft_encoder_args;
size_t vtgb_raft_encoder_workspace_bytes(const vtgb_raft_encoder_args* a);
int vtgb_raft_encoder(const vtgb_raft_encoder_args* a, vtgb_stream_t stream);

/* ---- LLM decode-step building blocks ("next" row) ----------------------------
 * The LLM itself is third-party on both sides (HF weights and GEMMs); these fuse the small
 * per-layer ops of a KV-cached greedy decode step with the exact rounding points of
 * transformers' modeling_llama (LlamaRMSNorm, apply_rotary_pos_emb, LlamaMLP).  `dtype` is the
 * activation dtype (VTGB_BF16 / VTGB_F32); `pos` is a DEVICE int64 (current position), so one
 * captured hipGraph serves every step.
 *   vtgb_llm_rmsnorm:          x[rows,H] (+= delta, written back if delta != NULL); h = w * norm(x)
 *   vtgb_llm_rope_cache:       qkv[B, nq+2nkv, hd] -> q_out[B,nq,hd] rotated; K/V cache [B,nkv,tmax,hd] row *pos (cos_t = sin_t = NULL: no rotary)
 *   vtgb_llm_rope_cache_prefill: qkv[B, S, (nq+2nkv)*hd]: q and k of every position rotated IN PLACE, k / v copied to cache rows 0..S-1
 *   vtgb_llm_decode_attention: out[B, nq*hd] = softmax(scale q K[0..*pos]^T) V[0..*pos]
 *   vtgb_llm_silu_mul:         act[rows, I] = silu(gu[:, :I]) * gu[:, I:]                       */
int vtgb_llm_rmsnorm(int dtype, void* x, const void* delta, const void* w, void* h, int64_t rows, int32_t H, float eps,
                     vtgb_stream_t stream);
int vtgb_llm_rope_cache(int dtype, const void* qkv, void* q_out, void* kc, void* vc, const void* cos_t, const void* sin_t,
                        const int64_t* pos, int32_t B, int32_t nq, int32_t nkv, int32_t hd, int32_t tmax, vtgb_stream_t stream);
int vtgb_llm_rope_cache_prefill(int dtype, void* qkv, void* kc, void* vc, const void* cos_t, const void* sin_t, int32_t B, int32_t S, int32_t nq,
                                int32_t nkv, int32_t hd, int32_t tmax, vtgb_stream_t stream);
int vtgb_llm_decode_attention(int dtype, const void* q, const void* kc, const void* vc, void* out, const int64_t* pos, int32_t B,
                              int32_t nq, int32_t nkv, int32_t hd, int32_t tmax, float scale, vtgb_stream_t stream);
int vtgb_llm_silu_mul(int dtype, const void* gu, void* act, int64_t rows, int32_t I, vtgb_stream_t stream);

/* The seq2seq language model of the BLIP-2 flavours (Flan-T5 under `language_model.generate`: eval/utils/model.py:427-437,
 * src/models/components/xblip2.py:1553-1556) with transformers' modeling_t5 arithmetic: T5LayerNorm = vtgb_llm_rmsnorm (no mean, no
 * bias), projections = vtgb_gemm / vtgb_gemm_skinny, cache append = vtgb_llm_rope_cache with cos_t = sin_t = NULL, and
 *   vtgb_llm_attention_rows: out[r, h, :] = softmax_k(scale q[r, h] . k[b, h, key] + bias[qpos, h, key]) v[b, h, key, :] over independent
 *     query rows r (b = r / rows_per_batch).  Keys [0, n_keys), or [0, *pos] with bias row *pos when `pos` (a DEVICE int64) is given --
 *     the decoder's self-attention over its static cache; pos == NULL: query position r % rows_per_batch -- the encoder's
 *     self-attention (rows = B x P, token-major q|k|v) and the decoder's cross-attention (bias NULL).  All strides in elements;
 *     heads are `head_dim` apart in q / out; `bias` has the activation dtype; scores and weights are rounded to it where HF does
 *     (softmax itself in fp32).  t_pad >= the largest key count (sizes the kernel's score buffer).
 *   vtgb_llm_gated_act: act[r, i] = f(gu[r, i]) (* gu[r, I + i] if gated); kind 0 SiLU, 1 gelu_new (Flan-T5's gated-gelu), 2 ReLU, 3 erf-GELU */
typedef struct {
    int32_t dtype, rows, heads, head_dim, rows_per_batch, n_keys, t_pad;
    float scale;
    const void* q;   int64_t q_row;
    const void* k;   const void* v;   int64_t kv_batch, kv_head, kv_tok;
    const void* bias; int64_t bias_pos, bias_head;
    const int64_t* pos;
    void* out;       int64_t o_row;
} vtgb_llm_attn_rows_args;
int vtgb_llm_attention_rows(const vtgb_llm_attn_rows_args* a, vtgb_stream_t stream);
int vtgb_llm_gated_act(int dtype, const void* gu, void* act, int64_t rows, int32_t I, int32_t kind, int32_t gated, vtgb_stream_t stream);

/* Skinny GEMM of the decode step: out[M, N] = x[M, K] . w[N, K]^T, bf16 operands, M <= 128 (one token per
 * clip), K a multiple of 64 -- what `F.linear(h, weight)` (hipBLASLt) computes under `language_model.generate`
 * (eval/utils/model.py:223-233; transformers LlamaDecoderLayer's q/k/v/o/gate/up/down projections and lm_head).  The weights
 * stream from HBM once: one workgroup per (128-row weight tile, K split) -- `n_splits` splits (0 = chosen by the library) where
 * the tiles alone would leave CUs without a stream.  Without a split the tile is rounded and stored straight to `out` (no
 * workspace: vtgb_gemm_skinny_workspace_bytes returns 0).  With a split every workgroup leaves an fp32 fragment in `workspace`
 * (n_tiles * n_splits * M * 128 * 4 bytes) and a second launch adds a tile's fragments in split order and rounds once to
 * `out_dtype` (deterministic: no atomics). */
typedef struct {
    int32_t M, N, K, n_splits;
    const void* x; int64_t ldx;          /* bf16 [M, K] */
    const void* w; int64_t ldw;          /* bf16 [N, K] (nn.Linear.weight) */
    void* out; int64_t ldo;              /* [M, N] */
    int32_t out_dtype;                   /* VTGB_BF16 | VTGB_F32 */
    int32_t w_tiled;                     /* 1: `w` was prepared by vtgb_pack_skinny_weight (ldw ignored): every (128-row tile, 64-deep
                                            k-tile) is one contiguous 16 KiB block in the kernel's LDS order -- 1 KiB reads instead of
                                            eight 128-byte row segments a DRAM page apart */
    void* workspace;
    size_t workspace_bytes;
    int32_t defer_reduce;                /* 1 (with a K split): skip the second launch -- `out` is not written; the consumer adds the fragments
                                            workspace[(tile * n_splits + split) * M + m][128] in split order and rounds once itself
                                            (vtgb_llm_rmsnorm_parts, vtgb_llm_rope_cache_parts; vtgb_gemm_skinny_splits tells n_splits) */
} vtgb_gemm_skinny_args;
int32_t vtgb_gemm_skinny_splits(const vtgb_gemm_skinny_args* a);   /* the K split the call will use (1: `out` is written directly) */
/* The decode step's consumers of a deferred split (bf16; rows = the GEMM's M <= 128): x += sum of the fragments (rounded once, as the reduce
 * launch would have stored it), h = rmsnorm(x) * w;  and rotary + cache append reading q | k | v from the fragments.  Same values as the two-launch
 * form, bit for bit (tests/test_decode.py). */
int vtgb_llm_rmsnorm_parts(int dtype, void* x, const float* part, int32_t n_splits, const void* w, void* h, int64_t rows, int32_t H, float eps,
                           vtgb_stream_t stream);
int vtgb_llm_rope_cache_parts(int dtype, const float* part, int32_t n_splits, void* q_out, void* kc, void* vc, const void* cos_t, const void* sin_t,
                              const int64_t* pos, int32_t B, int32_t nq, int32_t nkv, int32_t hd, int32_t tmax, vtgb_stream_t stream);
size_t vtgb_pack_skinny_weight_bytes(int32_t N, int32_t K);
int vtgb_pack_skinny_weight(const void* w, int64_t ldw, int32_t N, int32_t K, void* dst, vtgb_stream_t stream);
size_t vtgb_gemm_skinny_workspace_bytes(const vtgb_gemm_skinny_args* a);
int vtgb_gemm_skinny(const vtgb_gemm_skinny_args* a, vtgb_stream_t stream);

/* ---- attention for the TRAINABLE stages (config C5, SF flavours): forward + backward, fp32 ---------------------------
 * Replaces, with its autograd, the matmul / softmax / dropout / matmul sequence of InstructBlipQFormerMultiHeadAttention.forward
 * (xinstructblip.py:611-694; BLIP-2 twin xblip2.py) and RopeBertSelfAttention.forward (xropebert.py:243-332; the rotary
 * embedding is applied by the caller): out = dropout(softmax(q k^T * scale + key_mask)) v per (batch, head).  `drop` is the
 * multiplicative dropout mask on the probabilities (0 or 1 / (1 - p); NULL = no dropout): injectable, like the Gumbel noise.
 * q / k / v / out and their gradients are token-major [batch, s, heads * head_dim] with element strides (channel stride 1);
 * dq / dk / dv use the strides of q / k / v, dout those of out.  lse [batch, heads, s_q] is written by the forward and read by
 * the backward; delta is backward scratch of the same size.  head_dim <= 128. */
typedef struct {
    int32_t batch, heads, head_dim, s_q, s_kv;
    const float* q;
    const float* k;
    const float* v;
    int64_t q_tok, kv_tok, q_batch, kv_batch;
    const float* key_mask; /* additive fp32 [batch, s_kv] or NULL */
    const float* drop;     /* [batch, heads, s_q, s_kv] or NULL   */
    float scale;
    float* out;
    int64_t o_tok, o_batch;
    float* lse;
    const float* dout; /* backward only from here */
    float* dq;
    float* dk;
    float* dv;
    float* delta;
} vtgb_attn_train_args;
int vtgb_attn_train_forward(const vtgb_attn_train_args* a, vtgb_stream_t stream);
int vtgb_attn_train_backward(const vtgb_attn_train_args* a, vtgb_stream_t stream);

/* ---- the rest of the training graph (config C5 / the SF flavours) -------------------------------------
 * vtgb_gemm_train: out[M, N] fp32 = op(a) . op(b) (+ bias[N]) for the three GEMMs of a linear layer under autograd
 *   (nn.Linear inside InstructBlipQFormerLayer xinstructblip.py:814-883, RopeBertLayer xropebert.py:450-533 and their backward):
 *     forward  y  = x W^T     a = x  [M, K] k-contiguous, b = W  [N, K] k-contiguous
 *     dgrad    dX = dY W      a = dY [M, N'] k-contiguous (contraction N'), b = W read k-MAJOR (element (k_out, n') at W + n' * ldw + k_out)
 *     wgrad    dW = dY^T X    a = dY read k-major (element (n, m) at dY + m * ld + n), b = X read k-major
 *   x_kmajor = 0: element (row, k) at p + row * ld + k;  1: at p + k * ld + row.  x_dtype: storage, VTGB_F32 or VTGB_BF16 -- operands are
 *   used where they lie (no transposed or converted copy).  compute = VTGB_BF16: operands rounded to bf16 on the way into LDS, MFMA, fp32
 *   accumulation;  VTGB_F32: fp32 FMA with the contraction summed in index order (the exactness mode).  Any M, N, K, ld (16-byte aligned
 *   pointers and ld take the vector path).  Few output tiles under a long contraction (weight gradients): with the workspace given the
 *   contraction is cut into slices, summed afterwards in slice order (deterministic); without it the launch is unsplit.
 * vtgb_col_sum_f32: out[n] = sum_m x[m, n] in a fixed order (bias gradients); `partial` = fp32 workspace [vtgb_col_sum_parts(M), N].
 * vtgb_layernorm_train_forward: sum = x * mask + resid (mask = multiplicative dropout mask or NULL; resid or NULL; `sum` may be NULL when
 *   both are), y = LayerNorm(sum) * gamma + beta, mean / rstd [rows] kept for the backward.  _backward: from dy, sum (= x when there was no
 *   mask and no residual), mean, rstd, gamma -> ds (gradient of sum = of resid), dx = ds * mask (only with a mask; else dx = ds), dgamma,
 *   dbeta [D] -- summed per workgroup in row order into `partial` [vtgb_layernorm_train_partials(rows), 2, D] and then over the partials
 *   in order (deterministic).  D % 4 == 0, D <= 2048.
 * vtgb_gelu_forward / _backward: y = x Phi(x) (erf form, F.gelu's default) and dx = dy (Phi(x) + x phi(x)); n fp32 values. */
typedef struct {
    int32_t compute, M, N, K;
    const void* a;
    int64_t lda;
    int32_t a_dtype, a_kmajor;
    const void* b;
    int64_t ldb;
    int32_t b_dtype, b_kmajor;
    const float* bias; /* [N] or NULL */
    float* out;
    int64_t ldo;
    void* workspace;   /* vtgb_gemm_train_workspace_bytes() bytes, or NULL */
    size_t workspace_bytes;
} vtgb_gemm_train_args;
size_t vtgb_gemm_train_workspace_bytes(const vtgb_gemm_train_args* a);
int vtgb_gemm_train(const vtgb_gemm_train_args* a, vtgb_stream_t stream);
int32_t vtgb_col_sum_parts(int32_t M);
int vtgb_col_sum_f32(const float* x, int64_t ldx, int32_t M, int32_t N, float* out, float* partial, vtgb_stream_t stream);
typedef struct {
    int32_t rows, D;
    float eps;
    const float* x;
    const float* mask;
    const float* resid;
    const float* gamma;
    const float* beta;
    float* sum;
    float* y;
    float* mean;
    float* rstd;
    const float* dy; /* backward only from here */
    float* ds;
    float* dx;
    float* dgamma;
    float* dbeta;
    float* partial;
} vtgb_layernorm_train_args;
int32_t vtgb_layernorm_train_partials(int32_t rows);
int vtgb_layernorm_train_forward(const vtgb_layernorm_train_args* a, vtgb_stream_t stream);
int vtgb_layernorm_train_backward(const vtgb_layernorm_train_args* a, vtgb_stream_t stream);
int vtgb_gelu_forward(const float* x, float* y, int64_t n, vtgb_stream_t stream);
int vtgb_gelu_backward(const float* x, const float* dy, float* dx, int64_t n, vtgb_stream_t stream);

/* ---- gradient exchange (config C5; the one collective of the path) ------------------------------------------------
 * Thin wrapper over RCCL (SURVEY.md 8b "later: vtgb_allreduce_f32", 8e): replaces what Lightning's DDPStrategy does for the reference
 * (configs/trainer/ddp.yaml:4, find_unused_parameters: the sum / mean all-reduce of the trainable gradients once per optimizer
 * step).  One communicator per process (= per GPU); rank 0 creates the id with vtgb_comm_unique_id and the host side hands
 * its VTGB_COMM_ID_BYTES bytes to every rank (any channel: torch.distributed's store, a file, MPI); vtgb_comm_init is
 * collective.  vtgb_allreduce_f32 reduces `count` floats IN PLACE, asynchronously on `stream` (sum, or mean when `average`):
 * the caller allocates its gradients as views of one flat buffer and reduces it in a few large pieces -- xGMI is
 * point-to-point, ring collectives are per-link bound, large messages amortise the ring latency.  RCCL is bound with
 * dlopen at the first call; processes that never call these functions never load it. */
#define VTGB_COMM_ID_BYTES 128
typedef struct vtgb_comm vtgb_comm;
int vtgb_comm_unique_id(void* id_out /* host, VTGB_COMM_ID_BYTES */);
int vtgb_comm_init(vtgb_comm** comm, const void* unique_id /* host */, int32_t rank, int32_t world);
int vtgb_comm_destroy(vtgb_comm* comm);
int vtgb_allreduce_f32(vtgb_comm* comm, float* buf, size_t count, int32_t average, vtgb_stream_t stream);

/* ---- in-library launch timing (used by bench.py for the roofline figure) -------------------
 * While enabled, every GEMM / attention launch of the bf16 path is bracketed by a pair of HIP
 * events recorded on the launch stream (no synchronisation at record time).  vtgb_prof_summary
 * synchronises on the recorded events and returns launch count, summed kernel time and summed
 * algorithmic FLOPs (2*M*N*K per GEMM / convolution of the reference's form, 4*B*H*Sq*Skv*hd per attention) of one kind
 * since the last vtgb_prof_reset. */
#define VTGB_PROF_GEMM 0   /* plain GEMM launches (ViT / Q-Former / TGB / projections)             */
#define VTGB_PROF_ATTN 1
#define VTGB_PROF_CONV 2   /* implicit-GEMM convolution launches of the same kernel (RAFT)          */
void vtgb_prof_enable(int on);
void vtgb_prof_reset(void);
int vtgb_prof_summary(int kind, int64_t* launches, double* ms, double* flops);
/* FLOPs the recorded launches of `kind` actually EXECUTED: smaller than the algorithmic figure of vtgb_prof_summary where
 * loop-invariant work has been hoisted out of a launch (RAFT's GRU convolutions: the `inp` third, see vtgb_raft_update). */
int vtgb_prof_executed_flops(int kind, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* VTGB_H */
