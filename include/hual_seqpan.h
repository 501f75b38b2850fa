/* libhual_seqpan.so - C ABI of the MI355X-native SeqPAN hot path of renjie-liang/HUAL.
 *
 * The reference has no FFI, plugin or operator registry: its hot path is a TensorFlow-1 graph
 * (/root/reference/models/model.py:7-122) entered only through `sess.run(fetches, feed_dict)` at
 *   /root/reference/utils/runner_utils.py:147   [train_op, loss, start_index, end_index]
 *   /root/reference/utils/runner_utils.py:166   [start_index, end_index]
 *   /root/reference/utils/runner_utils.py:75-81 match_scores / [start_logits, end_logits]
 * This header is the boundary a binding for that path would sit on: the feeds of
 * model.py:15-27 (`_add_placeholders`) are `hual_batch` + `hual_labels`, the fetches are `hual_outputs`,
 * `train_op` (models/ops.py:119-132) is hual_seqpan_backward + hual_adamw_clip_step.
 *
 * Conventions
 *  - extern "C", plain C structs, plain pointers and sizes; no torch / C++ types.
 *  - EVERY pointer is DEVICE memory owned by the caller (parameters, inputs, outputs, workspace);
 *    the library never allocates, frees or synchronises.  All work is enqueued on `stream`
 *    (a hipStream_t passed as void*), so a whole step can be captured into a hipGraph.
 *  - return 0 on success, negative on error; message via hual_last_error() (thread local).
 *    No C++ exception crosses the ABI.
 *  - re-entrant: the compute entry points keep no global mutable state; one stream per device/rank is safe.  The one exception
 *    is the OPTIONAL per-kernel timing of bench.py's roofline leg (hual_prof_begin / hual_prof_end / hual_prof_get below): a
 *    thread-local recorder that is off unless armed, and whose hual_prof_end() is the only call that synchronises.
 *  - kernels are specialised for gfx950 and model.dim = 128, num_heads = 8 (the value in both
 *    configs/<task>/SeqPAN.yaml); other values are rejected by hual_seqpan_validate().
 */
#ifndef HUAL_SEQPAN_H
#define HUAL_SEQPAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HUAL_ABI_VERSION 8

#define HUAL_OK 0
#define HUAL_ERR_INVALID (-1)
#define HUAL_ERR_HIP (-2)
#define HUAL_ERR_UNSUPPORTED (-3)
#define HUAL_ERR_WORKSPACE (-4)

/* dropout call-site ids (Philox counter word c2); mirrors oracle/philox.py SITE_* and DESIGN.md */
#define HUAL_SITE_WORD 0
#define HUAL_SITE_CHAR 1
#define HUAL_SITE_VIDEO 2
#define HUAL_SITE_CONV 3
#define HUAL_SITE_DA 8
#define HUAL_SITE_TRI 24
#define HUAL_SITE_GUMBEL 28   /* gumbel noise of the matching head (layers.py:163-166): RNG row = b*T+t, one call = the 4 classes */
#define HUAL_SITE_FE 32

/* keys of configs/<task>/SeqPAN.yaml read by models/model.py (model.py:17,36-43,61,83,101,122) */
typedef struct hual_cfg {
  int32_t vdim;        /* model.vdim      */
  int32_t dim;         /* model.dim  (128) */
  int32_t num_heads;   /* model.num_heads (8) */
  int32_t word_dim;    /* model.word_dim  */
  int32_t char_dim;    /* model.char_dim  */
  int32_t max_vlen;    /* model.max_vlen = rows of both position tables */
  int32_t attn_layer;  /* model.attn_layer */
  int32_t num_chars;   /* configs.num_chars */
  int32_t num_words;   /* rows of [zero; unk; word_table] */
  int32_t no_gumbel;   /* loss.no_gumbel (both YAMLs set true); 0: gumbel noise on the matching logits, (logits + noise) / tau - needs rng_state */
  float match_lambda;  /* loss.match_lambda */
  float tau;           /* loss.tau (> 0; unused when no_gumbel) */
  float clip_norm;     /* train.clip_norm */
} hual_cfg;

int hual_abi_version(void);
const char* hual_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Parameters.  One flat fp32 device buffer in the layout reported by hual_seqpan_param_table():
 * every trainable TF variable of models/model.py under its TF scope name and TF
 * shape, each tensor 16-byte aligned.  Gradients and both Adam slots use the same layout.
 * The frozen GloVe table `word_embs/word_table` [num_words-2, word_dim] (modules.py:10) is separate.
 * ------------------------------------------------------------------------------------------ */
typedef struct hual_param_entry {
  char name[112];
  uint64_t offset;       /* in floats */
  uint64_t size;         /* in floats */
  int32_t ndim;
  int32_t shape[4];
  int32_t decay;         /* 1: AdamWeightDecay applies weight decay (ops.py:123,176-184) */
} hual_param_entry;

int hual_seqpan_validate(const hual_cfg* cfg);
/* *padded_floats = size of the flat buffer, *count = number of trainable scalars (1,186,508 for Charades) */
int hual_seqpan_param_count(const hual_cfg* cfg, uint64_t* padded_floats, uint64_t* count);
/* returns the number of entries (fills at most max_entries) or a negative error */
int hual_seqpan_param_table(const hual_cfg* cfg, hual_param_entry* out, int max_entries);

/* feeds of model.py:15-27 (`_add_placeholders`); all device pointers */
#define HUAL_DTYPE_F32 0
#define HUAL_DTYPE_BF16 1
typedef struct hual_batch {
  const void* video;              /* video_inputs [B,T,vdim] of `video_dtype`, rows beyond video_seq_len zero padded */
  const int32_t* video_seq_len;   /* [B]; max must equal T (model.py:31) */
  const int32_t* word_ids;        /* [B,L], 0 = PAD, 1 = unk */
  const int32_t* char_ids;        /* [B,L,C], 0 = PAD, C >= 4 */
  int32_t B, T, L, C;
  int32_t video_dtype;            /* HUAL_DTYPE_F32 (the reference's float32 placeholder, model.py:17) or HUAL_DTYPE_BF16:
                                     bfloat16 clip features (BASELINE.json configs[1]), read as the float32 values they
                                     are - half the bytes of the feature-load phase, same arithmetic behind the load.
                                     Needs vdim % 256 == 0 and vdim <= 1024 (the K-split feature-load kernel). */
} hual_batch;

typedef struct hual_labels {
  const float* y1;                /* start_indexes f32 [B,T] (soft labels) */
  const float* y2;                /* end_indexes   f32 [B,T] */
  const int32_t* match_labels;    /* i32 [B,T] in 0..3 */
  const float* inner_labels;      /* f32 [B,T] */
} hual_labels;

/* fetches used by runner_utils.py:75-81,147,166 */
typedef struct hual_outputs {
  float* start_logits;            /* f32 [B,T] raw (unmasked beyond v_len, as the reference) */
  float* end_logits;              /* f32 [B,T] */
  float* match_scores;            /* f32 [B,T,4] */
  int64_t* start_index;           /* i64 [B] */
  int64_t* end_index;             /* i64 [B] */
  float* loss_terms;              /* f32 [4]: loss, loc_loss, match_loss, align_loss (written only with labels) */
} hual_outputs;

typedef struct hual_run_opts {
  float drop_rate;                /* the `dropout_rate` placeholder (0 = inference) */
  const uint32_t* rng_state;      /* device u32[3] = {seed lo, seed hi, offset}; may be NULL when drop_rate == 0 and cfg.no_gumbel */
  float match_denom_override;     /* > 0: denominator of the masked matching loss (exact data parallel) */
  int32_t align_external;         /* 1: the [B,B] alignment loss is evaluated by the caller through
                                        hual_align_loss() on gathered features (exact data parallel) */
  int32_t static_tables;          /* 1: the caller guarantees that a previous hual_seqpan_backward ran with the SAME
                                        cfg, shapes, params / grads / workspace / batch pointers, so the device-resident
                                        job tables it left in the workspace are still valid and are not rewritten
                                        (saves six tiny launches per step inside a replayed hipGraph) */
  const float* match_denom_dev;   /* non-NULL: the denominator of the masked matching loss is read from this DEVICE scalar
                                        when the kernels run (takes precedence over match_denom_override): a data-parallel
                                        step can all-reduce the valid-frame count on the stream without a host round trip */
  int32_t debug_taps;             /* 1: the forward also writes the tensors that only parity tests read (the relu outputs of the
                                        conv_block layers, "cb.y*" / "fe*.y*" of the workspace table); the backward pass never
                                        reads them (it reads the bit planes "*.rb*" / "*.kb*") */
  float* grads_prezero;           /* non-NULL together with prezero_token, hual_seqpan_forward with labels: this flat gradient
                                        buffer is zeroed by the forward's first launch (one launch fewer per step) */
  uint64_t* prezero_token;        /* HOST word owned by the caller, the receipt of that zeroing: the forward stores the address of
                                        the buffer it zeroed there once the launch is enqueued; hual_seqpan_backward skips its own
                                        zeroing launch only if the word holds the address of its `grads`, and clears it.  Any
                                        sequence that breaks the pairing (two backward calls, a forward that failed or had no
                                        labels) therefore zeroes the bucket in backward as before.  NULL: no pre-zeroing. */
  float* deferred_loss_terms;     /* non-NULL (ABI 7), a TRAIN step whose backward call follows on the same stream: the forward leaves the
                                        closing of the loss (matching-loss denominator, the four reported terms) to the backward pass,
                                        which writes float[4] = {total, loc, match, align} HERE from inside its matching-head launch -
                                        one launch fewer per step.  hual_outputs.loss_terms is then not written by the forward.
                                        Pass the same options to both calls.  NULL: the forward closes the loss itself (a launch). */
  void* dw_table;                 /* non-NULL (ABI 8): caller-owned DEVICE storage of >= hual_seqpan_dw_table_bytes() bytes, 16-byte aligned,
                                        for the job table of the backward's weight-gradient launch INSTEAD of the copy inside the
                                        workspace.  A caller that runs several padded shapes in ONE workspace (the epoch loop of
                                        runner_utils.py:139-159) keeps one such table per shape: `static_tables` then holds per
                                        TABLE - whatever other shapes did to the workspace in between - and every replayed step graph
                                        drops the table-writing launches.  NULL: the table lives in the workspace. */
  uint64_t dw_table_bytes;        /* size of dw_table (checked) */
} hual_run_opts;

/* bytes a hual_run_opts.dw_table must hold (any cfg, any shape) */
uint64_t hual_seqpan_dw_table_bytes(void);

/* bytes of workspace needed for one forward(+backward) of this shape */
int hual_seqpan_query_workspace(const hual_cfg* cfg, int B, int T, int L, int C, uint64_t* bytes);

/* named intermediate tensors inside the workspace (debugging / parity taps) */
typedef struct hual_ws_entry {
  char name[48];
  uint64_t offset;       /* bytes */
  uint64_t rows, cols;   /* fp32 elements */
} hual_ws_entry;
int hual_seqpan_ws_table(const hual_cfg* cfg, int B, int T, int L, int C, hual_ws_entry* out, int max_entries);

/* the graph of model.py:29-118: all five fetch tensors in ONE pass (the reference runs five).  With `labels`
 * it also evaluates model.py:76-120 (losses) and keeps what backward needs in the workspace.
 * Dense weights travel as fp16 hi + lo images scaled by 2^10: a weight with |w| >= 63 does not fit.  The pass does not fail
 * silently on one: with labels the four loss terms are NaN, without labels the start / end logits are NaN and the span
 * indices -1. */
int hual_seqpan_forward(const hual_cfg* cfg, const float* params, const float* word_table, const hual_batch* batch,
                        const hual_labels* labels, const hual_outputs* out, const hual_run_opts* opts, void* workspace,
                        uint64_t ws_bytes, void* stream);

/* tf.gradients(loss, tvars) (ops.py:126): fills `grads` (flat layout, overwritten).  Must follow a forward with
 * labels on the same workspace, batch and rng_state. */
int hual_seqpan_backward(const hual_cfg* cfg, const float* params, const float* word_table, const hual_batch* batch,
                         const hual_labels* labels, const hual_run_opts* opts, float* grads, void* workspace,
                         uint64_t ws_bytes, void* stream);

/* clip_by_global_norm + AdamWeightDecayOptimizer.apply_gradients (ops.py:127-132,149-174).
 * decay: per-element weight decay rate in the flat layout; lr: device scalar; grad_prescale multiplies the
 * gradient first (1/world after a sum all-reduce); sqnorm: device scratch of 256 floats. */
int hual_adamw_clip_step(float* params, const float* grads, float* adam_m, float* adam_v, const float* decay,
                         uint64_t n_padded, const float* lr, float clip_norm, float grad_prescale, float* sqnorm,
                         void* stream);

/* the same, and the Philox offset rng_state[2] of the training loop is advanced by one in the same launch (the step
 * counter of the dropout stream: one launch fewer per captured step than a separate increment) */
int hual_adamw_clip_step_rng(float* params, const float* grads, float* adam_m, float* adam_v, const float* decay,
                             uint64_t n_padded, const float* lr, float clip_norm, float grad_prescale, float* sqnorm,
                             uint32_t* rng_state, void* stream);

/* the same for an EPOCH LOOP whose position lives on the device (ABI 8; runner_utils.py:139-159): `cursor` = device i64[2] =
 * {ids consumed so far, words written to the span bank so far}.  The launch also copies `span_words` 8-byte words from `spans` (the step's
 * predicted start / end indices in the caller's fetch buffer) to bank + cursor[1], then cursor[0] += sel_inc, cursor[1] += bank_inc.
 * Together with hual_assemble_batch_cursor (which reads its batch's ids at ids + cursor[0]) a whole step - batch assembly, forward,
 * backward, optimizer, span banking - has the SAME arguments every time it runs with a given padded shape: one hipGraph per shape,
 * nothing launched between two graphs, nothing uploaded or fetched until the epoch ends. */
int hual_adamw_clip_step_loop(float* params, const float* grads, float* adam_m, float* adam_v, const float* decay,
                              uint64_t n_padded, const float* lr, float clip_norm, float grad_prescale, float* sqnorm,
                              uint32_t* rng_state, int64_t* cursor, const int64_t* spans, int64_t* bank, int span_words, int sel_inc,
                              int bank_inc, void* stream);

/* ------------------------------------------------------------------------------------------
 * One-shot all-reduce of the flat gradient bucket over peer mappings (csrc/xgmi.hip; absent in the reference, which
 * pins one GPU: utils/runner_utils.py:11).  ONE launch per rank: flag barrier, reduce-scatter read straight from the peers' buckets
 * (rank order: identical bits on every rank), flag barrier, all-gather.  flat / scratch / flags: `world` device pointers each - entry
 * `rank` the rank's own memory, the others the peers' memory mapped into this process (hipIpcOpenMemHandle); scratch holds
 * ceil(n / world) floats rounded up to 4; flags are hual_xgmi_flags_bytes() of UNCACHED device memory (the setup helpers below own that
 * allocation - the one exception to "the library never allocates": the caller's allocator cannot provide it), zeroed once; seq: local
 * device u32, zero at start, advanced by the call itself (graph replays included); status: local device u32, 0 = ok, 1 = a peer did not
 * arrive within the spin limit (all waves left the kernel, the bucket is garbage).  Off by default (hual_amd/dist.py: HUAL_ALLREDUCE=custom). */
uint64_t hual_xgmi_flags_bytes(void);
int hual_xgmi_flags_alloc(void** p);
int hual_xgmi_flags_free(void* p);
int hual_xgmi_ipc_export(void* p, void* handle64, uint64_t* offset);      /* hipIpcGetMemHandle of p's allocation: 64 bytes + p's offset in it */
int hual_xgmi_ipc_open(const void* handle64, void** p);      /* hipIpcOpenMemHandle in ANOTHER process than the exporter's: the allocation's base */
int hual_xgmi_ipc_close(void* p);
int hual_xgmi_allreduce(int rank, int world, void* const* flat, void* const* scratch, void* const* flags, uint32_t* seq, uint32_t* status,
                        uint64_t n, uint64_t scratch_floats, void* stream);

/* cross-sample part of lossfun_aligment (layers.py:232-247) on [Bg,128] l2-normalised features
 * (all-gathered over ranks in exact data-parallel mode).  scratch: 2*Bg*Bg + Bg floats.
 * Writes d_that / d_vhat [Bg,128] (scaled by grad_scale) and WRITES the loss to *loss (device scalar: the row terms are summed
 * in row order by the second launch - no zeroing launch in front, no atomics). */
int hual_align_loss(const float* that, const float* vhat, int Bg, float* scratch, float* d_that, float* d_vhat,
                    float* loss, float grad_scale, void* stream);
/* the same for a rank of a data-parallel group: that / vhat are rows of stride `ld` floats (256 when the all-gather left
 * [that | vhat] side by side - the layout of the workspace buffer "align.tv" [B,256] the forward leaves), and only the gradient rows row0 .. row0 + nrows - 1 (the rank's own samples) are written,
 * to d_that / d_vhat [nrows,128] - straight into the workspace buffers "d.align.that" / "d.align.vhat" of the backward. */
int hual_align_loss_rows(const float* that, const float* vhat, int ld, int Bg, int row0, int nrows, float* scratch, float* d_that,
                         float* d_vhat, float* loss, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-block entry points: ONE block of the graph on caller-supplied activations, enqueued through the same launch
 * sequence the whole model uses (unit parity against the corresponding function of the reference).
 * Activations live in the unified row space: [B*T video rows, then B*L query rows] x 128 floats.  `batch` supplies the
 * shapes and the masks (video_seq_len, word_ids); `workspace` is the model workspace (hual_seqpan_query_workspace).  A *_bwd
 * call must follow the *_fwd call of the same block on the same workspace, batch and rng_state; it OVERWRITES `grads` (flat
 * parameter layout) with the gradients of the block's parameters (zero elsewhere).  Default kernel-fusion switches only.
 * ------------------------------------------------------------------------------------------ */
/* model.py:36-56: embeddings, query_conv1d / video_conv1d (the feature-load phase), q_/v_layer_norm, position embeddings
 * -> x0 [B*(T+L),128] */
int hual_video_proj_ln_fwd(const hual_cfg* cfg, const float* params, const float* word_table, const hual_batch* batch,
                           const hual_run_opts* opts, float* x0, void* workspace, uint64_t ws_bytes, void* stream);
/* modules.py:59-70 conv_block (shared weights, video and query rows in one pass): y = conv_block(x) */
int hual_conv_block_fwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, const float* x,
                        float* y, void* workspace, uint64_t ws_bytes, void* stream);
int hual_conv_block_bwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, const float* dy,
                        float* dx, float* grads, void* workspace, uint64_t ws_bytes, void* stream);
/* modules.py:73-89 + layers.py:59-111 dual_attn_block `layer` in both directions (v <- (v,q) and q <- (q,v), shared weights,
 * both from the OLD features, model.py:60-68): y = [dual_attn_block(v, q), dual_attn_block(q, v)] */
int hual_dual_attn_fwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, int layer,
                       const float* x, float* y, void* workspace, uint64_t ws_bytes, void* stream);
int hual_dual_attn_bwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, int layer,
                       const float* dy, float* dx, float* grads, void* workspace, uint64_t ws_bytes, void* stream);
/* layers.py:114-130 cq_attention in both directions (model.py:70-73): feats = [q2v_attn(v, q) on the video rows,
 * v2q_attn(q, v) on the query rows] */
int hual_cq_attn_fwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, const float* x,
                     float* feats, void* workspace, uint64_t ws_bytes, void* stream);
int hual_cq_attn_bwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, const float* dfeats,
                     float* dx, float* grads, void* workspace, uint64_t ws_bytes, void* stream);
/* modules.py:143-160 conditioned_predictor on `outputs` [B*T,128] -> raw start / end logits [B,T] and the span argmax of
 * layers.py:194-203; the backward takes the gradients of the two logit tensors */
int hual_predictor_fwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, const float* outputs,
                       float* start_logits, float* end_logits, int64_t* start_index, int64_t* end_index, void* workspace,
                       uint64_t ws_bytes, void* stream);
int hual_predictor_bwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, const float* d_start,
                       const float* d_end, float* d_outputs, float* grads, void* workspace, uint64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-kernel entry points (unit parity tests call these through ctypes).
 * ------------------------------------------------------------------------------------------ */

/* conv1d(kernel_size=1) == dense  (models/layers.py:20-29) on the 16-bit matrix cores with split operands (x = hi + lo, three
 * MFMA passes, fp32 accumulate; ~1e-6 relative to the fp32 product) - the kernel the model path uses for the dense layers outside
 * its fused kernels:
 *   trans_w = 0: Y[M,128] = act(A[M,K] . W[K,128] + bias), K % 8 == 0, act: 0 none, 1 relu; scratch >= ceil(K/128) * 65536 bytes
 *   trans_w = 1: Y[M,N]   = A[M,128] . W^T with W stored [N,128] (dX of a dense layer), N % 8 == 0;
 *                scratch >= ceil(N/128) * 65536 bytes
 * scratch (device) receives the pre-split weight image. */
int hual_linear_bf16x3(const float* A, int lda, const float* W, int trans_w, const float* bias, float* Y, int ldy, int M,
                       int K, int N, int act, void* scratch, uint64_t scratch_bytes, void* stream);

/* gradients of the dense above: dW[K,N] += A^T . dY ; db[N] += colsum(dY) (db may be NULL), through the persistent
 * weight-gradient launch of the training step: the 64-row tiles of the job dealt evenly to `workgroups` workgroups
 * (0 = one per CU), the job table living in `scratch` (device, >= 512 bytes).  N must be 128 (every dense layer of the
 * graph has 128 outputs).  Accumulates with float atomics: zero the destinations first.
 * Arithmetic (ABI 7): fp16-pair operands - A and dY each by a running power-of-two scale taken from the data (any magnitude: the
 * operands of a weight gradient are not bounded by construction - products of activations, unnormalised block outputs);
 * 2-5e-7 of the largest entry against a float64 product. */
int hual_linear_dw(const float* A, int lda, const float* dY, int ldy, float* dW, int ldw, float* db, int M, int K,
                   int N, int workgroups, void* scratch, uint64_t scratch_bytes, void* stream);

/* layer_norm (models/layers.py:7-17): y = (x - mean) * rsqrt(var + 1e-6) * gamma + beta over the 128 columns of each row;
 * mean / rstd (optional, [R]) are what the backward needs. */
int hual_layer_norm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int R,
                        void* stream);

/* multi-head attention core of dual_multihead_attention / top_self_attention (models/layers.py:80-96, modules.py:104-119):
 * 8 heads of size 16 kept merged in [rows,128]; scores / sqrt(16) + (1 - qmask x kmask) * (-1e30), softmax, P.V.
 * Q rows b*Tq + t, K/V rows b*Tk + t; masks are [B*Tq] / [B*Tk] floats (0/1).  Tq <= 256 and Tk <= 256
 * (longer queries: HUAL_ERR_INVALID - a launch's unit codes hold 16 query tiles per job).
 * Arithmetic (ABI 7): fp16-pair operands, Q / K / V scaled by 2^4 (|x| >= 4094 does not fit: Inf / NaN outputs, loudly); outputs within
 * 1e-6 of a float64 evaluation on O(1) inputs - the level of a float32 evaluation. */
int hual_attention_fwd(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* O, int ldo, int B, int Tq,
                       int Tk, const float* qmask, const float* kmask, void* stream);

/* The same with what a backward pass needs: stats [2][B*Tq*8] (row max of the scaled scores in the log2 domain, 1 / row sum,
 * per query and head) and - when drop_rate > 0 - the dropout of layers.py:86,91 / modules.py:114 on the probabilities with
 * the build's Philox stream (rng_state = device u32[3] {seed lo, seed hi, offset}, call site `drop_site`, RNG row =
 * query row * 8 + head; 16-bit decisions with the exact 1 / (1 - rate) scale, see DESIGN.md "Dropout") and its keep words: an opaque
 * buffer of B*Tq*8 rows of `ldm` bytes, ldm >= hual_attention_keep_row_bytes(Tk), 8-byte aligned (layout: csrc/attn.h). */
int hual_attention_fwd_save(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* O, int ldo, int B, int Tq,
                            int Tk, const float* qmask, const float* kmask, float* stats, uint8_t* keep_bytes, int ldm,
                            const uint32_t* rng_state, float drop_rate, int drop_site, void* stream);
int hual_attention_keep_row_bytes(int Tk);

/* gradient of the attention core (tf.gradients through layers.py:80-96): dQ, dK, dV [rows,128] (written, not accumulated)
 * from dO, the forward output O, `stats` and `keep_bytes` of hual_attention_fwd_save with the same arguments.  Tq, Tk <= 256. */
int hual_attention_bwd(const float* Q, int ldq, const float* K, const float* V, int ldkv, const float* O, int ldo,
                       const float* stats, const uint8_t* keep_bytes, int ldm, const float* dO, int lddo, float* dQ, int lddq,
                       float* dK, float* dV, int lddkv, int B, int Tq, int Tk, const float* qmask, const float* kmask,
                       const uint32_t* rng_state, float drop_rate, int drop_site, void* stream);

/* ans_predictor (models/layers.py:194-203): softmax of the masked logits, upper-triangular outer product, start = argmax
 * over rows of the row maxima, end = argmax over columns of the column maxima, first index on ties.  T <= 256. */
int hual_span_argmax(const float* start_logits, const float* end_logits, const float* vmask, int64_t* start_index,
                     int64_t* end_index, int B, int T, void* stream);

/* ------------------------------------------------------------------------------------------
 * Device-side batch assembly: TrainLoader.process_batch / TestLoader.process_batch
 * (/root/reference/utils/data_loader.py:30-98,145-164) from a training set that stays resident in HBM.
 * All arrays are device memory owned by the caller.  Videos: rows feat_off[v] .. feat_off[v+1] of feat_bank
 * [total_frames, vdim] (already down-sampled to <= max_vlen by visual_feature_sampling, data_utils.py:70-85).
 * Samples: video id, word ids word_bank[word_off[s] .. word_off[s+1]), chars of word w (global word index)
 * char_bank[char_off[w] .. char_off[w+1]), pseudo-label frame indices s_ind / e_ind (NULL for test sets).
 * ------------------------------------------------------------------------------------------ */
typedef struct hual_dataset {
  const float* feat_bank;
  const int64_t* feat_off;     /* [n_videos + 1] */
  int32_t vdim;
  const int32_t* sample_vid;   /* [n_samples] */
  const int32_t* word_off;     /* [n_samples + 1] */
  const int32_t* word_bank;
  const int32_t* char_off;     /* [n_words_total + 1] */
  const int32_t* char_bank;
  const int32_t* s_ind;        /* [n_samples] */
  const int32_t* e_ind;        /* [n_samples] */
} hual_dataset;

/* sel: i32 [B] sample ids of the batch.  T / L / C must be the maxima of the batch's video / word / char lengths
 * (the caller knows the lengths; C >= 4 for the model).  Writes the feeds of model.py:16-27: video f32 [B,T,vdim] zero
 * padded, video_seq_len i32 [B], word_ids i32 [B,L], char_ids i32 [B,L,C] and - unless y1 is NULL - the soft start/end
 * labels f32 [B,T], match_labels i32 [B,T], inner_labels f32 [B,T] exactly as data_loader.py:55-94 computes them. */
int hual_assemble_batch(const hual_dataset* ds, const int32_t* sel, int B, int T, int L, int C, float* video,
                        int32_t* video_seq_len, int32_t* word_ids, int32_t* char_ids, float* y1, float* y2,
                        int32_t* match_labels, float* inner_labels, void* stream);
/* The same launch with a CARRY: one of its workgroups also copies carry_n 8-byte words from carry_src to carry_dst (carry_n = 0: none).
 * For epoch loops that bank what the PREVIOUS step left in its fetch buffers (the predicted spans, runner_utils.py:150-156) without an
 * operation of their own between two replayed step graphs: every eager operation there costs ~13 us of idle device. */
int hual_assemble_batch_carry(const hual_dataset* ds, const int32_t* sel, int B, int T, int L, int C, float* video,
                              int32_t* video_seq_len, int32_t* word_ids, int32_t* char_ids, float* y1, float* y2,
                              int32_t* match_labels, float* inner_labels, const int64_t* carry_src, int64_t* carry_dst, int carry_n,
                              void* stream);

/* The same launch for an epoch loop with a device-side position: the batch's ids are ids[cursor[0] .. cursor[0] + B) (cursor: device
 * i64[2], advanced by hual_adamw_clip_step_loop at the end of the step). */
int hual_assemble_batch_cursor(const hual_dataset* ds, const int32_t* ids, const int64_t* cursor, int B, int T, int L, int C, float* video,
                               int32_t* video_seq_len, int32_t* word_ids, int32_t* char_ids, float* y1, float* y2,
                               int32_t* match_labels, float* inner_labels, void* stream);

/* ------------------------------------------------------------------------------------------
 * Active-learning label update (BASELINE.json configs[4]): what /root/reference/update_label.py does
 * per training sample between two training rounds, for the whole training set in two launches.
 *   hual_al_score  = the loop body of get_uncert_rank (update_label.py:125-169): sigmoid of the deterministic logits,
 *                    get_uncert_model (utils/utils_hual.py:144-161) of the two stochastic passes, get_distance_score
 *                    (:92-103) of the sample's active points, uncert_frame, uncert_video and the frame to annotate
 *                    (argmax of uncert_frame, update_label.py:194)
 *   hual_al_renew  = renew_label (update_label.py:85-123) for the selected samples, after the caller appended the
 *                    annotated frame to their active points (append_AP, utils_hual.py:133-139)
 * The ranking by uncert_video, the ground-truth lookup and the JSON/pickle files stay on the host (hual_amd/al.py).
 * A sample's logits occupy the first tlen[n] entries of its row (tlen = padded length of the batch the record came
 * from, `max_vlen = len(sprob)` in the reference); 2 <= tlen <= ld <= 1024, 1 <= vlen <= tlen.
 * ------------------------------------------------------------------------------------------ */
typedef struct hual_al_set {
  int32_t N, ld;
  const int32_t* vlen;     /* [N]   record['v_len'] */
  const int32_t* tlen;     /* [N]   len(prop_logits[0]) */
  const int32_t* ap_off;   /* [N+1] CSR offsets into ap_idx / ap_pos */
  const int32_t* ap_idx;   /* active points: frame index, in annotation order */
  const int8_t* ap_pos;    /* 1 = 'pos_idx' entry, 0 = 'neg_idx' entry */
} hual_al_set;

/* s0/e0: prop_logits, s1/e1: prop_logits1, s2/e2: prop_logits2, each f32 [N, ld].  Outputs: sprob, eprob f32 [N, ld];
 * uncert_frame f64 [N, ld]; uncert_video f32 [N]; observe_point i32 [N]. */
int hual_al_score(const hual_al_set* set, const float* s0, const float* e0, const float* s1, const float* e1,
                  const float* s2, const float* e2, float coff_uncert, float* sprob, float* eprob, double* uncert_frame,
                  float* uncert_video, int32_t* observe_point, void* stream);

/* sel: i32 [nsel] sample ids (NULL = all N); old_idx i32 [N,2]; coff6 (HOST pointer) = pos.{distance,model,old},
 * neg.{distance,model,old} of F_renew (update_label.py:11-37); new_idx i32 [N,2], written for the selected rows only. */
int hual_al_renew(const hual_al_set* set, const int32_t* sel, int nsel, const float* sprob, const float* eprob,
                  const int32_t* old_idx, const double* coff6, int32_t* new_idx, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement hook for bench.py's roofline leg (not part of the reference's surface): between begin and end every
 * kernel launch carries its own start / stop events (hipExtLaunchKernelGGL: the begin / end timestamps of that
 * kernel's dispatch, the quantity rocprofv3 --kernel-trace reports).  hual_prof_end() synchronises those events (the
 * only synchronising call in the library), aggregates per kernel symbol and returns the number of distinct kernels;
 * hual_prof_get(i, ...) reads entry i: kernel name, launches, microseconds, algorithmic FLOPs and bytes. */
int hual_prof_begin(void);
int hual_prof_end(void);
int hual_prof_get(int i, char* name, int name_cap, int64_t* launches, double* usec, double* flops, double* bytes);
/* the matrix pipe kernel `kernel` (a name hual_prof_get returned) runs its products on, and the MFMA passes one algorithmic
 * product costs there (split operands: 3) - what a TFLOP/s figure of that kernel has to be priced against */
#define HUAL_PIPE_NONE 0       /* no matrix instructions */
#define HUAL_PIPE_MATRIX32 1   /* v_mfma_f32_*_f32: 157.3 TFLOP/s dense peak */
#define HUAL_PIPE_MATRIX16 2   /* v_mfma_f32_*_{f16,bf16}: 2516.8 TFLOP/s dense peak */
int hual_prof_kernel_pipe(const char* kernel, int* pipe, int* passes);

#ifdef __cplusplus
}
#endif
#endif /* HUAL_SEQPAN_H */
