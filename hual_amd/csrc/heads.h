// Everything after the CQ fusion that is not a [rows,128]x[128,128] GEMM: pooling for cq_concat, the matching
// head + label embeddings, the start/end logit heads, the three losses and the span argmax.
#pragma once
#include "common.h"
#include "rowops.h"

namespace hual {

// loss accumulator slots (float[8], zeroed each step by the prologue launch)
enum { LA_MATCH_SUM = 0, LA_MASK_SUM = 1, LA_ORTHO = 2, LA_LOC = 3, LA_ALIGN = 4, LA_DENOM = 5 };

// ---- cq_concat: weighted pooling of v2q_feats + the pooled half of the 2D->D dense (layers.py:133-154)
struct PoolArgs {
  const float* F2;        // v2q_feats, unified rows (uses the q rows)
  const float* wp;        // cq_cat/weighted_pooling/weight [128]
  const float* Wbot;      // rows 128..255 of cq_cat/dense/kernel  [128,128]
  float* alpha;           // [B, L]
  float* pooled;          // [B, 128]
  float* PW;              // [B, 128]  pooled . Wbot
};
struct AlignPool;
struct AlignPoolBwd;
// weighted pooling (+ pooled . Wbot) and - ap non-null - the per-sample alignment pooling, one launch
int launch_pool_align_fwd(const PoolArgs& a, const AlignPool* ap, const RowSpace& rs, hipStream_t s);
struct PoolBwd {
  const float* dFuse;     // [Nv,128] gradient wrt the fuse GEMM output
  float* dPW;             // [B,128]  out: sum_t dFuse
  float* dF2;             // v2q_feats gradient (q rows), ACCUMULATED (+=)
  float* dwp;             // [128] accumulated (atomics)
};
// backward of both, one launch (alignment part first: it WRITES the query rows the pooling part accumulates into)
int launch_pool_align_bwd(const PoolArgs& a, const PoolBwd& g, const AlignPool& ap, const AlignPoolBwd& ab, const RowSpace& rs, hipStream_t s);

// ---- matching head, label embeddings, masked outputs (layers.py:157-174, model.py:82-97)
struct MatchArgs {
  const float* fuse;      // [Nv,128]
  const float* Wm;        // matching_loss/dense/kernel [128,4]
  const float* bm;        // [4]
  const float* E;         // label_emb [4,128]
  const int32_t* labels;  // [Nv] or null (inference)
  float* probs;           // [Nv,4]  == match_scores
  float* probs2;          // optional second copy (kept in the workspace for the backward pass)
  float* outputs;         // [Nv,128]
  float* loss_acc;        // accumulators (LA_*), may be null when labels is null
  float* part;            // [match_fwd_blocks(Nv)][2] per-block (cross-entropy sum, mask sum): summed by launch_loss_tail - float
                          // atomics queued on one address retire at ~30 ns each, 2 x 128 of them were half of this kernel
  const uint32_t* rng;    // gumbel branch (layers.py:163-166; loss.no_gumbel false): the Philox state {k0, k1, offset}, else null
  float inv_tau;          // 1 / loss.tau: logits = (logits + noise) * inv_tau; the backward scales d logits by it
};
int match_fwd_blocks(int Nv);
struct AlignSim;
int launch_match_fwd(const MatchArgs& a, const RowSpace& rs, const AlignSim* as, hipStream_t s);
struct LossTailArgs {
  float* loss_acc; const float* match_part; int match_nblk; const float* loc_part; int loc_nblk;
  float lambda, override_denom; const float* denom_dev; float* loss_out;
  const uint32_t* ovf; int novf;      // optional: the pack launch's overflow words (gemm.h PackExtra): any set -> the losses are NaN
  const float* align_rows; int nalign;      // optional: the alignment loss' row terms (AlignSim::row_loss), summed here in row order into
                                            // loss_acc[LA_ALIGN]; null: loss_acc[LA_ALIGN] was written by the caller (data parallel)
};
struct MatchBwd {
  const float* dOut;      // [Nv,128] gradient wrt outputs
  const float* dOut2;     // optional second part of it (added on the fly: the predictor's heads and encoders both read `outputs`)
  float* dFuse;           // [Nv,128] written
  float* dWm; float* dbm; float* dE;   // destinations (through `part` and the colsum job the caller queues)
  const float* dE_ortho;  // optional [4,128]: gradient of the orthogonality term left by the step's prologue (ortho.h), added to dE
  float* part;            // [match_bwd_blocks(Nv)][9][128] per-workgroup sums (dE rows, dWm flat, dbm): folded by launch_colsum
  float lambda;           // loss.match_lambda
  // deferred loss tail (hual_run_opts.deferred_loss_terms): every workgroup forms the matching-loss denominator itself from the forward's
  // partial sums (loss_acc[LA_DENOM] has not been written), workgroup 0 also closes the loss as loss_tail_kernel would have
  int do_tail; LossTailArgs tail;
};
int match_bwd_blocks(int Nv);
int launch_match_bwd(const MatchArgs& a, const MatchBwd& g, const RowSpace& rs, hipStream_t s);
// ---- the predictor's output end (heads.hip heads_kernel): start / end logits (modules.py:155-156), localizing loss and its
// gradient (layers.py:177-191), span argmax (layers.py:194-203) and the gradients of the two hidden layers' outputs
struct HeadsArgs {
  const float* h[2]; const float* w[2]; const float* b[2];   // hidden relu outputs [B*T,128], dense kernels [128], biases [1]; h null: logits are inputs
  float* logit[2];                                // [B,T] start / end logits (written, or read when h is null)
  const float* vmask;                             // [B,T] floats
  const float* y1; const float* y2;               // soft labels or null (inference)
  int64_t* start_index; int64_t* end_index;       // [B]
  float* ds; float* de;                           // optional [B,T]: d loss / d logits (grad_only: INPUTS)
  int grad_only;                                  // 1: only step 3 from the given d logits (per-block entry point of the predictor)
  float* dZ[2];                                   // optional [B*T,128]: dZ = dlogit * w * (h > 0), the operand of the hidden layers' backward
  float* part[2];                                 // with dZ: [B][2][128] per-clip sums (vector 0 = d w, vector 1[0] = d b), folded by colsum_kernel
  float* loc_part;                                // [B] per-clip loss terms (summed in a fixed order by the last workgroup)
  float inv_batch;                                // 1/B (reduce_mean over the batch)
  const uint32_t* ovf; int novf;                  // label-free calls (no loss launch to carry the flag): the pack launch's overflow words
                                                  // (gemm.h PackExtra); any set -> NaN logits, span indices -1
};
int launch_heads(const HeadsArgs& a, int B, int T, hipStream_t s);
// matching-loss denominator (layers.py:173) + the four reported loss terms (model.py:120) from the partial sums
int launch_loss_tail(const LossTailArgs& a, hipStream_t s);

// ---- alignment loss (layers.py:205-248)
struct AlignPool {
  const float* F2;        // v2q_feats (q rows of the unified space)
  const float* F1;        // q2v_feats (v rows)
  const float* inner;     // [B,T] inner_labels (float)
  float* tpre; float* vpre;     // [B,128] pre-normalisation
  float* that; float* vhat;     // [B,128] l2-normalised, rows of stride ld (256: side by side in ONE [B,256] buffer, the rows a
  int ld;                       // data-parallel rank hands to the all-gather as they are)
};
// all-rows similarity part.  that/vhat: [Bg,128] (gathered over ranks); gradients for ALL rows are produced.
struct AlignSim {
  const float* that; const float* vhat; int Bg;
  float* dq; float* da;         // [Bg,Bg] scratch
  float* row_loss;              // optional [Bg]: the rows' loss terms go here and the column launch WRITES their sum (fixed order) to
                                // loss_acc[LA_ALIGN]; null: every row adds its term to loss_acc[LA_ALIGN] (zeroed by the caller)
  float* dthat; float* dvhat;   // [nrows,128] written: gradient rows row0 .. row0 + nrows - 1
  float* loss_acc;
  float scale;                  // multiplies the gradients (world size in exact data-parallel mode)
  int ld;                       // row stride of that / vhat (128, or 256 when both sit side by side in one gathered buffer)
  int row0, nrows;              // window of gradient rows wanted (0, Bg: all)
};
int launch_align_sim(const AlignSim& a, hipStream_t s);
struct AlignPoolBwd {
  const float* dthat; const float* dvhat;   // [B,128] (local slice)
  float* dF2;   // v2q_feats gradient, q rows: WRITTEN (=)
  float* dF1;   // q2v_feats gradient, v rows: ACCUMULATED (+=)
  // optional: the column part of d vhat is still missing from dvhat (the similarity rows ran inside launch_match_fwd): it is
  // formed here from the [Bg,Bg] scratch matrices of that launch
  const float* col_dq; const float* col_da; int col_Bg;
};

}  // namespace hual
