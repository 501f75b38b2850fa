// Everything after the CQ fusion that is not a [rows,128]x[128,128] GEMM: pooling for cq_concat, the matching
// head + label embeddings, the start/end logit heads, the three losses and the span argmax.
#pragma once
#include "common.h"
#include "rowops.h"

namespace hual {

// loss accumulator slots (float[8], zeroed each step)
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
};
int match_fwd_blocks(int Nv);
int launch_match_fwd(const MatchArgs& a, const RowSpace& rs, hipStream_t s);
struct MatchBwd {
  const float* dOut;      // [Nv,128] gradient wrt outputs
  const float* dOut2;     // optional second part of it (added on the fly: the predictor's heads and encoders both read `outputs`)
  float* dFuse;           // [Nv,128] written
  float* dWm; float* dbm; float* dE;   // destinations (through `part` and the colsum job the caller queues)
  const float* dE_ortho;  // optional [4,128]: gradient of the orthogonality term left by launch_loss_tail, added to dE
  float* part;            // [match_bwd_blocks(Nv)][9][128] per-workgroup sums (dE rows, dWm flat, dbm): folded by launch_colsum
  float lambda;           // loss.match_lambda
};
int match_bwd_blocks(int Nv);
int launch_match_bwd(const MatchArgs& a, const MatchBwd& g, const RowSpace& rs, hipStream_t s);
// denominator of the masked mean: loss_acc[LA_DENOM] = override > 0 ? override : loss_acc[LA_MASK_SUM] + 1e-12
int launch_match_denominator(float* loss_acc, float override_denom, hipStream_t s);
// ortho term ||(E E^T) * (1-I)||_F (model.py:88-91): loss_acc[LA_ORTHO] = norm ; dE += lambda * d norm / dE
int launch_ortho(const float* E, float* dE, float* loss_acc, float lambda, hipStream_t s);
// forward: ortho term + match denominator + the reported loss terms (loss_out[4], may be null) in one launch
int launch_loss_tail(const float* E, float* loss_acc, float lambda, float override_denom, const float* denom_dev, float* loss_out,
                     const float* match_part, int match_nblk, float* dE_ortho, hipStream_t s);

// ---- start/end logit heads: logit = h . w + b  (predictor/{start,end}_dense, modules.py:155-156)
struct DotArgs {
  const float* h[2]; const float* w[2]; const float* b[2]; float* logit[2];   // two heads per launch
  int R;
};
int launch_rowdot_fwd(const DotArgs& a, hipStream_t s);
struct DotBwd {
  const float* dlogit[2]; float* dZ[2];   // dZ = dlogit * w * (h > 0)   (h = relu output of the hidden dense)
  float* dw[2]; float* db[2];
};
int launch_rowdot_bwd(const DotArgs& a, const DotBwd& g, hipStream_t s);

// ---- localizing loss (layers.py:177-191) + span argmax (layers.py:194-203); one block per clip
struct LocArgs {
  const float* s_logit; const float* e_logit;     // [B,T]
  const float* vmask;                             // [B,T] floats
  const float* y1; const float* y2;               // labels or null
  int64_t* start_index; int64_t* end_index;       // [B]
  float* ds; float* de;                           // [B,T] gradients (null: skip)
  float* loss_acc;
  float inv_batch;                                // 1/B (reduce_mean over the batch)
};
int launch_loc(const LocArgs& a, int B, int T, hipStream_t s);

// ---- alignment loss (layers.py:205-248)
struct AlignPool {
  const float* F2;        // v2q_feats (q rows of the unified space)
  const float* F1;        // q2v_feats (v rows)
  const float* inner;     // [B,T] inner_labels (float)
  float* tpre; float* vpre;     // [B,128] pre-normalisation
  float* that; float* vhat;     // [B,128] l2-normalised
};
// all-rows similarity part.  that/vhat: [Bg,128] (gathered over ranks); gradients for ALL rows are produced.
struct AlignSim {
  const float* that; const float* vhat; int Bg;
  float* dq; float* da;         // [Bg,Bg] scratch
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
};

}  // namespace hual
