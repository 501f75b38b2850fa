// Row-wise (per activation row of 128 floats) kernels: layer norm, depthwise conv, position embeddings and the
// small elementwise glue between the GEMMs.  Rows live in the "unified row space":
//   rows [0, Nv)       = video rows  b*T + t
//   rows [Nv, Nv + Nq) = query rows  Nv + b*L + l
// so that blocks whose weights are shared between the video and the query side
// (/root/reference/models/model.py:53-68: conv_block and dual_attn_block with reuse=True) run as ONE launch.
#pragma once
#include "common.h"
#include "embed_args.h"

namespace hual {

struct RowSpace {
  int B, T, L, Nv, Nq, R;
  const float* rowmask;   // [R] 1.0 valid / 0.0 padded (v: t < video_seq_len[b]; q: word_ids != 0)
};

// y1 = LN(x; g1,b1) (+pos[t]) ; y2 = LN(x; g2,b2)          models/layers.py:7-17
struct LnFwd {
  const float* x; int R;
  const float* g1; const float* b1; float* y1;
  const float* g2; const float* b2; float* y2;     // optional
  float* mean; float* rstd;                        // optional [R]
  const float* pos;                                // optional [max_vlen,128]: y1 += pos[t]   (modules.py:41-56)
  int row0;                                        // unified row index of x's first row (pos / clip lookup)
  // optional: x is not given ready but as nparts partial sums (feature_ksplit_kernel): x = sum_q part[q*part_stride + ...]
  // + part_bias; the sum is also written to x_out (the pre-LN tensor backward needs)
  const float* part; int nparts; size_t part_stride; const float* part_bias; float* x_out;
  // optional second parameter set for the rows >= split (video rows and query rows of the unified row space in one launch:
  // v_layer_norm / q_layer_norm and the two projection biases, model.py:43,49); split <= 0: one set for all rows
  int split; const float* g1_hi; const float* b1_hi; const float* part_bias_hi;
};
int launch_ln_fwd(const LnFwd& a, const RowSpace& rs, const DropCfg& drop, hipStream_t s);

// dx = LNbwd(x; dy1,g1) + LNbwd(x; dy2,g2) + add1 + add2 ; dgamma/dbeta accumulated with atomics
struct LnBwd {
  const float* x; const float* mean; const float* rstd; int R;
  const float* dy1; const float* g1; float* dg1; float* db1;
  const float* dy2; const float* g2; float* dg2; float* db2;
  const float* add1; const float* add2;
  float* dx;
  // optional [ln_bwd_blocks(R)][4][128] scratch: per-block column sums (dg1, db1, dg2, db2) are written there with plain
  // stores instead of atomics on dg/db, so the launch can use every CU; launch_colsum() folds them into the gradients
  float* part;
  // optional second output for the consumer of dx: dz = dropout'(dx) with the keep bits the forward left in the bit plane
  // dz_bits ([R][16] bytes, csrc/tilecore.h; null: dz = dx) - saves the elementwise launch in front of the next dX product
  float* dz; const uint8_t* dz_bits;
  // optional second scale vector for the rows >= split (the two input layer norms in one launch, as LnFwd::split): the first
  // ln_bwd_blocks(split) workgroups take the rows below split, the others the rows from split on; needs `part`
  int split; const float* g1_hi;
};
int ln_bwd_blocks(int R);
struct PosBwdJob;
// pos / npos / rs (optional): position-table jobs (below) that ride in the same launch
int launch_ln_bwd(const LnBwd& a, const DropCfg& drop, hipStream_t s, const PosBwdJob* pos = nullptr, int npos = 0, const RowSpace* rs = nullptr);

// dst[v][c] += sum_blk src[(blk*nvec + v)*128 + c] for the per-block partial sums left by the backward kernels.
// All jobs of a backward pass go into ONE launch at its end.
#define HUAL_COLSUM_MAX_VEC 9
#define HUAL_COLSUM_MAX_JOBS 40
struct ColsumJob {
  const float* src; int nblk; int nvec;
  float* dst[HUAL_COLSUM_MAX_VEC];     // null entries are skipped
  int last_ncols;                      // > 0: the LAST vector only has this many columns (a destination shorter than 128)
};
// unpack (optional): the char-CNN filter-gradient unpack (embed_gather.h embed_unpack_task) as further workgroups of the launch
// (+ optionally the last step of the text encoder's backward - embed_gather.h embed_finish_block: finish_blocks workgroups with
//  finish_lds bytes of dynamic LDS - which depends on nothing the weight-gradient launch produces and so needs no launch of its own)
struct EmbedUnpack { EmbedArgs a; EmbedGrads g; int CP; int ntasks; int finish_blocks; int finish_lds; int nrows; DropCfg drop; };
int launch_colsum(const ColsumJob* jobs, int n, hipStream_t s, const EmbedUnpack* unpack = nullptr);

// dpos[t] += sum_b dx[b,t] (over the video rows, the query rows or both: the shared table serves both sides, model.py:53,56)
// for up to HUAL_POS_MAX_JOBS tables in one launch; a job sums up to two gradient tensors (the predictor's two encoder
// passes share one table).  Jobs must have distinct dpos.
#define HUAL_POS_MAX_JOBS 2
struct PosBwdJob { const float* dx[2]; float* dpos; int do_v, do_q; };
struct PosBwdBatch { PosBwdJob j[HUAL_POS_MAX_JOBS]; };
int launch_pos_bwd(const PosBwdJob* jobs, int njobs, const RowSpace& rs, hipStream_t s);

// out = a + b on [R,128] tensors (per-block entry points: sum of two gradient tensors)
int launch_add_rows(const float* a, const float* b, float* out, int R, hipStream_t s);

}  // namespace hual
