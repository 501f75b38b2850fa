// Context-query attention (video<->query fusion): /root/reference/models/layers.py:114-130 (cq_attention),
// /root/reference/models/ops.py:94-116 (trilinear_attention), layers.py:133-154 (weighted_pooling, cq_concat).
// Both directions (q2v: x1 = video, x2 = query; v2q: x1 = query, x2 = video) run in one launch, one workgroup
// per (clip, direction); the [N1 x N2] score matrix and its two softmaxes live in LDS.
#pragma once
#include "common.h"
#include "rowops.h"

namespace hual {

struct CqParams {          // per direction: 0 = q2v_attn, 1 = v2q_attn
  const float* w0[2];      // efficient_trilinear/linear_kernel4arg0 [128]
  const float* w1[2];      // linear_kernel4arg1 [128]
  const float* wm[2];      // linear_kernel4mul  [128]
};
struct CqGrads {
  float* w0[2]; float* w1[2]; float* wm[2];
};

// workspace tensors (all [R,128] in the unified row space unless noted)
struct CqBufs {
  const float* X;          // final dual-attention features (v rows, then q rows)
  float* D1W;              // dropout(x1) * wm   (x1 role of each row's own direction)
  float* D2;               // dropout(x2)
  float* S0; float* S1;    // [R] row dots  d1.w0 , d2.w1
  float* C2Q; float* Q2C;  // outputs
  float* SR; float* SC;    // [2][B][Tp*Lp] saved softmaxes (row / column)
  float* M2;               // [2][B][max(Tp,Lp)][128] scratch
  float* GS;               // [2][B][Tp*Lp] score scratch of the global-memory form of the forward (cq_fwd_global(): else unused / null)
};

int cq_padded(int n);      // rows padded to a multiple of 16
int launch_tri_prep(const CqBufs& b, const CqParams& p, const RowSpace& rs, const DropCfg& drop, hipStream_t s);
int launch_cq_fwd(const CqBufs& b, const CqParams& p, const RowSpace& rs, const DropCfg& drop, hipStream_t s);      // tri_prep included

struct CqBwdBufs {
  const float* dCat; int ldcat;   // [R,512]: gradient wrt [x1, c2q, x1*c2q, x1*q2c]
  float* dC2Q; float* dQ2C;       // [R,128] scratch
  float* dX;                      // [R,128] gradient wrt X (written by cq_bwd_pre, accumulated afterwards)
  float* dD1W; float* dD2;        // [R,128] gradient wrt D1W / D2
  float* dS0; float* dS1;         // [R]
  float* dM2;                     // scratch like M2
  float* GD;                      // [2][B][2][Tp*Lp] gradient scratch of the global-memory form of the backward (cq_bwd_global(): else null)
};
int launch_cq_bwd_pre(const CqBufs& b, const CqBwdBufs& g, const RowSpace& rs, hipStream_t s);
// dXa / dXb: dedicated [R,128] scratch (partial dX of the x1-role rows / x2-role rows of each direction)
int launch_cq_bwd_impl(const CqBufs& b, const CqBwdBufs& g, const RowSpace& rs, float* dXa, float* dXb, hipStream_t s);
// part: [tri_bwd_blocks_v + tri_bwd_blocks_q][3][128] per-workgroup sums of (d wm, d w0, d w1): the video-side workgroups first
// (-> wm[0], w0[0], w1[1]), then the query side (-> wm[1], w0[1], w1[0]); the caller folds them (colsum_kernel)
int tri_bwd_blocks_v(const RowSpace& rs);
int tri_bwd_blocks_q(const RowSpace& rs);
int launch_tri_bwd_impl(const CqBufs& b, const CqBwdBufs& g, const CqParams& p, float* part, const RowSpace& rs,
                        const DropCfg& drop, const float* dXa, const float* dXb, hipStream_t s);
// long clips (cqwide.hip): 128 < T <= 256, L <= 32
bool cq_wide_ok(const RowSpace& rs);
int launch_cq_fwd_wide(const CqBufs& b, const CqParams& p, const RowSpace& rs, const DropCfg& drop, hipStream_t s);
int launch_cq_bwd_wide(const CqBufs& b, const CqBwdBufs& g, const RowSpace& rs, float* dXa, float* dXb, hipStream_t s);
// shapes whose score matrices fit none of the LDS forms (long clips with queries of more than 32 words): the global-operand kernels keep
// the matrices in global memory too - slow, but every T, L <= 256 runs.  The orchestrator sizes GS / GD by these.
bool cq_fwd_global(int B, int T, int L);
bool cq_bwd_global(int B, int T, int L);
size_t cq_mat_elems_host(int T, int L);   // floats per saved softmax matrix (per clip, per direction)
size_t cq_m2_rows_host(int T, int L);     // rows of the per-clip M2 scratch

}  // namespace hual
