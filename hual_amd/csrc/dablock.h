// Fused row-local chains of dual_attn_block (/root/reference/models/modules.py:73-89, layers.py:59-111): everything
// between two attention kernels touches only its own rows, so a workgroup carries a tile of rows through it with the
// activations resident in LDS (operand planes) or registers (epilogue tiles), streaming the weight images by LDS-DMA.
//   ln_proj  : LN1 / LN_t of the layer input and the projections that read them (query, f_key, f_value | t_key, t_value);
//              also the predictor encoder's LN1 + dropout + query / key / value (modules.py:127-131, 92-102)
//   da_post  : s/x projections -> cross gating -> guided dense -> bilinear gate.value -> dense_1 + residual -> LN2 -> dense_2
//              + residual  (11 weight images, 12 saved tensors)
// Same arithmetic as the launch sequences they replace (ln_fwd_kernel, gemm_bf16_kernel / gemm_chain_kernel), bit for bit.
#pragma once
#include "common.h"
#include "rowops.h"

namespace hual {

#define HUAL_LNPROJ_MAX 5
struct LnProjArgs {
  const float* x;                                  // [R,128]
  const float* xa; int pre_site; float* x_out;     // optional: x := dropout(xa, pre_site) + x, written to x_out (modules.py:132)
  uint8_t* pre_bits; uint8_t* y1_bits;             // keep-bit planes [R][16] bytes (tilecore.h) of pre_site / drop_site1 for backward
  const float* x2;                                 // optional raw second operand [R,128] (exclusive with g2): src = 1 reads it
  const float* g1; const float* b1; float* y1;     // y1 = dropout(LN(x; g1, b1), drop_site1)   [R,128]
  int drop_site1;                                  // < 0: no dropout
  const float* g2; const float* b2; float* y2;     // optional second layer norm of the same rows (null: absent)
  float* mean; float* rstd;                        // [R]
  int nproj;
  const float* wimg[HUAL_LNPROJ_MAX];              // forward weight images
  const float* bias[HUAL_LNPROJ_MAX];              // [128]
  float* out[HUAL_LNPROJ_MAX]; int ldo[HUAL_LNPROJ_MAX];
  int src[HUAL_LNPROJ_MAX];                        // 0: reads y1, 1: reads y2 (or x2)
  int accum[HUAL_LNPROJ_MAX];                      // 1: no epilogue, the product is added to the next projection's (K-concatenation)
  int act[HUAL_LNPROJ_MAX];                        // 1: relu
  int out_site[HUAL_LNPROJ_MAX];                   // >= 0: dropout on the output ...
  uint8_t* out_bits[HUAL_LNPROJ_MAX];              // ... and where its keep bits go
  int add_x[HUAL_LNPROJ_MAX];                      // 1: + the layer-norm input rows (residual; needs g2 == x2 == null)
  int R; int MT;                                   // rows, rows per workgroup (1..64)
  int Nv;                                          // unified row space: its first Nv rows are video rows (0: one row space) - XCD order, common.h
  uint32_t drop_row0;
};
int ln_proj_rows(int R, int Nv = 0);       // (Nv: video rows of a unified row space - XCD order by clips, common.h)
int launch_ln_proj(const LnProjArgs& a, const DropCfg& drop, hipStream_t s);
int check_ln_proj_args(const LnProjArgs& a);
bool ln_proj_plain(const LnProjArgs& a);         // the query / key / value shape (lnproj_body.h PLAIN): a leaner instantiation      // (the argument checks of launch_ln_proj, for the launch that carries it as a tail: convblock.h)
int launch_ln_proj_pair(const LnProjArgs& a0, const LnProjArgs& a1, const DropCfg& drop, hipStream_t s);   // two problems, one launch
int ln_proj_pair_rows(int R);              // rows per workgroup for the pair launch: both problems within one workgroup per CU

struct DaPostArgs {
  const float* s_att; const float* x_att; const float* ln1; const float* x;    // [R,128] inputs (x: the layer input, residual)
  const float* rowmask;                                                         // [R]
  const float* w[11];      // images: s_dense, x_dense, s_gate, x_gate, guided, bl1.dense_1, bl1.dense_2, bl2.dense_1, bl2.dense_2, dense_1, dense_2
  const float* b[9];       // biases: s_dense, x_dense, s_gate, x_gate, guided, bilinear_1, bilinear_2, dense_1, dense_2
  const float* ln2_g; const float* ln2_b;
  float *sv, *xv, *sg, *xg, *o, *gd, *gate, *val, *mha, *res, *l2, *out;       // [R,128] saved tensors
  float *mean2, *rstd2;                                                         // [R]
  int site;                // dropout sites site+2 (dense_1 output), site+3 (LN2 output), site+4 (dense_2 output)
  uint8_t *bits2, *bits3, *bits4;   // their keep-bit planes [R][16] bytes (tilecore.h), read by the backward kernels
  int R; int MT;           // MT = 1..48
  int Nv;                  // unified row space: video rows (0: one row space) - XCD order, common.h
  uint32_t drop_row0;
};
int da_post_rows(int R, int Nv = 0);
// tail (optional): the next layer's ln_proj launch on the block output (x = a.out, same rows, plain shape) rides at the end of this one
int launch_da_post(const DaPostArgs& a, const DropCfg& drop, hipStream_t s, const LnProjArgs* tail = nullptr);

}  // namespace hual

namespace hual {

// dX products into a layer norm's output gradient(s), then the layer norm(s) backward, in one launch:
//   dy_o = sum_{k: dst[k] = o} f(A[k]) . W[k]^T (+ add_dy1 for o = 0);  dx = LNbwd(x; dropout'(dy_0), g1) (+ LNbwd(x; dy_1, g2)) + add1
// covers layer_norm_1 / layer_norm_t behind the five projections (6 products), layer_norm_2 behind dense_2 (1 product) and the
// predictor encoder's two layer norms (3 products / 1 product).  Same arithmetic as gemm_bf16_kernel + ln_bwd_kernel.
#define HUAL_LNBWD_MAX 6
struct LnProjBwdArgs {
  int nsteps;
  const float* A[HUAL_LNBWD_MAX]; int lda[HUAL_LNBWD_MAX];     // gradient operands: rows of 128 floats with row stride lda
  const uint8_t* a_bits[HUAL_LNBWD_MAX];                       // non-null: the operand is dropout'(A) with these keep bits (forward) ...
  float* a_save[HUAL_LNBWD_MAX];                               // ... and (optional) is stored here [R,128] for the weight-gradient job
  const float* wimg_t[HUAL_LNBWD_MAX];                         // images of the transposed weights
  int dst[HUAL_LNBWD_MAX];                                     // 0 / 1: which layer norm's output gradient the product belongs to
  const float* add_dy1;                                        // optional [R,128] added to dy_0
  const uint8_t* dy1_bits;                                     // non-null: the first layer norm's output went through dropout (keep bits)
  const float* x; const float* mean; const float* rstd;        // layer-norm input and statistics
  const float* g1; const float* g2;                            // g2 null: one layer norm
  const float* add1;                                           // optional gradient arriving through the residual path
  float* dx;
  float* dz; const uint8_t* dz_bits;                           // optional: dz = dropout'(dx) with the keep bits dz_bits (null: dz = dx)
  float* part;                                                 // [grid][4][128] partial sums: dgamma1, dbeta1, dgamma2, dbeta2
  int R; int MT; uint32_t drop_row0;
  int Nv;                  // unified row space: video rows (0: one row space) - XCD order, common.h
  // optional prologue (needs g2 == null, add1 == null): the gradient operand of product 0 is itself a layer norm's input gradient,
  //   dxp = LNbwd(pre_x; pre_dy, pre_g) (+ pre_add);   A[0] := dxp (a_bits[0] / a_save[0] apply as usual);   add1 := dxp
  // and that layer norm's dgamma / dbeta sums take the slots of the second layer norm in `part` (ln_bwd_kernel's arithmetic)
  const float* pre_x; const float* pre_mean; const float* pre_rstd; const float* pre_g; const float* pre_dy; const float* pre_add;
};
int ln_proj_bwd_rows(int R, int Nv = 0);
int ln_proj_bwd_blocks(int R, int Nv = 0);
int launch_ln_proj_bwd(const LnProjBwdArgs& a, const DropCfg& drop, hipStream_t s);

// backward of the gated middle of dual_multihead_attention (layers.py:93-110): from dZ1 = dropout'(d res) down to the
// gradients of the two attention outputs; every intermediate gradient that a weight-gradient job needs is written once.
struct DaMidBwdArgs {
  const float* dz1;                                            // [R,128]
  const float *gate, *val, *sg, *xg, *sv, *xv;                 // saved by the forward chain
  const float* w[10];     // transposed images: dense_1, bl1.dense_1, bl2.dense_1, bl1.dense_2, bl2.dense_2, guided, s_gate, x_gate, s_dense, x_dense
  float *d_sc, *d_val, *d_ln1a, *d_g, *dz_sg, *dz_xg, *d_sv, *d_xv, *d_satt, *d_xatt;
  int R; int MT; int Nv;   // (Nv: video rows of a unified row space, 0: one row space - XCD order, common.h)
};
int launch_da_mid_bwd(const DaMidBwdArgs& a, hipStream_t s);

}  // namespace hual
