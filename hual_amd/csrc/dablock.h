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
  int out_site[HUAL_LNPROJ_MAX];                   // >= 0: dropout on the output
  int add_x[HUAL_LNPROJ_MAX];                      // 1: + the layer-norm input rows (residual; needs g2 == x2 == null)
  int R; int MT;                                   // rows, rows per workgroup (16, 32, 48 or 64)
  uint32_t drop_row0;
};
int ln_proj_rows(int R);
int launch_ln_proj(const LnProjArgs& a, const DropCfg& drop, hipStream_t s);

struct DaPostArgs {
  const float* s_att; const float* x_att; const float* ln1; const float* x;    // [R,128] inputs (x: the layer input, residual)
  const float* rowmask;                                                         // [R]
  const float* w[11];      // images: s_dense, x_dense, s_gate, x_gate, guided, bl1.dense_1, bl1.dense_2, bl2.dense_1, bl2.dense_2, dense_1, dense_2
  const float* b[9];       // biases: s_dense, x_dense, s_gate, x_gate, guided, bilinear_1, bilinear_2, dense_1, dense_2
  const float* ln2_g; const float* ln2_b;
  float *sv, *xv, *sg, *xg, *o, *gd, *gate, *val, *mha, *res, *l2, *out;       // [R,128] saved tensors
  float *mean2, *rstd2;                                                         // [R]
  int site;                // dropout sites site+2 (dense_1 output), site+3 (LN2 output), site+4 (dense_2 output)
  int R; int MT;           // MT = 16, 32 or 48
  uint32_t drop_row0;
};
int da_post_rows(int R);
int launch_da_post(const DaPostArgs& a, const DropCfg& drop, hipStream_t s);

}  // namespace hual
