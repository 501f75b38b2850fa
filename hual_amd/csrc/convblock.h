// Fused conv_block kernels: ALL FOUR layers of /root/reference/models/modules.py:59-70
//   x_{l+1} = dropout(relu(pointwise(depthwise7(layer_norm_l(x_l))) + bias_l)) + x_l        (layers.py:32-45)
// in ONE launch per direction.  A workgroup owns MT consecutive rows of the unified row space and carries them through
// the four layers with the activations resident in LDS; the three neighbour rows each depthwise convolution needs on
// either side are RECOMPUTED (halo of 3 rows per remaining layer), never exchanged between workgroups, so there is no
// inter-workgroup dependency.  Everything backward needs (c_l, relu output y_l, x_{l+1}, LN statistics) is written for
// the owned rows only.  Arithmetic is the unfused path's (ln_dwconv_fwd_kernel + gemm_bf16_kernel), operation for
// operation, so both paths produce the same bits.
#pragma once
#include "common.h"
#include "rowops.h"

namespace hual {

struct CbLayerFwd {
  const float* ln_g; const float* ln_b;     // [128]
  const float* dw;                           // depthwise filter [7,128]
  const float* wimg;                         // forward image of the pointwise weight [128,128] (pack_weights_kernel)
  const float* bias;                         // [128]
  float* c; float* xout;                     // [R,128] each: depthwise output, layer output
  float* y;                                  // optional [R,128]: the relu output (parity taps only; backward reads the bit planes)
  uint8_t* relu_bits; uint8_t* keep_bits;    // bit planes [R][16] bytes (tilecore.h): y > 0, dropout keep decisions (written when dropout is on)
  float* mean; float* rstd;                  // [R]
  int drop_site;
};
struct CbFwdArgs {
  const float* x0;                           // [R,128] block input ...
  const float* pos; float* x0_out;           // ... or (pos != null) x0 = x0 + pos[t] (modules.py:41-56), also written to x0_out
  CbLayerFwd l[4];
  int MT;                                    // rows owned by a workgroup (<= HUAL_CB_MAXMT)
  uint32_t drop_row0;
};
#define HUAL_CB_MAXMT 46
int conv_block_fused_rows(int R, int Nv = 0);            // the MT launch_conv_block_fwd picks for R rows
// tail (optional): the layer norm(s) + projections launch that would follow on the block output (dablock.h: x = l[3].xout, same rows,
// no xa / x2) rides at the end of this one, tile by tile, its input rows taken from LDS (conv_block_fwd_lnproj_kernel)
struct LnProjArgs;
int launch_conv_block_fwd(const CbFwdArgs& a, const RowSpace& rs, const DropCfg& drop, hipStream_t s, const LnProjArgs* tail = nullptr);

struct CbLayerBwd {
  const float* ln_g; const float* ln_b; const float* dw;
  const float* wimg_t;                       // image of the transposed pointwise weight (dX)
  const float* x;                            // layer input x_l [R,128]
  const float* mean; const float* rstd;      // [R]
  const uint8_t* relu_prev; const uint8_t* keep_prev;   // bit planes of layer l-1 (relu active set, dropout keep set; null for l = 0)
  float* dz;                                 // [R,128] dZ_l = dropout'(dx_{l+1}) * relu'(y_l): operand of this layer's dW job
  float* dz_prev;                            // [R,128] dZ_{l-1} (null for l = 0), written for the owned rows
  float* part;                               // [grid][9][128] per-workgroup partial sums: ddw[0..6], dgamma, dbeta
};
struct CbBwdArgs {
  const float* dx_in;                        // gradient wrt the block output x_4 [R,128]
  const uint8_t* relu_bits3; const uint8_t* keep_bits3;   // bit planes of layer 3
  float* dx_out;                             // gradient wrt x_0 [R,128]
  CbLayerBwd l[4];
  int MT;
  uint32_t drop_row0;
};
#define HUAL_CB_BWD_MAXMT 40
int conv_block_fused_rows_bwd(int R, int Nv = 0);
int conv_block_bwd_blocks(int R, int Nv = 0);
int launch_conv_block_bwd(const CbBwdArgs& a, const RowSpace& rs, const DropCfg& drop, hipStream_t s);

}  // namespace hual
