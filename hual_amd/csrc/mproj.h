// Multi-step dense kernel on the tile machinery of the fused row-local kernels (tilecore.h): a workgroup owns MT <= 64 rows and
// walks a list of up to MP_MAX weight steps.  A step multiplies one 128-deep operand block of its rows - read from HBM with an
// optional prologue (bfloat16 widening, elementwise factor, dropout) or reused from the previous step - by one [128,128] weight
// image streamed into LDS, into an accumulator tile that is started / closed per step; a closed tile gets bias / relu / a
// broadcast addend and is stored, or (last tile, LN mode) goes through layer norm + position embeddings in a row phase.
// Covers what used to be separate generic-dense launches and the K-split feature-load kernel:
//   feature load   video_conv1d / query_conv1d + v/q layer norm + pos emb (model.py:42-49,53-56): K / 128 steps, LN mode
//   cq_attention   dense over [x, c2q, x*c2q, x*q2c] (layers.py:127-130) and its dX (4 column blocks of one operand)
//   cq_concat      dense + pooled addend (layers.py:145-154) and its dX; heads' dX; query_conv1d dX; the char-CNN products
#pragma once
#include "common.h"
#include "rowops.h"

namespace hual {

#define MP_MAX 8
struct MProjStep {
  const void* A; int lda;          // operand rows: A + row * lda elements (float, or bfloat16 when a_bf16), kw valid columns
  int a_bf16;
  const float* A2; int lda2;       // optional elementwise factor (float rows)
  int kw;                          // 1..128 valid operand columns of this step (the rest of the 128-deep block is zero)
  int reuse;                       // 1: the operand planes of the previous step are used again (no refill)
  int drop_site; int col0;         // drop_site >= 0: dropout on the operand (16-bit decisions, csrc/tilecore.h); col0 = column of
                                   // A's element 0 inside the dropped tensor (counter + keep-byte position)
  uint8_t* keep_out; int ld_keep;  // optional: keep bytes (bit plane, one byte per 8 columns) at keep_out + row * ld_keep + col / 8
  const float* wimg; int wrows;    // weight image (pack_weights_kernel) and its valid K rows (rows beyond are clamped)
  int first, last;                 // first: the accumulator tile starts with this step; last: it is closed behind it
  int rep;                         // >= 1: the step stands for `rep` consecutive steps over one deep operand (feature load): repetition i
                                   // reads columns 128 i .. of A (kw = min(128, ktot - 128 i)), image rows 128 i .., col0 + 128 i;
                                   // `first` applies to repetition 0, `last` to the last one
  int ktot;                        // rep > 1: total valid columns of A
  // closing a tile:
  const float* bias;               // optional [128]
  int act;                         // 1: relu
  const float* add; int ldadd; int add_div;   // optional addend rows: add + (row / add_div) * ldadd
  float* out; int ldo; int ncol;   // destination rows out + row * ldo, ncol (multiple of 4, <= 128) valid columns; null in LN mode
};
struct MProjArgs {
  int nsteps; MProjStep s[MP_MAX];
  int R, MT;                       // rows, rows per workgroup (<= 64)
  uint32_t drop_row0;
  // LN mode (ln_g non-null): the LAST closed tile (+ bias) is x; x_out = x, y_out = LN(x; ln_g, ln_b) + pos[(row0 + row) % Tc]
  const float* ln_g; const float* ln_b; const float* pos; int row_in_clip0; int Tc;
  float* x_out; float* y_out; float* mean; float* rstd;
  // "quad" epilogue (quad_x non-null; exactly four steps, each closing a tile, no destinations): the four tiles d0 .. d3 of a row are
  // the gradient of [x, c2q, x * c2q, x * q2c] (cq_attention's concat, layers.py:127-130) and leave the kernel already split,
  //   dc2q = d1 + d2 * x,   dq2c = d3 * x,   dx = d0 + d2 * c2q + d3 * q2c      (rows of 128 floats, the problem's row index)
  const float* quad_x; const float* quad_c2q; const float* quad_q2c;
  float* quad_dc2q; float* quad_dq2c; float* quad_dx;
  // "pool" epilogue (pool_cat non-null; the char CNN, modules.py:26-38): the LAST closed tile (+ bias) holds, for every word, its
  // pool_C consecutive window rows x the 100 channels of the four filter banks; what leaves the kernel is relu + max over the valid
  // window starts of each channel, pool_cat[word * pool_ldcat + pool_col0 + ch], and the arg-max start (-1: none positive) for the
  // backward pass, pool_arg[word * 100 + ch].  Needs pool_C in {1, 2, 4, 8, 16} and MT % pool_C == 0; no tile is stored.
  float* pool_cat; int pool_ldcat; int pool_col0; int32_t* pool_arg; int pool_C;
};
int mproj_rows(int R0, int R1 = 0);      // rows per workgroup for one problem / a pair launched together
// one or two independent problems (blockIdx.y) in one launch
int launch_mproj(const MProjArgs* a, int nprob, const DropCfg& drop, hipStream_t s);

}  // namespace hual
