// Word + char embedding front end of the text encoder (see embed.hip).
#pragma once
#include "common.h"
#include "gemm.h"

namespace hual {

struct EmbedArgs {
  const int32_t* word_ids;      // [Nq]
  const int32_t* char_ids;      // [Nq, C]
  const float* word_table;      // frozen [num_words-2, word_dim]
  const float* unk;             // [word_dim]
  const float* char_table;      // [num_chars-1, char_dim]
  const float* filt[4];         // filter_i [k_i, char_dim, ch_i]
  const float* fbias[4];        // bias_i [ch_i]
  float* cat; int ldcat;        // [Nq, word_dim + 100]
  int32_t* char_arg;            // [Nq, 100] arg-max window start per channel (-1: relu clipped)
  int word_dim, char_dim, C, num_chars;
  // scratch (embed_layout), see embed.hip: cemb [(M+4) x CP] dropped char embeddings (M = Nq*C slot rows);
  // wall [4CP x 128] / wallt [128 x 4CP] packed filter banks; ball [128]; yall [M x 128] conv outputs (forward) then
  // their gradient (backward); dxall [M x 4CP] window gradients; dfall [4CP x 128 + 128] packed filter / bias gradients
  float* cemb; float* wall; float* wallt; float* ball; float* yall; float* dxall; float* dfall;
};
struct EmbedGrads {
  const float* dcat; int lddcat;
  float* dunk; float* dchar_table; float* dfilt[4]; float* dfbias[4];
};
// offsets (floats) of the scratch pieces inside one buffer of `total` floats
struct EmbedLayout { int CP; size_t cemb, wall, wallt, ball, yall, dxall, dfall, total; };
EmbedLayout embed_layout(int nrows, int C, int char_dim);
int launch_embed_fwd(const EmbedArgs& a, int nrows, const DropCfg& drop, hipStream_t s);
// backward: launches everything except the filter / bias gradient product, which is returned as a job for the step's
// weight-gradient launch (embed_dw_job fills the same job without launching anything: workspace planning);
// launch_embed_unpack must run after that launch.
void embed_dw_job(const EmbedArgs& a, int nrows, DwJob* dwjob);
int launch_embed_bwd(const EmbedArgs& a, const EmbedGrads& g, int nrows, const DropCfg& drop, hipStream_t s, DwJob* dwjob);
int launch_embed_unpack(const EmbedArgs& a, const EmbedGrads& g, hipStream_t s);

}  // namespace hual
