// Word + char embedding front end of the text encoder (see embed.hip).
#pragma once
#include "common.h"

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
};
struct EmbedGrads {
  const float* dcat; int lddcat;
  float* dunk; float* dchar_table; float* dfilt[4]; float* dfbias[4];
  float* partial;               // scratch: embed_bwd_partial_floats() floats
};
size_t embed_bwd_partial_floats(int nrows, int word_dim, int char_dim, int num_chars);
int launch_embed_fwd(const EmbedArgs& a, int nrows, const DropCfg& drop, hipStream_t s);
int launch_embed_bwd(const EmbedArgs& a, const EmbedGrads& g, int nrows, const DropCfg& drop, hipStream_t s);

}  // namespace hual
