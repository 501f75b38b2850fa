// Argument structs of the text-encoder front end (embed.hip), separate from embed.h so that gemm.h can carry them into the
// step's prologue launch (pack_weights_kernel runs the embedding gather and packs the char-CNN filter bank).
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
  // scratch (embed_layout), see embed.hip: cemb [(M+4) x CP] dropped char embeddings (M = Nq*C slot rows);
  // ball [128] packed biases (the packed filter bank "Wall" [4CP x 128] exists only as pre-split images, gemm.h PackWall); yall [M x 128] conv outputs (forward) then
  // their gradient (backward); dxall [M x 4CP] window gradients; dfall [4CP x 128 + 128] packed filter / bias gradients
  float* cemb; float* ball; float* yall; float* dxall; float* dfall;
  const float* wall_img; const float* wall_img_t;    // pre-split images of Wall / of its transpose (pack_weights_kernel)
};
struct EmbedGrads {
  const float* dcat; int lddcat;
  float* dunk; float* dchar_table; float* dfilt[4]; float* dfbias[4];
};

}  // namespace hual
