// Device-side batch assembly (see assemble.hip).
#pragma once
#include "common.h"

namespace hual {

struct AssembleArgs {
  // resident training set
  const float* feat_bank; const int64_t* feat_off; int vdim;
  const int32_t* sample_vid;
  const int32_t* word_off; const int32_t* word_bank;
  const int32_t* char_off; const int32_t* char_bank;
  const int32_t* s_ind; const int32_t* e_ind;
  // the batch
  const int32_t* sel;
  int B, T, L, C;
  float* video; int32_t* lens; int32_t* word_ids; int32_t* char_ids;
  float* y1; float* y2; int32_t* match; float* inner;
  const int64_t* carry_src; int64_t* carry_dst; int carry_n;      // optional: 8-byte words copied by one block of the launch (0: none)
  const int64_t* cursor;      // optional (device): the batch's ids are sel[cursor[0] .. cursor[0] + B) - the epoch loop's position, advanced
                              // on the device by the step's last launch (optim.h AdamArgs::cursor), so the launch is the same every step
};

int launch_assemble(const AssembleArgs& a, hipStream_t s);

}  // namespace hual
