// Fused clip_by_global_norm + AdamWeightDecay over the flat parameter buffer (see optim.hip).
#pragma once
#include "common.h"

namespace hual {

struct AdamArgs {
  float* p; float* g; float* m; float* v;
  const float* decay;     // per-element weight-decay rate (0.01 or 0)
  size_t n;               // padded flat size (multiple of 4)
  const float* lr_dev;    // device scalar: learning rate of this step (fed per step, main.py:61)
  float clip_norm;
  float prescale;         // multiplies the gradient first (1/world after a sum all-reduce)
  float* sqnorm;          // device scalar scratch: squared global norm of prescale*g
};
int launch_adamw(const AdamArgs& a, hipStream_t s);

}  // namespace hual
