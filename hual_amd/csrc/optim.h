// Fused clip_by_global_norm + AdamWeightDecay over the flat parameter buffer (see optim.hip).
#pragma once
#include "common.h"

#define HUAL_SQNORM_SLOTS 256

namespace hual {

struct AdamArgs {
  float* p; float* g; float* m; float* v;
  const float* decay;     // per-element weight-decay rate (0.01 or 0)
  size_t n;               // padded flat size (multiple of 4)
  const float* lr_dev;    // device scalar: learning rate of this step (fed per step, main.py:61)
  float clip_norm;
  float prescale;         // multiplies the gradient first (1/world after a sum all-reduce)
  float* sqnorm;          // device scratch, HUAL_SQNORM_SLOTS floats: per-block partial sums of |prescale*g|^2
  uint32_t* rng_state;    // optional: Philox state {k0, k1, offset}; offset += 1 after the update (next step's dropout)
  // optional: the epoch loop's device-side position (hual_loop_step): block 0 copies span_words 8-byte words (the step's predicted
  // spans) to bank + cursor[1], then cursor[0] += sel_inc (ids consumed), cursor[1] += bank_inc
  int64_t* cursor; const int64_t* spans; int64_t* bank; int span_words, sel_inc, bank_inc;
};
int launch_adamw(const AdamArgs& a, hipStream_t s);
// p[0..n) = 0 with a kernel (no memset node inside captured graphs)
int launch_zero(float* p, size_t n, hipStream_t s);

}  // namespace hual
