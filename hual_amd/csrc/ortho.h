// The orthogonality term of the label embeddings, ||(E E^T) * (1 - I)||_F (model.py:88-91): it depends on the parameters only,
// so it is evaluated by one workgroup of the step's prologue launch (pack_weights_kernel, gemm.hip) - value into
// loss_acc[LA_ORTHO], gradient (times lambda) into scratch that match_bwd_kernel folds into d label_emb.
#pragma once
#include "common.h"

// call with >= 128 threads of one workgroup, all of them (barriers inside); sm: 16 + 4 floats of LDS
__device__ __forceinline__ void ortho_body(const float* E, float* loss_slot, float lambda, float* dE_store, float* sm) {
  float* M = sm;            // [16]
  float* red = sm + 16;     // [4]: per-wave partials (threads 0..127 = 2 waves take part in the sums)
  const int c = threadIdx.x;
  const bool on = c < HUAL_D;
  float e[4] = {0.f, 0.f, 0.f, 0.f};
  if (on)
    for (int i = 0; i < 4; ++i) e[i] = E[i * HUAL_D + c];
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 4; ++k) {
      const float d = wave_sum64(on ? e[i] * e[k] : 0.f);
      __syncthreads();
      if (on && (c & 63) == 0) red[c >> 6] = d;
      __syncthreads();
      if (c == 0) M[i * 4 + k] = (i == k) ? 0.f : red[0] + red[1];
    }
  __syncthreads();
  float ss = 0.f;
  for (int i = 0; i < 16; ++i) ss += M[i] * M[i];
  const float nrm = sqrtf(ss);
  if (c == 0) *loss_slot = nrm;
  if (on && dE_store) {
    // one row at a time: unrolled, two rows' sums were paired into v_pk_fma_f32 with the low lane reading the high register of the
    // (e0, e1) pair - the op_sel form hual_amd/build.py _check_isa refuses (unreliable when two queues share the GPU)
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += M[i * 4 + k] * e[k];
      dE_store[i * HUAL_D + c] = nrm > 0.f ? lambda * 2.0f * s / nrm : 0.f;
    }
  }
}
