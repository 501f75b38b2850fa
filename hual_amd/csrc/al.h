// Active-learning label update kernels (see al.hip).
#pragma once
#include "common.h"

#define HUAL_AL_MAX_T 1024

namespace hual {

struct AlScoreArgs {
  const float *s0, *e0, *s1, *e1, *s2, *e2;   // [N, ld] start / end logits: deterministic pass, two stochastic passes
  int ld, N;
  const int32_t* vlen;                        // [N] valid frames
  const int32_t* tlen;                        // [N] length of the logits record (padded length of its batch)
  const int32_t* ap_off;                      // [N+1] CSR offsets of the active points
  const int32_t* ap_idx;                      // frame index of each active point
  const int8_t* ap_pos;                       // 1 = inside the ground-truth span, 0 = outside
  float coff_uncert;
  float* sprob;                               // [N, ld]
  float* eprob;                               // [N, ld]
  double* uncert_frame;                       // [N, ld]
  float* uncert_video;                        // [N]
  int32_t* observe;                           // [N] argmax of uncert_frame
};

struct AlRenewArgs {
  const int32_t* sel;                         // [nsel] sample ids to update (NULL: all)
  const float* sprob;
  const float* eprob;
  int ld;
  const int32_t* vlen;
  const int32_t* tlen;
  const int32_t* ap_off;
  const int32_t* ap_idx;
  const int8_t* ap_pos;
  const int32_t* old_idx;                     // [N, 2]
  double coff[6];                             // pos.distance, pos.model, pos.old, neg.distance, neg.model, neg.old
  int32_t* new_idx;                           // [N, 2] (rows of unselected samples are left untouched)
};

int launch_al_score(const AlScoreArgs& a, hipStream_t s);
int launch_al_renew(const AlRenewArgs& a, int nsel, hipStream_t s);

}  // namespace hual
