// Optional per-kernel timing, used only by bench.py's roofline leg.
// Off by default: hual_prof_begin() arms it for the calling thread; while armed every launch goes through
// hipExtLaunchKernelGGL with its own start / stop events, which take the begin / end timestamps of that kernel's own
// dispatch (the numbers rocprofv3 --kernel-trace reports), not the time between two markers on the stream.
// hual_prof_end() synchronises the recorded events (the ONLY place the library ever synchronises) and aggregates
// launches / microseconds / algorithmic FLOPs and bytes per kernel symbol.
#pragma once
#include <hip/hip_ext.h>
#include "common.h"

namespace hual {

bool prof_on();
// a fresh event pair for one launch of kernel `name` (the stringified kernel expression)
void prof_events(const char* name, double flops, double bytes, hipEvent_t* start, hipEvent_t* stop);

// debug aid (HUAL_DEBUG_LDS_POISON=1 in the environment, read once): before EVERY launch a filler launch writes NaN patterns over the whole
// LDS of every CU, so a kernel that reads shared memory it has not written itself computes NaNs instead of living off what the previous
// launch happened to leave there - the shared-memory counterpart of the 0xFF workspace poison of the parity tests.  Off: one predictable branch.
bool lds_poison_on();
void lds_poison(hipStream_t stream);

}  // namespace hual

// Launch KERN (parenthesise template instantiations that contain commas).  FLOPS / BYTES = algorithmic work of this
// launch for the roofline report (0.0 when not meaningful).
#define HUAL_LAUNCH(FLOPS, BYTES, KERN, GRID, BLOCK, LDS, STREAM, ...)                              \
  do {                                                                                              \
    if (hual::lds_poison_on()) hual::lds_poison(STREAM);                                            \
    if (hual::prof_on()) {                                                                          \
      hipEvent_t hual_e0_ = nullptr, hual_e1_ = nullptr;                                            \
      hual::prof_events(#KERN, (FLOPS), (BYTES), &hual_e0_, &hual_e1_);                             \
      hipExtLaunchKernelGGL(KERN, GRID, BLOCK, LDS, STREAM, hual_e0_, hual_e1_, 0, __VA_ARGS__);    \
    } else {                                                                                        \
      hipLaunchKernelGGL(KERN, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                              \
    }                                                                                               \
  } while (0)
