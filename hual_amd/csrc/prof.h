// Optional per-kernel-family timing with HIP events, used only by bench.py's roofline leg.
// Off by default: hual_prof_begin() arms it for the calling thread, hual_prof_end() synchronises the recorded
// events (the ONLY place the library ever synchronises) and reports launches / microseconds / algorithmic FLOPs
// and bytes per family.
#pragma once
#include "common.h"

namespace hual {

enum ProfKind {
  PK_GEMM = 0, PK_GEMM_DUAL, PK_DW, PK_ATTN_FWD, PK_ATTN_BWD_DQ, PK_ATTN_BWD_DKV, PK_LN_FWD, PK_LN_BWD, PK_CONV_FWD,
  PK_CONV_BWD, PK_EW, PK_CQ, PK_EMBED, PK_HEADS, PK_OPTIM, PK_COUNT
};

bool prof_on();
// bracket one launch: call before and after the hipLaunchKernelGGL
void prof_start(int kind, hipStream_t s);
void prof_stop(int kind, hipStream_t s, double flops, double bytes);

struct ProfScope {
  int kind; hipStream_t s; double flops, bytes; bool on;
  ProfScope(int k, hipStream_t st, double f = 0.0, double b = 0.0) : kind(k), s(st), flops(f), bytes(b), on(prof_on()) {
    if (on) prof_start(kind, s);
  }
  ~ProfScope() { if (on) prof_stop(kind, s, flops, bytes); }
};

}  // namespace hual
