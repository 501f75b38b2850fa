#include "prof.h"
#include <vector>

namespace hual {

namespace {
struct Rec { int kind; hipEvent_t a, b; double flops, bytes; };
struct State {
  bool on = false;
  std::vector<Rec> recs;
  hipEvent_t pending = nullptr;
};
thread_local State g_prof;
}  // namespace

bool prof_on() { return g_prof.on; }

void prof_start(int kind, hipStream_t s) {
  (void)kind;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  hipEventRecord(e, s);
  g_prof.pending = e;
}

void prof_stop(int kind, hipStream_t s, double flops, double bytes) {
  if (!g_prof.pending) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  hipEventRecord(e, s);
  g_prof.recs.push_back(Rec{kind, g_prof.pending, e, flops, bytes});
  g_prof.pending = nullptr;
}

}  // namespace hual

extern "C" {

int hual_prof_begin(void) {
  hual::g_prof.on = true;
  hual::g_prof.recs.clear();
  return 0;
}

// out arrays of length HUAL_PROF_KINDS: launches, microseconds, flops, bytes.  Synchronises the recorded events.
int hual_prof_end(int64_t* launches, double* usec, double* flops, double* bytes, int n) {
  using namespace hual;
  g_prof.on = false;
  for (int i = 0; i < n; ++i) { launches[i] = 0; usec[i] = 0.0; flops[i] = 0.0; bytes[i] = 0.0; }
  for (auto& r : g_prof.recs) {
    hipEventSynchronize(r.b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.a, r.b);
    if (r.kind < n) {
      launches[r.kind] += 1; usec[r.kind] += (double)ms * 1e3; flops[r.kind] += r.flops; bytes[r.kind] += r.bytes;
    }
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  g_prof.recs.clear();
  return 0;
}

const char* hual_prof_kind_name(int k) {
  static const char* names[] = {"gemm_kernel<false>", "gemm_kernel<true>", "dw_kernel", "attn_fwd_kernel",
                                "attn_bwd_kernel", "(unused)", "ln_fwd_kernel", "ln_bwd_kernel",
                                "ln_dwconv_fwd_kernel", "dwconv_ln_bwd_kernel", "ew_kernel", "cq_kernels",
                                "embed_kernels", "head_kernels", "optim_kernels"};
  return (k >= 0 && k < hual::PK_COUNT) ? names[k] : "";
}

}  // extern "C"
