#include <stdlib.h>
#include "prof.h"
#include <map>
#include <string>
#include <vector>
#include <string.h>
#include "../../include/hual_seqpan.h"

namespace hual {

namespace {
struct Rec { std::string name; hipEvent_t a, b; double flops, bytes; };
struct Agg { std::string name; int64_t launches = 0; double usec = 0.0, flops = 0.0, bytes = 0.0; };
struct State {
  bool on = false;
  std::vector<Rec> recs;
  std::vector<Agg> aggs;
};
thread_local State g_prof;

// "(gemm_lds_kernel<false, 3>)" -> "gemm_lds_kernel<false, 3>"
std::string clean(const char* s) {
  std::string n(s);
  while (!n.empty() && (n.front() == '(' || n.front() == ' ')) n.erase(n.begin());
  while (!n.empty() && (n.back() == ')' || n.back() == ' ')) n.pop_back();
  return n;
}
}  // namespace

bool prof_on() { return g_prof.on; }
bool lds_poison_on() {
  static const bool on = getenv("HUAL_DEBUG_LDS_POISON") != nullptr && atoi(getenv("HUAL_DEBUG_LDS_POISON")) != 0;
  return on;
}

void prof_events(const char* name, double flops, double bytes, hipEvent_t* start, hipEvent_t* stop) {
  *start = *stop = nullptr;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess) return;
  if (hipEventCreate(&b) != hipSuccess) { hipEventDestroy(a); return; }
  g_prof.recs.push_back(Rec{clean(name), a, b, flops, bytes});
  *start = a;
  *stop = b;
}

}  // namespace hual

extern "C" {

int hual_prof_begin(void) {
  hual::g_prof.on = true;
  hual::g_prof.recs.clear();
  hual::g_prof.aggs.clear();
  return 0;
}

int hual_prof_end(void) {
  using namespace hual;
  g_prof.on = false;
  std::map<std::string, size_t> idx;
  g_prof.aggs.clear();
  for (auto& r : g_prof.recs) {
    hipEventSynchronize(r.b);
    float ms = 0.f;
    const bool ok = hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess;
    auto it = idx.find(r.name);
    if (it == idx.end()) {
      it = idx.emplace(r.name, g_prof.aggs.size()).first;
      g_prof.aggs.push_back(Agg{r.name});
    }
    Agg& a = g_prof.aggs[it->second];
    if (ok) { a.launches += 1; a.usec += (double)ms * 1e3; a.flops += r.flops; a.bytes += r.bytes; }
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  g_prof.recs.clear();
  return (int)g_prof.aggs.size();
}

int hual_prof_get(int i, char* name, int name_cap, int64_t* launches, double* usec, double* flops, double* bytes) {
  using namespace hual;
  if (i < 0 || i >= (int)g_prof.aggs.size() || !name || name_cap < 1) return HUAL_ERR_INVALID;
  const Agg& a = g_prof.aggs[i];
  strncpy(name, a.name.c_str(), (size_t)name_cap - 1);
  name[name_cap - 1] = 0;
  if (launches) *launches = a.launches;
  if (usec) *usec = a.usec;
  if (flops) *flops = a.flops;
  if (bytes) *bytes = a.bytes;
  return 0;
}

// Which matrix pipe a kernel's products run on and how many MFMA passes one algorithmic product costs there (bench.py prices
// TFLOP/s against that pipe's peak, MI355X_MICROARCH.md).  Keyed by the kernel symbol's prefix; keep in step with the kernels.
int hual_prof_kernel_pipe(const char* kernel, int* pipe, int* passes) {
  if (!kernel || !pipe || !passes) return HUAL_ERR_INVALID;
  struct Row { const char* prefix; int pipe, passes; };
  static const Row rows[] = {
      // fp16 / bf16 hi + lo splits: hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_{f16,bf16} / 32x32x16_bf16
      {"conv_block_", HUAL_PIPE_MATRIX16, 3}, {"ln_proj", HUAL_PIPE_MATRIX16, 3}, {"da_post", HUAL_PIPE_MATRIX16, 3},
      {"da_mid_bwd", HUAL_PIPE_MATRIX16, 3},  {"mproj_", HUAL_PIPE_MATRIX16, 3},
      {"feature_", HUAL_PIPE_MATRIX16, 3},    {"dw_f16_", HUAL_PIPE_MATRIX16, 3},
      {"attn_", HUAL_PIPE_MATRIX16, 3},      {"cq_fwd_staged", HUAL_PIPE_MATRIX16, 3}, {"cq_bwd_staged", HUAL_PIPE_MATRIX16, 3},
      // exact fp32: v_mfma_f32_16x16x4_f32 (context-query attention of clips beyond the staged kernels' LDS budget)
      {"cq_", HUAL_PIPE_MATRIX32, 1},
  };
  for (const Row& r : rows)
    if (strncmp(kernel, r.prefix, strlen(r.prefix)) == 0) { *pipe = r.pipe; *passes = r.passes; return 0; }
  *pipe = HUAL_PIPE_NONE; *passes = 0;
  return 0;
}

}  // extern "C"
