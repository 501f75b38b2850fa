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

}  // extern "C"
