// Error plumbing + small host helpers shared by every translation unit of libhual_seqpan.so.
#include "common.h"
#include <math.h>
#include <mutex>
#include <set>
#include <utility>

namespace hual {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
const char* last_error_cstr() { return g_last_error.c_str(); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember (device, kernel) pairs that
// have been raised already, under a mutex, so that a process driving several GPUs from several threads gets it set on
// every device exactly once (include/hual_seqpan.h: "re-entrant, one stream per device/rank is safe").
int ensure_dyn_lds(const void* fn, int bytes) {
  int dev = 0;
  HUAL_CHECK_HIP(hipGetDevice(&dev));
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  std::lock_guard<std::mutex> lock(mu);
  const std::pair<int, const void*> key(dev, fn);
  if (done.count(key)) return 0;
  HUAL_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert(key);
  return 0;
}

DropCfg make_dropcfg(const uint32_t* state, float rate) {
  DropCfg d;
  d.state = state;
  d.enabled = (rate > 0.0f && state != nullptr) ? 1 : 0;
  // identical double arithmetic to oracle/philox.py keep_threshold()/keep_scale()
  double keep = 1.0 - (double)rate;
  double t = floor(keep * 4294967296.0);
  if (t < 0.0) t = 0.0;
  if (t > 4294967295.0) t = 4294967295.0;
  d.thresh = (uint32_t)t;
  d.scale = 1.0f / (1.0f - rate);
  return d;
}

}  // namespace hual
