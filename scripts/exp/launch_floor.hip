// Micro-benchmark: per-kernel cost of back-to-back dependent launches inside a hipGraph (the shape of the training step:
// 256 workgroups x 512 threads, one per CU), for an empty kernel, a kernel whose arguments arrive in a by-value struct, and one
// with plain pointer arguments (eligible for kernel-argument preloading with -mllvm -amdgpu-kernarg-preload-count=16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Big { const float* p[20]; float* q[20]; int n; int pad; };
__global__ __launch_bounds__(512) void k_empty() {}
__global__ __launch_bounds__(512) void k_struct(Big a) {
  extern __shared__ char lds[];
  const int i = blockIdx.x * 512 + threadIdx.x;
  a.q[0][i] = a.p[0][i] + 1.0f;
}
__global__ __launch_bounds__(512) void k_struct_late(Big a) {      // second argument line touched after the first round trip
  extern __shared__ char lds[];
  const int i = blockIdx.x * 512 + threadIdx.x;
  float v = a.p[0][i];
  if (v > -1.0f) v += a.p[19][i];
  a.q[19][i] = v;
}
__global__ __launch_bounds__(512) void k_plain(const float* p, float* q) {
  extern __shared__ char lds[];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // <= 256 * 512 elements in every configuration below
  q[i] = p[i] + 1.0f;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class F>
static int run(const char* name, F launch, hipStream_t s, int n) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < n; ++i) launch();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int w = 0; w < 10; ++w) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-44s %7.2f us per kernel\n", name, ms * 1000.0f / (10.0f * n));
  return 0;
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const size_t N = 256 * 512;
  float *p, *q; CK(hipMalloc(&p, N * 4 * 2)); CK(hipMalloc(&q, N * 4 * 2)); CK(hipMemset(p, 0, N * 8));
  Big a{}; for (int i = 0; i < 20; ++i) { a.p[i] = p; a.q[i] = q; }
  CK(hipFuncSetAttribute((const void*)k_struct, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_struct_late, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_plain, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  const int n = 200;
  if (run("empty <<<256,512>>>", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, s); }, s, n)) return 1;
  if (run("empty <<<1,64>>>", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, s, n)) return 1;
  if (run("struct args, one load+store", [&] { hipLaunchKernelGGL(k_struct, dim3(256), dim3(512), 0, s, a); }, s, n)) return 1;
  if (run("struct args, 150 KB LDS", [&] { hipLaunchKernelGGL(k_struct, dim3(256), dim3(512), 150 * 1024, s, a); }, s, n)) return 1;
  if (run("struct args, late second line, 150 KB LDS", [&] { hipLaunchKernelGGL(k_struct_late, dim3(256), dim3(512), 150 * 1024, s, a); }, s, n)) return 1;
  if (run("plain pointer args, 150 KB LDS", [&] { hipLaunchKernelGGL(k_plain, dim3(256), dim3(512), 150 * 1024, s, (const float*)p, q); }, s, n)) return 1;
  if (run("plain pointer args, 2048 x 64 threads", [&] { hipLaunchKernelGGL(k_plain, dim3(2048), dim3(64), 0, s, (const float*)p, q); }, s, n)) return 1;
  return 0;
}
