// Micro-benchmark: does data written by one kernel stay in the writer's XCD-local L2 for the next kernel of a hipGraph?
// Kernel W: workgroup b writes its chunk of rows.  Kernel R: workgroup b reads the chunk of workgroup (b + shift) % G and reduces
// it.  Workgroups go to XCDs round-robin (b % 8), so shift 0 / 8 reads what the same XCD wrote, shift 1 what another XCD wrote.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int G = 256, ROWS = 37, D = 128;             // 256 x 37 rows x 512 B = 4.8 MB
__global__ __launch_bounds__(512) void k_write(float* buf, float v) {
  float4* p = reinterpret_cast<float4*>(buf + (size_t)blockIdx.x * ROWS * D);
  for (int i = threadIdx.x; i < ROWS * D / 4; i += 512) p[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(512) void k_read(const float* buf, float* out, int shift) {
  const int src = (blockIdx.x + shift) % G;
  const float4* p = reinterpret_cast<const float4*>(buf + (size_t)src * ROWS * D);
  float s = 0.f;
  for (int i = threadIdx.x; i < ROWS * D / 4; i += 512) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[blockIdx.x] = s;              // (never true: keeps the loads alive)
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float *buf, *out; CK(hipMalloc(&buf, (size_t)G * ROWS * D * 4)); CK(hipMalloc(&out, G * 4));
  const int n = 100;
  for (int mode = 0; mode < 2; ++mode)                  // 0: W,R pairs; 1: W only (subtract)
    for (int shift : {0, 8, 1, 3, 129}) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
      for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(k_write, dim3(G), dim3(512), 0, s, buf, (float)i);
        if (mode == 0) hipLaunchKernelGGL(k_read, dim3(G), dim3(512), 0, s, (const float*)buf, out, shift);
      }
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
      if (mode == 0) printf("write + read(shift %3d): %6.2f us per pair\n", shift, ms * 1000.f / (10.f * n));
      else { printf("write only            : %6.2f us per kernel\n", ms * 1000.f / (10.f * n)); break; }
    }
  return 0;
}
