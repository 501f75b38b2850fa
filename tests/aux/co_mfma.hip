// a co-runner for the shared-GPU soaks: back-to-back matrix instructions on a caller-chosen stream (the strongest trigger of the round-6 finding,
// profiles/r6_packed_fp32_opsel.txt).   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o build_exp/co_mfma.so tests/aux/co_mfma.hip
#include <hip/hip_runtime.h>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void co_mfma(float* sink, int iters) {
  h16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (c[0] == 12345.678f) sink[0] = c[0];
}
extern "C" int co_mfma_launch(void* stream, void* sink, int blocks, int iters) {
  co_mfma<<<blocks, 256, 0, (hipStream_t)stream>>>((float*)sink, iters);
  return (int)hipGetLastError();
}
