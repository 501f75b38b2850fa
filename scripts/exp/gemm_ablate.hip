// Ablation of the [M,128]x[128,128] dense kernel: which of {A loads, W staging, MFMA, stores} sets its duration?
// build: hipcc --offload-arch=gfx950 -O3 scripts/exp/gemm_ablate.hip -o scripts/exp/gemm_ablate ; run under rocprofv3 --kernel-trace
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GL_STAGE (64 * 128)
template <int MODE>   // bit0: W staging (LDS-DMA), bit1: A loads, bit2: MFMA, bit3: store, bit4: residual add load
__global__ __launch_bounds__(256) void k(const float* A, const float* W, const float* R, float* Y, int M) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int rowbase = blockIdx.x * 32 + (wave >> 1) * 16;
  const int n0 = (wave & 1) * 64;
  if (blockIdx.x * 32 >= M) return;
  const int arow = min(rowbase + j, M - 1);
  f32x4 acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 a[8];
  if (MODE & 1) {
    const int c4 = lane & 31, rr = lane >> 5;
    for (int s = 0; s < 2; ++s)
      for (int u = 0; u < 8; ++u) {
        const int pc = wave * 8 + u;
        const float* src = W + (size_t)(s * 64 + 2 * pc + rr) * 128 + 4 * c4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + s * GL_STAGE + pc * 256), 16, 0, 0);
      }
  }
  for (int c = 0; c < 8; ++c) a[c] = (MODE & 2) ? *reinterpret_cast<const float4*>(A + (size_t)arow * 128 + c * 16 + 4 * g) : make_float4(1.f, 2.f, 3.f, 4.f);
  float4 radd[4];
  for (int r = 0; r < 4; ++r) {
    const int row = min(rowbase + 4 * g + r, M - 1);
    radd[r] = (MODE & 16) ? *reinterpret_cast<const float4*>(R + (size_t)row * 128 + n0 + 4 * j) : make_float4(0, 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE & 4) {
    for (int c = 0; c < 8; ++c) {
      const float* wsb = lds + (c >> 2) * GL_STAGE + ((c & 3) * 16 + 4 * g) * 128 + n0 + 4 * j;
      float4 b[4];
      for (int q = 0; q < 4; ++q) b[q] = (MODE & 1) ? *reinterpret_cast<const float4*>(wsb + q * 128) : make_float4(1, 1, 1, 1);
      const float av[4] = {a[c].x, a[c].y, a[c].z, a[c].w};
      for (int q = 0; q < 4; ++q) {
        const float bv[4] = {b[q].x, b[q].y, b[q].z, b[q].w};
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[t], acc[t], 0, 0, 0);
      }
    }
  } else {
    for (int c = 0; c < 8; ++c) acc[c & 3][0] += a[c].x;
  }
  for (int r = 0; r < 4; ++r) {
    const int row = rowbase + 4 * g + r;
    if (row < M) {
      float4 v = make_float4(acc[0][r] + radd[r].x, acc[1][r] + radd[r].y, acc[2][r] + radd[r].z, acc[3][r] + radd[r].w);
      if ((MODE & 8) || v.x == 12345.678f) *reinterpret_cast<float4*>(Y + (size_t)row * 128 + n0 + 4 * j) = v;
    }
  }
}
__global__ void empty_k() {}
#define RUN(MODE) for (int it = 0; it < 30; ++it) hipLaunchKernelGGL(k<MODE>, dim3((M + 31) / 32), dim3(256), 65536, 0, A, W, R, Y, M);
int main() {
  for (int M : {8192, 9472}) {
    float *A, *W, *R, *Y;
    hipMalloc(&A, (size_t)M * 128 * 4); hipMalloc(&W, 128 * 128 * 4); hipMalloc(&R, (size_t)M * 128 * 4); hipMalloc(&Y, (size_t)M * 128 * 4);
    hipMemset(A, 0, (size_t)M * 128 * 4); hipMemset(W, 0, 128 * 128 * 4); hipMemset(R, 0, (size_t)M * 128 * 4);
    for (int it = 0; it < 2000; ++it) hipLaunchKernelGGL(k<31>, dim3((M + 31) / 32), dim3(256), 65536, 0, A, W, R, Y, M);   // warm clocks
    hipDeviceSynchronize();
    RUN(31) RUN(15) RUN(7) RUN(3) RUN(1) RUN(2) RUN(4) RUN(8) RUN(0) RUN(14) RUN(13)
    for (int it = 0; it < 30; ++it) hipLaunchKernelGGL(empty_k, dim3((M + 31) / 32), dim3(256), 0, 0);
    hipDeviceSynchronize();
    hipFree(A); hipFree(W); hipFree(R); hipFree(Y);
  }
  printf("done\n");
  return 0;
}
