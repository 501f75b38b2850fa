// v_fma_mixlo_f16 / v_fma_mixhi_f16 as the fp16-pair split with a scale folded in: bit-compare with the conversion sequence of
// bf16x3.h f16_split_pair on random and edge values.      hipcc --offload-arch=gfx950 -O3 scripts/exp/mix_split_test.hip -o build_exp/mix_split_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include "../../hual_amd/csrc/bf16x3.h"

__global__ void k(const float* x, float s, uint32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t h, l, h2, l2;
  f16_split_pair(x[2 * i] * s, x[2 * i + 1] * s, h, l);
  f16_split_pair_s(x[2 * i], x[2 * i + 1], s, h2, l2);
  out[4 * i] = h; out[4 * i + 1] = l; out[4 * i + 2] = h2; out[4 * i + 3] = l2;
}
int main() {
  const int n = 1 << 20;
  std::vector<float> hx(2 * n);
  uint32_t st = 12345u;
  for (int i = 0; i < 2 * n; ++i) {
    st = st * 1664525u + 1013904223u;
    const float u = (float)(st >> 8) / 16777216.0f * 2.0f - 1.0f;
    const int e = (int)((st >> 3) % 40) - 30;      // magnitudes 2^-30 .. 2^9
    hx[i] = ldexpf(u, e);
  }
  hx[0] = 0.f; hx[1] = -0.f; hx[2] = 1.0f; hx[3] = 65504.0f / 16.0f; hx[4] = 6.1e-5f / 16.0f; hx[5] = 1e-8f; hx[6] = 4095.9f; hx[7] = -3.14159f;
  float* dx; uint32_t* dout;
  hipMalloc(&dx, sizeof(float) * 2 * n); hipMalloc(&dout, sizeof(uint32_t) * 4 * n);
  hipMemcpy(dx, hx.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice);
  for (float s : {1.0f, 16.0f, 1024.0f, 0.125f}) {
    k<<<n / 256, 256>>>(dx, s, dout, n);
    std::vector<uint32_t> ho(4 * n);
    hipMemcpy(ho.data(), dout, sizeof(uint32_t) * 4 * n, hipMemcpyDeviceToHost);
    long bad_h = 0, bad_l = 0;
    for (int i = 0; i < n; ++i) { bad_h += ho[4 * i] != ho[4 * i + 2]; bad_l += ho[4 * i + 1] != ho[4 * i + 3]; }
    printf("scale %g: %d pairs, hi words differing %ld, lo words differing %ld   (first: %08x %08x | %08x %08x)\n", s, n, bad_h, bad_l, ho[0], ho[1], ho[2], ho[3]);
  }
  return 0;
}
