// Stand-alone probe for the round-6 finding (profiles/r6_packed_fp32_opsel.txt): do packed-fp32 instructions whose op_sel makes the LOW lane read
// the HIGH register of a source pair (v_pk_mul_f32 ... op_sel:[1,0] / [0,1]) lose their low result when another queue runs kernels beside them?
// The victim computes the matching head's logit pattern l_c = sum_j a_j * w_j[c] per lane (the compiler pairs it into exactly those instructions:
// check with  hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only), its twin is compiled without packed math; a second stream runs a co-runner.
//   hipcc --offload-arch=gfx950 -O3 -o build_exp/opsel_repro scripts/exp/opsel_repro.hip && build_exp/opsel_repro [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ float4 logits(const float4 a, const float4 w0, const float4 w1, const float4 w2, const float4 w3) {
  return make_float4(a.x * w0.x + a.y * w1.x + a.z * w2.x + a.w * w3.x, a.x * w0.y + a.y * w1.y + a.z * w2.y + a.w * w3.y,
                     a.x * w0.z + a.y * w1.z + a.z * w2.z + a.w * w3.z, a.x * w0.w + a.y * w1.w + a.z * w2.w + a.w * w3.w);
}
// rows x 32 lanes; every lane: its float4 of the row, its four weight rows (the matching head's layout), `iters` times on rotated inputs
__global__ __launch_bounds__(256) void victim(const float4* f, const float4* w, float4* out, int rows, int iters) {
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const float4 w0 = w[4 * l32], w1 = w[4 * l32 + 1], w2 = w[4 * l32 + 2], w3 = w[4 * l32 + 3];
  for (int row = blockIdx.x * 8 + grp; row < rows; row += gridDim.x * 8) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {        // (a fresh load per product, as in the kernel: a.y / a.w arrive in the HIGH registers of their pairs)
      const float4 a = f[(size_t)((row + 97 * it) % rows) * 32 + l32];
      const float4 l = logits(a, w0, w1, w2, w3);
      acc.x += l.x; acc.y += l.y; acc.z += l.z; acc.w += l.w;
    }
    out[(size_t)row * 32 + l32] = acc;
  }
}
// co-runners
__global__ __launch_bounds__(256) void co_trans(float* p, int n, int iters) {      // transcendental + DPP traffic
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float x = p[i % n];
  for (int it = 0; it < iters; ++it) { x = __expf(x * 0.001f) + __shfl_xor(x, 1) * 0.5f; x = x - floorf(x); }
  p[i % n] = x;
}
__global__ __launch_bounds__(256) void co_mem(const float4* a, float4* b, size_t n) {      // streaming copy
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// gfx950's LDS transpose read (ds_read_b64_tr_b16), back to back
__global__ __launch_bounds__(256) void co_ldstr(int* sink, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<int*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  int acc = 0;
  for (int it = 0; it < iters; ++it) {
    const int off = ((threadIdx.x * 8 + it * 512) & 32767) & ~7;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 0x7fffffff) sink[0] = acc;
}
// ordinary LDS reads (ds_read_b64), the same access pattern
__global__ __launch_bounds__(256) void co_lds(int* sink, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<int*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  int acc = 0;
  for (int it = 0; it < iters; ++it) {
    const int off = ((threadIdx.x * 8 + it * 512) & 32767) & ~7;
    const int2 v = *reinterpret_cast<const int2*>(lds + off);
    acc += v.x + v.y;
  }
  if (acc == 0x7fffffff) sink[0] = acc;
}
// matrix-core work (v_mfma_f32_16x16x32_f16), back to back
__global__ __launch_bounds__(256) void co_mfma(float* sink, int iters) {
  h16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (c[0] == 12345.678f) sink[0] = c[0];
}

// ONE kernel, one stream: waves 0-3 of every workgroup do the victim's arithmetic, waves 4-7 run matrix instructions beside them
__global__ __launch_bounds__(512) void mixed(const float4* f, const float4* w, float4* out, float* sink, int rows, int iters, int mfma_iters) {
  if (threadIdx.x >= 256) {
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < mfma_iters; ++it) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (c[0] == 12345.678f) sink[0] = c[0];
    return;
  }
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const float4 w0 = w[4 * l32], w1 = w[4 * l32 + 1], w2 = w[4 * l32 + 2], w3 = w[4 * l32 + 3];
  for (int row = blockIdx.x * 8 + grp; row < rows; row += gridDim.x * 8) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
      const float4 a = f[(size_t)((row + 97 * it) % rows) * 32 + l32];
      const float4 l = logits(a, w0, w1, w2, w3);
      acc.x += l.x; acc.y += l.y; acc.z += l.z; acc.w += l.w;
    }
    out[(size_t)row * 32 + l32] = acc;
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 2000;
  const int rows = 8192, iters = argc > 2 ? atoi(argv[2]) : 16;
  std::vector<float> hf((size_t)rows * 128), hw(128 * 4);
  srand(7);
  for (auto& x : hf) x = (float)(rand() % 2001 - 1000) / 500.f;
  for (auto& x : hw) x = (float)(rand() % 2001 - 1000) / 700.f;
  float4 *f, *w, *out, *ref, *big0, *big1;
  float* tr;
  const size_t nb = (size_t)rows * 32 * sizeof(float4);
  CHECK(hipMalloc(&f, nb)); CHECK(hipMalloc(&w, 128 * sizeof(float4))); CHECK(hipMalloc(&out, nb)); CHECK(hipMalloc(&ref, nb));
  CHECK(hipMalloc(&big0, 256 << 20)); CHECK(hipMalloc(&big1, 256 << 20)); CHECK(hipMalloc(&tr, 4 << 20));
  CHECK(hipMemset(big0, 0, 256 << 20)); CHECK(hipMemset(tr, 0, 4 << 20));
  CHECK(hipMemcpy(f, hf.data(), nb, hipMemcpyHostToDevice)); CHECK(hipMemcpy(w, hw.data(), 128 * sizeof(float4), hipMemcpyHostToDevice));
  hipStream_t s0, s1;
  CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
  // quiet reference (nothing else on the GPU)
  victim<<<512, 256, 0, s0>>>(f, w, ref, rows, iters);
  CHECK(hipStreamSynchronize(s0));
  std::vector<float> href((size_t)rows * 128), hout((size_t)rows * 128);
  CHECK(hipMemcpy(href.data(), ref, nb, hipMemcpyDeviceToHost));
  const char* names[8] = {"nothing", "the victim itself", "exp + shuffle loop", "streaming copy", "ds_read_b64_tr_b16 loop", "ds_read_b64 loop", "v_mfma 16x16x32 f16 loop",
                          "NOTHING - the victim's own workgroups hold 4 more waves running v_mfma (one kernel, one stream)"};
  for (int mode = 0; mode < 8; ++mode) {
    long bad_launches = 0, bad_elems = 0, upper_only = 0;
    for (int k = 0; k < launches; ++k) {
      if (mode == 1) victim<<<512, 256, 0, s1>>>(f, w, ref, rows, iters);
      if (mode == 2) co_trans<<<1024, 256, 0, s1>>>(tr, 1 << 20, 200);
      if (mode == 3) co_mem<<<1024, 256, 0, s1>>>(big0, big1, (size_t)(256 << 20) / 16);
      if (mode == 4) co_ldstr<<<1024, 256, 0, s1>>>(reinterpret_cast<int*>(tr), 4000);
      if (mode == 5) co_lds<<<1024, 256, 0, s1>>>(reinterpret_cast<int*>(tr), 4000);
      if (mode == 6) co_mfma<<<1024, 256, 0, s1>>>(tr, 4000);
      if (mode == 7) mixed<<<512, 512, 0, s0>>>(f, w, out, tr, rows, iters, 20000);
      else victim<<<512, 256, 0, s0>>>(f, w, out, rows, iters);
      CHECK(hipStreamSynchronize(s0));
      CHECK(hipMemcpy(hout.data(), out, nb, hipMemcpyDeviceToHost));
      if (memcmp(hout.data(), href.data(), nb) != 0) {
        ++bad_launches;
        long cls[4] = {0, 0, 0, 0}, lost_j1 = 0, nbad = 0;
        for (size_t i = 0; i < hout.size(); ++i)
          if (memcmp(&hout[i], &href[i], 4) != 0) {
            ++bad_elems; ++nbad;
            const int l32 = (int)((i / 4) % 32), row = (int)(i / 128), c = (int)(i % 4);
            const int lane = l32 + 32 * (row % 2);
            if (lane >= 48) ++upper_only;
            ++cls[c];
            if (iters == 1) {      // is the wrong value the right one without the product a.y * w1[c] (the op_sel:[0,1] instruction's low result)?
              const float* a = &hf[(size_t)row * 128 + 4 * l32];
              const float* wl = &hw[16 * l32];                    // w0 .. w3 of the lane, 4 classes each
              const float want = fmaf(wl[12 + c], a[3], fmaf(wl[8 + c], a[2], fmaf(wl[c], a[0], 0.0f)));
              if (fabsf(hout[i] - want) <= 1e-6f * fmaxf(1.f, fabsf(want))) ++lost_j1;
            }
          }
        if (bad_launches == 1)
          printf("   first bad launch: %ld wrong elements, by class (float4 component) %ld %ld %ld %ld%s\n", nbad, cls[0], cls[1], cls[2], cls[3],
                 iters == 1 ? (lost_j1 == nbad ? "; EVERY one equals the sum without the term a.y * w1[c]" : "; not all explained by a lost a.y * w1[c]") : "");
      }
      CHECK(hipStreamSynchronize(s1));
    }
    printf("second stream runs %-26s: %ld of %d victim launches differ from the quiet result (%ld elements, %ld of them in lanes 48-63)\n",
           names[mode], bad_launches, launches, bad_elems, upper_only);
    fflush(stdout);
  }
  return 0;
}
