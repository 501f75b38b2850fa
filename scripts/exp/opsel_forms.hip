// Which operand-select forms of the packed-fp32 instructions are affected by the round-6 finding (profiles/r6_packed_fp32_opsel.txt)?  Every form is
// issued from inline assembly by four waves of a workgroup while four more waves run v_mfma beside them (the configuration that failed 200 of 200 times
// in scripts/exp/opsel_repro.hip), and checked IN the kernel against the same products from scalar v_mul/v_fma/v_add instructions; mismatching lanes
// are counted by lane quarter.    hipcc --offload-arch=gfx950 -O3 -o build_exp/opsel_forms scripts/exp/opsel_forms.hip && build_exp/opsel_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// sel bits: lo lane of src0 / src1 / src2 reads the HIGH register (op_sel), hi lane of src0 / src1 / src2 reads the LOW register (!op_sel_hi)
#define PK2(NAME, INSTR, MODS)                                                                          \
  __device__ __forceinline__ f32x2 NAME(f32x2 a, f32x2 b) {                                              \
    f32x2 d;                                                                                             \
    asm volatile(INSTR " %0, %1, %2 " MODS : "=v"(d) : "v"(a), "v"(b));                                \
    return d;                                                                                            \
  }
#define PK3(NAME, MODS)                                                                                 \
  __device__ __forceinline__ f32x2 NAME(f32x2 a, f32x2 b, f32x2 c) {                                     \
    f32x2 d;                                                                                             \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=v"(d) : "v"(a), "v"(b), "v"(c));              \
    return d;                                                                                            \
  }
PK2(mul_plain, "v_pk_mul_f32", "")
PK2(mul_s0hi, "v_pk_mul_f32", "op_sel:[1,0]")            // lo lane: a.hi * b.lo
PK2(mul_s1hi, "v_pk_mul_f32", "op_sel:[0,1]")            // lo lane: a.lo * b.hi
PK2(mul_bothhi, "v_pk_mul_f32", "op_sel:[1,1]")          // lo lane: a.hi * b.hi
PK2(mul_h0lo, "v_pk_mul_f32", "op_sel_hi:[0,1]")         // hi lane: a.lo * b.hi
PK2(mul_h1lo, "v_pk_mul_f32", "op_sel_hi:[1,0]")         // hi lane: a.hi * b.lo
PK2(add_s1hi, "v_pk_add_f32", "op_sel:[0,1]")            // lo lane: a.lo + b.hi
PK3(fma_s1hi, "op_sel:[0,1,0]")                          // lo lane: a.lo * b.hi + c.lo
PK3(fma_s2hi, "op_sel:[0,0,1]")                          // lo lane: a.lo * b.lo + c.hi
PK3(fma_h1lo, "op_sel_hi:[1,0,1]")                       // hi lane: a.hi * b.lo + c.hi   (the form the library's kernels are full of)

#define NFORMS 10
__device__ __forceinline__ float smul(float x, float y) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }
__device__ __forceinline__ float sadd(float x, float y) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }
__device__ __forceinline__ float sfma(float x, float y, float z) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z)); return d; }

__global__ __launch_bounds__(512) void forms(const f32x2* in, unsigned* bad, float* sink, int n, int mfma_iters, int with_mfma) {
  if (threadIdx.x >= 256) {
    if (!with_mfma) return;
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < mfma_iters; ++it) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (c[0] == 12345.678f) sink[0] = c[0];
    return;
  }
  const int q = (threadIdx.x & 63) >> 4;      // lane quarter of the wave
  for (int i = blockIdx.x * 256 + threadIdx.x; i + 2 < n; i += gridDim.x * 256) {
    const f32x2 a = in[i], b = in[i + 1], c = in[i + 2];
    f32x2 got[NFORMS], want[NFORMS];
    got[0] = mul_plain(a, b);      want[0] = f32x2{smul(a.x, b.x), smul(a.y, b.y)};
    got[1] = mul_s0hi(a, b);       want[1] = f32x2{smul(a.y, b.x), smul(a.y, b.y)};
    got[2] = mul_s1hi(a, b);       want[2] = f32x2{smul(a.x, b.y), smul(a.y, b.y)};
    got[3] = mul_bothhi(a, b);     want[3] = f32x2{smul(a.y, b.y), smul(a.y, b.y)};
    got[4] = mul_h0lo(a, b);       want[4] = f32x2{smul(a.x, b.x), smul(a.x, b.y)};
    got[5] = mul_h1lo(a, b);       want[5] = f32x2{smul(a.x, b.x), smul(a.y, b.x)};
    got[6] = add_s1hi(a, b);       want[6] = f32x2{sadd(a.x, b.y), sadd(a.y, b.y)};
    got[7] = fma_s1hi(a, b, c);    want[7] = f32x2{sfma(a.x, b.y, c.x), sfma(a.y, b.y, c.y)};
    got[8] = fma_s2hi(a, b, c);    want[8] = f32x2{sfma(a.x, b.x, c.y), sfma(a.y, b.y, c.y)};
    got[9] = fma_h1lo(a, b, c);    want[9] = f32x2{sfma(a.x, b.x, c.x), sfma(a.y, b.x, c.y)};
#pragma unroll
    for (int f = 0; f < NFORMS; ++f) {
      if (__float_as_uint(got[f].x) != __float_as_uint(want[f].x)) atomicAdd(&bad[(f * 2 + 0) * 4 + q], 1u);
      if (__float_as_uint(got[f].y) != __float_as_uint(want[f].y)) atomicAdd(&bad[(f * 2 + 1) * 4 + q], 1u);
    }
  }
}

int main() {
  const int n = 1 << 20;
  f32x2* in; unsigned* bad; float* sink;
  CHECK(hipMalloc(&in, n * sizeof(f32x2))); CHECK(hipMalloc(&bad, NFORMS * 8 * 4)); CHECK(hipMalloc(&sink, 64));
  f32x2* h = (f32x2*)malloc(n * sizeof(f32x2));
  srand(3);
  for (int i = 0; i < n; ++i) h[i] = f32x2{(float)(rand() % 2001 - 1000) / 300.f, (float)(rand() % 2001 - 1000) / 700.f};
  CHECK(hipMemcpy(in, h, n * sizeof(f32x2), hipMemcpyHostToDevice));
  const char* names[NFORMS] = {"v_pk_mul_f32 (no modifier)", "v_pk_mul_f32 op_sel:[1,0]", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_mul_f32 op_sel:[1,1]", "v_pk_mul_f32 op_sel_hi:[0,1]",
                               "v_pk_mul_f32 op_sel_hi:[1,0]", "v_pk_add_f32 op_sel:[0,1]", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_fma_f32 op_sel_hi:[1,0,1]"};
  for (int with = 0; with < 2; ++with) {
    CHECK(hipMemset(bad, 0, NFORMS * 8 * 4));
    for (int rep = 0; rep < 50; ++rep) forms<<<512, 512>>>(in, bad, sink, n, 20000, with);
    CHECK(hipDeviceSynchronize());
    unsigned hb[NFORMS * 8];
    CHECK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
    printf("%s (50 launches x %d checks per form and lane half):\n", with ? "four v_mfma waves beside the four checking waves of every workgroup" : "no matrix instructions beside them", n - 2);
    for (int f = 0; f < NFORMS; ++f)
      printf("  %-34s wrong LOW results by lane quarter %8u %8u %8u %8u   wrong HIGH results %8u %8u %8u %8u\n", names[f], hb[f * 8], hb[f * 8 + 1], hb[f * 8 + 2], hb[f * 8 + 3],
             hb[f * 8 + 4], hb[f * 8 + 5], hb[f * 8 + 6], hb[f * 8 + 7]);
  }
  return 0;
}
