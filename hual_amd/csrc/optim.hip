// train_op of /root/reference/models/ops.py:119-204: clip_by_global_norm(1.0) + AdamWeightDecayOptimizer
// (beta .9/.999, eps 1e-6, NO bias correction, decoupled weight decay 0.01 on every variable whose name does not
// contain LayerNorm|layer_norm|bias) as two launches over the flat parameter / gradient buffers.
#include "optim.h"
#include "prof.h"

using namespace hual;

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* g, size_t n, float prescale, float* out) {
  __shared__ float sm[4];
  float s = 0.f;
  const size_t n4 = n >> 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = ld4(g + 4 * i);
    v = make_float4(v.x * prescale, v.y * prescale, v.z * prescale, v.w * prescale);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  s = wave_sum64(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];     // per-block partial: no zeroing, no atomics
}

__global__ __launch_bounds__(256) void zero_kernel(float* p, size_t n) {
  const size_t n4 = n >> 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    st4(p + 4 * i, f4zero());
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[4 * n4 + threadIdx.x] = 0.f;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, const float* decay,
                                                    size_t n, const float* lr_dev, float clip_norm, float prescale,
                                                    const float* sqnorm, float b1, float b2, float eps, uint32_t* rng_state,
                                                    int64_t* cursor, const int64_t* spans, int64_t* bank, int span_words, int sel_inc,
                                                    int bank_inc) {
  const float lr = lr_dev[0];
  if (rng_state && blockIdx.x == 0 && threadIdx.x == 0) rng_state[2] += 1u;     // nothing in this launch reads it
  if (cursor && blockIdx.x == gridDim.x - 1) {      // the epoch loop's position: bank this step's spans, move on (one block: ordered by its barrier)
    const int64_t bp = cursor[1];
    for (int i = threadIdx.x; i < span_words; i += blockDim.x) bank[bp + i] = spans[i];
    __syncthreads();
    if (threadIdx.x == 0) { cursor[0] += sel_inc; cursor[1] = bp + bank_inc; }
  }
  float part = 0.f;
  for (int i = threadIdx.x & 63; i < HUAL_SQNORM_SLOTS; i += 64) part += sqnorm[i];     // same order in every wave
  const float gn = sqrtf(wave_sum64(part));
  const float sc = prescale * (clip_norm / fmaxf(gn, clip_norm));     // tf.clip_by_global_norm
  const size_t n4 = n >> 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 gg = ld4(g + 4 * i), mm = ld4(m + 4 * i), vv = ld4(v + 4 * i), pp = ld4(p + 4 * i), dd = ld4(decay + 4 * i);
    float* gp = &gg.x; float* mp = &mm.x; float* vp = &vv.x; float* ppp = &pp.x; const float* dp = &dd.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = gp[k] * sc;
      const float nm = b1 * mp[k] + (1.0f - b1) * gr;
      const float nv = b2 * vp[k] + (1.0f - b2) * gr * gr;
      float upd = nm / (sqrtf(nv) + eps);          // ops.py:166-168
      upd += dp[k] * ppp[k];                       // ops.py:169-170 (decay[k] = 0 for excluded variables)
      ppp[k] = ppp[k] - lr * upd;
      mp[k] = nm;
      vp[k] = nv;
    }
    st4(p + 4 * i, pp);
    st4(m + 4 * i, mm);
    st4(v + 4 * i, vv);
  }
}

namespace hual {

int launch_adamw(const AdamArgs& a, hipStream_t s) {
  HUAL_REQUIRE(a.p && a.g && a.m && a.v && a.decay && a.lr_dev && a.sqnorm, "adamw: null pointer");
  HUAL_REQUIRE((a.n % 4) == 0, "adamw: flat size must be a multiple of 4");
  HUAL_REQUIRE(!a.cursor || (a.span_words == 0 || (a.spans && a.bank)), "adamw: loop cursor needs the span source and the bank");
  HUAL_LAUNCH(0.0, 4.0 * a.n, sqnorm_kernel, dim3(HUAL_SQNORM_SLOTS), dim3(256), 0, s, (const float*)a.g, a.n, a.prescale, a.sqnorm);
  HUAL_LAUNCH(0.0, 32.0 * a.n, adamw_kernel, dim3(512), dim3(256), 0, s, a.p, (const float*)a.g, a.m, a.v, a.decay, a.n, a.lr_dev,
                     a.clip_norm, a.prescale, (const float*)a.sqnorm, 0.9f, 0.999f, 1e-6f, a.rng_state,
                     a.cursor, a.spans, a.bank, a.span_words, a.sel_inc, a.bank_inc);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_zero(float* p, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  size_t g = (n / 4 + 255) / 256;
  g = g < 2048 ? (g > 0 ? g : 1) : 2048;
  HUAL_LAUNCH(0.0, 4.0 * n, zero_kernel, dim3((unsigned)g), dim3(256), 0, s, p, n);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
