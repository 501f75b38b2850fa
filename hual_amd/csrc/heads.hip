// Heads, losses and span argmax (see heads.h).  These are tiny, HBM/latency-bound row kernels.
#include "heads.h"
#include "philox.h"
#include "prof.h"

using namespace hual;

#if defined(HUAL_STAMPS) && HUAL_STAMPS == 8      // debug build: phase clock stamps of heads_kernel (scripts/exp/stamps_generic.py)
static __device__ unsigned long long g_heads_stamps[512 * 64];
extern "C" int hual_debug_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_heads_stamps), sizeof(unsigned long long) * (size_t)n);
}
#define HEADS_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_heads_stamps[blockIdx.x * 64 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define HEADS_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ float block_sum(float v, float* sm) {   // blockDim multiple of 64, <= 256
  v = wave_sum64(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sm[i];
  return s;   // sm must hold blockDim.x/64 floats
}
// double-precision block sum (the span selection's softmax denominator: the sum of <= 256 floats in double is exact
// to ~2^-53, so its float rounding does not depend on the order of the additions)
__device__ __forceinline__ double block_sum_d(double v, double* sm) {
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sm[i];
  return s;
}
__device__ __forceinline__ float block_max(float v, float* sm) {
  v = wave_max64(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float s = -INFINITY;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s = fmaxf(s, sm[i]);
  return s;
}

// ------------------------------------------------------------------------------------------------------
// weighted pooling + pooled half of cq_cat/dense.  One block of 256 threads per clip: the L row dots are spread over
// 8 groups of 32 lanes (float4 per lane, shuffle reduction) instead of L block-wide reductions in sequence.
// weighted_pooling + the pooled half of cq_concat for clip b (layers.py:133-154); written for 256 threads
__device__ __forceinline__ void pool_fwd_body(const PoolArgs& a, const RowSpace& rs, int b) {
  __shared__ float al[256];
  __shared__ float pooled[HUAL_D];
  __shared__ float part[HUAL_D];
  const int tid = threadIdx.x;
  const bool on = tid < 256;            // the work is laid out for 256 threads; the other waves only take part in the barriers
  const int l32 = tid & 31, grp = tid >> 5;
  const int L = rs.L;
  const float* F = a.F2 + (size_t)(rs.Nv + b * L) * HUAL_D;
  const float* m = rs.rowmask + rs.Nv + b * L;
  if (on) {
    const float4 w4 = ld4(a.wp + 4 * l32);
    for (int l = grp; l < L; l += 8) {
      const float4 f = ld4(F + (size_t)l * HUAL_D + 4 * l32);
      const float d = half_sum32(f.x * w4.x + f.y * w4.y + f.z * w4.z + f.w * w4.w);
      if (l32 == 0) al[l] = d * m[l] + HUAL_MASK_VALUE * (1.0f - m[l]);   // mask_logits, layers.py:139
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int l = 0; l < L; ++l) mx = fmaxf(mx, al[l]);
  float sum = 0.f;
  for (int l = 0; l < L; ++l) sum += __expf(al[l] - mx);
  const float inv = 1.0f / sum;
  const int c = tid & 127, half = (tid >> 7) & 1;
  if (on && half == 0) {
    float p = 0.f;
    for (int l = 0; l < L; ++l) {
      const float alpha = __expf(al[l] - mx) * inv;
      if (c == 0) a.alpha[b * L + l] = alpha;
      p = fmaf(alpha, F[(size_t)l * HUAL_D + c], p);
    }
    pooled[c] = p;
    a.pooled[b * HUAL_D + c] = p;
  }
  __syncthreads();
  float o = 0.f;
  if (on) {
#pragma unroll 8
    for (int k = half * 64; k < half * 64 + 64; ++k) o = fmaf(pooled[k], a.Wbot[(size_t)k * HUAL_D + c], o);
    if (half) part[c] = o;
  }
  __syncthreads();
  if (on && !half) a.PW[b * HUAL_D + c] = o + part[c];
}

// -log(-log(u + 1e-20) + 1e-20) with u = top 24 bits of a Philox word * 2^-24 (ops.py:6-9)
__device__ __forceinline__ float gumbel_noise(uint32_t w) {
  const float u = (float)(w >> 8) * 5.9604644775390625e-08f;
  return -logf(-logf(u + 1e-20f) + 1e-20f);
}
__device__ __forceinline__ void match_fwd_body(const MatchArgs& a, const RowSpace& rs, int bid, int nblk) {
  __shared__ float red[2][8];
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  float ce_sum = 0.f, m_sum = 0.f;
  // Wm rows col..col+3 (4 classes each)
  float4 w0 = ld4(a.Wm + (col + 0) * 4), w1 = ld4(a.Wm + (col + 1) * 4), w2 = ld4(a.Wm + (col + 2) * 4), w3 = ld4(a.Wm + (col + 3) * 4);
  const float4 bm = ld4(a.bm);
  float4 e0 = ld4(a.E + col), e1 = ld4(a.E + HUAL_D + col), e2 = ld4(a.E + 2 * HUAL_D + col), e3 = ld4(a.E + 3 * HUAL_D + col);
  uint3 gk = make_uint3(0u, 0u, 0u);
  if (a.rng) gk = make_uint3(a.rng[0], a.rng[1], a.rng[2]);
  for (int row = bid * 8 + grp; row < rs.Nv; row += nblk * 8) {
    const size_t off = (size_t)row * HUAL_D + col;
    const float4 f = ld4(a.fuse + off);
    float l0 = f.x * w0.x + f.y * w1.x + f.z * w2.x + f.w * w3.x;
    float l1 = f.x * w0.y + f.y * w1.y + f.z * w2.y + f.w * w3.y;
    float l2 = f.x * w0.z + f.y * w1.z + f.z * w2.z + f.w * w3.z;
    float l3 = f.x * w0.w + f.y * w1.w + f.z * w2.w + f.w * w3.w;
    l0 = half_sum32(l0) + bm.x; l1 = half_sum32(l1) + bm.y; l2 = half_sum32(l2) + bm.z; l3 = half_sum32(l3) + bm.w;
    if (a.rng) {      // gumbel noise (ops.py:6-9): u on the 2^-24 grid from the four words of one call, oracle/philox.py gumbel_uniform
      const uint4_ r = philox4x32(0u, (uint32_t)row, (uint32_t)HUAL_SITE_GUMBEL, gk.z, gk.x, gk.y);
      l0 = (l0 + gumbel_noise(r.x)) * a.inv_tau; l1 = (l1 + gumbel_noise(r.y)) * a.inv_tau;
      l2 = (l2 + gumbel_noise(r.z)) * a.inv_tau; l3 = (l3 + gumbel_noise(r.w)) * a.inv_tau;
    }
    const float mx = fmaxf(fmaxf(l0, l1), fmaxf(l2, l3));
    const float x0 = expf(l0 - mx), x1 = expf(l1 - mx), x2 = expf(l2 - mx), x3 = expf(l3 - mx);
    const float sum = x0 + x1 + x2 + x3;
    const float inv = 1.0f / sum;
    const float p0 = x0 * inv, p1 = x1 * inv, p2 = x2 * inv, p3 = x3 * inv;
    const float mk = rs.rowmask[row];
    if (l32 == 0) {
      *reinterpret_cast<float4*>(a.probs + (size_t)row * 4) = make_float4(p0, p1, p2, p3);
      if (a.probs2) *reinterpret_cast<float4*>(a.probs2 + (size_t)row * 4) = make_float4(p0, p1, p2, p3);
      if (a.labels) {
        const int lab = a.labels[row];
        const float ll = lab == 0 ? l0 : (lab == 1 ? l1 : (lab == 2 ? l2 : l3));
        ce_sum += (mx + logf(sum) - ll) * mk;
        m_sum += mk;
      }
    }
    float4 o = make_float4((f.x + p0 * e0.x + p1 * e1.x + p2 * e2.x + p3 * e3.x) * mk,
                           (f.y + p0 * e0.y + p1 * e1.y + p2 * e2.y + p3 * e3.y) * mk,
                           (f.z + p0 * e0.z + p1 * e1.z + p2 * e2.z + p3 * e3.z) * mk,
                           (f.w + p0 * e0.w + p1 * e1.w + p2 * e2.w + p3 * e3.w) * mk);
    st4(a.outputs + off, o);
  }
  if (a.labels && a.loss_acc) {
    if (l32 == 0) { red[0][grp] = ce_sum; red[1][grp] = m_sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float c = 0.f, m = 0.f;
      for (int i = 0; i < 8; ++i) { c += red[0][i]; m += red[1][i]; }
      if (a.part) { a.part[2 * bid] = c; a.part[2 * bid + 1] = m; }
      else { atomicAdd(a.loss_acc + LA_MATCH_SUM, c); atomicAdd(a.loss_acc + LA_MASK_SUM, m); }
    }
  }
}

__device__ __forceinline__ float loss_tail_body(const LossTailArgs& a, float* sm, bool write);      // (defined with loss_tail_kernel below)
__global__ __launch_bounds__(256) void match_bwd_kernel(MatchArgs a, MatchBwd g, RowSpace rs) {
  __shared__ float4 red[8][8][32];    // [vec][grp][lane]: 4 dE rows + 4 dWm columns(as float4 over cols)
  __shared__ float redb[8][4];
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  float4 w0 = ld4(a.Wm + (col + 0) * 4), w1 = ld4(a.Wm + (col + 1) * 4), w2 = ld4(a.Wm + (col + 2) * 4), w3 = ld4(a.Wm + (col + 3) * 4);
  float4 e0 = ld4(a.E + col), e1 = ld4(a.E + HUAL_D + col), e2 = ld4(a.E + 2 * HUAL_D + col), e3 = ld4(a.E + 3 * HUAL_D + col);
  // the first row of every group is requested BEFORE the loss is closed (its denominator is a workgroup reduction over the forward's
  // partial sums: a memory round trip and two barriers that the row loads do not depend on)
  const int row0 = min((int)blockIdx.x * 8 + grp, rs.Nv - 1);
  const size_t off0 = (size_t)row0 * HUAL_D + col;
  float pmk = rs.rowmask[row0];
  float4 pd = ld4(g.dOut + off0), pd2 = g.dOut2 ? ld4(g.dOut2 + off0) : f4zero();
  float4 pf = ld4(a.fuse + off0), pp = *reinterpret_cast<const float4*>(a.probs + (size_t)row0 * 4);
  float denom = 1.0f;
  if (g.do_tail) {      // (uniform) the forward left the loss open: the denominator from its partial sums; workgroup 0 closes the loss
    __shared__ float smt[12];
    denom = loss_tail_body(g.tail, smt, blockIdx.x == 0);
  } else if (a.labels) {
    denom = a.loss_acc[LA_DENOM];
  }
  const float ce_scale = a.labels ? g.lambda / denom : 0.f;
  float4 dE0 = f4zero(), dE1 = f4zero(), dE2 = f4zero(), dE3 = f4zero();
  float4 dW0 = f4zero(), dW1 = f4zero(), dW2 = f4zero(), dW3 = f4zero();   // dWk = column k of dWm over this lane's 4 rows
  float db0 = 0.f, db1 = 0.f, db2 = 0.f, db3 = 0.f;
  for (int row = blockIdx.x * 8 + grp; row < rs.Nv; row += gridDim.x * 8) {
    const size_t off = (size_t)row * HUAL_D + col;
    // this row's operands (requested one trip ahead); the next row's requests go out before this row's arithmetic
    const float mk = pmk;
    float4 d = pd;
    if (g.dOut2) d = make_float4(d.x + pd2.x, d.y + pd2.y, d.z + pd2.z, d.w + pd2.w);
    const float4 f = pf, p = pp;
    {
      const int rn = min(row + (int)gridDim.x * 8, rs.Nv - 1);
      const size_t offn = (size_t)rn * HUAL_D + col;
      pmk = rs.rowmask[rn];
      pd = ld4(g.dOut + offn);
      if (g.dOut2) pd2 = ld4(g.dOut2 + offn);
      pf = ld4(a.fuse + offn);
      pp = *reinterpret_cast<const float4*>(a.probs + (size_t)rn * 4);
    }
    d = make_float4(d.x * mk, d.y * mk, d.z * mk, d.w * mk);   // through the *v_mask of model.py:97
    float q0 = half_sum32(d.x * e0.x + d.y * e0.y + d.z * e0.z + d.w * e0.w);
    float q1 = half_sum32(d.x * e1.x + d.y * e1.y + d.z * e1.z + d.w * e1.w);
    float q2 = half_sum32(d.x * e2.x + d.y * e2.y + d.z * e2.z + d.w * e2.w);
    float q3 = half_sum32(d.x * e3.x + d.y * e3.y + d.z * e3.z + d.w * e3.w);
    const float dot = p.x * q0 + p.y * q1 + p.z * q2 + p.w * q3;
    float dl0 = p.x * (q0 - dot), dl1 = p.y * (q1 - dot), dl2 = p.z * (q2 - dot), dl3 = p.w * (q3 - dot);
    if (a.labels) {
      const int lab = a.labels[row];
      const float cs = ce_scale * mk;
      dl0 += cs * (p.x - (lab == 0 ? 1.f : 0.f));
      dl1 += cs * (p.y - (lab == 1 ? 1.f : 0.f));
      dl2 += cs * (p.z - (lab == 2 ? 1.f : 0.f));
      dl3 += cs * (p.w - (lab == 3 ? 1.f : 0.f));
    }
    if (a.rng) { dl0 *= a.inv_tau; dl1 *= a.inv_tau; dl2 *= a.inv_tau; dl3 *= a.inv_tau; }      // z = (logits + noise) / tau
    // dfuse = d + sum_c dl[c] * Wm[:,c]
    st4(g.dFuse + off, make_float4(d.x + dl0 * w0.x + dl1 * w0.y + dl2 * w0.z + dl3 * w0.w,
                                   d.y + dl0 * w1.x + dl1 * w1.y + dl2 * w1.z + dl3 * w1.w,
                                   d.z + dl0 * w2.x + dl1 * w2.y + dl2 * w2.z + dl3 * w2.w,
                                   d.w + dl0 * w3.x + dl1 * w3.y + dl2 * w3.z + dl3 * w3.w));
    dE0 = make_float4(dE0.x + p.x * d.x, dE0.y + p.x * d.y, dE0.z + p.x * d.z, dE0.w + p.x * d.w);
    dE1 = make_float4(dE1.x + p.y * d.x, dE1.y + p.y * d.y, dE1.z + p.y * d.z, dE1.w + p.y * d.w);
    dE2 = make_float4(dE2.x + p.z * d.x, dE2.y + p.z * d.y, dE2.z + p.z * d.z, dE2.w + p.z * d.w);
    dE3 = make_float4(dE3.x + p.w * d.x, dE3.y + p.w * d.y, dE3.z + p.w * d.z, dE3.w + p.w * d.w);
    dW0 = make_float4(dW0.x + f.x * dl0, dW0.y + f.y * dl0, dW0.z + f.z * dl0, dW0.w + f.w * dl0);
    dW1 = make_float4(dW1.x + f.x * dl1, dW1.y + f.y * dl1, dW1.z + f.z * dl1, dW1.w + f.w * dl1);
    dW2 = make_float4(dW2.x + f.x * dl2, dW2.y + f.y * dl2, dW2.z + f.z * dl2, dW2.w + f.w * dl2);
    dW3 = make_float4(dW3.x + f.x * dl3, dW3.y + f.y * dl3, dW3.z + f.z * dl3, dW3.w + f.w * dl3);
    if (l32 == 0) { db0 += dl0; db1 += dl1; db2 += dl2; db3 += dl3; }
  }
  red[0][grp][l32] = dE0; red[1][grp][l32] = dE1; red[2][grp][l32] = dE2; red[3][grp][l32] = dE3;
  red[4][grp][l32] = dW0; red[5][grp][l32] = dW1; red[6][grp][l32] = dW2; red[7][grp][l32] = dW3;
  if (l32 == 0) { redb[grp][0] = db0; redb[grp][1] = db1; redb[grp][2] = db2; redb[grp][3] = db3; }
  __syncthreads();
  // per-workgroup sums -> g.part[blk][9][128] (folded by launch_colsum): vectors 0-3 = rows of dE, 4-7 = dWm flat ([128,4]:
  // element c*4 + class), 8 = dbm in its first 4 entries.  Workgroup 0 adds the gradient of the orthogonality term
  // (model.py:88-91), computed by the forward's loss tail.
  float* part = g.part + (size_t)blockIdx.x * 9 * HUAL_D;
  for (int idx = threadIdx.x; idx < 8 * 128; idx += 256) {
    const int vec = idx >> 7, c = idx & 127;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += reinterpret_cast<const float*>(&red[vec][k][c >> 2])[c & 3];
    if (vec < 4) {
      if (g.dE_ortho && blockIdx.x == 0) s += g.dE_ortho[vec * HUAL_D + c];
      part[vec * HUAL_D + c] = s;
    } else {
      part[4 * HUAL_D + c * 4 + (vec - 4)] = s;
    }
  }
  if (threadIdx.x < HUAL_D) {
    float s = 0.f;
    if (threadIdx.x < 4)
      for (int k = 0; k < 8; ++k) s += redb[k][threadIdx.x];
    part[8 * HUAL_D + threadIdx.x] = s;
  }
}

// ------------------------------------------------------------------------------------------------------
// The predictor's output end in ONE launch, one workgroup (512 threads) per clip:
//   1. start / end logits = hidden . w + b                      (predictor/{start,end}_dense, modules.py:155-156)
//   2. localizing loss + its gradient, span argmax              (layers.py:177-203)
//   3. dZ = dlogit * w * (hidden > 0) for the two hidden layers and the per-clip sums of d w, d b (folded by colsum_kernel)
//   4. the workgroup that finishes last closes the loss: matching-loss denominator and the four reported terms
// With `h` null the logits are read instead of computed (hual_span_argmax).
// two block-wide reductions for one pair of barriers
__device__ __forceinline__ void block_max2(float& a, float& b, float* sm) {       // sm: 2 * (blockDim.x / 64) floats
  a = wave_max64(a); b = wave_max64(b);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sm[w] = a; sm[nw + w] = b; }
  __syncthreads();
  float x = -INFINITY, y = -INFINITY;
  for (int i = 0; i < nw; ++i) { x = fmaxf(x, sm[i]); y = fmaxf(y, sm[nw + i]); }
  a = x; b = y;
}
__device__ __forceinline__ void block_sum2_d(double& a, double& b, double* sm) {
  for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sm[w] = a; sm[nw + w] = b; }
  __syncthreads();
  double x = 0.0, y = 0.0;
  for (int i = 0; i < nw; ++i) { x += sm[i]; y += sm[nw + i]; }
  a = x; b = y;
}
__device__ __forceinline__ void block_sum3(float& a, float& b, float& c, float* sm) {
  a = wave_sum64(a); b = wave_sum64(b); c = wave_sum64(c);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sm[w] = a; sm[nw + w] = b; sm[2 * nw + w] = c; }
  __syncthreads();
  float x = 0.f, y = 0.f, z = 0.f;
  for (int i = 0; i < nw; ++i) { x += sm[i]; y += sm[nw + i]; z += sm[2 * nw + i]; }
  a = x; b = y; c = z;
}

// NR = rows of the clip per 32-lane group (T <= 16 NR): the hidden rows are loaded ONCE, all loads in flight together, and stay
// in registers for step 3
template <int NR>
__global__ __launch_bounds__(512) void heads_kernel(HeadsArgs a, int T, int B) {
  __shared__ float zs_[256], ze_[256], ps[256], pe[256], dls[256], dle[256];
  __shared__ float sm[24];
  __shared__ double smd[16];
  __shared__ float wtot[2][4];
  __shared__ float bestv[2][4];
  __shared__ int besti[2][4];
  __shared__ float4 red[2][16][32];
  __shared__ float redb[2][16];
  const int b = xcd_tile(blockIdx.x, gridDim.x), tid = threadIdx.x;      // XCD-aware clip order (common.h)
  if (b >= B) return;
  const int l32 = tid & 31, grp = tid >> 5;
  const int col = 4 * l32;
  HEADS_STAMP(0);
  float4 hs[NR], he[NR];
  float4 ws = f4zero(), we = f4zero();
  float poison = 0.f;      // NaN when a weight did not fit its fp16 image (label-free calls; with labels the loss tail carries the flag)
  if (a.ovf) {
    bool bad = false;
    for (int i = tid & 63; i < a.novf; i += 64) bad |= a.ovf[i] != 0u;
    if (__any(bad)) poison = __builtin_nanf("");
  }
  if (a.h[0]) {
    ws = ld4(a.w[0] + col); we = ld4(a.w[1] + col);
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int t = min(grp + 16 * u, T - 1);
      hs[u] = ld4(a.h[0] + (size_t)(b * T + t) * HUAL_D + col);
      he[u] = ld4(a.h[1] + (size_t)(b * T + t) * HUAL_D + col);
    }
  }
  if (a.grad_only) {      // per-block entry point: d logits are inputs, only step 3 runs
    if (tid < T) { dls[tid] = a.ds[b * T + tid]; dle[tid] = a.de[b * T + tid]; }
  } else if (a.h[0]) {
    // ---- 1. logits
    const float bs = a.b[0][0] + poison, be = a.b[1][0] + poison;
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int t = grp + 16 * u;
      const float ds = half_sum32(hs[u].x * ws.x + hs[u].y * ws.y + hs[u].z * ws.z + hs[u].w * ws.w) + bs;
      const float de = half_sum32(he[u].x * we.x + he[u].y * we.y + he[u].z * we.z + he[u].w * we.w) + be;
      if (l32 == 0 && t < T) { zs_[t] = ds; ze_[t] = de; a.logit[0][b * T + t] = ds; a.logit[1][b * T + t] = de; }
    }
  } else if (tid < T) {
    zs_[tid] = a.logit[0][b * T + tid];
    ze_[tid] = a.logit[1][b * T + tid];
  }
  __syncthreads();
  HEADS_STAMP(1);
  // ---- 2. localizing loss + span argmax (threads 0..T-1 hold one frame each; the block-wide reductions take all 512)
  if (!a.grad_only) {
    const int t = tid;
    const bool in = t < T;
    const float m = in ? a.vmask[b * T + t] : 0.f;
    float zs = -INFINITY, ze = -INFINITY;
    if (in) {
      zs = zs_[t] * m + HUAL_MASK_VALUE * (1.0f - m);
      ze = ze_[t] * m + HUAL_MASK_VALUE * (1.0f - m);
    }
    float mxs = zs, mxe = ze;
    block_max2(mxs, mxe, sm);
    // Reproducible float32 softmax (the span indices must be BIT EXACT, north_star): exp of the float32 difference is
    // evaluated in double and rounded once to float32 (= the correctly rounded float32 exp), the denominator is the
    // double-precision sum of those floats rounded once, the quotient is an IEEE float32 division.  None of the three
    // depends on the platform's libm or on a summation order; oracle/seqpan_ref.py::softmax_cr does the same arithmetic.
    const float xs = in ? (float)exp((double)(zs - mxs)) : 0.f;
    const float xe = in ? (float)exp((double)(ze - mxe)) : 0.f;
    double dss = (double)xs, dse = (double)xe;
    block_sum2_d(dss, dse, smd);
    const float sums = (float)dss, sume = (float)dse;
    const float p_s = __fdiv_rn(xs, sums), p_e = __fdiv_rn(xe, sume);
    HEADS_STAMP(2);
    if (a.y1) {
      const float y1 = in ? a.y1[b * T + t] : 0.f, y2 = in ? a.y2[b * T + t] : 0.f;
      const float lsm_s = zs - mxs - logf(sums), lsm_e = ze - mxe - logf(sume);
      float lsum = in ? -(y1 * lsm_s + y2 * lsm_e) : 0.f, y1s = y1, y2s = y2;
      block_sum3(lsum, y1s, y2s, sm);
      if (t == 0 && a.loc_part) a.loc_part[b] = lsum * a.inv_batch;
      if (in) {
        const float ds = (p_s * y1s - y1) * m * a.inv_batch, de = (p_e * y2s - y2) * m * a.inv_batch;
        dls[t] = ds; dle[t] = de;
        if (a.ds) { a.ds[b * T + t] = ds; a.de[b * T + t] = de; }
      }
    }
    HEADS_STAMP(3);
    // start = argmax_i p_s[i] * max_{j>=i} p_e[j] ;  end = argmax_j max_{i<=j} p_s[i] * p_e[j]   (first index on ties).
    // Inclusive prefix maximum of p_s / suffix maximum of p_e: in-wave scans (6 steps), wave totals combined through LDS.
    // (max is exact and order independent, so the scan gives the same numbers as a sequential pass)
    if (tid < 256) {
      const int lane = tid & 63, w = tid >> 6;
      float pre = in ? p_s : 0.f, suf = in ? p_e : 0.f;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const float up = __shfl_up(pre, off), dn = __shfl_down(suf, off);
        if (lane >= off) pre = fmaxf(pre, up);
        if (lane + off < 64) suf = fmaxf(suf, dn);
      }
      if (lane == 63) wtot[0][w] = pre;
      if (lane == 0) wtot[1][w] = suf;
      ps[tid] = pre; pe[tid] = suf;
    }
    // (a barrier that was there, now carrying a flag) NaN logits - an activation beyond the fp16 operand range, |x| >= 4094, turns into
    // Inf - Inf in a product and spreads: the spans of such a clip are not an argmax of anything and come back as -1, like the spans
    // of a step whose weights did not fit (poison)
    const int nanlogit = __syncthreads_or(in && (zs_[t] != zs_[t] || ze_[t] != ze_[t]));
    float vs = -1.f, ve = -1.f;
    if (in) {
      const int w = t >> 6;
      float premax = ps[t], sufmax = pe[t];
      for (int k = 0; k < w; ++k) premax = fmaxf(premax, wtot[0][k]);
      for (int k = w + 1; k < 4; ++k) sufmax = fmaxf(sufmax, wtot[1][k]);
      vs = p_s * sufmax;
      ve = premax * p_e;
    }
    if (tid < 256) {
      int is = t, ie = t;
      for (int off = 32; off >= 1; off >>= 1) {
        float ovs = __shfl_xor(vs, off); int ois = __shfl_xor(is, off);
        if (ovs > vs || (ovs == vs && ois < is)) { vs = ovs; is = ois; }
        float ove = __shfl_xor(ve, off); int oie = __shfl_xor(ie, off);
        if (ove > ve || (ove == ve && oie < ie)) { ve = ove; ie = oie; }
      }
      if ((t & 63) == 0) { bestv[0][t >> 6] = vs; besti[0][t >> 6] = is; bestv[1][t >> 6] = ve; besti[1][t >> 6] = ie; }
    }
    __syncthreads();
    if (t == 0) {
      float v = bestv[0][0]; int i = besti[0][0];
      for (int w = 1; w < 4; ++w) if (bestv[0][w] > v || (bestv[0][w] == v && besti[0][w] < i)) { v = bestv[0][w]; i = besti[0][w]; }
      a.start_index[b] = i;
      v = bestv[1][0]; i = besti[1][0];
      for (int w = 1; w < 4; ++w) if (bestv[1][w] > v || (bestv[1][w] == v && besti[1][w] < i)) { v = bestv[1][w]; i = besti[1][w]; }
      a.end_index[b] = i;
      if (poison != poison || nanlogit) { a.start_index[b] = -1; a.end_index[b] = -1; }
    }
  } else {
    __syncthreads();
  }
  HEADS_STAMP(4);
  HEADS_STAMP(5);
  // ---- 3. gradients of the two hidden layers' outputs + per-clip sums of d w / d b (the hidden rows are still in registers)
  if (a.dZ[0] && (a.y1 || a.grad_only)) {
    float4 dws = f4zero(), dwe = f4zero();
    float dbs = 0.f, dbe = 0.f;
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int tt = grp + 16 * u;
      if (tt >= T) continue;
      const size_t off = (size_t)(b * T + tt) * HUAL_D + col;
      const float dl0 = dls[tt], dl1 = dle[tt];
      st4(a.dZ[0] + off, make_float4(hs[u].x > 0.f ? dl0 * ws.x : 0.f, hs[u].y > 0.f ? dl0 * ws.y : 0.f, hs[u].z > 0.f ? dl0 * ws.z : 0.f,
                                     hs[u].w > 0.f ? dl0 * ws.w : 0.f));
      st4(a.dZ[1] + off, make_float4(he[u].x > 0.f ? dl1 * we.x : 0.f, he[u].y > 0.f ? dl1 * we.y : 0.f, he[u].z > 0.f ? dl1 * we.z : 0.f,
                                     he[u].w > 0.f ? dl1 * we.w : 0.f));
      dws = make_float4(dws.x + dl0 * hs[u].x, dws.y + dl0 * hs[u].y, dws.z + dl0 * hs[u].z, dws.w + dl0 * hs[u].w);
      dwe = make_float4(dwe.x + dl1 * he[u].x, dwe.y + dl1 * he[u].y, dwe.z + dl1 * he[u].z, dwe.w + dl1 * he[u].w);
      if (l32 == 0) { dbs += dl0; dbe += dl1; }
    }
    red[0][grp][l32] = dws; red[1][grp][l32] = dwe;
    if (l32 == 0) { redb[0][grp] = dbs; redb[1][grp] = dbe; }
    __syncthreads();
    // part[hd][b][2][128]: vector 0 = d w, vector 1 = d b in its first entry
    if (tid < 256) {
      const int hd = tid >> 7, c = tid & 127;
      float sw = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) sw += reinterpret_cast<const float*>(&red[hd][k][c >> 2])[c & 3];
      float sb = 0.f;
      if (c == 0)
        for (int k = 0; k < 16; ++k) sb += redb[hd][k];
      float* part = a.part[hd] + (size_t)b * 2 * HUAL_D;
      part[c] = sw;
      part[HUAL_D + c] = sb;
    }
  }
  HEADS_STAMP(6);
}

// Closes the loss (layers.py:173, model.py:91,120): sums the per-workgroup partials of match_fwd_kernel and heads_kernel in a
// fixed order, sets the matching-loss denominator and writes the four reported terms.  A launch of its own: taking a
// "last workgroup" ticket inside heads_kernel needs a device-scope release fence in every workgroup, which on this part
// writes back the L2 (measured: 19 k cycles per workgroup).
// (a device function: the launch of its own when the forward closes the loss, the head of match_bwd_kernel's workgroup 0 when the
//  close is deferred to the backward pass - hual_run_opts.deferred_loss_terms.  Returns the denominator.)
__device__ __forceinline__ float loss_tail_body(const LossTailArgs& a, float* sm, bool write) {
  const int tid = threadIdx.x;
  float cs = 0.f, ms = 0.f, ls = 0.f;
  for (int i = tid; i < a.match_nblk; i += 256) { cs += a.match_part[2 * i]; ms += a.match_part[2 * i + 1]; }
  if (write)
    for (int i = tid; i < a.loc_nblk; i += 256) ls += a.loc_part[i];
  int bad = 0;
  if (write) {
    for (int i = tid; i < a.novf; i += 256) bad |= a.ovf[i] != 0u;
    bad = __syncthreads_or(bad);
  }
  block_sum3(cs, ms, ls, sm);
  const float denom = a.denom_dev ? *a.denom_dev : (a.override_denom > 0.f ? a.override_denom : ms + 1e-12f);
  if (write && tid == 0) {
    float* la = a.loss_acc;
    la[LA_MATCH_SUM] = cs; la[LA_MASK_SUM] = ms; la[LA_LOC] = ls;
    la[LA_DENOM] = denom;
    if (a.align_rows) {
      float l = 0.f;
      for (int i = 0; i < a.nalign; ++i) l += a.align_rows[i];
      la[LA_ALIGN] = l;
    }
    if (a.loss_out) {
      const float match = cs / denom + la[LA_ORTHO];
      const float align = la[LA_ALIGN];
      // a dense weight outside the range of the scaled fp16 images (|w| >= 63): the products of this step are not to be trusted
      const float poison = bad ? __builtin_nanf("") : 0.f;
      a.loss_out[0] = ls + a.lambda * match + align + poison;
      a.loss_out[1] = ls + poison;
      a.loss_out[2] = match + poison;
      a.loss_out[3] = align + poison;
    }
  }
  return denom;
}
__global__ __launch_bounds__(256) void loss_tail_kernel(LossTailArgs a) {
  __shared__ float sm[12];
  loss_tail_body(a, sm, true);
}

// ------------------------------------------------------------------------------------------------------
// alignment loss
#define L2_EPS 1e-12f
__device__ __forceinline__ void align_pool_body(const AlignPool& a, const RowSpace& rs, int b) {
  __shared__ float sm[8];
  __shared__ float part[2][4][HUAL_D];
  const int c = threadIdx.x & 127, grp = threadIdx.x >> 7;
  const int L = rs.L, T = rs.T;
  float tc = 0.f, vc = 0.f;
  for (int l = threadIdx.x; l < L; l += 512) tc += rs.rowmask[rs.Nv + b * L + l];
  for (int t = threadIdx.x; t < T; t += 512) vc += rs.rowmask[b * T + t];
  tc = block_sum(tc, sm);       // exact: the mask is 0/1
  vc = block_sum(vc, sm);
  float ts = 0.f, vs = 0.f;
#pragma unroll 8
  for (int l = grp; l < L; l += 4) ts += a.F2[(size_t)(rs.Nv + b * L + l) * HUAL_D + c];     // padded words included (layers.py:214)
#pragma unroll 8
  for (int t = grp; t < T; t += 4) vs += a.F1[(size_t)(b * T + t) * HUAL_D + c] * (a.inner[b * T + t] / vc);
  part[0][grp][c] = ts;
  part[1][grp][c] = vs;
  __syncthreads();
  ts = part[0][0][c] + part[0][1][c] + part[0][2][c] + part[0][3][c];
  vs = part[1][0][c] + part[1][1][c] + part[1][2][c] + part[1][3][c];
  const float tp = ts / tc;
  const float tn = block_sum(grp == 0 ? tp * tp : 0.f, sm);
  const float vn = block_sum(grp == 0 ? vs * vs : 0.f, sm);
  if (grp == 0) {
    a.tpre[b * HUAL_D + c] = tp;
    a.vpre[b * HUAL_D + c] = vs;
    a.that[(size_t)b * a.ld + c] = tp * rsqrtf(fmaxf(tn, L2_EPS));
    a.vhat[(size_t)b * a.ld + c] = vs * rsqrtf(fmaxf(vn, L2_EPS));
  }
}

// row i of the [Bg,Bg] similarity matrices; one block (256 threads) per row, Bg <= 1024.
// Round 5: evaluated in DOUBLE.  The gradient of a row, p (log p + 1 - 2 p' - dot), is a small difference of O(1) terms (log-sum-exp
// ~ ln Bg against similarities in [-1, 1]): in float32 its rounding put 1.4e-5 of noise on d that (a float32 PyTorch evaluation: 5e-6)
// that every query-side gradient downstream carried - at [Bg, Bg] <= 1024 x 1024 elements per step the double pipe costs nothing.
__device__ __forceinline__ double block_max_d(double v, double* sm) {
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  double s = sm[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) s = fmax(s, sm[i]);
  return s;
}
__device__ __forceinline__ void align_sim_rows_body(const AlignSim& a, int i) {
  __shared__ float ti[HUAL_D], vi[HUAL_D];
  __shared__ double sm[4];
  __shared__ double sa[1024], sq[1024];
  const int tid = threadIdx.x, Bg = a.Bg;
  const int ld = a.ld;
  if (tid < HUAL_D) { ti[tid] = a.that[(size_t)i * ld + tid]; vi[tid] = a.vhat[(size_t)i * ld + tid]; }
  __syncthreads();
  double mxa = -INFINITY, mxq = -INFINITY;
  // four threads per column j, a quarter of the 128 dimensions each (at Bg = 64 every thread of the block works)
  for (int j0 = 0; j0 < Bg; j0 += 64) {
    const int j = j0 + (tid >> 2), kq = (tid & 3) * 32;
    const float* vj = a.vhat + (size_t)min(j, Bg - 1) * ld + kq;
    double da = 0.0, dq = 0.0;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) { const double v = (double)vj[k]; da = fma((double)vi[kq + k], v, da); dq = fma((double)ti[kq + k], v, dq); }
    da += __shfl_xor(da, 1); dq += __shfl_xor(dq, 1);
    da += __shfl_xor(da, 2); dq += __shfl_xor(dq, 2);
    if ((tid & 3) == 0 && j < Bg) { sa[j] = da; sq[j] = dq; }
  }
  __syncthreads();
  for (int j = tid; j < Bg; j += 256) { mxa = fmax(mxa, sa[j]); mxq = fmax(mxq, sq[j]); }
  mxa = block_max_d(mxa, sm);
  mxq = block_max_d(mxq, sm);
  double suma = 0.0, sumq = 0.0;
  for (int j = tid; j < Bg; j += 256) { suma += exp(sa[j] - mxa); sumq += exp(sq[j] - mxq); }
  suma = block_sum_d(suma, sm);
  sumq = block_sum_d(sumq, sm);
  const double lsa = mxa + log(suma), lsq = mxq + log(sumq);
  // loss_i = sum Pq logPq + sum Pv logPv - 2 sum Pq Pv ; dPq = logPq + 1 - 2Pv ; dPv = logPv + 1 - 2Pq
  double li = 0.0, dotq = 0.0, dota = 0.0;
  for (int j = tid; j < Bg; j += 256) {
    const double lpv = sa[j] - lsa, lpq = sq[j] - lsq;
    const double pv = exp(lpv), pq = exp(lpq);
    li += pq * lpq + pv * lpv - 2.0 * pq * pv;
    dotq += pq * (lpq + 1.0 - 2.0 * pv);
    dota += pv * (lpv + 1.0 - 2.0 * pq);
  }
  li = block_sum_d(li, sm);
  dotq = block_sum_d(dotq, sm);
  dota = block_sum_d(dota, sm);
  if (tid == 0) {
    if (a.row_loss) a.row_loss[i] = (float)li;
    else if (a.loss_acc) atomicAdd(a.loss_acc + LA_ALIGN, (float)li);
  }
  __syncthreads();
  for (int j = tid; j < Bg; j += 256) {
    const double lpv = sa[j] - lsa, lpq = sq[j] - lsq;
    const double pv = exp(lpv), pq = exp(lpq);
    const double gq = pq * ((lpq + 1.0 - 2.0 * pv) - dotq) * (double)a.scale;
    const double ga = pv * ((lpv + 1.0 - 2.0 * pq) - dota) * (double)a.scale;
    a.dq[(size_t)i * Bg + j] = (float)gq;
    a.da[(size_t)i * Bg + j] = (float)ga;
    sq[j] = gq; sa[j] = ga;
  }
  __syncthreads();
  // dthat_i = sum_j dq[i][j] vhat_j ; dvhat_i (row part) = sum_j da[i][j] vhat_j
  // only the rows [row0, row0 + nrows) are wanted (exact data parallel: a rank keeps the gradient rows of its own samples)
  if (tid < HUAL_D && i >= a.row0 && i < a.row0 + a.nrows) {
    double st = 0.0, sv = 0.0;
#pragma unroll 8
    for (int j = 0; j < Bg; ++j) {
      const double v = (double)a.vhat[(size_t)j * ld + tid];
      st = fma(sq[j], v, st);
      sv = fma(sa[j], v, sv);
    }
    a.dthat[(i - a.row0) * HUAL_D + tid] = (float)st;
    a.dvhat[(i - a.row0) * HUAL_D + tid] = (float)sv;
  }
}
__global__ __launch_bounds__(256) void align_sim_rows_kernel(AlignSim a) { align_sim_rows_body(a, blockIdx.x); }
// the matching head (workgroups [0, nmatch)) and the rows of the alignment similarity (the rest) in one launch: both only
// depend on kernels further up the stream
__global__ __launch_bounds__(256) void match_align_fwd_kernel(MatchArgs ma, RowSpace rs, AlignSim as, int nmatch) {
  if ((int)blockIdx.x < nmatch) match_fwd_body(ma, rs, blockIdx.x, nmatch);
  else align_sim_rows_body(as, (int)blockIdx.x - nmatch);
}
// column part: dvhat_j += sum_i dq[i][j] that_i + da[i][j] vhat_i
__global__ __launch_bounds__(256) void align_sim_cols_kernel(AlignSim a) {
  __shared__ float cq[1024], ca[1024];
  __shared__ float part[HUAL_D];
  const int j = a.row0 + blockIdx.x, c = threadIdx.x & 127, half = threadIdx.x >> 7, Bg = a.Bg, ld = a.ld;
  for (int i = threadIdx.x; i < Bg; i += 256) { cq[i] = a.dq[(size_t)i * Bg + j]; ca[i] = a.da[(size_t)i * Bg + j]; }
  __syncthreads();
  float s = 0.f;
#pragma unroll 4
  for (int i = half; i < Bg; i += 2)
    s += cq[i] * a.that[(size_t)i * ld + c] + ca[i] * a.vhat[(size_t)i * ld + c];
  if (half) part[c] = s;
  __syncthreads();
  if (!half) a.dvhat[blockIdx.x * HUAL_D + c] += s + part[c];
  if (a.row_loss && blockIdx.x == 0 && threadIdx.x == 0) {      // the loss: sum of the row terms in row order, written (no zeroing launch)
    float l = 0.f;
    for (int i = 0; i < Bg; ++i) l += a.row_loss[i];
    a.loss_acc[LA_ALIGN] = l;
  }
}

// per-clip forward kernels that only read cq.feats, in ONE launch: blockIdx.y = 0 weighted pooling (+ pooled . Wbot),
// 1 = the per-sample part of the alignment loss (when labels are present)
__global__ __launch_bounds__(512) void pool_align_fwd_kernel(PoolArgs pa, AlignPool ap, RowSpace rs) {
  const int clip = xcd_tile(blockIdx.x, gridDim.x);      // XCD-aware clip order (common.h)
  if (clip >= rs.B) return;
  if (blockIdx.y == 0) {
    pool_fwd_body(pa, rs, clip);
  } else {
    align_pool_body(ap, rs, clip);
  }
}
// per-clip backward of the alignment pooling (layers.py:213-229) and of weighted_pooling / cq_concat's pooled half
// (layers.py:133-154), one workgroup of 512 threads per clip: thread (c, grp) owns column c of the rows t = grp + 4 k.
// Every global operand is requested in the first phase (the rows of d fuse and of d cq.feats of this clip sit in registers,
// up to 64 + 64 + 8 per thread), so the kernel pays ONE memory round trip instead of one per reduction step:
//   d cq.feats[v rows] += d vpre * inner / n_v                                  (alignment, video side)
//   d cq.feats[q rows]  = d tpre / n_q + alpha_l * d pooled + d alpha_l * w_pool  (alignment + pooling, query side)
// KT = rows of the video side per thread (T <= 4 KT), KL likewise for the query side.
template <int KT, int KL>
__global__ __launch_bounds__(512) void pool_align_bwd_kernel(PoolArgs pa, PoolBwd pb, AlignPool ap, AlignPoolBwd ab, RowSpace rs) {
  __shared__ float sm[48];
  __shared__ float part[4][HUAL_D];
  __shared__ float vecs[2][HUAL_D];      // [1]: dPW
  __shared__ float da[256], al[256], qm[256], dap[2][256];
  const int b = xcd_tile(blockIdx.x, gridDim.x), tid = threadIdx.x;      // XCD-aware clip order (common.h)
  if (b >= rs.B) return;
  const int c = tid & 127, grp = tid >> 7;
  const int L = rs.L, T = rs.T;
  const size_t vrow0 = (size_t)b * T, qrow0 = (size_t)rs.Nv + (size_t)b * L;
  // ---- phase 0: every load
  // (the inner labels of the clip go through LDS, one float per frame: as a third register array of KT they cost the 64-row
  //  instantiation - T = 256 - 53 spilled registers, 47.6 us per launch at B32 T256)
  __shared__ float inl[256];
  float rdf[KT], rf1[KT], rF[KL];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const int t = min(grp + 4 * k, T - 1);
    rdf[k] = pb.dFuse[(vrow0 + t) * HUAL_D + c];
    rf1[k] = ab.dF1[(vrow0 + t) * HUAL_D + c];
  }
  for (int t = tid; t < T; t += 512) inl[t] = ap.inner[vrow0 + t];      // (published by the barriers of phase 1)
#pragma unroll
  for (int k = 0; k < KL; ++k) rF[k] = pa.F2[(qrow0 + min(grp + 4 * k, L - 1)) * HUAL_D + c];
  const float tp = ap.tpre[b * HUAL_D + c], vp = ap.vpre[b * HUAL_D + c];
  const float th = ap.that[(size_t)b * ap.ld + c], vh = ap.vhat[(size_t)b * ap.ld + c];
  const float dth = ab.dthat[b * HUAL_D + c];
  float dvh = ab.dvhat[b * HUAL_D + c];
  const float wpc = pa.wp[c];
  float tc = 0.f, vc = 0.f;
  for (int l = tid; l < L; l += 512) { const float m = rs.rowmask[qrow0 + l]; qm[l] = m; al[l] = pa.alpha[b * L + l]; tc += m; }
  for (int t = tid; t < T; t += 512) vc += rs.rowmask[vrow0 + t];
  // column part of d vhat_b (align_sim_cols_kernel's sum, when the similarity was evaluated inside the forward)
  float colsum = 0.f;
  if (ab.col_dq) {
    const int Bg = ab.col_Bg;
#pragma unroll 4
    for (int i = grp; i < Bg; i += 4)
      colsum += ab.col_dq[(size_t)i * Bg + b] * ap.that[(size_t)i * ap.ld + c] + ab.col_da[(size_t)i * Bg + b] * ap.vhat[(size_t)i * ap.ld + c];
  }
  // ---- phase 1: workgroup sums.  dPW[c] = sum_t dFuse[t][c]; the column part; the six scalars of the two l2-normalisations
  float sdf = 0.f;
#pragma unroll
  for (int k = 0; k < KT; ++k) sdf += (grp + 4 * k < T) ? rdf[k] : 0.f;
  part[grp][c] = sdf;
  __syncthreads();
  const float dpw = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
  if (grp == 0) { pb.dPW[b * HUAL_D + c] = dpw; vecs[1][c] = dpw; }
  __syncthreads();
  if (ab.col_dq) {
    part[grp][c] = colsum;
    __syncthreads();
    dvh += (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
    __syncthreads();
  }
  {
    // l2_normalize backward: x_hat = x * r, r = rsqrt(max(|x|^2, eps)); dx = r * (dxh - x_hat * (x_hat . dxh)) when |x|^2 > eps
    const float w0 = grp == 0 ? 1.f : 0.f;       // every group holds the same 128 values: count them once
    float v[6] = {w0 * tp * tp, w0 * vp * vp, w0 * th * dth, w0 * vh * dvh, tc, vc};
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = wave_sum64(v[q]);
    if ((tid & 63) == 0)
      for (int q = 0; q < 6; ++q) sm[q * 8 + (tid >> 6)] = v[q];
  }
  __syncthreads();
  float red6[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    float x = 0.f;
    for (int w = 0; w < 8; ++w) x += sm[q * 8 + w];
    red6[q] = x;
  }
  const float tn = red6[0], vn = red6[1], tdot = red6[2], vdot = red6[3];
  tc = red6[4]; vc = red6[5];
  // (the difference dxh - x_hat (x_hat . dxh) in double: it cancels to a few per cent of its terms)
  const float rt = rsqrtf(fmaxf(tn, L2_EPS)), rv = rsqrtf(fmaxf(vn, L2_EPS));
  const float dtp = tn > L2_EPS ? (float)((double)rt * ((double)dth - (double)th * (double)tdot)) : rt * dth;
  const float dvp = vn > L2_EPS ? (float)((double)rv * ((double)dvh - (double)vh * (double)vdot)) : rv * dvh;
  const float dts = dtp / tc;
  // ---- video rows: d cq.feats += d vpre * inner / n_v
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const int t = grp + 4 * k;
    if (t < T) ab.dF1[(vrow0 + t) * HUAL_D + c] = rf1[k] + dvp * (inl[t] / vc);
  }
  // ---- d pooled[c] = sum_n dPW[n] * Wbot[c][n]  (four slices of n)
  {
    float dp = 0.f;
    const float* wrow = pa.Wbot + (size_t)c * HUAL_D + grp * 32;
#pragma unroll
    for (int n = 0; n < 32; n += 4) {
      const float4 w = ld4(wrow + n);
      const float* d = vecs[1] + grp * 32 + n;
      dp += d[0] * w.x + d[1] * w.y + d[2] * w.z + d[3] * w.w;
    }
    part[grp][c] = dp;
  }
  __syncthreads();
  const float dpc = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
  __syncthreads();
  // ---- d alpha[l] = d pooled . F[l]: row l = grp + 4 k belongs to this thread's group; reduce over c within the 128 threads
  //      (two waves) of the group
#pragma unroll
  for (int k = 0; k < KL; ++k) {
    const int l = grp + 4 * k;
    const float d = wave_sum64(dpc * rF[k]);
    if ((tid & 63) == 0 && l < L) dap[(tid >> 6) & 1][l] = d;        // two partials per row: waves 2 grp, 2 grp + 1
  }
  __syncthreads();
  if (tid < L) da[tid] = dap[0][tid] + dap[1][tid];
  __syncthreads();
  float dot_acc = 0.f;
  for (int l = 0; l < L; ++l) dot_acc += al[l] * da[l];              // identical in every thread
  // ---- query rows: d cq.feats = d tpre / n_q + alpha * d pooled + d alpha_l * w_pool ; d w_pool[c] += sum_l d alpha_l F[l][c]
  float dw = 0.f;
#pragma unroll
  for (int k = 0; k < KL; ++k) {
    const int l = grp + 4 * k;
    if (l >= L) continue;
    const float alpha = al[l];
    const float dal = alpha * (da[l] - dot_acc) * qm[l];
    ab.dF2[(qrow0 + l) * HUAL_D + c] = dts + alpha * dpc + dal * wpc;
    dw = fmaf(dal, rF[k], dw);
  }
  part[grp][c] = dw;
  __syncthreads();
  if (grp == 0) atomicAdd(pb.dwp + c, (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]));
}

namespace hual {

int launch_pool_align_fwd(const PoolArgs& a, const AlignPool* ap, const RowSpace& rs, hipStream_t s) {
  HUAL_REQUIRE(rs.L <= 256, "pool: L <= 256");
  AlignPool z{};
  // bytes: cq.feats rows in (query rows for the pooling; all rows for the alignment pooling), the [B,128] results out
  HUAL_LAUNCH(0.0, 512.0 * (rs.Nq + (ap ? rs.R : 0) + 4.0 * rs.B), pool_align_fwd_kernel, dim3(xcd_round8(rs.B), ap ? 2 : 1), dim3(512), 0, s, a, ap ? *ap : z, rs);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_pool_align_bwd(const PoolArgs& a, const PoolBwd& g, const AlignPool& ap, const AlignPoolBwd& ab, const RowSpace& rs, hipStream_t s) {
  HUAL_REQUIRE(rs.T <= 256 && rs.L <= 256, "pool_align_bwd: T, L <= 256");
  HUAL_REQUIRE(ab.dF2 == g.dF2, "pool_align_bwd: the two parts write the same query rows");
  // bytes: d fuse and d cq.feats of the video rows in, d cq.feats of the video rows back out, cq.feats query rows in, their gradient out
  const double pab = 512.0 * (3.0 * rs.Nv + 2.0 * rs.Nq);
  // (KT / KL = rows per thread held in registers: the 64 / 64 form is a wall of spills - 87.6 us at B32 T256 L20, where 64 / 8 does)
  // KT / KL = rows per thread of the video / query side (T <= 4 KT, L <= 4 KL): the register arrays are sized by them, so a shape takes
  // the smallest instantiation that holds it (queries of 33-79 words are a quarter of the reference's ActivityNet batches,
  // tests/golden/lengths_anet.npz: they ran the <64, 64> form - 41 us against 13 - whatever the clip length)
#define PAB(KT, KL) HUAL_LAUNCH(0.0, pab, (pool_align_bwd_kernel<KT, KL>), dim3(xcd_round8(rs.B)), dim3(512), 0, s, a, g, ap, ab, rs)
  if (rs.T <= 128) {
    if (rs.L <= 32) PAB(32, 8); else if (rs.L <= 64) PAB(32, 16); else if (rs.L <= 128) PAB(32, 32); else PAB(64, 64);
  } else {
    if (rs.L <= 32) PAB(64, 8); else if (rs.L <= 64) PAB(64, 16); else if (rs.L <= 128) PAB(64, 32); else PAB(64, 64);
  }
#undef PAB
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
// Grid caps of the kernels below that end in atomics on a handful of addresses: float atomics queued on ONE address
// retire at ~30 ns each on MI355X (match_fwd with 1024 blocks spent 30 us on 2 x 1024 of them), so those kernels use
// few blocks with a grid-stride loop over the rows.
static int rowgrid(int R, int cap) {
  int g = cdiv(R, 8);
  return g < cap ? (g > 0 ? g : 1) : cap;
}
int match_fwd_blocks(int Nv) { return rowgrid(Nv, 512); }
int match_bwd_blocks(int Nv) { return rowgrid(Nv, 256); }
// matching head; `as` non-null: the rows of the alignment similarity ride in the same launch (their column part is left to
// pool_align_bwd_kernel: AlignPoolBwd::col_*)
int launch_match_fwd(const MatchArgs& a, const RowSpace& rs, const AlignSim* as, hipStream_t s) {
  HUAL_REQUIRE(a.part != nullptr, "match_fwd: partial-sum scratch");
  const int nm = match_fwd_blocks(rs.Nv);
  AlignSim z{};
  if (as) {
    HUAL_REQUIRE(as->Bg >= 1 && as->Bg <= 1024 && as->ld >= HUAL_D && as->row0 == 0 && as->nrows == as->Bg, "align: global batch <= 1024");
    z = *as;
  }
  // bytes: fuse in, outputs out (+ the [Nv,4] scores twice)
  HUAL_LAUNCH(0.0, (1024.0 + 32.0) * rs.Nv, match_align_fwd_kernel, dim3(nm + (as ? as->Bg : 0)), dim3(256), 0, s, a, rs, z, nm);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_match_bwd(const MatchArgs& a, const MatchBwd& g, const RowSpace& rs, hipStream_t s) {
  HUAL_REQUIRE(g.part != nullptr, "match_bwd: partial-sum scratch");
  // bytes: two gradient tensors and fuse in, d fuse out
  HUAL_LAUNCH(0.0, 4.0 * 512.0 * rs.Nv, match_bwd_kernel, dim3(match_bwd_blocks(rs.Nv)), dim3(256), 0, s, a, g, rs);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_heads(const HeadsArgs& a, int B, int T, hipStream_t s) {
  HUAL_REQUIRE(T >= 1 && T <= 256 && B >= 1, "heads: T <= 256");
  HUAL_REQUIRE(a.grad_only ? (a.ds && a.de && a.dZ[0]) : (a.logit[0] && a.logit[1] && a.vmask && a.start_index && a.end_index), "heads: null tensor");
  HUAL_REQUIRE(!a.h[0] || (a.h[1] && a.w[0] && a.w[1] && a.b[0] && a.b[1]), "heads: hidden layers incomplete");
  HUAL_REQUIRE(!a.dZ[0] || (a.dZ[1] && a.part[0] && a.part[1] && a.h[0]), "heads: gradient outputs incomplete");
  // bytes: the two hidden layers in, their gradients out (labels present), logits
  const double hb = (a.dZ[0] ? 4.0 : 2.0) * 512.0 * B * T + 16.0 * B * T;
  if (T <= 128) HUAL_LAUNCH(0.0, hb, heads_kernel<8>, dim3(xcd_round8(B)), dim3(512), 0, s, a, T, B);
  else HUAL_LAUNCH(0.0, hb, heads_kernel<16>, dim3(xcd_round8(B)), dim3(512), 0, s, a, T, B);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_loss_tail(const LossTailArgs& a, hipStream_t s) {
  HUAL_REQUIRE(a.loss_acc && a.match_part && a.loc_part, "loss_tail: null");
  HUAL_LAUNCH(0.0, 0.0, loss_tail_kernel, dim3(1), dim3(256), 0, s, a);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_align_sim(const AlignSim& a, hipStream_t s) {
  HUAL_REQUIRE(a.Bg >= 1 && a.Bg <= 1024, "align: global batch <= 1024");
  HUAL_REQUIRE(a.ld >= HUAL_D && a.row0 >= 0 && a.nrows >= 1 && a.row0 + a.nrows <= a.Bg, "align: row window / leading dimension");
  HUAL_LAUNCH(0.0, 0.0, align_sim_rows_kernel, dim3(a.Bg), dim3(256), 0, s, a);
  HUAL_LAUNCH(0.0, 0.0, align_sim_cols_kernel, dim3(a.nrows), dim3(256), 0, s, a);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
