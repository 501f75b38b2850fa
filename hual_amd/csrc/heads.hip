// Heads, losses and span argmax (see heads.h).  These are tiny, HBM/latency-bound row kernels.
#include "heads.h"
#include "prof.h"

using namespace hual;

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
  const int l32 = tid & 31, grp = tid >> 5;
  const int L = rs.L;
  const float* F = a.F2 + (size_t)(rs.Nv + b * L) * HUAL_D;
  const float* m = rs.rowmask + rs.Nv + b * L;
  const float4 w4 = ld4(a.wp + 4 * l32);
  for (int l = grp; l < L; l += 8) {
    const float4 f = ld4(F + (size_t)l * HUAL_D + 4 * l32);
    const float d = half_sum32(f.x * w4.x + f.y * w4.y + f.z * w4.z + f.w * w4.w);
    if (l32 == 0) al[l] = d * m[l] + HUAL_MASK_VALUE * (1.0f - m[l]);   // mask_logits, layers.py:139
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int l = 0; l < L; ++l) mx = fmaxf(mx, al[l]);
  float sum = 0.f;
  for (int l = 0; l < L; ++l) sum += __expf(al[l] - mx);
  const float inv = 1.0f / sum;
  const int c = tid & 127, half = tid >> 7;
  if (half == 0) {
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
#pragma unroll 8
  for (int k = half * 64; k < half * 64 + 64; ++k) o = fmaf(pooled[k], a.Wbot[(size_t)k * HUAL_D + c], o);
  if (half) part[c] = o;
  __syncthreads();
  if (!half) a.PW[b * HUAL_D + c] = o + part[c];
}

// stage 1 (512 threads): dPW[b] = sum_t dFuse[b,t,:]   (4 row groups, then LDS)
// backward of weighted_pooling + cq_concat's pooled half for clip b, 512 threads; dPW = sum_t dFuse is formed here
__device__ __forceinline__ void pool_bwd_body(const PoolArgs& a, const PoolBwd& g, const RowSpace& rs, int b) {
  __shared__ float part4[4][HUAL_D];
  __shared__ float dpw[HUAL_D];
  __shared__ float dps[HUAL_D];
  __shared__ float part[HUAL_D];
  __shared__ float da[256];
  const int tid = threadIdx.x;
  const int L = rs.L, T = rs.T;
  {
    const int c = tid & 127, grp = tid >> 7;
    float s = 0.f;
    const float* dfu = g.dFuse + (size_t)b * T * HUAL_D;
#pragma unroll 8
    for (int t = grp; t < T; t += 4) s += dfu[(size_t)t * HUAL_D + c];
    part4[grp][c] = s;
    __syncthreads();
    if (grp == 0) {
      const float v = part4[0][c] + part4[1][c] + part4[2][c] + part4[3][c];
      g.dPW[b * HUAL_D + c] = v;
      dpw[c] = v;
    }
  }
  __syncthreads();
  const bool on = tid < 256;
  const int c = tid & 127, half = (tid >> 7) & 1;
  const int l32 = tid & 31, grp = (tid >> 5) & 7;
  // dpooled[c] = sum_n dPW[n] * Wbot[c][n]   (two halves of n, float4 loads along the row of Wbot)
  float dp = 0.f;
  if (on) {
    const float* wrow = a.Wbot + (size_t)c * HUAL_D + half * 64;
#pragma unroll 4
    for (int n = 0; n < 64; n += 4) {
      const float4 w = ld4(wrow + n);
      const float* d = dpw + half * 64 + n;
      dp += d[0] * w.x + d[1] * w.y + d[2] * w.z + d[3] * w.w;
    }
    if (half) part[c] = dp;
  }
  __syncthreads();
  if (on && !half) dps[c] = dp + part[c];
  __syncthreads();
  const float* F = a.F2 + (size_t)(rs.Nv + b * L) * HUAL_D;
  const float* m = rs.rowmask + rs.Nv + b * L;
  // dalpha[l] = dpooled . F[l]
  if (on) {
    const float4 dp4 = *reinterpret_cast<const float4*>(dps + 4 * l32);
    for (int l = grp; l < L; l += 8) {
      const float4 f = ld4(F + (size_t)l * HUAL_D + 4 * l32);
      const float d = half_sum32(f.x * dp4.x + f.y * dp4.y + f.z * dp4.z + f.w * dp4.w);
      if (l32 == 0) da[l] = d;
    }
  }
  __syncthreads();
  if (!on || half) return;
  float dot_acc = 0.f;
  for (int l = 0; l < L; ++l) dot_acc += a.alpha[b * L + l] * da[l];     // identical in every thread
  dp = dps[c];
  const float w = a.wp[c];
  float dw = 0.f;
  float* dF = g.dF2 + (size_t)(rs.Nv + b * L) * HUAL_D;
#pragma unroll 4
  for (int l = 0; l < L; ++l) {
    const float alpha = a.alpha[b * L + l];
    const float dal = alpha * (da[l] - dot_acc) * m[l];
    const float f = F[(size_t)l * HUAL_D + c];
    dF[(size_t)l * HUAL_D + c] += alpha * dp + dal * w;
    dw = fmaf(dal, f, dw);
  }
  atomicAdd(g.dwp + c, dw);
}
__global__ __launch_bounds__(256) void match_fwd_kernel(MatchArgs a, RowSpace rs) {
  __shared__ float red[2][8];
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  float ce_sum = 0.f, m_sum = 0.f;
  // Wm rows col..col+3 (4 classes each)
  float4 w0 = ld4(a.Wm + (col + 0) * 4), w1 = ld4(a.Wm + (col + 1) * 4), w2 = ld4(a.Wm + (col + 2) * 4), w3 = ld4(a.Wm + (col + 3) * 4);
  const float4 bm = ld4(a.bm);
  float4 e0 = ld4(a.E + col), e1 = ld4(a.E + HUAL_D + col), e2 = ld4(a.E + 2 * HUAL_D + col), e3 = ld4(a.E + 3 * HUAL_D + col);
  for (int row = blockIdx.x * 8 + grp; row < rs.Nv; row += gridDim.x * 8) {
    const size_t off = (size_t)row * HUAL_D + col;
    const float4 f = ld4(a.fuse + off);
    float l0 = f.x * w0.x + f.y * w1.x + f.z * w2.x + f.w * w3.x;
    float l1 = f.x * w0.y + f.y * w1.y + f.z * w2.y + f.w * w3.y;
    float l2 = f.x * w0.z + f.y * w1.z + f.z * w2.z + f.w * w3.z;
    float l3 = f.x * w0.w + f.y * w1.w + f.z * w2.w + f.w * w3.w;
    l0 = half_sum32(l0) + bm.x; l1 = half_sum32(l1) + bm.y; l2 = half_sum32(l2) + bm.z; l3 = half_sum32(l3) + bm.w;
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
      if (a.part) { a.part[2 * blockIdx.x] = c; a.part[2 * blockIdx.x + 1] = m; }
      else { atomicAdd(a.loss_acc + LA_MATCH_SUM, c); atomicAdd(a.loss_acc + LA_MASK_SUM, m); }
    }
  }
}

__global__ void match_denominator_kernel(float* loss_acc, float override_denom) {
  loss_acc[LA_DENOM] = override_denom > 0.f ? override_denom : loss_acc[LA_MASK_SUM] + 1e-12f;
}

__global__ __launch_bounds__(256) void match_bwd_kernel(MatchArgs a, MatchBwd g, RowSpace rs) {
  __shared__ float4 red[8][8][32];    // [vec][grp][lane]: 4 dE rows + 4 dWm columns(as float4 over cols)
  __shared__ float redb[8][4];
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  float4 w0 = ld4(a.Wm + (col + 0) * 4), w1 = ld4(a.Wm + (col + 1) * 4), w2 = ld4(a.Wm + (col + 2) * 4), w3 = ld4(a.Wm + (col + 3) * 4);
  float4 e0 = ld4(a.E + col), e1 = ld4(a.E + HUAL_D + col), e2 = ld4(a.E + 2 * HUAL_D + col), e3 = ld4(a.E + 3 * HUAL_D + col);
  const float ce_scale = a.labels ? g.lambda / a.loss_acc[LA_DENOM] : 0.f;
  float4 dE0 = f4zero(), dE1 = f4zero(), dE2 = f4zero(), dE3 = f4zero();
  float4 dW0 = f4zero(), dW1 = f4zero(), dW2 = f4zero(), dW3 = f4zero();   // dWk = column k of dWm over this lane's 4 rows
  float db0 = 0.f, db1 = 0.f, db2 = 0.f, db3 = 0.f;
#pragma unroll 2
  for (int row = blockIdx.x * 8 + grp; row < rs.Nv; row += gridDim.x * 8) {
    const size_t off = (size_t)row * HUAL_D + col;
    const float mk = rs.rowmask[row];
    float4 d = ld4(g.dOut + off);
    if (g.dOut2) { const float4 d2 = ld4(g.dOut2 + off); d = make_float4(d.x + d2.x, d.y + d2.y, d.z + d2.z, d.w + d2.w); }
    d = make_float4(d.x * mk, d.y * mk, d.z * mk, d.w * mk);   // through the *v_mask of model.py:97
    const float4 f = ld4(a.fuse + off);
    const float4 p = *reinterpret_cast<const float4*>(a.probs + (size_t)row * 4);
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

// ortho: one block of 128 threads
// forward use (tail != 0) also closes the loss: match denominator (layers.py:173) and the four reported loss terms
__global__ __launch_bounds__(128) void ortho_kernel(const float* E, float* dE, float* loss_acc, float lambda, int tail,
                                                    float override_denom, const float* denom_dev, float* loss_out,
                                                    const float* match_part, int match_nblk, float* dE_store) {
  __shared__ float M[16];
  __shared__ float sm[4];
  const int c = threadIdx.x;
  if (tail && match_part) {      // per-block partial sums of match_fwd_kernel, in a fixed order
    float cs = 0.f, ms = 0.f;
    for (int i = c; i < match_nblk; i += 128) { cs += match_part[2 * i]; ms += match_part[2 * i + 1]; }
    cs = block_sum(cs, sm);
    ms = block_sum(ms, sm);
    if (c == 0) { loss_acc[LA_MATCH_SUM] = cs; loss_acc[LA_MASK_SUM] = ms; }
    __syncthreads();
  }
  float e[4];
  for (int i = 0; i < 4; ++i) e[i] = E[i * HUAL_D + c];
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 4; ++k) {
      float d = block_sum(e[i] * e[k], sm);
      if (c == 0) M[i * 4 + k] = (i == k) ? 0.f : d;
    }
  __syncthreads();
  float ss = 0.f;
  for (int i = 0; i < 16; ++i) ss += M[i] * M[i];
  const float nrm = sqrtf(ss);
  if (c == 0) {
    loss_acc[LA_ORTHO] = nrm;
    if (tail) {
      const float denom = denom_dev ? *denom_dev : (override_denom > 0.f ? override_denom : loss_acc[LA_MASK_SUM] + 1e-12f);
      loss_acc[LA_DENOM] = denom;
      if (loss_out) {
        const float match = loss_acc[LA_MATCH_SUM] / denom + nrm;             // layers.py:173 + model.py:91
        const float loc = loss_acc[LA_LOC], align = loss_acc[LA_ALIGN];
        loss_out[0] = loc + lambda * match + align;                           // model.py:120
        loss_out[1] = loc;
        loss_out[2] = match;
        loss_out[3] = align;
      }
    }
  }
  if (dE && nrm > 0.f) {
    for (int i = 0; i < 4; ++i) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += M[i * 4 + k] * e[k];
      dE[i * HUAL_D + c] += lambda * 2.0f * s / nrm;
    }
  }
  if (dE_store) {      // the same gradient, left in scratch for match_bwd_kernel to fold in (one launch fewer in backward)
    for (int i = 0; i < 4; ++i) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += M[i * 4 + k] * e[k];
      dE_store[i * HUAL_D + c] = nrm > 0.f ? lambda * 2.0f * s / nrm : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(DotArgs a) {
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  const int hd = blockIdx.y;
  const float4 w = ld4(a.w[hd] + col);
  const float b = a.b[hd][0];
  for (int row = blockIdx.x * 8 + grp; row < a.R; row += gridDim.x * 8) {
    const float4 h = ld4(a.h[hd] + (size_t)row * HUAL_D + col);
    float d = half_sum32(h.x * w.x + h.y * w.y + h.z * w.z + h.w * w.w);
    if (l32 == 0) a.logit[hd][row] = d + b;
  }
}

__global__ __launch_bounds__(256) void rowdot_bwd_kernel(DotArgs a, DotBwd g) {
  __shared__ float4 red[8][32];
  __shared__ float redb[8];
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  const int hd = blockIdx.y;
  const float4 w = ld4(a.w[hd] + col);
  float4 dw = f4zero();
  float db = 0.f;
  for (int row = blockIdx.x * 8 + grp; row < a.R; row += gridDim.x * 8) {
    const size_t off = (size_t)row * HUAL_D + col;
    const float4 h = ld4(a.h[hd] + off);
    const float dl = g.dlogit[hd][row];
    st4(g.dZ[hd] + off, make_float4(h.x > 0.f ? dl * w.x : 0.f, h.y > 0.f ? dl * w.y : 0.f, h.z > 0.f ? dl * w.z : 0.f,
                                    h.w > 0.f ? dl * w.w : 0.f));
    dw = make_float4(dw.x + dl * h.x, dw.y + dl * h.y, dw.z + dl * h.z, dw.w + dl * h.w);
    if (l32 == 0) db += dl;
  }
  red[grp][l32] = dw;
  if (l32 == 0) redb[grp] = db;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += reinterpret_cast<const float*>(&red[k][c >> 2])[c & 3];
    atomicAdd(g.dw[hd] + c, s);
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += redb[k];
    atomicAdd(g.db[hd], s);
  }
}

// ------------------------------------------------------------------------------------------------------
// localizing loss + span argmax.  One block (256 threads) per clip, T <= 256.
__global__ __launch_bounds__(256) void loc_kernel(LocArgs a, int T) {
  __shared__ float ps[256], pe[256];
  __shared__ float sm[4];
  __shared__ double smd[4];
  __shared__ float bestv[2][4];
  __shared__ int besti[2][4];
  const int b = blockIdx.x, t = threadIdx.x;
  const bool in = t < T;
  const float m = in ? a.vmask[b * T + t] : 0.f;
  float zs = -INFINITY, ze = -INFINITY;
  if (in) {
    zs = a.s_logit[b * T + t] * m + HUAL_MASK_VALUE * (1.0f - m);
    ze = a.e_logit[b * T + t] * m + HUAL_MASK_VALUE * (1.0f - m);
  }
  const float mxs = block_max(zs, sm);
  const float mxe = block_max(ze, sm);
  // Reproducible float32 softmax (the span indices must be BIT EXACT, north_star): exp of the float32 difference is
  // evaluated in double and rounded once to float32 (= the correctly rounded float32 exp), the denominator is the
  // double-precision sum of those floats rounded once, the quotient is an IEEE float32 division.  None of the three
  // depends on the platform's libm or on a summation order; oracle/seqpan_ref.py::softmax_cr does the same arithmetic.
  const float xs = in ? (float)exp((double)(zs - mxs)) : 0.f;
  const float xe = in ? (float)exp((double)(ze - mxe)) : 0.f;
  const float sums = (float)block_sum_d((double)xs, smd);
  const float sume = (float)block_sum_d((double)xe, smd);
  const float p_s = __fdiv_rn(xs, sums), p_e = __fdiv_rn(xe, sume);
  ps[t] = p_s;
  pe[t] = p_e;
  if (a.y1) {
    const float y1 = in ? a.y1[b * T + t] : 0.f, y2 = in ? a.y2[b * T + t] : 0.f;
    const float lsm_s = zs - mxs - logf(sums), lsm_e = ze - mxe - logf(sume);
    float l = in ? -(y1 * lsm_s + y2 * lsm_e) : 0.f;
    const float lsum = block_sum(l, sm);
    const float y1s = block_sum(y1, sm), y2s = block_sum(y2, sm);
    if (t == 0 && a.loss_acc) atomicAdd(a.loss_acc + LA_LOC, lsum * a.inv_batch);
    if (a.ds && in) {
      a.ds[b * T + t] = (p_s * y1s - y1) * m * a.inv_batch;
      a.de[b * T + t] = (p_e * y2s - y2) * m * a.inv_batch;
    }
  }
  __syncthreads();
  // start = argmax_i p_s[i] * max_{j>=i} p_e[j] ;  end = argmax_j max_{i<=j} p_s[i] * p_e[j]   (first index on ties)
  float vs = -1.f, ve = -1.f;
  if (in) {
    float sufmax = 0.f, premax = 0.f;
    for (int jx = t; jx < T; ++jx) sufmax = fmaxf(sufmax, pe[jx]);
    for (int ix = 0; ix <= t; ++ix) premax = fmaxf(premax, ps[ix]);
    vs = p_s * sufmax;
    ve = premax * p_e;
  }
  // wave argmax with first-index ties, then across the 4 waves
  int is = t, ie = t;
  for (int off = 32; off >= 1; off >>= 1) {
    float ovs = __shfl_xor(vs, off); int ois = __shfl_xor(is, off);
    if (ovs > vs || (ovs == vs && ois < is)) { vs = ovs; is = ois; }
    float ove = __shfl_xor(ve, off); int oie = __shfl_xor(ie, off);
    if (ove > ve || (ove == ve && oie < ie)) { ve = ove; ie = oie; }
  }
  if ((t & 63) == 0) { bestv[0][t >> 6] = vs; besti[0][t >> 6] = is; bestv[1][t >> 6] = ve; besti[1][t >> 6] = ie; }
  __syncthreads();
  if (t == 0) {
    float v = bestv[0][0]; int i = besti[0][0];
    for (int w = 1; w < 4; ++w) if (bestv[0][w] > v || (bestv[0][w] == v && besti[0][w] < i)) { v = bestv[0][w]; i = besti[0][w]; }
    a.start_index[b] = i;
    v = bestv[1][0]; i = besti[1][0];
    for (int w = 1; w < 4; ++w) if (bestv[1][w] > v || (bestv[1][w] == v && besti[1][w] < i)) { v = bestv[1][w]; i = besti[1][w]; }
    a.end_index[b] = i;
  }
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
    a.that[b * HUAL_D + c] = tp * rsqrtf(fmaxf(tn, L2_EPS));
    a.vhat[b * HUAL_D + c] = vs * rsqrtf(fmaxf(vn, L2_EPS));
  }
}

// row i of the [Bg,Bg] similarity matrices; one block (256 threads) per row, Bg <= 1024
__global__ __launch_bounds__(256) void align_sim_rows_kernel(AlignSim a) {
  __shared__ float ti[HUAL_D], vi[HUAL_D];
  __shared__ float sm[4];
  __shared__ float sa[1024], sq[1024];
  const int i = blockIdx.x, tid = threadIdx.x, Bg = a.Bg;
  const int ld = a.ld;
  if (tid < HUAL_D) { ti[tid] = a.that[(size_t)i * ld + tid]; vi[tid] = a.vhat[(size_t)i * ld + tid]; }
  __syncthreads();
  float mxa = -INFINITY, mxq = -INFINITY;
  for (int j = tid; j < Bg; j += 256) {
    const float* vj = a.vhat + (size_t)j * ld;
    float da = 0.f, dq = 0.f;
#pragma unroll 16
    for (int k = 0; k < HUAL_D; ++k) { da = fmaf(vi[k], vj[k], da); dq = fmaf(ti[k], vj[k], dq); }
    sa[j] = da; sq[j] = dq;
    mxa = fmaxf(mxa, da); mxq = fmaxf(mxq, dq);
  }
  mxa = block_max(mxa, sm);
  mxq = block_max(mxq, sm);
  float suma = 0.f, sumq = 0.f;
  for (int j = tid; j < Bg; j += 256) { suma += expf(sa[j] - mxa); sumq += expf(sq[j] - mxq); }
  suma = block_sum(suma, sm);
  sumq = block_sum(sumq, sm);
  const float lsa = mxa + logf(suma), lsq = mxq + logf(sumq);
  // loss_i = sum Pq logPq + sum Pv logPv - 2 sum Pq Pv ; dPq = logPq + 1 - 2Pv ; dPv = logPv + 1 - 2Pq
  float li = 0.f, dotq = 0.f, dota = 0.f;
  for (int j = tid; j < Bg; j += 256) {
    const float lpv = sa[j] - lsa, lpq = sq[j] - lsq;
    const float pv = expf(lpv), pq = expf(lpq);
    li += pq * lpq + pv * lpv - 2.0f * pq * pv;
    dotq += pq * (lpq + 1.0f - 2.0f * pv);
    dota += pv * (lpv + 1.0f - 2.0f * pq);
  }
  li = block_sum(li, sm);
  dotq = block_sum(dotq, sm);
  dota = block_sum(dota, sm);
  if (tid == 0 && a.loss_acc) atomicAdd(a.loss_acc + LA_ALIGN, li);
  __syncthreads();
  for (int j = tid; j < Bg; j += 256) {
    const float lpv = sa[j] - lsa, lpq = sq[j] - lsq;
    const float pv = expf(lpv), pq = expf(lpq);
    const float gq = pq * ((lpq + 1.0f - 2.0f * pv) - dotq) * a.scale;
    const float ga = pv * ((lpv + 1.0f - 2.0f * pq) - dota) * a.scale;
    a.dq[(size_t)i * Bg + j] = gq;
    a.da[(size_t)i * Bg + j] = ga;
    sq[j] = gq; sa[j] = ga;
  }
  __syncthreads();
  // dthat_i = sum_j dq[i][j] vhat_j ; dvhat_i (row part) = sum_j da[i][j] vhat_j
  // only the rows [row0, row0 + nrows) are wanted (exact data parallel: a rank keeps the gradient rows of its own samples)
  if (tid < HUAL_D && i >= a.row0 && i < a.row0 + a.nrows) {
    float st = 0.f, sv = 0.f;
#pragma unroll 8
    for (int j = 0; j < Bg; ++j) {
      const float v = a.vhat[(size_t)j * ld + tid];
      st = fmaf(sq[j], v, st);
      sv = fmaf(sa[j], v, sv);
    }
    a.dthat[(i - a.row0) * HUAL_D + tid] = st;
    a.dvhat[(i - a.row0) * HUAL_D + tid] = sv;
  }
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
}

__device__ __forceinline__ void align_pool_bwd_body(const AlignPool& a, const AlignPoolBwd& g, const RowSpace& rs, int b) {
  __shared__ float sm[8];
  const int c = threadIdx.x & 127, grp = threadIdx.x >> 7;
  const int L = rs.L, T = rs.T;
  // l2_normalize backward: x_hat = x * r, r = rsqrt(max(|x|^2, eps)); dx = r * (dxh - x_hat * (x_hat . dxh)) when |x|^2 > eps
  const float tp = a.tpre[b * HUAL_D + c], vp = a.vpre[b * HUAL_D + c];
  const float th = a.that[b * HUAL_D + c], vh = a.vhat[b * HUAL_D + c];
  const float dth = g.dthat[b * HUAL_D + c], dvh = g.dvhat[b * HUAL_D + c];
  const float w0 = grp == 0 ? 1.f : 0.f;     // every group holds the same 128 values: count them once
  const float tn = block_sum(w0 * tp * tp, sm), vn = block_sum(w0 * vp * vp, sm);
  const float tdot = block_sum(w0 * th * dth, sm), vdot = block_sum(w0 * vh * dvh, sm);
  const float rt = rsqrtf(fmaxf(tn, L2_EPS)), rv = rsqrtf(fmaxf(vn, L2_EPS));
  const float dtp = tn > L2_EPS ? rt * (dth - th * tdot) : rt * dth;
  const float dvp = vn > L2_EPS ? rv * (dvh - vh * vdot) : rv * dvh;
  float tc = 0.f, vc = 0.f;
  for (int l = threadIdx.x; l < L; l += 512) tc += rs.rowmask[rs.Nv + b * L + l];
  for (int t = threadIdx.x; t < T; t += 512) vc += rs.rowmask[b * T + t];
  tc = block_sum(tc, sm);
  vc = block_sum(vc, sm);
  const float dts = dtp / tc;
  for (int l = grp; l < L; l += 4) g.dF2[(size_t)(rs.Nv + b * L + l) * HUAL_D + c] = dts;
#pragma unroll 8
  for (int t = grp; t < T; t += 4) g.dF1[(size_t)(b * T + t) * HUAL_D + c] += dvp * (a.inner[b * T + t] / vc);
}

// per-clip forward kernels that only read cq.feats, in ONE launch: blockIdx.y = 0 weighted pooling (+ pooled . Wbot),
// 1 = the per-sample part of the alignment loss (when labels are present)
__global__ __launch_bounds__(512) void pool_align_fwd_kernel(PoolArgs pa, AlignPool ap, RowSpace rs) {
  if (blockIdx.y == 0) {
    if (threadIdx.x >= 256) return;      // whole waves leave: the body is written for 256 threads
    pool_fwd_body(pa, rs, blockIdx.x);
  } else {
    align_pool_body(ap, rs, blockIdx.x);
  }
}
// per-clip backward: alignment pooling (writes the query rows of d cq.feats, accumulates the video rows), then weighted
// pooling / cq_concat's pooled half (accumulates into the query rows the first part just wrote - same workgroup, same clip)
__global__ __launch_bounds__(512) void pool_align_bwd_kernel(PoolArgs pa, PoolBwd pb, AlignPool ap, AlignPoolBwd ab, RowSpace rs) {
  align_pool_bwd_body(ap, ab, rs, blockIdx.x);
  __syncthreads();
  pool_bwd_body(pa, pb, rs, blockIdx.x);
}

namespace hual {

int launch_pool_align_fwd(const PoolArgs& a, const AlignPool* ap, const RowSpace& rs, hipStream_t s) {
  HUAL_REQUIRE(rs.L <= 256, "pool: L <= 256");
  AlignPool z{};
  HUAL_LAUNCH(0.0, 0.0, pool_align_fwd_kernel, dim3(rs.B, ap ? 2 : 1), dim3(512), 0, s, a, ap ? *ap : z, rs);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_pool_align_bwd(const PoolArgs& a, const PoolBwd& g, const AlignPool& ap, const AlignPoolBwd& ab, const RowSpace& rs, hipStream_t s) {
  HUAL_LAUNCH(0.0, 0.0, pool_align_bwd_kernel, dim3(rs.B), dim3(512), 0, s, a, g, ap, ab, rs);
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
int launch_match_fwd(const MatchArgs& a, const RowSpace& rs, hipStream_t s) {
  HUAL_LAUNCH(0.0, 0.0, match_fwd_kernel, dim3(a.part ? match_fwd_blocks(rs.Nv) : rowgrid(rs.Nv, 128)), dim3(256), 0, s, a, rs);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_match_denominator(float* loss_acc, float override_denom, hipStream_t s) {
  HUAL_LAUNCH(0.0, 0.0, match_denominator_kernel, dim3(1), dim3(1), 0, s, loss_acc, override_denom);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_match_bwd(const MatchArgs& a, const MatchBwd& g, const RowSpace& rs, hipStream_t s) {
  HUAL_REQUIRE(g.part != nullptr, "match_bwd: partial-sum scratch");
  HUAL_LAUNCH(0.0, 0.0, match_bwd_kernel, dim3(match_bwd_blocks(rs.Nv)), dim3(256), 0, s, a, g, rs);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_ortho(const float* E, float* dE, float* loss_acc, float lambda, hipStream_t s) {
  HUAL_LAUNCH(0.0, 0.0, ortho_kernel, dim3(1), dim3(128), 0, s, E, dE, loss_acc, lambda, 0, 0.f, (const float*)nullptr, (float*)nullptr,
              (const float*)nullptr, 0, (float*)nullptr);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_loss_tail(const float* E, float* loss_acc, float lambda, float override_denom, const float* denom_dev, float* loss_out,
                     const float* match_part, int match_nblk, float* dE_ortho, hipStream_t s) {
  HUAL_LAUNCH(0.0, 0.0, ortho_kernel, dim3(1), dim3(128), 0, s, E, (float*)nullptr, loss_acc, lambda, 1, override_denom, denom_dev, loss_out,
              match_part, match_nblk, dE_ortho);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_rowdot_fwd(const DotArgs& a, hipStream_t s) {
  HUAL_LAUNCH(0.0, 0.0, rowdot_fwd_kernel, dim3(rowgrid(a.R, 1024), 2), dim3(256), 0, s, a);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_rowdot_bwd(const DotArgs& a, const DotBwd& g, hipStream_t s) {
  HUAL_LAUNCH(0.0, 0.0, rowdot_bwd_kernel, dim3(rowgrid(a.R, 64), 2), dim3(256), 0, s, a, g);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_loc(const LocArgs& a, int B, int T, hipStream_t s) {
  HUAL_REQUIRE(T <= 256, "loc: T <= 256");
  HUAL_LAUNCH(0.0, 0.0, loc_kernel, dim3(B), dim3(256), 0, s, a, T);
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
