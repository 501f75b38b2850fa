// Context-query attention kernels (see cq.h).  All small matrix products of one clip run on
// v_mfma_f32_16x16x4_f32 through one device helper (tile_mma) whose operands may sit in LDS or global memory,
// K-contiguous (float4 fragment loads) or K-strided (4 scalar loads) - same fragment maps as gemm.hip/attn.hip.
#include <stdlib.h>
#include "cq.h"
#include "philox.h"
#include "prof.h"

using namespace hual;

namespace hual {
int cq_padded(int n) { return (n + 15) & ~15; }
}

__device__ __forceinline__ f32x4 mfma16c(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// C tile [i0,i0+16) x [n0,n0+16) of  sum_k A(i,k) B(k,n),  K % 16 == 0.
//   AK: A(i,k) = A[i*lda + k]  else A(i,k) = A[k*lda + i]
//   BK: B(k,n) = B[n*ldb + k]  else B(k,n) = B[k*ldb + n]
// Row indices on the non-K axis are clamped to [0,imax) / [0,nmax); K-axis indices of a strided operand are
// clamped to [0,kmaxA) / [0,kmaxB) (exactly one of the two operands must be zero in the K padding).  Result: lane (j,g) reg r = C[i0+4g+r][n0+j].
template <bool AK, bool BK>
__device__ __forceinline__ f32x4 tile_mma(const float* A, int lda, int imax, const float* B, int ldb, int nmax, int K,
                                          int kmaxA, int kmaxB, int i0, int n0, int j, int g, f32x4 acc) {
  const int ia = min(i0 + j, imax - 1);
  const int nb = min(n0 + j, nmax - 1);
#pragma unroll 4
  for (int k0 = 0; k0 < K; k0 += 16) {
    float a[4], b[4];
    if (AK) {
      float4 v = ld4(A + (size_t)ia * lda + k0 + 4 * g);
      a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) a[c] = A[(size_t)min(k0 + 4 * g + c, kmaxA - 1) * lda + ia];
    }
    if (BK) {
      float4 v = ld4(B + (size_t)nb * ldb + k0 + 4 * g);
      b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = B[(size_t)min(k0 + 4 * g + c, kmaxB - 1) * ldb + nb];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc = mfma16c(a[c], b[c], acc);
  }
  return acc;
}

// per-clip kernels: one block = one (clip, direction); 16 waves share the tiles / rows of every phase
#define CQ_MAX_THREADS 1024
#define CQ_THREADS ((int)blockDim.x)
#define CQ_WAVES ((int)blockDim.x >> 6)
struct ClipGeom {
  int N1, N2, N1p, N2p, ld, x1base, x2base;
};
__device__ __forceinline__ ClipGeom clip_geom(const RowSpace& rs, int clip, int dir) {
  ClipGeom c;
  if (dir == 0) { c.N1 = rs.T; c.N2 = rs.L; c.x1base = clip * rs.T; c.x2base = rs.Nv + clip * rs.L; }
  else { c.N1 = rs.L; c.N2 = rs.T; c.x1base = rs.Nv + clip * rs.L; c.x2base = clip * rs.T; }
  c.N1p = (c.N1 + 15) & ~15;
  c.N2p = (c.N2 + 15) & ~15;
  c.ld = c.N2p + 4;
  return c;
}
__host__ __device__ inline size_t cq_mat_elems(int T, int L) {   // max over both directions of N1p*(N2p+4)
  int Tp = (T + 15) & ~15, Lp = (L + 15) & ~15;
  size_t a = (size_t)Tp * (Lp + 4), b = (size_t)Lp * (Tp + 4);
  return a > b ? a : b;
}
__host__ __device__ inline size_t cq_m2_rows(int T, int L) {
  int Tp = (T + 15) & ~15, Lp = (L + 15) & ~15;
  return Tp > Lp ? Tp : Lp;
}

// ------------------------------------------------------------------------------------------------------
// tri_prep: dropout on both roles of every row + the two rank-1 terms of the trilinear score (ops.py:104-114)
__global__ __launch_bounds__(256) void tri_prep_kernel(CqBufs b, CqParams p, RowSpace rs, DropCfg drop) {
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  for (int row = blockIdx.x * 8 + grp; row < rs.R; row += gridDim.x * 8) {
    const bool isv = row < rs.Nv;
    const int d1 = isv ? 0 : 1;          // direction in which this row plays x1
    const int d2 = isv ? 1 : 0;          // direction in which this row plays x2
    const uint32_t site1 = (uint32_t)HUAL_SITE_TRI + (isv ? 0u : 2u);
    const uint32_t site2 = (uint32_t)HUAL_SITE_TRI + (isv ? 3u : 1u);
    const size_t off = (size_t)row * HUAL_D + col;
    float4 x = ld4(b.X + off);
    float4 a = x, c = x;
    if (drop.enabled) {
      a = apply_drop4(drop, site1, (uint32_t)row, (uint32_t)l32, x);
      c = apply_drop4(drop, site2, (uint32_t)row, (uint32_t)l32, x);
    }
    float4 w0 = ld4(p.w0[d1] + col), wm = ld4(p.wm[d1] + col), w1 = ld4(p.w1[d2] + col);
    float s0 = half_sum32(a.x * w0.x + a.y * w0.y + a.z * w0.z + a.w * w0.w);
    float s1 = half_sum32(c.x * w1.x + c.y * w1.y + c.z * w1.z + c.w * w1.w);
    st4(b.D1W + off, make_float4(a.x * wm.x, a.y * wm.y, a.z * wm.z, a.w * wm.w));
    st4(b.D2 + off, c);
    if (l32 == 0) { b.S0[row] = s0; b.S1[row] = s1; }
  }
}

// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CQ_MAX_THREADS) void cq_fwd_kernel(CqBufs b, RowSpace rs) {
  extern __shared__ float lds[];
  const int clip = blockIdx.x, dir = blockIdx.y;
  const ClipGeom c = clip_geom(rs, clip, dir);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int msz = c.N1p * c.ld;
  float* S = lds;
  float* Sr = lds + msz;
  float* Sc = lds + 2 * msz;
  const float* m1 = rs.rowmask + c.x1base;
  const float* m2 = rs.rowmask + c.x2base;
  const float* X1 = b.X + (size_t)c.x1base * HUAL_D;
  const float* X2 = b.X + (size_t)c.x2base * HUAL_D;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // ---- score = d1w . d2^T + s0 + s1
  const int nj = c.N2p >> 4, ni = c.N1p >> 4;
  for (int tile = wave; tile < ni * nj; tile += CQ_WAVES) {
    const int i0 = (tile / nj) * 16, n0 = (tile % nj) * 16;
    f32x4 acc = tile_mma<true, true>(b.D1W + (size_t)c.x1base * HUAL_D, HUAL_D, c.N1, b.D2 + (size_t)c.x2base * HUAL_D,
                                     HUAL_D, c.N2, HUAL_D, HUAL_D, HUAL_D, i0, n0, j, g, zero);
    const float s1 = b.S1[c.x2base + min(n0 + j, c.N2 - 1)];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * g + r;
      S[i * c.ld + n0 + j] = acc[r] + b.S0[c.x1base + min(i, c.N1 - 1)] + s1;
    }
  }
  __syncthreads();
  // ---- row softmax over j with mask2 (layers.py:122-123) ; zero outside the valid block
  for (int i = wave; i < c.N1p; i += CQ_WAVES) {
    float mx = -INFINITY;
    if (i < c.N1)
      for (int jj = lane; jj < c.N2; jj += 64) {
        const float mk = m2[jj];
        mx = fmaxf(mx, S[i * c.ld + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk));
      }
    mx = wave_max64(mx);
    float sum = 0.f;
    if (i < c.N1)
      for (int jj = lane; jj < c.N2; jj += 64) {
        const float mk = m2[jj];
        sum += __expf(S[i * c.ld + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk) - mx);
      }
    sum = wave_sum64(sum);
    const float inv = 1.0f / sum;
    for (int jj = lane; jj < c.N2p; jj += 64) {
      float v = 0.f;
      if (i < c.N1 && jj < c.N2) {
        const float mk = m2[jj];
        v = __expf(S[i * c.ld + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk) - mx) * inv;
      }
      Sr[i * c.ld + jj] = v;
    }
  }
  // ---- column softmax over i with mask1 (layers.py:124-125)
  for (int jj = wave; jj < c.N2p; jj += CQ_WAVES) {
    float mx = -INFINITY;
    if (jj < c.N2)
      for (int i = lane; i < c.N1; i += 64) {
        const float mk = m1[i];
        mx = fmaxf(mx, S[i * c.ld + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk));
      }
    mx = wave_max64(mx);
    float sum = 0.f;
    if (jj < c.N2)
      for (int i = lane; i < c.N1; i += 64) {
        const float mk = m1[i];
        sum += __expf(S[i * c.ld + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk) - mx);
      }
    sum = wave_sum64(sum);
    const float inv = 1.0f / sum;
    for (int i = lane; i < c.N1p; i += 64) {
      float v = 0.f;
      if (i < c.N1 && jj < c.N2) {
        const float mk = m1[i];
        v = __expf(S[i * c.ld + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk) - mx) * inv;
      }
      Sc[i * c.ld + jj] = v;
    }
  }
  __syncthreads();
  // ---- save both softmaxes for the backward pass
  const size_t mat = cq_mat_elems(rs.T, rs.L);
  float* gSr = b.SR + ((size_t)dir * rs.B + clip) * mat;
  float* gSc = b.SC + ((size_t)dir * rs.B + clip) * mat;
  for (int idx = threadIdx.x; idx < msz; idx += CQ_THREADS) { gSr[idx] = Sr[idx]; gSc[idx] = Sc[idx]; }
  // ---- c2q = Sr . x2   and   M2 = Sc^T . x1
  float* M2 = b.M2 + ((size_t)dir * rs.B + clip) * cq_m2_rows(rs.T, rs.L) * HUAL_D;
  for (int tile = wave; tile < ni * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
    f32x4 acc = tile_mma<true, false>(Sr, c.ld, c.N1p, X2, HUAL_D, HUAL_D, c.N2p, c.N2p, c.N2, i0, n0, j, g, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * g + r;
      if (i < c.N1) b.C2Q[(size_t)(c.x1base + i) * HUAL_D + n0 + j] = acc[r];
    }
  }
  for (int tile = wave; tile < nj * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;   // rows of M2 = index j of the score
    f32x4 acc = tile_mma<false, false>(Sc, c.ld, c.N2p, X1, HUAL_D, HUAL_D, c.N1p, c.N1p, c.N1, i0, n0, j, g, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) M2[(size_t)(i0 + 4 * g + r) * HUAL_D + n0 + j] = acc[r];
  }
  __syncthreads();
  // ---- q2c = Sr . M2          (= (Sr.Sc^T).x1 of layers.py:127, re-associated)
  for (int tile = wave; tile < ni * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
    f32x4 acc = tile_mma<true, false>(Sr, c.ld, c.N1p, M2, HUAL_D, HUAL_D, c.N2p, c.N2p, c.N2p, i0, n0, j, g, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * g + r;
      if (i < c.N1) b.Q2C[(size_t)(c.x1base + i) * HUAL_D + n0 + j] = acc[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// backward, step 1 (row kernel): split the gradient of [x1, c2q, x1*c2q, x1*q2c]
__global__ __launch_bounds__(256) void cq_bwd_pre_kernel(CqBufs b, CqBwdBufs gb, RowSpace rs) {
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  for (int row = blockIdx.x * 8 + grp; row < rs.R; row += gridDim.x * 8) {
    const size_t off = (size_t)row * HUAL_D + col;
    const float* dc = gb.dCat + (size_t)row * gb.ldcat + col;
    float4 d0 = ld4(dc), d1 = ld4(dc + HUAL_D), d2 = ld4(dc + 2 * HUAL_D), d3 = ld4(dc + 3 * HUAL_D);
    float4 x = ld4(b.X + off), c2q = ld4(b.C2Q + off), q2c = ld4(b.Q2C + off);
    st4(gb.dC2Q + off, make_float4(d1.x + d2.x * x.x, d1.y + d2.y * x.y, d1.z + d2.z * x.z, d1.w + d2.w * x.w));
    st4(gb.dQ2C + off, make_float4(d3.x * x.x, d3.y * x.y, d3.z * x.z, d3.w * x.w));
    st4(gb.dX + off, make_float4(d0.x + d2.x * c2q.x + d3.x * q2c.x, d0.y + d2.y * c2q.y + d3.y * q2c.y,
                                 d0.z + d2.z * c2q.z + d3.z * q2c.z, d0.w + d2.w * c2q.w + d3.w * q2c.w));
  }
}

// backward, step 2 (per clip).  Outputs: dD1W (x1 rows), dD2 (x2 rows), dS0, dS1 and two partial dX:
//   dXa (x1-role rows, via M2)  is ADDED into gb.dX rows of x1;  dXb (x2-role rows, via c2q) goes to dD2's
//   companion buffer - to stay race free between the two directions of a clip it is folded into dD2 itself is
//   not possible (different dropout), so it is written to gb.dC2Q rows?  -> see below: uses dedicated slices.
__global__ __launch_bounds__(CQ_MAX_THREADS) void cq_bwd_kernel(CqBufs b, CqBwdBufs gb, RowSpace rs, float* dXa, float* dXb) {
  extern __shared__ float lds[];
  const int clip = blockIdx.x, dir = blockIdx.y;
  const ClipGeom c = clip_geom(rs, clip, dir);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int msz = c.N1p * c.ld;
  float* Sr = lds;
  float* Sc = lds + msz;
  float* dSr = lds + 2 * msz;    // becomes dscore
  float* dSc = lds + 3 * msz;
  const float* m1 = rs.rowmask + c.x1base;
  const float* m2 = rs.rowmask + c.x2base;
  const float* X1 = b.X + (size_t)c.x1base * HUAL_D;
  const float* X2 = b.X + (size_t)c.x2base * HUAL_D;
  const float* dC2Q = gb.dC2Q + (size_t)c.x1base * HUAL_D;
  const float* dQ2C = gb.dQ2C + (size_t)c.x1base * HUAL_D;
  const size_t mat = cq_mat_elems(rs.T, rs.L);
  const float* gSr = b.SR + ((size_t)dir * rs.B + clip) * mat;
  const float* gSc = b.SC + ((size_t)dir * rs.B + clip) * mat;
  const size_t m2off = ((size_t)dir * rs.B + clip) * cq_m2_rows(rs.T, rs.L) * HUAL_D;
  const float* M2 = b.M2 + m2off;
  float* dM2 = gb.dM2 + m2off;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int nj = c.N2p >> 4, ni = c.N1p >> 4;
  for (int idx = threadIdx.x; idx < msz; idx += CQ_THREADS) { Sr[idx] = gSr[idx]; Sc[idx] = gSc[idx]; dSr[idx] = 0.f; dSc[idx] = 0.f; }
  __syncthreads();
  // dSr = dc2q . x2^T + dq2c . M2^T
  for (int tile = wave; tile < ni * nj; tile += CQ_WAVES) {
    const int i0 = (tile / nj) * 16, n0 = (tile % nj) * 16;
    f32x4 acc = tile_mma<true, true>(dC2Q, HUAL_D, c.N1, X2, HUAL_D, c.N2, HUAL_D, HUAL_D, HUAL_D, i0, n0, j, g, zero);
    acc = tile_mma<true, true>(dQ2C, HUAL_D, c.N1, M2, HUAL_D, c.N2p, HUAL_D, HUAL_D, HUAL_D, i0, n0, j, g, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) dSr[(i0 + 4 * g + r) * c.ld + n0 + j] = ((i0 + 4 * g + r) < c.N1 && (n0 + j) < c.N2) ? acc[r] : 0.f;
  }
  // dM2 = Sr^T . dq2c ;  dXb (x2 rows) = Sr^T . dc2q
  for (int tile = wave; tile < nj * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
    f32x4 acc = tile_mma<false, false>(Sr, c.ld, c.N2p, dQ2C, HUAL_D, HUAL_D, c.N1p, c.N1p, c.N1, i0, n0, j, g, zero);
    f32x4 acc2 = tile_mma<false, false>(Sr, c.ld, c.N2p, dC2Q, HUAL_D, HUAL_D, c.N1p, c.N1p, c.N1, i0, n0, j, g, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = i0 + 4 * g + r;
      dM2[(size_t)jj * HUAL_D + n0 + j] = acc[r];
      if (jj < c.N2) dXb[(size_t)(c.x2base + jj) * HUAL_D + n0 + j] = acc2[r];
    }
  }
  __syncthreads();
  // dSc = x1 . dM2^T ;  dXa (x1 rows) = Sc . dM2
  for (int tile = wave; tile < ni * nj; tile += CQ_WAVES) {
    const int i0 = (tile / nj) * 16, n0 = (tile % nj) * 16;
    f32x4 acc = tile_mma<true, true>(X1, HUAL_D, c.N1, dM2, HUAL_D, c.N2p, HUAL_D, HUAL_D, HUAL_D, i0, n0, j, g, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) dSc[(i0 + 4 * g + r) * c.ld + n0 + j] = ((i0 + 4 * g + r) < c.N1 && (n0 + j) < c.N2) ? acc[r] : 0.f;
  }
  for (int tile = wave; tile < ni * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
    f32x4 acc = tile_mma<true, false>(Sc, c.ld, c.N1p, dM2, HUAL_D, HUAL_D, c.N2p, c.N2p, c.N2p, i0, n0, j, g, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * g + r;
      if (i < c.N1) dXa[(size_t)(c.x1base + i) * HUAL_D + n0 + j] = acc[r];
    }
  }
  __syncthreads();
  // softmax backward -> dscore (in dSr).  mask_logits is multiplicative, so its derivative is the mask.
  for (int i = wave; i < c.N1; i += CQ_WAVES) {
    float dot = 0.f;
    for (int jj = lane; jj < c.N2; jj += 64) dot += Sr[i * c.ld + jj] * dSr[i * c.ld + jj];
    dot = wave_sum64(dot);
    for (int jj = lane; jj < c.N2; jj += 64)
      dSr[i * c.ld + jj] = Sr[i * c.ld + jj] * (dSr[i * c.ld + jj] - dot) * m2[jj];
  }
  __syncthreads();
  for (int jj = wave; jj < c.N2; jj += CQ_WAVES) {
    float dot = 0.f;
    for (int i = lane; i < c.N1; i += 64) dot += Sc[i * c.ld + jj] * dSc[i * c.ld + jj];
    dot = wave_sum64(dot);
    float colsum = 0.f;
    for (int i = lane; i < c.N1; i += 64) {
      const float v = dSr[i * c.ld + jj] + Sc[i * c.ld + jj] * (dSc[i * c.ld + jj] - dot) * m1[i];
      dSr[i * c.ld + jj] = v;
      colsum += v;
    }
    colsum = wave_sum64(colsum);
    if (lane == 0) gb.dS1[c.x2base + jj] = colsum;
  }
  __syncthreads();
  for (int i = wave; i < c.N1; i += CQ_WAVES) {
    float rowsum = 0.f;
    for (int jj = lane; jj < c.N2; jj += 64) rowsum += dSr[i * c.ld + jj];
    rowsum = wave_sum64(rowsum);
    if (lane == 0) gb.dS0[c.x1base + i] = rowsum;
  }
  // dD1W = dscore . d2 ;  dD2 = dscore^T . d1w
  for (int tile = wave; tile < ni * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
    f32x4 acc = tile_mma<true, false>(dSr, c.ld, c.N1p, b.D2 + (size_t)c.x2base * HUAL_D, HUAL_D, HUAL_D, c.N2p, c.N2p, c.N2,
                                      i0, n0, j, g, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * g + r;
      if (i < c.N1) gb.dD1W[(size_t)(c.x1base + i) * HUAL_D + n0 + j] = acc[r];
    }
  }
  for (int tile = wave; tile < nj * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
    f32x4 acc = tile_mma<false, false>(dSr, c.ld, c.N2p, b.D1W + (size_t)c.x1base * HUAL_D, HUAL_D, HUAL_D, c.N1p,
                                       c.N1p, c.N1, i0, n0, j, g, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = i0 + 4 * g + r;
      if (jj < c.N2) gb.dD2[(size_t)(c.x2base + jj) * HUAL_D + n0 + j] = acc[r];
    }
  }
}

// backward, step 3 (row kernel): through the two dropouts and the rank-1 terms; parameter gradients.
// Rows [row_lo,row_hi) must all be video rows or all query rows (the small weights differ per side).
// The first `nvb` workgroups stride over the video rows, the others over the query rows (the small weights differ per side).
__global__ __launch_bounds__(256) void tri_bwd_kernel(CqBufs b, CqBwdBufs gb, CqParams p, CqGrads pg, RowSpace rs,
                                                      DropCfg drop, const float* dXa, const float* dXb, int nvb) {
  __shared__ float4 red[3][8][32];
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  const bool isv = (int)blockIdx.x < nvb;
  const int row_lo = isv ? 0 : rs.Nv, row_hi = isv ? rs.Nv : rs.R;
  const int bid = isv ? (int)blockIdx.x : (int)blockIdx.x - nvb, nblk = isv ? nvb : (int)gridDim.x - nvb;
  const int d1 = isv ? 0 : 1, d2 = isv ? 1 : 0;
  const uint32_t site1 = (uint32_t)HUAL_SITE_TRI + (isv ? 0u : 2u);
  const uint32_t site2 = (uint32_t)HUAL_SITE_TRI + (isv ? 3u : 1u);
  const float4 w0 = ld4(p.w0[d1] + col), wm = ld4(p.wm[d1] + col), w1 = ld4(p.w1[d2] + col);
  float4 gwm = f4zero(), gw0 = f4zero(), gw1 = f4zero();
  for (int row = row_lo + bid * 8 + grp; row < row_hi; row += nblk * 8) {
    const size_t off = (size_t)row * HUAL_D + col;
    float4 x = ld4(b.X + off);
    float4 mk1 = make_float4(1.f, 1.f, 1.f, 1.f), mk2 = mk1;
    if (drop.enabled) {
      mk1 = drop_mask4(drop, site1, (uint32_t)row, (uint32_t)l32);
      mk2 = drop_mask4(drop, site2, (uint32_t)row, (uint32_t)l32);
    }
    const float4 dd1w = ld4(gb.dD1W + off), dd2 = ld4(gb.dD2 + off);
    const float ds0 = gb.dS0[row], ds1 = gb.dS1[row];
    const float4 a = make_float4(x.x * mk1.x, x.y * mk1.y, x.z * mk1.z, x.w * mk1.w);   // dropout(x1)
    const float4 c = ld4(b.D2 + off);                                                   // dropout(x2)
    gwm = make_float4(gwm.x + dd1w.x * a.x, gwm.y + dd1w.y * a.y, gwm.z + dd1w.z * a.z, gwm.w + dd1w.w * a.w);
    gw0 = make_float4(gw0.x + ds0 * a.x, gw0.y + ds0 * a.y, gw0.z + ds0 * a.z, gw0.w + ds0 * a.w);
    gw1 = make_float4(gw1.x + ds1 * c.x, gw1.y + ds1 * c.y, gw1.z + ds1 * c.z, gw1.w + ds1 * c.w);
    float4 dx = ld4(gb.dX + off);
    const float4 xa = ld4(dXa + off), xb = ld4(dXb + off);
    dx.x += xa.x + xb.x + mk1.x * (dd1w.x * wm.x + ds0 * w0.x) + mk2.x * (dd2.x + ds1 * w1.x);
    dx.y += xa.y + xb.y + mk1.y * (dd1w.y * wm.y + ds0 * w0.y) + mk2.y * (dd2.y + ds1 * w1.y);
    dx.z += xa.z + xb.z + mk1.z * (dd1w.z * wm.z + ds0 * w0.z) + mk2.z * (dd2.z + ds1 * w1.z);
    dx.w += xa.w + xb.w + mk1.w * (dd1w.w * wm.w + ds0 * w0.w) + mk2.w * (dd2.w + ds1 * w1.w);
    st4(gb.dX + off, dx);
  }
  red[0][grp][l32] = gwm; red[1][grp][l32] = gw0; red[2][grp][l32] = gw1;
  __syncthreads();
  for (int idx = threadIdx.x; idx < 3 * 128; idx += 256) {
    const int vec = idx >> 7, cc = idx & 127;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += reinterpret_cast<const float*>(&red[vec][k][cc >> 2])[cc & 3];
    float* dst = vec == 0 ? pg.wm[d1] : (vec == 1 ? pg.w0[d1] : pg.w1[d2]);
    atomicAdd(dst + cc, s);
  }
}

namespace hual {

static int cq_threads() {
  static const int t = []() { const char* e = getenv("HUAL_CQ_THREADS"); int v = e ? atoi(e) : 1024; return (v == 256 || v == 512 || v == 1024) ? v : 1024; }();
  return t;
}
static int cq_lds_bytes(const RowSpace& rs, int nmats) { return (int)(cq_mat_elems(rs.T, rs.L) * nmats * sizeof(float)); }

int launch_tri_prep(const CqBufs& b, const CqParams& p, const RowSpace& rs, const DropCfg& drop, hipStream_t s) {
  int g = cdiv(rs.R, 8);
  g = g < 2048 ? g : 2048;
  HUAL_LAUNCH(0.0, 0.0, tri_prep_kernel, dim3(g), dim3(256), 0, s, b, p, rs, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_cq_fwd(const CqBufs& b, const RowSpace& rs, hipStream_t s) {
  const int bytes = cq_lds_bytes(rs, 3);
  HUAL_REQUIRE(bytes <= 160 * 1024, "cq_fwd: T x L score matrix does not fit LDS");
  HUAL_DYN_LDS(cq_fwd_kernel, 160 * 1024);
  HUAL_LAUNCH(2.0 * 8.0 * rs.B * rs.T * rs.L * HUAL_D, 0.0, cq_fwd_kernel, dim3(rs.B, 2), dim3(cq_threads()), bytes, s, b, rs);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_cq_bwd_pre(const CqBufs& b, const CqBwdBufs& g, const RowSpace& rs, hipStream_t s) {
  int n = cdiv(rs.R, 8);
  n = n < 2048 ? n : 2048;
  HUAL_LAUNCH(0.0, 0.0, cq_bwd_pre_kernel, dim3(n), dim3(256), 0, s, b, g, rs);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

// dXa/dXb scratch = g.dC2Q / g.dQ2C can NOT be reused (read by the kernel); callers pass dedicated buffers
int launch_cq_bwd_impl(const CqBufs& b, const CqBwdBufs& g, const RowSpace& rs, float* dXa, float* dXb, hipStream_t s) {
  const int bytes = cq_lds_bytes(rs, 4);
  HUAL_REQUIRE(bytes <= 160 * 1024, "cq_bwd: T x L score matrix does not fit LDS");
  HUAL_DYN_LDS(cq_bwd_kernel, 160 * 1024);
  HUAL_LAUNCH(2.0 * 18.0 * rs.B * rs.T * rs.L * HUAL_D, 0.0, cq_bwd_kernel, dim3(rs.B, 2), dim3(cq_threads()), bytes, s, b, g, rs, dXa, dXb);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_tri_bwd_impl(const CqBufs& b, const CqBwdBufs& g, const CqParams& p, const CqGrads& pg, const RowSpace& rs,
                        const DropCfg& drop, const float* dXa, const float* dXb, hipStream_t s) {
  // every block ends in 384 float atomics on the same 384 addresses (~30 ns each when queued on one address): cap the
  // grid and let the blocks stride over the rows
  int nv = cdiv(rs.Nv, 8);
  nv = nv < 128 ? nv : 128;
  int nq = cdiv(rs.Nq, 8);
  nq = nq < 64 ? nq : 64;
  HUAL_LAUNCH(0.0, 0.0, tri_bwd_kernel, dim3(nv + nq), dim3(256), 0, s, b, g, p, pg, rs, drop, dXa, dXb, nv);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

size_t cq_mat_elems_host(int T, int L) { return cq_mat_elems(T, L); }
size_t cq_m2_rows_host(int T, int L) { return cq_m2_rows(T, L); }

}  // namespace hual
