// Context-query attention kernels (see cq.h).  All small matrix products of one clip run on
// v_mfma_f32_16x16x4_f32 through one device helper (tile_mma) whose operands may sit in LDS or global memory,
// K-contiguous (float4 fragment loads) or K-strided (4 scalar loads) - same fragment maps as gemm.hip/attn.hip.
#include <stdlib.h>
#include "cq.h"
#include "bf16x3.h"
#include "philox.h"
#include "tilecore.h"
#include "prof.h"
#include "cqimg.h"

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
#pragma unroll 8
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

#ifdef HUAL_STAMPS
// debug: clock stamps of the per-clip kernels' phases (scripts/exp/cq_stamps.py); slot 0..15 forward, 16..31 backward
__device__ unsigned long long g_cq_stamps[256 * 32];
extern "C" int hual_debug_cq_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cq_stamps), sizeof(unsigned long long) * (size_t)n);
}
#define CQ_STAMP(i) do { if (threadIdx.x == 0) g_cq_stamps[(blockIdx.y * gridDim.x + blockIdx.x) % 256 * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define CQ_STAMP(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------------
// tri_prep: dropout on both roles of every row + the two rank-1 terms of the trilinear score (ops.py:104-114)
__global__ __launch_bounds__(256) void tri_prep_kernel(CqBufs b, CqParams p, RowSpace rs, DropCfg drop) {
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  const DropRegs dr = drop_load(drop);
  for (int row = blockIdx.x * 8 + grp; row < rs.R; row += gridDim.x * 8) {
    const bool isv = row < rs.Nv;
    const int d1 = isv ? 0 : 1;          // direction in which this row plays x1
    const int d2 = isv ? 1 : 0;          // direction in which this row plays x2
    const uint32_t site1 = (uint32_t)HUAL_SITE_TRI + (isv ? 0u : 2u);
    const uint32_t site2 = (uint32_t)HUAL_SITE_TRI + (isv ? 3u : 1u);
    const size_t off = (size_t)row * HUAL_D + col;
    float4 x = ld4(b.X + off);
    float4 a = x, c = x;
    if (dr.enabled) {      // 16-bit decisions, one call per lane for both roles of the row (tilecore.h)
      uint32_t n1, n2;
      drop_nib2_sites_r(dr, site1, (uint32_t)row, site2, (uint32_t)row, (uint32_t)l32, n1, n2);
      const float4 xs = make_float4(x.x * dr.scale, x.y * dr.scale, x.z * dr.scale, x.w * dr.scale);
      a = f4_select(n1, xs);
      c = f4_select(n2, xs);
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
// glob != 0: the three matrices do not fit LDS (T > 128 with a query of more than 32 words, ..): the scores live in the global scratch
// b.GS and the two softmaxes are written straight into their save buffers - the same code on global pointers (a workgroup's waves share
// the CU's vector L1, which is write-through: __syncthreads() orders one wave's stores before another's loads); only the masks use LDS
__global__ __launch_bounds__(CQ_MAX_THREADS) void cq_fwd_kernel(CqBufs b, RowSpace rs, int glob) {
  extern __shared__ float lds[];
  const int clip = xcd_tile(blockIdx.x, gridDim.x), dir = blockIdx.y;      // XCD-aware clip order (common.h)
  if (clip >= rs.B) return;
  const ClipGeom c = clip_geom(rs, clip, dir);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int msz = c.N1p * c.ld;
  const size_t gslot = ((size_t)dir * rs.B + clip) * cq_mat_elems(rs.T, rs.L);
  float* S = glob ? b.GS + gslot : lds;
  float* Sr = glob ? b.SR + gslot : lds + msz;
  float* Sc = glob ? b.SC + gslot : lds + 2 * msz;
  // the two row masks of the clip, staged once: the softmax passes below read them per element (from global memory every
  // pass paid an L2 round trip)
  float* m1 = glob ? lds : lds + 3 * msz;
  float* m2 = m1 + c.N1p;
  for (int idx = threadIdx.x; idx < c.N1p + c.N2p; idx += CQ_THREADS) {
    const bool first = idx < c.N1p;
    const int k = first ? idx : idx - c.N1p;
    m1[idx] = (k < (first ? c.N1 : c.N2)) ? rs.rowmask[(first ? c.x1base : c.x2base) + k] : 0.f;
  }
  const float* X1 = b.X + (size_t)c.x1base * HUAL_D;
  const float* X2 = b.X + (size_t)c.x2base * HUAL_D;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  CQ_STAMP(0);
  // ---- score = d1w . d2^T + s0 + s1
  const int nj = c.N2p >> 4, ni = c.N1p >> 4;
  for (int tile = wave; tile < ni * nj; tile += CQ_WAVES) {
    const int ti = small_div(tile, nj), i0 = ti * 16, n0 = (tile - ti * nj) * 16;      // (no integer division: common.h)
    // (the rank-1 terms are requested before the product: one memory round trip for the whole tile)
    const float s1 = b.S1[c.x2base + min(n0 + j, c.N2 - 1)];
    float s0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) s0[r] = b.S0[c.x1base + min(i0 + 4 * g + r, c.N1 - 1)];
    f32x4 acc = tile_mma<true, true>(b.D1W + (size_t)c.x1base * HUAL_D, HUAL_D, c.N1, b.D2 + (size_t)c.x2base * HUAL_D,
                                     HUAL_D, c.N2, HUAL_D, HUAL_D, HUAL_D, i0, n0, j, g, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(i0 + 4 * g + r) * c.ld + n0 + j] = acc[r] + s0[r] + s1;
  }
  __syncthreads();
  CQ_STAMP(1);
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
  CQ_STAMP(2);
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
  CQ_STAMP(3);
  // ---- save both softmaxes for the backward pass
  const size_t mat = cq_mat_elems(rs.T, rs.L);
  float* gSr = b.SR + ((size_t)dir * rs.B + clip) * mat;
  float* gSc = b.SC + ((size_t)dir * rs.B + clip) * mat;
  if (!glob)
    for (int idx = threadIdx.x; idx < msz; idx += CQ_THREADS) { gSr[idx] = Sr[idx]; gSc[idx] = Sc[idx]; }
  CQ_STAMP(4);
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
  CQ_STAMP(5);
  for (int tile = wave; tile < nj * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;   // rows of M2 = index j of the score
    f32x4 acc = tile_mma<false, false>(Sc, c.ld, c.N2p, X1, HUAL_D, HUAL_D, c.N1p, c.N1p, c.N1, i0, n0, j, g, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) M2[(size_t)(i0 + 4 * g + r) * HUAL_D + n0 + j] = acc[r];
  }
  __syncthreads();
  CQ_STAMP(6);
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
  CQ_STAMP(7);
}

// ------------------------------------------------------------------------------------------------------

#define CQ_STAGE_MAX 5          // float4 per thread of a 1024-thread workgroup: (N1q + N2q) * 32 / 1024
struct CqRows { float4 v[CQ_STAGE_MAX]; };
__device__ __forceinline__ int cq_r32(int n) { return (n + 31) & ~31; }
// request rows [0, n1) of A (N1q staged rows) and [0, n2) of B (N2q staged rows), zero beyond
__device__ __forceinline__ void cq_rows_load(CqRows& r, const float* A, int n1, int N1q, const float* B, int n2, int N2q) {
#pragma unroll
  for (int u = 0; u < CQ_STAGE_MAX; ++u) {
    const int idx = threadIdx.x + CQ_MAX_THREADS * u, row = idx >> 5, c4 = idx & 31;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < N1q) { if (row < n1) v = ld4(A + (size_t)row * HUAL_D + 4 * c4); }
    else if (row - N1q < n2) v = ld4(B + (size_t)(row - N1q) * HUAL_D + 4 * c4);
    r.v[u] = v;
  }
}
// split the rows into the images of bufA (N1q rows) and bufB (N2q rows; 0: bufB is left alone)
template <int FMT>
__device__ __forceinline__ void cq_rows_store(const CqRows& r, const CqImg& A, int N1q, const CqImg& B, int N2q) {
#pragma unroll
  for (int u = 0; u < CQ_STAGE_MAX; ++u) {
    const int idx = threadIdx.x + CQ_MAX_THREADS * u, row = idx >> 5, c4 = idx & 31;
    if (row < N1q) cq_img_store4<FMT>(A, row, 4 * c4, r.v[u]);
    else if (row - N1q < N2q) cq_img_store4<FMT>(B, row - N1q, 4 * c4, r.v[u]);
  }
}
struct CqLds { int s, sri, sci, m, bufa, bufb, total, N1q, N2q, Sq; };
// byte offsets of the staged kernels' LDS regions: nf32 fp32 score matrices [N1p][ld], two score images, masks / rank-1 terms, bufA, bufB
__host__ __device__ inline CqLds cq_lds_map(int N1, int N2, int nf32) {
  CqLds l;
  const int N1p = (N1 + 15) & ~15, N2p = (N2 + 15) & ~15;
  l.N1q = (N1 + 31) & ~31; l.N2q = (N2 + 31) & ~31;
  l.Sq = l.N1q >= l.N2q ? l.N2q : l.N1q;
  int o = 0;
  l.s = o; o += nf32 * N1p * (N2p + 4) * 4;
  l.sri = o; o += l.Sq * 512;
  l.sci = o; o += l.Sq * 512;
  l.m = o; o += 2 * (N1p + N2p) * 4 + 64;
  o = (o + 15) & ~15;
  l.bufa = o; o += l.N1q * 512;
  l.bufb = o; o += l.N2q * 512;
  l.total = o;
  return l;
}
// (the tri_prep step - dropout on both roles of a row, the rank-1 terms, ops.py:104-114 - happens here on the rows as they are
//  staged: in direction `dir` the x1 rows of the clip get their D1W / S0, the x2 rows their D2 / S1, so over the two directions
//  every row is prepared exactly once in each role; the prepared rows are also written out for the backward pass)
template <bool LONG1>
__device__ __forceinline__ void cq_fwd_staged_body(const CqBufs& b, const CqParams& p, const RowSpace& rs, const DropCfg& drop, int clip, int dir,
                                                   char* lds) {
  const ClipGeom c = clip_geom(rs, clip, dir);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const CqLds L = cq_lds_map(c.N1, c.N2, 1);
  const int N1q = L.N1q, N2q = L.N2q;
  float* S = reinterpret_cast<float*>(lds + L.s);
  const CqImg SrI = cq_img(lds + L.sri, L.Sq, CQ_SCALE_PROB), ScI = cq_img(lds + L.sci, L.Sq, CQ_SCALE_PROB);
  float* m1 = reinterpret_cast<float*>(lds + L.m);            // row masks and rank-1 terms of the clip: m1, s0 [N1p]; m2, s1 [N2p]
  float* m2 = m1 + c.N1p;
  float* s0 = m2 + c.N2p;
  float* s1 = s0 + c.N1p;
  const CqImg bufA = cq_img(lds + L.bufa, N1q, CQ_SCALE_ACT), bufB = cq_img(lds + L.bufb, N2q, CQ_SCALE_ACT);
  const float* X1 = b.X + (size_t)c.x1base * HUAL_D;
  const float* X2 = b.X + (size_t)c.x2base * HUAL_D;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int nj = c.N2p >> 4, ni = c.N1p >> 4;
  CQ_STAMP(0);
  // ---- stage the rows (X1, X2) and the masks: one round trip; D1W / D2 and the rank-1 terms are formed from the rows in registers
  CqRows rows, xrows;
  cq_rows_load(xrows, X1, c.N1, N1q, X2, c.N2, N2q);
  {
    const int idx = threadIdx.x;
    if (idx < c.N1p + c.N2p) {
      const bool first = idx < c.N1p;
      const int k = first ? idx : idx - c.N1p, n = first ? c.N1 : c.N2, base = first ? c.x1base : c.x2base;
      m1[idx] = k < n ? rs.rowmask[base + k] : 0.f;      // (m2 follows m1)
    }
  }
  {
    const uint32_t site1 = (uint32_t)HUAL_SITE_TRI + (dir == 0 ? 0u : 2u), site2 = (uint32_t)HUAL_SITE_TRI + (dir == 0 ? 1u : 3u);
    const int c4 = threadIdx.x & 31;
    const float4 w0 = ld4(p.w0[dir] + 4 * c4), wm = ld4(p.wm[dir] + 4 * c4), w1 = ld4(p.w1[dir] + 4 * c4);
    // dropout: 16-bit decisions, the rows of a thread in pairs - one Philox call per lane and pair (tilecore.h drop_nib2_sites_r)
    const DropRegs dr = drop_load(drop);
    uint32_t nib[CQ_STAGE_MAX + 1];
    auto stage_row = [&](int u, uint32_t& site, uint32_t& grow) {
      const int row = (threadIdx.x + CQ_MAX_THREADS * u) >> 5;
      const bool first = row < N1q;
      const int k = first ? row : row - N1q;
      const bool live = first ? k < c.N1 : k < c.N2;
      site = first ? site1 : site2;
      grow = (uint32_t)((first ? c.x1base : c.x2base) + (live ? k : 0));
    };
    if (dr.enabled) {
#pragma unroll
      for (int u = 0; u < CQ_STAGE_MAX; u += 2) {
        uint32_t sA, rA, sB, rB;
        stage_row(u, sA, rA);
        stage_row(u + 1 < CQ_STAGE_MAX ? u + 1 : u, sB, rB);
        drop_nib2_sites_r(dr, sA, rA, sB, rB, (uint32_t)c4, nib[u], nib[u + 1]);
      }
    }
#pragma unroll
    for (int u = 0; u < CQ_STAGE_MAX; ++u) {
      const int idx = threadIdx.x + CQ_MAX_THREADS * u, row = idx >> 5;
      const bool first = row < N1q;
      const int k = first ? row : row - N1q;
      const bool live = first ? k < c.N1 : k < c.N2;
      const int grow = (first ? c.x1base : c.x2base) + (live ? k : 0);      // unified row (dropout counter, destination)
      float4 x = xrows.v[u];
      if (dr.enabled) x = f4_select(nib[u], make_float4(x.x * dr.scale, x.y * dr.scale, x.z * dr.scale, x.w * dr.scale));
      const float4 w = first ? w0 : w1;
      const float sv = half_sum32(x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w);
      const float4 o = first ? make_float4(x.x * wm.x, x.y * wm.y, x.z * wm.z, x.w * wm.w) : x;
      rows.v[u] = o;
      if (row < N1q + N2q && k < (first ? c.N1p : c.N2p)) {
        if (c4 == 0) (first ? s0 : s1)[k] = live ? sv : 0.f;
        if (live) {
          st4((first ? b.D1W : b.D2) + (size_t)grow * HUAL_D + 4 * c4, o);
          if (c4 == 0) (first ? b.S0 : b.S1)[grow] = sv;
        }
      }
    }
  }
  cq_rows_store<1>(rows, bufA, N1q, bufB, N2q);
  cq_barrier();
  CQ_STAMP(1);
  // ---- score = d1w . d2^T + s0 + s1
  for (int tile = wave; tile < ni * nj; tile += CQ_WAVES) {
    const int ti = small_div(tile, nj), i0 = ti * 16, n0 = (tile - ti * nj) * 16;      // (no integer division: common.h)
    const f32x4 acc = cq_mma<false, false, 1>(bufA, i0, bufB, n0, HUAL_D, lane, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) S[__mul24(i0 + 4 * g + r, c.ld) + n0 + j] = acc[r] + s0[i0 + 4 * g + r] + s1[n0 + j];
  }
  cq_barrier();
  CQ_STAMP(2);
  cq_rows_store<1>(xrows, bufA, N1q, bufB, N2q);      // X1, X2
  CQ_STAMP(8);
  // ---- row softmax over j with mask2 (layers.py:122-123); zero outside the valid block (the images are read up to N1q x N2q).
  // Rows of at most 32 columns go two per wave (a 32-lane half each)
  if (N2q <= 32) {
    const int hh = lane >> 5, l32 = lane & 31;
    for (int i = 2 * wave + hh; i < N1q; i += 2 * CQ_WAVES) {
      const bool ok = i < c.N1 && l32 < c.N2;
      float lg = -INFINITY;
      if (ok) {
        const float mk = m2[l32];
        lg = S[__mul24(i, c.ld) + l32] * mk + HUAL_MASK_VALUE * (1.0f - mk);
      }
      const float mx = half_max32(lg);
      const float e = ok ? __expf(lg - mx) : 0.f;
      const float inv = 1.0f / half_sum32(e);
      cq_sc_store<LONG1, 1>(SrI, i, l32, ok ? e * inv : 0.f);
    }
  } else {
    for (int i = wave; i < N1q; i += CQ_WAVES) {
      float mx = -INFINITY;
      if (i < c.N1)
        for (int jj = lane; jj < c.N2; jj += 64) {
          const float mk = m2[jj];
          mx = fmaxf(mx, S[__mul24(i, c.ld) + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk));
        }
      mx = wave_max64(mx);
      float sum = 0.f;
      if (i < c.N1)
        for (int jj = lane; jj < c.N2; jj += 64) {
          const float mk = m2[jj];
          sum += __expf(S[__mul24(i, c.ld) + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk) - mx);
        }
      sum = wave_sum64(sum);
      const float inv = 1.0f / sum;
      for (int jj = lane; jj < N2q; jj += 64) {
        float v = 0.f;
        if (i < c.N1 && jj < c.N2) {
          const float mk = m2[jj];
          v = __expf(S[__mul24(i, c.ld) + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk) - mx) * inv;
        }
        cq_sc_store<LONG1, 1>(SrI, i, jj, v);
      }
    }
  }
  CQ_STAMP(9);
  // ---- column softmax over i with mask1 (layers.py:124-125); columns of at most 32 rows go two per wave
  if (N1q <= 32) {
    const int hh = lane >> 5, l32 = lane & 31;
    for (int jj = 2 * wave + hh; jj < N2q; jj += 2 * CQ_WAVES) {
      const bool ok = jj < c.N2 && l32 < c.N1;
      float lg = -INFINITY;
      if (ok) {
        const float mk = m1[l32];
        lg = S[__mul24(l32, c.ld) + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk);
      }
      const float mx = half_max32(lg);
      const float e = ok ? __expf(lg - mx) : 0.f;
      const float inv = 1.0f / half_sum32(e);
      cq_sc_store<LONG1, 1>(ScI, l32, jj, ok ? e * inv : 0.f);
    }
  } else {
    for (int jj = wave; jj < N2q; jj += CQ_WAVES) {
      float mx = -INFINITY;
      if (jj < c.N2)
        for (int i = lane; i < c.N1; i += 64) {
          const float mk = m1[i];
          mx = fmaxf(mx, S[__mul24(i, c.ld) + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk));
        }
      mx = wave_max64(mx);
      float sum = 0.f;
      if (jj < c.N2)
        for (int i = lane; i < c.N1; i += 64) {
          const float mk = m1[i];
          sum += __expf(S[__mul24(i, c.ld) + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk) - mx);
        }
      sum = wave_sum64(sum);
      const float inv = 1.0f / sum;
      for (int i = lane; i < N1q; i += 64) {
        float v = 0.f;
        if (i < c.N1 && jj < c.N2) {
          const float mk = m1[i];
          v = __expf(S[__mul24(i, c.ld) + jj] * mk + HUAL_MASK_VALUE * (1.0f - mk) - mx) * inv;
        }
        cq_sc_store<LONG1, 1>(ScI, i, jj, v);
      }
    }
  }
  CQ_STAMP(10);
  cq_barrier();
  CQ_STAMP(3);
  // ---- save both softmaxes for the backward pass: the images as they stand (both planes)
  {
    const size_t mat = cq_mat_elems(rs.T, rs.L);
    float4* gSr = reinterpret_cast<float4*>(b.SR + ((size_t)dir * rs.B + clip) * mat);
    float4* gSc = reinterpret_cast<float4*>(b.SC + ((size_t)dir * rs.B + clip) * mat);
    const float4* lr = reinterpret_cast<const float4*>(SrI.p);
    const float4* lc = reinterpret_cast<const float4*>(ScI.p);
    for (int idx = threadIdx.x; idx < L.Sq * 32; idx += CQ_THREADS) { gSr[idx] = lr[idx]; gSc[idx] = lc[idx]; }
  }
  CQ_STAMP(4);
  // ---- c2q = Sr . x2   and   M2 = Sc^T . x1 (kept in registers until every wave is done with X2)
  float* M2 = b.M2 + ((size_t)dir * rs.B + clip) * cq_m2_rows(rs.T, rs.L) * HUAL_D;
  for (int tile = wave; tile < ni * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
    const f32x4 acc = cq_mma<LONG1, true, 1>(SrI, i0, bufB, n0, N2q, lane, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * g + r;
      if (i < c.N1) b.C2Q[(size_t)(c.x1base + i) * HUAL_D + n0 + j] = acc[r];
    }
  }
  CQ_STAMP(5);
  const int njq = N2q >> 4;              // M2 rows up to N2q (zero beyond N2: Sc is zero there), the contraction length of q2c
  f32x4 m2acc[4];                        // njq * 8 <= 64 tiles over 16 waves
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int tile = wave + CQ_WAVES * q;
    m2acc[q] = zero;
    if (tile < njq * 8) {
      const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;   // rows of M2 = index j of the score
      m2acc[q] = cq_mma<!LONG1, true, 1>(ScI, i0, bufA, n0, N1q, lane, zero);
    }
  }
  cq_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int tile = wave + CQ_WAVES * q;
    if (tile < njq * 8) {
      const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (i0 + 4 * g + r < c.N2p) M2[(size_t)(i0 + 4 * g + r) * HUAL_D + n0 + j] = m2acc[q][r];
        cq_img_store1<1>(bufB, i0 + 4 * g + r, n0 + j, m2acc[q][r]);
      }
    }
  }
  cq_barrier();
  CQ_STAMP(6);
  // ---- q2c = Sr . M2          (= (Sr.Sc^T).x1 of layers.py:127, re-associated)
  for (int tile = wave; tile < ni * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
    const f32x4 acc = cq_mma<LONG1, true, 1>(SrI, i0, bufB, n0, N2q, lane, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * g + r;
      if (i < c.N1) b.Q2C[(size_t)(c.x1base + i) * HUAL_D + n0 + j] = acc[r];
    }
  }
  CQ_STAMP(7);
}

__global__ __launch_bounds__(CQ_MAX_THREADS) void cq_fwd_staged_kernel(CqBufs b, CqParams p, RowSpace rs, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char cq_lds[];
  const int clip = xcd_tile(blockIdx.x, gridDim.x), dir = blockIdx.y;      // XCD-aware clip order (common.h)
  if (clip >= rs.B) return;
  const int N1 = dir == 0 ? rs.T : rs.L, N2 = dir == 0 ? rs.L : rs.T;
  if (cq_r32(N1) >= cq_r32(N2)) cq_fwd_staged_body<true>(b, p, rs, drop, clip, dir, cq_lds);
  else cq_fwd_staged_body<false>(b, p, rs, drop, clip, dir, cq_lds);
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
// (glob != 0: the saved softmaxes are read where they are, the two gradient matrices live in the global scratch gb.GD - see cq_fwd_kernel)
__global__ __launch_bounds__(CQ_MAX_THREADS) void cq_bwd_kernel(CqBufs b, CqBwdBufs gb, RowSpace rs, float* dXa, float* dXb, int glob) {
  extern __shared__ float lds[];
  const int clip = xcd_tile(blockIdx.x, gridDim.x), dir = blockIdx.y;      // XCD-aware clip order (common.h)
  if (clip >= rs.B) return;
  const ClipGeom c = clip_geom(rs, clip, dir);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int msz = c.N1p * c.ld;
  const size_t gslot = ((size_t)dir * rs.B + clip) * cq_mat_elems(rs.T, rs.L);
  float* Sr = glob ? b.SR + gslot : lds;
  float* Sc = glob ? b.SC + gslot : lds + msz;
  float* dSr = glob ? gb.GD + 2 * gslot : lds + 2 * msz;    // becomes dscore
  float* dSc = glob ? gb.GD + 2 * gslot + cq_mat_elems(rs.T, rs.L) : lds + 3 * msz;
  float* m1 = glob ? lds : lds + 4 * msz;     // the two row masks of the clip (see cq_fwd_kernel)
  float* m2 = m1 + c.N1p;
  for (int idx = threadIdx.x; idx < c.N1p + c.N2p; idx += CQ_THREADS) {
    const bool first = idx < c.N1p;
    const int k = first ? idx : idx - c.N1p;
    m1[idx] = (k < (first ? c.N1 : c.N2)) ? rs.rowmask[(first ? c.x1base : c.x2base) + k] : 0.f;
  }
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
  CQ_STAMP(16);
  if (glob) { for (int idx = threadIdx.x; idx < msz; idx += CQ_THREADS) { dSr[idx] = 0.f; dSc[idx] = 0.f; } }
  else for (int idx = threadIdx.x; idx < msz; idx += CQ_THREADS) { Sr[idx] = gSr[idx]; Sc[idx] = gSc[idx]; dSr[idx] = 0.f; dSc[idx] = 0.f; }
  __syncthreads();
  CQ_STAMP(17);
  // dSr = dc2q . x2^T + dq2c . M2^T
  for (int tile = wave; tile < ni * nj; tile += CQ_WAVES) {
    const int ti = small_div(tile, nj), i0 = ti * 16, n0 = (tile - ti * nj) * 16;      // (no integer division: common.h)
    f32x4 acc = tile_mma<true, true>(dC2Q, HUAL_D, c.N1, X2, HUAL_D, c.N2, HUAL_D, HUAL_D, HUAL_D, i0, n0, j, g, zero);
    acc = tile_mma<true, true>(dQ2C, HUAL_D, c.N1, M2, HUAL_D, c.N2p, HUAL_D, HUAL_D, HUAL_D, i0, n0, j, g, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) dSr[__mul24(i0 + 4 * g + r, c.ld) + n0 + j] = ((i0 + 4 * g + r) < c.N1 && (n0 + j) < c.N2) ? acc[r] : 0.f;
  }
  CQ_STAMP(18);
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
  CQ_STAMP(19);
  // dSc = x1 . dM2^T ;  dXa (x1 rows) = Sc . dM2
  for (int tile = wave; tile < ni * nj; tile += CQ_WAVES) {
    const int ti = small_div(tile, nj), i0 = ti * 16, n0 = (tile - ti * nj) * 16;      // (no integer division: common.h)
    f32x4 acc = tile_mma<true, true>(X1, HUAL_D, c.N1, dM2, HUAL_D, c.N2p, HUAL_D, HUAL_D, HUAL_D, i0, n0, j, g, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) dSc[__mul24(i0 + 4 * g + r, c.ld) + n0 + j] = ((i0 + 4 * g + r) < c.N1 && (n0 + j) < c.N2) ? acc[r] : 0.f;
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
  CQ_STAMP(20);
  // softmax backward -> dscore (in dSr).  mask_logits is multiplicative, so its derivative is the mask.
  for (int i = wave; i < c.N1; i += CQ_WAVES) {
    float dot = 0.f;
    for (int jj = lane; jj < c.N2; jj += 64) dot += Sr[__mul24(i, c.ld) + jj] * dSr[__mul24(i, c.ld) + jj];
    dot = wave_sum64(dot);
    for (int jj = lane; jj < c.N2; jj += 64)
      dSr[__mul24(i, c.ld) + jj] = Sr[__mul24(i, c.ld) + jj] * (dSr[__mul24(i, c.ld) + jj] - dot) * m2[jj];
  }
  __syncthreads();
  for (int jj = wave; jj < c.N2; jj += CQ_WAVES) {
    float dot = 0.f;
    for (int i = lane; i < c.N1; i += 64) dot += Sc[__mul24(i, c.ld) + jj] * dSc[__mul24(i, c.ld) + jj];
    dot = wave_sum64(dot);
    float colsum = 0.f;
    for (int i = lane; i < c.N1; i += 64) {
      const float v = dSr[__mul24(i, c.ld) + jj] + Sc[__mul24(i, c.ld) + jj] * (dSc[__mul24(i, c.ld) + jj] - dot) * m1[i];
      dSr[__mul24(i, c.ld) + jj] = v;
      colsum += v;
    }
    colsum = wave_sum64(colsum);
    if (lane == 0) gb.dS1[c.x2base + jj] = colsum;
  }
  __syncthreads();
  CQ_STAMP(21);
  for (int i = wave; i < c.N1; i += CQ_WAVES) {
    float rowsum = 0.f;
    for (int jj = lane; jj < c.N2; jj += 64) rowsum += dSr[__mul24(i, c.ld) + jj];
    rowsum = wave_sum64(rowsum);
    if (lane == 0) gb.dS0[c.x1base + i] = rowsum;
  }
  CQ_STAMP(22);
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
  CQ_STAMP(23);
}

// cq_bwd_staged_kernel: cq_bwd_kernel with the [rows,128] operands of every product staged in LDS as split images and requested
// one phase ahead (see cq_fwd_staged_body).   bufA (x1 rows): dC2Q, dQ2C, X1, D1W      bufB (x2 rows): X2, M2, dM2, D2
// The saved softmaxes arrive as the forward's images and are used as they are; dscore takes the place of the Sr image once both softmax
// backward passes are done.
// Round 5: EVERY image is an fp16 pair (FMT 1, 22 significant bits; round 4 ran the backward on bf16 pairs - 16 bits - and its 2^-16 per
// product was the largest single source of the 2-4e-5 gradient noise that every tensor upstream of this block carried).  Activations
// (X1, X2, M2, D1W, D2) use the forward's fixed scale 2^4, probabilities 2^10; the four GRADIENT images (dC2Q, dQ2C, dM2, dscore) take a
// power-of-two scale from their own largest element (cq_img_autoscale: one workgroup maximum each - two of them ride on barriers that
// were there already, the first two share one extra barrier).
__device__ __forceinline__ float cq_rows_absmax_a(const CqRows& r, int N1q) {      // largest |element| of the bufA part of staged rows
  float m = 0.f;
#pragma unroll
  for (int u = 0; u < CQ_STAGE_MAX; ++u) {
    const int idx = threadIdx.x + CQ_MAX_THREADS * u, row = idx >> 5;
    if (row < N1q) m = fmaxf(m, f4absmax(r.v[u]));
  }
  return m;
}
template <bool LONG1>
__device__ __forceinline__ void cq_bwd_staged_body(const CqBufs& b, const CqBwdBufs& gb, const RowSpace& rs, float* dXa, float* dXb, int clip, int dir,
                                                   char* lds) {
  const ClipGeom c = clip_geom(rs, clip, dir);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const CqLds L = cq_lds_map(c.N1, c.N2, 2);
  const int N1q = L.N1q, N2q = L.N2q;
  const int msz = c.N1p * c.ld;
  float* dSr = reinterpret_cast<float*>(lds + L.s);      // becomes dscore
  float* dSc = dSr + msz;
  const CqImg SrI = cq_img(lds + L.sri, L.Sq, CQ_SCALE_PROB), ScI = cq_img(lds + L.sci, L.Sq, CQ_SCALE_PROB);
  float* m1 = reinterpret_cast<float*>(lds + L.m);
  float* m2 = m1 + c.N1p;
  float* mx0 = m1 + c.N1p + c.N2p;                       // two rows of 16 per-wave maxima (cq_wgmax_put / _get): >= 48 floats are free here
  float* mx1 = mx0 + 16;
  CqImg bufA = cq_img(lds + L.bufa, N1q, CQ_SCALE_ACT), bufB = cq_img(lds + L.bufb, N2q, CQ_SCALE_ACT);
  const size_t x1off = (size_t)c.x1base * HUAL_D, x2off = (size_t)c.x2base * HUAL_D;
  const size_t mat = cq_mat_elems(rs.T, rs.L);
  const float4* gSr = reinterpret_cast<const float4*>(b.SR + ((size_t)dir * rs.B + clip) * mat);
  const float4* gSc = reinterpret_cast<const float4*>(b.SC + ((size_t)dir * rs.B + clip) * mat);
  const size_t m2off = ((size_t)dir * rs.B + clip) * cq_m2_rows(rs.T, rs.L) * HUAL_D;
  float* dM2 = gb.dM2 + m2off;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int nj = c.N2p >> 4, ni = c.N1p >> 4, njq = N2q >> 4;
  CQ_STAMP(16);
  // ---- requests of the first two operand pairs, the saved softmaxes and the masks: one round trip
  CqRows r1, r2;
  cq_rows_load(r1, gb.dC2Q + x1off, c.N1, N1q, b.X + x2off, c.N2, N2q);
  cq_rows_load(r2, gb.dQ2C + x1off, c.N1, N1q, b.M2 + m2off, c.N2p, N2q);
  if ((int)threadIdx.x < c.N1p + c.N2p) {
    const int idx = threadIdx.x;
    const bool first = idx < c.N1p;
    const int k = first ? idx : idx - c.N1p;
    m1[idx] = (k < (first ? c.N1 : c.N2)) ? rs.rowmask[(first ? c.x1base : c.x2base) + k] : 0.f;     // (m2 follows m1)
  }
  {
    // the forward saved fp16 pairs of p * 2^10 (both planes, Sq x 256 bytes each) in the very layout of the images: plain copies
    const int nch = L.Sq * 32;                         // 16-byte chunks of both planes
    uint4* lr = reinterpret_cast<uint4*>(SrI.p);
    uint4* lc = reinterpret_cast<uint4*>(ScI.p);
    for (int idx = threadIdx.x; idx < nch; idx += CQ_THREADS) {
      lr[idx] = __builtin_bit_cast(uint4, gSr[idx]);
      lc[idx] = __builtin_bit_cast(uint4, gSc[idx]);
    }
  }
  cq_wgmax_put(mx0, cq_rows_absmax_a(r1, N1q));          // max |dC2Q|, max |dQ2C| of the clip
  cq_wgmax_put(mx1, cq_rows_absmax_a(r2, N1q));
  cq_barrier();
  CqImg bufA2 = bufA;                                    // the dQ2C image (same bytes as bufA, its own scale)
  cq_img_autoscale(bufA, cq_wgmax_get(mx0));
  cq_img_autoscale(bufA2, cq_wgmax_get(mx1));
  cq_rows_store<1>(r1, bufA, N1q, bufB, N2q);            // dC2Q, X2
  cq_barrier();
  CQ_STAMP(17);
  // ---- first half of dSr = dc2q . x2^T (+ dq2c . M2^T below) ;  dXb (x2 rows) = Sr^T . dc2q
  cq_rows_load(r1, b.X + x1off, c.N1, N1q, nullptr, 0, N2q);      // X1, for the dSc product
  f32x4 sacc[2] = {zero, zero};                          // ni * nj <= 25 tiles over 16 waves
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int tile = wave + CQ_WAVES * q;
    if (tile < ni * nj) {
      const int ti = small_div(tile, nj), i0 = ti * 16, n0 = (tile - ti * nj) * 16;      // (no integer division: common.h)
      sacc[q] = cq_mma<false, false, 1>(bufA, i0, bufB, n0, HUAL_D, lane, zero);
    }
  }
  for (int tile = wave; tile < nj * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
    const f32x4 acc2 = cq_mma<!LONG1, true, 1>(SrI, i0, bufA, n0, N1q, lane, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = i0 + 4 * g + r;
      if (jj < c.N2) dXb[(size_t)(c.x2base + jj) * HUAL_D + n0 + j] = acc2[r];
    }
  }
  cq_barrier();
  CQ_STAMP(18);
  cq_rows_store<1>(r2, bufA2, N1q, bufB, N2q);         // dQ2C, M2
  cq_barrier();
  cq_rows_load(r2, b.D1W + x1off, c.N1, N1q, b.D2 + x2off, c.N2, N2q);     // for the last two products
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int tile = wave + CQ_WAVES * q;
    if (tile < ni * nj) {
      const int ti = small_div(tile, nj), i0 = ti * 16, n0 = (tile - ti * nj) * 16;      // (no integer division: common.h)
      const f32x4 acc = cq_mma<false, false, 1>(bufA2, i0, bufB, n0, HUAL_D, lane, sacc[q]);
#pragma unroll
      for (int r = 0; r < 4; ++r) dSr[__mul24(i0 + 4 * g + r, c.ld) + n0 + j] = ((i0 + 4 * g + r) < c.N1 && (n0 + j) < c.N2) ? acc[r] : 0.f;
    }
  }
  // dM2 = Sr^T . dq2c (kept in registers until every wave is done with M2); rows up to N2q: zero beyond N2 (Sr is zero there)
  f32x4 macc[4];                                          // njq * 8 <= 64 tiles over 16 waves
  float mmax = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int tile = wave + CQ_WAVES * q;
    macc[q] = zero;
    if (tile < njq * 8) {
      const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
      macc[q] = cq_mma<!LONG1, true, 1>(SrI, i0, bufA2, n0, N1q, lane, zero);
      mmax = fmaxf(mmax, fmaxf(fmaxf(fabsf(macc[q][0]), fabsf(macc[q][1])), fmaxf(fabsf(macc[q][2]), fabsf(macc[q][3]))));
    }
  }
  cq_wgmax_put(mx0, mmax);                               // (mx0 was read behind the first barrier: four barriers ago)
  cq_barrier();
  CQ_STAMP(19);
  bufA.scale = CQ_SCALE_ACT; bufA.inv = 1.0f / CQ_SCALE_ACT;
  CqImg bufB2 = bufB;                                    // the dM2 image
  cq_img_autoscale(bufB2, cq_wgmax_get(mx0));
  cq_rows_store<1>(r1, bufA, N1q, bufB, 0);                 // X1 (bufB receives dM2 below)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int tile = wave + CQ_WAVES * q;
    if (tile < njq * 8) {
      const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (i0 + 4 * g + r < c.N2p) dM2[(size_t)(i0 + 4 * g + r) * HUAL_D + n0 + j] = macc[q][r];
        cq_img_store1<1>(bufB2, i0 + 4 * g + r, n0 + j, macc[q][r]);
      }
    }
  }
  cq_barrier();
  // ---- dSc = x1 . dM2^T ;  dXa (x1 rows) = Sc . dM2
  for (int tile = wave; tile < ni * nj; tile += CQ_WAVES) {
    const int ti = small_div(tile, nj), i0 = ti * 16, n0 = (tile - ti * nj) * 16;      // (no integer division: common.h)
    const f32x4 acc = cq_mma<false, false, 1>(bufA, i0, bufB2, n0, HUAL_D, lane, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) dSc[__mul24(i0 + 4 * g + r, c.ld) + n0 + j] = ((i0 + 4 * g + r) < c.N1 && (n0 + j) < c.N2) ? acc[r] : 0.f;
  }
  for (int tile = wave; tile < ni * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
    const f32x4 acc = cq_mma<LONG1, true, 1>(ScI, i0, bufB2, n0, N2q, lane, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * g + r;
      if (i < c.N1) dXa[(size_t)(c.x1base + i) * HUAL_D + n0 + j] = acc[r];
    }
  }
  cq_barrier();
  CQ_STAMP(20);
  cq_rows_store<1>(r2, bufA, N1q, bufB, N2q);          // D1W, D2 (activation scale on both)
  // ---- softmax backward -> dscore (in dSr).  mask_logits is multiplicative, so its derivative is the mask.
  // (rows / columns of at most 32 elements go two per wave, as in the forward kernel)
  const int hh = lane >> 5, l32 = lane & 31;
  if (c.N2p <= 32) {
    for (int i = 2 * wave + hh; i < c.N1p; i += 2 * CQ_WAVES) {
      const bool ok = i < c.N1 && l32 < c.N2;
      const float sr = ok ? cq_sc_load<LONG1, 1>(SrI, i, l32) : 0.f, ds = ok ? dSr[__mul24(i, c.ld) + l32] : 0.f;
      const float dot = half_sum32(sr * ds);
      if (ok) dSr[__mul24(i, c.ld) + l32] = sr * (ds - dot) * m2[l32];
    }
  } else {
    for (int i = wave; i < c.N1; i += CQ_WAVES) {
      float dot = 0.f;
      for (int jj = lane; jj < c.N2; jj += 64) dot += cq_sc_load<LONG1, 1>(SrI, i, jj) * dSr[__mul24(i, c.ld) + jj];
      dot = wave_sum64(dot);
      for (int jj = lane; jj < c.N2; jj += 64)
        dSr[__mul24(i, c.ld) + jj] = cq_sc_load<LONG1, 1>(SrI, i, jj) * (dSr[__mul24(i, c.ld) + jj] - dot) * m2[jj];
    }
  }
  cq_barrier();
  float dmax = 0.f;                                       // largest |dscore| this thread produced
  if (c.N1p <= 32) {
    for (int jj = 2 * wave + hh; jj < c.N2p; jj += 2 * CQ_WAVES) {
      const bool ok = jj < c.N2 && l32 < c.N1;
      const float sc = ok ? cq_sc_load<LONG1, 1>(ScI, l32, jj) : 0.f, ds = ok ? dSc[__mul24(l32, c.ld) + jj] : 0.f;
      const float dot = half_sum32(sc * ds);
      float v = 0.f;
      if (ok) {
        v = dSr[__mul24(l32, c.ld) + jj] + sc * (ds - dot) * m1[l32];
        dSr[__mul24(l32, c.ld) + jj] = v;
      }
      dmax = fmaxf(dmax, fabsf(v));
      const float colsum = half_sum32(v);
      if (l32 == 0 && jj < c.N2) gb.dS1[c.x2base + jj] = colsum;
    }
  } else {
    for (int jj = wave; jj < c.N2; jj += CQ_WAVES) {
      float dot = 0.f;
      for (int i = lane; i < c.N1; i += 64) dot += cq_sc_load<LONG1, 1>(ScI, i, jj) * dSc[__mul24(i, c.ld) + jj];
      dot = wave_sum64(dot);
      float colsum = 0.f;
      for (int i = lane; i < c.N1; i += 64) {
        const float v = dSr[__mul24(i, c.ld) + jj] + cq_sc_load<LONG1, 1>(ScI, i, jj) * (dSc[__mul24(i, c.ld) + jj] - dot) * m1[i];
        dSr[__mul24(i, c.ld) + jj] = v;
        dmax = fmaxf(dmax, fabsf(v));
        colsum += v;
      }
      colsum = wave_sum64(colsum);
      if (lane == 0) gb.dS1[c.x2base + jj] = colsum;
    }
  }
  cq_wgmax_put(mx1, dmax);
  cq_barrier();
  CQ_STAMP(21);
  // row sums (d s0) and the dscore image (in place of the Sr image: both softmaxes are done with it)
  CqImg dscI = SrI;
  cq_img_autoscale(dscI, cq_wgmax_get(mx1));
  if (c.N2p <= 32) {
    for (int i = 2 * wave + hh; i < N1q; i += 2 * CQ_WAVES) {
      const float v = (i < c.N1 && l32 < c.N2) ? dSr[__mul24(i, c.ld) + l32] : 0.f;
      const float rowsum = half_sum32(v);
      if (l32 == 0 && i < c.N1) gb.dS0[c.x1base + i] = rowsum;
      cq_sc_store<LONG1, 1>(dscI, i, l32, v);
    }
  } else {
    for (int i = wave; i < N1q; i += CQ_WAVES) {
      float rowsum = 0.f;
      for (int jj = lane; jj < N2q; jj += 64) {
        const float v = (i < c.N1 && jj < c.N2) ? dSr[__mul24(i, c.ld) + jj] : 0.f;
        rowsum += v;
        cq_sc_store<LONG1, 1>(dscI, i, jj, v);
      }
      rowsum = wave_sum64(rowsum);
      if (lane == 0 && i < c.N1) gb.dS0[c.x1base + i] = rowsum;
    }
  }
  cq_barrier();
  CQ_STAMP(22);
  // ---- dD1W = dscore . d2 ;  dD2 = dscore^T . d1w
  for (int tile = wave; tile < ni * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
    const f32x4 acc = cq_mma<LONG1, true, 1>(dscI, i0, bufB, n0, N2q, lane, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * g + r;
      if (i < c.N1) gb.dD1W[(size_t)(c.x1base + i) * HUAL_D + n0 + j] = acc[r];
    }
  }
  for (int tile = wave; tile < nj * 8; tile += CQ_WAVES) {
    const int i0 = (tile >> 3) * 16, n0 = (tile & 7) * 16;
    const f32x4 acc = cq_mma<!LONG1, true, 1>(dscI, i0, bufA, n0, N1q, lane, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = i0 + 4 * g + r;
      if (jj < c.N2) gb.dD2[(size_t)(c.x2base + jj) * HUAL_D + n0 + j] = acc[r];
    }
  }
  CQ_STAMP(23);
}

__global__ __launch_bounds__(CQ_MAX_THREADS) void cq_bwd_staged_kernel(CqBufs b, CqBwdBufs gb, RowSpace rs, float* dXa, float* dXb) {
  extern __shared__ __attribute__((aligned(16))) char cq_lds[];
  const int clip = xcd_tile(blockIdx.x, gridDim.x), dir = blockIdx.y;      // XCD-aware clip order (common.h)
  if (clip >= rs.B) return;
  const int N1 = dir == 0 ? rs.T : rs.L, N2 = dir == 0 ? rs.L : rs.T;
  if (cq_r32(N1) >= cq_r32(N2)) cq_bwd_staged_body<true>(b, gb, rs, dXa, dXb, clip, dir, cq_lds);
  else cq_bwd_staged_body<false>(b, gb, rs, dXa, dXb, clip, dir, cq_lds);
}

// backward, step 3 (row kernel): through the two dropouts and the rank-1 terms; parameter gradients.
// The first `nvb` workgroups take the video rows, the others the query rows (the small weights differ per side), 8 rows
// each: one memory round trip per workgroup (a grid-stride loop paid one per iteration, ~2 us each).  The parameter sums
// leave as per-workgroup partials.
__global__ __launch_bounds__(256) void tri_bwd_kernel(CqBufs b, CqBwdBufs gb, CqParams p, float* part, RowSpace rs,
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
  const DropRegs dr = drop_load(drop);
  float4 gwm = f4zero(), gw0 = f4zero(), gw1 = f4zero();
  for (int row = row_lo + bid * 8 + grp; row < row_hi; row += nblk * 8) {
    const size_t off = (size_t)row * HUAL_D + col;
    float4 x = ld4(b.X + off);
    float4 mk1 = make_float4(1.f, 1.f, 1.f, 1.f), mk2 = mk1;
    if (dr.enabled) {
      uint32_t n1, n2;
      drop_nib2_sites_r(dr, site1, (uint32_t)row, site2, (uint32_t)row, (uint32_t)l32, n1, n2);
      mk1 = mask_from_bits4(n1, dr.scale);
      mk2 = mask_from_bits4(n2, dr.scale);
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
    part[((size_t)blockIdx.x * 3 + vec) * HUAL_D + cc] = s;      // [block][wm, w0, w1][128], folded by colsum_kernel
  }
}

namespace hual {

static int cq_threads() { return 1024; }
static int cq_lds_bytes(const RowSpace& rs, int nmats) {      // the matrices + the two staged row masks
  return (int)((cq_mat_elems(rs.T, rs.L) * nmats + cq_padded(rs.T) + cq_padded(rs.L)) * sizeof(float));
}

int launch_tri_prep(const CqBufs& b, const CqParams& p, const RowSpace& rs, const DropCfg& drop, hipStream_t s) {
  int g = cdiv(rs.R, 8);
  g = g < 2048 ? g : 2048;
  HUAL_LAUNCH(0.0, 0.0, tri_prep_kernel, dim3(g), dim3(256), 0, s, b, p, rs, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

// LDS of the staged kernels (cq_lds_map): the larger of the two directions; both kernels must fit (the forward saves its softmaxes in
// the image layout only the staged backward reads)
static int cq_staged_bytes(const RowSpace& rs, int nf32) {
  const int a = cq_lds_map(rs.T, rs.L, nf32).total, c = cq_lds_map(rs.L, rs.T, nf32).total;
  return a > c ? a : c;
}
static bool cq_staged_ok(const RowSpace& rs) {
  const int Tq = (rs.T + 31) & ~31, Lq = (rs.L + 31) & ~31;
  return Tq <= 128 && Lq <= 128 && (Tq + Lq) * 32 <= CQ_STAGE_MAX * CQ_MAX_THREADS && cq_staged_bytes(rs, 1) <= 160 * 1024 &&
         cq_staged_bytes(rs, 2) <= 160 * 1024;
}
// algorithmic HBM bytes per launch: forward - the [R,128] rows in, D1W / D2 / C2Q / Q2C out, M2 and the two saved softmaxes; backward -
// 8 row tensors in, 5 out, softmaxes + M2 in, dM2 out
static double cq_fwd_bytes(const RowSpace& rs) { return 4.0 * (5.0 * rs.R * HUAL_D + 2.0 * rs.B * (3.0 * cq_mat_elems(rs.T, rs.L))); }
static double cq_bwd_bytes(const RowSpace& rs) { return 4.0 * (13.0 * rs.R * HUAL_D + 2.0 * rs.B * (4.0 * cq_mat_elems(rs.T, rs.L))); }

// tri_prep (dropout on both roles of every row, rank-1 terms) + the attention itself: one launch when the clip fits the staged
// kernel (it prepares the rows as it stages them), else two
// Which kernels serve a shape: queries of at most 32 words against clips of L <= T <= 256 frames - every shape of the YAML configs and of
// BASELINE.json - run the kernels of cqwide.hip (round 5: at the bench shape 20.8 + 26.1 us against 26.3 + 33.6 us for the staged kernels
// below, at B32 T256 33 + 47 us against 65 + 6 + 109 us for the global-operand kernels); longer queries (or queries longer than the clip)
// the staged kernels when everything fits LDS, else the global-operand kernels.  HUAL_CQ_NO_WIDE=1 takes cqwide.hip out (A/B timings, tests).
static bool cq_use_wide(const RowSpace& rs) {
  static const bool off = getenv("HUAL_CQ_NO_WIDE") != nullptr && atoi(getenv("HUAL_CQ_NO_WIDE")) != 0;
  return !off && cq_wide_ok(rs);
}
static RowSpace cq_shape(int B, int T, int L) {
  RowSpace rs{};
  rs.B = B; rs.T = T; rs.L = L; rs.Nv = B * T; rs.Nq = B * L; rs.R = rs.Nv + rs.Nq;
  return rs;
}
bool cq_fwd_global(int B, int T, int L) {
  const RowSpace rs = cq_shape(B, T, L);
  return !cq_use_wide(rs) && !cq_staged_ok(rs) && cq_lds_bytes(rs, 3) > 160 * 1024;
}
bool cq_bwd_global(int B, int T, int L) {
  const RowSpace rs = cq_shape(B, T, L);
  return !cq_use_wide(rs) && !cq_staged_ok(rs) && cq_lds_bytes(rs, 4) > 160 * 1024;
}
int launch_cq_fwd(const CqBufs& b, const CqParams& p, const RowSpace& rs, const DropCfg& drop, hipStream_t s) {
  if (cq_use_wide(rs)) return launch_cq_fwd_wide(b, p, rs, drop, s);
  if (cq_staged_ok(rs)) {
    HUAL_DYN_LDS(cq_fwd_staged_kernel, 160 * 1024);
    HUAL_LAUNCH(2.0 * 8.0 * rs.B * rs.T * rs.L * HUAL_D, cq_fwd_bytes(rs), cq_fwd_staged_kernel, dim3(xcd_round8(rs.B), 2), dim3(CQ_MAX_THREADS), cq_staged_bytes(rs, 1), s, b, p, rs, drop);
    HUAL_CHECK_HIP(hipGetLastError());
    return 0;
  }
  int rc = launch_tri_prep(b, p, rs, drop, s);
  if (rc) return rc;
  const bool glob = cq_fwd_global(rs.B, rs.T, rs.L);
  const int bytes = glob ? cq_lds_bytes(rs, 0) : cq_lds_bytes(rs, 3);
  HUAL_REQUIRE(!glob || b.GS != nullptr, "cq_fwd: the global-memory form needs the score scratch (CqBufs::GS)");
  HUAL_DYN_LDS(cq_fwd_kernel, 160 * 1024);
  HUAL_LAUNCH(2.0 * 8.0 * rs.B * rs.T * rs.L * HUAL_D, cq_fwd_bytes(rs), cq_fwd_kernel, dim3(xcd_round8(rs.B), 2), dim3(cq_threads()), bytes, s, b, rs, glob ? 1 : 0);
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
  if (cq_use_wide(rs)) return launch_cq_bwd_wide(b, g, rs, dXa, dXb, s);
  if (cq_staged_ok(rs)) {
    HUAL_DYN_LDS(cq_bwd_staged_kernel, 160 * 1024);
    HUAL_LAUNCH(2.0 * 18.0 * rs.B * rs.T * rs.L * HUAL_D, cq_bwd_bytes(rs), cq_bwd_staged_kernel, dim3(xcd_round8(rs.B), 2), dim3(CQ_MAX_THREADS), cq_staged_bytes(rs, 2), s, b, g, rs, dXa, dXb);
    HUAL_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const bool glob = cq_bwd_global(rs.B, rs.T, rs.L);
  const int bytes = glob ? cq_lds_bytes(rs, 0) : cq_lds_bytes(rs, 4);
  HUAL_REQUIRE(!glob || g.GD != nullptr, "cq_bwd: the global-memory form needs the gradient scratch (CqBwdBufs::GD)");
  HUAL_DYN_LDS(cq_bwd_kernel, 160 * 1024);
  HUAL_LAUNCH(2.0 * 18.0 * rs.B * rs.T * rs.L * HUAL_D, cq_bwd_bytes(rs), cq_bwd_kernel, dim3(xcd_round8(rs.B), 2), dim3(cq_threads()), bytes, s, b, g, rs, dXa, dXb, glob ? 1 : 0);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int tri_bwd_blocks_v(const RowSpace& rs) { return cdiv(rs.Nv, 8); }
int tri_bwd_blocks_q(const RowSpace& rs) { return cdiv(rs.Nq, 8); }
int launch_tri_bwd_impl(const CqBufs& b, const CqBwdBufs& g, const CqParams& p, float* part, const RowSpace& rs,
                        const DropCfg& drop, const float* dXa, const float* dXb, hipStream_t s) {
  HUAL_REQUIRE(part != nullptr, "tri_bwd: null partial-sum buffer");
  const int nv = tri_bwd_blocks_v(rs), nq = tri_bwd_blocks_q(rs);
  // bytes: X, D2, dD1W, dD2, dXa, dXb, dX in; dX out
  HUAL_LAUNCH(0.0, 8.0 * 512.0 * rs.R, tri_bwd_kernel, dim3(nv + nq), dim3(256), 0, s, b, g, p, part, rs, drop, dXa, dXb, nv);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

size_t cq_mat_elems_host(int T, int L) { return cq_mat_elems(T, L); }
size_t cq_m2_rows_host(int T, int L) { return cq_m2_rows(T, L); }

}  // namespace hual
