// Context-query attention for LONG clips: 128 < T <= 256 frames against queries of at most 32 words (BASELINE configs[3] / [4]: ActivityNet,
// T = 256) - the shapes for which the staged kernels of cq.hip (every [rows,128] operand of a clip as a split image in LDS) do not fit:
// 256 rows of fp16 pairs are 128 KB.  /root/reference/models/layers.py:114-130 (cq_attention), ops.py:94-116 (trilinear_attention).
//
// One workgroup of 16 waves per (clip, direction), written in the LONG x SHORT view U[l][s] of the score matrix whatever the direction
// (direction 0: x1 = the clip's frames = long side, direction 1: x1 = the query's words = short side, U = score^T):
//   * wave w OWNS the 16 long-side rows 16 w .. 16 w + 15.  It reads them from global memory straight into the A-operand layout of
//     v_mfma_f32_16x16x32_f16 (lane (j, g): row j, columns 32 s + 8 g .. + 7 of k-step s - which is also one 16-byte chunk of an LDS image
//     row, and the 8 columns ONE Philox call decides), so the score tile of its rows, the softmax along the short axis (16-lane DPP
//     reductions) and every "own rows x short image" product need no LDS staging of the long side at all;
//   * the softmax along the long axis crosses the waves once: per-wave column maxima / sums through LDS, combined by every lane (the
//     two-level form of the same softmax);
//   * the two probability matrices live in LDS as split images [32 short rows][long columns] in blocks of 128 columns (the layout they
//     are saved in for the backward pass), the short-side operand as ONE 16 KB row image that changes content between phases;
//   * products that CONTRACT over the long side (M2 = Sc^T x1, dD2 = dscore^T d1w, ...) take their long-side operand through a 128-row
//     chunk image (64 KB) filled by the waves that own those rows - two rounds for 256 rows.
// Arithmetic as in the staged kernels: every product on fp16 pairs (22-bit operands, three passes), activations at the fixed scale 2^4,
// probabilities at 2^10, gradient operands at a power-of-two scale taken from the largest element of the clip's tensor (cqimg.h).
#include <stdlib.h>
#include "cq.h"
#include "bf16x3.h"
#include "philox.h"
#include "tilecore.h"
#include "prof.h"
#include "cqimg.h"

using namespace hual;

#define CQW_SQ 32                      // short-side rows of every image
#define CQW_BLK (CQW_SQ * 512)         // bytes of one [32][128] block of a probability image (both planes)
struct CqwLds { int simg, ps, pl, chunk, vec, total; };
__host__ __device__ inline CqwLds cqw_lds_map() {
  CqwLds l;
  int o = 0;
  l.simg = o; o += CQW_SQ * 512;       // short-side row image
  l.ps = o; o += 2 * CQW_BLK;          // softmax along the short axis (later: dscore)
  l.pl = o; o += 2 * CQW_BLK;          // softmax along the long axis
  l.chunk = o; o += 128 * 512;         // 128 long-side rows
  l.vec = o; o += 1632 * 4;            // masks, rank-1 terms, cross-wave statistics (offsets below)
  l.total = o;
  return l;
}
#define CQW_V_MLONG 0
#define CQW_V_TLONG 256
#define CQW_V_MSHORT 512
#define CQW_V_TSHORT 544
#define CQW_V_CA 576
#define CQW_V_CB 1088
#define CQW_V_MX0 1600
#define CQW_V_MX1 1616

struct CqwFrag { uint4 h[4], l[4]; };      // 16 rows x 128 columns of a wave as A fragments: k-step s = columns 32 s .. 32 s + 31
// this lane's 32 columns of a row (rowp = the row's first element): x[2 s], x[2 s + 1] = columns 32 s + 8 g .. + 7
__device__ __forceinline__ void cqw_row_load(const float* rowp, int g, bool ok, float4 (&x)[8]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    x[2 * s] = ld4(rowp + 32 * s + 8 * g);
    x[2 * s + 1] = ld4(rowp + 32 * s + 8 * g + 4);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) x[u] = f4_pick(ok, x[u], f4zero());
}
__device__ __forceinline__ void cqw_split(const float4 (&x)[8], float scale, CqwFrag& f) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    uint2 h0, l0, h1, l1;
    f16_split4(f4scale1(x[2 * s], scale), h0, l0);
    f16_split4(f4scale1(x[2 * s + 1], scale), h1, l1);
    f.h[s] = make_uint4(h0.x, h0.y, h1.x, h1.y);
    f.l[s] = make_uint4(l0.x, l0.y, l1.x, l1.y);
  }
}
__device__ __forceinline__ float cqw_absmax(const float4 (&x)[8]) {
  float m = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) m = fmaxf(m, f4absmax(x[u]));
  return m;
}
// a wave's fragments <-> row `row` of a row image (one 16-byte chunk per k-step and plane)
__device__ __forceinline__ void cqw_frag_store(const CqImg& im, int row, int g, const CqwFrag& f) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int off = tile256_off(row, 4 * s + g);
    *reinterpret_cast<uint4*>(im.p + off) = f.h[s];
    *reinterpret_cast<uint4*>(im.p + im.plane + off) = f.l[s];
  }
}
__device__ __forceinline__ void cqw_frag_load(const CqImg& im, int row, int g, CqwFrag& f) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int off = tile256_off(row, 4 * s + g);
    f.h[s] = *reinterpret_cast<const uint4*>(im.p + off);
    f.l[s] = *reinterpret_cast<const uint4*>(im.p + im.plane + off);
  }
}
__device__ __forceinline__ f32x4 cqw_mfma3(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x4 t) {
  t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), t, 0, 0, 0);
  t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bl), t, 0, 0, 0);
  t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, bh), t, 0, 0, 0);
  return t;
}
// own rows x (row image)^T over the 128 columns: tile [l0 + 4 g + r][n0 + j] in lane (j, g) register r; raw sum (scales not divided out)
__device__ __forceinline__ f32x4 cqw_mma_rows(const CqwFrag& a, const CqImg& B, int n0, int lane) {
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    uint4 bh, bl;
    cq_frag<false>(B, n0, 32 * s, lane, bh, bl);
    t = cqw_mfma3(a.h[s], a.l[s], bh, bl, t);
  }
  return t;
}
// "alpha" products: OUT[l] = sum over s of P[l][s] S[s] for the wave's own rows (contraction over the 32 short rows = one k-step);
// ah / al = the transposed fragment of the wave's 16 columns of the P image
__device__ __forceinline__ f32x4 cqw_alpha_tile(const uint4& ah, const uint4& al, const CqImg& S, int n0, int lane) {
  uint4 bh, bl;
  cq_frag<true>(S, n0, 0, lane, bh, bl);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return cqw_mfma3(ah, al, bh, bl, z);
}
// "beta" products: OUT[s] = sum over the K rows l of a chunk of P[l][s] LONG[l]: tile [s0 + 4 g + r][n0 + j]
__device__ __forceinline__ f32x4 cqw_beta(f32x4 acc, const CqImg& Pb, int s0, const CqImg& chunk, int n0, int K, int lane) {
  for (int k0 = 0; k0 < K; k0 += 32) {
    uint4 ah, al, bh, bl;
    cq_frag<false>(Pb, s0, k0, lane, ah, al);
    cq_frag<true>(chunk, n0, k0, lane, bh, bl);
    acc = cqw_mfma3(ah, al, bh, bl, acc);
  }
  return acc;
}
__device__ __forceinline__ float cqw_max16(float v) {      // over the 16 lanes of a row group (same g)
  v = fmaxf(v, dpp_xor_partner(v, 1)); v = fmaxf(v, dpp_xor_partner(v, 2));
  v = fmaxf(v, dpp_xor_partner(v, 4)); v = fmaxf(v, dpp_xor_partner(v, 8));
  return v;
}
__device__ __forceinline__ float cqw_sum16(float v) {
  v += dpp_xor_partner(v, 1); v += dpp_xor_partner(v, 2);
  v += dpp_xor_partner(v, 4); v += dpp_xor_partner(v, 8);
  return v;
}
__device__ __forceinline__ float cqw_gmax(float v) {       // over the four row groups (same j)
  v = fmaxf(v, lane_xor16_partner(v));
  return fmaxf(v, lane_xor32_partner(v));
}
__device__ __forceinline__ float cqw_gsum(float v) {
  v += lane_xor16_partner(v);
  return v + lane_xor32_partner(v);
}
// elements (row, col .. col + 3) of an image, col % 4 == 0
__device__ __forceinline__ f32x4 cqw_img_load4(const CqImg& im, int row, int col) {
  const int off = tile256_off(row, col >> 3) + 2 * (col & 7);
  const uint2 h = *reinterpret_cast<const uint2*>(im.p + off), l = *reinterpret_cast<const uint2*>(im.p + im.plane + off);
  const f32x2_t h0 = __builtin_convertvector(__builtin_bit_cast(f16x2_t, h.x), f32x2_t), h1 = __builtin_convertvector(__builtin_bit_cast(f16x2_t, h.y), f32x2_t);
  const f32x2_t l0 = __builtin_convertvector(__builtin_bit_cast(f16x2_t, l.x), f32x2_t), l1 = __builtin_convertvector(__builtin_bit_cast(f16x2_t, l.y), f32x2_t);
  f32x4 v;
  v[0] = (h0[0] + l0[0]) * im.inv; v[1] = (h0[1] + l0[1]) * im.inv; v[2] = (h1[0] + l1[0]) * im.inv; v[3] = (h1[1] + l1[1]) * im.inv;
  return v;
}
__device__ __forceinline__ float4 cqw_f4(const float (&v)[4]) { return make_float4(v[0], v[1], v[2], v[3]); }

// geometry shared by the two kernels
struct CqwGeom {
  int lane, wave, j, g, Nl, Ns, Nlq, lbase, sbase, l0, lc0, lblk, nlive, K1, lrow, lrc, k, c4, kc, s0, n0;
  bool live, lok, sok;
};
__device__ __forceinline__ CqwGeom cqw_geom(const RowSpace& rs, int clip) {
  CqwGeom q;
  q.lane = threadIdx.x & 63; q.wave = threadIdx.x >> 6; q.j = q.lane & 15; q.g = q.lane >> 4;
  q.Nl = rs.T; q.Ns = rs.L; q.Nlq = (rs.T + 31) & ~31;
  q.lbase = clip * rs.T; q.sbase = rs.Nv + clip * rs.L;
  q.l0 = 16 * q.wave; q.lc0 = q.l0 & 127; q.lblk = q.l0 >> 7;
  q.live = q.l0 < q.Nlq; q.nlive = q.Nlq >> 4; q.K1 = q.Nlq - 128;
  q.lrow = q.l0 + q.j; q.lok = q.live && q.lrow < q.Nl; q.lrc = min(q.lrow, q.Nl - 1);
  q.k = threadIdx.x >> 5; q.c4 = threadIdx.x & 31; q.sok = q.k < q.Ns; q.kc = min(q.k, q.Ns - 1);      // short rows: one float4 per thread
  q.s0 = 16 * (q.wave >> 3); q.n0 = 16 * (q.wave & 7);                                                     // tile of a beta product
  return q;
}
__device__ __forceinline__ CqImg cqw_blk(char* lds, int base, int c, float scale = CQ_SCALE_PROB) { return cq_img(lds + base + c * CQW_BLK, CQW_SQ, scale); }

// ------------------------------------------------------------------------------------------------------
// forward.  DIR 0: long = x1 (d1w = dropout(x) * wm, s0 = dropout(x) . w0), short = x2 (d2 = dropout(x), s1 = d2 . w1); Sr = softmax along
// the short axis, Sc along the long one.  DIR 1: the roles swap.
template <int DIR>
__device__ __forceinline__ void cqw_fwd_body(const CqBufs& b, const CqParams& p, const RowSpace& rs, const DropCfg& drop, int clip, char* lds) {
  const CqwGeom q = cqw_geom(rs, clip);
  const int lane = q.lane, j = q.j, g = q.g;
  const CqwLds L = cqw_lds_map();
  const CqImg simg = cq_img(lds + L.simg, CQW_SQ, CQ_SCALE_ACT), chunk = cq_img(lds + L.chunk, 128, CQ_SCALE_ACT);
  float* vec = reinterpret_cast<float*>(lds + L.vec);
  float* mlong = vec + CQW_V_MLONG; float* tlong = vec + CQW_V_TLONG; float* mshort = vec + CQW_V_MSHORT; float* tshort = vec + CQW_V_TSHORT;
  float* ca = vec + CQW_V_CA; float* cb = vec + CQW_V_CB;
  const DropRegs dr = drop_load(drop);
  const float* w_lterm = DIR == 0 ? p.w0[0] : p.w1[1];
  const float* w_sterm = DIR == 0 ? p.w1[0] : p.w0[1];
  const float* w_mul = DIR == 0 ? p.wm[0] : p.wm[1];                  // applies to the x1 role: the long rows (DIR 0) / the short rows (DIR 1)
  const uint32_t site_long = (uint32_t)HUAL_SITE_TRI + (DIR == 0 ? 0u : 3u), site_short = (uint32_t)HUAL_SITE_TRI + (DIR == 0 ? 1u : 2u);
  float* Dlong = DIR == 0 ? b.D1W : b.D2; float* Tlong = DIR == 0 ? b.S0 : b.S1;
  float* Dshort = DIR == 0 ? b.D2 : b.D1W; float* Tshort = DIR == 0 ? b.S1 : b.S0;
  const float* xrow = b.X + (size_t)(q.lbase + q.lrc) * HUAL_D;
  // ---- short rows: one float4 per thread (dropout, rank-1 term, the prepared row out, its image)
  float4 xs = f4_pick(q.sok, ld4(b.X + (size_t)(q.sbase + q.kc) * HUAL_D + 4 * q.c4), f4zero());
  // ---- the wave's long rows
  CqwFrag fa;
  if (q.live) {
    float4 x[8];
    cqw_row_load(xrow, g, q.lok, x);
    float term = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float4 a = x[2 * s], c = x[2 * s + 1];
      if (dr.enabled) {
        const uint32_t bits = drop_bits8_r(dr, site_long, (uint32_t)(q.lbase + q.lrc), (uint32_t)(4 * s + g));
        a = f4_select(bits & 15u, f4scale1(a, dr.scale));
        c = f4_select(bits >> 4, f4scale1(c, dr.scale));
      }
      const float4 w0 = ld4(w_lterm + 32 * s + 8 * g), w1 = ld4(w_lterm + 32 * s + 8 * g + 4);
      term += (a.x * w0.x + a.y * w0.y + a.z * w0.z + a.w * w0.w) + (c.x * w1.x + c.y * w1.y + c.z * w1.z + c.w * w1.w);
      if (DIR == 0) {
        const float4 m0 = ld4(w_mul + 32 * s + 8 * g), m1 = ld4(w_mul + 32 * s + 8 * g + 4);
        a = make_float4(a.x * m0.x, a.y * m0.y, a.z * m0.z, a.w * m0.w);
        c = make_float4(c.x * m1.x, c.y * m1.y, c.z * m1.z, c.w * m1.w);
      }
      x[2 * s] = a; x[2 * s + 1] = c;
      if (q.lok) {
        float* o = Dlong + (size_t)(q.lbase + q.lrow) * HUAL_D + 32 * s + 8 * g;
        st4(o, a); st4(o + 4, c);
      }
    }
    term = cqw_gsum(term);
    if (g == 0) {
      tlong[q.lrow] = q.lok ? term : 0.f;
      if (q.lok) Tlong[q.lbase + q.lrow] = term;
    }
    cqw_split(x, CQ_SCALE_ACT, fa);
  }
  {
    float4 o = xs;
    if (dr.enabled) {
      const uint32_t bits = drop_bits8_r(dr, site_short, (uint32_t)(q.sbase + q.kc), (uint32_t)(q.c4 >> 1));
      o = f4_select((bits >> (4 * (q.c4 & 1))) & 15u, f4scale1(xs, dr.scale));
    }
    const float4 w = ld4(w_sterm + 4 * q.c4);
    const float sv = half_sum32(o.x * w.x + o.y * w.y + o.z * w.z + o.w * w.w);
    if (DIR == 1) {
      const float4 m = ld4(w_mul + 4 * q.c4);
      o = make_float4(o.x * m.x, o.y * m.y, o.z * m.z, o.w * m.w);
    }
    if (q.c4 == 0) tshort[q.k] = q.sok ? sv : 0.f;
    if (q.sok) {
      st4(Dshort + (size_t)(q.sbase + q.k) * HUAL_D + 4 * q.c4, o);
      if (q.c4 == 0) Tshort[q.sbase + q.k] = sv;
    }
    cq_img_store4<1>(simg, q.k, 4 * q.c4, o);
  }
  if ((int)threadIdx.x < 256) mlong[threadIdx.x] = (int)threadIdx.x < q.Nl ? rs.rowmask[q.lbase + threadIdx.x] : 0.f;
  else if ((int)threadIdx.x < 256 + CQW_SQ) {
    const int kk = threadIdx.x - 256;
    mshort[kk] = kk < q.Ns ? rs.rowmask[q.sbase + kk] : 0.f;
  }
  cq_barrier();
  // ---- scores of the wave's rows, the softmax along the short axis, the wave's part of the softmax along the long axis
  float lgl[2][4];
  bool valid[2][4];
  if (q.live) {
    float Ps[2][4], mxs[4], wmax[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) mxs[r] = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 t = cqw_mma_rows(fa, simg, 16 * nt, lane);
      const int col = 16 * nt + j;
      const float ts = tshort[col], ms = mshort[col];
      wmax[nt] = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int l = q.l0 + 4 * g + r;
        const float u = fmaf(t[r], 1.0f / (CQ_SCALE_ACT * CQ_SCALE_ACT), tlong[l] + ts), ml = mlong[l];
        valid[nt][r] = l < q.Nl && col < q.Ns;
        Ps[nt][r] = valid[nt][r] ? u * ms + HUAL_MASK_VALUE * (1.0f - ms) : -INFINITY;
        lgl[nt][r] = valid[nt][r] ? u * ml + HUAL_MASK_VALUE * (1.0f - ml) : -INFINITY;
        mxs[r] = fmaxf(mxs[r], Ps[nt][r]);
        wmax[nt] = fmaxf(wmax[nt], lgl[nt][r]);
      }
      wmax[nt] = cqw_gmax(wmax[nt]);
    }
    float sums[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mxs[r] = cqw_max16(mxs[r]);
      float e = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        Ps[nt][r] = valid[nt][r] ? __expf(Ps[nt][r] - mxs[r]) : 0.f;
        e += Ps[nt][r];
      }
      sums[r] = 1.0f / cqw_sum16(e);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float ws = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Ps[nt][r] = valid[nt][r] ? Ps[nt][r] * sums[r] : 0.f;
        ws += valid[nt][r] ? __expf(lgl[nt][r] - wmax[nt]) : 0.f;
      }
      ws = cqw_gsum(ws);
      if (g == 0) { ca[q.wave * 32 + 16 * nt + j] = wmax[nt]; cb[q.wave * 32 + 16 * nt + j] = ws; }
      cq_img_store4<1>(cqw_blk(lds, L.ps, q.lblk), 16 * nt + j, q.lc0 + 4 * g, cqw_f4(Ps[nt]));
    }
  }
  cq_barrier();
  if (q.live) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = 16 * nt + j;
      float M = -INFINITY;
      for (int w = 0; w < q.nlive; ++w) M = fmaxf(M, ca[w * 32 + col]);
      float Ls = 0.f;
      for (int w = 0; w < q.nlive; ++w) Ls += cb[w * 32 + col] * __expf(ca[w * 32 + col] - M);
      const float inv = 1.0f / Ls;
      float Pl[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) Pl[r] = valid[nt][r] ? __expf(lgl[nt][r] - M) * inv : 0.f;
      cq_img_store4<1>(cqw_blk(lds, L.pl, q.lblk), col, q.lc0 + 4 * g, cqw_f4(Pl));
    }
  }
  // the raw rows for the products with the probabilities: short side into its image, the first 128 long rows into the chunk
  cq_img_store4<1>(simg, q.k, 4 * q.c4, xs);
  auto chunk_raw = [&](int blk) {
    if (q.live && q.lblk == blk) {
      float4 x[8];
      cqw_row_load(xrow, g, q.lok, x);
      CqwFrag f;
      cqw_split(x, CQ_SCALE_ACT, f);
      cqw_frag_store(chunk, q.lc0 + j, g, f);
    }
  };
  chunk_raw(0);
  cq_barrier();
  // ---- both softmaxes out for the backward pass: the images as they stand
  {
    const size_t mat = cq_mat_elems(rs.T, rs.L);
    uint4* gps = reinterpret_cast<uint4*>((DIR == 0 ? b.SR : b.SC) + ((size_t)DIR * rs.B + clip) * mat);
    uint4* gpl = reinterpret_cast<uint4*>((DIR == 0 ? b.SC : b.SR) + ((size_t)DIR * rs.B + clip) * mat);
    const uint4* lps = reinterpret_cast<const uint4*>(lds + L.ps);
    const uint4* lpl = reinterpret_cast<const uint4*>(lds + L.pl);
    for (int idx = threadIdx.x; idx < 2 * CQW_BLK / 16; idx += CQ_MAX_THREADS) { gps[idx] = lps[idx]; gpl[idx] = lpl[idx]; }
  }
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const float inv_pa = 1.0f / (CQ_SCALE_PROB * CQ_SCALE_ACT);
  float* M2 = b.M2 + ((size_t)DIR * rs.B + clip) * cq_m2_rows(rs.T, rs.L) * HUAL_D;
  uint4 ah = make_uint4(0u, 0u, 0u, 0u), al = ah;
  if (q.live) cq_frag<true>(cqw_blk(lds, L.ps, q.lblk), q.lc0, 0, lane, ah, al);      // the wave's columns of the short-axis softmax
  // OUT[own rows] = Ps . (short image) -> out (rows of the long side)
  auto alpha_out = [&](float* out) {
    if (q.live) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const f32x4 t = cqw_alpha_tile(ah, al, simg, 16 * nt, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int l = q.l0 + 4 * g + r;
          if (l < q.Nl) out[(size_t)(q.lbase + l) * HUAL_D + 16 * nt + j] = t[r] * inv_pa;
        }
      }
    }
  };
  if (DIR == 0) {
    alpha_out(b.C2Q);                                                           // c2q = Sr . x2
    f32x4 acc = cqw_beta(zero, cqw_blk(lds, L.pl, 0), q.s0, chunk, q.n0, 128, lane);      // M2 = Sc^T . x1
    cq_barrier();
    chunk_raw(1);
    cq_barrier();
    acc = cqw_beta(acc, cqw_blk(lds, L.pl, 1), q.s0, chunk, q.n0, q.K1, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = acc[r] * inv_pa;
      M2[(size_t)(q.s0 + 4 * g + r) * HUAL_D + q.n0 + j] = v;
      cq_img_store1<1>(simg, q.s0 + 4 * g + r, q.n0 + j, v);
    }
    cq_barrier();
    alpha_out(b.Q2C);                                                           // q2c = Sr . M2
  } else {
    f32x4 acc = cqw_beta(zero, cqw_blk(lds, L.pl, 0), q.s0, chunk, q.n0, 128, lane);      // c2q = Sr . x2 (Sr: along the long axis)
    f32x4 m2o[8];                                                               // M2 = Sc^T . x1: the wave's rows
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) m2o[nt] = zero;
    if (q.live) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const f32x4 t = cqw_alpha_tile(ah, al, simg, 16 * nt, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          m2o[nt][r] = t[r] * inv_pa;
          const int l = q.l0 + 4 * g + r;
          if (l < q.Nl) M2[(size_t)l * HUAL_D + 16 * nt + j] = m2o[nt][r];
        }
      }
    }
    cq_barrier();
    chunk_raw(1);
    cq_barrier();
    acc = cqw_beta(acc, cqw_blk(lds, L.pl, 1), q.s0, chunk, q.n0, q.K1, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = q.s0 + 4 * g + r;
      if (s < q.Ns) b.C2Q[(size_t)(q.sbase + s) * HUAL_D + q.n0 + j] = acc[r] * inv_pa;
    }
    auto chunk_m2 = [&](int blk) {
      if (q.live && q.lblk == blk) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) cq_img_store1<1>(chunk, q.lc0 + 4 * g + r, 16 * nt + j, m2o[nt][r]);
      }
    };
    cq_barrier();
    chunk_m2(0);
    cq_barrier();
    acc = cqw_beta(zero, cqw_blk(lds, L.pl, 0), q.s0, chunk, q.n0, 128, lane);            // q2c = Sr . M2
    cq_barrier();
    chunk_m2(1);
    cq_barrier();
    acc = cqw_beta(acc, cqw_blk(lds, L.pl, 1), q.s0, chunk, q.n0, q.K1, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = q.s0 + 4 * g + r;
      if (s < q.Ns) b.Q2C[(size_t)(q.sbase + s) * HUAL_D + q.n0 + j] = acc[r] * inv_pa;
    }
  }
}

__global__ __launch_bounds__(CQ_MAX_THREADS) void cq_fwd_wide_kernel(CqBufs b, CqParams p, RowSpace rs, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char cqw_lds[];
  const int clip = xcd_tile(blockIdx.x, gridDim.x);      // XCD-aware clip order (common.h)
  if (clip >= rs.B) return;
  if (blockIdx.y == 0) cqw_fwd_body<0>(b, p, rs, drop, clip, cqw_lds);
  else cqw_fwd_body<1>(b, p, rs, drop, clip, cqw_lds);
}

// ------------------------------------------------------------------------------------------------------
// backward (the math of cq_bwd_kernel in cq.hip, in the long x short view; Ps / Pl = the saved softmax along the short / long axis):
//   DIR 0:  dPs = dc2q . x2^T + dq2c . M2^T (own rows)      dXb = Ps^T . dc2q, dM2 = Ps^T . dq2c (beta)     dPl = x1 . dM2^T (own rows)
//           dXa = Pl . dM2 (alpha)     dD1W = dscore . d2 (alpha)     dD2 = dscore^T . d1w (beta)
//   DIR 1:  dPl = x2 . dc2q^T + M2 . dq2c^T (own rows)      dXb = Pl . dc2q, dM2 = Pl . dq2c (alpha)        dPs = dM2 . x1^T (own rows)
//           dXa = Ps^T . dM2 (beta)    dD2 = dscore . d1w (alpha)     dD1W = dscore^T . d2 (beta)
//   dscore = Ps (dPs - <Ps, dPs> along s) mask_short[s] + Pl (dPl - <Pl, dPl> along l) mask_long[l]; its row / column sums are d s0 / d s1.
// Softmax backward pass common to both directions; dps / dpl = the gradients of the two softmaxes on the wave's rows (tile nt, register r)
struct CqwSm { float ds[2][4]; };
template <int DIR>
__device__ __forceinline__ void cqw_bwd_body(const CqBufs& b, const CqBwdBufs& gb, const RowSpace& rs, float* dXa, float* dXb, int clip, char* lds) {
  const CqwGeom q = cqw_geom(rs, clip);
  const int lane = q.lane, j = q.j, g = q.g;
  const CqwLds L = cqw_lds_map();
  const CqImg simg = cq_img(lds + L.simg, CQW_SQ, CQ_SCALE_ACT), chunk = cq_img(lds + L.chunk, 128, CQ_SCALE_ACT);
  float* vec = reinterpret_cast<float*>(lds + L.vec);
  float* mlong = vec + CQW_V_MLONG; float* mshort = vec + CQW_V_MSHORT;
  float* ca = vec + CQW_V_CA; float* cb = vec + CQW_V_CB; float* mx0 = vec + CQW_V_MX0; float* mx1 = vec + CQW_V_MX1;
  const size_t lrowoff = (size_t)(q.lbase + q.lrc) * HUAL_D, srowoff = (size_t)(q.sbase + q.kc) * HUAL_D + 4 * q.c4;
  const float* M2 = b.M2 + ((size_t)DIR * rs.B + clip) * cq_m2_rows(rs.T, rs.L) * HUAL_D;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // ---- the saved softmaxes (plain copies of the forward's images) and the masks
  {
    const size_t mat = cq_mat_elems(rs.T, rs.L);
    const uint4* gps = reinterpret_cast<const uint4*>((DIR == 0 ? b.SR : b.SC) + ((size_t)DIR * rs.B + clip) * mat);
    const uint4* gpl = reinterpret_cast<const uint4*>((DIR == 0 ? b.SC : b.SR) + ((size_t)DIR * rs.B + clip) * mat);
    uint4* lps = reinterpret_cast<uint4*>(lds + L.ps);
    uint4* lpl = reinterpret_cast<uint4*>(lds + L.pl);
    for (int idx = threadIdx.x; idx < 2 * CQW_BLK / 16; idx += CQ_MAX_THREADS) { lps[idx] = gps[idx]; lpl[idx] = gpl[idx]; }
  }
  if ((int)threadIdx.x < 256) mlong[threadIdx.x] = (int)threadIdx.x < q.Nl ? rs.rowmask[q.lbase + threadIdx.x] : 0.f;
  else if ((int)threadIdx.x < 256 + CQW_SQ) {
    const int kk = threadIdx.x - 256;
    mshort[kk] = kk < q.Ns ? rs.rowmask[q.sbase + kk] : 0.f;
  }
  float dps[2][4], dpl[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) dps[nt][r] = dpl[nt][r] = 0.f;
  auto own_rows = [&](const float* base, float scale, CqwFrag& f) {      // the wave's rows of a [R,128] tensor as fragments
    float4 x[8];
    cqw_row_load(base + lrowoff, g, q.lok, x);
    cqw_split(x, scale, f);
  };
  auto short_row = [&](const float* base) { return f4_pick(q.sok, ld4(base + srowoff), f4zero()); };
  float* dS_long = DIR == 0 ? gb.dS0 : gb.dS1;
  float* dS_short = DIR == 0 ? gb.dS1 : gb.dS0;
  CqImg dsc0 = cqw_blk(lds, L.ps, 0), dsc1 = cqw_blk(lds, L.ps, 1);      // the dscore image (takes the place of Ps)

  // softmax backward on the wave's rows: part 1 (before the barrier that publishes the long-axis dot products) ...
  float Ps[2][4], Pl[2][4], dots[4];
  auto sm_part1 = [&]() {
    if (q.live) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dots[r] = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 a = cqw_img_load4(cqw_blk(lds, L.ps, q.lblk), 16 * nt + j, q.lc0 + 4 * g);
        const f32x4 c = cqw_img_load4(cqw_blk(lds, L.pl, q.lblk), 16 * nt + j, q.lc0 + 4 * g);
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Ps[nt][r] = a[r]; Pl[nt][r] = c[r];
          dots[r] = fmaf(a[r], dps[nt][r], dots[r]);
          part = fmaf(c[r], dpl[nt][r], part);
        }
        part = cqw_gsum(part);
        if (g == 0) ca[q.wave * 32 + 16 * nt + j] = part;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dots[r] = cqw_sum16(dots[r]);
    }
  };
  // ... part 2: dscore in registers, its sums along the short axis out, its sums along the long axis and its maximum into LDS
  float ds[2][4];
  auto sm_part2 = [&]() {
    float dmax = 0.f;
    if (q.live) {
      float rs4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = 16 * nt + j;
        float dotl = 0.f;
        for (int w = 0; w < q.nlive; ++w) dotl += ca[w * 32 + col];
        const float ms = mshort[col];
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = Ps[nt][r] * (dps[nt][r] - dots[r]) * ms + Pl[nt][r] * (dpl[nt][r] - dotl) * mlong[q.l0 + 4 * g + r];
          ds[nt][r] = v;
          rs4[r] += v;
          part += v;
          dmax = fmaxf(dmax, fabsf(v));
        }
        part = cqw_gsum(part);
        if (g == 0) cb[q.wave * 32 + col] = part;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = cqw_sum16(rs4[r]);
        const int l = q.l0 + 4 * g + r;
        if (j == 0 && l < q.Nl) dS_long[q.lbase + l] = v;
      }
    }
    cq_wgmax_put(mx1, dmax);
  };
  // ... part 3 (behind the barrier): the dscore image at its own scale, the sums along the long axis out
  auto sm_part3 = [&]() {
    const float m = cq_wgmax_get(mx1);
    cq_img_autoscale(dsc0, m);
    cq_img_autoscale(dsc1, m);
    if (q.live) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) cq_img_store4<1>(q.lblk == 0 ? dsc0 : dsc1, 16 * nt + j, q.lc0 + 4 * g, cqw_f4(ds[nt]));
    }
    if ((int)threadIdx.x < q.Ns) {
      float v = 0.f;
      for (int w = 0; w < q.nlive; ++w) v += cb[w * 32 + threadIdx.x];
      dS_short[q.sbase + threadIdx.x] = v;
    }
  };
  // OUT[own rows] = P . (short image S), P = the wave's columns of an image block
  auto alpha_out = [&](const CqImg& Pb, const CqImg& S, float scale, float* out) {
    if (q.live) {
      uint4 ah, al;
      cq_frag<true>(Pb, q.lc0, 0, lane, ah, al);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const f32x4 t = cqw_alpha_tile(ah, al, S, 16 * nt, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int l = q.l0 + 4 * g + r;
          if (l < q.Nl) out[(size_t)(q.lbase + l) * HUAL_D + 16 * nt + j] = t[r] * scale;
        }
      }
    }
  };
  auto beta_out = [&](const f32x4& acc, float scale, float* out) {      // a beta tile -> rows of the short side
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = q.s0 + 4 * g + r;
      if (s < q.Ns) out[(size_t)(q.sbase + s) * HUAL_D + q.n0 + j] = acc[r] * scale;
    }
  };

  if (DIR == 0) {
    const float4 xs = short_row(b.X), m2s = ld4(M2 + (size_t)q.k * HUAL_D + 4 * q.c4), d2s = short_row(b.D2);
    cq_img_store4<1>(simg, q.k, 4 * q.c4, xs);
    float4 x1[8];
    cqw_row_load(gb.dC2Q + lrowoff, g, q.lok, x1);
    cq_wgmax_put(mx0, q.live ? cqw_absmax(x1) : 0.f);
    {
      float4 x2[8];
      cqw_row_load(gb.dQ2C + lrowoff, g, q.lok, x2);
      cq_wgmax_put(mx1, q.live ? cqw_absmax(x2) : 0.f);
    }
    cq_barrier();                                                               // 1
    CqImg cg1 = chunk, cg2 = chunk;
    cq_img_autoscale(cg1, cq_wgmax_get(mx0));
    cq_img_autoscale(cg2, cq_wgmax_get(mx1));
    CqwFrag f;
    cqw_split(x1, cg1.scale, f);
    if (q.live) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 t = cqw_mma_rows(f, simg, 16 * nt, lane);                   // dc2q . x2^T
#pragma unroll
        for (int r = 0; r < 4; ++r) dps[nt][r] = t[r] * (cg1.inv * (1.0f / CQ_SCALE_ACT));
      }
      if (q.lblk == 0) cqw_frag_store(cg1, q.lc0 + j, g, f);
    }
    cq_barrier();                                                               // 2
    f32x4 acc = cqw_beta(zero, cqw_blk(lds, L.ps, 0), q.s0, cg1, q.n0, 128, lane);        // dXb = Sr^T . dc2q
    cq_img_store4<1>(simg, q.k, 4 * q.c4, m2s);
    cq_barrier();                                                               // 3
    if (q.live && q.lblk == 1) cqw_frag_store(cg1, q.lc0 + j, g, f);
    cq_barrier();                                                               // 4
    acc = cqw_beta(acc, cqw_blk(lds, L.ps, 1), q.s0, cg1, q.n0, q.K1, lane);
    beta_out(acc, cg1.inv * (1.0f / CQ_SCALE_PROB), dXb);
    own_rows(gb.dQ2C, cg2.scale, f);
    if (q.live) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 t = cqw_mma_rows(f, simg, 16 * nt, lane);                   // + dq2c . M2^T
#pragma unroll
        for (int r = 0; r < 4; ++r) dps[nt][r] = fmaf(t[r], cg2.inv * (1.0f / CQ_SCALE_ACT), dps[nt][r]);
      }
    }
    cq_barrier();                                                               // 5
    if (q.live && q.lblk == 0) cqw_frag_store(cg2, q.lc0 + j, g, f);
    cq_barrier();                                                               // 6
    acc = cqw_beta(zero, cqw_blk(lds, L.ps, 0), q.s0, cg2, q.n0, 128, lane);              // dM2 = Sr^T . dq2c
    cq_barrier();                                                               // 7
    if (q.live && q.lblk == 1) cqw_frag_store(cg2, q.lc0 + j, g, f);
    cq_barrier();                                                               // 8
    acc = cqw_beta(acc, cqw_blk(lds, L.ps, 1), q.s0, cg2, q.n0, q.K1, lane);
    float dm[4], dmmax = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dm[r] = acc[r] * (cg2.inv * (1.0f / CQ_SCALE_PROB));
      dmmax = fmaxf(dmmax, fabsf(dm[r]));
    }
    cq_wgmax_put(mx0, dmmax);
    cq_barrier();                                                               // 9
    CqImg sdm = simg;
    cq_img_autoscale(sdm, cq_wgmax_get(mx0));
#pragma unroll
    for (int r = 0; r < 4; ++r) cq_img_store1<1>(sdm, q.s0 + 4 * g + r, q.n0 + j, dm[r]);
    cq_barrier();                                                               // 10
    own_rows(b.X, CQ_SCALE_ACT, f);
    if (q.live) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 t = cqw_mma_rows(f, sdm, 16 * nt, lane);                    // dSc = x1 . dM2^T
#pragma unroll
        for (int r = 0; r < 4; ++r) dpl[nt][r] = t[r] * (sdm.inv * (1.0f / CQ_SCALE_ACT));
      }
    }
    alpha_out(cqw_blk(lds, L.pl, q.lblk), sdm, sdm.inv * (1.0f / CQ_SCALE_PROB), dXa);      // dXa = Sc . dM2
    sm_part1();
    cq_barrier();                                                               // 11
    cq_img_store4<1>(simg, q.k, 4 * q.c4, d2s);
    own_rows(b.D1W, CQ_SCALE_ACT, f);
    if (q.live && q.lblk == 0) cqw_frag_store(chunk, q.lc0 + j, g, f);
    sm_part2();
    cq_barrier();                                                               // 12
    sm_part3();
    cq_barrier();                                                               // 13
    alpha_out(q.lblk == 0 ? dsc0 : dsc1, simg, dsc0.inv * (1.0f / CQ_SCALE_ACT), gb.dD1W);  // dD1W = dscore . d2
    acc = cqw_beta(zero, dsc0, q.s0, chunk, q.n0, 128, lane);                   // dD2 = dscore^T . d1w
    cq_barrier();                                                               // 14
    if (q.live && q.lblk == 1) cqw_frag_store(chunk, q.lc0 + j, g, f);
    cq_barrier();                                                               // 15
    acc = cqw_beta(acc, dsc1, q.s0, chunk, q.n0, q.K1, lane);
    beta_out(acc, dsc0.inv * (1.0f / CQ_SCALE_ACT), gb.dD2);
  } else {
    const float4 g1s = short_row(gb.dC2Q), g2s = short_row(gb.dQ2C);
    cq_wgmax_put(mx0, f4absmax(g1s));
    cq_wgmax_put(mx1, f4absmax(g2s));
    CqwFrag fx, fm;
    own_rows(b.X, CQ_SCALE_ACT, fx);
    {
      float4 x[8];
      cqw_row_load(M2 + (size_t)q.lrc * HUAL_D, g, q.lok, x);
      cqw_split(x, CQ_SCALE_ACT, fm);
    }
    cq_barrier();                                                               // 1
    CqImg sa = simg, sb = cq_img(lds + L.chunk, CQW_SQ, CQ_SCALE_ACT);          // dc2q, dq2c images (the second one in the idle chunk buffer)
    cq_img_autoscale(sa, cq_wgmax_get(mx0));
    cq_img_autoscale(sb, cq_wgmax_get(mx1));
    cq_img_store4<1>(sa, q.k, 4 * q.c4, g1s);
    cq_img_store4<1>(sb, q.k, 4 * q.c4, g2s);
    cq_barrier();                                                               // 2
    f32x4 dm[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) dm[nt] = zero;
    float dmmax = 0.f;
    if (q.live) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 t = cqw_mma_rows(fx, sa, 16 * nt, lane), u = cqw_mma_rows(fm, sb, 16 * nt, lane);      // dSr^T = x2 . dc2q^T + M2 . dq2c^T
#pragma unroll
        for (int r = 0; r < 4; ++r) dpl[nt][r] = t[r] * (sa.inv * (1.0f / CQ_SCALE_ACT)) + u[r] * (sb.inv * (1.0f / CQ_SCALE_ACT));
      }
    }
    alpha_out(cqw_blk(lds, L.pl, q.lblk), sa, sa.inv * (1.0f / CQ_SCALE_PROB), dXb);        // dXb = Sr^T . dc2q (rows of the long side)
    if (q.live) {
      uint4 ah, al;
      cq_frag<true>(cqw_blk(lds, L.pl, q.lblk), q.lc0, 0, lane, ah, al);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {                                          // dM2 = Sr^T . dq2c: the wave's rows
        const f32x4 t = cqw_alpha_tile(ah, al, sb, 16 * nt, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dm[nt][r] = t[r] * (sb.inv * (1.0f / CQ_SCALE_PROB));
          dmmax = fmaxf(dmmax, fabsf(dm[nt][r]));
        }
      }
    }
    cq_wgmax_put(mx0, dmmax);
    const float4 xs = short_row(b.X), d1s = short_row(b.D1W);
    cq_barrier();                                                               // 3
    CqImg cdm = chunk;
    cq_img_autoscale(cdm, cq_wgmax_get(mx0));
    cq_img_store4<1>(simg, q.k, 4 * q.c4, xs);
    // dM2 rows into the chunk (also the transposition from the accumulator layout to fragments of the wave's rows)
    auto chunk_dm = [&](int blk) {
      if (q.live && q.lblk == blk) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) cq_img_store1<1>(cdm, q.lc0 + 4 * g + r, 16 * nt + j, dm[nt][r]);
      }
    };
    auto dps_rows = [&](int blk) {                                              // dSc^T = dM2 . x1^T on the wave's rows
      if (q.live && q.lblk == blk) {
        CqwFrag fd;
        cqw_frag_load(cdm, q.lc0 + j, g, fd);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x4 t = cqw_mma_rows(fd, simg, 16 * nt, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) dps[nt][r] = t[r] * (cdm.inv * (1.0f / CQ_SCALE_ACT));
        }
      }
    };
    chunk_dm(0);
    cq_barrier();                                                               // 4
    dps_rows(0);
    f32x4 acc = cqw_beta(zero, cqw_blk(lds, L.ps, 0), q.s0, cdm, q.n0, 128, lane);        // dXa = Sc . dM2 (rows of the short side)
    cq_barrier();                                                               // 5
    chunk_dm(1);
    cq_barrier();                                                               // 6
    dps_rows(1);
    acc = cqw_beta(acc, cqw_blk(lds, L.ps, 1), q.s0, cdm, q.n0, q.K1, lane);
    beta_out(acc, cdm.inv * (1.0f / CQ_SCALE_PROB), dXa);
    sm_part1();
    cq_barrier();                                                               // 7
    cq_img_store4<1>(simg, q.k, 4 * q.c4, d1s);
    CqwFrag f;
    own_rows(b.D2, CQ_SCALE_ACT, f);
    if (q.live && q.lblk == 0) cqw_frag_store(chunk, q.lc0 + j, g, f);
    sm_part2();
    cq_barrier();                                                               // 8
    sm_part3();
    cq_barrier();                                                               // 9
    alpha_out(q.lblk == 0 ? dsc0 : dsc1, simg, dsc0.inv * (1.0f / CQ_SCALE_ACT), gb.dD2);   // dD2 = dscore^T . d1w (rows of the long side)
    acc = cqw_beta(zero, dsc0, q.s0, chunk, q.n0, 128, lane);                   // dD1W = dscore . d2
    cq_barrier();                                                               // 10
    if (q.live && q.lblk == 1) cqw_frag_store(chunk, q.lc0 + j, g, f);
    cq_barrier();                                                               // 11
    acc = cqw_beta(acc, dsc1, q.s0, chunk, q.n0, q.K1, lane);
    beta_out(acc, dsc0.inv * (1.0f / CQ_SCALE_ACT), gb.dD1W);
  }
}

__global__ __launch_bounds__(CQ_MAX_THREADS) void cq_bwd_wide_kernel(CqBufs b, CqBwdBufs gb, RowSpace rs, float* dXa, float* dXb) {
  extern __shared__ __attribute__((aligned(16))) char cqw_lds[];
  const int clip = xcd_tile(blockIdx.x, gridDim.x);      // XCD-aware clip order (common.h)
  if (clip >= rs.B) return;
  if (blockIdx.y == 0) cqw_bwd_body<0>(b, gb, rs, dXa, dXb, clip, cqw_lds);
  else cqw_bwd_body<1>(b, gb, rs, dXa, dXb, clip, cqw_lds);
}

namespace hual {

bool cq_wide_ok(const RowSpace& rs) { return rs.T > 128 && rs.T <= 256 && rs.L >= 1 && rs.L <= CQW_SQ; }

// algorithmic bytes as for the staged kernels (cq.hip): rows in / out, M2, the two saved softmaxes (as images: 32 KB each)
int launch_cq_fwd_wide(const CqBufs& b, const CqParams& p, const RowSpace& rs, const DropCfg& drop, hipStream_t s) {
  HUAL_REQUIRE(cq_wide_ok(rs), "cq_fwd_wide: needs 128 < T <= 256 and L <= 32");
  HUAL_DYN_LDS(cq_fwd_wide_kernel, 160 * 1024);
  HUAL_LAUNCH(2.0 * 8.0 * rs.B * rs.T * rs.L * HUAL_D, 4.0 * 5.0 * rs.R * HUAL_D + 2.0 * rs.B * (4.0 * CQW_BLK + 4.0 * CQW_SQ * HUAL_D), cq_fwd_wide_kernel,
              dim3(xcd_round8(rs.B), 2), dim3(CQ_MAX_THREADS), cqw_lds_map().total, s, b, p, rs, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_cq_bwd_wide(const CqBufs& b, const CqBwdBufs& g, const RowSpace& rs, float* dXa, float* dXb, hipStream_t s) {
  HUAL_REQUIRE(cq_wide_ok(rs), "cq_bwd_wide: needs 128 < T <= 256 and L <= 32");
  HUAL_DYN_LDS(cq_bwd_wide_kernel, 160 * 1024);
  HUAL_LAUNCH(2.0 * 18.0 * rs.B * rs.T * rs.L * HUAL_D, 4.0 * 13.0 * rs.R * HUAL_D + 2.0 * rs.B * (4.0 * CQW_BLK + 4.0 * CQW_SQ * HUAL_D), cq_bwd_wide_kernel,
              dim3(xcd_round8(rs.B), 2), dim3(CQ_MAX_THREADS), cqw_lds_map().total, s, b, g, rs, dXa, dXb);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
