// Context-query attention for queries of at most 32 words against clips of up to 256 frames (every shape of the YAML configs and of
// BASELINE.json: Charades T <= 128, ActivityNet T = 256): /root/reference/models/layers.py:114-130 (cq_attention), ops.py:94-116
// (trilinear_attention).  The staged kernels of cq.hip keep EVERY [rows,128] operand of a clip as a split image in LDS - 256 rows of fp16
// pairs are 128 KB, so they stop at 128 frames - and re-stage the long side for every phase; here the long side never goes through LDS
// except for the products that contract over it.
//
// One workgroup per (clip, direction) - 16 waves for clips of more than 128 frames, 8 waves (and 256 registers per lane: every global
// load requested a phase ahead) for shorter ones - written in the LONG x SHORT view U[l][s] of the score matrix whatever the direction
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
__host__ __device__ inline CqwLds cqw_lds_map(int nblk) {      // nblk: blocks of 128 long-side entries (1 or 2)
  CqwLds l;
  int o = 0;
  l.simg = o; o += CQW_SQ * 512;       // short-side row image
  l.ps = o; o += nblk * CQW_BLK;       // softmax along the short axis (later: dscore)
  l.pl = o; o += nblk * CQW_BLK;       // softmax along the long axis
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

#ifdef HUAL_STAMPS
// debug: clock stamps of wave 0 before and after every barrier of the long-clip kernels (scripts/exp/cqw_stamps.py): 64 slots per
// workgroup, forward kernel in [0, 32), backward in [32, 64)
__device__ unsigned long long g_cqw_stamps[256 * 64];
extern "C" int hual_debug_cqw_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cqw_stamps), sizeof(unsigned long long) * (size_t)n);
}
#define CQW_STAMP_INIT(base) int cqw_si = (base)
// per-wave stamps of the backward kernel's first phase: slot s (0..3) of wave w at [wg][s][w]
__device__ unsigned long long g_cqw_wstamps[256 * 64];
extern "C" int hual_debug_cqw_wstamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cqw_wstamps), sizeof(unsigned long long) * (size_t)n);
}
#define CQW_WSTAMP(s, drain) do { if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if ((threadIdx.x & 63) == 0) g_cqw_wstamps[(blockIdx.y * gridDim.x + blockIdx.x) % 256 * 64 + (s) * 16 + (threadIdx.x >> 6)] = __builtin_readcyclecounter(); } while (0)
#define CQW_STAMP() do { if (threadIdx.x == 0) g_cqw_stamps[(blockIdx.y * gridDim.x + blockIdx.x) % 256 * 64 + cqw_si] = __builtin_readcyclecounter(); ++cqw_si; } while (0)
#else
#define CQW_STAMP_INIT(base) do { } while (0)
#define CQW_STAMP() do { } while (0)
#define CQW_WSTAMP(s, drain) do { } while (0)
#endif
#define CQW_BARRIER() do { CQW_STAMP(); cq_barrier(); CQW_STAMP(); } while (0)

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
// The products whose results leave the kernel (or enter an image) are computed TRANSPOSED - the operand that carries the output's
// columns goes in as A - so that lane (j, g) register r holds OUT[row0 + j][col0 + 4 g + r]: four consecutive columns of one row = one
// 16-byte global store / one 8-byte store per image plane (the direct orientation would hold four ROWS: four scalar stores each).
// "alpha" products: OUT[l] = sum over s of P[l][s] S[s] for the wave's own rows (contraction over the 32 short rows = one k-step);
// ph / pl = the transposed fragment of the wave's 16 columns of the P image; lane (j, g) register r = OUT[l0 + j][n0 + 4 g + r]
__device__ __forceinline__ f32x4 cqw_alpha_tile(const uint4& ph, const uint4& pl, const CqImg& S, int n0, int lane) {
  uint4 sh, sl;
  cq_frag<true>(S, n0, 0, lane, sh, sl);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return cqw_mfma3(sh, sl, ph, pl, z);
}
// "beta" products: OUT[s] = sum over the K rows l of a chunk of P[l][s] LONG[l]; lane (j, g) register r = OUT[s0 + j][n0 + 4 g + r]
__device__ __forceinline__ f32x4 cqw_beta(f32x4 acc, const CqImg& Pb, int s0, const CqImg& chunk, int n0, int K, int lane) {
  for (int k0 = 0; k0 < K; k0 += 32) {
    uint4 ph, pl, ch, cl;
    cq_frag<false>(Pb, s0, k0, lane, ph, pl);
    cq_frag<true>(chunk, n0, k0, lane, ch, cl);
    acc = cqw_mfma3(ch, cl, ph, pl, acc);
  }
  return acc;
}
// range-checked 16-byte / 4-byte row stores (common.h: a store that is not to happen gets the offset ROW_SKIP - no branch around it)
__device__ __forceinline__ void cqw_st4(__amdgpu_buffer_rsrc_t r, bool ok, int row, int col, const f32x4& v, float scale) {
  bst4(r, ok ? (uint32_t)(row * HUAL_D + col) * 4u : ROW_SKIP, make_float4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale));
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

// geometry shared by the two kernels (NW waves: 16 or 8)
struct CqwGeom {
  int lane, wave, j, g, c4, Nl, Ns, Nlq, lbase, sbase, l0, lc0, lblk, nlive, nblk, K0, K1, lrow, lrc;
  bool live, lok;
};
__device__ __forceinline__ CqwGeom cqw_geom(const RowSpace& rs, int clip) {
  CqwGeom q;
  q.lane = threadIdx.x & 63; q.wave = threadIdx.x >> 6; q.j = q.lane & 15; q.g = q.lane >> 4; q.c4 = threadIdx.x & 31;
  q.Nl = rs.T; q.Ns = rs.L; q.Nlq = (rs.T + 31) & ~31;
  q.lbase = clip * rs.T; q.sbase = rs.Nv + clip * rs.L;
  q.l0 = 16 * q.wave; q.lc0 = q.l0 & 127; q.lblk = q.l0 >> 7;
  q.live = q.l0 < q.Nlq; q.nlive = q.Nlq >> 4; q.nblk = (q.Nlq + 127) >> 7;
  q.K0 = q.Nlq < 128 ? q.Nlq : 128; q.K1 = q.Nlq - q.K0;      // rows of the two chunks (the second one may be empty)
  q.lrow = q.l0 + q.j; q.lok = q.live && q.lrow < q.Nl; q.lrc = min(q.lrow, q.Nl - 1);
  return q;
}
__device__ __forceinline__ CqImg cqw_blk(char* lds, int base, int c, float scale = CQ_SCALE_PROB) { return cq_img(lds + base + c * CQW_BLK, CQW_SQ, scale); }
// this lane's 32 columns of a row: the loads alone (cqw_rows_zero finishes what cqw_row_load does in one piece)
__device__ __forceinline__ void cqw_row_issue(const float* rowp, int g, float4 (&x)[8]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    x[2 * s] = ld4(rowp + 32 * s + 8 * g);
    x[2 * s + 1] = ld4(rowp + 32 * s + 8 * g + 4);
  }
}
__device__ __forceinline__ void cqw_row_zero(bool ok, float4 (&x)[8]) {
#pragma unroll
  for (int u = 0; u < 8; ++u) x[u] = f4_pick(ok, x[u], f4zero());
}
// the mask element a thread stages (threads 0 .. 255: long side, 256 .. 287: short side; zero beyond the clip): selects only - a branch
// in front of the load made the compiler wait for it before the next loads were requested
__device__ __forceinline__ float cqw_mask_load(const RowSpace& rs, const CqwGeom& q) {
  const int t = threadIdx.x;
  const bool isl = t < 256;
  const int idx = isl ? t : t - 256, n = isl ? q.Nl : q.Ns, base = isl ? q.lbase : q.sbase;
  const float mv = rs.rowmask[base + min(idx, n - 1)];
  return idx < n ? mv : 0.f;
}
// the NW waves' beta tiles: tile t = wave + NW u -> short rows 16 (t >> 3), columns 16 (t & 7).  With 8 waves a wave's two tiles share
// their columns (one chunk fragment feeds both)
template <int NW>
__device__ __forceinline__ void cqw_beta_all(f32x4 (&acc)[16 / NW], const CqImg& Pb, const CqImg& chunk, int K, int wave, int lane) {
  if (NW == 16) {
    acc[0] = cqw_beta(acc[0], Pb, 16 * (wave >> 3), chunk, 16 * (wave & 7), K, lane);
  } else {
    for (int k0 = 0; k0 < K; k0 += 32) {
      uint4 bh, bl;
      cq_frag<true>(chunk, 16 * wave, k0, lane, bh, bl);
#pragma unroll
      for (int u = 0; u < 16 / NW; ++u) {
        uint4 ph, pl;
        cq_frag<false>(Pb, 16 * u, k0, lane, ph, pl);
        acc[u] = cqw_mfma3(bh, bl, ph, pl, acc[u]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// forward.  DIR 0: long = x1 (d1w = dropout(x) * wm, s0 = dropout(x) . w0), short = x2 (d2 = dropout(x), s1 = d2 . w1); Sr = softmax along
// the short axis, Sc along the long one.  DIR 1: the roles swap.
template <int DIR, int NW>
__device__ __forceinline__ void cqw_fwd_body(const CqBufs& b, const CqParams& p, const RowSpace& rs, const DropCfg& drop, int clip, char* lds) {
  constexpr int NT = NW * 64, SR = 1024 / NT, TPW = 16 / NW;      // threads; short-row float4 per thread; beta tiles per wave
  const CqwGeom q = cqw_geom(rs, clip);
  const int lane = q.lane, j = q.j, g = q.g;
  const CqwLds L = cqw_lds_map(NW == 16 ? 2 : 1);
  const CqImg simg = cq_img(lds + L.simg, CQW_SQ, CQ_SCALE_ACT), chunk = cq_img(lds + L.chunk, 128, CQ_SCALE_ACT);
  float* vec = reinterpret_cast<float*>(lds + L.vec);
  float* mlong = vec + CQW_V_MLONG; float* tlong = vec + CQW_V_TLONG; float* mshort = vec + CQW_V_MSHORT; float* tshort = vec + CQW_V_TSHORT;
  float* ca = vec + CQW_V_CA; float* cb = vec + CQW_V_CB;
  const DropRegs dr = drop_load(drop);
  CQW_STAMP_INIT(0);
  CQW_STAMP();
  const float* w_lterm = DIR == 0 ? p.w0[0] : p.w1[1];
  const float* w_sterm = DIR == 0 ? p.w1[0] : p.w0[1];
  const float* w_mul = DIR == 0 ? p.wm[0] : p.wm[1];                  // applies to the x1 role: the long rows (DIR 0) / the short rows (DIR 1)
  const uint32_t site_long = (uint32_t)HUAL_SITE_TRI + (DIR == 0 ? 0u : 3u), site_short = (uint32_t)HUAL_SITE_TRI + (DIR == 0 ? 1u : 2u);
  float* Dlong = DIR == 0 ? b.D1W : b.D2; float* Tlong = DIR == 0 ? b.S0 : b.S1;
  float* Dshort = DIR == 0 ? b.D2 : b.D1W; float* Tshort = DIR == 0 ? b.S1 : b.S0;
  const float* xrow = b.X + (size_t)(q.lbase + q.lrc) * HUAL_D;
  const uint32_t rbytes = (uint32_t)rs.R * HUAL_D * 4u;
  const __amdgpu_buffer_rsrc_t rDlong = row_rsrc(Dlong, rbytes), rTlong = row_rsrc(Tlong, (uint32_t)rs.R * 4u);
  const __amdgpu_buffer_rsrc_t rDshort = row_rsrc(Dshort, rbytes), rTshort = row_rsrc(Tshort, (uint32_t)rs.R * 4u);
  auto srow = [&](int u) { return (int)(threadIdx.x + NT * u) >> 5; };      // short row of this thread's u-th float4 (column 4 c4)
  // ---- every load of the first phase before any store (stores would pin the order of the loads behind them)
  float mval;
  {
    mval = cqw_mask_load(rs, q);
  }
  float4 xs[SR];
#pragma unroll
  for (int u = 0; u < SR; ++u) xs[u] = ld4(b.X + (size_t)(q.sbase + min(srow(u), q.Ns - 1)) * HUAL_D + 4 * q.c4);
  const float4 wS = ld4(w_sterm + 4 * q.c4), mS = ld4(w_mul + 4 * q.c4);
  float4 x[8];
  cqw_row_issue(xrow, g, x);
  float4 wl[8], wm[8];                 // (8 waves: the long side's weights with the first round of loads; 16 waves: no registers for that)
  if (NW == 8 && q.live) {
    cqw_row_issue(w_lterm, g, wl);
    if (DIR == 0) cqw_row_issue(w_mul, g, wm);
  }
#pragma unroll
  for (int u = 0; u < SR; ++u) xs[u] = f4_pick(srow(u) < q.Ns, xs[u], f4zero());
  cqw_row_zero(q.lok, x);
  // ---- the wave's long rows: the raw rows of the first 128 go into the chunk image at once (it is free until the first product
  // that contracts over the long side), then dropout, the rank-1 term, the prepared rows out and as fragments
  CqwFrag fa;
  if (q.live) {
    if (q.lblk == 0) {
      cqw_split(x, CQ_SCALE_ACT, fa);
      cqw_frag_store(chunk, q.lc0 + j, g, fa);
    }
    float term = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float4 a = x[2 * s], c = x[2 * s + 1];
      if (dr.enabled) {
        const uint32_t bits = drop_bits8_r(dr, site_long, (uint32_t)(q.lbase + q.lrc), (uint32_t)(4 * s + g));
        a = f4_select(bits & 15u, f4scale1(a, dr.scale));
        c = f4_select(bits >> 4, f4scale1(c, dr.scale));
      }
      const float4 w0 = NW == 8 ? wl[2 * s] : ld4(w_lterm + 32 * s + 8 * g), w1 = NW == 8 ? wl[2 * s + 1] : ld4(w_lterm + 32 * s + 8 * g + 4);
      term += (a.x * w0.x + a.y * w0.y + a.z * w0.z + a.w * w0.w) + (c.x * w1.x + c.y * w1.y + c.z * w1.z + c.w * w1.w);
      if (DIR == 0) {
        const float4 m0 = NW == 8 ? wm[2 * s] : ld4(w_mul + 32 * s + 8 * g), m1 = NW == 8 ? wm[2 * s + 1] : ld4(w_mul + 32 * s + 8 * g + 4);
        a = make_float4(a.x * m0.x, a.y * m0.y, a.z * m0.z, a.w * m0.w);
        c = make_float4(c.x * m1.x, c.y * m1.y, c.z * m1.z, c.w * m1.w);
      }
      x[2 * s] = a; x[2 * s + 1] = c;
    }
    term = cqw_gsum(term);
    if (g == 0) tlong[q.lrow] = q.lok ? term : 0.f;
    cqw_split(x, CQ_SCALE_ACT, fa);
    const uint32_t ro = (uint32_t)((q.lbase + q.lrow) * HUAL_D + 8 * g) * 4u;      // (ROW_SKIP per store: it must not be offset)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bst4(rDlong, q.lok ? ro + 128u * s : ROW_SKIP, x[2 * s]);
      bst4(rDlong, q.lok ? ro + 128u * s + 16u : ROW_SKIP, x[2 * s + 1]);
    }
    bst1(rTlong, (q.lok && g == 0) ? (uint32_t)(q.lbase + q.lrow) * 4u : ROW_SKIP, term);
  }
  // ---- short rows: dropout, rank-1 term, the prepared row out, its image
#pragma unroll
  for (int u = 0; u < SR; ++u) {
    const int k = srow(u);
    const bool sok = k < q.Ns;
    float4 o = xs[u];
    if (dr.enabled) {
      const uint32_t bits = drop_bits8_r(dr, site_short, (uint32_t)(q.sbase + min(k, q.Ns - 1)), (uint32_t)(q.c4 >> 1));
      o = f4_select((bits >> (4 * (q.c4 & 1))) & 15u, f4scale1(xs[u], dr.scale));
    }
    const float sv = half_sum32(o.x * wS.x + o.y * wS.y + o.z * wS.z + o.w * wS.w);
    if (DIR == 1) o = make_float4(o.x * mS.x, o.y * mS.y, o.z * mS.z, o.w * mS.w);
    if (q.c4 == 0) tshort[k] = sok ? sv : 0.f;
    cq_img_store4<1>(simg, k, 4 * q.c4, o);
    bst4(rDshort, sok ? (uint32_t)((q.sbase + k) * HUAL_D + 4 * q.c4) * 4u : ROW_SKIP, o);
    bst1(rTshort, (sok && q.c4 == 0) ? (uint32_t)(q.sbase + k) * 4u : ROW_SKIP, sv);
  }
  if ((int)threadIdx.x < 256) mlong[threadIdx.x] = mval;
  else if ((int)threadIdx.x < 256 + CQW_SQ) mshort[threadIdx.x - 256] = mval;
  CQW_BARRIER();
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
  CQW_BARRIER();
  if (NW == 16 && q.live && q.lblk == 1) cqw_row_issue(xrow, g, x);      // the raw rows of the second chunk, requested two phases ahead
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
  // the raw short rows for the products with the probabilities
#pragma unroll
  for (int u = 0; u < SR; ++u) cq_img_store4<1>(simg, srow(u), 4 * q.c4, xs[u]);
  CQW_BARRIER();
  const float inv_pa = 1.0f / (CQ_SCALE_PROB * CQ_SCALE_ACT);
  float* M2 = b.M2 + ((size_t)DIR * rs.B + clip) * cq_m2_rows(rs.T, rs.L) * HUAL_D;
  uint4 ah = make_uint4(0u, 0u, 0u, 0u), al = ah;
  if (q.live) cq_frag<true>(cqw_blk(lds, L.ps, q.lblk), q.lc0, 0, lane, ah, al);      // the wave's columns of the short-axis softmax
  // OUT[own rows] = Ps . (short image) -> out (rows of the long side)
  auto alpha_out = [&](float* out) {
    if (q.live) {
      const __amdgpu_buffer_rsrc_t ro = row_rsrc(out, rbytes);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) cqw_st4(ro, q.lok, q.lbase + q.lrow, 16 * nt + 4 * g, cqw_alpha_tile(ah, al, simg, 16 * nt, lane), inv_pa);
    }
  };
  auto chunk_raw1 = [&]() {      // (16 waves) the raw rows of the second chunk, requested above
    if (NW == 16 && q.live && q.lblk == 1) {
      cqw_row_zero(q.lok, x);
      CqwFrag f;
      cqw_split(x, CQ_SCALE_ACT, f);
      cqw_frag_store(chunk, q.lc0 + j, g, f);
    }
  };
  // both softmaxes out for the backward pass: the images as they stand
  auto save_images = [&]() {
    const size_t mat = cq_mat_elems(rs.T, rs.L);
    uint4* gps = reinterpret_cast<uint4*>((DIR == 0 ? b.SR : b.SC) + ((size_t)DIR * rs.B + clip) * mat);
    uint4* gpl = reinterpret_cast<uint4*>((DIR == 0 ? b.SC : b.SR) + ((size_t)DIR * rs.B + clip) * mat);
    const uint4* lps = reinterpret_cast<const uint4*>(lds + L.ps);
    const uint4* lpl = reinterpret_cast<const uint4*>(lds + L.pl);
    for (int idx = threadIdx.x; idx < q.nblk * (CQW_BLK / 16); idx += NT) { gps[idx] = lps[idx]; gpl[idx] = lpl[idx]; }
  };
  auto tile_s0 = [&](int u) { return 16 * ((q.wave + NW * u) >> 3); };
  auto tile_n0 = [&](int u) { return 16 * ((q.wave + NW * u) & 7); };
  f32x4 acc[TPW];
  auto acc_zero = [&]() {
#pragma unroll
    for (int u = 0; u < TPW; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto beta_short_out = [&](float* out) {      // beta tiles -> rows of the short side
    const __amdgpu_buffer_rsrc_t ro = row_rsrc(out, rbytes);
#pragma unroll
    for (int u = 0; u < TPW; ++u) cqw_st4(ro, tile_s0(u) + j < q.Ns, q.sbase + tile_s0(u) + j, tile_n0(u) + 4 * g, acc[u], inv_pa);
  };
  acc_zero();
  if (DIR == 0) {
    cqw_beta_all<NW>(acc, cqw_blk(lds, L.pl, 0), chunk, q.K0, q.wave, lane);       // M2 = Sc^T . x1
    alpha_out(b.C2Q);                                                           // c2q = Sr . x2
    save_images();
    CQW_BARRIER();
    if (NW == 16 && q.nblk == 2) {
      chunk_raw1();
      CQW_BARRIER();
      cqw_beta_all<NW>(acc, cqw_blk(lds, L.pl, 1), chunk, q.K1, q.wave, lane);
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const float4 v = make_float4(acc[u][0] * inv_pa, acc[u][1] * inv_pa, acc[u][2] * inv_pa, acc[u][3] * inv_pa);
      st4(M2 + (size_t)(tile_s0(u) + j) * HUAL_D + tile_n0(u) + 4 * g, v);
      cq_img_store4<1>(simg, tile_s0(u) + j, tile_n0(u) + 4 * g, v);
    }
    CQW_BARRIER();
    alpha_out(b.Q2C);                                                           // q2c = Sr . M2
  } else {
    cqw_beta_all<NW>(acc, cqw_blk(lds, L.pl, 0), chunk, q.K0, q.wave, lane);       // c2q = Sr . x2 (Sr: along the long axis)
    f32x4 m2o[8];                                                               // M2 = Sc^T . x1: the wave's rows
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) m2o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q.live) {
      const __amdgpu_buffer_rsrc_t rm = row_rsrc(M2, (uint32_t)cq_m2_rows(rs.T, rs.L) * HUAL_D * 4u);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const f32x4 t = cqw_alpha_tile(ah, al, simg, 16 * nt, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) m2o[nt][r] = t[r] * inv_pa;
        cqw_st4(rm, q.lok, q.lrow, 16 * nt + 4 * g, m2o[nt], 1.0f);
      }
    }
    save_images();
    if (NW == 16 && q.nblk == 2) {
      CQW_BARRIER();
      chunk_raw1();
      CQW_BARRIER();
      cqw_beta_all<NW>(acc, cqw_blk(lds, L.pl, 1), chunk, q.K1, q.wave, lane);
    }
    beta_short_out(b.C2Q);
    auto chunk_m2 = [&](int blk) {
      if (q.live && q.lblk == blk) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) cq_img_store4<1>(chunk, q.lc0 + j, 16 * nt + 4 * g, make_float4(m2o[nt][0], m2o[nt][1], m2o[nt][2], m2o[nt][3]));
      }
    };
    CQW_BARRIER();
    chunk_m2(0);
    CQW_BARRIER();
    acc_zero();
    cqw_beta_all<NW>(acc, cqw_blk(lds, L.pl, 0), chunk, q.K0, q.wave, lane);       // q2c = Sr . M2
    if (NW == 16 && q.nblk == 2) {
      CQW_BARRIER();
      chunk_m2(1);
      CQW_BARRIER();
      cqw_beta_all<NW>(acc, cqw_blk(lds, L.pl, 1), chunk, q.K1, q.wave, lane);
    }
    beta_short_out(b.Q2C);
  }
  CQW_STAMP();
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void cq_fwd_wide_kernel(CqBufs b, CqParams p, RowSpace rs, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char cqw_lds[];
  const int clip = xcd_tile(blockIdx.x, gridDim.x);      // XCD-aware clip order (common.h)
  if (clip >= rs.B) return;
  if (blockIdx.y == 0) cqw_fwd_body<0, NW>(b, p, rs, drop, clip, cqw_lds);
  else cqw_fwd_body<1, NW>(b, p, rs, drop, clip, cqw_lds);
}

// ------------------------------------------------------------------------------------------------------
// backward (the math of cq_bwd_kernel in cq.hip, in the long x short view; Ps / Pl = the saved softmax along the short / long axis):
//   DIR 0:  dPs = dc2q . x2^T + dq2c . M2^T (own rows)      dXb = Ps^T . dc2q, dM2 = Ps^T . dq2c (beta)     dPl = x1 . dM2^T (own rows)
//           dXa = Pl . dM2 (alpha)     dD1W = dscore . d2 (alpha)     dD2 = dscore^T . d1w (beta)
//   DIR 1:  dPl = x2 . dc2q^T + M2 . dq2c^T (own rows)      dXb = Pl . dc2q, dM2 = Pl . dq2c (alpha)        dPs = dM2 . x1^T (own rows)
//           dXa = Ps^T . dM2 (beta)    dD2 = dscore . d1w (alpha)     dD1W = dscore^T . d2 (beta)
//   dscore = Ps (dPs - <Ps, dPs> along s) mask_short[s] + Pl (dPl - <Pl, dPl> along l) mask_long[l]; its row / column sums are d s0 / d s1.
// With 8 waves (256 registers per lane) the rows of every later phase are requested one or two phases ahead (PF); with 16 waves they
// are loaded where they are used.
template <int DIR, int NW>
__device__ __forceinline__ void cqw_bwd_body(const CqBufs& b, const CqBwdBufs& gb, const RowSpace& rs, float* dXa, float* dXb, int clip, char* lds) {
  constexpr int NT = NW * 64, SR = 1024 / NT, TPW = 16 / NW;
  constexpr bool PF = NW == 8;
  const CqwGeom q = cqw_geom(rs, clip);
  const int lane = q.lane, j = q.j, g = q.g;
  const CqwLds L = cqw_lds_map(NW == 16 ? 2 : 1);
  const CqImg simg = cq_img(lds + L.simg, CQW_SQ, CQ_SCALE_ACT), chunk = cq_img(lds + L.chunk, 128, CQ_SCALE_ACT);
  float* vec = reinterpret_cast<float*>(lds + L.vec);
  float* mlong = vec + CQW_V_MLONG; float* mshort = vec + CQW_V_MSHORT;
  float* ca = vec + CQW_V_CA; float* cb = vec + CQW_V_CB; float* mx0 = vec + CQW_V_MX0; float* mx1 = vec + CQW_V_MX1;
  const size_t lrowoff = (size_t)(q.lbase + q.lrc) * HUAL_D;
  const float* M2 = b.M2 + ((size_t)DIR * rs.B + clip) * cq_m2_rows(rs.T, rs.L) * HUAL_D;
  CQW_STAMP_INIT(32);
  CQW_STAMP();
  CQW_WSTAMP(0, false);
  auto srow = [&](int u) { return (int)(threadIdx.x + NT * u) >> 5; };
  auto short_rows = [&](const float* base, float4 (&v)[SR]) {      // this thread's float4 of the short-side rows of a [R,128] tensor (zero beyond Ns)
#pragma unroll
    for (int u = 0; u < SR; ++u) v[u] = f4_pick(srow(u) < q.Ns, ld4(base + (size_t)(q.sbase + min(srow(u), q.Ns - 1)) * HUAL_D + 4 * q.c4), f4zero());
  };
  auto short_store = [&](const CqImg& im, const float4 (&v)[SR]) {
#pragma unroll
    for (int u = 0; u < SR; ++u) cq_img_store4<1>(im, srow(u), 4 * q.c4, v[u]);
  };
  auto short_absmax = [&](const float4 (&v)[SR]) {
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < SR; ++u) m = fmaxf(m, f4absmax(v[u]));
    return m;
  };
  // ---- the saved softmaxes (plain copies of the forward's images: exactly two 16-byte pieces per thread and image with either wave
  // count) and the masks: requested here, written to LDS by prologue_finish() once the direction's own loads are on their way too (as
  // a loop of load / wait / store per piece these were three memory round trips in front of everything else)
  uint4 cps[2], cpl[2];
  float mval;
  {
    const size_t mat = cq_mat_elems(rs.T, rs.L);
    const uint4* gps = reinterpret_cast<const uint4*>((DIR == 0 ? b.SR : b.SC) + ((size_t)DIR * rs.B + clip) * mat);
    const uint4* gpl = reinterpret_cast<const uint4*>((DIR == 0 ? b.SC : b.SR) + ((size_t)DIR * rs.B + clip) * mat);
#pragma unroll
    for (int it = 0; it < 2; ++it) { cps[it] = gps[threadIdx.x + NT * it]; cpl[it] = gpl[threadIdx.x + NT * it]; }
    mval = cqw_mask_load(rs, q);
  }
  auto prologue_finish = [&]() {
    uint4* lps = reinterpret_cast<uint4*>(lds + L.ps);
    uint4* lpl = reinterpret_cast<uint4*>(lds + L.pl);
#pragma unroll
    for (int it = 0; it < 2; ++it) { lps[threadIdx.x + NT * it] = cps[it]; lpl[threadIdx.x + NT * it] = cpl[it]; }
    if ((int)threadIdx.x < 256) mlong[threadIdx.x] = mval;
    else if ((int)threadIdx.x < 256 + CQW_SQ) mshort[threadIdx.x - 256] = mval;
  };
  float dps[2][4], dpl[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) dps[nt][r] = dpl[nt][r] = 0.f;
  float4 xa[8], xb[8];                                                          // staging of the wave's rows of two [R,128] tensors
  auto rows_issue = [&](const float* base, float4 (&x)[8]) { cqw_row_issue(base + lrowoff, g, x); };
  auto rows_split = [&](float4 (&x)[8], float scale, CqwFrag& f) {
    cqw_row_zero(q.lok, x);
    cqw_split(x, scale, f);
  };
  float* dS_long = DIR == 0 ? gb.dS0 : gb.dS1;
  const __amdgpu_buffer_rsrc_t rdsl = row_rsrc(dS_long, (uint32_t)rs.R * 4u);
  float* dS_short = DIR == 0 ? gb.dS1 : gb.dS0;
  CqImg dsc0 = cqw_blk(lds, L.ps, 0), dsc1 = cqw_blk(lds, L.ps, NW == 16 ? 1 : 0);      // the dscore image (takes the place of Ps)
  auto tile_s0 = [&](int u) { return 16 * ((q.wave + NW * u) >> 3); };
  auto tile_n0 = [&](int u) { return 16 * ((q.wave + NW * u) & 7); };
  f32x4 acc[TPW];
  auto acc_zero = [&]() {
#pragma unroll
    for (int u = 0; u < TPW; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

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
        bst1(rdsl, (j == 0 && l < q.Nl) ? (uint32_t)(q.lbase + l) * 4u : ROW_SKIP, v);
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
  const uint32_t rbytes = (uint32_t)rs.R * HUAL_D * 4u;
  auto alpha_out = [&](const CqImg& Pb, const CqImg& S, float scale, float* out) {
    if (q.live) {
      const __amdgpu_buffer_rsrc_t ro = row_rsrc(out, rbytes);
      uint4 ah, al;
      cq_frag<true>(Pb, q.lc0, 0, lane, ah, al);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) cqw_st4(ro, q.lok, q.lbase + q.lrow, 16 * nt + 4 * g, cqw_alpha_tile(ah, al, S, 16 * nt, lane), scale);
    }
  };
  auto beta_out = [&](float scale, float* out) {      // the beta tiles -> rows of the short side
    const __amdgpu_buffer_rsrc_t ro = row_rsrc(out, rbytes);
#pragma unroll
    for (int u = 0; u < TPW; ++u) cqw_st4(ro, tile_s0(u) + j < q.Ns, q.sbase + tile_s0(u) + j, tile_n0(u) + 4 * g, acc[u], scale);
  };
  // one beta product over both chunks; the wave's fragments f go into the chunk image (block 0 by the caller BEFORE the barrier in front
  // of this call when `stored0`, block 1 in here)
  auto beta_rounds = [&](const CqImg& P0, const CqImg& P1, const CqImg& ch, const CqwFrag& f) {
    cqw_beta_all<NW>(acc, P0, ch, q.K0, q.wave, lane);
    if (NW == 16 && q.nblk == 2) {
      CQW_BARRIER();
      if (q.live && q.lblk == 1) cqw_frag_store(ch, q.lc0 + j, g, f);
      CQW_BARRIER();
      cqw_beta_all<NW>(acc, P1, ch, q.K1, q.wave, lane);
    }
  };

  if (DIR == 0) {
    float4 xs[SR], m2s[SR], d2s[SR];
    short_rows(b.X, xs);
#pragma unroll
    for (int u = 0; u < SR; ++u) m2s[u] = ld4(M2 + (size_t)srow(u) * HUAL_D + 4 * q.c4);
    if (PF) short_rows(b.D2, d2s);
    rows_issue(gb.dC2Q, xa);
    CQW_WSTAMP(1, false);
    CQW_WSTAMP(2, true);
    prologue_finish();
    short_store(simg, xs);
    cqw_row_zero(q.lok, xa);
    cq_wgmax_put(mx0, cqw_absmax(xa));
    CQW_WSTAMP(3, false);
    CQW_BARRIER();                                                              // 1
    // d q2c is not needed before the second half of dSr: its rows are requested HERE, not with the first burst (the first phase of
    // 128 workgroups asks for 200 KB each at once and is bound by that burst), and its scale rides on the barrier in front of its use
    rows_issue(gb.dQ2C, xb);
    CqImg cg1 = chunk, cg2 = chunk;
    cq_img_autoscale(cg1, cq_wgmax_get(mx0));
    CqwFrag f;
    cqw_split(xa, cg1.scale, f);
    if (q.live) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 t = cqw_mma_rows(f, simg, 16 * nt, lane);                   // dc2q . x2^T
#pragma unroll
        for (int r = 0; r < 4; ++r) dps[nt][r] = t[r] * (cg1.inv * (1.0f / CQ_SCALE_ACT));
      }
      if (q.lblk == 0) cqw_frag_store(cg1, q.lc0 + j, g, f);
    }
    CQW_BARRIER();                                                              // 2
    short_store(simg, m2s);
    acc_zero();
    beta_rounds(cqw_blk(lds, L.ps, 0), cqw_blk(lds, L.ps, NW == 16 ? 1 : 0), cg1, f);      // dXb = Sr^T . dc2q
    beta_out(cg1.inv * (1.0f / CQ_SCALE_PROB), dXb);
    cqw_row_zero(q.lok, xb);
    cq_wgmax_put(mx1, cqw_absmax(xb));
    if (PF) rows_issue(b.X, xa);
    CQW_BARRIER();                                                              // 5 (the M2 image is complete; every read of the dc2q chunk is done)
    cq_img_autoscale(cg2, cq_wgmax_get(mx1));
    cqw_split(xb, cg2.scale, f);
    if (q.live) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 t = cqw_mma_rows(f, simg, 16 * nt, lane);                   // + dq2c . M2^T
#pragma unroll
        for (int r = 0; r < 4; ++r) dps[nt][r] = fmaf(t[r], cg2.inv * (1.0f / CQ_SCALE_ACT), dps[nt][r]);
      }
      if (q.lblk == 0) cqw_frag_store(cg2, q.lc0 + j, g, f);
    }
    CQW_BARRIER();                                                              // 6
    acc_zero();
    beta_rounds(cqw_blk(lds, L.ps, 0), cqw_blk(lds, L.ps, NW == 16 ? 1 : 0), cg2, f);      // dM2 = Sr^T . dq2c
    if (PF) rows_issue(b.D1W, xb);
    float dmmax = 0.f;
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[u][r] *= cg2.inv * (1.0f / CQ_SCALE_PROB);
        dmmax = fmaxf(dmmax, fabsf(acc[u][r]));
      }
    cq_wgmax_put(mx0, dmmax);
    CQW_BARRIER();                                                              // 9
    CqImg sdm = simg;
    cq_img_autoscale(sdm, cq_wgmax_get(mx0));
#pragma unroll
    for (int u = 0; u < TPW; ++u) cq_img_store4<1>(sdm, tile_s0(u) + j, tile_n0(u) + 4 * g, make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]));
    if (!PF) rows_issue(b.X, xa);
    rows_split(xa, CQ_SCALE_ACT, f);
    CQW_BARRIER();                                                              // 10
    if (q.live) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 t = cqw_mma_rows(f, sdm, 16 * nt, lane);                    // dSc = x1 . dM2^T
#pragma unroll
        for (int r = 0; r < 4; ++r) dpl[nt][r] = t[r] * (sdm.inv * (1.0f / CQ_SCALE_ACT));
      }
    }
    if (!PF) { short_rows(b.D2, d2s); rows_issue(b.D1W, xb); }
    sm_part1();
    alpha_out(cqw_blk(lds, L.pl, q.lblk), sdm, sdm.inv * (1.0f / CQ_SCALE_PROB), dXa);      // dXa = Sc . dM2
    CQW_BARRIER();                                                              // 11
    short_store(simg, d2s);
    rows_split(xb, CQ_SCALE_ACT, f);
    if (q.live && q.lblk == 0) cqw_frag_store(chunk, q.lc0 + j, g, f);
    sm_part2();
    CQW_BARRIER();                                                              // 12
    sm_part3();
    CQW_BARRIER();                                                              // 13
    acc_zero();
    beta_rounds(dsc0, dsc1, chunk, f);                                          // dD2 = dscore^T . d1w
    alpha_out(q.lblk == 0 ? dsc0 : dsc1, simg, dsc0.inv * (1.0f / CQ_SCALE_ACT), gb.dD1W);  // dD1W = dscore . d2
    beta_out(dsc0.inv * (1.0f / CQ_SCALE_ACT), gb.dD2);
  } else {
    float4 g1s[SR], g2s[SR], xs[SR], d1s[SR];
    short_rows(gb.dC2Q, g1s);
    short_rows(gb.dQ2C, g2s);
    if (PF) { short_rows(b.X, xs); short_rows(b.D1W, d1s); }
    rows_issue(b.X, xa);
    prologue_finish();
    cq_wgmax_put(mx0, short_absmax(g1s));
    cq_wgmax_put(mx1, short_absmax(g2s));
    CQW_BARRIER();                                                              // 1
    cqw_row_issue(M2 + (size_t)q.lrc * HUAL_D, g, xb);      // (behind the first burst, used last in the next phase: see direction 0)
    CqImg sa = simg, sb = cq_img(lds + L.chunk, CQW_SQ, CQ_SCALE_ACT);          // dc2q, dq2c images (the second one in the idle chunk buffer)
    cq_img_autoscale(sa, cq_wgmax_get(mx0));
    cq_img_autoscale(sb, cq_wgmax_get(mx1));
    short_store(sa, g1s);
    short_store(sb, g2s);
    CqwFrag fx;
    rows_split(xa, CQ_SCALE_ACT, fx);
    if (PF) rows_issue(b.D2, xa);
    CQW_BARRIER();                                                              // 2
    f32x4 dm[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) dm[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dmmax = 0.f;
    if (q.live) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 t = cqw_mma_rows(fx, sa, 16 * nt, lane);                    // dSr^T = x2 . dc2q^T (+ M2 . dq2c^T below)
#pragma unroll
        for (int r = 0; r < 4; ++r) dpl[nt][r] = t[r] * (sa.inv * (1.0f / CQ_SCALE_ACT));
      }
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
    alpha_out(cqw_blk(lds, L.pl, q.lblk), sa, sa.inv * (1.0f / CQ_SCALE_PROB), dXb);        // dXb = Sr^T . dc2q (rows of the long side)
    {
      CqwFrag fm;
      rows_split(xb, CQ_SCALE_ACT, fm);
      if (q.live) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x4 u = cqw_mma_rows(fm, sb, 16 * nt, lane);                  // + M2 . dq2c^T
#pragma unroll
          for (int r = 0; r < 4; ++r) dpl[nt][r] = fmaf(u[r], sb.inv * (1.0f / CQ_SCALE_ACT), dpl[nt][r]);
        }
      }
    }
    if (!PF) { short_rows(b.X, xs); short_rows(b.D1W, d1s); }
    CQW_BARRIER();                                                              // 3
    CqImg cdm = chunk;
    cq_img_autoscale(cdm, cq_wgmax_get(mx0));
    short_store(simg, xs);
    // dM2 rows into the chunk (also the transposition from the accumulator layout to fragments of the wave's rows)
    auto chunk_dm = [&](int blk) {
      if (q.live && q.lblk == blk) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) cq_img_store4<1>(cdm, q.lc0 + j, 16 * nt + 4 * g, make_float4(dm[nt][0], dm[nt][1], dm[nt][2], dm[nt][3]));
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
    CQW_BARRIER();                                                              // 4
    acc_zero();
    cqw_beta_all<NW>(acc, cqw_blk(lds, L.ps, 0), cdm, q.K0, q.wave, lane);         // dXa = Sc . dM2 (rows of the short side)
    dps_rows(0);
    if (NW == 16 && q.nblk == 2) {
      CQW_BARRIER();                                                            // 5
      chunk_dm(1);
      CQW_BARRIER();                                                            // 6
      cqw_beta_all<NW>(acc, cqw_blk(lds, L.ps, 1), cdm, q.K1, q.wave, lane);
      dps_rows(1);
    }
    beta_out(cdm.inv * (1.0f / CQ_SCALE_PROB), dXa);
    if (!PF) rows_issue(b.D2, xa);
    sm_part1();
    CQW_BARRIER();                                                              // 7
    short_store(simg, d1s);
    CqwFrag f;
    rows_split(xa, CQ_SCALE_ACT, f);
    if (q.live && q.lblk == 0) cqw_frag_store(chunk, q.lc0 + j, g, f);
    sm_part2();
    CQW_BARRIER();                                                              // 8
    sm_part3();
    CQW_BARRIER();                                                              // 9
    acc_zero();
    beta_rounds(dsc0, dsc1, chunk, f);                                          // dD1W = dscore . d2
    alpha_out(q.lblk == 0 ? dsc0 : dsc1, simg, dsc0.inv * (1.0f / CQ_SCALE_ACT), gb.dD2);   // dD2 = dscore^T . d1w (rows of the long side)
    beta_out(dsc0.inv * (1.0f / CQ_SCALE_ACT), gb.dD1W);
  }
  CQW_STAMP();
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void cq_bwd_wide_kernel(CqBufs b, CqBwdBufs gb, RowSpace rs, float* dXa, float* dXb) {
  extern __shared__ __attribute__((aligned(16))) char cqw_lds[];
  const int clip = xcd_tile(blockIdx.x, gridDim.x);      // XCD-aware clip order (common.h)
  if (clip >= rs.B) return;
  if (blockIdx.y == 0) cqw_bwd_body<0, NW>(b, gb, rs, dXa, dXb, clip, cqw_lds);
  else cqw_bwd_body<1, NW>(b, gb, rs, dXa, dXb, clip, cqw_lds);
}

namespace hual {

bool cq_wide_ok(const RowSpace& rs) { return rs.T >= 1 && rs.T <= 256 && rs.L >= 1 && rs.L <= CQW_SQ && rs.T >= rs.L; }

// algorithmic bytes as for the staged kernels (cq.hip): rows in / out, M2, the two saved softmaxes (as images: 16 KB per 128 frames each)
int launch_cq_fwd_wide(const CqBufs& b, const CqParams& p, const RowSpace& rs, const DropCfg& drop, hipStream_t s) {
  HUAL_REQUIRE(cq_wide_ok(rs), "cq_fwd_wide: needs L <= T <= 256 and L <= 32");
  const int nblk = rs.T > 128 ? 2 : 1;
  const double flops = 2.0 * 8.0 * rs.B * rs.T * rs.L * HUAL_D, bytes = 4.0 * 5.0 * rs.R * HUAL_D + 2.0 * rs.B * (2.0 * nblk * CQW_BLK + 4.0 * CQW_SQ * HUAL_D);
  if (nblk == 2) {
    HUAL_DYN_LDS(cq_fwd_wide_kernel<16>, 160 * 1024);
    HUAL_LAUNCH(flops, bytes, cq_fwd_wide_kernel<16>, dim3(xcd_round8(rs.B), 2), dim3(1024), cqw_lds_map(2).total, s, b, p, rs, drop);
  } else {
    HUAL_DYN_LDS(cq_fwd_wide_kernel<8>, 160 * 1024);
    HUAL_LAUNCH(flops, bytes, cq_fwd_wide_kernel<8>, dim3(xcd_round8(rs.B), 2), dim3(512), cqw_lds_map(1).total, s, b, p, rs, drop);
  }
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_cq_bwd_wide(const CqBufs& b, const CqBwdBufs& g, const RowSpace& rs, float* dXa, float* dXb, hipStream_t s) {
  HUAL_REQUIRE(cq_wide_ok(rs), "cq_bwd_wide: needs L <= T <= 256 and L <= 32");
  const int nblk = rs.T > 128 ? 2 : 1;
  const double flops = 2.0 * 18.0 * rs.B * rs.T * rs.L * HUAL_D, bytes = 4.0 * 13.0 * rs.R * HUAL_D + 2.0 * rs.B * (2.0 * nblk * CQW_BLK + 4.0 * CQW_SQ * HUAL_D);
  if (nblk == 2) {
    HUAL_DYN_LDS(cq_bwd_wide_kernel<16>, 160 * 1024);
    HUAL_LAUNCH(flops, bytes, cq_bwd_wide_kernel<16>, dim3(xcd_round8(rs.B), 2), dim3(1024), cqw_lds_map(2).total, s, b, g, rs, dXa, dXb);
  } else {
    HUAL_DYN_LDS(cq_bwd_wide_kernel<8>, 160 * 1024);
    HUAL_LAUNCH(flops, bytes, cq_bwd_wide_kernel<8>, dim3(xcd_round8(rs.B), 2), dim3(512), cqw_lds_map(1).total, s, b, g, rs, dXa, dXb);
  }
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
