// Fused conv_block kernels (see convblock.h).
//
// Forward, one workgroup (512 threads = 8 waves) per MT owned rows:
//   LDS  X   [MT+24][128] fp32     x_l of the rows r0-12 .. r0+MT+11 (updated in place layer by layer)
//        A   hi / lo planes        the GEMM operand c_l = depthwise7(LN(x_l)) pre-split into scaled fp16 pairs (bf16x3.h)
//        W   64 KB                 the pointwise weight image of the current layer (LDS-DMA, requested one phase ahead)
//   per layer l (halo H_l = 9, 6, 3, 0 rows on either side of the owned rows):
//     P1  every 32-lane group slides a 7-row window over its chunk of rows: layer norm (shuffle reductions), depthwise
//         taps masked at clip boundaries (modules.py:66 zero padding), row scale, split, write A; c / mean / rstd of owned rows
//         go to HBM for backward
//     P2  wave (mt, ch) multiplies row tile mt by column half ch: A fragments by ds_read_b128, weight fragments by
//         ds_read_b64_tr_b16, 48 x v_mfma_f32_16x16x32_f16
//     P3  bias, relu (saved), Philox dropout, + x_l from LDS -> x_{l+1} back into X (+ HBM for the owned rows)
#include "convblock.h"
#include "bf16x3.h"
#include "philox.h"
#include "prof.h"

using namespace hual;

#define LN_EPS 1e-6f   // models/layers.py:15
#define CB_THREADS 512
#define CB_TILE 16384            // one [64][128 x 16 bit] tile
#define CB_STAGE (2 * CB_TILE)   // hi + lo tile of 64 K rows
#define CB_WBYTES (2 * CB_STAGE) // a whole [128,128] weight image

__device__ __forceinline__ float4 cb_fma(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 cb_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 cb_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float cb_hsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

// clip segment [lo, hi) of unified row `row` (rowops.h RowSpace)
__device__ __forceinline__ void cb_segment(int row, const RowSpace& rs, int& lo, int& hi) {
  if (row < rs.Nv) { const int b = row / rs.T; lo = b * rs.T; hi = lo + rs.T; }
  else { const int q = row - rs.Nv; const int b = q / rs.L; lo = rs.Nv + b * rs.L; hi = lo + rs.L; }
}

// LDS-DMA of one [128,128] weight image (pack_weights_kernel layout: per K row 256 B of fp16 high parts, 256 B of
// residuals) into two stages of {hi tile, lo tile}; the XOR swizzle of tile256_off is applied on the global side
__device__ __forceinline__ void cb_dma_weight(const float* wimg, char* Wl, int wave, int lane, int nwaves) {
  const char* img = reinterpret_cast<const char*>(wimg);
  const int chp = lane & 15, rr = lane >> 4;
  for (int pc = wave; pc < 64; pc += nwaves) {
    const int st = pc >> 5, pl = pc & 31;
    const int r = 4 * (pl & 15) + rr;
    const int ch = chp ^ (((r & 3) << 2) | ((r >> 2) & 3));
    const char* src = img + (size_t)(64 * st + r) * 512 + (pl >> 4) * 256 + 16 * ch;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(Wl + st * CB_STAGE + pl * 1024), 16, 0, 0);
  }
}

// 16 x 64 output tile of A[16 rows of the LDS planes] . W: wave-level, accumulators in the column order of the epilogue
// (accumulator t, register r, lane (j, g) = row 4g + r, column 64 ch + 4j + t)
__device__ __forceinline__ void cb_tile_mma(const char* Ahi, const char* Alo, const char* Wl, int mt, int ch, int lane,
                                            f32x4 (&acc)[4]) {
  const int j = lane & 15, g = lane >> 4;
  const int tq = (lane >> 2) & 3, tp = lane & 3;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int aoff = tile256_off(16 * mt + j, 4 * ks + g);
    const f16x8 ah = *reinterpret_cast<const f16x8*>(Ahi + aoff);
    const f16x8 al = *reinterpret_cast<const f16x8*>(Alo + aoff);
    const char* hi = Wl + (ks >> 1) * CB_STAGE;
    const int r0 = 32 * (ks & 1) + 8 * g + tq, r1 = r0 + 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int chunk = 8 * ch + 2 * t + (tp >> 1);
      const int o0 = tile256_off(r0, chunk) + 8 * (tp & 1), o1 = tile256_off(r1, chunk) + 8 * (tp & 1);
      const f16x8 wh = join_tr_f16(lds_read_tr16(hi, o0), lds_read_tr16(hi, o1));
      const f16x8 wl = join_tr_f16(lds_read_tr16(hi + CB_TILE, o0), lds_read_tr16(hi + CB_TILE, o1));
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh, acc[t], 0, 0, 0);
    }
  }
}

// row of the operand planes: scale to fp16 range, split, store (8 bytes per lane and plane); returns the inverse scale
__device__ __forceinline__ float cb_store_operand(char* Ahi, char* Alo, int arow, int l32, float4 v) {
  float inv;
  const float sc = f16_row_scale(half_max32(f4absmax(v)), inv);
  uint2 h, l;
  f16_split4(f4scale1(v, sc), h, l);
  const int off = tile256_off(arow, l32 >> 1) + 8 * (l32 & 1);
  *reinterpret_cast<uint2*>(Ahi + off) = h;
  *reinterpret_cast<uint2*>(Alo + off) = l;
  return inv;
}

__global__ __launch_bounds__(CB_THREADS) void conv_block_fwd_kernel(CbFwdArgs a, RowSpace rs, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char cb_lds[];
  const int MT = a.MT;
  const int XR = MT + 24;                                   // rows of X
  float4* X = reinterpret_cast<float4*>(cb_lds);            // [XR][32] float4
  char* Ahi = cb_lds + (size_t)XR * 512;                    // [64][256 B]
  char* Alo = Ahi + 64 * 256;
  char* Wl = Alo + 64 * 256;                                // CB_WBYTES
  float* ainv = reinterpret_cast<float*>(Wl + CB_WBYTES);   // [64] inverse operand scale per A row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;          // 16 groups of 32 lanes: one row = 32 x float4
  const int col = 4 * l32;
  const int R = rs.R;
  const int r0 = blockIdx.x * MT;
  const int xbase = r0 - 12;                                 // global row of X[0]

  cb_dma_weight(a.l[0].wimg, Wl, wave, lane, CB_THREADS / 64);
  // ---- block input (+ position embeddings for the predictor's feature encoder): every group requests its (at most 5)
  // rows before it touches any of them - one memory round trip instead of one per row
  {
    float4 xv[5], pv[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = grp + 16 * u, row = xbase + i;
      const bool ok = i < XR && row >= 0 && row < R;
      xv[u] = ok ? ld4(a.x0 + (size_t)row * HUAL_D + col) : f4zero();
      pv[u] = f4zero();
      if (ok && a.pos) {
        int lo, hi;
        cb_segment(row, rs, lo, hi);
        pv[u] = ld4(a.pos + (size_t)(row - lo) * HUAL_D + col);
      }
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = grp + 16 * u, row = xbase + i;
      if (i >= XR) continue;
      float4 v = xv[u];
      if (a.pos) {
        v = cb_add(v, pv[u]);
        if (row >= r0 && row < r0 + MT && row < R) st4(a.x0_out + (size_t)row * HUAL_D + col, v);
      }
      X[i * 32 + l32] = v;
    }
  }
  __syncthreads();

#pragma unroll 1
  for (int l = 0; l < 4; ++l) {
    const CbLayerFwd& L = a.l[l];
    const int H = 9 - 3 * l;                                 // halo of this layer's OUTPUT rows
    const int nout = MT + 2 * H;                             // output rows r0-H .. r0+MT+H-1
    const int ntile = (nout + 15) >> 4;
    const int obase = r0 - H;                                // global row of output / operand row 0
    const int mt = wave >> 1, ch = wave & 1;                 // P2 / P3: row tile and column half of this wave
    float4 bias = ld4(L.bias + 64 * ch + 4 * (lane & 15));
    // ---------------- P1: layer norm + depthwise conv -> operand planes
    {
      const float4 gam = ld4(L.ln_g + col), bet = ld4(L.ln_b + col);
      float4 w[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) w[k] = ld4(L.dw + k * HUAL_D + col);
      const int chunk = (16 * ntile + 15) >> 4;              // operand rows per group (incl. the zero rows up to 16*ntile)
      const int la = grp * chunk, lb = min(la + chunk, 16 * ntile);
      float4 h0 = f4zero(), h1 = f4zero(), h2 = f4zero(), h3 = f4zero(), h4 = f4zero(), h5 = f4zero(), h6 = f4zero();
      for (int li = la - 3; li < lb + 3; ++li) {
        // h of global row t = LN(x_l[t]) (zero outside the tensor): enters the window as its newest row
        const int t = obase + li;
        float4 hn = f4zero();
        if (li < nout + 3 && t >= 0 && t < R) {
          const float4 v = X[(t - xbase) * 32 + l32];
          const float mean = half_sum32(cb_hsum(v)) * (1.0f / HUAL_D);
          const float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
          const float var = half_sum32(cb_hsum(cb_mul(d, d))) * (1.0f / HUAL_D);
          const float rstd = rsqrtf(var + LN_EPS);
          hn = cb_fma(make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd), gam, bet);
          if (l32 == 0 && li >= la && li < lb && t >= r0 && t < r0 + MT) { L.mean[t] = mean; L.rstd[t] = rstd; }
        }
        h0 = h1; h1 = h2; h2 = h3; h3 = h4; h4 = h5; h5 = h6; h6 = hn;
        const int lo_ = li - 3;                              // operand row whose window (lo_-3 .. lo_+3) is complete now
        if (lo_ < la) continue;
        const int o = obase + lo_;
        float4 c = f4zero();
        const bool live = lo_ < nout && o >= 0 && o < R;
        if (live) {
          int slo, shi;
          cb_segment(o, rs, slo, shi);
          // taps outside the clip are zero padding (SAME, modules.py:66); same accumulation order as ln_dwconv_fwd_kernel
          c = cb_fma((o - 3 >= slo) ? h0 : f4zero(), w[0], c);
          c = cb_fma((o - 2 >= slo) ? h1 : f4zero(), w[1], c);
          c = cb_fma((o - 1 >= slo) ? h2 : f4zero(), w[2], c);
          c = cb_fma(h3, w[3], c);
          c = cb_fma((o + 1 < shi) ? h4 : f4zero(), w[4], c);
          c = cb_fma((o + 2 < shi) ? h5 : f4zero(), w[5], c);
          c = cb_fma((o + 3 < shi) ? h6 : f4zero(), w[6], c);
          if (o >= r0 && o < r0 + MT) st4(L.c + (size_t)o * HUAL_D + col, c);
        }
        const float inv = cb_store_operand(Ahi, Alo, lo_, l32, c);
        if (l32 == 0) ainv[lo_] = live ? inv : 0.f;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's share of the weight image has landed
    // make the compiler place its own wait for `bias` here, where nothing is in flight: a wait it inserted later would
    // be a vmcnt(0) that also drains the next layer's LDS-DMA (cdna_hip_programming.md, "Pipelining across barriers")
    asm volatile("" : "+v"(bias.x), "+v"(bias.y), "+v"(bias.z), "+v"(bias.w));
    __syncthreads();
    // ---------------- P2: pointwise convolution on the matrix cores
    f32x4 acc[4];
    if (mt < ntile) cb_tile_mma(Ahi, Alo, Wl, mt, ch, lane, acc);
    __syncthreads();                                         // operand planes and weight image are free again
    if (l + 1 < 4) cb_dma_weight(a.l[l + 1].wimg, Wl, wave, lane, CB_THREADS / 64);
    // ---------------- P3: bias, relu, dropout, residual
    if (mt < ntile) {
      const int j = lane & 15, g = lane >> 4;
      const int ecol = 64 * ch + 4 * j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = 16 * mt + 4 * g + r;
        const int o = obase + lr;
        if (lr >= nout || o < 0 || o >= R) continue;
        const float ir = ainv[lr];
        float4 v = make_float4(fmaf(acc[0][r], ir, 0.f) + bias.x, fmaf(acc[1][r], ir, 0.f) + bias.y,
                               fmaf(acc[2][r], ir, 0.f) + bias.z, fmaf(acc[3][r], ir, 0.f) + bias.w);
        v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        const bool own = o >= r0 && o < r0 + MT;
        if (own) st4(L.y + (size_t)o * HUAL_D + ecol, v);
        if (L.drop_site >= 0 && drop.enabled)
          v = apply_drop4(drop, (uint32_t)L.drop_site, a.drop_row0 + (uint32_t)o, (uint32_t)(ecol >> 2), v);
        float4* xp = X + (o - xbase) * 32 + (ecol >> 2);
        v = cb_add(v, *xp);
        *xp = v;
        if (own) st4(L.xout + (size_t)o * HUAL_D + ecol, v);
      }
    }
    __syncthreads();
  }
}

namespace hual {

int conv_block_fused_rows(int R) {
  // as many workgroups as there are CUs when the rows allow it (each workgroup streams all four weight images, so fewer,
  // taller workgroups cost nothing extra), never more than HUAL_CB_MAXMT rows: the halo'd operand must fit 64 rows
  int mt = cdiv(R, 256);
  if (mt < 16) mt = 16;
  if (mt > HUAL_CB_MAXMT) mt = HUAL_CB_MAXMT;
  return mt;
}

static size_t cb_fwd_lds(int MT) { return (size_t)(MT + 24) * 512 + 2 * 64 * 256 + CB_WBYTES + 64 * sizeof(float); }

int launch_conv_block_fwd(const CbFwdArgs& a, const RowSpace& rs, const DropCfg& drop, hipStream_t s) {
  HUAL_REQUIRE(a.x0 && rs.R > 0 && a.MT >= 1 && a.MT <= HUAL_CB_MAXMT, "conv_block_fwd: bad arguments");
  HUAL_REQUIRE(!a.pos || a.x0_out, "conv_block_fwd: pos needs x0_out");
  for (int l = 0; l < 4; ++l) {
    const CbLayerFwd& L = a.l[l];
    HUAL_REQUIRE(L.ln_g && L.ln_b && L.dw && L.wimg && L.bias && L.c && L.y && L.xout && L.mean && L.rstd, "conv_block_fwd: null layer tensor");
  }
  HUAL_DYN_LDS(conv_block_fwd_kernel, 160 * 1024);
  const double rows = (double)rs.R;
  // algorithmic work: 4 pointwise products; bytes: x0 in, (c, y, x) out per layer, weights
  HUAL_LAUNCH(4.0 * 2.0 * rows * HUAL_D * HUAL_D, 4.0 * (rows * HUAL_D * 13.0 + 4.0 * HUAL_D * HUAL_D), conv_block_fwd_kernel,
              dim3(cdiv(rs.R, a.MT)), dim3(CB_THREADS), cb_fwd_lds(a.MT), s, a, rs, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
