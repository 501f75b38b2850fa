// Fused conv_block kernels (see convblock.h).
//
// Forward, one workgroup (512 threads = 8 waves) per MT owned rows:
//   LDS  X   [MT+24][128] fp32     x_l of the rows r0-12 .. r0+MT+11 (updated in place layer by layer)
//        A   hi / lo planes        the GEMM operand c_l = depthwise7(LN(x_l)) pre-split into scaled fp16 pairs (bf16x3.h)
//        (the pointwise weights stay in REGISTERS: wave w keeps the fragments of its 16 output columns, loaded one layer ahead
//         from the fragment-major T image - tilecore.h "T-form"; no weight buffer in LDS)
//   per layer l (halo H_l = 9, 6, 3, 0 rows on either side of the owned rows):
//     P1  every 32-lane group slides a 7-row window over its chunk of rows: layer norm (shuffle reductions), depthwise
//         taps masked at clip boundaries (modules.py:66 zero padding), row scale, split, write A; c / mean / rstd of owned rows
//         go to HBM for backward
//     P2  wave w multiplies its 16 output columns (register-resident weight fragments) by every row tile of the workgroup:
//         activation fragments by ds_read_b128, 3 x v_mfma_f32_16x16x32_f16 per row tile and 32-deep k-step
//     P3  bias, relu (saved), Philox dropout, + x_l from LDS -> x_{l+1} back into X (+ HBM for the owned rows)
#include "convblock.h"
#include "tilecore.h"
#include "prof.h"

using namespace hual;

#include "lnproj_body.h"      // the tail of conv_block_fwd_lnproj_kernel

#define CB_XS 33          // float4 per X row: 528-byte rows spread the 128-byte column quarters of the statistics pass over the banks
#define CB_NPAR 10        // small parameter vectors of a layer staged in LDS: w[0..6], gamma, beta, bias
#define CB_TP_F4 (10 * 32) // float4 reserved for the tail's vectors (>= LNP_TP_VECS * 32)

// TAIL (round 5): the launch that follows a conv block in the step is the layer norm(s) + projections of the SAME rows (dual attention
// layer 0: LN1 / LN_t + five projections; the predictor's encoders: LN1 + dropout + query / key / value) - the block output is still in
// X, so the tile goes straight on (lnproj_body.h) instead of ending the launch and reading x_4 back: one launch and one memory round trip
// less per site
template <bool TAIL>
__device__ __forceinline__ void conv_block_fwd_body(const CbFwdArgs& a, const RowSpace& rs, const DropCfg& drop, const LnProjArgs* lp, char* cb_lds) {
  const int MT = a.MT;
  const int XR = MT + 24;                                   // rows of X
  float4* X = reinterpret_cast<float4*>(cb_lds);            // [XR][CB_XS] float4
  float4* TP = X + XR * CB_XS;                              // TAIL: [LNP_TP_VECS][32] float4, the tail's small vectors (lnproj_body.h)
  char* Ahi = reinterpret_cast<char*>(TP + (TAIL ? CB_TP_F4 : 0));      // [64][256 B]
  char* Alo = Ahi + 64 * 256;
  float* ainv = reinterpret_cast<float*>(Alo + 64 * 256);   // [64] inverse operand scale per A row
  float* smean = ainv + 64;                                 // [72] layer-norm statistics of the X rows
  float* srstd = smean + 72;                                // [72]
  float4* par = reinterpret_cast<float4*>(srstd + 72);      // [4][CB_NPAR][32] float4
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;          // 16 groups of 32 lanes: one row = 32 x float4
  const int col = 4 * l32;
  const int R = rs.R;
  const int tile_ = xcd_tile_clip(blockIdx.x, rs.R, rs.Nq > 0 ? rs.Nv : 0, MT);      // XCD-aware tile order by clips (common.h)
  if (tile_ < 0) return;
  const int r0 = tile_ * MT;
  if (r0 >= rs.R) return;
  const int xbase = r0 - 12;                                 // global row of X[0]

  const DropRegs dr = drop_load(drop);
  HUAL_STAMP_K(2, 0);
  // T-form (tilecore.h): wave `wave` owns output columns 16 wave .. 16 wave + 15 of all row tiles of a layer; its weight fragments
  // (T images) come straight from L2 into registers, a layer ahead
  TfW wc, wn;
  tf_load_w(wc, a.l[0].wimg, wave, lane);
  // ---- block input (+ position embeddings for the predictor's feature encoder) and the small parameters of all four
  // layers: everything is requested before anything is used - one memory round trip.  The loads are unconditional on
  // clamped rows (a load behind a lane-dependent branch is waited for inside that branch: one round trip per load)
  {
    float4 xv[5], pv[5], qv[4];
    const float* posp = a.pos ? a.pos : a.x0;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = grp + 16 * u, row = min(max(xbase + i, 0), R - 1);
      int lo, hi;
      cb_segment(row, rs, lo, hi);
      xv[u] = ld4(a.x0 + (size_t)row * HUAL_D + col);
      pv[u] = ld4(posp + (size_t)(a.pos ? row - lo : row) * HUAL_D + col);
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) {                            // group k < CB_NPAR stages parameter vector k of every layer
      const CbLayerFwd& L = a.l[l];
      const float* src = grp < 7 ? L.dw + grp * HUAL_D : (grp == 7 ? L.ln_g : (grp == 8 ? L.ln_b : L.bias));
      qv[l] = ld4(src + col);
    }
    float4 tq = f4zero();
    if (TAIL) {                                              // group k < LNP_TP_VECS: vector k of the tail (absent ones: any valid address)
      const float* src = lp->g1;
      if (grp == 1) src = lp->b1;
      if (grp == 2 && lp->g2) src = lp->g2;
      if (grp == 3 && lp->g2) src = lp->b2;
#pragma unroll
      for (int k = 0; k < HUAL_LNPROJ_MAX; ++k)
        if (grp == 4 + k && k < lp->nproj && lp->bias[k]) src = lp->bias[k];
      tq = ld4(src + col);
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = grp + 16 * u, row = xbase + i;
      if (i >= XR) continue;
      const bool ok = row >= 0 && row < R;
      float4 v = ok ? xv[u] : f4zero();
      if (a.pos) {
        v = ok ? cb_add(v, pv[u]) : v;
        if (row >= r0 && row < r0 + MT && row < R) st4(a.x0_out + (size_t)row * HUAL_D + col, v);
      }
      X[__mul24(i, CB_XS) + l32] = v;
    }
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (grp < CB_NPAR) par[(l * CB_NPAR + grp) * 32 + l32] = qv[l];
    if (TAIL && grp < LNP_TP_VECS) TP[grp * 32 + l32] = tq;
  }
  cb_barrier();
  HUAL_STAMP_K(2, 1);

#pragma unroll 1
  for (int l = 0; l < 4; ++l) {
    const CbLayerFwd& L = a.l[l];
    const int H = 9 - 3 * l;                                 // halo of this layer's OUTPUT rows
    const int nout = MT + 2 * H;                             // output rows r0-H .. r0+MT+H-1
    const int ntile = (nout + 15) >> 4;
    const int obase = r0 - H;                                // global row of output / operand row 0
    const int j = lane & 15, g = lane >> 4, ecol = 16 * wave + 4 * g;      // P2 / P3: accumulator rt of lane (j, g) = row 16 rt + j, columns ecol ..
    const float4* lp = par + l * CB_NPAR * 32;
    // ---------------- P1a: layer-norm statistics of the nout + 6 input rows, once per row.  Thread (row, quarter) owns 32
    // columns; the partial sums are associated exactly like the 32-lane butterfly of row_stats() in rowops.hip (lane =
    // float4 column group: xor 1, 2, 4 inside the thread, xor 8 / 16 across the four quarter threads), so the values are
    // bit-identical to the unfused kernels'
    {
      const int ri = threadIdx.x >> 2, q = threadIdx.x & 3;
      const int xi = ri + 9 - H;                             // X row of input row obase - 3 + ri
      const int t = xbase + xi;
      const bool ok = ri < nout + 6 && t >= 0 && t < R;      // (quad-uniform)
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ok ? X[__mul24(xi, CB_XS) + 8 * q + k] : f4zero();
      float p[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = cb_hsum(v[k]);
      float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
      s += dpp_xor_partner(s, 1);
      s += dpp_xor_partner(s, 2);
      const float mean = s * (1.0f / HUAL_D);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 d = make_float4(v[k].x - mean, v[k].y - mean, v[k].z - mean, v[k].w - mean);
        p[k] = cb_hsum(cb_mul(d, d));
      }
      s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
      s += dpp_xor_partner(s, 1);
      s += dpp_xor_partner(s, 2);
      const float rstd = rsqrtf(s * (1.0f / HUAL_D) + LN_EPS);
      if (ok && q == 0) {
        smean[xi] = mean;
        srstd[xi] = rstd;
        if (t >= r0 && t < r0 + MT) { L.mean[t] = mean; L.rstd[t] = rstd; }
      }
    }
    cb_barrier();
    HUAL_STAMP_K(2, 2 + 6 * l);
    // ---------------- P1b: depthwise conv over a sliding window of normalised rows -> operand planes
    if (l + 1 < 4) tf_load_w(wn, a.l[l + 1].wimg, wave, lane);      // the next layer's fragments (requested in front of this phase's stores)
    {
      const float4 gam = lp[7 * 32 + l32], bet = lp[8 * 32 + l32];
      float4 w[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) w[k] = lp[k * 32 + l32];
      // operand rows of this group (16 groups x ntile = all 16*ntile rows).  Balanced chunks of the nout live rows ((grp * nout) >> 4,
      // as in the backward's row phase, where they bought 3.8 us per step) were measured here too: +1.3 us on the same box - kept as is
      const int la = grp * ntile, lb = la + ntile;
      // the group's ntile + 6 input rows, normalised, in registers: every LDS read of the phase is issued before anything
      // depends on one (with two waves per SIMD a read per window step was a full LDS round trip per step)
      float4 hv[10];
      {
        float4 xr[10];
        float mr[10], sr[10];
#pragma unroll
        for (int s = 0; s < 10; ++s) {
          const int t = obase + la - 3 + s;
          const int xi = min(max(t - xbase, 0), XR - 1);
          xr[s] = X[__mul24(xi, CB_XS) + l32];      // (24-bit multiply: the 32-bit one is quarter rate)
          mr[s] = smean[xi];
          sr[s] = srstd[xi];
        }
#pragma unroll
        for (int s = 0; s < 10; ++s) {
          const int li = la - 3 + s, t = obase + li;
          const bool in = s < ntile + 6 && li < nout + 3 && t >= 0 && t < R;      // zero outside the tensor
          const float4 v = xr[s];
          const float mean = mr[s], rstd = sr[s];
          const float4 hn = cb_fma(make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd), gam, bet);
          hv[s] = in ? hn : f4zero();
        }
      }
      // clip segment of the group's first output row; the following rows step through it (one integer division per group)
      int slo, shi;
      cb_segment(min(max(obase + la, 0), R - 1), rs, slo, shi);
#pragma unroll
      for (int u = 0; u < 4; ++u) {                          // output row la + u: window hv[u] .. hv[u + 6]
        const int lo_ = la + u;
        if (lo_ >= lb) continue;
        const int o = obase + lo_;
        float4 c = f4zero();
        const bool live = lo_ < nout && o >= 0 && o < R;
        if (live) {
          if (o >= shi) { slo = shi; shi = slo + (slo < rs.Nv ? rs.T : rs.L); }
          // taps outside the clip are zero padding (SAME, modules.py:66); accumulation order of the taps: 0 .. 6
          c = cb_fma((o - 3 >= slo) ? hv[u] : f4zero(), w[0], c);
          c = cb_fma((o - 2 >= slo) ? hv[u + 1] : f4zero(), w[1], c);
          c = cb_fma((o - 1 >= slo) ? hv[u + 2] : f4zero(), w[2], c);
          c = cb_fma(hv[u + 3], w[3], c);
          c = cb_fma((o + 1 < shi) ? hv[u + 4] : f4zero(), w[4], c);
          c = cb_fma((o + 2 < shi) ? hv[u + 5] : f4zero(), w[5], c);
          c = cb_fma((o + 3 < shi) ? hv[u + 6] : f4zero(), w[6], c);
          if (o >= r0 && o < r0 + MT) st4_nt(L.c + (size_t)o * HUAL_D + col, c);
        }
        // (fixed operand scale: c is a 7-tap sum of layer-norm outputs; the per-row scale's maximum + butterfly cost 4.8 us per step here)
        const float inv = cb_store_operand_fx(Ahi, Alo, lo_, l32, c);
        if (l32 == 0) ainv[lo_] = live ? inv : 0.f;
      }
    }
    HUAL_STAMP_K(2, 3 + 6 * l);
    cb_barrier();                                         // operand planes complete
    HUAL_STAMP_K(2, 4 + 6 * l);
    // ---------------- P2: pointwise convolution on the matrix cores (three or four row tiles: H = 9, 6 / 3, 0 at 37 .. 46 owned rows)
    f32x4 acc[4];
    if (ntile > 3) tf_mma_lean<4, 64 * 256>(Ahi, wc, lane, acc);
    else tf_mma_lean<3, 64 * 256>(Ahi, wc, lane, reinterpret_cast<f32x4(&)[3]>(acc));
    HUAL_STAMP_K(2, 5 + 6 * l);
    // ---------------- P3: bias, relu, dropout, residual.  The relu active set and the dropout keep set of the owned rows leave
    // as bit planes (tilecore.h): all the backward pass needs of y_l
    {
      const float4 bias = lp[9 * 32 + (ecol >> 2)];
      const bool dropping = L.drop_site >= 0 && dr.enabled;      // (wave-uniform)
      int orow[4];
      bool own[4], live[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int lr = 16 * rt + j;
        orow[rt] = obase + lr;
        live[rt] = rt < ntile && lr < nout && orow[rt] >= 0 && orow[rt] < R;
        own[rt] = live[rt] && orow[rt] >= r0 && orow[rt] < r0 + MT;
      }
      uint32_t nib[4] = {15u, 15u, 15u, 15u};
      if (dropping) {      // two calls per lane for its 4 rows; the keep bytes of the owned rows go to the plane
        drop_nib2_store_t(dr, (uint32_t)L.drop_site, a.drop_row0, orow[0], orow[1], own[0], own[1], (uint32_t)(ecol >> 2), L.keep_bits, nib[0], nib[1], lane);
        drop_nib2_store_t(dr, (uint32_t)L.drop_site, a.drop_row0, orow[2], orow[3], own[2], own[3], (uint32_t)(ecol >> 2), L.keep_bits, nib[2], nib[3], lane);
      }
      uint32_t rb[4] = {0u, 0u, 0u, 0u};
      // the LDS reads per row (operand scale, residual row) are issued up front: the stores into X below would otherwise
      // order every later read behind them
      float irv[4];
      float4 xres[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        irv[rt] = ainv[16 * rt + j];
        xres[rt] = X[__mul24(min(max(orow[rt] - xbase, 0), XR - 1), CB_XS) + (ecol >> 2)];
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        if (!live[rt]) continue;
        const int o = orow[rt];
        const float ir = irv[rt];
        float4 v = make_float4(fmaf(acc[rt][0], ir, 0.f) + bias.x, fmaf(acc[rt][1], ir, 0.f) + bias.y,
                               fmaf(acc[rt][2], ir, 0.f) + bias.z, fmaf(acc[rt][3], ir, 0.f) + bias.w);
        v = relu_nan4(v);
        if (own[rt] && L.y) st4(L.y + (size_t)o * HUAL_D + ecol, v);
        rb[rt] = f4_posbits(v);
        if (dropping) v = f4_select(nib[rt], make_float4(v.x * dr.scale, v.y * dr.scale, v.z * dr.scale, v.w * dr.scale));
        v = cb_add(v, xres[rt]);
        X[__mul24(o - xbase, CB_XS) + (ecol >> 2)] = v;
        if (own[rt]) st4(L.xout + (size_t)o * HUAL_D + ecol, v);
      }
      bits_store2_t(L.relu_bits, orow[0], orow[1], own[0], own[1], ecol >> 2, rb[0], rb[1], lane);
      bits_store2_t(L.relu_bits, orow[2], orow[3], own[2], own[3], ecol >> 2, rb[2], rb[3], lane);
    }
    if (l + 1 < 4) wc = wn;
    cb_barrier();
    HUAL_STAMP_K(2, 7 + 6 * l);
  }
  if (TAIL) {      // x_4 of the owned rows is in X (rows r0 - xbase ..); the tail's operand slots take over everything from the operand planes on
    if (MT <= 32) ln_proj_body<2, true, true>(*lp, drop, Ahi, r0, X, xbase, CB_XS, XR, TP);
    else ln_proj_body<3, true, true>(*lp, drop, Ahi, r0, X, xbase, CB_XS, XR, TP);
  }
}

__global__ __launch_bounds__(CB_THREADS) void conv_block_fwd_kernel(CbFwdArgs a, RowSpace rs, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char cb_lds[];
  conv_block_fwd_body<false>(a, rs, drop, nullptr, cb_lds);
}
__global__ __launch_bounds__(CB_THREADS) void conv_block_fwd_lnproj_kernel(CbFwdArgs a, RowSpace rs, DropCfg drop, LnProjArgs lp) {
  extern __shared__ __attribute__((aligned(16))) char cb_lds[];
  conv_block_fwd_body<true>(a, rs, drop, &lp, cb_lds);
}

// ------------------------------------------------------------------------------------------------------
// Backward of the four layers in one launch.  Workgroup = MT owned rows (MT <= 40) + halo; per layer i = 3..0 with
// E1 = 3(i+1) halo rows of its input gradient and E0 = 3i of its output gradient:
//   LDS  DX  [MT+24][128] fp32   gradient wrt x_{i+1}, replaced in place by the gradient wrt x_i
//        A / DC (one region)     operand planes of dZ_i = dropout'(dx_{i+1}) * relu'(y_i) for the dX product, then its result
//                                dC_i = dZ_i . W_i^T as fp32 rows (the depthwise taps need it at t-3 .. t+3)
//        W   64 KB               image of W_i^T, requested while layer i+1 is in its row phase
//   G   wave (mt, ch): dC tile on the matrix cores (rows r0-E1 .. r0+MT+E1)
//   R   32-lane groups, <= 4 rows each (rows r0-E0 .. r0+MT+E0): transposed depthwise conv over a sliding window of dC,
//       layer-norm backward, + residual gradient -> DX; operand dZ_{i-1} of the next product; the parameter sums
//       (7 depthwise taps, gamma, beta) are taken over the OWNED rows only and leave as one partial row set per workgroup
//       (folded by colsum_kernel), dZ_i of the owned rows goes to HBM for the pointwise weight-gradient job.
// Same arithmetic as dwconv_ln_bwd_kernel + gemm_bf16_kernel for dx / dZ (bit for bit); the parameter sums are
// associated differently (per row instead of per tap position).
__global__ __launch_bounds__(CB_THREADS) void conv_block_bwd_kernel(CbBwdArgs a, RowSpace rs, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char cb_lds[];
  const int MT = a.MT;
  const int XR = MT + 24;
  float4* DX = reinterpret_cast<float4*>(cb_lds);            // [XR][32] float4
  char* Ahi = cb_lds + (size_t)XR * 512;                     // operand planes [64][256 B] x 2 ...
  char* Alo = Ahi + 64 * 256;
  float4* DC = reinterpret_cast<float4*>(Ahi);               // ... or dC as [64][32] float4 (same 32 KB)
  float* ainv = reinterpret_cast<float*>(Ahi + 64 * 512);    // [64]
  float4* par = reinterpret_cast<float4*>(ainv + 64);        // [9][32] float4: w[0..6], gamma, beta of the current layer
  float4* pbuf = par + 9 * 32;                               // [4][9][32] float4: per-wave parameter-gradient sums (one batch)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  const int R = rs.R;
  const int tile_ = xcd_tile_clip(blockIdx.x, rs.R, rs.Nq > 0 ? rs.Nv : 0, MT);      // XCD-aware tile order by clips (common.h)
  if (tile_ < 0) return;
  const int r0 = tile_ * MT;
  if (r0 >= rs.R) return;
  const int xbase = r0 - 12;

  const DropRegs dr = drop_load(drop);
  const float dscale3 = dr.enabled ? dr.scale : 1.0f;         // every dropout site of the block has the same rate
  HUAL_STAMP_K(3, 0);
  // T-form (tilecore.h): wave `wave` owns columns 16 wave .. 16 wave + 15 of all row tiles of dC; its fragments of W_i (N image) come
  // straight from L2 into registers - ONE set: the next layer's are requested right behind the product that frees it and arrive
  // under the row phase
  TfW wc;
  tf_load_w(wc, a.l[3].wimg_t, wave, lane);
  // ---- prologue: gradient wrt the block output -> DX; dZ_3 -> operand planes (+ HBM for the owned rows)
  {
    float4 dv[4];
    uint32_t zn[4];                                           // keep & relu' nibbles of layer 3 (bit planes of the forward)
#pragma unroll
    for (int u = 0; u < 4; ++u) {                             // unconditional loads on clamped rows, see the forward kernel
      const int i = grp + 16 * u, row = min(max(xbase + i, 0), R - 1);
      dv[u] = ld4(a.dx_in + (size_t)row * HUAL_D + col);
      // (both planes read unconditionally: a load behind the dropout branch would be waited for inside it, a round trip per row)
      const uint32_t kn = bits_nibble(a.keep_bits3, row, l32);
      zn[u] = bits_nibble(a.relu_bits3, row, l32) & (dr.enabled ? kn : 15u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = grp + 16 * u, row = xbase + i;
      if (i >= 64) continue;
      const bool ok = i < XR && row >= 0 && row < R;
      float4 v = ok ? dv[u] : f4zero();
      if (i < XR) DX[i * 32 + l32] = v;
      v = f4_select(ok ? zn[u] : 0u, make_float4(v.x * dscale3, v.y * dscale3, v.z * dscale3, v.w * dscale3));
      if (ok && row >= r0 && row < r0 + MT) st4(a.l[3].dz + (size_t)row * HUAL_D + col, v);
      const float inv = cb_store_operand(Ahi, Alo, i, l32, v);
      if (l32 == 0) ainv[i] = ok ? inv : 0.f;
    }
  }

  HUAL_STAMP_K(3, 1);
#pragma unroll 1
  for (int i = 3; i >= 0; --i) {
    const CbLayerBwd& L = a.l[i];
    const int E1 = 3 * (i + 1), E0 = 3 * i;
    const int nG = MT + 2 * E1, ntile = (nG + 15) >> 4, gbase = r0 - E1;      // dC rows
    const int nR = MT + 2 * E0, rbase = r0 - E0;                               // rows of the row phase
    const int ra = (grp * nR) >> 4, rb = ((grp + 1) * nR) >> 4;                // balanced chunks of <= 4 rows per group: every SIMD carries the same number of rows (122.6 -> 118.9 us per step)
    // the layer's small parameters go through LDS (9 x 512 B; group k stages vector k): registers are scarce in the row phase.
    // One unconditional load through a selected pointer (groups 9 .. 15 read a vector they do not store): a load behind the
    // lane-dependent branch would be waited for inside it.
    const float4 pv = ld4((grp < 7 ? L.dw + grp * HUAL_D : (grp == 7 ? L.ln_g : L.ln_b)) + col);
    // ---- operands of the row phase, requested before the matrix phase: their latency hides under it (unconditional loads on clamped
    // rows; plain loads now that no hand-placed wait for an LDS-DMA stands between them and their use)
    float4 xv[4];
    float mu[4], rsd[4];
    uint32_t zb[4], zk[4];                                    // relu' / keep nibbles of layer i-1 for this group's rows
    const uint8_t* rprev = L.relu_prev ? L.relu_prev : a.relu_bits3;      // layer 0 has no layer below it: the loaded bytes are ignored
    const uint8_t* kprev = L.keep_prev ? L.keep_prev : a.keep_bits3;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tc = min(max(rbase + ra + u, 0), R - 1);
      xv[u] = ld4(L.x + (size_t)tc * HUAL_D + col);
      zk[u] = kprev[(size_t)tc * 16 + (l32 >> 1)];
      zb[u] = rprev[(size_t)tc * 16 + (l32 >> 1)];
      mu[u] = L.mean[tc];
      rsd[u] = L.rstd[tc];
    }
    if (grp < 9) par[grp * 32 + l32] = pv;
    HUAL_STAMP_K(3, 2 + 7 * (3 - i));
    cb_barrier();                                          // (1) operand planes + parameters complete
    HUAL_STAMP_K(3, 3 + 7 * (3 - i));
    // ---- G: dC_i = dZ_i . W_i^T (three or four row tiles)
    f32x4 acc[4];
    if (ntile > 3) tf_mma_lean<4, 64 * 256>(Ahi, wc, lane, acc);
    else tf_mma_lean<3, 64 * 256>(Ahi, wc, lane, reinterpret_cast<f32x4(&)[3]>(acc));
    if (i > 0) tf_load_w(wc, a.l[i - 1].wimg_t, wave, lane);      // the next layer's fragments: under the row phase
    float irv[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) irv[rt] = ainv[16 * rt + (lane & 15)];
    cb_barrier();                                          // (2) every wave is through the planes: they become dC
    HUAL_STAMP_K(3, 4 + 7 * (3 - i));
    {
      // accumulator rt of lane (j, g) = row 16 rt + j, columns 16 wave + 4 g .. + 3.  The 16 lanes j of a store are 16 rows of one
      // column group: the float4 column index is XOR-swizzled with 2 (row & 15) - 16 different 16-byte slots - and the row phase
      // reads a row through the same permutation (still one contiguous 512 B per group)
      const int j = lane & 15, g = lane >> 4;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        if (rt >= ntile) continue;
        const int lr = 16 * rt + j;
        const float ir = irv[rt];
        DC[lr * 32 + ((4 * wave + g) ^ (2 * j))] = make_float4(acc[rt][0] * ir, acc[rt][1] * ir, acc[rt][2] * ir, acc[rt][3] * ir);
      }
    }
    cb_barrier();                                          // (3) dC visible
    // keep & relu' of dZ_{i-1}, 4 bits per row (bit c = column col + c), rows packed into one register
    uint32_t zbits = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t sh = 4 * (l32 & 1);
      zbits |= (((zb[u] >> sh) & 15u) & (dr.enabled ? ((zk[u] >> sh) & 15u) : 15u)) << (4 * u);
    }
    HUAL_STAMP_K(3, 5 + 7 * (3 - i));
    // ---- R: transposed depthwise conv + layer-norm backward
    float4 sw[7], sg = f4zero(), sb = f4zero();
#pragma unroll
    for (int k = 0; k < 7; ++k) sw[k] = f4zero();
    {
      float4 d0 = f4zero(), d1 = f4zero(), d2 = f4zero(), d3 = f4zero(), d4 = f4zero(), d5 = f4zero(), d6 = f4zero();
      // window rows tt = first-3 .. last+3 ; after pushing tt the window is centred on t = tt - 3
#pragma unroll
      for (int s = 0; s < 10; ++s) {
        const int tt = rbase + ra - 3 + s;
        const int di = tt - gbase;
        float4 dn = f4zero();
        if (s < (rb - ra) + 6 && di >= 0 && di < nG && tt >= 0 && tt < R) dn = DC[di * 32 + (l32 ^ (2 * (di & 15)))];
        d0 = d1; d1 = d2; d2 = d3; d3 = d4; d4 = d5; d5 = d6; d6 = dn;
        if (s < 6) continue;
        const int u = s - 6;                                  // compile-time index of this group's row
        const int lr = ra + u, t = rbase + lr;
        if (!(lr < rb && t >= 0 && t < R)) continue;
        int slo, shi;
        cb_segment(t, rs, slo, shi);
        // dh[t] = sum_k dC[t - k + 3] * w[k] over the taps whose source position lies in the clip (order k = 0..6)
        const float4 e0 = (t + 3 < shi) ? d6 : f4zero(), e1 = (t + 2 < shi) ? d5 : f4zero(), e2 = (t + 1 < shi) ? d4 : f4zero();
        const float4 e4 = (t - 1 >= slo) ? d2 : f4zero(), e5 = (t - 2 >= slo) ? d1 : f4zero(), e6 = (t - 3 >= slo) ? d0 : f4zero();
        float4 dh = f4zero();
        dh = cb_fma(e0, par[0 * 32 + l32], dh); dh = cb_fma(e1, par[1 * 32 + l32], dh); dh = cb_fma(e2, par[2 * 32 + l32], dh);
        dh = cb_fma(d3, par[3 * 32 + l32], dh);
        dh = cb_fma(e4, par[4 * 32 + l32], dh); dh = cb_fma(e5, par[5 * 32 + l32], dh); dh = cb_fma(e6, par[6 * 32 + l32], dh);
        const float4 gam = par[7 * 32 + l32];
        const float4 v = xv[u];
        const float mean = mu[u], rstd = rsd[u];
        const float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
        if (t >= r0 && t < r0 + MT) {                         // parameter gradients: owned rows only
          const float4 h = cb_fma(xh, gam, par[8 * 32 + l32]);
          sw[0] = cb_fma(h, e0, sw[0]); sw[1] = cb_fma(h, e1, sw[1]); sw[2] = cb_fma(h, e2, sw[2]); sw[3] = cb_fma(h, d3, sw[3]);
          sw[4] = cb_fma(h, e4, sw[4]); sw[5] = cb_fma(h, e5, sw[5]); sw[6] = cb_fma(h, e6, sw[6]);
          sb = cb_add(sb, dh);
          sg = cb_fma(dh, xh, sg);
        }
        const float4 gv = cb_mul(dh, gam);
        const float m1 = fast_sum32(cb_hsum(gv)) * (1.0f / HUAL_D);
        const float m2 = fast_sum32(cb_hsum(cb_mul(gv, xh))) * (1.0f / HUAL_D);
        float4 dx = make_float4(rstd * (gv.x - m1 - xh.x * m2), rstd * (gv.y - m1 - xh.y * m2),
                                rstd * (gv.z - m1 - xh.z * m2), rstd * (gv.w - m1 - xh.w * m2));
        float4* dxp = DX + (t - xbase) * 32 + l32;
        // + gradient through the residual connection; __fadd_rn keeps the product above and this sum two separately
        // rounded operations, as in dwconv_ln_bwd_kernel (where the sum sits behind a branch and is never contracted)
        const float4 dres = *dxp;
        dx = make_float4(__fadd_rn(dx.x, dres.x), __fadd_rn(dx.y, dres.y), __fadd_rn(dx.z, dres.z), __fadd_rn(dx.w, dres.w));
        *dxp = dx;
        if (i == 0) st4(a.dx_out + (size_t)t * HUAL_D + col, dx);
        // (i > 0: the operand of the next product, dZ_{i-1} = dropout'(dx_i) * relu'(y_{i-1}), is formed behind the parameter sums
        //  from DX and the keep & relu' bits the forward left)
      }
    }
    HUAL_STAMP_K(3, 6 + 7 * (3 - i));
    // workgroup sums of the parameter gradients, without atomics (an LDS float atomic was measured at ~600 stall cycles per
    // wave-instruction here): the two 32-lane groups of a wave are added in registers (v_permlane32_swap), the 8 per-wave
    // partials go through LDS in two batches of four, and thread e owns sums e, e + 512, e + 1024 of the 9 x 128
    {
      auto xor32_sum = [&](float v) {
        const unsigned x = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return v + __builtin_bit_cast(float, (lane & 32) ? r[0] : r[1]);
      };
      auto xor32_sum4 = [&](float4 v) { return make_float4(xor32_sum(v.x), xor32_sum(v.y), xor32_sum(v.z), xor32_sum(v.w)); };
#pragma unroll
      for (int k = 0; k < 7; ++k) sw[k] = xor32_sum4(sw[k]);
      sg = xor32_sum4(sg);
      sb = xor32_sum4(sb);
      float pacc[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int batch = 0; batch < 2; ++batch) {
        if ((wave >> 2) == batch && lane < 32) {
          float4* dst = pbuf + (wave & 3) * 9 * 32 + l32;
#pragma unroll
          for (int k = 0; k < 7; ++k) dst[k * 32] = sw[k];
          dst[7 * 32] = sg;
          dst[8 * 32] = sb;
        }
        cb_barrier();                                      // (4) on the first pass: dC consumed as well
        const float* pf = reinterpret_cast<const float*>(pbuf);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int e = threadIdx.x + CB_THREADS * q;
          if (e < 9 * HUAL_D) pacc[q] += (pf[e] + pf[e + 9 * HUAL_D]) + (pf[e + 2 * 9 * HUAL_D] + pf[e + 3 * 9 * HUAL_D]);
        }
        cb_barrier();
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int e = threadIdx.x + CB_THREADS * q;
        if (e < 9 * HUAL_D) L.part[(size_t)tile_ * 9 * HUAL_D + e] = pacc[q];
      }
    }
    HUAL_STAMP_K(3, 7 + 7 * (3 - i));
    if (i > 0) {
      // operand rows of the next product: index 0 = global row r0 - E0 (= rbase); the rows between nR and the next 16-row tile edge keep
      // what they held (their dC rows are never read: the next layer's nG is this nR)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int lr = ra + u, t = rbase + lr;
        if (lr >= rb) continue;
        const bool ok = lr < rb && t >= 0 && t < R;
        float4 v = f4zero();
        if (ok) {
          v = cb_mul(DX[(t - xbase) * 32 + l32], mask_from_bits4((zbits >> (4 * u)) & 0xfu, dscale3));
        }
        if (ok && t >= r0 && t < r0 + MT) st4(L.dz_prev + (size_t)t * HUAL_D + col, v);
        const float inv = cb_store_operand(Ahi, Alo, lr, l32, v);
        if (l32 == 0) ainv[lr] = ok ? inv : 0.f;
      }
    }
    HUAL_STAMP_K(3, 8 + 7 * (3 - i));
  }
}

#if defined(HUAL_STAMPS) && (HUAL_STAMPS == 2 || HUAL_STAMPS == 3 || HUAL_STAMPS == 7)
extern "C" int hual_debug_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hual_stamps), sizeof(unsigned long long) * (size_t)n);
}
extern "C" int hual_debug_stamps_reset() {
  static unsigned long long zeros[512 * HUAL_STAMP_SLOTS];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_hual_stamps), zeros, sizeof(zeros));
}
#endif

namespace hual {

int conv_block_fused_rows(int R, int Nv) {
  // as many workgroups as there are CUs when the rows allow it (each workgroup streams all four weight images, so fewer,
  // taller workgroups cost nothing extra), never more than HUAL_CB_MAXMT rows: the halo'd operand must fit 64 rows
  return xcd_clip_rows(R, Nv, 16, HUAL_CB_MAXMT);
}

static size_t cb_fwd_lds(int MT) { return (size_t)(MT + 24) * CB_XS * 16 + 2 * 64 * 256 + (64 + 72 + 72) * sizeof(float) + 4 * CB_NPAR * 512; }

int launch_conv_block_fwd(const CbFwdArgs& a, const RowSpace& rs, const DropCfg& drop, hipStream_t s, const LnProjArgs* tail) {
  HUAL_REQUIRE(a.x0 && rs.R > 0 && a.MT >= 1 && a.MT <= HUAL_CB_MAXMT, "conv_block_fwd: bad arguments");
  HUAL_REQUIRE(!a.pos || a.x0_out, "conv_block_fwd: pos needs x0_out");
  for (int l = 0; l < 4; ++l) {
    const CbLayerFwd& L = a.l[l];
    HUAL_REQUIRE(L.ln_g && L.ln_b && L.dw && L.wimg && L.bias && L.c && L.xout && L.mean && L.rstd && L.relu_bits && L.keep_bits, "conv_block_fwd: null layer tensor");
  }
  const double rows = (double)rs.R;
  const dim3 grid(xcd_clip_grid(rs.R, rs.Nq > 0 ? rs.Nv : 0, a.MT));
  // algorithmic work: 4 pointwise products; bytes: x0 in, (c, y, x) out per layer, weights
  double flops = 4.0 * 2.0 * rows * HUAL_D * HUAL_D, bytes = 4.0 * (rows * HUAL_D * 13.0 + 4.0 * HUAL_D * HUAL_D);
  if (!tail) {
    HUAL_DYN_LDS(conv_block_fwd_kernel, 160 * 1024);
    HUAL_LAUNCH(flops, bytes, conv_block_fwd_kernel, grid, dim3(CB_THREADS), cb_fwd_lds(a.MT), s, a, rs, drop);
  } else {
    // the tail works on this launch's tiles: the block output as its layer-norm input, rows straight from LDS
    LnProjArgs lp = *tail;
    lp.MT = a.MT;
    int rc = check_ln_proj_args(lp);
    if (rc) return rc;
    HUAL_REQUIRE(lp.x == a.l[3].xout && lp.R == rs.R && ln_proj_plain(lp), "conv_block_fwd: the tail must read the block output and be of the plain shape");
    const size_t tail_from_planes = std::max((size_t)LN_PROJ_LDS, cb_fwd_lds(a.MT) - (size_t)(a.MT + 24) * CB_XS * 16);
    const size_t lds = (size_t)(a.MT + 24) * CB_XS * 16 + CB_TP_F4 * 16 + tail_from_planes;
    HUAL_REQUIRE(lds <= 160 * 1024, "conv_block_fwd: LDS of the tail");
    flops += 2.0 * rows * HUAL_D * HUAL_D * lp.nproj;
    bytes += 4.0 * rows * HUAL_D * (lp.nproj + 1.0 + (lp.g2 ? 1.0 : 0.0)) - 4.0 * rows * HUAL_D;      // (x_4 is not read back)
    HUAL_DYN_LDS(conv_block_fwd_lnproj_kernel, 160 * 1024);
    HUAL_LAUNCH(flops, bytes, conv_block_fwd_lnproj_kernel, grid, dim3(CB_THREADS), lds, s, a, rs, drop, lp);
  }
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int conv_block_fused_rows_bwd(int R, int Nv) { return xcd_clip_rows(R, Nv, 16, HUAL_CB_BWD_MAXMT); }
int conv_block_bwd_blocks(int R, int Nv) { return cdiv(R, conv_block_fused_rows_bwd(R, Nv)); }

static size_t cb_bwd_lds(int MT) { return (size_t)(MT + 24) * 512 + 64 * 512 + 64 * sizeof(float) + (1 + 4) * 9 * HUAL_D * sizeof(float); }

int launch_conv_block_bwd(const CbBwdArgs& a, const RowSpace& rs, const DropCfg& drop, hipStream_t s) {
  HUAL_REQUIRE(a.dx_in && a.relu_bits3 && a.keep_bits3 && a.dx_out && rs.R > 0 && a.MT >= 1 && a.MT <= HUAL_CB_BWD_MAXMT, "conv_block_bwd: bad arguments");
  for (int l = 0; l < 4; ++l) {
    const CbLayerBwd& L = a.l[l];
    HUAL_REQUIRE(L.ln_g && L.ln_b && L.dw && L.wimg_t && L.x && L.mean && L.rstd && L.dz && L.part, "conv_block_bwd: null layer tensor");
    HUAL_REQUIRE(l == 0 || (L.relu_prev && L.keep_prev && L.dz_prev), "conv_block_bwd: layers 1-3 need the bit planes of the layer below / dz_prev");
  }
  HUAL_DYN_LDS(conv_block_bwd_kernel, 160 * 1024);
  const double rows = (double)rs.R;
  HUAL_LAUNCH(4.0 * 2.0 * rows * HUAL_D * HUAL_D, 4.0 * (rows * HUAL_D * 14.0 + 4.0 * HUAL_D * HUAL_D), conv_block_bwd_kernel,
              dim3(xcd_clip_grid(rs.R, rs.Nq > 0 ? rs.Nv : 0, a.MT)), dim3(CB_THREADS), cb_bwd_lds(a.MT), s, a, rs, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
