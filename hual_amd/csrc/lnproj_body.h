// The ln_proj row-tile body (dablock.h: layer norm(s) of a tile of rows + the projections that read them), shared by ln_proj_kernel
// (dablock.hip) and - round 5 - by the tail of conv_block_fwd_kernel (convblock.hip: the block output is still in LDS when the next
// launch would read it back; FUSED = true takes the rows from there).  Same arithmetic, bit for bit, wherever it runs.
#pragma once
#include "dablock.h"
#include "tilecore.h"

// layer norm of one row held as one float4 per lane of a 32-lane group (models/layers.py:7-17; the butterfly of row_stats())
__device__ __forceinline__ float4 ln_row(float4 v, float& mean, float& rstd) {
  mean = fast_sum32(cb_hsum(v)) * (1.0f / HUAL_D);
  const float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
  const float var = fast_sum32(cb_hsum(cb_mul(d, d))) * (1.0f / HUAL_D);
  rstd = rsqrtf(var + LN_EPS);
  return make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
}

#define LP_ROWS 64
#define LN_PROJ_LDS ((size_t)4 * LP_ROWS * 256 + 2 * LP_ROWS * sizeof(float))      // two operand slots (hi + lo planes) + their row scales
// FUSED: the tile is rows [r0f, r0f + a.MT) and its layer-norm input sits in LDS - row t at xl[(t - xl_row0) * xl_stride + float4 column],
// xl_rows rows (conv_block_fwd_kernel's X) - instead of a.x; a.xa / a.x2 are not supported there.  The small vectors were staged by the
// host kernel's prologue (a global load here would be waited for at once): tp[k][32] float4, k = g1, b1, g2, b2, bias[0 .. 4] (LNP_TP_*)
#define LNP_TP_VECS 9
// PLAIN (host: ln_proj_plain()): the query / key / value shape - no residual input (xa), no raw second operand, every projection a plain
// "product + bias -> store" (no relu / dropout / residual / K-concatenation).  The generic body evaluates those options as selects on
// uniform flags, and these phases are bound by VALU issue (two waves per SIMD, ~1500 instructions per wave in the row phase alone)
template <int NT, bool FUSED = false, bool PLAIN = false>
__device__ __forceinline__ void ln_proj_body(const hual::LnProjArgs& a, const hual::DropCfg& drop, char* lp_lds, const int r0f = 0,
                                             const float4* xl = nullptr, const int xl_row0 = 0, const int xl_stride = 0, const int xl_rows = 0, const float4* tp = nullptr) {
  char* P1 = lp_lds;                                  // hi plane [64][256 B] | lo plane
  char* P2 = P1 + 2 * LP_ROWS * 256;
  float* ainv1 = reinterpret_cast<float*>(P2 + 2 * LP_ROWS * 256);
  float* ainv2 = ainv1 + LP_ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  const int MT = a.MT, R = a.R;
  int r0 = r0f;
  if (!FUSED) {
    const int tile_ = xcd_tile_clip(blockIdx.x, R, a.Nv, MT);
    if (tile_ < 0) return;
    r0 = tile_ * MT;                                // (grid rounded up to whole XCD rounds)
  }
  const int RE = min(R, r0 + MT);             // rows [r0, RE) belong to this workgroup (MT need not be a multiple of 16)
  const int j = lane & 15, g = lane >> 4, ecol = 16 * wave + 4 * g;
  const DropRegs dr = drop_load(drop);

  // every operand of the kernel is requested up front: the first projection's weight fragments, rows (unconditional loads on
  // clamped rows), layer-norm parameters, biases
  TfW w[2];
  tf_load_w(w[0], a.wimg[0], wave, lane);
  float4 xv[4], av[4], rv[4], bias[HUAL_LNPROJ_MAX];
  const float* xap = a.xa ? a.xa : a.x;
  const float* x2p = a.x2 ? a.x2 : a.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (FUSED) {
      xv[u] = xl[__mul24(min(max(r0 + grp + 16 * u - xl_row0, 0), xl_rows - 1), xl_stride) + l32];
      av[u] = xv[u];
      rv[u] = xv[u];
    } else {
      const size_t off = (size_t)min(r0 + grp + 16 * u, R - 1) * HUAL_D + col;
      xv[u] = ld4(a.x + off);
      av[u] = PLAIN ? xv[u] : ld4(xap + off);
      rv[u] = PLAIN ? xv[u] : ld4(x2p + off);
    }
  }
  float4 g1, b1, g2, b2;
  if (FUSED) {
    g1 = tp[l32]; b1 = tp[32 + l32];
    g2 = f4_pick(a.g2 != nullptr, tp[64 + l32], f4zero()); b2 = f4_pick(a.g2 != nullptr, tp[96 + l32], f4zero());
#pragma unroll
    for (int p = 0; p < HUAL_LNPROJ_MAX; ++p) bias[p] = f4_pick(p < a.nproj && a.bias[p], tp[(4 + p) * 32 + (ecol >> 2)], f4zero());
  } else {
    g1 = ld4(a.g1 + col); b1 = ld4(a.b1 + col);
    g2 = a.g2 ? ld4(a.g2 + col) : f4zero(); b2 = a.g2 ? ld4(a.b2 + col) : f4zero();
#pragma unroll
    for (int p = 0; p < HUAL_LNPROJ_MAX; ++p) bias[p] = (p < a.nproj && a.bias[p]) ? ld4(a.bias[p] + ecol) : f4zero();
  }
  float4* scratch = reinterpret_cast<float4*>(P2);           // fp32 rows [64][32] float4 of the layer-norm input (residual)
  // dropout decisions of the two row-layout sites for this group's four rows (two rows per call, tilecore.h); the keep bytes
  // go to the bit planes the backward pass reads
  uint32_t nbp[4] = {15u, 15u, 15u, 15u}, nb1[4] = {15u, 15u, 15u, 15u};
  if (dr.enabled) {
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int lrA = grp + 32 * pr, lrB = lrA + 16;
      const bool okA = lrA < MT && r0 + lrA < RE, okB = lrB < MT && r0 + lrB < RE;
      if (a.xa && a.pre_site >= 0)
        drop_nib2_store_r(dr, (uint32_t)a.pre_site, a.drop_row0, r0 + lrA, r0 + lrB, okA, okB, (uint32_t)l32, a.pre_bits, nbp[2 * pr], nbp[2 * pr + 1]);
      if (a.drop_site1 >= 0)
        drop_nib2_store_r(dr, (uint32_t)a.drop_site1, a.drop_row0, r0 + lrA, r0 + lrB, okA, okB, (uint32_t)l32, a.y1_bits, nb1[2 * pr], nb1[2 * pr + 1]);
    }
  }
  // ---- (residual) + layer norm(s) -> operand planes.  Straight-line code over the NT row tiles: every global store is a range-checked
  // buffer store (common.h: rows that are not this tile's get the offset ROW_SKIP and are dropped by the hardware, absent tensors a
  // resource of zero bytes) - a branch around a store made the compiler drain every outstanding load and store at the next wait
  // (scripts/exp/isa_vmcnt0.py: one full drain per row here, three per projection in the epilogues below)
  const uint32_t rbytes = (uint32_t)R * (uint32_t)(HUAL_D * 4);
  const __amdgpu_buffer_rsrc_t rs_xo = row_rsrc(a.x_out, a.xa ? rbytes : 0u), rs_y1 = row_rsrc(a.y1, rbytes), rs_y2 = row_rsrc(a.y2, a.g2 ? rbytes : 0u);
  const __amdgpu_buffer_rsrc_t rs_mean = row_rsrc(a.mean, (uint32_t)R * 4u), rs_rstd = row_rsrc(a.rstd, (uint32_t)R * 4u);
  const bool has_xa = a.xa != nullptr, pre_drop = a.pre_site >= 0 && dr.enabled, drop1 = a.drop_site1 >= 0 && dr.enabled;
  const bool second = a.g2 || a.x2, ln2 = a.g2 != nullptr;
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int lr = grp + 16 * u, row = r0 + lr;
    if (__builtin_amdgcn_readfirstlane(2 * wave + 16 * u) >= MT) {      // (wave-uniform, no vector-memory operation inside) both row groups of the wave lie beyond the tile
      const float i0 = cb_store_operand_fx(P1, P1 + LP_ROWS * 256, lr, l32, f4zero());
      if (l32 == 0) ainv1[lr] = 0.f * i0;
      if (second) { cb_store_operand_fx(P2, P2 + LP_ROWS * 256, lr, l32, f4zero()); if (l32 == 0) ainv2[lr] = 0.f; }
      continue;
    }
    const bool ok = lr < MT && row < RE;
    const uint32_t roff = ok ? (uint32_t)row * (uint32_t)(HUAL_D * 4) + (uint32_t)col * 4u : ROW_SKIP;
    const uint32_t soff = (ok && l32 == 0) ? (uint32_t)row * 4u : ROW_SKIP;
    float4 xr = xv[u];
    if (!PLAIN) {
      float4 t = av[u];
      const float4 td = f4_select(nbp[u], make_float4(t.x * dr.scale, t.y * dr.scale, t.z * dr.scale, t.w * dr.scale));
      t = f4_pick(pre_drop, td, t);
      xr = f4_pick(has_xa, cb_add(t, xr), xr);
      bst4(rs_xo, roff, xr);
    }
    float mean, rstd;
    const float4 xh = ln_row(xr, mean, rstd);
    float4 y1 = cb_fma(xh, g1, b1);
    {
      const float4 yd = f4_select(nb1[u], make_float4(y1.x * dr.scale, y1.y * dr.scale, y1.z * dr.scale, y1.w * dr.scale));
      y1 = f4_pick(drop1, yd, y1);
    }
    y1 = f4_pick(ok, y1, f4zero());
    bst4(rs_y1, roff, y1);
    bst1(rs_mean, soff, mean);
    bst1(rs_rstd, soff, rstd);
    const float i1 = cb_store_operand_fx(P1, P1 + LP_ROWS * 256, lr, l32, y1);      // layer-norm output: fixed operand scale (tilecore.h)
    if (l32 == 0) ainv1[lr] = ok ? i1 : 0.f;
    if (second) {      // (uniform; LDS stores only)
      const float4 y2 = f4_pick(ok, (PLAIN ? cb_fma(xh, g2, b2) : f4_pick(ln2, cb_fma(xh, g2, b2), rv[u])), f4zero());
      const float i2 = (PLAIN || ln2) ? cb_store_operand_fx(P2, P2 + LP_ROWS * 256, lr, l32, y2) : cb_store_operand(P2, P2 + LP_ROWS * 256, lr, l32, y2);
      if (l32 == 0) ainv2[lr] = ok ? i2 : 0.f;
    } else if (!PLAIN) {
      scratch[lr * 32 + l32] = xr;
    }
    bst4(rs_y2, roff, f4_pick(ok, cb_fma(xh, g2, b2), f4zero()));      // (zero-byte resource without a second layer norm)
  }
  // ---- the projections.  T-form (tilecore.h): wave `wave` owns output columns 16 wave .. 16 wave + 15 of all NT row tiles and holds
  // its weight fragments (T images, straight from L2, requested a projection ahead) in registers; the operand planes are never
  // rewritten, so ONE barrier serves all projections.  Accumulator rt of lane (j, g) = row 16 rt + j, columns 16 wave + 4 g .. + 3
  if (FUSED) HUAL_STAMP_K(2, 26);
  float4 acc[NT];
  f32x4 accp[NT];
  bool fresh = true;
  TfA<NT <= 2 ? NT : 1> xa;
  int xa_src = -1;
#pragma unroll
  for (int p = 0; p < HUAL_LNPROJ_MAX; ++p) {
    if (p >= a.nproj) break;                                 // uniform
    tf_load_w_if(w[(p + 1) & 1], a.wimg[p + 1 < a.nproj ? p + 1 : p], p + 1 < a.nproj, wave, lane);      // (straight-line: no traffic behind the last one)
    if (p == 0) cb_barrier();
    if (FUSED && p == 0) HUAL_STAMP_K(2, 27);
    const char* P = a.src[p] ? P2 : P1;
    const float* ai = a.src[p] ? ainv2 : ainv1;
    if constexpr (NT <= 2) {      // fragments of the operand slot stay in registers while consecutive projections read the same one (tilecore.h TfA;
                                  // at three row tiles the up-front read of all four k-steps cost more than the re-reads: +1.5 us per launch)
      if (a.src[p] != xa_src) { tf_load_a<NT, LP_ROWS * 256>(xa, P, lane); xa_src = a.src[p]; }      // (uniform)
      tf_mma_regs<NT>(xa, w[p & 1], accp);
    } else {
      tf_mma_lean<NT, LP_ROWS * 256>(P, w[p & 1], lane, accp);
    }
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const float ir = ai[16 * rt + j];
      if (PLAIN || fresh) acc[rt] = make_float4(fmaf(accp[rt][0], ir, 0.f), fmaf(accp[rt][1], ir, 0.f), fmaf(accp[rt][2], ir, 0.f), fmaf(accp[rt][3], ir, 0.f));
      else acc[rt] = make_float4(fmaf(accp[rt][0], ir, acc[rt].x), fmaf(accp[rt][1], ir, acc[rt].y), fmaf(accp[rt][2], ir, acc[rt].z), fmaf(accp[rt][3], ir, acc[rt].w));
    }
    const __amdgpu_buffer_rsrc_t rs_out = row_rsrc(a.out[p], ((uint32_t)(R - 1) * (uint32_t)a.ldo[p] + (uint32_t)HUAL_D) * 4u);
    if (PLAIN) {
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) {
        const int row = r0 + 16 * rt + j;
        const float4 v = make_float4(acc[rt].x + bias[p].x, acc[rt].y + bias[p].y, acc[rt].z + bias[p].z, acc[rt].w + bias[p].w);
        bst4(rs_out, row < RE ? ((uint32_t)row * (uint32_t)a.ldo[p] + (uint32_t)ecol) * 4u : ROW_SKIP, v);
      }
      if (FUSED) HUAL_STAMP_K(2, 28 + p);
      continue;
    }
    fresh = a.accum[p] == 0;
    if (a.accum[p]) continue;
    uint32_t nbo[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) nbo[rt] = 15u;
    const bool dropo = a.out_site[p] >= 0 && dr.enabled;
    if (dropo) drop_rows_t<NT>(dr, (uint32_t)a.out_site[p], a.drop_row0, r0 + j, RE, (uint32_t)(ecol >> 2), a.out_bits[p], nbo, lane);
    const bool relu = a.act[p] != 0, addx = a.add_x[p] != 0;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const int lr = 16 * rt + j, row = r0 + lr;
      float4 v = make_float4(acc[rt].x + bias[p].x, acc[rt].y + bias[p].y, acc[rt].z + bias[p].z, acc[rt].w + bias[p].w);
      const float4 vr = relu_nan4(v);
      v = f4_pick(relu, vr, v);
      const float4 vd = f4_select(nbo[rt], make_float4(v.x * dr.scale, v.y * dr.scale, v.z * dr.scale, v.w * dr.scale));
      v = f4_pick(dropo, vd, v);
      const float4 vx = cb_add(v, scratch[lr * 32 + (ecol >> 2)]);
      v = f4_pick(addx, vx, v);
      bst4(rs_out, row < RE ? ((uint32_t)row * (uint32_t)a.ldo[p] + (uint32_t)ecol) * 4u : ROW_SKIP, v);
    }
    if (FUSED) HUAL_STAMP_K(2, 28 + p);
  }
}
