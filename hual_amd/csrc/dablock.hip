// Fused row-local chains of the dual attention block (see dablock.h).  Built from the pieces of tilecore.h: operand planes
// in LDS, one weight image at a time by LDS-DMA (requested as soon as the previous image has been consumed, so that it
// lands under the epilogue), wave (mt, ch) = 16 rows x 64 columns on the matrix cores, epilogue tiles in registers.
#include "dablock.h"
#include "tilecore.h"
#include "prof.h"

using namespace hual;

// maximum over the 16 lanes that share lane >> 4 (one DPP row): the lanes holding the 64 columns of a tile row
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_xor_partner(v, 1));
  v = fmaxf(v, dpp_xor_partner(v, 2));
  v = fmaxf(v, dpp_xor_partner(v, 4));
  v = fmaxf(v, dpp_xor_partner(v, 8));
  return v;
}
__device__ __forceinline__ float4 sig4(float4 v) { return make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)); }

#include "lnproj_body.h"      // ln_row(), ln_proj_body<NT>

template <int NT, bool PLAIN>
__global__ __launch_bounds__(CB_THREADS) void ln_proj_kernel(LnProjArgs a, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char lp_lds[];
  ln_proj_body<NT, false, PLAIN>(a, drop, lp_lds);
}
// two independent problems of the same row count in one launch (the start / end hidden layers of the predictor heads)
template <int NT>
__global__ __launch_bounds__(CB_THREADS) void ln_proj_pair_kernel(LnProjArgs a0, LnProjArgs a1, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char lp_lds[];
  if (blockIdx.y == 0) ln_proj_body<NT>(a0, drop, lp_lds);
  else ln_proj_body<NT>(a1, drop, lp_lds);
}

// ------------------------------------------------------------------------------------------------------
// Stores of the T-form kernels: a wave writes 64 B of a row (its 16 columns), the neighbouring wave the other half of the 128-byte
// line.  As streaming (nontemporal) stores the halves reached HBM separately - +22 % write traffic measured (WRITE_SIZE) and no
// faster - so the tensors only the backward reads are stored plainly too: the halves merge in L2
#define DA_ST_NT st4
#define DP_ROWS 48
#define DP_PLANE (DP_ROWS * 256)          // one plane of an operand slot
#define DP_SLOT (2 * DP_PLANE)
#define DP_NB 11                          // small vectors in LDS: 9 biases + ln2 gamma, beta
#define DA_ACT_SCALE 16.0f                // fixed operand scale of the chained tiles of da_post_kernel

// T-form (tilecore.h), like da_mid_bwd_kernel below: wave `wave` owns output columns 16 wave .. 16 wave + 15 of all NT row tiles and
// keeps its weight fragments (T images, straight from L2, requested a step ahead) in registers.  Accumulator rt of lane (j, g) = row
// 16 rt + j, columns 16 wave + 4 g .. + 3.  The row phases RP0 - RP2 keep the 32-lane row layout (thread group grp: rows grp + 16 u).
// TAIL (round 5, as in conv_block_fwd_lnproj_kernel): the next dual attention layer's layer norms + projections (lnproj_body.h) go on with
// the block output of the same rows instead of a launch of their own - the output tiles pass through slot 0 as fp32 rows, the tail's
// operand slots take slots 1 .. 3, its small vectors are staged with this kernel's
template <int NT, bool TAIL>
__device__ __forceinline__ void da_post_body(const DaPostArgs& a, const DropCfg& drop, const LnProjArgs* lp, char* dp_lds) {
  char* P0 = dp_lds;
  char* P1 = P0 + DP_SLOT;
  char* P2 = P1 + DP_SLOT;
  char* P3 = P2 + DP_SLOT;
  float4* bl = reinterpret_cast<float4*>(P3 + DP_SLOT);             // [DP_NB][32] float4
  float* ainv0 = reinterpret_cast<float*>(bl + DP_NB * 32);       // [48] per slot
  float* ainv1 = ainv0 + DP_ROWS;
  float* ainv2 = ainv1 + DP_ROWS;
  float* ainv3 = ainv2 + DP_ROWS;
  float4* TP = reinterpret_cast<float4*>(ainv3 + DP_ROWS);         // TAIL: [LNP_TP_VECS][32] float4
  float4* scratch = reinterpret_cast<float4*>(P0);                 // fp32 rows [48][32] float4 for the LN2 pass (= slot 0)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  const int MT = a.MT, R = a.R;
  const int tile_ = xcd_tile_clip(blockIdx.x, R, a.Nv, MT);
  if (tile_ < 0) return;
  const int r0 = tile_ * MT;                        // (grid rounded up to whole XCD rounds)
  const int RE = min(R, r0 + MT);             // rows [r0, RE) belong to this workgroup (MT need not be a multiple of 16)
  const int j = lane & 15, g = lane >> 4, ecol = 16 * wave + 4 * g;
  const DropRegs dr = drop_load(drop);

  HUAL_STAMP(0);
  TfW wa, wb;
  tf_load_w(wa, a.w[0], wave, lane);
  // ---- everything read from HBM is requested up front: the attention outputs and ln1 rows of the row phases, the residual
  // rows / row mask of the epilogues, the small vectors
  float4 sa[3], xa[3], l1[3], xin[NT];
  float rm[NT];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const size_t off = (size_t)min(r0 + grp + 16 * u, R - 1) * HUAL_D + col;
    sa[u] = ld4(a.s_att + off);
    xa[u] = ld4(a.x_att + off);
    l1[u] = ld4(a.ln1 + off);
  }
  // element offsets of the lane's NT tile rows (clamped to the tensor: loads are unconditional, stores guarded by row < RE) as 32-bit
  // values: uniform base + 32-bit lane offset addressing, one register per row for every tensor of the kernel
  uint32_t eoff[NT];
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) {
    const int row = min(r0 + 16 * rt + j, R - 1);
    eoff[rt] = (uint32_t)row * (uint32_t)HUAL_D + (uint32_t)ecol;
    xin[rt] = ld4(a.x + eoff[rt]);
    rm[rt] = a.rowmask[row];
  }
  {
    // the group's vector.  Indexing the pointer array of the argument struct with the lane-dependent group is a vector load of the
    // POINTER followed by the load through it - two dependent round trips; the two groups of a wave are wave-uniform, so their
    // pointers come out of the argument struct by scalar loads and the lane picks its half's (groups 11 .. 15 read a vector they
    // do not store)
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    auto vec_ptr = [&](int k) -> const float* { return k < 9 ? a.b[k] : (k == 9 ? a.ln2_g : a.ln2_b); };
    const float* pe = vec_ptr(min(2 * wv, 10));
    const float* po = vec_ptr(min(2 * wv + 1, 10));
    const float4 pv = ld4_global(((lane & 32) ? po : pe) + col);
    if (grp < DP_NB) bl[grp * 32 + l32] = pv;
    if (TAIL) {                                              // group k < LNP_TP_VECS: vector k of the tail (absent ones: any valid address)
      auto tail_ptr = [&](int k) -> const float* {
        const float* q = lp->g1;
        if (k == 1) q = lp->b1;
        if (k == 2 && lp->g2) q = lp->g2;
        if (k == 3 && lp->g2) q = lp->b2;
#pragma unroll
        for (int i = 0; i < HUAL_LNPROJ_MAX; ++i)
          if (k == 4 + i && i < lp->nproj && lp->bias[i]) q = lp->bias[i];
        return q;
      };
      const float* te = tail_ptr(min(2 * wv, LNP_TP_VECS - 1));
      const float* to = tail_ptr(min(2 * wv + 1, LNP_TP_VECS - 1));
      const float4 tv = ld4_global(((lane & 32) ? to : te) + col);
      if (grp < LNP_TP_VECS) TP[grp * 32 + l32] = tv;
    }
  }
  tf_load_w(wb, a.w[1], wave, lane);
  // ---- RP0: attention outputs -> slots 0, 1
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int lr = grp + 16 * u;
    if (lr >= MT) continue;
    const bool ok = r0 + lr < RE;
    const float i0 = cb_store_operand(P0, P0 + DP_PLANE, lr, l32, ok ? sa[u] : f4zero());
    const float i1 = cb_store_operand(P1, P1 + DP_PLANE, lr, l32, ok ? xa[u] : f4zero());
    if (l32 == 0) { ainv0[lr] = ok ? i0 : 0.f; ainv1[lr] = ok ? i1 : 0.f; }
  }

  f32x4 accp[NT];
  // acc (+)= accp * inverse operand scale of the rows
  auto fold = [&](float4 (&acc)[NT], const float* ai, bool first) {
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const float ir = ai[16 * rt + j];
      if (first) acc[rt] = make_float4(fmaf(accp[rt][0], ir, 0.f), fmaf(accp[rt][1], ir, 0.f), fmaf(accp[rt][2], ir, 0.f), fmaf(accp[rt][3], ir, 0.f));
      else acc[rt] = make_float4(fmaf(accp[rt][0], ir, acc[rt].x), fmaf(accp[rt][1], ir, acc[rt].y), fmaf(accp[rt][2], ir, acc[rt].z), fmaf(accp[rt][3], ir, acc[rt].w));
    }
  };
  auto addb = [&](float4 v, int k) { const float4 b = bl[k * 32 + (ecol >> 2)]; return make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w); };
  auto save = [&](float* dst, int rt, float4 v) {
    if (r0 + 16 * rt + j < RE) st4(dst + eoff[rt], v);
  };
  auto save_nt = [&](float* dst, int rt, float4 v) {      // tensors only the backward pass reads
    if (r0 + 16 * rt + j < RE) DA_ST_NT(dst + eoff[rt], v);
  };
  // tile -> operand slot.  The tiles of the chain are layer-norm outputs, gated products of them and O(1) projections: a FIXED
  // power-of-two scale (2^4; fp16 pairs then carry 22 bits for |x| >= 2^-7 and an absolute 2^-28 below - |x| must stay under
  // 2^12) replaces the per-row scale, whose maximum needed an exchange of the waves' 16-column slice maxima
  // through LDS and a barrier of its own.  With a fourth operand slot no product overwrites a slot another wave may still
  // read: ONE barrier per chained product (behind the stores), where the per-row form had two.
  auto put = [&](char* P, float* ai, const float4 (&v)[NT]) {
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const int lr = 16 * rt + j;
      uint2 h, l;
      f16_split4(f4scale1(v[rt], DA_ACT_SCALE), h, l);
      const int off = tile256_off(lr, ecol >> 3) + 8 * (g & 1);
      *reinterpret_cast<uint2*>(P + off) = h;
      *reinterpret_cast<uint2*>(P + DP_PLANE + off) = l;
      if (wave == 0 && g == 0) ai[lr] = (r0 + lr < RE) ? 1.0f / (DA_ACT_SCALE * HUAL_F16_WSCALE) : 0.f;
    }
  };
  auto zero_invalid = [&](float4 (&v)[NT]) {      // rows beyond the tensor carry zeros through the chain
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) if (r0 + 16 * rt + j >= RE) v[rt] = f4zero();
  };
  auto drop_rows = [&](uint32_t site, uint8_t* plane, uint32_t (&nib)[NT]) {
    drop_rows_t<NT>(dr, site, a.drop_row0, r0 + j, RE, (uint32_t)(ecol >> 2), plane, nib, lane);
  };

  float4 SV[NT], XV[NT], T1[NT], T2[NT];
  // ---- s_value = s_att . Ws + b ; x_value = x_att . Wx + b  (layers.py:93-94)
  cb_barrier();
  tf_mma_lean<NT, DP_PLANE>(P0, wa, lane, accp);
  fold(SV, ainv0, true);
  tf_load_w(wa, a.w[2], wave, lane);
  tf_mma_lean<NT, DP_PLANE>(P1, wb, lane, accp);
  tf_load_w(wb, a.w[3], wave, lane);                   // (requests go in front of the epilogue's stores, here and below)
  fold(XV, ainv1, true);
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) {
    SV[rt] = addb(SV[rt], 0);
    XV[rt] = addb(XV[rt], 1);
    save_nt(a.sv, rt, SV[rt]);
    save_nt(a.xv, rt, XV[rt]);
  }
  zero_invalid(SV);
  zero_invalid(XV);
  put(P2, ainv2, SV);
  put(P3, ainv3, XV);
  cb_barrier();
  // ---- cross gating (layers.py:96-103): o = sigmoid(s_value . Wsg + b) * x_value + sigmoid(x_value . Wxg + b) * s_value
  tf_mma_lean<NT, DP_PLANE>(P2, wa, lane, accp);
  fold(T1, ainv2, true);
  tf_load_w(wa, a.w[4], wave, lane);
  tf_mma_lean<NT, DP_PLANE>(P3, wb, lane, accp);
  tf_load_w(wb, a.w[5], wave, lane);
  fold(T2, ainv3, true);
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) {
    const float4 sg = sig4(addb(T1[rt], 2)), xg = sig4(addb(T2[rt], 3));
    save_nt(a.sg, rt, sg);
    save_nt(a.xg, rt, xg);
    T1[rt] = cb_add(cb_mul(sg, XV[rt]), cb_mul(xg, SV[rt]));
    save_nt(a.o, rt, T1[rt]);
  }
  zero_invalid(T1);
  put(P0, ainv0, T1);                                        // (slot 0: last read by the s_value product, a barrier ago)
  cb_barrier();
  // ---- guided dense (layers.py:104)
  tf_mma_lean<NT, DP_PLANE>(P0, wa, lane, accp);
  tf_load_w(wa, a.w[6], wave, lane);
  fold(T1, ainv0, true);
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) { T1[rt] = addb(T1[rt], 4); save_nt(a.gd, rt, T1[rt]); }
  zero_invalid(T1);
  put(P2, ainv2, T1);                                        // (slot 2: last read by the s_gate product, a barrier ago)
  // RP1: the layer-normed input (ln1) -> slot 1 (last read by the x_value product)
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int lr = grp + 16 * u;
    if (lr >= MT) continue;
    const bool ok = r0 + lr < RE;
    const float i0 = cb_store_operand_fx(P1, P1 + DP_PLANE, lr, l32, ok ? l1[u] : f4zero());
    if (l32 == 0) ainv1[lr] = ok ? i0 : 0.f;
  }
  cb_barrier();
  // ---- bilinear gate and value (layers.py:48-56, 106-110): scores = ln1 . W11 + g . W12 + b1 ; values = ln1 . W21 + g . W22 + b2
  tf_mma_lean<NT, DP_PLANE>(P1, wb, lane, accp);
  fold(T1, ainv1, true);
  tf_load_w(wb, a.w[7], wave, lane);
  tf_mma_lean<NT, DP_PLANE>(P2, wa, lane, accp);
  fold(T1, ainv2, false);
  tf_load_w(wa, a.w[8], wave, lane);
  tf_mma_lean<NT, DP_PLANE>(P1, wb, lane, accp);
  fold(T2, ainv1, true);
  tf_load_w(wb, a.w[9], wave, lane);
  tf_mma_lean<NT, DP_PLANE>(P2, wa, lane, accp);
  tf_load_w(wa, a.w[10], wave, lane);
  fold(T2, ainv2, false);
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) {
    float4 gate = addb(T1[rt], 5);
    gate = rm[rt] != 0.f ? sig4(gate) : f4zero();             // sigmoid(mask_logits(scores, mask)) (layers.py:110)
    const float4 val = addb(T2[rt], 6);
    save_nt(a.gate, rt, gate);
    save_nt(a.val, rt, val);
    T1[rt] = cb_mul(gate, val);
    save_nt(a.mha, rt, T1[rt]);
  }
  zero_invalid(T1);
  put(P3, ainv3, T1);                                        // (slot 3: last read by the x_gate product, two barriers ago)
  cb_barrier();
  // ---- dense_1 + dropout + residual (modules.py:82-83); the rows also go to LDS as fp32 for the layer norm
  tf_mma_lean<NT, DP_PLANE>(P3, wb, lane, accp);
  fold(SV, ainv3, true);                                      // SV now holds `res`
  {
    uint32_t nbd[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) nbd[rt] = 15u;
    if (dr.enabled) drop_rows((uint32_t)(a.site + 2), a.bits2, nbd);
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const int lr = 16 * rt + j;
      float4 v = addb(SV[rt], 7);
      if (dr.enabled) v = f4_select(nbd[rt], make_float4(v.x * dr.scale, v.y * dr.scale, v.z * dr.scale, v.w * dr.scale));
      v = cb_add(v, xin[rt]);
      SV[rt] = v;
      save(a.res, rt, v);
      scratch[lr * 32 + (ecol >> 2)] = v;                     // (slot 0 was last read two products ago)
    }
  }
  cb_barrier();
  // ---- RP2: layer_norm_2 + dropout (modules.py:85-86) -> slot 2
  {
    const float4 g2 = bl[9 * 32 + l32], b2 = bl[10 * 32 + l32];
    uint32_t nb3[4] = {15u, 15u, 15u, 15u};
    if (dr.enabled) {      // rows grp, grp + 16 share one call per lane pair, row grp + 32 takes one alone
      const int lrA = grp, lrB = grp + 16, lrC = grp + 32;
      drop_nib2_store_r(dr, (uint32_t)(a.site + 3), a.drop_row0, r0 + lrA, r0 + lrB, lrA < MT && r0 + lrA < RE, lrB < MT && r0 + lrB < RE,
                        (uint32_t)l32, a.bits3, nb3[0], nb3[1]);
      drop_nib2_store_r(dr, (uint32_t)(a.site + 3), a.drop_row0, r0 + lrC, r0 + lrC, lrC < MT && r0 + lrC < RE, lrC < MT && r0 + lrC < RE,
                        (uint32_t)l32, a.bits3, nb3[2], nb3[3]);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int lr = grp + 16 * u, row = r0 + lr;
      if (lr >= MT) continue;
      const bool ok = row < RE;
      float mean, rstd;
      const float4 xh = ln_row(scratch[lr * 32 + l32], mean, rstd);
      float4 y = cb_fma(xh, g2, b2);
      if (dr.enabled) y = f4_select(nb3[u], make_float4(y.x * dr.scale, y.y * dr.scale, y.z * dr.scale, y.w * dr.scale));
      if (!ok) y = f4zero();
      if (ok) {
        st4_nt(a.l2 + (size_t)row * HUAL_D + col, y);
        if (l32 == 0) { a.mean2[row] = mean; a.rstd2[row] = rstd; }
      }
      const float i2 = cb_store_operand_fx(P2, P2 + DP_PLANE, lr, l32, y);
      if (l32 == 0) ainv2[lr] = ok ? i2 : 0.f;
    }
  }
  cb_barrier();
  // ---- dense_2 + dropout + residual (modules.py:87-88)
  tf_mma_lean<NT, DP_PLANE>(P2, wa, lane, accp);
  fold(T1, ainv2, true);
  {
    uint32_t nbe[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) nbe[rt] = 15u;
    if (dr.enabled) drop_rows((uint32_t)(a.site + 4), a.bits4, nbe);
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      float4 v = addb(T1[rt], 8);
      if (dr.enabled) v = f4_select(nbe[rt], make_float4(v.x * dr.scale, v.y * dr.scale, v.z * dr.scale, v.w * dr.scale));
      v = cb_add(v, SV[rt]);
      save(a.out, rt, v);
      if (TAIL) scratch[(16 * rt + j) * 32 + (ecol >> 2)] = v;      // (slot 0 has been free since the LN2 pass)
    }
  }
  if (TAIL) {
    cb_barrier();      // the rows are complete, and every wave is through its last product (slot 2)
    ln_proj_body<NT, true, true>(*lp, drop, P1, r0, scratch, r0, 32, DP_ROWS, TP);
  }
}
template <int NT>
__global__ __launch_bounds__(CB_THREADS) void da_post_kernel(DaPostArgs a, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char dp_lds[];
  da_post_body<NT, false>(a, drop, nullptr, dp_lds);
}
template <int NT>
__global__ __launch_bounds__(CB_THREADS) void da_post_lnproj_kernel(DaPostArgs a, DropCfg drop, LnProjArgs lp) {
  extern __shared__ __attribute__((aligned(16))) char dp_lds[];
  da_post_body<NT, true>(a, drop, &lp, dp_lds);
}

// ------------------------------------------------------------------------------------------------------
// dX products -> layer norm(s) backward (LnProjBwdArgs).  Two operand slots are filled alternately from HBM rows (the
// rows of product k+1 are requested before product k's wait and written behind its matrix phase); the two output tiles
// stay in registers until the last product, then go to LDS as fp32 rows for the row phase.
#define LB_ROWS 48
#define LB_U (LB_ROWS / 16)     // rows of a 32-lane group in the row layout
// PRE: with the layer-norm prologue (LnProjBwdArgs::pre_*) - a compile-time switch, the plain launches do not pay for its branches
// SIX: the dual attention's shape - six products, the last two into the second layer norm, no dropout' on an operand, add_dy1, both
// layer norms, a residual addend - as compile-time facts (ln_proj_bwd_six() on the host): per product ~100 of ~575 instructions of a
// wave were selects and compares on those uniform options, and a pair of waves saturates its SIMD's issue in these phases
template <bool PRE, int NT, bool SIX = false>
__global__ __launch_bounds__(CB_THREADS) void ln_proj_bwd_kernel(LnProjBwdArgs a, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char lb_lds[];
  char* S0 = lb_lds;                                   // operand slot 0: hi | lo planes [48][256 B]; later dy_0 as fp32 rows
  char* S1 = S0 + 2 * LB_ROWS * 256;                   // operand slot 1; later dy_1
  char* Ps = S1 + 2 * LB_ROWS * 256;                   // the per-wave parameter sums [8][4][32] float4
  float* ainv0 = reinterpret_cast<float*>(Ps + 8 * 4 * 32 * 16);
  float* ainv1 = ainv0 + LB_ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  const int MT = a.MT, R = a.R;
  const int tile_ = xcd_tile_clip(blockIdx.x, R, a.Nv, MT);
  if (tile_ < 0) return;
  const int r0 = tile_ * MT;                        // (grid rounded up to whole XCD rounds)
  const int RE = min(R, r0 + MT);             // rows [r0, RE) belong to this workgroup (MT need not be a multiple of 16)
  const int j = lane & 15, g = lane >> 4, ecol = 16 * wave + 4 * g;
  const DropRegs dr = drop_load(drop);

  // T-form (tilecore.h): wave `wave` owns gradient columns 16 wave .. 16 wave + 15 of all NT row tiles; its weight fragments (N images)
  // come straight from L2 into registers, a product ahead.  Accumulator rt of lane (j, g) = row 16 rt + j, columns 16 wave + 4 g .. + 3
  TfW w[2];
  tf_load_w(w[0], a.wimg_t[0], wave, lane);
  // operand rows of product 0 and everything the row phase needs (unconditional loads on clamped rows)
  float4 nv[LB_U], xv[LB_U], a1v[LB_U];
  float mu[LB_U], rsd[LB_U];
  constexpr bool pre = PRE;
  const float* add1p = pre ? (a.pre_add ? a.pre_add : a.x) : (a.add1 ? a.add1 : a.x);
  const float* a0p = pre ? a.pre_dy : a.A[0];
  const int lda0 = pre ? HUAL_D : a.lda[0];
  uint32_t nkb[LB_U];                                  // keep-bit bytes of the operand rows in flight
  const int nsteps = SIX ? 6 : a.nsteps;
  auto has_bits = [&](int k) { return !SIX && a.a_bits[k] != nullptr; };
  auto to_second = [&](int k) { return SIX ? k >= 4 : a.dst[k] != 0; };
  const bool has_g2 = SIX || a.g2 != nullptr, has_add_dy1 = SIX || a.add_dy1 != nullptr, has_dy1_bits = !SIX && a.dy1_bits != nullptr;
  const uint8_t* kb0p = has_bits(0) ? a.a_bits[0] : reinterpret_cast<const uint8_t*>(a.x);
  float4 pxv[LB_U];                                    // prologue: rows / statistics of the layer norm in front
  float pmu[LB_U], prs[LB_U];
#pragma unroll
  for (int u = 0; u < LB_U; ++u) {
    const int row = min(r0 + grp + 16 * u, R - 1);
    nv[u] = ld4(a0p + (size_t)row * lda0 + col);
    nkb[u] = SIX ? 0u : kb0p[(size_t)row * 16 + (l32 >> 1)];
    xv[u] = ld4(a.x + (size_t)row * HUAL_D + col);
    a1v[u] = ld4(add1p + (size_t)row * HUAL_D + col);
    mu[u] = a.mean[row];
    rsd[u] = a.rstd[row];
    if (pre) { pxv[u] = ld4(a.pre_x + (size_t)row * HUAL_D + col); pmu[u] = a.pre_mean[row]; prs[u] = a.pre_rstd[row]; }
  }
  const float4 g1 = ld4(a.g1 + col);
  float4 sg2 = f4zero(), sb2 = f4zero();               // second layer norm's parameter sums - or the prologue layer norm's
  if (pre) {      // dxp = LNbwd(pre_x; pre_dy, pre_g) + pre_add -> operand of product 0 and the residual addend of the row phase
    const float4 pg = ld4(a.pre_g + col);
#pragma unroll
    for (int u = 0; u < LB_U; ++u) {
      const int lr = grp + 16 * u;
      const bool ok = lr < MT && r0 + lr < RE;
      const float4 v = pxv[u], dy = nv[u];
      const float mean = pmu[u], rstd = prs[u];
      const float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
      if (ok) { sb2 = cb_add(sb2, dy); sg2 = cb_fma(dy, xh, sg2); }
      const float4 gv = cb_mul(dy, pg);
      const float m1 = fast_sum32(cb_hsum(gv)) * (1.0f / HUAL_D);
      const float m2 = fast_sum32(cb_hsum(cb_mul(gv, xh))) * (1.0f / HUAL_D);
      float4 dx = make_float4(rstd * (gv.x - m1 - xh.x * m2), rstd * (gv.y - m1 - xh.y * m2),
                              rstd * (gv.z - m1 - xh.z * m2), rstd * (gv.w - m1 - xh.w * m2));
      if (a.pre_add) dx = make_float4(__fadd_rn(dx.x, a1v[u].x), __fadd_rn(dx.y, a1v[u].y), __fadd_rn(dx.z, a1v[u].z), __fadd_rn(dx.w, a1v[u].w));
      nv[u] = dx;
      a1v[u] = dx;
    }
  }
  const float4 g2 = has_g2 ? ld4(a.g2 + col) : f4zero();
  float4 addt[NT];                                     // add_dy1 in the accumulator layout
  const float* addp = has_add_dy1 ? a.add_dy1 : a.x;
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) addt[rt] = SIX ? f4zero() : ld4(addp + (uint32_t)min(r0 + 16 * rt + j, R - 1) * (uint32_t)HUAL_D + (uint32_t)ecol);      // (SIX: requested in front of the last product - 12 registers less across the loop)
  // rows -> operand planes of slot `k & 1` (with the operand's dropout', saved for the weight-gradient job)
  auto fill = [&](int k) {
    char* S = (k & 1) ? S1 : S0;
    float* ai = (k & 1) ? ainv1 : ainv0;
#pragma unroll
    for (int u = 0; u < LB_U; ++u) {
      const int lr = grp + 16 * u, row = r0 + lr;
      if (lr >= MT) continue;
      const bool ok = row < RE;
      float4 v = ok ? nv[u] : f4zero();
      if (has_bits(k)) {      // dropout' with the keep bits the forward left (requested with the rows)
        if (dr.enabled) v = f4_select((nkb[u] >> (4 * (l32 & 1))) & 15u, make_float4(v.x * dr.scale, v.y * dr.scale, v.z * dr.scale, v.w * dr.scale));
        if (ok && a.a_save[k]) st4_nt(a.a_save[k] + (size_t)row * HUAL_D + col, v);
      }
      const float inv = cb_store_operand(S, S + LB_ROWS * 256, lr, l32, v);
      if (l32 == 0) ai[lr] = ok ? inv : 0.f;
    }
  };
  HUAL_STAMP_K(4, 0);
  fill(0);
  HUAL_STAMP_K(4, 1);
  float4 acc0[NT], acc1[NT];
  bool first0 = true, first1 = true;
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) { acc0[rt] = f4zero(); acc1[rt] = f4zero(); }
  // one barrier per product: slot k & 1 complete (filled behind product k - 1, whose reads of the other slot every wave has finished
  // before it arrives here); the next product's rows and weight fragments are requested in front of it
#pragma unroll
  for (int k = 0; k < HUAL_LNBWD_MAX; ++k) {
    if (k >= nsteps) break;                            // uniform
    const bool more = k + 1 < nsteps;
    if (SIX && k == 5) {
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) addt[rt] = ld4(addp + (uint32_t)min(r0 + 16 * rt + j, R - 1) * (uint32_t)HUAL_D + (uint32_t)ecol);
    }
    if (more) {
      const uint8_t* nbp = has_bits(k + 1) ? a.a_bits[k + 1] : reinterpret_cast<const uint8_t*>(a.x);      // (no bits: a byte that is ignored)
      const float* nap = a.A[k + 1];
      const int nld = a.lda[k + 1];
#pragma unroll
      for (int u = 0; u < LB_U; ++u) {
        const size_t row = (size_t)min(r0 + grp + 16 * u, R - 1);
        nv[u] = ld4(nap + row * nld + col);
        nkb[u] = SIX ? 0u : nbp[row * 16 + (l32 >> 1)];
      }
      tf_load_w(w[(k + 1) & 1], a.wimg_t[k + 1], wave, lane);
    }
    cb_barrier();
    HUAL_STAMP_K(4, 2 + 3 * k);
    const char* S = (k & 1) ? S1 : S0;
    const float* ai = (k & 1) ? ainv1 : ainv0;
    f32x4 accp[NT];
    tf_mma_lean<NT, LB_ROWS * 256>(S, w[k & 1], lane, accp);
    HUAL_STAMP_K(4, 3 + 3 * k);
    {
      const bool to1 = to_second(k);
      const bool first = to1 ? first1 : first0;
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) {
        const float ir = ai[16 * rt + j];
        const float4 o = to1 ? acc1[rt] : acc0[rt];
        const float4 n = first ? make_float4(fmaf(accp[rt][0], ir, 0.f), fmaf(accp[rt][1], ir, 0.f), fmaf(accp[rt][2], ir, 0.f), fmaf(accp[rt][3], ir, 0.f))
                               : make_float4(fmaf(accp[rt][0], ir, o.x), fmaf(accp[rt][1], ir, o.y), fmaf(accp[rt][2], ir, o.z), fmaf(accp[rt][3], ir, o.w));
        if (to1) acc1[rt] = n; else acc0[rt] = n;
      }
    }
    if (to_second(k)) first1 = false; else first0 = false;
    if (more) fill(k + 1);                             // slot (k+1)&1 was last read by product k-1
    HUAL_STAMP_K(4, 4 + 3 * k);
  }
  cb_barrier();                                        // every wave is through the last product: both slots are free
  HUAL_STAMP_K(4, 20);
  // ---- the two output-gradient tiles -> LDS as fp32 rows
  float4* D0 = reinterpret_cast<float4*>(S0);
  float4* D1 = reinterpret_cast<float4*>(S1);
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) {
    const int lr = 16 * rt + j;
    float4 v = acc0[rt];
    if (has_add_dy1) v = cb_add(v, addt[rt]);
    D0[lr * 32 + (ecol >> 2)] = v;
    if (has_g2) D1[lr * 32 + (ecol >> 2)] = acc1[rt];
  }
  // keep bits of the row phase (dropout' of dy_0 / of dx): requested together in front of the barrier - a load behind a branch
  // inside the row loop is a round trip of its own per row (absent planes: a byte of x that is not used)
  uint32_t kb1[LB_U], kbz[LB_U];
  {
    const uint8_t* b1p = has_dy1_bits ? a.dy1_bits : reinterpret_cast<const uint8_t*>(a.x);
    const uint8_t* bzp = a.dz_bits ? a.dz_bits : reinterpret_cast<const uint8_t*>(a.x);
#pragma unroll
    for (int u = 0; u < LB_U; ++u) {
      const size_t row = (size_t)min(r0 + grp + 16 * u, R - 1);
      kb1[u] = SIX ? 0u : b1p[row * 16 + (l32 >> 1)];
      kbz[u] = bzp[row * 16 + (l32 >> 1)];
    }
  }
  cb_barrier();
  HUAL_STAMP_K(4, 21);
  // ---- row phase: layer norm(s) backward.  dy = dy*g ; dx = rstd * (gv - mean(gv) - xhat * mean(gv * xhat))   (ln_bwd_kernel)
  float4 sg1 = f4zero(), sb1 = f4zero();
#pragma unroll
  for (int u = 0; u < LB_U; ++u) {
    const int lr = grp + 16 * u, row = r0 + lr;
    if (lr >= MT || row >= R) continue;
    const size_t off = (size_t)row * HUAL_D + col;
    const float4 v = xv[u];
    const float mean = mu[u], rstd = rsd[u];
    const float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
    float4 dy = D0[lr * 32 + l32];
    if (has_dy1_bits && dr.enabled) dy = f4_select((kb1[u] >> (4 * (l32 & 1))) & 15u, make_float4(dy.x * dr.scale, dy.y * dr.scale, dy.z * dr.scale, dy.w * dr.scale));
    sb1 = cb_add(sb1, dy);
    sg1 = cb_fma(dy, xh, sg1);
    float4 gv = cb_mul(dy, g1);
    if (has_g2) {
      const float4 dy2 = D1[lr * 32 + l32];
      sb2 = cb_add(sb2, dy2);
      sg2 = cb_fma(dy2, xh, sg2);
      gv = cb_fma(dy2, g2, gv);
    }
    const float m1 = fast_sum32(cb_hsum(gv)) * (1.0f / HUAL_D);
    const float m2 = fast_sum32(cb_hsum(cb_mul(gv, xh))) * (1.0f / HUAL_D);
    float4 dx = make_float4(rstd * (gv.x - m1 - xh.x * m2), rstd * (gv.y - m1 - xh.y * m2),
                            rstd * (gv.z - m1 - xh.z * m2), rstd * (gv.w - m1 - xh.w * m2));
    if (SIX || a.add1 || pre) dx = make_float4(__fadd_rn(dx.x, a1v[u].x), __fadd_rn(dx.y, a1v[u].y), __fadd_rn(dx.z, a1v[u].z), __fadd_rn(dx.w, a1v[u].w));
    st4(a.dx + off, dx);
    if (a.dz) {
      if (a.dz_bits && dr.enabled) dx = f4_select((kbz[u] >> (4 * (l32 & 1))) & 15u, make_float4(dx.x * dr.scale, dx.y * dr.scale, dx.z * dr.scale, dx.w * dr.scale));
      st4(a.dz + off, dx);
    }
  }
  HUAL_STAMP_K(4, 22);
  // ---- parameter sums of the workgroup: halves of a wave in registers, the 8 waves through LDS (the weight buffer is free)
  {
    auto xor32_sum = [&](float v) {
      const unsigned x = __builtin_bit_cast(unsigned, v);
      const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
      return v + __builtin_bit_cast(float, (lane & 32) ? r[0] : r[1]);
    };
    auto xor32_sum4 = [&](float4 v) { return make_float4(xor32_sum(v.x), xor32_sum(v.y), xor32_sum(v.z), xor32_sum(v.w)); };
    sg1 = xor32_sum4(sg1); sb1 = xor32_sum4(sb1); sg2 = xor32_sum4(sg2); sb2 = xor32_sum4(sb2);
    float4* pb = reinterpret_cast<float4*>(Ps);        // [8 waves][4 vectors][32] float4
    if (lane < 32) {
      float4* dst = pb + wave * 4 * 32 + l32;
      dst[0] = sg1; dst[32] = sb1; dst[64] = sg2; dst[96] = sb2;
    }
    cb_barrier();
    const float* pf = reinterpret_cast<const float*>(pb);
    const int e = threadIdx.x;                         // 4 x 128 sums, one per thread
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += pf[w * 4 * HUAL_D + e];
    a.part[(size_t)tile_ * 4 * HUAL_D + e] = s;
  }
  HUAL_STAMP_K(4, 23);
}

// ------------------------------------------------------------------------------------------------------
// Backward of the gated middle of the dual attention (DaMidBwdArgs): ten weight steps, three operand slots.
// T-form (tilecore.h): wave `wave` owns the 16 output columns 16 wave .. 16 wave + 15 of ALL NT row tiles of the workgroup - every
// wave works whatever the row count (the 16 x 64 tiles of the LDS-image form leave two of eight waves idle at three row tiles, on
// two of the four SIMDs) - and reads its weight fragments straight from the L2-resident N images into registers, a step ahead:
// no weight buffer in LDS, no DMA wait, no barrier for weights.  Accumulator rt of lane (j, g) = row 16 rt + j, columns
// 16 wave + 4 g .. + 3.
template <int NT>
__global__ __launch_bounds__(CB_THREADS) void da_mid_bwd_kernel(DaMidBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char dm_lds[];
  char* P0 = dm_lds;
  char* P1 = P0 + DP_SLOT;
  char* P2 = P1 + DP_SLOT;
  float* ainv0 = reinterpret_cast<float*>(P2 + DP_SLOT);
  float* ainv1 = ainv0 + DP_ROWS;
  float* ainv2 = ainv1 + DP_ROWS;
  float* smaxA = ainv2 + DP_ROWS;                      // [48][8] maxima of the 16-column slices of the (first) tile being written
  float* smaxB = smaxA + 8 * DP_ROWS;                  // ... and of the second
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  const int MT = a.MT, R = a.R;
  const int tile_ = xcd_tile_clip(blockIdx.x, R, a.Nv, MT);
  if (tile_ < 0) return;
  const int r0 = tile_ * MT;                        // (grid rounded up to whole XCD rounds)
  const int RE = min(R, r0 + MT);             // rows [r0, RE) belong to this workgroup (MT need not be a multiple of 16)
  const int j = lane & 15, g = lane >> 4, ecol = 16 * wave + 4 * g;

  TfW wa, wb;
  tf_load_w(wa, a.w[0], wave, lane);
  {
    float4 zv[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) zv[u] = ld4(a.dz1 + (size_t)min(r0 + grp + 16 * u, R - 1) * HUAL_D + col);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int lr = grp + 16 * u;
      if (lr >= MT) continue;
      const bool ok = r0 + lr < RE;
      const float inv = cb_store_operand(P1, P1 + DP_PLANE, lr, l32, ok ? zv[u] : f4zero());
      if (l32 == 0) ainv1[lr] = ok ? inv : 0.f;
    }
  }
  f32x4 accp[NT];
  auto fold = [&](float4 (&acc)[NT], const float* ai, bool first) {
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const float ir = ai[16 * rt + j];
      if (first) acc[rt] = make_float4(fmaf(accp[rt][0], ir, 0.f), fmaf(accp[rt][1], ir, 0.f), fmaf(accp[rt][2], ir, 0.f), fmaf(accp[rt][3], ir, 0.f));
      else acc[rt] = make_float4(fmaf(accp[rt][0], ir, acc[rt].x), fmaf(accp[rt][1], ir, acc[rt].y), fmaf(accp[rt][2], ir, acc[rt].z), fmaf(accp[rt][3], ir, acc[rt].w));
    }
  };
  // element offsets of the lane's NT tile rows (clamped to the tensor: loads are unconditional, stores are guarded by row < RE) as
  // 32-bit values: uniform base + 32-bit lane offset addressing, one register per row for every tensor of the kernel
  uint32_t eoff[NT];
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) eoff[rt] = (uint32_t)min(r0 + 16 * rt + j, R - 1) * (uint32_t)HUAL_D + (uint32_t)ecol;
  auto tile_ld = [&](const float* src, float4 (&t)[NT]) {
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) t[rt] = ld4(src + eoff[rt]);
  };
  auto save = [&](float* dst, int rt, float4 v) {
    if (r0 + 16 * rt + j < RE) st4(dst + eoff[rt], v);
  };
  auto save_nt = [&](float* dst, int rt, float4 v) {      // operands of the weight-gradient launch only
    if (r0 + 16 * rt + j < RE) DA_ST_NT(dst + eoff[rt], v);
  };
  auto put_max = [&](float* sm, const float4 (&v)[NT]) {
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const float m = slice16_max(f4absmax(v[rt]), lane);
      if (g == 0) sm[(16 * rt + j) * 8 + wave] = m;
    }
  };
  auto put_planes = [&](char* P, float* ai, const float* sm, const float4 (&v)[NT]) {
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const int lr = 16 * rt + j;
      const float4 ma = *reinterpret_cast<const float4*>(sm + lr * 8), mb = *reinterpret_cast<const float4*>(sm + lr * 8 + 4);
      float inv;
      const float sc = f16_row_scale(fmaxf(fmaxf(fmaxf(ma.x, ma.y), fmaxf(ma.z, ma.w)), fmaxf(fmaxf(mb.x, mb.y), fmaxf(mb.z, mb.w))), inv);
      uint2 h, l;
      f16_split4(f4scale1(v[rt], sc), h, l);
      const int off = tile256_off(lr, ecol >> 3) + 8 * (g & 1);
      *reinterpret_cast<uint2*>(P + off) = h;
      *reinterpret_cast<uint2*>(P + DP_PLANE + off) = l;
      if (wave == 0 && g == 0) ai[lr] = (r0 + lr < RE) ? inv : 0.f;
    }
  };
  auto zero_invalid = [&](float4 (&v)[NT]) {
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) if (r0 + 16 * rt + j >= RE) v[rt] = f4zero();
  };

  float4 T1[NT], T2[NT], U1[NT], U2[NT], U3[NT], U4[NT];
  // ---- d mha = dZ1 . Wd1^T ; bilinear backward (layers.py:110): d scores = d mha * val * gate * (1 - gate), d values = d mha * gate
  tf_load_w(wb, a.w[1], wave, lane);
  tile_ld(a.gate, U1);
  tile_ld(a.val, U2);
  cb_barrier();
  tf_mma_lean<NT, DP_PLANE>(P1, wa, lane, accp);
  tf_load_w(wa, a.w[2], wave, lane);                   // (requests go in front of the epilogue's stores, here and below)
  fold(T1, ainv1, true);
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) {
    const float4 d0 = T1[rt], gt = U1[rt], vl = U2[rt];
    T1[rt] = make_float4(d0.x * vl.x * gt.x * (1.f - gt.x), d0.y * vl.y * gt.y * (1.f - gt.y), d0.z * vl.z * gt.z * (1.f - gt.z),
                         d0.w * vl.w * gt.w * (1.f - gt.w));
    T2[rt] = cb_mul(d0, gt);
    save_nt(a.d_sc, rt, T1[rt]);
    save_nt(a.d_val, rt, T2[rt]);
  }
  zero_invalid(T1);
  zero_invalid(T2);
  put_max(smaxA, T1);
  put_max(smaxB, T2);
  cb_barrier();
  put_planes(P0, ainv0, smaxA, T1);
  put_planes(P2, ainv2, smaxB, T2);
  cb_barrier();
  // ---- gradient of ln1 through the two bilinear layers: d scores . W11^T + d values . W21^T
  tf_mma_lean<NT, DP_PLANE>(P0, wb, lane, accp);
  fold(T1, ainv0, true);
  tf_load_w(wb, a.w[3], wave, lane);
  tf_mma_lean<NT, DP_PLANE>(P2, wa, lane, accp);
  fold(T1, ainv2, false);                              // d ln1a: stored below, BEHIND the loads of the next phase (the vector-memory
                                                       // pipe is in order: requests issued behind stores wait for the stores' turn)
  // ---- gradient of the guided features: d scores . W12^T + d values . W22^T
  tf_load_w(wa, a.w[4], wave, lane);
  tf_mma_lean<NT, DP_PLANE>(P0, wb, lane, accp);
  fold(T2, ainv0, true);
  tf_load_w(wb, a.w[5], wave, lane);
  tile_ld(a.sg, U1);
  tile_ld(a.xg, U2);
  tile_ld(a.sv, U3);
  tile_ld(a.xv, U4);
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) save(a.d_ln1a, rt, T1[rt]);
  tf_mma_lean<NT, DP_PLANE>(P2, wa, lane, accp);
  fold(T2, ainv2, false);
  tf_load_w(wa, a.w[6], wave, lane);
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) save_nt(a.d_g, rt, T2[rt]);
  zero_invalid(T2);
  put_max(smaxA, T2);
  cb_barrier();
  put_planes(P1, ainv1, smaxA, T2);
  cb_barrier();
  // ---- d o = d g . Wg^T ; cross gating backward (layers.py:96-103): o = sg * x + xg * s
  tf_mma_lean<NT, DP_PLANE>(P1, wb, lane, accp);
  tf_load_w(wb, a.w[7], wave, lane);
  fold(T1, ainv1, true);
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) {
    const float4 d0 = T1[rt], sg = U1[rt], xg = U2[rt], sv = U3[rt], xv = U4[rt];
    T1[rt] = make_float4(d0.x * xv.x * sg.x * (1.f - sg.x), d0.y * xv.y * sg.y * (1.f - sg.y), d0.z * xv.z * sg.z * (1.f - sg.z),
                         d0.w * xv.w * sg.w * (1.f - sg.w));                        // dZ of s_gate
    T2[rt] = make_float4(d0.x * sv.x * xg.x * (1.f - xg.x), d0.y * sv.y * xg.y * (1.f - xg.y), d0.z * sv.z * xg.z * (1.f - xg.z),
                         d0.w * sv.w * xg.w * (1.f - xg.w));                        // dZ of x_gate
    U3[rt] = cb_mul(d0, xg);                                                         // direct part of d s_value
    U4[rt] = cb_mul(d0, sg);                                                         // direct part of d x_value
    save_nt(a.dz_sg, rt, T1[rt]);
    save_nt(a.dz_xg, rt, T2[rt]);
  }
  zero_invalid(T1);
  zero_invalid(T2);
  put_max(smaxA, T1);
  put_max(smaxB, T2);
  cb_barrier();
  put_planes(P0, ainv0, smaxA, T1);
  put_planes(P2, ainv2, smaxB, T2);
  cb_barrier();
  // ---- d s_value = dZ_sg . Wsg^T + d o * xg ; d x_value = dZ_xg . Wxg^T + d o * sg
  tf_mma_lean<NT, DP_PLANE>(P0, wa, lane, accp);
  fold(T1, ainv0, true);
  tf_load_w(wa, a.w[8], wave, lane);
  tf_mma_lean<NT, DP_PLANE>(P2, wb, lane, accp);
  tf_load_w(wb, a.w[9], wave, lane);
  fold(T2, ainv2, true);
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) {
    T1[rt] = cb_add(T1[rt], U3[rt]);
    T2[rt] = cb_add(T2[rt], U4[rt]);
    save_nt(a.d_sv, rt, T1[rt]);
    save_nt(a.d_xv, rt, T2[rt]);
  }
  zero_invalid(T1);
  zero_invalid(T2);
  put_max(smaxA, T1);
  put_max(smaxB, T2);
  cb_barrier();
  put_planes(P1, ainv1, smaxA, T1);
  put_planes(P0, ainv0, smaxB, T2);
  cb_barrier();
  // ---- gradients of the two attention outputs
  tf_mma_lean<NT, DP_PLANE>(P1, wa, lane, accp);
  fold(T1, ainv1, true);
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) save(a.d_satt, rt, T1[rt]);
  tf_mma_lean<NT, DP_PLANE>(P0, wb, lane, accp);
  fold(T2, ainv0, true);
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) save(a.d_xatt, rt, T2[rt]);
}

#if defined(HUAL_STAMPS) && (HUAL_STAMPS == 1 || (HUAL_STAMPS >= 4 && HUAL_STAMPS <= 6))
extern "C" int hual_debug_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hual_stamps), sizeof(unsigned long long) * (size_t)n);
}
extern "C" int hual_debug_stamps_reset() {
  static unsigned long long zeros[512 * HUAL_STAMP_SLOTS];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_hual_stamps), zeros, sizeof(zeros));
}
#endif

namespace hual {

// rows per workgroup: one workgroup per CU when the rows allow it (every workgroup streams all the weight images of its
// launch, so fewer, taller workgroups cost nothing extra).  Any count from 16 to max_rows: the matrix phase works on
// whole 16-row tiles, rows of the last tile beyond the workgroup's own are computed on stale operands and discarded.
static int tile_rows(int R, int Nv, int max_rows) { return xcd_clip_rows(R, Nv, 16, max_rows); }
int ln_proj_rows(int R, int Nv) { return tile_rows(R, Nv, LP_ROWS); }
// the pair launch: BOTH problems in one round of workgroups (a workgroup fills a CU: 2 x 256 workgroups of 32 rows ran as two rounds,
// each paying the cold start - weights, parameters, layer-norm phase)
int ln_proj_pair_rows(int R) {
  int mt = 16;
  while (mt < LP_ROWS && 2 * xcd_clip_grid(R, 0, mt) > 256) ++mt;
  return mt;
}
int da_post_rows(int R, int Nv) { return tile_rows(R, Nv, DP_ROWS); }

static int check_ln_proj(const LnProjArgs& a) {
  HUAL_REQUIRE(a.x && a.g1 && a.b1 && a.y1 && a.mean && a.rstd && a.R > 0, "ln_proj: null / empty");
  HUAL_REQUIRE(a.MT >= 1 && a.MT <= LP_ROWS, "ln_proj: MT must be 1..64");
  HUAL_REQUIRE(a.nproj >= 1 && a.nproj <= HUAL_LNPROJ_MAX, "ln_proj: projection count");
  HUAL_REQUIRE(!a.g2 || (a.b2 && a.y2), "ln_proj: second layer norm incomplete");
  HUAL_REQUIRE(!(a.g2 && a.x2), "ln_proj: second layer norm and raw second operand are exclusive");
  HUAL_REQUIRE(!a.xa || a.x_out, "ln_proj: xa needs x_out");
  for (int p = 0; p < a.nproj; ++p) {
    HUAL_REQUIRE(a.wimg[p] && (a.accum[p] || (a.out[p] && (a.ldo[p] % 4) == 0)), "ln_proj: projection operand");
    HUAL_REQUIRE(a.src[p] == 0 || a.g2 || a.x2, "ln_proj: projection reads an absent second operand");
    HUAL_REQUIRE(!a.add_x[p] || !(a.g2 || a.x2), "ln_proj: the residual rows share LDS with the second operand");
    HUAL_REQUIRE(!a.accum[p] || p + 1 < a.nproj, "ln_proj: the last projection cannot be marked accum");
  }
  return 0;
}
int check_ln_proj_args(const LnProjArgs& a) { return check_ln_proj(a); }
bool ln_proj_plain(const LnProjArgs& a) {
  if (a.xa || a.x2 || a.pre_site >= 0) return false;
  for (int p = 0; p < a.nproj; ++p)
    if (a.accum[p] || a.act[p] || a.out_site[p] >= 0 || a.add_x[p]) return false;
  return true;
}
static const size_t kLnProjLds = LN_PROJ_LDS;
static void ln_proj_work(const LnProjArgs& a, double& flops, double& bytes) {
  const double rows = (double)a.R;
  flops += 2.0 * rows * HUAL_D * HUAL_D * a.nproj;
  bytes += 4.0 * (rows * HUAL_D * (2.0 + (a.g2 ? 1.0 : 0.0) + a.nproj) + (double)a.nproj * HUAL_D * HUAL_D);
}

int launch_ln_proj(const LnProjArgs& a, const DropCfg& drop, hipStream_t s) {
  int rc = check_ln_proj(a);
  if (rc) return rc;
  double flops = 0.0, bytes = 0.0;
  ln_proj_work(a, flops, bytes);
  const dim3 grid(xcd_clip_grid(a.R, a.Nv, a.MT));
  const bool plain = ln_proj_plain(a);
#define LN_PROJ_NT(NT) { if (plain) { HUAL_DYN_LDS((ln_proj_kernel<NT, true>), 160 * 1024); HUAL_LAUNCH(flops, bytes, (ln_proj_kernel<NT, true>), grid, dim3(CB_THREADS), kLnProjLds, s, a, drop); } \
                         else { HUAL_DYN_LDS((ln_proj_kernel<NT, false>), 160 * 1024); HUAL_LAUNCH(flops, bytes, (ln_proj_kernel<NT, false>), grid, dim3(CB_THREADS), kLnProjLds, s, a, drop); } break; }
  switch (cdiv(a.MT, 16)) {      // row tiles per workgroup
    case 1: LN_PROJ_NT(1)
    case 2: LN_PROJ_NT(2)
    case 3: LN_PROJ_NT(3)
    default: LN_PROJ_NT(4)
  }
#undef LN_PROJ_NT
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_ln_proj_pair(const LnProjArgs& a0, const LnProjArgs& a1, const DropCfg& drop, hipStream_t s) {
  int rc = check_ln_proj(a0);
  if (rc) return rc;
  if ((rc = check_ln_proj(a1))) return rc;
  HUAL_REQUIRE(a0.R == a1.R && a0.MT == a1.MT, "ln_proj_pair: the two problems must have the same rows / tile");
  double flops = 0.0, bytes = 0.0;
  ln_proj_work(a0, flops, bytes);
  ln_proj_work(a1, flops, bytes);
  const dim3 grid(xcd_clip_grid(a0.R, a0.Nv, a0.MT), 2);
#define LN_PROJ_NT(NT) { HUAL_DYN_LDS(ln_proj_pair_kernel<NT>, 160 * 1024); HUAL_LAUNCH(flops, bytes, ln_proj_pair_kernel<NT>, grid, dim3(CB_THREADS), kLnProjLds, s, a0, a1, drop); break; }
  switch (cdiv(a0.MT, 16)) {
    case 1: LN_PROJ_NT(1)
    case 2: LN_PROJ_NT(2)
    case 3: LN_PROJ_NT(3)
    default: LN_PROJ_NT(4)
  }
#undef LN_PROJ_NT
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_da_post(const DaPostArgs& a, const DropCfg& drop, hipStream_t s, const LnProjArgs* tail) {
  HUAL_REQUIRE(a.s_att && a.x_att && a.ln1 && a.x && a.rowmask && a.ln2_g && a.ln2_b && a.R > 0, "da_post: null / empty");
  HUAL_REQUIRE(a.MT >= 1 && a.MT <= DP_ROWS, "da_post: MT must be 1..48");
  for (int k = 0; k < 11; ++k) HUAL_REQUIRE(a.w[k] != nullptr, "da_post: null weight image");
  for (int k = 0; k < 9; ++k) HUAL_REQUIRE(a.b[k] != nullptr, "da_post: null bias");
  HUAL_REQUIRE(a.sv && a.xv && a.sg && a.xg && a.o && a.gd && a.gate && a.val && a.mha && a.res && a.l2 && a.out && a.mean2 && a.rstd2,
               "da_post: null output");
  const size_t lds = (size_t)4 * DP_SLOT + DP_NB * 512 + 4 * DP_ROWS * sizeof(float) + (tail ? 10 * 512 : 0);
  const double rows = (double)a.R;
  double flops = 11.0 * 2.0 * rows * HUAL_D * HUAL_D, bytes = 4.0 * (rows * HUAL_D * 16.0 + 11.0 * HUAL_D * HUAL_D);
  const dim3 grid(xcd_clip_grid(a.R, a.Nv, a.MT));
  if (tail) {
    // the tail works on this launch's tiles (lnproj_body.h): rows of the block output from LDS, operand slots over slots 1 .. 3
    LnProjArgs lp = *tail;
    lp.MT = a.MT;
    int rc = check_ln_proj(lp);
    if (rc) return rc;
    HUAL_REQUIRE(lp.x == a.out && lp.R == a.R && ln_proj_plain(lp), "da_post: the tail must read the block output and be of the plain shape");
    static_assert(DP_SLOT + LN_PROJ_LDS <= 4 * DP_SLOT, "the tail's operand slots must fit slots 1 .. 3");
    flops += 2.0 * rows * HUAL_D * HUAL_D * lp.nproj;
    bytes += 4.0 * rows * HUAL_D * (lp.nproj + (lp.g2 ? 1.0 : 0.0));
    switch (cdiv(a.MT, 16)) {
      case 1: { HUAL_DYN_LDS(da_post_lnproj_kernel<1>, 160 * 1024); HUAL_LAUNCH(flops, bytes, da_post_lnproj_kernel<1>, grid, dim3(CB_THREADS), lds, s, a, drop, lp); break; }
      case 2: { HUAL_DYN_LDS(da_post_lnproj_kernel<2>, 160 * 1024); HUAL_LAUNCH(flops, bytes, da_post_lnproj_kernel<2>, grid, dim3(CB_THREADS), lds, s, a, drop, lp); break; }
      default: { HUAL_DYN_LDS(da_post_lnproj_kernel<3>, 160 * 1024); HUAL_LAUNCH(flops, bytes, da_post_lnproj_kernel<3>, grid, dim3(CB_THREADS), lds, s, a, drop, lp); break; }
    }
    HUAL_CHECK_HIP(hipGetLastError());
    return 0;
  }
  switch (cdiv(a.MT, 16)) {      // row tiles per workgroup
    case 1: { HUAL_DYN_LDS(da_post_kernel<1>, 160 * 1024); HUAL_LAUNCH(flops, bytes, da_post_kernel<1>, grid, dim3(CB_THREADS), lds, s, a, drop); break; }
    case 2: { HUAL_DYN_LDS(da_post_kernel<2>, 160 * 1024); HUAL_LAUNCH(flops, bytes, da_post_kernel<2>, grid, dim3(CB_THREADS), lds, s, a, drop); break; }
    default: { HUAL_DYN_LDS(da_post_kernel<3>, 160 * 1024); HUAL_LAUNCH(flops, bytes, da_post_kernel<3>, grid, dim3(CB_THREADS), lds, s, a, drop); break; }
  }
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int ln_proj_bwd_rows(int R, int Nv) { return tile_rows(R, Nv, LB_ROWS); }
int ln_proj_bwd_blocks(int R, int Nv) { return cdiv(R, ln_proj_bwd_rows(R, Nv)); }

// the dual attention's shape (ln_proj_bwd_kernel SIX)
static bool ln_proj_bwd_six(const LnProjBwdArgs& a) {
  if (a.pre_x || a.nsteps != 6 || !a.g2 || !a.add_dy1 || a.dy1_bits || !a.add1) return false;
  for (int k = 0; k < 6; ++k)
    if (a.a_bits[k] || a.a_save[k] || (a.dst[k] != 0) != (k >= 4)) return false;
  return true;
}

int launch_ln_proj_bwd(const LnProjBwdArgs& a, const DropCfg& drop, hipStream_t s) {
  HUAL_REQUIRE(a.nsteps >= 1 && a.nsteps <= HUAL_LNBWD_MAX && a.R > 0, "ln_proj_bwd: step count / rows");
  HUAL_REQUIRE(a.MT >= 1 && a.MT <= LB_ROWS, "ln_proj_bwd: MT must be 1..48");
  HUAL_REQUIRE(a.x && a.mean && a.rstd && a.g1 && a.dx && a.part, "ln_proj_bwd: null tensor");
  bool any1 = false;
  for (int k = 0; k < a.nsteps; ++k) {
    HUAL_REQUIRE((a.A[k] || (k == 0 && a.pre_x)) && a.wimg_t[k] && (a.lda[k] % 4) == 0, "ln_proj_bwd: product operand");
    HUAL_REQUIRE(a.dst[k] == 0 || a.g2, "ln_proj_bwd: product for the absent second layer norm");
    any1 = any1 || a.dst[k] != 0;
  }
  HUAL_REQUIRE(!a.g2 || any1, "ln_proj_bwd: second layer norm without a product");
  HUAL_REQUIRE(!a.pre_x || (!a.g2 && !a.add1 && a.pre_mean && a.pre_rstd && a.pre_g && a.pre_dy), "ln_proj_bwd: layer-norm prologue");
  const size_t lds = (size_t)4 * LB_ROWS * 256 + 8 * 4 * 32 * 16 + 2 * LB_ROWS * sizeof(float);
  const double rows = (double)a.R;
  const double flops = 2.0 * rows * HUAL_D * HUAL_D * a.nsteps;
  const double bytes = 4.0 * (rows * HUAL_D * ((a.pre_x ? 7.0 : 4.0) + a.nsteps) + (double)a.nsteps * HUAL_D * HUAL_D);
  const dim3 grid(xcd_clip_grid(a.R, a.Nv, a.MT));
  const bool six = ln_proj_bwd_six(a);
#define LN_BWD_NT(NT)                                                                                                   \
  {                                                                                                                     \
    if (a.pre_x) { HUAL_DYN_LDS((ln_proj_bwd_kernel<true, NT>), 160 * 1024); HUAL_LAUNCH(flops, bytes, (ln_proj_bwd_kernel<true, NT>), grid, dim3(CB_THREADS), lds, s, a, drop); } \
    else if (six) { HUAL_DYN_LDS((ln_proj_bwd_kernel<false, NT, true>), 160 * 1024); HUAL_LAUNCH(flops, bytes, (ln_proj_bwd_kernel<false, NT, true>), grid, dim3(CB_THREADS), lds, s, a, drop); } \
    else { HUAL_DYN_LDS((ln_proj_bwd_kernel<false, NT>), 160 * 1024); HUAL_LAUNCH(flops, bytes, (ln_proj_bwd_kernel<false, NT>), grid, dim3(CB_THREADS), lds, s, a, drop); }      \
    break;                                                                                                              \
  }
  switch (cdiv(a.MT, 16)) {      // row tiles per workgroup
    case 1: LN_BWD_NT(1)
    case 2: LN_BWD_NT(2)
    default: LN_BWD_NT(3)
  }
#undef LN_BWD_NT
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_da_mid_bwd(const DaMidBwdArgs& a, hipStream_t s) {
  HUAL_REQUIRE(a.dz1 && a.gate && a.val && a.sg && a.xg && a.sv && a.xv && a.R > 0, "da_mid_bwd: null / empty");
  HUAL_REQUIRE(a.MT >= 1 && a.MT <= DP_ROWS, "da_mid_bwd: MT must be 1..48");
  for (int k = 0; k < 10; ++k) HUAL_REQUIRE(a.w[k] != nullptr, "da_mid_bwd: null weight image");
  HUAL_REQUIRE(a.d_sc && a.d_val && a.d_ln1a && a.d_g && a.dz_sg && a.dz_xg && a.d_sv && a.d_xv && a.d_satt && a.d_xatt, "da_mid_bwd: null output");
  const size_t lds = (size_t)3 * DP_SLOT + (3 * DP_ROWS + 16 * DP_ROWS) * sizeof(float);
  const double rows = (double)a.R;
  const double flops = 10.0 * 2.0 * rows * HUAL_D * HUAL_D, bytes = 4.0 * (rows * HUAL_D * 17.0 + 10.0 * HUAL_D * HUAL_D);
  const dim3 grid(xcd_clip_grid(a.R, a.Nv, a.MT));
  switch (cdiv(a.MT, 16)) {      // row tiles per workgroup
    case 1: { HUAL_DYN_LDS(da_mid_bwd_kernel<1>, 160 * 1024); HUAL_LAUNCH(flops, bytes, da_mid_bwd_kernel<1>, grid, dim3(CB_THREADS), lds, s, a); break; }
    case 2: { HUAL_DYN_LDS(da_mid_bwd_kernel<2>, 160 * 1024); HUAL_LAUNCH(flops, bytes, da_mid_bwd_kernel<2>, grid, dim3(CB_THREADS), lds, s, a); break; }
    default: { HUAL_DYN_LDS(da_mid_bwd_kernel<3>, 160 * 1024); HUAL_LAUNCH(flops, bytes, da_mid_bwd_kernel<3>, grid, dim3(CB_THREADS), lds, s, a); break; }
  }
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
