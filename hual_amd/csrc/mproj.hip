// Multi-step dense kernel (see mproj.h).  Two operand slots are filled alternately from HBM rows - the rows of step k+1 are
// requested before step k's matrix phase and written behind it -; the weights never touch LDS: wave ws keeps the fragments of
// its 16 output columns in registers ("T-form", tilecore.h), requested one step ahead straight from the L2-resident image.
// One workgroup barrier per step (operand planes complete), no wait on vector memory in front of the matrix phase.
#include "mproj.h"
#include "tilecore.h"
#include "prof.h"
#include <cstdlib>
#include <cstddef>

using namespace hual;

// Every pointer of this kernel comes out of the LDS copy of the step descriptors (or sits next to them in the argument struct): the
// compiler cannot see its address space and would emit FLAT loads / stores, which count in lgkmcnt as well as vmcnt - every wait
// for an LDS read of the matrix phase would then wait for the operand prefetch too.  All memory traffic goes through the
// global-address-space helpers of common.h.
#define ld4 ld4_global
#define st4 st4_global

#define MP_ROWS 64
// compile-time feature set of a launch (any step of any problem uses ...): the loop body is straight-line for the features
// that are off and branch-free (pointer selects, predicated lanes) for those that are on - a uniform branch around a vector
// load makes the wait-count pass give up on the loads in flight across it (vmcnt(0) right behind the prefetch)
enum { MPF_A2 = 1, MPF_BF16 = 2, MPF_DROP = 4, MPF_ADD = 8, MPF_LN = 16, MPF_REUSE = 32, MPF_QUAD = 64, MPF_POOL = 128 };

// NT = row tiles of a workgroup (MT <= 16 NT)
// (tile < 0: the workgroup derives its tile from blockIdx.x - the XCD-aware order over this problem's own tiles, common.h)
template <int NT, int F>
__device__ __forceinline__ void mproj_body(const MProjArgs& a, const DropCfg& drop, int tile = -1) {
  extern __shared__ __attribute__((aligned(16))) char mp_lds[];
  char* S0 = mp_lds;                                   // operand slot 0: hi | lo planes [64][256 B]; LN mode: x as fp32 rows
  char* S1 = S0 + 2 * MP_ROWS * 256;
  float* ainv0 = reinterpret_cast<float*>(S1 + 2 * MP_ROWS * 256);
  float* ainv1 = ainv0 + MP_ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  const int MT = a.MT, R = a.R;
  if (tile < 0) {
    const int nblk = xcd_round8((R + MT - 1) / MT);
    if ((int)blockIdx.x >= nblk) return;
    tile = xcd_tile(blockIdx.x, nblk);
  }
  const int r0 = tile * MT;
  if (r0 >= R) return;
  const int RE = min(R, r0 + MT);
  const int j = lane & 15, g = lane >> 4, ecol = 16 * wave + 4 * g;      // T-form: lane = row j of a tile, columns ecol .. ecol + 3
  const DropRegs dr = drop_load(drop);
  constexpr int NU = NT;                               // 16-row groups a thread loads / fills (MT <= 16 NT)

  // the step descriptors go through LDS once: a scalar load of a kernel-argument line that is not in the scalar cache yet is a
  // full memory round trip, and a walk over the descriptors takes one (or two, dependent) of those per step
  constexpr int SW = sizeof(MProjStep) / 4;
  __shared__ uint32_t sdesc[MP_MAX * SW];
  {
    const uint32_t* ka = reinterpret_cast<const uint32_t*>(&a.s[0]);
    for (int i = threadIdx.x; i < a.nsteps * SW; i += CB_THREADS) sdesc[i] = ka[i];
  }
  __syncthreads();
  // (one LDS read per lane - lane q holds dword q of the descriptor - and a readlane per field: reading the dwords one by one
  //  was a chain of ~10 dependent LDS round trips per step)
  static_assert(SW <= 64, "a step descriptor fits one dword per lane");
  auto step_at = [&](int si) {
    struct alignas(8) Raw { uint32_t w[SW]; } r;
    const uint32_t mine = sdesc[si * SW + min(lane, SW - 1)];
#pragma unroll
    for (int q = 0; q < SW; ++q) r.w[q] = __builtin_amdgcn_readlane(mine, q);
    return __builtin_bit_cast(MProjStep, r);
  };
  // step descriptors with the repetitions unrolled: (descriptor index, repetition) of flat step k
  int nflat = 0;
  for (int i = 0; i < a.nsteps; ++i) nflat += (int)__builtin_amdgcn_readfirstlane(sdesc[i * SW + offsetof(MProjStep, rep) / 4]);
  auto expand = [&](int si, int ri) {
    MProjStep st = step_at(si);
    if (st.rep > 1) {
      st.A = st.a_bf16 ? (const void*)(reinterpret_cast<const uint16_t*>(st.A) + 128 * ri) : (const void*)(reinterpret_cast<const float*>(st.A) + 128 * ri);
      st.kw = min(128, st.ktot - 128 * ri);
      st.col0 += 128 * ri;
      st.wimg = reinterpret_cast<const float*>(reinterpret_cast<const char*>(st.wimg) + (size_t)ri * 128 * 512);
      st.first = st.first && ri == 0;
      st.last = st.last && ri == st.rep - 1;
    }
    return st;
  };
  TfW wc, wn;
  tf_load_w(wc, step_at(0).wimg, wave, lane);
  // raw operand rows of one step -> registers (unconditional loads on clamped rows / columns)
  float4 nv[NU], n2[NU];
  auto rows_load = [&](const MProjStep& st) {      // (bfloat16 rows travel raw in .x / .y, widened in fill)
    const int kc = min(col, max(st.kw - 4, 0));        // kw is a multiple of 4: a lane's 4 columns are in or out together
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const size_t row = (size_t)min(r0 + grp + 16 * u, R - 1);
      if ((F & MPF_BF16) && st.a_bf16) {      // (the widening happens at the consumer: the raw 8 bytes travel in .x / .y)
        const uint2 raw = ld2_global(reinterpret_cast<const uint16_t*>(st.A) + row * st.lda + kc);
        nv[u] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), 0.f, 0.f);
      } else {
        nv[u] = ld4(reinterpret_cast<const float*>(st.A) + row * st.lda + kc);
      }
      if (F & MPF_A2) {      // no factor: the operand itself is read once more (a hit) and not used
        const float* p2 = st.A2 ? st.A2 : reinterpret_cast<const float*>(st.A);
        const int ld2 = st.A2 ? st.lda2 : st.lda;
        n2[u] = ld4(p2 + row * ld2 + kc);
      }
    }
  };
  // registers -> operand planes of slot `slot` (prologue: factor, dropout; columns >= kw and rows beyond the tensor are zero)
  auto fill = [&](const MProjStep& st, int slot) {
    char* S = slot ? S1 : S0;
    float* ai = slot ? ainv1 : ainv0;
    const bool cin = col < st.kw;
    uint32_t nb[4] = {15u, 15u, 15u, 15u};
    constexpr int NPR = (NU + 1) / 2;
    if ((F & MPF_DROP) && st.drop_site >= 0 && dr.enabled) {
      const uint32_t c4 = (uint32_t)((st.col0 + col) >> 2);
#pragma unroll
      for (int pr = 0; pr < NPR; ++pr) {
        const int lrA = grp + 32 * pr, lrB = lrA + 16;
        const bool okA = cin && lrA < MT && r0 + lrA < RE, okB = cin && lrB < MT && r0 + lrB < RE;
        uint32_t na, nbb;
        const uint32_t byte = drop_nib2_r(dr, (uint32_t)st.drop_site, a.drop_row0 + (uint32_t)(r0 + lrA), a.drop_row0 + (uint32_t)(r0 + lrB), c4, na, nbb);
        nb[2 * pr] = na; nb[2 * pr + 1] = nbb;
        if (st.keep_out) {      // even lane: row A's byte, odd lane: row B's (tilecore.h drop_nib2_r)
          const bool odd = (c4 & 1u) != 0u;
          if (odd ? okB : okA) st1b_global(st.keep_out + (size_t)(r0 + (odd ? lrB : lrA)) * st.ld_keep + (c4 >> 1), (uint8_t)byte);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int lr = grp + 16 * u, row = r0 + lr;
      const bool ok = row < RE && cin && lr < MT;
      float4 v = nv[u];
      if ((F & MPF_BF16) && st.a_bf16) {
        const uint32_t w0 = __float_as_uint(v.x), w1 = __float_as_uint(v.y);
        v = make_float4(__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u), __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u));
      }
      if ((F & MPF_A2) && st.A2) v = cb_mul(v, n2[u]);
      if (!ok) v = f4zero();
      if ((F & MPF_DROP) && st.drop_site >= 0 && dr.enabled) v = f4_select(nb[u], make_float4(v.x * dr.scale, v.y * dr.scale, v.z * dr.scale, v.w * dr.scale));
      const float inv = cb_store_operand(S, S + MP_ROWS * 256, lr, l32, v);      // (rows MT .. 16 NT - 1 of a slot: zeros)
      if (l32 == 0) ai[lr] = row < RE ? inv : 0.f;
    }
  };
#define MP_STAMP(i) HUAL_STAMP_K(9, i)
  MP_STAMP(0);
  MProjStep cur = expand(0, 0);
  rows_load(cur);
  fill(cur, 0);
  MP_STAMP(1);
  int slot = 0, si = 0, ri = 0;
  float4 acc[NT];
  float4 qd[(F & MPF_QUAD) ? 4 : 1][NT];               // quad epilogue: the closed tiles of the four steps
  float4 lastb = f4zero();                             // bias of the step in flight (the LN / pool epilogues add the last step's)
  constexpr bool QPRE = (F & MPF_QUAD) && NT <= 3;     // (four row tiles: the 48 extra registers would spill)
  float4 qx[QPRE ? NT : 1], qc[QPRE ? NT : 1], qq[QPRE ? NT : 1];
  if (QPRE && a.quad_x) {      // what the quad epilogue reads: requested here, arrives under the four steps
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const size_t off = (size_t)min(r0 + 16 * rt + j, R - 1) * HUAL_D + ecol;
      qx[rt] = ld4(a.quad_x + off); qc[rt] = ld4(a.quad_c2q + off); qq[rt] = ld4(a.quad_q2c + off);
    }
  }
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) acc[rt] = f4zero();
#pragma unroll 1
  for (int k = 0; k < nflat; ++k) {
    MP_STAMP(2 + 7 * k);
    const MProjStep st = cur;
    const bool more = k + 1 < nflat;
    MProjStep nxt = st;
    if (more) {
      if (++ri == (int)__builtin_amdgcn_readfirstlane(sdesc[si * SW + offsetof(MProjStep, rep) / 4])) { ri = 0; ++si; }
      nxt = expand(si, ri);
    }
    MP_STAMP(3 + 7 * k);
    // what the closing phase reads from memory, requested FIRST: vmcnt counts in order, so a wait for
    // these must not have the prefetches below in front of it
    const bool closes = st.last && st.out && ecol < st.ncol;
    const float* wdummy = st.wimg + 4 * lane;
    const float4 bias = ld4(st.bias ? st.bias + ecol : wdummy);
    float4 addv[NT];
    if (F & MPF_ADD) {
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) {
        const int row = min(r0 + 16 * rt + j, R - 1);
        addv[rt] = ld4((st.add && closes) ? st.add + (size_t)(row / st.add_div) * st.ldadd + ecol : wdummy);
      }
    }
    // next step's weight fragments and operand rows: in flight under this step (the last step asks for its own again)
    tf_load_w(wn, nxt.wimg, wave, lane);
    bool refill = true;
    if (F & MPF_REUSE) {
      refill = more && !nxt.reuse;
      if (refill) rows_load(nxt);
    } else {
      rows_load(nxt);
    }
    MP_STAMP(4 + 7 * k);
    cb_barrier();                                       // operand planes of this step complete (LDS only: nothing waits for HBM here)
    MP_STAMP(5 + 7 * k);
    const char* S = slot ? S1 : S0;
    const float* ai = slot ? ainv1 : ainv0;
    f32x4 accp[NT];
    tf_mma<NT, MP_ROWS * 256>(S, wc, lane, accp);
    MP_STAMP(6 + 7 * k);
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const float ir = ai[16 * rt + j];
      const float4 o = st.first ? f4zero() : acc[rt];
      acc[rt] = make_float4(fmaf(accp[rt][0], ir, o.x), fmaf(accp[rt][1], ir, o.y), fmaf(accp[rt][2], ir, o.z), fmaf(accp[rt][3], ir, o.w));
    }
    lastb = st.bias ? bias : f4zero();
    {      // close the tile: bias, relu, addend, store (lanes outside the tensor / a step that does not close: no store)
      float* outp = st.out ? st.out : const_cast<float*>(wdummy);
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) {
        const int row = r0 + 16 * rt + j;
        float4 v = st.bias ? cb_add(acc[rt], bias) : acc[rt];
        if (st.act) v = relu_nan4(v);
        if ((F & MPF_ADD) && st.add) v = cb_add(v, addv[rt]);
        if (closes && 16 * rt + j < MT && row < RE) st4(outp + (size_t)row * st.ldo + ecol, v);
        if (F & MPF_QUAD) {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (k == q) qd[q][rt] = v;
        }
      }
    }
    MP_STAMP(7 + 7 * k);
    if (refill && more) { slot ^= 1; fill(nxt, slot); }      // the other slot was last read by product k - 1: every wave is past it (barrier above)
    MP_STAMP(8 + 7 * k);
    wc = wn;
    cur = nxt;
  }
  if ((F & MPF_QUAD) && a.quad_x) {      // the four tiles leave split (mproj.h)
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const int row = r0 + 16 * rt + j;
      if (16 * rt + j >= MT || row >= RE) continue;
      const size_t off = (size_t)row * HUAL_D + ecol;
      const float4 x = QPRE ? qx[QPRE ? rt : 0] : ld4(a.quad_x + off), c2q = QPRE ? qc[QPRE ? rt : 0] : ld4(a.quad_c2q + off),
                   q2c = QPRE ? qq[QPRE ? rt : 0] : ld4(a.quad_q2c + off);
      const float4 d0 = qd[0][rt], d1 = qd[1][rt], d2 = qd[2][rt], d3 = qd[3][rt];
      st4(a.quad_dc2q + off, make_float4(d1.x + d2.x * x.x, d1.y + d2.y * x.y, d1.z + d2.z * x.z, d1.w + d2.w * x.w));
      st4(a.quad_dq2c + off, make_float4(d3.x * x.x, d3.y * x.y, d3.z * x.z, d3.w * x.w));
      st4(a.quad_dx + off, make_float4(d0.x + d2.x * c2q.x + d3.x * q2c.x, d0.y + d2.y * c2q.y + d3.y * q2c.y,
                                       d0.z + d2.z * c2q.z + d3.z * q2c.z, d0.w + d2.w * c2q.w + d3.w * q2c.w));
    }
    return;
  }
  if ((F & MPF_POOL) && a.pool_cat) {      // relu + max over the window starts of a word (mproj.h): lanes j of a tile = rows
    const float4 bias = lastb;                           // (loaded with the last step's operands: no round trip here)
    const int C = a.pool_C;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const int lr = 16 * rt + j, row = r0 + lr;
      const int p = lr & (C - 1);                        // window start inside the word (MT % C == 0: words do not straddle workgroups)
      const float4 v4 = cb_add(acc[rt], bias);
      const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
      float best[4];
      int arg[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = ecol + q;
        const int k = ch < 10 ? 1 : (ch < 30 ? 2 : (ch < 60 ? 3 : 4));      // filter width of the channel's bank
        best[q] = (p + k <= C) ? vv[q] : -INFINITY;
        arg[q] = p;
      }
      // butterfly over the C rows of the word (DPP: lanes j of a 16-lane row): larger value, earlier start on ties
      auto merge = [&](int step) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float ov = dpp_xor_partner(best[q], step);
          const int oa = __builtin_bit_cast(int, dpp_xor_partner(__builtin_bit_cast(float, arg[q]), step));
          const bool take = ov > best[q] || (ov == best[q] && oa < arg[q]);
          best[q] = take ? ov : best[q];
          arg[q] = take ? oa : arg[q];
        }
      };
      if (C > 1) merge(1);
      if (C > 2) merge(2);
      if (C > 4) merge(4);
      if (C > 8) merge(8);
      if (p == 0 && lr < MT && row < RE && ecol < 100) {
        const int word = row / C;
        float o[4];
        int oa[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool pos = best[q] > 0.f;
          o[q] = pos ? best[q] : 0.f;
          oa[q] = pos ? arg[q] : -1;
        }
        float* cp = a.pool_cat + (size_t)word * a.pool_ldcat + a.pool_col0 + ecol;
        if (((a.pool_ldcat | a.pool_col0) & 3) == 0) st4(cp, make_float4(o[0], o[1], o[2], o[3]));      // (16-byte aligned rows: one store)
        else { st1f_global(cp, o[0]); st1f_global(cp + 1, o[1]); st1f_global(cp + 2, o[2]); st1f_global(cp + 3, o[3]); }
        st4i_global(a.pool_arg + (size_t)word * 100 + ecol, oa[0], oa[1], oa[2], oa[3]);
      }
    }
    return;
  }
  if (!(F & MPF_LN) || !a.ln_g) return;
  // ---- LN mode: the last tile (+ bias) -> LDS as fp32 rows -> layer norm (+ position embeddings) row by row
  float4* D0 = reinterpret_cast<float4*>(S0);                // both slots are free behind the last matrix phase (barrier above)
  cb_barrier();                                         // every wave is past the last matrix phase
  {
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) D0[(16 * rt + j) * 32 + (ecol >> 2)] = cb_add(acc[rt], lastb);
  }
  const float4 gam = ld4(a.ln_g + col), bet = ld4(a.ln_b + col);
  cb_barrier();
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int lr = grp + 16 * u, row = r0 + lr;
    if (lr >= MT || row >= RE) continue;
    const float4 x = D0[lr * 32 + l32];
    float mean, rstd;
    const float m0 = fast_sum32(cb_hsum(x)) * (1.0f / HUAL_D);
    const float4 d = make_float4(x.x - m0, x.y - m0, x.z - m0, x.w - m0);
    const float var = fast_sum32(cb_hsum(cb_mul(d, d))) * (1.0f / HUAL_D);
    mean = m0; rstd = rsqrtf(var + LN_EPS);
    float4 y = cb_fma(make_float4(d.x * rstd, d.y * rstd, d.z * rstd, d.w * rstd), gam, bet);
    if (a.pos) y = cb_add(y, ld4(a.pos + (size_t)((row + a.row_in_clip0) % a.Tc) * HUAL_D + col));
    const size_t off = (size_t)row * HUAL_D + col;
    if (a.x_out) st4(a.x_out + off, x);
    st4(a.y_out + off, y);
    if (l32 == 0) { st1f_global(a.mean + row, mean); st1f_global(a.rstd + row, rstd); }
  }
}

template <int NT, int F>
__global__ __launch_bounds__(CB_THREADS) void mproj_kernel(MProjArgs a, DropCfg drop) { mproj_body<NT, F>(a, drop); }
// Two problems in one launch, ONE grid row: XCD x (= blockIdx.x & 7, the dispatcher's round robin) takes its eighth of the tiles of
// problem 0, then its eighth of problem 1 (mproj_pair_tiles: the same split on the host sizes the grid).  Round 5: the launch used to
// be (tiles of the LARGER problem) x 2 with the second row mostly empty - at 8192 + 640 rows 221 of 480 workgroups returned at once,
// but each of them takes a CU slot (96 KB of LDS: one per CU) through dispatch, argument load and exit, and the 19 real workgroups of
// the second row queued behind them: 29.9 / 28.2 us against 19.1 / 17.9 us for the same work at 8192 + 1280 rows.
__host__ __device__ inline int mproj_pair_lo(int nb, int x) { return nb * x / 8; }
template <int NT, int F>
__global__ __launch_bounds__(CB_THREADS) void mproj_pair_kernel(MProjArgs a0, MProjArgs a1, DropCfg drop) {
  const int x = blockIdx.x & 7;
  int sl = blockIdx.x >> 3;
  const int nb0 = (a0.R + a0.MT - 1) / a0.MT, nb1 = (a1.R + a1.MT - 1) / a1.MT;
  const int lo0 = mproj_pair_lo(nb0, x), n0 = mproj_pair_lo(nb0, x + 1) - lo0;
  if (sl < n0) { mproj_body<NT, F>(a0, drop, lo0 + sl); return; }
  sl -= n0;
  const int lo1 = mproj_pair_lo(nb1, x), n1 = mproj_pair_lo(nb1, x + 1) - lo1;
  if (sl < n1) mproj_body<NT, F>(a1, drop, lo1 + sl);
}

#undef ld4
#undef st4

#if defined(HUAL_STAMPS) && HUAL_STAMPS == 9
extern "C" int hual_debug_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hual_stamps), sizeof(unsigned long long) * (size_t)n);
}
#endif

namespace hual {

// rows per workgroup for problems of R0 (+ R1) rows in one launch: the smallest tile (>= 16) with which ALL workgroups of the
// launch are resident at once - one per CU; a launch with one workgroup more than there are CUs takes two rounds
// workgroups of a pair launch: 8 x the largest per-XCD count of the two problems' tiles (mproj_pair_kernel)
static int mproj_pair_grid(int R0, int R1, int t) {
  const int nb0 = cdiv(R0, t), nb1 = cdiv(R1, t);
  int m = 0;
  for (int x = 0; x < 8; ++x) {
    const int n = (mproj_pair_lo(nb0, x + 1) - mproj_pair_lo(nb0, x)) + (mproj_pair_lo(nb1, x + 1) - mproj_pair_lo(nb1, x));
    m = n > m ? n : m;
  }
  return 8 * m;
}
int mproj_rows(int R0, int R1) {
  int t = 16;
  if (R1 > 0) { while (t < MP_ROWS && mproj_pair_grid(R0, R1, t) > 256) ++t; }      // (no XCD gets more than its 32 CUs' worth)
  else { while (t < MP_ROWS && cdiv(R0, t) > 256) ++t; }
  return t;
}

static int check_mproj(const MProjArgs& a, double& flops, double& bytes, int& feat) {
  HUAL_REQUIRE(a.nsteps >= 1 && a.nsteps <= MP_MAX && a.R > 0 && a.MT >= 1 && a.MT <= MP_ROWS, "mproj: steps / rows");
  HUAL_REQUIRE(a.s[0].first && !a.s[0].reuse && a.s[a.nsteps - 1].last, "mproj: first step starts a tile, last step closes one");
  for (int k = 0; k < a.nsteps; ++k) {
    const MProjStep& s = a.s[k];
    HUAL_REQUIRE(s.rep >= 1 && (s.rep == 1 || (!s.reuse && s.ktot > 128 * (s.rep - 1) && s.ktot <= 128 * s.rep && (s.ktot % 4) == 0 && !s.A2)), "mproj: repetitions");
    HUAL_REQUIRE(s.reuse || (s.A && (s.rep > 1 || (s.kw >= 4 && s.kw <= 128 && (s.kw % 4) == 0)) && (s.lda % 4) == 0), "mproj: operand");
    HUAL_REQUIRE(!s.reuse || k > 0, "mproj: nothing to reuse");
    HUAL_REQUIRE(s.wimg && s.wrows >= 1, "mproj: weight image");
    feat |= (s.A2 ? MPF_A2 : 0) | (s.a_bf16 ? MPF_BF16 : 0) | (s.drop_site >= 0 ? MPF_DROP : 0) | (s.add ? MPF_ADD : 0) | (s.reuse ? MPF_REUSE : 0);
    HUAL_REQUIRE(!s.last || (a.ln_g && k == a.nsteps - 1 && !s.out) || (a.quad_x && !s.out) || (a.pool_cat && k == a.nsteps - 1 && !s.out) ||
                 (s.out && (s.ldo % 4) == 0 && s.ncol >= 4 && s.ncol <= 128 && (s.ncol % 4) == 0), "mproj: closing step needs a destination");
    HUAL_REQUIRE(!s.add || s.add_div >= 1, "mproj: add_div");
    HUAL_REQUIRE(k == 0 || s.first == a.s[k - 1].last, "mproj: a tile starts exactly behind a closed one");
    const double kdeep = s.rep > 1 ? s.ktot : (s.reuse ? a.s[k - 1].kw : s.kw);
    flops += 2.0 * a.R * kdeep * 128.0;
    if (!s.reuse) bytes += (s.a_bf16 ? 2.0 : 4.0) * a.R * kdeep + (s.A2 ? 4.0 * a.R * kdeep : 0.0);
    bytes += 4.0 * 128.0 * (s.rep > 1 ? s.ktot : (s.wrows < 128 ? s.wrows : 128));
    if (s.last && s.out) bytes += 4.0 * a.R * s.ncol;
  }
  if (a.quad_x) {
    feat |= MPF_QUAD;
    HUAL_REQUIRE(a.nsteps == 4 && a.quad_c2q && a.quad_q2c && a.quad_dc2q && a.quad_dq2c && a.quad_dx && !a.ln_g, "mproj: quad epilogue");
    for (int k = 0; k < 4; ++k) HUAL_REQUIRE(a.s[k].first && a.s[k].last && a.s[k].rep == 1 && !a.s[k].out, "mproj: quad epilogue takes four closing steps");
    bytes += 4.0 * a.R * 128.0 * 6.0;
  }
  if (a.pool_cat) {
    feat |= MPF_POOL;
    const int C = a.pool_C;
    HUAL_REQUIRE(a.pool_arg && (C == 1 || C == 2 || C == 4 || C == 8 || C == 16) && (a.MT % C) == 0 && (a.R % C) == 0 && !a.ln_g && !a.quad_x,
                 "mproj: pool epilogue needs a power-of-two window count <= 16 that divides the rows of a workgroup");
    bytes += 8.0 * (a.R / C) * 100.0;
  }
  if (a.ln_g) {
    feat |= MPF_LN;
    HUAL_REQUIRE(a.ln_b && a.y_out && a.mean && a.rstd && (!a.pos || a.Tc >= 1), "mproj: layer-norm outputs");
    bytes += 4.0 * a.R * 128.0 * (a.x_out ? 2.0 : 1.0);
  }
  return 0;
}

int launch_mproj(const MProjArgs* a, int nprob, const DropCfg& drop, hipStream_t s) {
  HUAL_REQUIRE(a && (nprob == 1 || nprob == 2), "mproj: one or two problems");
  double flops = 0.0, bytes = 0.0;
  int blocks = 0, feat = 0;
  for (int i = 0; i < nprob; ++i) {
    int rc = check_mproj(a[i], flops, bytes, feat);
    if (rc) return rc;
    const int nb = cdiv(a[i].R, a[i].MT);
    blocks = nb > blocks ? nb : blocks;
  }
  blocks = xcd_round8(blocks);
  const size_t lds = (size_t)4 * MP_ROWS * 256 + 2 * MP_ROWS * sizeof(float);
  if (nprob == 2) HUAL_REQUIRE(a[0].MT == a[1].MT, "mproj: problems of one launch share the rows per workgroup");
  const int nt = (a[0].MT + 15) / 16;
  // instantiated feature sets: none, reuse, factor, addend, reuse + addend, everything
  const int fsets[] = {0, MPF_REUSE, MPF_A2, MPF_ADD, MPF_REUSE | MPF_ADD, MPF_A2 | MPF_BF16 | MPF_DROP | MPF_ADD | MPF_LN | MPF_REUSE,
                       MPF_REUSE | MPF_QUAD};
  int fs = 5;
  for (int i = 4; i >= 0; --i) if ((feat & ~fsets[i]) == 0) fs = i;
  if (feat & MPF_QUAD) {
    HUAL_REQUIRE((feat & ~fsets[6]) == 0, "mproj: the quad epilogue goes with plain / reused operands only");
    fs = 6;
  }
  if (feat & MPF_POOL) {
    HUAL_REQUIRE(feat == MPF_POOL, "mproj: the pool epilogue goes with plain operands only");
    fs = 7;
  }
#define MPROJ_LAUNCH(NT, FS)                                                                                               \
  do {                                                                                                                     \
    HUAL_DYN_LDS((mproj_kernel<NT, FS>), 96 * 1024);                                                                       \
    HUAL_DYN_LDS((mproj_pair_kernel<NT, FS>), 96 * 1024);                                                                  \
    if (nprob == 1) HUAL_LAUNCH(flops, bytes, (mproj_kernel<NT, FS>), dim3(blocks), dim3(CB_THREADS), lds, s, a[0], drop); \
    else HUAL_LAUNCH(flops, bytes, (mproj_pair_kernel<NT, FS>), dim3(mproj_pair_grid(a[0].R, a[1].R, a[0].MT)), dim3(CB_THREADS), lds, s, a[0], a[1], drop); \
  } while (0)
#define MPROJ_NT(FS)                                                                                                       \
  do {                                                                                                                     \
    if (nt <= 2) MPROJ_LAUNCH(2, FS);                                                                                      \
    else if (nt == 3) MPROJ_LAUNCH(3, FS);                                                                                 \
    else MPROJ_LAUNCH(4, FS);                                                                                              \
  } while (0)
  // (numeric literals: the launch macro stringifies its kernel argument for the profiler, and the names must be the ones rocprofv3
  //  prints - mproj_kernel<2, 32>, not mproj_kernel<2, MPF_REUSE>)
  static_assert(MPF_A2 == 1 && MPF_ADD == 8 && MPF_REUSE == 32 && MPF_QUAD == 64 && MPF_POOL == 128 && (MPF_A2 | MPF_BF16 | MPF_DROP | MPF_ADD | MPF_LN | MPF_REUSE) == 63,
                "feature-set literals below");
  switch (fs) {
    case 0: MPROJ_NT(0); break;
    case 1: MPROJ_NT(32); break;      // reuse
    case 2: MPROJ_NT(1); break;       // operand factor
    case 3: MPROJ_NT(8); break;       // addend
    case 4: MPROJ_NT(40); break;      // reuse + addend
    case 6: MPROJ_NT(96); break;      // reuse + quad epilogue
    case 7: MPROJ_NT(128); break;     // pool epilogue
    default: MPROJ_NT(63); break;     // everything but the quad epilogue
  }
#undef MPROJ_NT
#undef MPROJ_LAUNCH
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
