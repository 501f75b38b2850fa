// Multi-step dense kernel (see mproj.h).  Product loop of ln_proj_bwd_kernel (dablock.hip): two operand slots filled
// alternately from HBM rows - the rows of step k+1 are requested before step k's wait and written behind its matrix phase -,
// one weight image at a time by LDS-DMA, wave (mt, ch) = 16 rows x 64 columns on the matrix cores.
#include "mproj.h"
#include "tilecore.h"
#include "prof.h"

using namespace hual;

#define MP_ROWS 64

// LDS-DMA of a weight image whose K rows beyond `wrows` do not exist (K not a multiple of 128): those tile rows are fetched from
// the last valid row (finite numbers; the operand columns they meet are zero).  Same piece order as cb_dma_weight.
__device__ __forceinline__ void mp_dma_weight(const float* wimg, char* Wl, int wave, int lane, int wrows) {
  if (wrows >= 128) { cb_dma_weight(wimg, Wl, wave, lane, CB_THREADS / 64); return; }
  const int chp = lane & 15, rr = lane >> 4;
  const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds_addr_of(Wl) + 1024u * (uint32_t)wave);
  const int sw = (rr << 2) | (wave & 3);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int row = 4 * wave + rr + 64 * (k >> 2) + 32 * (k & 1);
    const int kk = min(row, wrows - 1);
    const char* src = reinterpret_cast<const char*>(wimg) + (size_t)kk * 512 + 256 * ((k >> 1) & 1) + 16 * (chp ^ sw);
    glds16_asm(src, ldsw + (uint32_t)(CB_STAGE * (k >> 2) + 8192 * (k & 3)));
  }
}

__device__ __forceinline__ void mproj_body(const MProjArgs& a, const DropCfg& drop) {
  extern __shared__ __attribute__((aligned(16))) char mp_lds[];
  char* S0 = mp_lds;                                   // operand slot 0: hi | lo planes [64][256 B]; LN mode: x as fp32 rows
  char* S1 = S0 + 2 * MP_ROWS * 256;
  char* Wl = S1 + 2 * MP_ROWS * 256;
  float* ainv0 = reinterpret_cast<float*>(Wl + CB_WBYTES);
  float* ainv1 = ainv0 + MP_ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  const int MT = a.MT, ntile = (MT + 15) >> 4, R = a.R;
  const int r0 = blockIdx.x * MT;
  if (r0 >= R) return;                                 // (pair launches: the smaller problem has fewer workgroups)
  const int RE = min(R, r0 + MT);
  const int mt = wave >> 1, ch = wave & 1;
  const int j = lane & 15, g = lane >> 4, ecol = 64 * ch + 4 * j;
  const bool act = mt < ntile;
  const DropRegs dr = drop_load(drop);

  // step descriptors with the repetitions unrolled: (descriptor index, repetition) of flat step k
  int nflat = 0;
  for (int i = 0; i < a.nsteps; ++i) nflat += a.s[i].rep;
  auto expand = [&](int si, int ri) {
    MProjStep st = a.s[si];
    if (st.rep > 1) {
      st.A = st.a_bf16 ? (const void*)(reinterpret_cast<const uint16_t*>(st.A) + 128 * ri) : (const void*)(reinterpret_cast<const float*>(st.A) + 128 * ri);
      st.kw = min(128, st.ktot - 128 * ri);
      st.col0 += 128 * ri;
      st.wimg = reinterpret_cast<const float*>(reinterpret_cast<const char*>(st.wimg) + (size_t)ri * 128 * 512);
      st.wrows = st.wrows - 128 * ri;
      st.first = st.first && ri == 0;
      st.last = st.last && ri == st.rep - 1;
    }
    return st;
  };
  mp_dma_weight(a.s[0].wimg, Wl, wave, lane, a.s[0].wrows);
  // raw operand rows of one step -> registers (unconditional loads on clamped rows / columns)
  float4 nv[4], n2[4];
  auto rows_load_to = [&](const MProjStep& st, float4 (&dst)[4]) {      // (bfloat16 rows travel raw in .x / .y, widened in fill)
    const int kc = min(col, max(st.kw - 4, 0));        // kw is a multiple of 4: a lane's 4 columns are in or out together
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t row = (size_t)min(r0 + grp + 16 * u, R - 1);
      if (st.a_bf16) {      // (the widening happens at the consumer: the raw 8 bytes travel in .x / .y)
        const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(st.A) + row * st.lda + kc);
        dst[u] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), 0.f, 0.f);
      } else {
        dst[u] = ld4(reinterpret_cast<const float*>(st.A) + row * st.lda + kc);
      }
    }
  };
  auto rows_load = [&](const MProjStep& st) {
    rows_load_to(st, nv);
    if (st.A2) {
      const int kc = min(col, max(st.kw - 4, 0));
#pragma unroll
      for (int u = 0; u < 4; ++u) n2[u] = ld4(st.A2 + (size_t)min(r0 + grp + 16 * u, R - 1) * st.lda2 + kc);
    }
  };
  // registers -> operand planes of slot `slot` (prologue: factor, dropout; columns >= kw and rows beyond the tensor are zero)
  auto fill = [&](const MProjStep& st, int slot) {
    char* S = slot ? S1 : S0;
    float* ai = slot ? ainv1 : ainv0;
    const bool cin = col < st.kw;
    uint32_t nb[4] = {15u, 15u, 15u, 15u};
    if (st.drop_site >= 0 && dr.enabled) {
      const uint32_t c4 = (uint32_t)((st.col0 + col) >> 2);
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int lrA = grp + 32 * pr, lrB = lrA + 16;
        const bool okA = cin && lrA < MT && r0 + lrA < RE, okB = cin && lrB < MT && r0 + lrB < RE;
        uint32_t na, nbb;
        const uint32_t byte = drop_nib2_r(dr, (uint32_t)st.drop_site, a.drop_row0 + (uint32_t)(r0 + lrA), a.drop_row0 + (uint32_t)(r0 + lrB), c4, na, nbb);
        nb[2 * pr] = na; nb[2 * pr + 1] = nbb;
        if (st.keep_out) {      // even lane: row A's byte, odd lane: row B's (tilecore.h drop_nib2_r)
          const bool odd = (c4 & 1u) != 0u;
          if (odd ? okB : okA) st.keep_out[(size_t)(r0 + (odd ? lrB : lrA)) * st.ld_keep + (c4 >> 1)] = (uint8_t)byte;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int lr = grp + 16 * u, row = r0 + lr;
      if (lr >= MT) continue;
      const bool ok = row < RE && cin;
      float4 v = nv[u];
      if (st.a_bf16) {
        const uint32_t w0 = __float_as_uint(v.x), w1 = __float_as_uint(v.y);
        v = make_float4(__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u), __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u));
      }
      if (!ok) v = f4zero();
      if (st.A2 && ok) v = cb_mul(v, n2[u]);
      if (st.drop_site >= 0 && dr.enabled) v = f4_select(nb[u], make_float4(v.x * dr.scale, v.y * dr.scale, v.z * dr.scale, v.w * dr.scale));
      const float inv = cb_store_operand(S, S + MP_ROWS * 256, lr, l32, v);
      if (l32 == 0) ai[lr] = row < RE ? inv : 0.f;
    }
  };
  MProjStep cur = expand(0, 0);
  rows_load(cur);
  fill(cur, 0);
  int slot = 0, si = 0, ri = 0;
  float4 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = f4zero();
#pragma unroll 1
  for (int k = 0; k < nflat; ++k) {
    const MProjStep st = cur;
    const bool more = k + 1 < nflat;
    MProjStep nxt = st;
    if (more) {
      if (++ri == a.s[si].rep) { ri = 0; ++si; }
      nxt = expand(si, ri);
    }
    const bool refill = more && !nxt.reuse;
    if (refill) rows_load(nxt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    cb_barrier();
    const char* S = slot ? S1 : S0;
    const float* ai = slot ? ainv1 : ainv0;
    f32x4 accp[4];
    if (act) cb_tile_mma_t<MP_ROWS * 256>(S, Wl, mt, ch, lane, accp);
    cb_barrier();
    if (more) mp_dma_weight(nxt.wimg, Wl, wave, lane, nxt.wrows);
    if (act) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ir = ai[16 * mt + 4 * g + r];
        const float4 o = acc[r];
        acc[r] = st.first ? make_float4(fmaf(accp[0][r], ir, 0.f), fmaf(accp[1][r], ir, 0.f), fmaf(accp[2][r], ir, 0.f), fmaf(accp[3][r], ir, 0.f))
                          : make_float4(fmaf(accp[0][r], ir, o.x), fmaf(accp[1][r], ir, o.y), fmaf(accp[2][r], ir, o.z), fmaf(accp[3][r], ir, o.w));
      }
      if (st.last && st.out) {      // close the tile: bias, relu, addend, store
        const float4 bias = st.bias ? ld4(st.bias + ecol) : f4zero();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = r0 + 16 * mt + 4 * g + r;
          if (row >= RE || ecol >= st.ncol) continue;
          float4 v = cb_add(acc[r], bias);
          if (st.act) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
          if (st.add) v = cb_add(v, ld4(st.add + (size_t)(row / st.add_div) * st.ldadd + ecol));
          st4(st.out + (size_t)row * st.ldo + ecol, v);
        }
      }
    }
    if (refill) { slot ^= 1; fill(nxt, slot); }      // the other slot was last read by product k - 1 (or never)
    cur = nxt;
  }
  if (!a.ln_g) return;
  // ---- LN mode: the last tile (+ bias) -> LDS as fp32 rows -> layer norm (+ position embeddings) row by row
  float4* D0 = reinterpret_cast<float4*>(S0);                // both slots are free behind the last matrix phase (barrier above)
  cb_barrier();
  if (act) {
    const float* lb = a.s[a.nsteps - 1].bias;
    const float4 bias = lb ? ld4(lb + ecol) : f4zero();
#pragma unroll
    for (int r = 0; r < 4; ++r) D0[(16 * mt + 4 * g + r) * 32 + (ecol >> 2)] = cb_add(acc[r], bias);
  }
  const float4 gam = ld4(a.ln_g + col), bet = ld4(a.ln_b + col);
  cb_barrier();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
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
    if (l32 == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
  }
}

__global__ __launch_bounds__(CB_THREADS) void mproj_kernel(MProjArgs a, DropCfg drop) { mproj_body(a, drop); }
__global__ __launch_bounds__(CB_THREADS) void mproj_pair_kernel(MProjArgs a0, MProjArgs a1, DropCfg drop) {
  if (blockIdx.y == 0) mproj_body(a0, drop);
  else mproj_body(a1, drop);
}

namespace hual {

// rows per workgroup for problems of R0 (+ R1) rows in one launch: the smallest tile (>= 16) with which ALL workgroups of the
// launch are resident at once - one per CU; a launch with one workgroup more than there are CUs takes two rounds
int mproj_rows(int R0, int R1) {
  int t = 16;
  while (t < MP_ROWS && cdiv(R0, t) + (R1 > 0 ? cdiv(R1, t) : 0) > 256) ++t;
  return t;
}

static int check_mproj(const MProjArgs& a, double& flops, double& bytes) {
  HUAL_REQUIRE(a.nsteps >= 1 && a.nsteps <= MP_MAX && a.R > 0 && a.MT >= 1 && a.MT <= MP_ROWS, "mproj: steps / rows");
  HUAL_REQUIRE(a.s[0].first && !a.s[0].reuse && a.s[a.nsteps - 1].last, "mproj: first step starts a tile, last step closes one");
  for (int k = 0; k < a.nsteps; ++k) {
    const MProjStep& s = a.s[k];
    HUAL_REQUIRE(s.rep >= 1 && (s.rep == 1 || (!s.reuse && s.ktot > 128 * (s.rep - 1) && s.ktot <= 128 * s.rep && (s.ktot % 4) == 0 && !s.A2)), "mproj: repetitions");
    HUAL_REQUIRE(s.reuse || (s.A && (s.rep > 1 || (s.kw >= 4 && s.kw <= 128 && (s.kw % 4) == 0)) && (s.lda % 4) == 0), "mproj: operand");
    HUAL_REQUIRE(!s.reuse || k > 0, "mproj: nothing to reuse");
    HUAL_REQUIRE(s.wimg && s.wrows >= 1, "mproj: weight image");
    HUAL_REQUIRE(!s.last || (a.ln_g && k == a.nsteps - 1 && !s.out) || (s.out && (s.ldo % 4) == 0 && s.ncol >= 4 && s.ncol <= 128 && (s.ncol % 4) == 0),
                 "mproj: closing step needs a destination");
    HUAL_REQUIRE(!s.add || s.add_div >= 1, "mproj: add_div");
    HUAL_REQUIRE(k == 0 || s.first == a.s[k - 1].last, "mproj: a tile starts exactly behind a closed one");
    const double kdeep = s.rep > 1 ? s.ktot : (s.reuse ? a.s[k - 1].kw : s.kw);
    flops += 2.0 * a.R * kdeep * 128.0;
    if (!s.reuse) bytes += (s.a_bf16 ? 2.0 : 4.0) * a.R * kdeep + (s.A2 ? 4.0 * a.R * kdeep : 0.0);
    bytes += 4.0 * 128.0 * (s.rep > 1 ? s.ktot : (s.wrows < 128 ? s.wrows : 128));
    if (s.last && s.out) bytes += 4.0 * a.R * s.ncol;
  }
  if (a.ln_g) {
    HUAL_REQUIRE(a.ln_b && a.y_out && a.mean && a.rstd && (!a.pos || a.Tc >= 1), "mproj: layer-norm outputs");
    bytes += 4.0 * a.R * 128.0 * (a.x_out ? 2.0 : 1.0);
  }
  return 0;
}

int launch_mproj(const MProjArgs* a, int nprob, const DropCfg& drop, hipStream_t s) {
  HUAL_REQUIRE(a && (nprob == 1 || nprob == 2), "mproj: one or two problems");
  double flops = 0.0, bytes = 0.0;
  int blocks = 0;
  for (int i = 0; i < nprob; ++i) {
    int rc = check_mproj(a[i], flops, bytes);
    if (rc) return rc;
    const int nb = cdiv(a[i].R, a[i].MT);
    blocks = nb > blocks ? nb : blocks;
  }
  const size_t lds = (size_t)4 * MP_ROWS * 256 + CB_WBYTES + 2 * MP_ROWS * sizeof(float);
  if (nprob == 2) HUAL_REQUIRE(a[0].MT == a[1].MT, "mproj: problems of one launch share the rows per workgroup");
  if (nprob == 1) {
    HUAL_DYN_LDS(mproj_kernel, 160 * 1024);
    HUAL_LAUNCH(flops, bytes, mproj_kernel, dim3(blocks), dim3(CB_THREADS), lds, s, a[0], drop);
  } else {
    HUAL_DYN_LDS(mproj_pair_kernel, 160 * 1024);
    HUAL_LAUNCH(flops, bytes, mproj_pair_kernel, dim3(blocks, 2), dim3(CB_THREADS), lds, s, a[0], a[1], drop);
  }
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
