// Attention kernels (see attn.h), head size 16, on v_mfma_f32_16x16x32_bf16 with split fp32 operands (bf16x3.h:
// x = hi + lo, a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi, fp32 accumulate).
//
// One workgroup (4 waves) = one (job, clip, head).  The head's K / V (forward) or Q / dO / K (backward) panels are split
// once into bf16 planes in LDS; everything between the products stays in registers:
//
//   forward  - a wave owns 16 queries and ALL keys.  It computes S^T = K.Q^T per 16-key tile (K = 32 slots of the MFMA =
//              [hi | lo] of the 16 head dims, so hi.hi + lo.hi is ONE instruction and hi.lo the second), so lane
//              (j = lane & 15, g = lane >> 4) register r holds S[query j][key 16 kt + 4 g + r]: a query's row is spread over
//              4 lanes x 4 nkt registers (row max / sum = in-lane + two xor shuffles), and the same registers of two key
//              tiles are, unchanged, the A operand of P.V over 32 keys (B = transposing reads of the V panel).
//   backward - ONE pass over the scores for dQ, dK and dV.  A wave owns 32 keys and sweeps the queries in the mirrored
//              orientation (S = Q.K^T, lane (j = key, g) register r = query 4 g + r), where P^T and dS^T are directly the A
//              operands of dV += P^T.dO and dK += dS^T.Q (contraction over queries).  dQ += dS.K contracts over keys: the
//              dS tile goes through a wave-private LDS transpose and the per-wave partial products are summed at the end
//              from one fp32 LDS slot per wave.  The forward leaves (row max in the log2 domain, 1 / row sum) and the keep bits of
//              the dropout; delta = dO . O.
//
// Dropout on the probabilities (layers.py:86,91; modules.py:114): ONE Philox4x32-10 call yields 16 decisions from its 16
// bytes (keep iff byte < t8, t8 = round(keep_prob * 256), kept values scaled by 256 / t8 - unbiased at the 8-bit keep
// probability); the call with counter c0 = g + 4 (kt >> 2) covers keys 16 kt + 4 g + r for the four tiles kt of a group:
// word kt & 3, byte r.  oracle/philox.py `mask8` is the same draw.  (The other dropout sites keep one decision per word.)
#include "attn.h"
#include "bf16x3.h"
#include "philox.h"
#include "prof.h"
#include <string.h>
#include <stdlib.h>

using namespace hual;

namespace hual {
void attn_job_init(AttnJob& j) {
  ::memset((void*)&j, 0, sizeof(j));
  j.drop_site = -1;
  j.dmask = nullptr;
}
int attn_ldm(int Tk) { return 4 * ((cdiv(Tk, 16) + 3) & ~3); }
}  // namespace hual

#define ATT_LOG2E 1.4426950408889634f
#define ATT_C1 (0.25f * ATT_LOG2E)              // 1/sqrt(head_size = 16) (layers.py:82), scores kept in the log2 domain
#define ATT_NEGL (HUAL_MASK_VALUE * ATT_LOG2E)  // the additive mask value (ops.py:89 via layers.py:84) in the log2 domain

__device__ __forceinline__ f32x4 mfma_bf(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_bf8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ bf16x8 as_bf8(uint2 a, uint2 b) { return as_bf8(make_uint4(a.x, a.y, b.x, b.y)); }
// eight floats -> packed high parts and residuals
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
  bf16_split_pair(a.x, a.y, hi.x, lo.x);
  bf16_split_pair(a.z, a.w, hi.y, lo.y);
  bf16_split_pair(b.x, b.y, hi.z, lo.z);
  bf16_split_pair(b.z, b.w, hi.w, lo.w);
}

// XCD-aware block order (cdna_hip_programming.md T1, bijective form).  The dispatcher deals consecutive linear block ids
// round-robin over the 8 XCDs, each with a private L2; all blocks of one clip (jobs x heads) read that clip's rows, so the
// remap hands each XCD a contiguous run of logical ids, i.e. whole clips.
__device__ __forceinline__ int xcd_logical_id() {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// 8-bit dropout decisions: threshold / scale from the 32-bit threshold of DropCfg (same arithmetic as oracle/philox.py)
__device__ __forceinline__ uint32_t drop_t8(const DropCfg& d) {
  uint32_t t = (uint32_t)(((uint64_t)d.thresh + (1ull << 23)) >> 24);
  return t < 1u ? 1u : (t > 256u ? 256u : t);
}

// ---- LDS panels -----------------------------------------------------------------------------------------
// A panel holds the 16 head dims of up to 256 rows, split: row r = [16 x bf16 high parts | 16 x bf16 residuals] (64 bytes).
// It serves every operand shape of the kernels:
//   * rows as the M / N index of a product whose contraction runs over the head dims (S = Q.K^T, dP = dO.V^T): one
//     16-byte read per lane, slots [hi | lo] (panel_a) or [hi | hi], [lo | 0] (panel_b1 / panel_b2);
//   * rows as the CONTRACTION index (P.V, dS^T.Q, P^T.dO, dS.K): ds_read_b64_tr_b16 delivers a 4-row x 16-column block
//     column-major, i.e. lane (col = lane & 15, g) receives rows 4 g .. 4 g + 3 of its column (panel_tr).
__device__ __forceinline__ void panel_store(char* dst, int row, int c4, const float4& v) {
  uint2 h, l;
  bf16_split4(v, h, l);
  *reinterpret_cast<uint2*>(dst + row * 64 + 8 * c4) = h;
  *reinterpret_cast<uint2*>(dst + row * 64 + 32 + 8 * c4) = l;
}
__device__ __forceinline__ bf16x8 panel_a(const char* panel, int row, int g) {
  return *reinterpret_cast<const bf16x8*>(panel + row * 64 + 16 * g);
}
__device__ __forceinline__ bf16x8 panel_b1(const char* panel, int row, int g) {     // [hi | hi]
  return *reinterpret_cast<const bf16x8*>(panel + row * 64 + 16 * (g & 1));
}
__device__ __forceinline__ bf16x8 panel_b2(const char* panel, int row, int g) {     // [lo | 0]
  const uint4 v = *reinterpret_cast<const uint4*>(panel + row * 64 + 32 + 16 * (g & 1));
  return as_bf8(g >= 2 ? make_uint4(0u, 0u, 0u, 0u) : v);
}
// B operand over the 32 contraction rows row0 .. row0 + 31 for column lane & 15: slots e < 4 <-> rows row0 + 4 g + e,
// e >= 4 <-> rows row0 + 16 + 4 g + (e - 4) (the order in which the accumulators of two 16-row tiles hold them).
// plane = 0: high parts, 32: residuals.  EXEC must be all ones.
__device__ __forceinline__ bf16x8 panel_tr(const char* panel, int plane, int row0, int lane) {
  const int off = (row0 + 4 * (lane >> 4) + ((lane & 15) >> 2)) * 64 + plane + 8 * (lane & 3);
  return join_tr(lds_read_tr16(panel, off), lds_read_tr16(panel, off + 16 * 64));
}

// ======================================================================================================
// forward
// ======================================================================================================
// NKT = key tiles of 16 staged and processed (compile time: the tile loops carry no conditions; tiles beyond Tk hold zero
// keys with an additive term of -inf, i.e. probability 0).  One Philox call per lane covers a group of 4 tiles.
// DROP: dropout on the probabilities (compile time: the mask code carries no run-time branches)
template <int NKT, bool DROP, int NT>      // NT threads per workgroup (NT / 64 query tiles in flight)
__device__ __forceinline__ void attn_fwd_body(const AttnJob& job, int b, int h, char* lds, const DropCfg& drop) {
  constexpr int Tkp = 16 * NKT, NKQ = (NKT + 3) / 4, NST = (Tkp * 4 + NT - 1) / NT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Tq = job.Tq, Tk = job.Tk;
  const int qbase = job.qrow0 + b * Tq, kbase = job.krow0 + b * Tk;
  char* Kp = lds;                                   // [Tkp][64]
  char* Vp = Kp + Tkp * 64;                         // [Tkp][64]
  float* Bias = reinterpret_cast<float*>(Vp + Tkp * 64);      // [2][Tkp]: additive term of a key for padded / valid queries
  const int nqt = (Tq + 15) >> 4;
  // the dropout stream's key / offset live in device memory: read at the top (behind the barrier the load was a round trip of its own)
  constexpr bool dodrop = DROP;
  uint32_t k0 = 0, k1 = 0, off = 0;
  if (dodrop) {      // (through the constant address space: scalar loads, no vector-memory wait in front of the staging loads)
    const __attribute__((address_space(4))) uint32_t* sp = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)drop.state;
    k0 = sp[0]; k1 = sp[1]; off = sp[2];
  }
  // the wave's first query tile is requested before the staging so that it arrives under it, every further one while the tile
  // in front of it is computed (two register sets); the query mask is read in the staging phase, from LDS afterwards
  float4 qv0 = f4zero(), qv1 = f4zero(), nq0 = f4zero(), nq1 = f4zero();
  auto load_q = [&](int qt, float4& d0, float4& d1) {
    const int qrow = qbase + min(16 * qt + j, Tq - 1);
    const float* qp = job.Q + (size_t)qrow * job.ldq + 16 * h + 8 * (g & 1);
    d0 = ld4(qp); d1 = ld4(qp + 4);
  };
  if (wave < nqt) load_q(wave, qv0, qv1);
  float qmk = 0.f;                                   // mask of query wave * 16 + j (the wave's first tile): requested behind the panels
  {
    // all loads of the block's panels first, then the splits and LDS stores
    const float* Kg = job.K + (size_t)kbase * job.ldkv + 16 * h;
    const float* Vg = job.V + (size_t)kbase * job.ldkv + 16 * h;
    float4 kv[NST], vv[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int idx = threadIdx.x + NT * it, row = min(idx >> 2, Tk - 1), c4 = idx & 3;
      kv[it] = ld4(Kg + (size_t)row * job.ldkv + 4 * c4);
      vv[it] = ld4(Vg + (size_t)row * job.ldkv + 4 * c4);
    }
    // (unconditional on a clamped index: a load behind a lane-dependent branch is waited for inside that branch)
    const float km = job.kmask[kbase + min((int)threadIdx.x, Tk - 1)];
    qmk = job.qmask[qbase + min(16 * min(wave, nqt - 1) + j, Tq - 1)];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int idx = threadIdx.x + NT * it, row = idx >> 2, c4 = idx & 3;
      if (row < Tkp) {
        panel_store(Kp, row, c4, row < Tk ? kv[it] : f4zero());
        panel_store(Vp, row, c4, row < Tk ? vv[it] : f4zero());
      }
    }
    if (threadIdx.x < Tkp) {
      const bool in = (int)threadIdx.x < Tk;
      Bias[threadIdx.x] = in ? ATT_NEGL : -INFINITY;                                  // layers.py:84: (1 - mq mk) * -1e30, mq = 0
      Bias[Tkp + threadIdx.x] = in ? (km != 0.f ? 0.f : ATT_NEGL) : -INFINITY;
    }
  }
  __syncthreads();
  const uint32_t t8 = drop_t8(drop);
  const float scale8 = 256.0f / (float)t8;
  for (int qt = wave; qt < nqt; qt += NT / 64) {
    const int q0 = qt * 16;
    const bool qok = q0 + j < Tq;
    const int qrow = qbase + min(q0 + j, Tq - 1);
    if (qt != wave) { qv0 = nq0; qv1 = nq1; }
    const bool more_q = qt + NT / 64 < nqt;
    const int qnext = more_q ? qt + NT / 64 : qt;     // (the last tile asks for itself again: no branch around the loads)
    load_q(qnext, nq0, nq1);
    const float mqn = job.qmask[qbase + min(16 * qnext + j, Tq - 1)];
    const float mq = qmk;
    qmk = mqn;
    // B operands: [Q_hi | Q_hi] and [Q_lo | 0] over the 32 slots (lane g covers head dims 8 (g & 1) .. + 7)
    uint4 qh, ql;
    split8(qv0, qv1, qh, ql);
    if (g >= 2) ql = make_uint4(0u, 0u, 0u, 0u);
    const bf16x8 B1 = as_bf8(qh), B2 = as_bf8(ql);
    const float* bias = Bias + (mq != 0.f ? Tkp : 0);
    f32x4 s[NKT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const bf16x8 a = panel_a(Kp, 16 * kt + j, g);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = mfma_bf(a, B1, acc);
      acc = mfma_bf(a, B2, acc);
      const float4 b4 = *reinterpret_cast<const float4*>(bias + 16 * kt + 4 * g);
      acc[0] = fmaf(acc[0], ATT_C1, b4.x); acc[1] = fmaf(acc[1], ATT_C1, b4.y);
      acc[2] = fmaf(acc[2], ATT_C1, b4.z); acc[3] = fmaf(acc[3], ATT_C1, b4.w);
      mx = fmaxf(fmaxf(mx, fmaxf(acc[0], acc[1])), fmaxf(acc[2], acc[3]));
      s[kt] = acc;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx);
        s[kt][r] = e;
        sum += e;
      }
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    const int ql_ = b * Tq + q0 + j;                  // job-local query index
    if (job.stats && g == 0 && qok) {
      job.stats[ql_ * 8 + h] = mx;
      job.stats[job.B * Tq * 8 + ql_ * 8 + h] = inv;
    }
    const uint32_t drow = (job.drop_row0 + (uint32_t)qrow) * 8u + (uint32_t)h;
    uint8_t* mrow = (job.dmask && dodrop && qok) ? job.dmask + ((size_t)ql_ * 8 + h) * job.ldm + g * (job.ldm >> 2) : nullptr;
    const float keepv = dodrop ? inv * scale8 : inv;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq) {
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      uint32_t bits = 0u;
      if (dodrop) {
        const uint4_ rnd = philox4x32_10((uint32_t)(g + 4 * kq), drow, (uint32_t)job.drop_site, off, k0, k1);
        w[0] = rnd.x; w[1] = rnd.y; w[2] = rnd.z; w[3] = rnd.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kt = 4 * kq + i;
        if (kt < NKT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool keep = !dodrop || ((w[i] >> (8 * r)) & 0xffu) < t8;
            s[kt][r] *= keep ? keepv : 0.f;
            bits |= keep ? (1u << (8 * i + r)) : 0u;
          }
        }
      }
      if (mrow) *reinterpret_cast<uint32_t*>(mrow + 4 * kq) = bits;
#pragma unroll
      for (int ip = 0; ip < 2; ++ip) {
        const int kp = 2 * kq + ip;                 // key tiles 2 kp, 2 kp + 1
        if (2 * kp < NKT) {
          uint4 ph, pl;
          split8(make_float4(s[2 * kp][0], s[2 * kp][1], s[2 * kp][2], s[2 * kp][3]),
                 make_float4(s[2 * kp + 1][0], s[2 * kp + 1][1], s[2 * kp + 1][2], s[2 * kp + 1][3]), ph, pl);
          const bf16x8 vh = panel_tr(Vp, 0, 32 * kp, lane), vl = panel_tr(Vp, 32, 32 * kp, lane);
          o = mfma_bf(as_bf8(ph), vh, o);
          o = mfma_bf(as_bf8(ph), vl, o);
          o = mfma_bf(as_bf8(pl), vh, o);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + 4 * g + r;
      if (q < Tq) job.O[(size_t)(qbase + q) * job.ldo + 16 * h + j] = o[r];
    }
  }
}

// MAXNKT: the largest tile count among the jobs of the launch (the kernel's register budget is that of its largest body)
// (limiting the kernel to 128 registers - 4 waves per SIMD - spills 17 of them in the 8-tile body and is SLOWER, with 4- or
//  8-wave workgroups alike: 13.4 vs 10.7 us for the 128 x 128 job.  The kernel is VALU-issue bound: ~25 vector instructions per
//  score - 10 for mask / max / exp / sum, 7.5 Philox, 3 keep decision + keep bits, 3.5 operand split - not latency bound.)
template <int MAXNKT, int NT>
__global__ __launch_bounds__(NT) void attn_fwd_kernel(AttnBatch batch, int njobs, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int lid = xcd_logical_id();
  const int h = lid & 7; lid >>= 3;
  const AttnJob& job = batch.j[lid % njobs];       // clip-major: an XCD gets whole clips with ALL their jobs (balanced)
  const int b = lid / njobs;
  if (b >= job.B) return;   // block-uniform
  const int nkt = (job.Tk + 15) >> 4;
  if (job.drop_site >= 0 && drop.enabled) {
    if (nkt <= 2) attn_fwd_body<2, true, NT>(job, b, h, lds, drop);
    else if (MAXNKT >= 4 && nkt <= 4) attn_fwd_body<4, true, NT>(job, b, h, lds, drop);
    else if (MAXNKT >= 8 && nkt <= 8) attn_fwd_body<8, true, NT>(job, b, h, lds, drop);
    else if (MAXNKT >= 16) attn_fwd_body<16, true, NT>(job, b, h, lds, drop);
  } else {
    if (nkt <= 2) attn_fwd_body<2, false, NT>(job, b, h, lds, drop);
    else if (MAXNKT >= 4 && nkt <= 4) attn_fwd_body<4, false, NT>(job, b, h, lds, drop);
    else if (MAXNKT >= 8 && nkt <= 8) attn_fwd_body<8, false, NT>(job, b, h, lds, drop);
    else if (MAXNKT >= 16) attn_fwd_body<16, false, NT>(job, b, h, lds, drop);
  }
}

// ======================================================================================================
// backward
// ======================================================================================================
// LDS map of one (job, clip, head), Tqp = Tq rounded up to 32 queries, Tkp likewise:
//   Qp, Dp   [Tqp][64]       split Q / dO panels
//   Kp       [Tkp][64]       split K panel (the V rows of a wave's keys go from HBM straight into its B operands)
//   St       [6][Tqp]        row max (log2 domain), 1 / row sum, delta, additive term for masked / valid / padding keys
//   Mk       [Tqp][ldm]      keep bytes of the head (forward layout: byte g * (ldm / 4) + kt, bit r)
//   dQw      [4 waves][Tqp][16]  fp32 dQ partial products, one slot per wave (plain stores: float atomics on LDS retire at
//                            about one lane per clock and cost more than the rest of the kernel)
//   Xs       [4 waves][32][20]   transposition scratch
struct BwdLds { int qp, dp, kp, st, mk, dqw, xs, total; };
__host__ __device__ inline BwdLds bwd_lds(int Tq, int Tk, int ldm) {
  const int Tqp = (Tq + 31) & ~31, Tkp = (Tk + 31) & ~31;
  BwdLds l;
  int o = 0;
  l.qp = o; o += Tqp * 64;
  l.dp = o; o += Tqp * 64;
  l.kp = o; o += Tkp * 64;
  l.st = o; o += 6 * Tqp * 4;
  l.mk = o; o += Tqp * ldm;
  l.dqw = o; o += 4 * Tqp * 64;
  l.xs = o; o += 4 * 32 * 20 * 4;
  l.total = o;
  return l;
}

template <bool DROP>
__device__ __forceinline__ void attn_bwd_body(const AttnJob& job, int b, int h, char* lds, const DropCfg& drop) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Tq = job.Tq, Tk = job.Tk;
  const int Tqp = (Tq + 31) & ~31, Tkp = (Tk + 31) & ~31;
  const int qbase = job.qrow0 + b * Tq, kbase = job.krow0 + b * Tk;
  constexpr bool dodrop = DROP;
  const int ldm = dodrop ? job.ldm : 0;
  const BwdLds L = bwd_lds(Tq, Tk, ldm);
  char* Qp = lds + L.qp; char* Dp = lds + L.dp; char* Kp = lds + L.kp;
  float* St = reinterpret_cast<float*>(lds + L.st);
  uint8_t* Mk = reinterpret_cast<uint8_t*>(lds + L.mk);
  float* dQw = reinterpret_cast<float*>(lds + L.dqw);
  float* Xs = reinterpret_cast<float*>(lds + L.xs) + wave * 32 * 20;
  const float* Qg = job.Q + (size_t)qbase * job.ldq + 16 * h;
  const float* Dg = job.dO + (size_t)qbase * job.lddo + 16 * h;
  const float* Og = job.O + (size_t)qbase * job.ldo + 16 * h;
  const float* Kg = job.K + (size_t)kbase * job.ldkv + 16 * h;
  const float* Vg = job.V + (size_t)kbase * job.ldkv + 16 * h;
  const int nkp = Tkp >> 5, nqp = Tqp >> 5;
  // V rows of the wave's first 32 keys: requested before the staging so that they arrive under it
  float4 vreg[2][2];
  float kmv[2];
  auto load_v = [&](int kp) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int krow = min(32 * kp + 16 * t + j, Tk - 1);
      const float* vptr = Vg + (size_t)krow * job.ldkv + 8 * (g & 1);
      vreg[t][0] = ld4(vptr); vreg[t][1] = ld4(vptr + 4);
      kmv[t] = job.kmask[kbase + krow];
    }
  };
  // ONE block of <= 32 keys and at least four 32-query pairs (the video -> query jobs): the waves split the QUERIES instead of the
  // keys - all four work on key block 0, wave w takes the query pairs w, w + 4, ..; dQ rows are then disjoint (one slot), and the
  // partial dK / dV of the waves are summed through the three free slots
  const bool qsplit = nkp == 1 && nqp >= 4;
  if (wave < nkp || qsplit) load_v(qsplit ? 0 : wave);
  // ---- staging: per pass every thread requests its piece of all three panels (+ O for delta = dO . O) before any split
  const int stat_n = job.B * Tq * 8;
  const int npass = (max(Tqp, Tkp) * 4 + 255) >> 8;
  for (int it = 0; it < npass; ++it) {
    const int idx = threadIdx.x + 256 * it, row = idx >> 2, c4 = idx & 3;
    const int qr = min(row, Tq - 1), kr = min(row, Tk - 1);
    const float4 qv = ld4(Qg + (size_t)qr * job.ldq + 4 * c4);
    const float4 dv = ld4(Dg + (size_t)qr * job.lddo + 4 * c4);
    const float4 ov = ld4(Og + (size_t)qr * job.ldo + 4 * c4);
    const float4 kv = ld4(Kg + (size_t)kr * job.ldkv + 4 * c4);
    float smx = 0.f, sinv = 0.f, qm = 0.f;
    const bool qok = row < Tq;
    if (c4 == 0 && qok) {
      const int si = (b * Tq + row) * 8 + h;
      smx = job.stats[si];
      sinv = job.stats[stat_n + si];
      qm = job.qmask[qbase + row];
    }
    if (row < Tqp) {
      panel_store(Qp, row, c4, qok ? qv : f4zero());
      panel_store(Dp, row, c4, qok ? dv : f4zero());
    }
    if (row < Tkp) panel_store(Kp, row, c4, row < Tk ? kv : f4zero());
    float part = qok ? (dv.x * ov.x + dv.y * ov.y) + (dv.z * ov.z + dv.w * ov.w) : 0.f;
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    if (c4 == 0 && row < Tqp) {
      St[row] = smx;
      St[Tqp + row] = sinv;                                                      // 1 / row sum = 0 for padding queries -> p = 0
      St[2 * Tqp + row] = part;
      St[3 * Tqp + row] = ATT_NEGL;                                              // key masked
      St[4 * Tqp + row] = (qok && qm != 0.f) ? 0.f : ATT_NEGL;                   // key valid: (1 - mq) * -1e30
      St[5 * Tqp + row] = -INFINITY;                                             // key beyond Tk (tile padding)
    }
  }
  if (dodrop) {
    const int wpr = ldm >> 2;                          // words per row
    const uint32_t* src = reinterpret_cast<const uint32_t*>(job.dmask + ((size_t)(b * Tq) * 8 + h) * ldm);
    for (int idx = threadIdx.x; idx < Tq * wpr; idx += 256) {
      const int q = idx / wpr, w = idx - q * wpr;
      reinterpret_cast<uint32_t*>(Mk)[idx] = src[(size_t)q * 8 * wpr + w];
    }
  }
  __syncthreads();
  const float scale8 = 256.0f / (float)drop_t8(drop);
  float* slot = dQw + (qsplit ? 0 : wave) * Tqp * 16;
  bool first = true;
  const int qp0 = qsplit ? wave : 0, qpstep = qsplit ? 4 : 1;
  for (int kp = qsplit ? 0 : wave; kp < nkp; kp += 4) {
    // B operands of this wave's 32 keys: [K_hi | K_hi], [K_lo | 0] and the same of V, per 16-key tile t
    bf16x8 Kb1[2], Kb2[2], Vb1[2], Vb2[2];
    int bsel[2];
    if (kp != wave && !qsplit) load_v(kp);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = 32 * kp + 16 * t + j;
      Kb1[t] = panel_b1(Kp, key, g); Kb2[t] = panel_b2(Kp, key, g);
      uint4 h4, l4;
      split8(key < Tk ? vreg[t][0] : f4zero(), key < Tk ? vreg[t][1] : f4zero(), h4, l4);
      if (g >= 2) l4 = make_uint4(0u, 0u, 0u, 0u);
      Vb1[t] = as_bf8(h4); Vb2[t] = as_bf8(l4);
      bsel[t] = key < Tk ? (kmv[t] != 0.f ? 4 : 3) : 5;
    }
    const bf16x8 kh = panel_tr(Kp, 0, 32 * kp, lane), kl = panel_tr(Kp, 32, 32 * kp, lane);      // B operand of dQ
    f32x4 dk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int qp = qp0; qp < nqp; qp += qpstep) {
      float pd[2][2][4], ds[2][2][4];          // [t][u][r]
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q0 = 32 * qp + 16 * u;
        const bf16x8 aq = panel_a(Qp, q0 + j, g), ad = panel_a(Dp, q0 + j, g);
        const float4 m4 = *reinterpret_cast<const float4*>(St + q0 + 4 * g);
        const float4 i4 = *reinterpret_cast<const float4*>(St + Tqp + q0 + 4 * g);
        const float4 d4 = *reinterpret_cast<const float4*>(St + 2 * Tqp + q0 + 4 * g);
        const float mxv[4] = {m4.x, m4.y, m4.z, m4.w}, inv[4] = {i4.x, i4.y, i4.z, i4.w}, dlv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = mfma_bf(aq, Kb1[t], s);
          s = mfma_bf(aq, Kb2[t], s);          // lane: S[query q0 + 4 g + r][key 32 kp + 16 t + j]
          dp = mfma_bf(ad, Vb1[t], dp);
          dp = mfma_bf(ad, Vb2[t], dp);
          const float4 b4 = *reinterpret_cast<const float4*>(St + bsel[t] * Tqp + q0 + 4 * g);
          const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
          uint32_t mbits = 0xfu;
          if (dodrop) {
            // keep bits of queries q0 + 4 g + r for this lane's key: byte (j >> 2) * (ldm / 4) + kt of each query row, bit j & 3
            const uint8_t* mp = Mk + (q0 + 4 * g) * ldm + (j >> 2) * (ldm >> 2) + 2 * kp + t;
            mbits = ((mp[0] >> (j & 3)) & 1u) | (((mp[ldm] >> (j & 3)) & 1u) << 1) | (((mp[2 * ldm] >> (j & 3)) & 1u) << 2) |
                    (((mp[3 * ldm] >> (j & 3)) & 1u) << 3);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = fmaf(s[r], ATT_C1, bv[r]);
            const float p = __builtin_amdgcn_exp2f(v - mxv[r]) * inv[r];
            const float m = dodrop ? (((mbits >> r) & 1u) ? scale8 : 0.f) : 1.0f;
            pd[t][u][r] = p * m;                                  // dropped probability: A operand of dV
            ds[t][u][r] = p * fmaf(dp[r], m, -dlv[r]);            // dS / 0.25: A operand of dK, dQ
          }
        }
      }
      // dV += Pd^T . dO, dK += dS^T . Q over the 32 queries of the pair
      const bf16x8 doh = panel_tr(Dp, 0, 32 * qp, lane), dol = panel_tr(Dp, 32, 32 * qp, lane);
      const bf16x8 qh = panel_tr(Qp, 0, 32 * qp, lane), ql = panel_tr(Qp, 32, 32 * qp, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint4 ah, al;
        split8(make_float4(pd[t][0][0], pd[t][0][1], pd[t][0][2], pd[t][0][3]),
               make_float4(pd[t][1][0], pd[t][1][1], pd[t][1][2], pd[t][1][3]), ah, al);
        dv[t] = mfma_bf(as_bf8(ah), doh, dv[t]);
        dv[t] = mfma_bf(as_bf8(ah), dol, dv[t]);
        dv[t] = mfma_bf(as_bf8(al), doh, dv[t]);
        split8(make_float4(ds[t][0][0], ds[t][0][1], ds[t][0][2], ds[t][0][3]),
               make_float4(ds[t][1][0], ds[t][1][1], ds[t][1][2], ds[t][1][3]), ah, al);
        dk[t] = mfma_bf(as_bf8(ah), qh, dk[t]);
        dk[t] = mfma_bf(as_bf8(ah), ql, dk[t]);
        dk[t] = mfma_bf(as_bf8(al), qh, dk[t]);
      }
      // dQ partial of the pair's two query tiles: transpose dS through the wave's scratch (key-major rows of 16 queries)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          *reinterpret_cast<float4*>(Xs + (16 * t + j) * 20 + 4 * g) = make_float4(ds[t][u][0], ds[t][u][1], ds[t][u][2], ds[t][u][3]);
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = Xs[(16 * (e >> 2) + 4 * g + (e & 3)) * 20 + j];      // dS[query j][key 16 t + 4 g + r]
        uint4 ah, al;
        split8(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), ah, al);
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
        dq = mfma_bf(as_bf8(ah), kh, dq);
        dq = mfma_bf(as_bf8(ah), kl, dq);
        dq = mfma_bf(as_bf8(al), kh, dq);      // lane: dQ[query 4 g + r][head dim j]
        float* dst = slot + (32 * qp + 16 * u + 4 * g) * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[16 * r] = first ? dq[r] : dst[16 * r] + dq[r];
      }
    }
    first = false;
    if (qsplit) {      // partial dK / dV of this wave's queries -> slots 1 .. 3 as [wave][dk | dv][t][r][lane]; summed below
      float* red = dQw + Tqp * 16 + wave * 1024;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          red[(t * 4 + r) * 64 + lane] = dk[t][r];
          red[512 + (t * 4 + r) * 64 + lane] = dv[t][r];
        }
      continue;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = 32 * kp + 16 * t + 4 * g + r;
        if (kk < Tk) {
          job.dK[(size_t)(kbase + kk) * job.lddkv + 16 * h + j] = dk[t][r] * 0.25f;
          job.dV[(size_t)(kbase + kk) * job.lddkv + 16 * h + j] = dv[t][r];
        }
      }
  }
  __syncthreads();
  if (qsplit) {      // dK / dV of key block 0: the four waves' partials, in wave order
    const float* red = dQw + Tqp * 16;
    for (int idx = threadIdx.x; idx < 1024; idx += 256) {
      const int which = idx >> 9, e = (idx >> 6) & 7, ln = idx & 63;      // e = t * 4 + r, lane ln = (j, g)
      const float v = (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
      const int kk = 16 * (e >> 2) + 4 * (ln >> 4) + (e & 3);
      if (kk < Tk) {
        float* dst = which ? job.dV : job.dK;
        dst[(size_t)(kbase + kk) * job.lddkv + 16 * h + (ln & 15)] = which ? v : v * 0.25f;
      }
    }
  }
  // dQ = 0.25 * sum of the slots of the waves that had keys
  const int nw = qsplit ? 1 : min(4, nkp);
  for (int idx = threadIdx.x; idx < Tq * 4; idx += 256) {
    const int q = idx >> 2, c4 = idx & 3;
    float4 v = reinterpret_cast<const float4*>(dQw)[idx];
    for (int w = 1; w < nw; ++w) {
      const float4 a = reinterpret_cast<const float4*>(dQw + w * Tqp * 16)[idx];
      v = make_float4(v.x + a.x, v.y + a.y, v.z + a.z, v.w + a.w);
    }
    st4(job.dQ + (size_t)(qbase + q) * job.lddq + 16 * h + 4 * c4, make_float4(v.x * 0.25f, v.y * 0.25f, v.z * 0.25f, v.w * 0.25f));
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnBatch batch, int njobs, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int lid = xcd_logical_id();
  const int h = lid & 7; lid >>= 3;
  const AttnJob& job = batch.j[lid % njobs];
  const int b = lid / njobs;
  if (b >= job.B) return;   // block-uniform
  if (job.drop_site >= 0 && drop.enabled) attn_bwd_body<true>(job, b, h, lds, drop);
  else attn_bwd_body<false>(job, b, h, lds, drop);
}

namespace hual {

static int check_jobs(const AttnJob* jobs, int n, bool bwd, const DropCfg& drop, int& maxTq, int& maxTk, int& maxB) {
  HUAL_REQUIRE(n >= 1 && n <= HUAL_MAX_ATTN_JOBS, "attn: job count");
  maxTq = maxTk = maxB = 0;
  for (int i = 0; i < n; ++i) {
    const AttnJob& j = jobs[i];
    HUAL_REQUIRE(j.Q && j.K && j.V && j.qmask && j.kmask, "attn: null operand");
    HUAL_REQUIRE(j.B > 0 && j.Tq > 0 && j.Tk > 0 && j.Tk <= 256, "attn: need 0 < Tk <= 256, Tq > 0");
    HUAL_REQUIRE((j.ldq % 4) == 0 && (j.ldkv % 4) == 0, "attn: leading dims must be multiples of 4");
    HUAL_REQUIRE(!j.dmask || (j.ldm >= attn_ldm(j.Tk) && (j.ldm % 16) == 0 && (reinterpret_cast<uintptr_t>(j.dmask) & 3) == 0),
                 "attn: dropout keep-byte rows need ldm >= attn_ldm(Tk), multiple of 16");
    if (bwd) {
      HUAL_REQUIRE(j.dO && j.dQ && j.dK && j.dV && j.O && j.stats && (j.lddo % 4) == 0 && (j.ldo % 4) == 0 && (j.lddq % 4) == 0,
                   "attn bwd: needs dO, dQ, dK, dV, the forward output O and the forward softmax statistics");
      HUAL_REQUIRE(j.Tq <= 256, "attn bwd: Tq <= 256");
      HUAL_REQUIRE(!(j.drop_site >= 0 && drop.enabled) || j.dmask, "attn bwd: dropout needs the keep bytes of the forward");
    } else {
      HUAL_REQUIRE(j.O != nullptr, "attn fwd: null output");
    }
    maxTq = j.Tq > maxTq ? j.Tq : maxTq;
    maxTk = j.Tk > maxTk ? j.Tk : maxTk;
    maxB = j.B > maxB ? j.B : maxB;
  }
  return 0;
}

int launch_attn_fwd(const AttnJob* jobs, int n, const DropCfg& drop, hipStream_t s) {
  int maxTq, maxTk, maxB;
  int rc = check_jobs(jobs, n, false, drop, maxTq, maxTk, maxB);
  if (rc) return rc;
  AttnBatch b;
  ::memset((void*)&b, 0, sizeof(b));
  for (int i = 0; i < n; ++i) b.j[i] = jobs[i];
  dim3 grid(maxB * 8 * n), block(256);
  double flops = 0.0;
  for (int i = 0; i < n; ++i) flops += 4.0 * jobs[i].B * 8.0 * jobs[i].Tq * jobs[i].Tk * 16.0;   // QK^T + PV
  const int nkt = cdiv(maxTk, 16);
  const int Tkp = 16 * (nkt <= 2 ? 2 : nkt <= 4 ? 4 : nkt <= 8 ? 8 : 16);
  const size_t lds = (size_t)2 * Tkp * 64 + 2 * Tkp * sizeof(float);
  // (8 waves per workgroup - one 8-wave workgroup per CU instead of two 4-wave ones - measured equal)
  if (nkt <= 2) HUAL_LAUNCH(flops, 0.0, (attn_fwd_kernel<2, 256>), grid, block, lds, s, b, n, drop);
  else if (nkt <= 4) HUAL_LAUNCH(flops, 0.0, (attn_fwd_kernel<4, 256>), grid, block, lds, s, b, n, drop);
  else if (nkt <= 8) HUAL_LAUNCH(flops, 0.0, (attn_fwd_kernel<8, 256>), grid, block, lds, s, b, n, drop);
  else HUAL_LAUNCH(flops, 0.0, (attn_fwd_kernel<16, 256>), grid, block, lds, s, b, n, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_attn_bwd(const AttnJob* jobs, int n, const DropCfg& drop, hipStream_t s) {
  int maxTq, maxTk, maxB;
  int rc = check_jobs(jobs, n, true, drop, maxTq, maxTk, maxB);
  if (rc) return rc;
  AttnBatch b;
  ::memset((void*)&b, 0, sizeof(b));
  size_t lds = 0;
  double flops = 0.0;
  for (int i = 0; i < n; ++i) {
    b.j[i] = jobs[i];
    const bool dd = jobs[i].drop_site >= 0 && drop.enabled;
    const size_t need = (size_t)bwd_lds(jobs[i].Tq, jobs[i].Tk, dd ? jobs[i].ldm : 0).total;
    lds = need > lds ? need : lds;
    flops += 2.0 * jobs[i].B * 8.0 * jobs[i].Tq * jobs[i].Tk * 16.0;
  }
  HUAL_REQUIRE(lds <= 160 * 1024, "attn bwd: LDS footprint");
  HUAL_DYN_LDS(attn_bwd_kernel, 160 * 1024);
  dim3 grid(maxB * 8 * n), block(256);
  // algorithmic work of the backward: FOUR products (dP = dO.V^T, dV = P^T.dO, dK = dS^T.Q, dQ = dS.K) = 8.B.H.Tq.Tk.16; the
  // recomputation of S = Q.K^T is the kernel's choice (it saves storing P) and is not counted
  HUAL_LAUNCH(4.0 * flops, 0.0, attn_bwd_kernel, grid, block, lds, s, b, n, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
