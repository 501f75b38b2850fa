// Attention kernels (see attn.h), head size 16, on v_mfma_f32_16x16x32_f16 with split fp32 operands (bf16x3.h, "f16x3":
// x s = hi + lo with fp16 hi / lo and a power-of-two scale s, a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi, fp32 accumulate - 22-bit
// operands; rounds 2-4 used bf16 pairs, 16 bits: see the header of the backward section for the scales and what they cost).
//
// One workgroup (4 waves) = one (job, clip, head).  The head's K / V (forward) or Q / dO / K (backward) panels are split
// once into fp16 hi / lo planes in LDS; everything between the products stays in registers:
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
// Dropout on the probabilities (layers.py:86,91; modules.py:114): 16-bit decisions, EIGHT per Philox4x32-7 call (keep iff half < t16,
// t16 = round(keep_prob * 65536); kept values scaled by exactly 1 / (1 - rate), as tf.nn.dropout does); key k = 16 kt + 4 g + r of RNG
// row (query row, head) takes the call with counter c0 = g + 4 (kt >> 1), word 2 (kt & 1) + (r >> 1), half r & 1.  oracle/philox.py
// `mask_attn` is the same draw.  The forward stores the decisions as keep WORDS: for (query tile qt, key tile kt, r) the 64-bit lane
// mask of the comparison - bit 16 g + jq = (query 16 qt + jq, key 16 kt + 4 g + r) - at word (qt * nkt + kt) * 4 + r of the
// (clip, head)'s block; the backward reads 16 bits of a word per (key, query tile).
#include "attn.h"
#include "bf16x3.h"
#include "philox.h"
#include "prof.h"
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>

using namespace hual;

namespace hual {
void attn_job_init(AttnJob& j) {
  ::memset((void*)&j, 0, sizeof(j));
  j.drop_site = -1;
  j.dmask = nullptr;
}
int attn_ldm(int Tk) { return 32 * cdiv(Tk, 16); }
size_t attn_keep_bytes(int B, int Tq, int Tk) { return (size_t)B * 8 * cdiv(Tq, 16) * cdiv(Tk, 16) * 32; }
}  // namespace hual

#define ATT_LOG2E 1.4426950408889634f
#define ATT_C1 (0.25f * ATT_LOG2E)              // 1/sqrt(head_size = 16) (layers.py:82), scores kept in the log2 domain
#define ATT_NEGL (HUAL_MASK_VALUE * ATT_LOG2E)  // the additive mask value (ops.py:89 via layers.py:84) in the log2 domain

// (bf16x8 is the register type of an operand fragment: the bytes are format blind, mfma_h below reads them as fp16)
__device__ __forceinline__ bf16x8 as_bf8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ bf16x8 as_bf8(uint2 a, uint2 b) { return as_bf8(make_uint4(a.x, a.y, b.x, b.y)); }
// XCD-aware block order (cdna_hip_programming.md T1, bijective form).  The dispatcher deals consecutive linear block ids
// round-robin over the 8 XCDs, each with a private L2; all blocks of one clip (jobs x heads) read that clip's rows, so the
// remap hands each XCD a contiguous run of logical ids, i.e. whole clips.
__device__ __forceinline__ int xcd_logical_id() {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---- LDS panels -----------------------------------------------------------------------------------------
// A panel holds the 16 head dims of up to 256 rows, split: row r = [16 x fp16 high parts | 16 x fp16 residuals] (64 bytes).
// It serves every operand shape of the kernels:
//   * rows as the M / N index of a product whose contraction runs over the head dims (S = Q.K^T, dP = dO.V^T): one
//     16-byte read per lane, slots [hi | lo] (panel_a) or [hi | hi], [lo | 0] (panel_b1 / panel_b2);
//   * rows as the CONTRACTION index (P.V, dS^T.Q, P^T.dO, dS.K): ds_read_b64_tr_b16 delivers a 4-row x 16-column block
//     column-major, i.e. lane (col = lane & 15, g) receives rows 4 g .. 4 g + 3 of its column (panel_tr).
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

// ---- fp16-pair operands (round 5: forward and backward; scales in the header of the backward section)
#define ATT_SX 16.0f                            // Q, K, V
#define ATT_SP 1024.0f                          // probabilities
__device__ __forceinline__ f32x4 mfma_h(bf16x8 a, bf16x8 b, f32x4 c) {      // (operand bytes are format blind: the panel helpers serve both)
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// (the same structure on bf16 instructions was measured to take the same time - the instruction kind is not what the 22-bit form costs.
//  Panels - LDS stores - split with the scale folded into the mixed-precision FMA, bf16x3.h f16_split_pair_s)
__device__ __forceinline__ void split8h(const float4& a, const float4& b, float sc, uint4& hi, uint4& lo) {
  // (the conversion sequence the compiler sees, NOT the asm form of bf16x3.h f16_split_pair_s: these halves go straight into MFMA operands,
  //  and the hazard recogniser does not look inside inline asm - a VALU write of an MFMA source needs wait states: wrong products at Tk = 256)
  f16_split_pair(a.x * sc, a.y * sc, hi.x, lo.x);
  f16_split_pair(a.z * sc, a.w * sc, hi.y, lo.y);
  f16_split_pair(b.x * sc, b.y * sc, hi.z, lo.z);
  f16_split_pair(b.z * sc, b.w * sc, hi.w, lo.w);
}
__device__ __forceinline__ void panel_store_h(char* dst, int row, int c4, const float4& v, float sc) {
  uint2 h, l;
#ifdef ATT_EXP_RAWPANEL     // timing experiment: what the splits of the fixed-scale panels (Q, K, V) cost - raw bits instead (numerically wrong)
  if (sc == ATT_SX) {
    h = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)); l = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
    *reinterpret_cast<uint2*>(dst + row * 64 + 8 * c4) = h;
    *reinterpret_cast<uint2*>(dst + row * 64 + 32 + 8 * c4) = l;
    return;
  }
#endif
  f16_split4_s(v, sc, h, l);
  *reinterpret_cast<uint2*>(dst + row * 64 + 8 * c4) = h;
  *reinterpret_cast<uint2*>(dst + row * 64 + 32 + 8 * c4) = l;
}
// power-of-two scale that brings amax >= 0 into [2^13, 2^14) and its exact inverse (amax = 0: 2^113, finite)
__device__ __forceinline__ float att_pow2_scale(float amax, float& inv) {
  uint32_t eb = (__float_as_uint(amax) >> 23) & 0xffu;
  eb = eb < 27u ? 27u : (eb > 240u ? 240u : eb);
  inv = __uint_as_float((eb - 13u) << 23);
  return __uint_as_float((267u - eb) << 23);
}
// workgroup barrier that waits for the wave's LDS operations only (__syncthreads() also drains the global stores of the job before)
__device__ __forceinline__ void att_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ======================================================================================================
// forward: ONE workgroup per (clip, head) for ALL jobs of the launch
// ======================================================================================================
// The four attention problems of a dual_attn_block call (layers.py:80-96: video self 128 x 128, video -> query 128 x 20, query self
// 20 x 20, query -> video 20 x 128 at the bench shape) used to be four workgroups per (clip, head); the small ones cost almost as
// much as the large one (a cold start, a staging round trip and a barrier each: the 4-job launch took the SUM of its jobs).  Now a
// workgroup stages the K / V panels of every job once and its waves walk a list of units = (job, 16-query tile), sorted by cost on
// the host and dealt to the waves boustrophedon (near-LPT): at the bench shape 20 units of 8 or 2 key tiles, 25 +- 1 key tiles per wave.
//
// A unit (lane j = query of the tile, g = lane >> 4):
//   S^T = K.Q^T per 16-key tile with [hi | lo] of the 16 head dims in the 32 contraction slots (two MFMAs per tile); row max / row sum
//   in-lane + two cross-lane steps; e = exp2(s - max) stays UNNORMALISED (e in [0, 1]) and the dropped, unnormalised probabilities are
//   the A operand of e.V; 1 / sum (x the dropout scale) multiplies the 16 x 16 output tile instead of the 16 x Tk probabilities.
//   Dropout: 16-bit decisions, 8 per Philox call (pair of key tiles), kept iff half < t16, scale exactly 1 / (1 - rate) as
//   tf.nn.dropout.  The comparison's lane mask IS the keep word of (query tile, key tile, r): it selects e or 0 (one v_cndmask per
//   score) and is stored as 8 bytes for the backward pass - no per-score bit arithmetic.
//   P operand: the bf16 hi + lo split of e (three MFMAs per 32 keys).  e as ONE fp16 operand (two MFMAs, no split) was built and
//   measured in round 4: 2.2e-4 .. 5.3e-4 absolute on O(1) outputs (peaked rows: 2^-12 |V|), and the backward's delta = dO . O no
//   longer matches the exact probabilities it recomputes (key-bias gradients, exact zeros by cancellation, came out at 1e-3): the
//   backward would have to round its probabilities the same way (+2 vector instructions per score there for -3 here) - not taken.
// NKT = key tiles of 16 of the unit's job (compile time per unit body; tiles beyond Tk hold zero keys with an additive term of -inf).
__device__ __forceinline__ int nkt_pad(int Tk) {
  const int n = (Tk + 15) >> 4;
  return n <= 2 ? 2 : n <= 4 ? 4 : n <= 8 ? 8 : 16;
}
__device__ __forceinline__ uint32_t attn_t16(const DropCfg& d) {            // = oracle/philox.py keep_threshold16
  const uint32_t t = (uint32_t)(((uint64_t)d.thresh + (1ull << 15)) >> 16);
  return t < 1u ? 1u : (t > 65536u ? 65536u : t);
}

// compile-time loops (the loop index is a constant expression inside the body: template arguments, asm immediates)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(static_cast<F&&>(f));
  }
}
// lanes L0 .. L0 + 7 of (lo, hi) = the eight 64-bit lane masks m[0..7] (wave-uniform values in SGPRs).  v_writelane_b32 reads its
// scalar data operand EARLY: issued right behind the v_cmp that produced the mask it read the register's previous contents (measured:
// every second keep word carried its predecessor's low half; the compiler's hazard tables know this for its own instructions, not for
// asm) - so all eight masks are computed first and the block opens with the wait states (ISA: VALU writes SGPR -> lane instruction: 4).
template <int L0>
__device__ __forceinline__ void put_lanes8(uint32_t& lo, uint32_t& hi, const unsigned long long (&m)[8]) {
  asm volatile("s_nop 4\n\t"
               "v_writelane_b32 %0, %2, %18\n\tv_writelane_b32 %1, %3, %18\n\t"
               "v_writelane_b32 %0, %4, %19\n\tv_writelane_b32 %1, %5, %19\n\t"
               "v_writelane_b32 %0, %6, %20\n\tv_writelane_b32 %1, %7, %20\n\t"
               "v_writelane_b32 %0, %8, %21\n\tv_writelane_b32 %1, %9, %21\n\t"
               "v_writelane_b32 %0, %10, %22\n\tv_writelane_b32 %1, %11, %22\n\t"
               "v_writelane_b32 %0, %12, %23\n\tv_writelane_b32 %1, %13, %23\n\t"
               "v_writelane_b32 %0, %14, %24\n\tv_writelane_b32 %1, %15, %24\n\t"
               "v_writelane_b32 %0, %16, %25\n\tv_writelane_b32 %1, %17, %25"
               : "+v"(lo), "+v"(hi)
               : "s"((uint32_t)m[0]), "s"((uint32_t)(m[0] >> 32)), "s"((uint32_t)m[1]), "s"((uint32_t)(m[1] >> 32)),
                 "s"((uint32_t)m[2]), "s"((uint32_t)(m[2] >> 32)), "s"((uint32_t)m[3]), "s"((uint32_t)(m[3] >> 32)),
                 "s"((uint32_t)m[4]), "s"((uint32_t)(m[4] >> 32)), "s"((uint32_t)m[5]), "s"((uint32_t)(m[5] >> 32)),
                 "s"((uint32_t)m[6]), "s"((uint32_t)(m[6] >> 32)), "s"((uint32_t)m[7]), "s"((uint32_t)(m[7] >> 32)),
                 "n"(L0), "n"(L0 + 1), "n"(L0 + 2), "n"(L0 + 3), "n"(L0 + 4), "n"(L0 + 5), "n"(L0 + 6), "n"(L0 + 7));
}

struct FwdUnitCtx {
  const char* Kp; const char* Vp; const float* Bias; int RT;      // the job's panels (already offset) and the stride between the two bias rows
  uint32_t k0, k1, off, t16; float scale;
};

template <int NKT, bool DROP>
__device__ __forceinline__ void attn_fwd_unit(const AttnJob& job, int b, int h, int qt, const FwdUnitCtx& c, const float4& qv0, const float4& qv1,
                                              float mq, int lane) {
  constexpr int NKP = NKT / 2;
  const int j = lane & 15, g = lane >> 4;
  const int Tq = job.Tq, Tk = job.Tk;
  const int qbase = job.qrow0 + b * Tq;
  const int q0 = qt * 16;
  const bool qok = q0 + j < Tq;
  const int qrow = qbase + min(q0 + j, Tq - 1);
  // B operands: [Q_hi | Q_hi] and [Q_lo | 0] over the 32 slots (lane g covers head dims 8 (g & 1) .. + 7)
  uint4 qh, ql;
  split8h(qv0, qv1, ATT_SX, qh, ql);
  if (g >= 2) ql = make_uint4(0u, 0u, 0u, 0u);
  const bf16x8 B1 = as_bf8(qh), B2 = as_bf8(ql);
  const float* bias = c.Bias + (mq != 0.f ? c.RT : 0);
  f32x4 s[NKT];
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const bf16x8 a = panel_a(c.Kp, 16 * kt + j, g);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = mfma_h(a, B1, acc);
    acc = mfma_h(a, B2, acc);                              // (16 K) . (16 Q)
    const float4 b4 = *reinterpret_cast<const float4*>(bias + 16 * kt + 4 * g);
    constexpr float c_s = ATT_C1 / (ATT_SX * ATT_SX);
    acc[0] = fmaf(acc[0], c_s, b4.x); acc[1] = fmaf(acc[1], c_s, b4.y);
    acc[2] = fmaf(acc[2], c_s, b4.z); acc[3] = fmaf(acc[3], c_s, b4.w);
    mx = fmaxf(fmaxf(mx, fmaxf(acc[0], acc[1])), fmaxf(acc[2], acc[3]));
    s[kt] = acc;
  }
  mx = fmaxf(mx, lane_xor16_partner(mx));      // (v_permlane16_swap / v_permlane32_swap: no LDS round trip)
  mx = fmaxf(mx, lane_xor32_partner(mx));
  // e = 2^10 exp2(s - max): the operand scale of the probabilities rides in the exponent (no multiplication), the row sum and with
  // it 1 / sum carry it too, so the normalisation of the output tile needs no correction for it; the SAVED 1 / sum is the true one
  const float mxs = mx - 10.0f;                     // log2(ATT_SP)
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __builtin_amdgcn_exp2f(s[kt][r] - mxs);
      s[kt][r] = e;
      sum += e;
    }
  }
  sum += lane_xor16_partner(sum);
  sum += lane_xor32_partner(sum);
  const float inv = 1.0f / sum;                     // = (true 1 / sum) / 2^(mx - mxs)
  const int ql_ = b * Tq + q0 + j;                  // job-local query index
  if (job.stats && g == 0 && qok) {
    job.stats[ql_ * 8 + h] = mx;
    // (mx - mxs is 10, except for rows whose every key is masked: there max = -1.44e30 absorbs the 10 and e = 1)
    job.stats[job.B * Tq * 8 + ql_ * 8 + h] = inv * __builtin_amdgcn_exp2f(mx - mxs);
  }
  // factor of output row 4 g + r: 1 / sum (x dropout scale) of query 4 g + r, which lane 4 g + r holds (x 2^-4: the V panel holds 16 V)
  const float fac = (DROP ? inv * c.scale : inv) * (1.0f / ATT_SX);
  float fr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) fr[r] = __shfl(fac, 4 * g + r);
  const uint32_t drow = (job.drop_row0 + (uint32_t)qrow) * 8u + (uint32_t)h;
  uint32_t wlo = 0u, whi = 0u;                      // lane 4 kt + r: keep word of (this query tile, key tile kt, r)
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  static_for<0, NKP>([&](auto kpc) {                // key tiles 2 kp, 2 kp + 1
    constexpr int kp = decltype(kpc)::value;
    if (DROP) {
      const uint4_ rnd = philox4x32((uint32_t)(g + 4 * kp), drow, (uint32_t)job.drop_site, c.off, c.k0, c.k1);
      const uint32_t w[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
      unsigned long long mk[8];
      static_for<0, 8>([&](auto ic) {
        constexpr int i = decltype(ic)::value >> 2, r = decltype(ic)::value & 3;
        const uint32_t word = w[2 * i + (r >> 1)];
        const bool keep = ((r & 1) ? (word >> 16) : (word & 0xffffu)) < c.t16;
        mk[decltype(ic)::value] = __ballot(keep);               // = the comparison's own lane mask
        s[2 * kp + i][r] = keep ? s[2 * kp + i][r] : 0.f;
      });
      put_lanes8<8 * kp>(wlo, whi, mk);                         // lane 4 kt + r collects the keep word of (kt, r)
    }
    uint4 ph, pl;
    split8h(make_float4(s[2 * kp][0], s[2 * kp][1], s[2 * kp][2], s[2 * kp][3]),
            make_float4(s[2 * kp + 1][0], s[2 * kp + 1][1], s[2 * kp + 1][2], s[2 * kp + 1][3]), 1.0f, ph, pl);
    const bf16x8 vh = panel_tr(c.Vp, 0, 32 * kp, lane), vl = panel_tr(c.Vp, 32, 32 * kp, lane);
    o = mfma_h(as_bf8(ph), vh, o);
    o = mfma_h(as_bf8(ph), vl, o);
    o = mfma_h(as_bf8(pl), vh, o);
  });
  if (DROP && job.dmask) {      // keep words of the tile's real key tiles: one 8-byte store per lane, contiguous over the lanes
    const int nkt = (Tk + 15) >> 4, nqt = (Tq + 15) >> 4;
    if (lane < 4 * nkt)
      *reinterpret_cast<uint2*>(job.dmask + (((size_t)(b * 8 + h) * nqt + qt) * nkt * 4 + lane) * 8) = make_uint2(wlo, whi);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + 4 * g + r;
    if (q < Tq) job.O[(size_t)(qbase + q) * job.ldo + 16 * h + j] = o[r] * fr[r];
  }
}

// MAXNKT: the largest tile count among the jobs of the launch (the kernel's register budget is that of its largest body)
// NT threads per workgroup: the units are latency chains (LDS read -> MFMA -> cross-lane reductions -> exp -> Philox -> MFMA chain), so a
// SIMD wants as many waves as the registers allow: 512 threads x 2 workgroups per CU = 4 waves per SIMD where the body fits 128 registers
template <int MAXNKT, int NT>
__global__ __launch_bounds__(NT, (NT == 512 && MAXNKT <= 8) ? 4 : 2) void attn_fwd_kernel(AttnBatch batch, int njobs, AttnUnits units, DropCfg drop) {
  constexpr int NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lid = xcd_logical_id();
  const int h = lid & 7, b = lid >> 3;
  if (b >= batch.j[0].B) return;   // block-uniform (grid rounded up to 8 clips)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 15, g = lane >> 4;
  // panel rows of the jobs: job i owns rows ko[i] .. ko[i] + 16 nkt_pad(Tk_i) - 1 of the K and the V panel
  int ko1, ko2, ko3, RT;
  {
    const int t0 = 16 * nkt_pad(batch.j[0].Tk), t1 = njobs > 1 ? 16 * nkt_pad(batch.j[1].Tk) : 0;
    const int t2 = njobs > 2 ? 16 * nkt_pad(batch.j[2].Tk) : 0, t3 = njobs > 3 ? 16 * nkt_pad(batch.j[3].Tk) : 0;
    ko1 = t0; ko2 = ko1 + t1; ko3 = ko2 + t2; RT = ko3 + t3;
  }
  char* Kp = lds;                                   // [RT][64]
  char* Vp = Kp + RT * 64;                          // [RT][64]
  float* Bias = reinterpret_cast<float*>(Vp + RT * 64);      // [2][RT]: additive term of a key for padded / valid queries
  uint32_t k0 = 0, k1 = 0, off = 0;
  if (drop.enabled) {      // (through the constant address space: scalar loads, no vector-memory wait in front of the staging loads)
    const __attribute__((address_space(4))) uint32_t* sp = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)drop.state;
    k0 = sp[0]; k1 = sp[1]; off = sp[2];
  }
  // the wave's units: entries NW k + w (k even) / NW k + NW - 1 - w (k odd) of the cost-sorted list
  const int nun = units.n;
  auto unit_of = [&](int k) { const int i = NW * k + ((k & 1) ? NW - 1 - wave : wave); return i < nun ? (int)units.u[i] : -1; };
  float4 qv0 = f4zero(), qv1 = f4zero(), nq0 = f4zero(), nq1 = f4zero();
  float mq = 0.f, nmq = 0.f;
  auto load_q = [&](int u, float4& d0, float4& d1, float& m) {
    const AttnJob& J = batch.j[u >> 4];
    const int qrow = J.qrow0 + b * J.Tq + min(16 * (u & 15) + j, J.Tq - 1);
    const float* qp = J.Q + (size_t)qrow * J.ldq + 16 * h + 8 * (g & 1);
    d0 = ld4(qp); d1 = ld4(qp + 4);
    m = J.qmask[qrow];
  };
  int ucur = unit_of(0);
  if (ucur >= 0) load_q(ucur, qv0, qv1, mq);        // requested before the staging so that it arrives under it
  {
    // staging: every K / V row of every job once; ALL loads first, then the splits and LDS stores.  The loop over the jobs is
    // static (job fields by constant index: their scalar loads sit at the top of the kernel - indexing the job table by a computed
    // job id put a scalar-memory round trip in front of every batch of loads: 7.5 us for a launch that computed nothing)
    constexpr int ITER = (1024 + NT - 1) / NT;        // a job has <= 256 panel rows x 4 pieces
    float kmv[HUAL_MAX_ATTN_JOBS];
    float4 kv[HUAL_MAX_ATTN_JOBS][ITER], vv[HUAL_MAX_ATTN_JOBS][ITER];
#pragma unroll
    for (int jb = 0; jb < HUAL_MAX_ATTN_JOBS; ++jb) {      // (unconditional on clamped indices: a load behind a branch is waited for inside it)
      const AttnJob& J = batch.j[jb < njobs ? jb : 0];
      kmv[jb] = J.kmask[J.krow0 + b * J.Tk + min((int)threadIdx.x, J.Tk - 1)];
    }
#pragma unroll
    for (int jb = 0; jb < HUAL_MAX_ATTN_JOBS; ++jb) {
      const AttnJob& J = batch.j[jb];
      const int items = jb < njobs ? 64 * nkt_pad(J.Tk) : 0;
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        kv[jb][it] = f4zero(); vv[jb][it] = f4zero();
        if (it * NT < items) {                        // (workgroup-uniform)
          const int idx = it * NT + (int)threadIdx.x;
          const int r = min(idx >> 2, J.Tk - 1), c4 = idx & 3;
          const size_t goff = (size_t)(J.krow0 + b * J.Tk + r) * J.ldkv + 16 * h + 4 * c4;
          kv[jb][it] = ld4(J.K + goff);
          vv[jb][it] = ld4(J.V + goff);
        }
      }
    }
#pragma unroll
    for (int jb = 0; jb < HUAL_MAX_ATTN_JOBS; ++jb) {
      const int Tk = batch.j[jb].Tk;
      const int items = jb < njobs ? 64 * nkt_pad(Tk) : 0;
      const int ko = jb == 0 ? 0 : jb == 1 ? ko1 : jb == 2 ? ko2 : ko3;
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int idx = it * NT + (int)threadIdx.x;
        if (idx < items) {
          const int row = idx >> 2, c4 = idx & 3;
          const bool in = row < Tk;
          panel_store_h(Kp, ko + row, c4, in ? kv[jb][it] : f4zero(), ATT_SX);
          panel_store_h(Vp, ko + row, c4, in ? vv[jb][it] : f4zero(), ATT_SX);
        }
      }
    }
#pragma unroll
    for (int jb = 0; jb < HUAL_MAX_ATTN_JOBS; ++jb) {
      if (jb < njobs) {
        const int ko = jb == 0 ? 0 : jb == 1 ? ko1 : jb == 2 ? ko2 : ko3;
        const int Tk = batch.j[jb].Tk, Tkp = 16 * nkt_pad(Tk);
        if ((int)threadIdx.x < Tkp) {
          const bool in = (int)threadIdx.x < Tk;
          Bias[ko + threadIdx.x] = in ? ATT_NEGL : -INFINITY;                                  // layers.py:84: (1 - mq mk) * -1e30, mq = 0
          Bias[RT + ko + threadIdx.x] = in ? (kmv[jb] != 0.f ? 0.f : ATT_NEGL) : -INFINITY;
        }
      }
    }
  }
  __syncthreads();
  FwdUnitCtx c;
  c.RT = RT; c.k0 = k0; c.k1 = k1; c.off = off; c.t16 = attn_t16(drop); c.scale = drop.scale;
  for (int k = 0; ucur >= 0; ++k) {
    const int unext = unit_of(k + 1);
    load_q(unext >= 0 ? unext : ucur, nq0, nq1, nmq);      // (the last unit asks for itself again: no branch around the loads)
    const int jb = ucur >> 4, qt = ucur & 15;
    const AttnJob& J = batch.j[jb];
    const int ko = jb == 0 ? 0 : jb == 1 ? ko1 : jb == 2 ? ko2 : ko3;
    c.Kp = Kp + ko * 64; c.Vp = Vp + ko * 64; c.Bias = Bias + ko;
    const int nktp = nkt_pad(J.Tk);
    if (J.drop_site >= 0 && drop.enabled) {
      if (nktp == 2) attn_fwd_unit<2, true>(J, b, h, qt, c, qv0, qv1, mq, lane);
      else if (MAXNKT >= 4 && nktp == 4) attn_fwd_unit<4, true>(J, b, h, qt, c, qv0, qv1, mq, lane);
      else if (MAXNKT >= 8 && nktp == 8) attn_fwd_unit<8, true>(J, b, h, qt, c, qv0, qv1, mq, lane);
      else if (MAXNKT >= 16) attn_fwd_unit<16, true>(J, b, h, qt, c, qv0, qv1, mq, lane);
    } else {
      if (nktp == 2) attn_fwd_unit<2, false>(J, b, h, qt, c, qv0, qv1, mq, lane);
      else if (MAXNKT >= 4 && nktp == 4) attn_fwd_unit<4, false>(J, b, h, qt, c, qv0, qv1, mq, lane);
      else if (MAXNKT >= 8 && nktp == 8) attn_fwd_unit<8, false>(J, b, h, qt, c, qv0, qv1, mq, lane);
      else if (MAXNKT >= 16) attn_fwd_unit<16, false>(J, b, h, qt, c, qv0, qv1, mq, lane);
    }
    qv0 = nq0; qv1 = nq1; mq = nmq;
    ucur = unext;
  }
}

// ======================================================================================================
// backward
// ======================================================================================================
// Round 5: the backward's five products run on FP16 pairs (x s = hi + lo, 22 significant bits, v_mfma_f32_16x16x32_f16) instead of bf16
// pairs (16 bits): the 2^-16 per product of rounds 2-4 was, with the context-query backward, what put every gradient tensor upstream
// of an attention block 2-4e-5 of its maximum from the float64 oracle (a float32 PyTorch implementation: 5e-6).  fp16's narrow range
// needs a power-of-two scale that is constant along each contraction:
//   Q, K, V   fixed 2^4 (projections of layer-norm outputs, the forward's activation scale: |x| >= 4094 ends in Inf / NaN gradients).
//             The scale is NOT cosmetic: measured with the operands as they are, every gradient tensor moved from 5.3e-6 back to 8.6e-6
//             of its maximum - an element below 2^-3 has a residual below fp16's normal range, and the subnormal residuals do not
//             survive (conversion / matrix pipe flush them): such an element keeps 11 bits instead of 22.  2^4 moves that edge to 2^-7;
//             it costs one v_pk_mul per element pair in the staging (+5.7 us per step over the six attention launches' issue-bound
//             split-and-store phase, same-box A/B);
//   P         fixed 2^10 (p <= 1, x 1 / (1 - rate));
//   dO        ONE scale per (clip, head) workgroup from the largest |dO| of the head's panel (a workgroup maximum in the staging:
//             one extra LDS-only barrier per job);
//   dS        one scale per (wave, 32-key block) from a BOUND: |dS| <= p (16 max|dO| max|V_block| / (1 - rate) + max|delta|), with max|V|
//             of the block from the wave's own V registers and max|delta| from the staging - no reduction inside the query loop;
//             the bound is loose by 2^10..2^14 against typical values, which fp16's 2^-24 .. 2^15 span absorbs (hi + lo still carry
//             >= 19 bits of the largest element of a contraction).
// Same instruction count in the products (13 MFMAs per 16 x 16 tile), the same split cost (v_cvt_pk_f16_f32 + v_cvt_f32_f16 + v_sub
// against v_cvt_pk_bf16_f32 + shifts + v_sub); the scales ride on multiplications that were there (ATT_C1, 1 / sum, the 0.25 of the
// epilogues) except for one v_cndmask and one v_mul per score.
// largest |dO| and |delta| of the head: every wave leaves its maxima in Red[wave], Red[4 + wave] during the staging (plain stores, no
// barrier of their own); attn_bwd_compute reads all eight behind the barrier that ends the staging
__device__ __forceinline__ float att_wave_max(float v) {      // DPP / permlane butterflies only (no LDS round trip)
  v = fast_max32(v);
  return fmaxf(v, lane_xor32_partner(v));
}
template <int NW = 4>
__device__ __forceinline__ void att_wave_max2_put(float* Red, float a, float b) {
  a = att_wave_max(a); b = att_wave_max(b);
  if ((threadIdx.x & 63) == 0) { Red[threadIdx.x >> 6] = a; Red[NW + (threadIdx.x >> 6)] = b; }
}

// LDS map of one (job, clip, head), Tqp = Tq rounded up to 32 queries, Tkp likewise:
//   Qp, Dp   [Tqp][64]       split Q / dO panels
//   Kp       [Tkp][64]       split K panel (the V rows of a wave's keys go from HBM straight into its B operands)
//   St       [7][Tqp]        row max (log2 domain), 1 / row sum, delta, additive term for masked / valid / padding keys, 1 / scale of the dO row
//   Mk       [nqt][nkt][4]   keep words of the head (8 bytes each, forward layout)
//   dQw      [4 waves][Tqp][16]  fp32 dQ partial products, one slot per wave (plain stores: float atomics on LDS retire at
//                            about one lane per clock and cost more than the rest of the kernel)
//   Xs       [waves][32][20]     transposition scratch
//   Red      [16]            per-wave maxima of |dO| and |delta| (staging)
// (nw = waves of the workgroup: 4, or 8 for the large job of a launch whose jobs do not fit two workgroups per CU - attn_bwd_big_kernel)
struct BwdLds { int qp, dp, kp, st, mk, dqw, xs, red, total; };
__host__ __device__ inline BwdLds bwd_lds(int Tq, int Tk, bool dropout, int nw = 4) {
  const int Tqp = (Tq + 31) & ~31, Tkp = (Tk + 31) & ~31;
  BwdLds l;
  int o = 0;
  l.qp = o; o += Tqp * 64;
  l.dp = o; o += Tqp * 64;
  l.kp = o; o += Tkp * 64;
  l.st = o; o += 7 * Tqp * 4;
  l.mk = o; o += dropout ? ((Tq + 15) >> 4) * ((Tk + 15) >> 4) * 32 : 0;
  l.dqw = o; o += 4 * Tqp * 64;
  l.xs = o; o += nw * 32 * 20 * 4;
  l.red = o; o += 64;
  l.total = o;
  return l;
}

#ifdef HUAL_STAMPS
// debug: clock stamps of the backward kernel's phases per workgroup (scripts/exp/attn_stamps.py)
__device__ unsigned long long g_attn_stamps[4096 * 8];
extern "C" int hual_debug_attn_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * (size_t)n);
}
#define ATT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_attn_stamps[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define ATT_STAMP(i) do { } while (0)
#endif

// The backward of one job is cut in three: the V rows of a wave's first key block (registers), the staging of the panels (every load
// requested before the first LDS store), and the products + epilogue.  A workgroup of the plain kind runs them back to back
// (attn_bwd_body); the workgroup that serves the SMALL jobs of a four-job launch runs the next job's loads under the current job's
// products (attn_bwd_chain below).
struct BwdV { float4 v[2][2]; float km[2]; };
template <bool DROP>
__device__ __forceinline__ void attn_bwd_load_v(const AttnJob& job, int b, int h, int kp, BwdV& bv) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int Tk = job.Tk, kbase = job.krow0 + b * Tk;
  const float* Vg = job.V + (size_t)kbase * job.ldkv + 16 * h;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int krow = min(32 * kp + 16 * t + j, Tk - 1);
    const float* vptr = Vg + (size_t)krow * job.ldkv + 8 * (g & 1);
    bv.v[t][0] = ld4(vptr); bv.v[t][1] = ld4(vptr + 4);
    bv.km[t] = job.kmask[kbase + krow];
  }
}
__device__ __forceinline__ bool attn_bwd_qsplit(int Tq, int Tk) { return (((Tk + 31) & ~31) >> 5) == 1 && (((Tq + 31) & ~31) >> 5) >= 4; }

// ---- staging of a job (any shape up to 256 x 256): loads and LDS stores in one piece
template <bool DROP, int NW = 4>
__device__ __forceinline__ void attn_bwd_stage(const AttnJob& job, int b, int h, char* lds) {
  constexpr int NT = 64 * NW;                           // threads of the workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Tq = job.Tq, Tk = job.Tk;
  const int Tqp = (Tq + 31) & ~31, Tkp = (Tk + 31) & ~31;
  const int qbase = job.qrow0 + b * Tq, kbase = job.krow0 + b * Tk;
  constexpr bool dodrop = DROP;
  const BwdLds L = bwd_lds(Tq, Tk, dodrop, NW);
  const int nqt = (Tq + 15) >> 4, nkt = (Tk + 15) >> 4;
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
  (void)j; (void)g; (void)Xs; (void)dQw; (void)Vg; (void)nkp; (void)nqp; (void)wave;
  ATT_STAMP(0);
  // ---- staging: EVERY load of the workgroup's panels (+ O for delta = dO . O, the softmax statistics, the keep words) is requested
  // before the first split / LDS store: one memory round trip (a pass per 64 rows used to wait for its own loads, and the loads
  // of the statistics sat behind a lane-dependent branch: three to five serial round trips in front of the products)
  const int stat_n = job.B * Tq * 8;
  const int npass = (max(Tqp, Tkp) * 4 + NT - 1) / NT;
  constexpr int MAXP = 1024 / NT;                       // Tq, Tk <= 256: four passes of 256 threads, two of 512
  float4 sq[MAXP], sd[MAXP], so[MAXP], sk[MAXP];
  float ssm[MAXP], ssi[MAXP], sqm[MAXP];
  uint2 mkw[MAXP];                                      // keep words of the head (<= 16 x 16 x 4 = 1024: MAXP per thread)
#pragma unroll
  for (int it = 0; it < MAXP; ++it) mkw[it] = make_uint2(0u, 0u);
  const int nmk = dodrop ? nqt * nkt * 4 : 0;
  // order of the requests = order of arrival = order of use: dO and O first (the workgroup maximum of |dO| - the scale of the dO panel -
  // is the one thing everything else has to wait for), then Q / K, whose panels (fixed scale) are split and stored while that
  // maximum crosses the workgroup, then the statistics and the keep words
#pragma unroll
  for (int it = 0; it < MAXP; ++it) {
    sd[it] = so[it] = f4zero();
    if (it < npass) {                                   // (workgroup-uniform)
      const int idx = threadIdx.x + NT * it, row = idx >> 2, c4 = idx & 3;
      const int qr = min(row, Tq - 1);
      sd[it] = ld4(Dg + (size_t)qr * job.lddo + 4 * c4);
      so[it] = ld4(Og + (size_t)qr * job.ldo + 4 * c4);
    }
  }
#pragma unroll
  for (int it = 0; it < MAXP; ++it) {
    sq[it] = sk[it] = f4zero();
    ssm[it] = ssi[it] = sqm[it] = 0.f;
    if (it < npass) {
      const int idx = threadIdx.x + NT * it, row = idx >> 2, c4 = idx & 3;
      const int qr = min(row, Tq - 1), kr = min(row, Tk - 1);
      sq[it] = ld4(Qg + (size_t)qr * job.ldq + 4 * c4);
      sk[it] = ld4(Kg + (size_t)kr * job.ldkv + 4 * c4);
      const int si = (b * Tq + qr) * 8 + h;             // (unconditional on the clamped row: every lane of a row reads the same words)
      ssm[it] = job.stats[si];
      ssi[it] = job.stats[stat_n + si];
      sqm[it] = job.qmask[qbase + qr];
    }
  }
  if (dodrop) {
    const uint2* src = reinterpret_cast<const uint2*>(job.dmask + (size_t)(b * 8 + h) * nqt * nkt * 32);
#pragma unroll
    for (int it = 0; it < MAXP; ++it) mkw[it] = src[min((int)threadIdx.x + NT * it, nmk - 1)];
  }
  // delta = dO . O per query row; the head's largest |dO| and |delta| (scales of the dO panel and of dS): per-wave maxima into LDS
  float part[MAXP], gmax = 0.f, dmax = 0.f;
#pragma unroll
  for (int it = 0; it < MAXP; ++it) {
    part[it] = 0.f;
    if (it < npass) {
      const int row = (threadIdx.x + NT * it) >> 2;
      const bool qok = row < Tq;
      const float4 dv = sd[it], ov = so[it];
      float pt = qok ? (dv.x * ov.x + dv.y * ov.y) + (dv.z * ov.z + dv.w * ov.w) : 0.f;
      pt += dpp_xor_partner(pt, 1);            // (quad butterflies on the VALU: __shfl_xor is an LDS round trip each)
      pt += dpp_xor_partner(pt, 2);
      part[it] = pt;
      gmax = fmaxf(gmax, qok ? f4absmax(dv) : 0.f);
      dmax = fmaxf(dmax, fabsf(pt));
    }
  }
  // the per-wave maxima go to LDS as they are: attn_bwd_compute combines them behind the barrier that ends the staging anyway
  att_wave_max2_put<NW>(reinterpret_cast<float*>(lds + L.red), gmax, dmax);
#pragma unroll
  for (int it = 0; it < MAXP; ++it) {
    if (it < npass) {
      const int idx = threadIdx.x + NT * it, row = idx >> 2, c4 = idx & 3;
      const bool qok = row < Tq;
      // dO row: its OWN power-of-two scale (the four lanes of a row agree on the row maximum with two quad steps) - no workgroup-wide
      // quantity in front of the panel stores.  dP = dO . V^T contracts along the row (any row scale divides out per row, St[6]);
      // dV = Pd^T . dO contracts over the rows: there Pd carries the ratio to the head's common scale (attn_bwd_compute)
      float rmax = qok ? f4absmax(sd[it]) : 0.f;
      rmax = fmaxf(rmax, dpp_xor_partner(rmax, 1));
      rmax = fmaxf(rmax, dpp_xor_partner(rmax, 2));
      float rinv;
#ifdef ATT_EXP_DOFIX      // timing experiment: what the per-row scale of the dO panel costs (numerically wrong for small gradients)
      const float rsc = 1024.0f; rinv = 1.0f / 1024.0f; (void)rmax;
#else
      const float rsc = att_pow2_scale(rmax, rinv);
#endif
      if (row < Tqp) {
        panel_store_h(Qp, row, c4, qok ? sq[it] : f4zero(), ATT_SX);
        panel_store_h(Dp, row, c4, qok ? sd[it] : f4zero(), rsc);
      }
      if (row < Tkp) panel_store_h(Kp, row, c4, row < Tk ? sk[it] : f4zero(), ATT_SX);
      if (c4 == 0 && row < Tqp) {
        St[row] = qok ? ssm[it] : 0.f;
        St[Tqp + row] = qok ? ssi[it] : 0.f;                                       // 1 / row sum = 0 for padding queries -> p = 0
        St[2 * Tqp + row] = part[it];
        St[3 * Tqp + row] = ATT_NEGL;                                              // key masked
        St[4 * Tqp + row] = (qok && sqm[it] != 0.f) ? 0.f : ATT_NEGL;              // key valid: (1 - mq) * -1e30
        St[5 * Tqp + row] = -INFINITY;                                             // key beyond Tk (tile padding)
        St[6 * Tqp + row] = rinv;                                                  // 1 / scale of the dO row
      }
    }
  }
  if (dodrop) {
    uint2* mk2 = reinterpret_cast<uint2*>(Mk);
#pragma unroll
    for (int it = 0; it < MAXP; ++it)
      if ((int)threadIdx.x + NT * it < nmk) mk2[threadIdx.x + NT * it] = mkw[it];
  }
}

// ---- products + epilogue of a staged job (ends with the dQ stores; the caller separates it from the next staging by a barrier)
// NW = 8 (attn_bwd_big_kernel: a job of more than 128 queries AND keys, whose 137 KB of LDS leave room for one workgroup per CU - with
// four waves that is ONE wave per SIMD, and neither pipe of a SIMD works under the other's latency): every wave takes ONE block of 32
// keys, so dK / dV stay private to a wave; the dQ slot of key blocks w and w + 4 is shared by waves w and w + 4, which walk the query
// pairs in lockstep - a workgroup barrier per pair - half a revolution apart (wave w + 4 starts at pair h = ceil(nqp / 2)): at every
// step the two waves touch different rows of the slot, and the barrier orders a row's first write (by the wave that gets there first:
// w for pairs < h, w + 4 for pairs >= h) before the other wave's read-modify-write.
template <bool DROP, int NW = 4>
__device__ __forceinline__ void attn_bwd_compute(const AttnJob& job, int b, int h, char* lds, const DropCfg& drop, BwdV& bv, bool have_v) {
  constexpr int NT = 64 * NW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Tq = job.Tq, Tk = job.Tk;
  const int Tqp = (Tq + 31) & ~31, Tkp = (Tk + 31) & ~31;
  const int qbase = job.qrow0 + b * Tq, kbase = job.krow0 + b * Tk;
  constexpr bool dodrop = DROP;
  const BwdLds L = bwd_lds(Tq, Tk, dodrop, NW);
  const int nqt = (Tq + 15) >> 4, nkt = (Tk + 15) >> 4;
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
  (void)Qg; (void)Dg; (void)Og; (void)Kg; (void)Vg; (void)nqt;
  float4 (&vreg)[2][2] = bv.v;
  float (&kmv)[2] = bv.km;
  auto load_v = [&](int kp) { attn_bwd_load_v<DROP>(job, b, h, kp, bv); };
  // ONE block of <= 32 keys and at least four 32-query pairs (the video -> query jobs): the waves split the QUERIES instead of the
  // keys - all four work on key block 0, wave w takes the query pairs w, w + 4, ..; dQ rows are then disjoint (one slot), and the
  // partial dK / dV of the waves are summed through the three free slots
  const bool qsplit = NW == 4 && nkp == 1 && nqp >= 4;
  if (!have_v && (wave < nkp || qsplit)) load_v(qsplit ? 0 : wave);
  const float scale8 = drop.scale;                  // exactly 1 / (1 - rate)
  // scales (see the header of this section): the head's largest |dO| and |delta| come from the staging
  const float* Red = reinterpret_cast<const float*>(lds + L.red);
  // (wave-uniform values: through readfirstlane into scalar registers, with everything derived from them)
  float gmax, dmax;
  {
    float4 va = *reinterpret_cast<const float4*>(Red), vb = *reinterpret_cast<const float4*>(Red + NW);
    if (NW == 8) {
      const float4 va2 = *reinterpret_cast<const float4*>(Red + 4), vb2 = *reinterpret_cast<const float4*>(Red + 12);
      va = make_float4(fmaxf(va.x, va2.x), fmaxf(va.y, va2.y), fmaxf(va.z, va2.z), fmaxf(va.w, va2.w));
      vb = make_float4(fmaxf(vb.x, vb2.x), fmaxf(vb.y, vb2.y), fmaxf(vb.z, vb2.z), fmaxf(vb.w, vb2.w));
    }
    gmax = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fmaxf(fmaxf(va.x, va.y), fmaxf(va.z, va.w)))));
    dmax = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fmaxf(fmaxf(vb.x, vb.y), fmaxf(vb.z, vb.w)))));
  }
  // the dO panel holds dO_q * s_q with the row's own scale s_q >= sg = the scale of the head's largest |dO|; St[6] = 1 / s_q
  float sg_inv;
  const float sg = att_pow2_scale(gmax, sg_inv);
  const float c_s = ATT_C1 / (ATT_SX * ATT_SX);     // scores: (16 Q) . (16 K)
  const float m_dp0 = (dodrop ? scale8 : 1.0f) / ATT_SX;      // dP = dO . V^T: (s_q dO) . (16 V) -> x 1 / (16 s_q) per row
  const float c_dv = sg_inv / ATT_SP;               // dV = Pd^T . dO with Pd_q x (sg / s_q) <= 1: sum_q (2^10 Pd sg / s_q) . (s_q dO)
  float c_dk = 0.f;                                 // 1 / (16 s_ds) of the current key block (dK, dQ)
  float* slot = dQw + (qsplit ? 0 : (wave & 3)) * Tqp * 16;
  bool first = true;
  const int qp0 = qsplit ? wave : 0, qpstep = qsplit ? 4 : 1;
  const int hrev = (nqp + 1) >> 1;                  // (NW = 8) half a revolution of the query pairs
  const bool partner = NW == 8 && wave < 4 && wave + 4 < nkp;      // (NW = 8, waves 0 .. 3) a wave shares this wave's dQ slot
  for (int kp = qsplit ? 0 : wave; kp < nkp; kp += NW) {
    // B operands of this wave's 32 keys: [K_hi | K_hi], [K_lo | 0] and the same of V, per 16-key tile t
    bf16x8 Kb1[2], Kb2[2], Vb1[2], Vb2[2];
    int bsel[2];
    if (kp != wave && !qsplit) load_v(kp);
    float vmax = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = 32 * kp + 16 * t + j;
      Kb1[t] = panel_b1(Kp, key, g); Kb2[t] = panel_b2(Kp, key, g);
      const float4 v0 = key < Tk ? vreg[t][0] : f4zero(), v1 = key < Tk ? vreg[t][1] : f4zero();
      vmax = fmaxf(vmax, fmaxf(f4absmax(v0), f4absmax(v1)));
      uint4 h4, l4;
      split8h(v0, v1, ATT_SX, h4, l4);
      if (g >= 2) l4 = make_uint4(0u, 0u, 0u, 0u);
      Vb1[t] = as_bf8(h4); Vb2[t] = as_bf8(l4);
      bsel[t] = key < Tk ? (kmv[t] != 0.f ? 4 : 3) : 5;
    }
    // |dS| <= p (|dP| / (1 - rate) + |delta|) <= 16 max|dO| max|V of these keys| / (1 - rate) + max|delta|   (p <= 1 up to rounding: x 2)
#ifdef ATT_EXP_NOVMAX     // timing experiment
    vmax = 1.0f;
#else
    vmax = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wave_max64(vmax))));
#endif
    float s_ds_inv;
    // (the bound itself goes to [2^14, 2^15): p <= 1 + 1e-4 keeps p |t| below fp16's 65504, and every binade the scale gives away is a binade
    //  of small dS elements whose residual falls below fp16's normal range and is lost)
    float s_ds = att_pow2_scale(16.0f * gmax * vmax * (dodrop ? scale8 : 1.0f) + dmax, s_ds_inv);
    s_ds *= 2.0f; s_ds_inv *= 0.5f;
    c_dk = s_ds_inv / ATT_SX;
    const float m_pd0 = (dodrop ? scale8 : 1.0f) * ATT_SP * s_ds_inv * sg;      // pd = (p s_ds) m_pd0 / s_q = 2^10 (sg / s_q) x dropped probability
    const bf16x8 kh = panel_tr(Kp, 0, 32 * kp, lane), kl = panel_tr(Kp, 32, 32 * kp, lane);      // B operand of dQ
    f32x4 dk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int qi = qp0; qi < nqp; qi += qpstep) {
      int qp = qi;
      bool fst = first;
      if (NW == 8) {
        if (wave >= 4) { qp = qi + hrev; qp = qp >= nqp ? qp - nqp : qp; fst = qp >= hrev; }
        else fst = !partner || qp < hrev;
      }
      float pd[2][2][4], ds[2][2][4];          // [t][u][r]
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q0 = 32 * qp + 16 * u;
        const bf16x8 aq = panel_a(Qp, q0 + j, g), ad = panel_a(Dp, q0 + j, g);
        const float4 m4 = *reinterpret_cast<const float4*>(St + q0 + 4 * g);
        const float4 i4 = *reinterpret_cast<const float4*>(St + Tqp + q0 + 4 * g);
        const float4 d4 = *reinterpret_cast<const float4*>(St + 2 * Tqp + q0 + 4 * g);
        // (1 / sum x s_ds: the scale of the dS operand rides on the normalisation)
        const float4 g4 = *reinterpret_cast<const float4*>(St + 6 * Tqp + q0 + 4 * g);
        const float mxv[4] = {m4.x, m4.y, m4.z, m4.w}, inv[4] = {i4.x * s_ds, i4.y * s_ds, i4.z * s_ds, i4.w * s_ds}, dlv[4] = {d4.x, d4.y, d4.z, d4.w};
#ifdef ATT_EXP_MCONST     // timing experiment: per-row factors of the dO scale replaced by constants (numerically wrong)
        const float m_pd[4] = {m_pd0, m_pd0, m_pd0, m_pd0}, m_dp[4] = {m_dp0, m_dp0, m_dp0, m_dp0}; (void)g4;
#else
        const float m_pd[4] = {g4.x * m_pd0, g4.y * m_pd0, g4.z * m_pd0, g4.w * m_pd0}, m_dp[4] = {g4.x * m_dp0, g4.y * m_dp0, g4.z * m_dp0, g4.w * m_dp0};
#endif
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = mfma_h(aq, Kb1[t], s);
          s = mfma_h(aq, Kb2[t], s);           // lane: 256 S[query q0 + 4 g + r][key 32 kp + 16 t + j]
          dp = mfma_h(ad, Vb1[t], dp);
          dp = mfma_h(ad, Vb2[t], dp);         // 16 sg dP
          const float4 b4 = *reinterpret_cast<const float4*>(St + bsel[t] * Tqp + q0 + 4 * g);
          const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
          uint32_t mbits = 0xfu;
          if (dodrop) {
            // keep bits of queries q0 + 4 g + r for this lane's key 16 kt + j: bits 4 g .. 4 g + 3 of the 16-bit field j >> 2 of
            // keep word (qt, kt, r = j & 3); tiles beyond the last real one are clamped (their probabilities are 0)
            const int qtu = min(2 * qp + u, nqt - 1), ktt = min(2 * kp + t, nkt - 1);
            const uint32_t f = *reinterpret_cast<const uint16_t*>(Mk + ((qtu * nkt + ktt) * 4 + (j & 3)) * 8 + 2 * (j >> 2));
            mbits = (f >> (4 * g)) & 0xfu;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = fmaf(s[r], c_s, bv[r]);
            const float p = __builtin_amdgcn_exp2f(v - mxv[r]) * inv[r];          // s_ds x probability
            const bool keep = !dodrop || ((mbits >> r) & 1u);
            pd[t][u][r] = p * (keep ? m_pd[r] : 0.f);                            // 2^10 (sg / s_q) x dropped probability: A operand of dV
            ds[t][u][r] = p * fmaf(dp[r], keep ? m_dp[r] : 0.f, -dlv[r]);        // s_ds x dS / 0.25: A operand of dK, dQ
          }
        }
      }
      // dV += Pd^T . dO, dK += dS^T . Q over the 32 queries of the pair
      const bf16x8 doh = panel_tr(Dp, 0, 32 * qp, lane), dol = panel_tr(Dp, 32, 32 * qp, lane);
      const bf16x8 qh = panel_tr(Qp, 0, 32 * qp, lane), ql = panel_tr(Qp, 32, 32 * qp, lane);
      uint4 dsh[2], dsl[2];                      // split dS of key tile t: halves .x .y = query tile u = 0 (r = 0..3), .z .w = u = 1
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint4 ah, al;
        split8h(make_float4(pd[t][0][0], pd[t][0][1], pd[t][0][2], pd[t][0][3]),
                make_float4(pd[t][1][0], pd[t][1][1], pd[t][1][2], pd[t][1][3]), 1.0f, ah, al);
        dv[t] = mfma_h(as_bf8(ah), doh, dv[t]);
        dv[t] = mfma_h(as_bf8(ah), dol, dv[t]);
        dv[t] = mfma_h(as_bf8(al), doh, dv[t]);
        split8h(make_float4(ds[t][0][0], ds[t][0][1], ds[t][0][2], ds[t][0][3]),
                make_float4(ds[t][1][0], ds[t][1][1], ds[t][1][2], ds[t][1][3]), 1.0f, dsh[t], dsl[t]);
        dk[t] = mfma_h(as_bf8(dsh[t]), qh, dk[t]);
        dk[t] = mfma_h(as_bf8(dsh[t]), ql, dk[t]);
        dk[t] = mfma_h(as_bf8(dsl[t]), qh, dk[t]);
      }
      // dQ partial of the pair's two query tiles: dS . K contracts over the keys, i.e. over the LANES of the dS tiles.  The halves that
      // were just split for dK go through the wave's scratch as a [32 keys][16 queries] 16-bit tile per plane and come back through
      // the transposing read - the A operand of the product, keys in the order the K panel's transposed B operand has them (round 4
      // moved the fp32 values and split them a second time: 16 more values to split per lane and iteration)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        char* Xw = reinterpret_cast<char*>(Xs);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          *reinterpret_cast<uint2*>(Xw + (16 * t + j) * 32 + 8 * g) = u == 0 ? make_uint2(dsh[t].x, dsh[t].y) : make_uint2(dsh[t].z, dsh[t].w);
          *reinterpret_cast<uint2*>(Xw + 1024 + (16 * t + j) * 32 + 8 * g) = u == 0 ? make_uint2(dsl[t].x, dsl[t].y) : make_uint2(dsl[t].z, dsl[t].w);
        }
        asm volatile("" ::: "memory");
        const int xoff = (4 * g + (j >> 2)) * 32 + 8 * (j & 3);      // lane 4 q + p of a 16-lane group: row q of the group's four keys, queries 4 p ..
        const bf16x8 xh = join_tr(lds_read_tr16(Xw, xoff), lds_read_tr16(Xw, xoff + 16 * 32));
        const bf16x8 xl = join_tr(lds_read_tr16(Xw, 1024 + xoff), lds_read_tr16(Xw, 1024 + xoff + 16 * 32));
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
        dq = mfma_h(xh, kh, dq);
        dq = mfma_h(xh, kl, dq);
        dq = mfma_h(xl, kh, dq);               // lane: 16 s_ds dQ[query 4 g + r][head dim j] of these keys
        float* dst = slot + (32 * qp + 16 * u + 4 * g) * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[16 * r] = fst ? dq[r] * c_dk : fmaf(dq[r], c_dk, dst[16 * r]);
      }
      if (NW == 8) att_lds_barrier();          // the slot's rows of this step are written before the other wave comes to them
    }
    first = false;
    if (qsplit) {      // partial dK / dV of this wave's queries -> slots 1 .. 3 as [wave][dk | dv][t][r][lane]; summed below
      float* red = dQw + Tqp * 16 + wave * 1024;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          red[(t * 4 + r) * 64 + lane] = dk[t][r] * c_dk;      // (every wave works on key block 0: the same scales)
          red[512 + (t * 4 + r) * 64 + lane] = dv[t][r] * c_dv;
        }
      continue;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = 32 * kp + 16 * t + 4 * g + r;
        if (kk < Tk) {
          job.dK[(size_t)(kbase + kk) * job.lddkv + 16 * h + j] = dk[t][r] * (0.25f * c_dk);
          job.dV[(size_t)(kbase + kk) * job.lddkv + 16 * h + j] = dv[t][r] * c_dv;
        }
      }
  }
  if (NW == 8 && !(wave < nkp)) {               // a wave without a key block (Tk <= 224) keeps the others' step barriers company
    for (int qi = 0; qi < nqp; ++qi) att_lds_barrier();
  }
  ATT_STAMP(3);
  __syncthreads();
  ATT_STAMP(4);
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
  for (int idx = threadIdx.x; idx < Tq * 4; idx += NT) {
    const int q = idx >> 2, c4 = idx & 3;
    float4 v = reinterpret_cast<const float4*>(dQw)[idx];
    for (int w = 1; w < nw; ++w) {
      const float4 a = reinterpret_cast<const float4*>(dQw + w * Tqp * 16)[idx];
      v = make_float4(v.x + a.x, v.y + a.y, v.z + a.z, v.w + a.w);
    }
    st4(job.dQ + (size_t)(qbase + q) * job.lddq + 16 * h + 4 * c4, make_float4(v.x * 0.25f, v.y * 0.25f, v.z * 0.25f, v.w * 0.25f));
  }
  ATT_STAMP(5);
}

template <bool DROP, int NW = 4>
__device__ __forceinline__ void attn_bwd_body(const AttnJob& job, int b, int h, char* lds, const DropCfg& drop) {
  const int wave = threadIdx.x >> 6;
  const int nkp = ((job.Tk + 31) & ~31) >> 5;
  const bool qsplit = NW == 4 && attn_bwd_qsplit(job.Tq, job.Tk);
  BwdV bv;
  ATT_STAMP(0);
  // V rows of the wave's first 32 keys: requested before the staging so that they arrive under it
  if (wave < nkp || qsplit) attn_bwd_load_v<DROP>(job, b, h, qsplit ? 0 : wave, bv);
  attn_bwd_stage<DROP, NW>(job, b, h, lds);
  ATT_STAMP(1);
  __syncthreads();
  ATT_STAMP(2);
  attn_bwd_compute<DROP, NW>(job, b, h, lds, drop, bv, true);
}

// ---- the small jobs of a four-job launch in ONE workgroup.  Per (clip, head) the three small jobs cost 13-20 k cycles each as workgroups
// of their own, more than half of it the staging round trip (scripts/exp/attn_stamps.py).  Here the rows of job n + 1 are requested
// (BwdPre: 25 registers - a job with Tq <= 32, Tk <= 128 has one pass of query rows and two of key rows) before job n's products and
// stored behind them: two of the three round trips disappear under products.  Measured (same-box A/B, B64 T128 L20): the four-job launches
// 42.0 -> 38.2 us each - the load latency hides, the split + LDS stores + barrier of a staging do not.
// (st: lane c4 = 0 / 1 / 2 of a row's four lanes holds the row's softmax maximum / 1 / row sum / query mask - one register instead of three:
//  at 256 registers a spilled prefetch register is stored behind an s_waitcnt vmcnt(0), i.e. the prefetch is waited for up front)
struct BwdPre { float4 q, d, o, k[2]; float st; uint2 mk; };
__host__ __device__ __forceinline__ bool attn_bwd_pre_ok(int Tq, int Tk) { return Tq <= 32 && Tk <= 128; }
template <bool DROP>
__device__ __forceinline__ void attn_bwd_pre_load(const AttnJob& job, int b, int h, BwdPre& s) {
  const int Tq = job.Tq, Tk = job.Tk;
  const int qbase = job.qrow0 + b * Tq, kbase = job.krow0 + b * Tk;
  const int row = threadIdx.x >> 2, c4 = threadIdx.x & 3;
  const int qr = min(row, Tq - 1);
  s.q = ld4(job.Q + (size_t)(qbase + qr) * job.ldq + 16 * h + 4 * c4);
  s.d = ld4(job.dO + (size_t)(qbase + qr) * job.lddo + 16 * h + 4 * c4);
  s.o = ld4(job.O + (size_t)(qbase + qr) * job.ldo + 16 * h + 4 * c4);
  s.k[0] = ld4(job.K + (size_t)(kbase + min(row, Tk - 1)) * job.ldkv + 16 * h + 4 * c4);
  s.k[1] = ld4(job.K + (size_t)(kbase + min(row + 64, Tk - 1)) * job.ldkv + 16 * h + 4 * c4);
  const int si = (b * Tq + qr) * 8 + h;
  const float* sp = c4 == 2 ? job.qmask + (qbase + qr) : job.stats + (c4 == 1 ? job.B * Tq * 8 + si : si);
  s.st = *sp;
  s.mk = make_uint2(0u, 0u);
  if (DROP) {
    const int nmk = ((Tq + 15) >> 4) * ((Tk + 15) >> 4) * 4;      // <= 2 x 8 x 4 keep words
    s.mk = reinterpret_cast<const uint2*>(job.dmask + (size_t)(b * 8 + h) * nmk * 8)[min((int)threadIdx.x, nmk - 1)];
  }
}
template <bool DROP>
__device__ __forceinline__ void attn_bwd_pre_store(const AttnJob& job, char* lds, const BwdPre& s) {
  const int Tq = job.Tq, Tk = job.Tk;
  const int Tqp = (Tq + 31) & ~31, Tkp = (Tk + 31) & ~31;
  const BwdLds L = bwd_lds(Tq, Tk, DROP);
  char* Qp = lds + L.qp; char* Dp = lds + L.dp; char* Kp = lds + L.kp;
  float* St = reinterpret_cast<float*>(lds + L.st);
  const int row = threadIdx.x >> 2, c4 = threadIdx.x & 3;
  const bool qok = row < Tq;
  float part = qok ? (s.d.x * s.o.x + s.d.y * s.o.y) + (s.d.z * s.o.z + s.d.w * s.o.w) : 0.f;
  part += dpp_xor_partner(part, 1);
  part += dpp_xor_partner(part, 2);
  float rmax = qok ? f4absmax(s.d) : 0.f;
  att_wave_max2_put(reinterpret_cast<float*>(lds + L.red), rmax, fabsf(part));
  rmax = fmaxf(rmax, dpp_xor_partner(rmax, 1));
  rmax = fmaxf(rmax, dpp_xor_partner(rmax, 2));
  float rinv;
#ifdef ATT_EXP_DOFIX
  const float rsc = 1024.0f; rinv = 1.0f / 1024.0f;
#else
  const float rsc = att_pow2_scale(rmax, rinv);          // the row's own scale (attn_bwd_stage)
#endif
  if (row < Tqp) {
    panel_store_h(Qp, row, c4, qok ? s.q : f4zero(), ATT_SX);
    panel_store_h(Dp, row, c4, qok ? s.d : f4zero(), rsc);
  }
  if (row < Tkp) panel_store_h(Kp, row, c4, row < Tk ? s.k[0] : f4zero(), ATT_SX);
  if (row + 64 < Tkp) panel_store_h(Kp, row + 64, c4, row + 64 < Tk ? s.k[1] : f4zero(), ATT_SX);
  const int sti = __builtin_bit_cast(int, s.st);
  const float s_si = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(sti, sti, 0x55, 0xF, 0xF, false));      // quad lane 1
  const float s_qm = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(sti, sti, 0xAA, 0xF, 0xF, false));      // quad lane 2
  if (c4 == 0 && row < Tqp) {      // (the six statistic rows of attn_bwd_stage)
    St[row] = qok ? s.st : 0.f;
    St[Tqp + row] = qok ? s_si : 0.f;
    St[2 * Tqp + row] = part;
    St[3 * Tqp + row] = ATT_NEGL;
    St[4 * Tqp + row] = (qok && s_qm != 0.f) ? 0.f : ATT_NEGL;
    St[6 * Tqp + row] = rinv;
    St[5 * Tqp + row] = -INFINITY;
  }
  if (DROP) {
    const int nmk = ((Tq + 15) >> 4) * ((Tk + 15) >> 4) * 4;
    if ((int)threadIdx.x < nmk) reinterpret_cast<uint2*>(lds + L.mk)[threadIdx.x] = s.mk;
  }
}
template <bool DROP>
__device__ __forceinline__ void attn_bwd_chain(const AttnJob& j1, const AttnJob& j2, const AttnJob& j3, int b, int h, char* lds, const DropCfg& drop) {
  const int wave = threadIdx.x >> 6;
  BwdV bv;
  BwdPre pre;
  {
    const int nkp = ((j1.Tk + 31) & ~31) >> 5;
    const bool qsplit = attn_bwd_qsplit(j1.Tq, j1.Tk);
    if (wave < nkp || qsplit) attn_bwd_load_v<DROP>(j1, b, h, qsplit ? 0 : wave, bv);
  }
  attn_bwd_stage<DROP>(j1, b, h, lds);
  __syncthreads();
  attn_bwd_pre_load<DROP>(j2, b, h, pre);
  attn_bwd_compute<DROP>(j1, b, h, lds, drop, bv, true);
  __syncthreads();                                   // every wave is through job 1's panels and dQ slots
  attn_bwd_pre_store<DROP>(j2, lds, pre);
  __syncthreads();
  attn_bwd_pre_load<DROP>(j3, b, h, pre);
  attn_bwd_compute<DROP>(j2, b, h, lds, drop, bv, false);
  __syncthreads();
  attn_bwd_pre_store<DROP>(j3, lds, pre);
  __syncthreads();
  attn_bwd_compute<DROP>(j3, b, h, lds, drop, bv, false);
}

// four jobs = two kinds of workgroup per (clip, head): kind 0 the largest job, kind 1 the chain of the other three
__global__ __launch_bounds__(256, 2) void attn_bwd_chain_kernel(AttnBatch batch, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int lid = xcd_logical_id();
  const int h = lid & 7; lid >>= 3;
  int kind = lid & 1, b = lid >> 1;
  {
    const int per = (int)(gridDim.x >> 6);           // (kind, clip) pairs per XCD: the chains of an XCD's clips first
    if ((gridDim.x & 63) == 0 && (per & 1) == 0) {
      const int cpx = per >> 1, x = lid / per, w = lid - x * per;
      // Order inside an XCD's run.  Round 4 put the chains of an XCD's clips first (the longer kind: ~45 k against ~30 k cycles) - but then
      // the two workgroups of a CU are always of one kind, start together and run their memory and compute phases in lockstep.  Round 5:
      // runs of FOUR (kind, clip) pairs alternate, so that the two slots of a CU (workgroups i and i + 32 of the XCD's dispatch order)
      // hold one chain and one large job: their phases interleave instead of coinciding (83.8 -> 79.0 us for the two launches, same-box A/B)
      if ((cpx & 3) == 0) { kind = 1 - ((w >> 2) & 1); b = x * cpx + (w & 3) + 4 * (w >> 3); }
      else { kind = 1 - w / cpx; b = x * cpx + (w - (w / cpx) * cpx); }
    }
  }
  if (b >= batch.j[0].B) return;   // block-uniform
  const bool dd = batch.j[0].drop_site >= 0 && drop.enabled;
  if (kind == 0) {
    if (dd) attn_bwd_body<true>(batch.j[0], b, h, lds, drop);
    else attn_bwd_body<false>(batch.j[0], b, h, lds, drop);
  } else {
    if (dd) attn_bwd_chain<true>(batch.j[1], batch.j[2], batch.j[3], b, h, lds, drop);
    else attn_bwd_chain<false>(batch.j[1], batch.j[2], batch.j[3], b, h, lds, drop);
  }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(AttnBatch batch, int njobs, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int lid = xcd_logical_id();
  const int h = lid & 7; lid >>= 3;
  // (job, clip) of this workgroup.  An XCD holds a contiguous run of logical ids = whole clips; inside the run the LARGEST job of
  // every clip goes first (launch_attn_bwd sorts the jobs by cost): the 64 resident slots of an XCD start on the 128 x 128
  // self-attention jobs together and the short jobs fill in behind them - with the jobs of a clip interleaved the four-job launches
  // ended on long jobs (44.5 -> 42.2 us each, A/B on one box).  Clip counts that do not give every XCD whole clips keep the
  // interleaved order
  int jb = lid % njobs, b = lid / njobs;
  {
    const int per = (int)(gridDim.x >> 6);      // (job, clip) pairs per XCD
    if ((gridDim.x & 63) == 0 && per % njobs == 0) {
      const int cpx = per / njobs, x = lid / per, w = lid - x * per;
      jb = w / cpx;
      b = x * cpx + (w - jb * cpx);
    }
  }
  const AttnJob& job = batch.j[jb];
  if (b >= job.B) return;   // block-uniform
  if (job.drop_site >= 0 && drop.enabled) attn_bwd_body<true>(job, b, h, lds, drop);
  else attn_bwd_body<false>(job, b, h, lds, drop);
}

// A launch whose LARGEST job has more than 128 queries and keys (dual attention at T = 256): that job on eight waves
// (attn_bwd_compute NW = 8); the other jobs of the launch as before on the first four waves of their workgroups - the other four
// end at once (s_barrier waits for the surviving waves of a workgroup only).  Workgroup -> (job, clip) as in attn_bwd_kernel.
__global__ __launch_bounds__(512) void attn_bwd_big_kernel(AttnBatch batch, int njobs, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int lid = xcd_logical_id();
  const int h = lid & 7; lid >>= 3;
  int jb = lid % njobs, b = lid / njobs;
  {
    const int per = (int)(gridDim.x >> 6);      // (job, clip) pairs per XCD
    if ((gridDim.x & 63) == 0 && per % njobs == 0) {
      const int cpx = per / njobs, x = lid / per, w = lid - x * per;
      jb = w / cpx;
      b = x * cpx + (w - jb * cpx);
    }
  }
  const AttnJob& job = batch.j[jb];
  if (b >= job.B) return;   // block-uniform
  const bool dd = job.drop_site >= 0 && drop.enabled;
  if (jb == 0) {
    if (dd) attn_bwd_body<true, 8>(job, b, h, lds, drop);
    else attn_bwd_body<false, 8>(job, b, h, lds, drop);
    return;
  }
  if (threadIdx.x >= 256) return;               // (wave-uniform)
  if (dd) attn_bwd_body<true>(job, b, h, lds, drop);
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
    HUAL_REQUIRE(!j.dmask || (reinterpret_cast<uintptr_t>(j.dmask) & 7) == 0, "attn: keep words must be 8-byte aligned");
    if (bwd) {
      HUAL_REQUIRE(j.dO && j.dQ && j.dK && j.dV && j.O && j.stats && (j.lddo % 4) == 0 && (j.ldo % 4) == 0 && (j.lddq % 4) == 0,
                   "attn bwd: needs dO, dQ, dK, dV, the forward output O and the forward softmax statistics");
      HUAL_REQUIRE(j.Tq <= 256, "attn bwd: Tq <= 256");
      HUAL_REQUIRE(!(j.drop_site >= 0 && drop.enabled) || j.dmask, "attn bwd: dropout needs the keep bytes of the forward");
    } else {
      HUAL_REQUIRE(j.O != nullptr, "attn fwd: null output");
      HUAL_REQUIRE(j.Tq <= 256, "attn fwd: Tq <= 256 (a unit code holds 16 query tiles per job)");
    }
    maxTq = j.Tq > maxTq ? j.Tq : maxTq;
    maxTk = j.Tk > maxTk ? j.Tk : maxTk;
    maxB = j.B > maxB ? j.B : maxB;
  }
  return 0;
}

static bool getenv_flag(const char* name) {
  const char* v = getenv(name);
  return v != nullptr && atoi(v) != 0;
}
static int nkt_pad_host(int Tk) {
  const int n = cdiv(Tk, 16);
  return n <= 2 ? 2 : n <= 4 ? 4 : n <= 8 ? 8 : 16;
}
// algorithmic HBM bytes of an attention job (forward): Q, K, V rows of a head read once, O written, softmax statistics and keep words
static double attn_job_bytes(const AttnJob& j, bool drop) {
  return (double)j.B * 8.0 * ((double)j.Tq * 64.0 * 2.0 + (double)j.Tk * 128.0 + (double)j.Tq * 8.0) +
         (drop && j.dmask ? (double)attn_keep_bytes(j.B, j.Tq, j.Tk) : 0.0);
}

int launch_attn_fwd(const AttnJob* jobs, int n, const DropCfg& drop, hipStream_t s) {
  int maxTq, maxTk, maxB;
  int rc = check_jobs(jobs, n, false, drop, maxTq, maxTk, maxB);
  if (rc) return rc;
  AttnBatch b;
  ::memset((void*)&b, 0, sizeof(b));
  for (int i = 0; i < n; ++i) {
    b.j[i] = jobs[i];
    HUAL_REQUIRE(jobs[i].B == jobs[0].B, "attn fwd: the jobs of a launch share the clip count");
  }
  // units = (job, 16-query tile), sorted by cost = padded key tiles (+ the fixed part of a unit: query split, statistics, stores)
  struct U { int cost, code; };
  std::vector<U> us;
  int RT = 0;
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const int nktp = nkt_pad_host(jobs[i].Tk);
    RT += 16 * nktp;
    for (int qt = 0; qt < cdiv(jobs[i].Tq, 16); ++qt) us.push_back(U{2 * nktp + 3, (i << 4) | qt});
    flops += 4.0 * jobs[i].B * 8.0 * jobs[i].Tq * jobs[i].Tk * 16.0;   // QK^T + PV
    bytes += attn_job_bytes(jobs[i], jobs[i].drop_site >= 0 && drop.enabled);
  }
  HUAL_REQUIRE((int)us.size() <= HUAL_MAX_ATTN_UNITS, "attn fwd: too many query tiles");
  std::stable_sort(us.begin(), us.end(), [](const U& a, const U& c) { return a.cost > c.cost; });
  AttnUnits un;
  ::memset((void*)&un, 0, sizeof(un));
  un.n = (int)us.size();
  for (size_t k = 0; k < us.size(); ++k) un.u[k] = (uint8_t)us[k].code;
  const size_t lds = (size_t)RT * 128 + (size_t)2 * RT * sizeof(float);
  HUAL_REQUIRE(lds <= 160 * 1024, "attn fwd: LDS footprint");
  const int nkt = cdiv(maxTk, 16);
  // eight waves per workgroup when there are units for them (four waves per SIMD at two workgroups per CU)
  const bool wide = us.size() >= 8;
  dim3 grid(xcd_round8(maxB) * 8), block(wide ? 512 : 256);
#define HUAL_ATTN_FWD_LAUNCH(NK)                                                                                         \
  do {                                                                                                                   \
    if (wide) { HUAL_DYN_LDS((attn_fwd_kernel<NK, 512>), 160 * 1024); HUAL_LAUNCH(flops, bytes, (attn_fwd_kernel<NK, 512>), grid, block, lds, s, b, n, un, drop); } \
    else { HUAL_DYN_LDS((attn_fwd_kernel<NK, 256>), 160 * 1024); HUAL_LAUNCH(flops, bytes, (attn_fwd_kernel<NK, 256>), grid, block, lds, s, b, n, un, drop); }      \
  } while (0)
  if (nkt <= 2) HUAL_ATTN_FWD_LAUNCH(2);
  else if (nkt <= 4) HUAL_ATTN_FWD_LAUNCH(4);
  else if (nkt <= 8) HUAL_ATTN_FWD_LAUNCH(8);
  else HUAL_ATTN_FWD_LAUNCH(16);
#undef HUAL_ATTN_FWD_LAUNCH
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
  double flops = 0.0, bytes = 0.0;
  // largest job first (attn_bwd_kernel's workgroup order)
  int order[HUAL_MAX_ATTN_JOBS];
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order, order + n, [&](int x, int y) { return (long long)jobs[x].Tq * jobs[x].Tk > (long long)jobs[y].Tq * jobs[y].Tk; });
  for (int k = 0; k < n; ++k) b.j[k] = jobs[order[k]];
  for (int i = 0; i < n; ++i) {
    const bool dd = jobs[i].drop_site >= 0 && drop.enabled;
    const size_t need = (size_t)bwd_lds(jobs[i].Tq, jobs[i].Tk, dd).total;
    lds = need > lds ? need : lds;
    flops += 2.0 * jobs[i].B * 8.0 * jobs[i].Tq * jobs[i].Tk * 16.0;
    // backward: the forward's reads + dO, dQ rows of the queries and dK, dV rows of the keys
    bytes += attn_job_bytes(jobs[i], dd) + (double)jobs[i].B * 8.0 * ((double)jobs[i].Tq * 128.0 + (double)jobs[i].Tk * 128.0);
  }
  HUAL_REQUIRE(lds <= 160 * 1024, "attn bwd: LDS footprint");
  // four jobs whose two smallest have <= 32 queries and <= 128 keys (the dual attention at T <= 128): the three small jobs share a
  // workgroup that prefetches across them (attn_bwd_chain); same dropout setting and clip count on all four
  bool chain = n == 4 && attn_bwd_pre_ok(b.j[2].Tq, b.j[2].Tk) && attn_bwd_pre_ok(b.j[3].Tq, b.j[3].Tk);
  for (int k = 1; k < n && chain; ++k)
    chain = b.j[k].B == b.j[0].B && ((b.j[k].drop_site >= 0) == (b.j[0].drop_site >= 0));
  if (chain) {
    HUAL_DYN_LDS(attn_bwd_chain_kernel, 160 * 1024);
    HUAL_LAUNCH(4.0 * flops, bytes, attn_bwd_chain_kernel, dim3(maxB * 8 * 2), dim3(256), lds, s, b, drop);
    HUAL_CHECK_HIP(hipGetLastError());
    return 0;
  }
  // the largest job beyond 128 queries AND keys (its workgroup fills a CU's LDS): eight waves for it (attn_bwd_big_kernel)
  if (b.j[0].Tq > 128 && b.j[0].Tk > 128 && !getenv_flag("HUAL_ATTN_NO_BIG")) {
    size_t lds8 = (size_t)bwd_lds(b.j[0].Tq, b.j[0].Tk, b.j[0].drop_site >= 0 && drop.enabled, 8).total;
    for (int k = 1; k < n; ++k) {
      const size_t need = (size_t)bwd_lds(b.j[k].Tq, b.j[k].Tk, b.j[k].drop_site >= 0 && drop.enabled).total;
      lds8 = need > lds8 ? need : lds8;
    }
    bool ok = lds8 <= 160 * 1024;
    for (int k = 1; k < n; ++k) ok = ok && !(b.j[k].Tq > 128 && b.j[k].Tk > 128) && b.j[k].B == b.j[0].B;      // one large job per launch
    if (ok) {
      HUAL_DYN_LDS(attn_bwd_big_kernel, 160 * 1024);
      HUAL_LAUNCH(4.0 * flops, bytes, attn_bwd_big_kernel, dim3(maxB * 8 * n), dim3(512), lds8, s, b, n, drop);
      HUAL_CHECK_HIP(hipGetLastError());
      return 0;
    }
  }
  HUAL_DYN_LDS(attn_bwd_kernel, 160 * 1024);
  dim3 grid(maxB * 8 * n), block(256);
  // algorithmic work of the backward: FOUR products (dP = dO.V^T, dV = P^T.dO, dK = dS^T.Q, dQ = dS.K) = 8.B.H.Tq.Tk.16; the
  // recomputation of S = Q.K^T is the kernel's choice (it saves storing P) and is not counted
  HUAL_LAUNCH(4.0 * flops, bytes, attn_bwd_kernel, grid, block, lds, s, b, n, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
