// Attention kernels (see attn.h) on v_mfma_f32_16x16x4_f32, dh = 16.
//
// Layout trick (no LDS, no transposes): a wave owns 16 queries and ALL keys of one (clip, head).
// It computes S^T = K.Q^T per 16-key tile, so lane (j = lane&15, g = lane>>4) register r holds
// S[query j][key 16*kt + 4g + r].  A query's row is then spread over 4 lanes x (4*nkt) registers:
// row max / row sum = in-lane reduction + two __shfl_xor (16, 32).  The same registers are, unchanged,
// the A operand (A[i = query j][k = g], k-step r) of the P.V product whose B operand is
// V[key 16kt+4g+r][dh = j] - the accumulator of one MFMA feeds the next without touching LDS.
// The backward dK/dV kernel uses the mirrored orientation (S = Q.K^T, lane holds 4 queries x 1 key).
#include "attn.h"
#include "philox.h"
#include "prof.h"
#include <string.h>

using namespace hual;

namespace hual {
void attn_job_init(AttnJob& j) {
  ::memset((void*)&j, 0, sizeof(j));
  j.drop_site = -1;
  j.dmask = nullptr;
}
}  // namespace hual

__device__ __forceinline__ f32x4 mfma16_(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 dot16(const float4& a, const float4& b) {   // K = 16 contraction, 4 k-steps
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = mfma16_(a.x, b.x, acc);
  acc = mfma16_(a.y, b.y, acc);
  acc = mfma16_(a.z, b.z, acc);
  acc = mfma16_(a.w, b.w, acc);
  return acc;
}

// XCD-aware block order (cdna_hip_programming.md T1, bijective form).  The dispatcher deals consecutive linear block ids
// round-robin over the 8 XCDs, each with a private L2; the 16+ blocks of one clip (8 heads x query tiles x halves) all read
// that clip's Q/K/V/dO rows, so in launch order every XCD fetched every clip (PMC: 222 MB of HBM traffic per backward
// launch against ~50 MB of operands).  The remap hands each XCD a contiguous run of logical ids, i.e. whole clips.
__device__ __forceinline__ int xcd_logical_id() {
  const int nwg = gridDim.x * gridDim.y * gridDim.z;
  const int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

#define ATT_SCALE 0.25f   // 1/sqrt(head_size=16)   layers.py:82

// ---- LDS staging ---------------------------------------------------------------------------------------
// A block = 4 waves = 4 consecutive 16-row tiles of ONE (clip, head).  The per-head K/V (or Q/dO) panels of the
// clip ([T,16] floats each, <= 16 KB at T = 256) are staged once per block with 16-byte loads and then read many
// times from LDS: rows as float4 (A operand / transposed B operand), columns as scalars (B operand of P.V).
__device__ __forceinline__ void stage_panel(float* dst, const float* src, int ld, int rows, int rows_padded) {
  // dst[row][16] <- src[row*ld + 0..15], zero beyond `rows`
  for (int idx = threadIdx.x; idx < rows_padded * 4; idx += 256) {
    const int row = idx >> 2, c4 = idx & 3;
    float4 v = f4zero();
    if (row < rows) v = ld4(src + (size_t)row * ld + 4 * c4);
    *reinterpret_cast<float4*>(dst + row * 16 + 4 * c4) = v;
  }
}

template <int MAXKT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnBatch batch, DropCfg drop) {
  extern __shared__ float lds[];
  int lid = xcd_logical_id();
  const int bx = lid % (int)gridDim.x; lid /= (int)gridDim.x;
  const int h = lid & 7; lid >>= 3;
  const int njobs = (int)gridDim.z;
  const AttnJob& job = batch.j[lid % njobs];       // clip-major: an XCD gets whole clips with ALL their jobs (balanced)
  const int b = lid / njobs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Tq = job.Tq, Tk = job.Tk;
  if (b >= job.B || bx * 64 >= Tq) return;   // block-uniform
  const int nkt = (Tk + 15) >> 4, Tkp = nkt * 16;
  const int qbase = job.qrow0 + b * Tq, kbase = job.krow0 + b * Tk;
  float* Ks = lds;
  float* Vs = lds + Tkp * 16;
  float* Ms = Vs + Tkp * 16;
  stage_panel(Ks, job.K + (size_t)kbase * job.ldkv + 16 * h, job.ldkv, Tk, Tkp);
  stage_panel(Vs, job.V + (size_t)kbase * job.ldkv + 16 * h, job.ldkv, Tk, Tkp);
  for (int k = threadIdx.x; k < Tkp; k += 256) Ms[k] = k < Tk ? job.kmask[kbase + k] : 0.f;
  __syncthreads();
  const int qt = bx * 4 + wave;
  if (qt * 16 >= Tq) return;   // wave-uniform, after the only barrier
  const int q0 = qt * 16;
  const int qrow = qbase + min(q0 + j, Tq - 1);
  const float4 qb = ld4(job.Q + (size_t)qrow * job.ldq + 16 * h + 4 * g);
  const float mq = job.qmask[qrow];
  f32x4 s[MAXKT];
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt) {
    if (kt < nkt) {
      const float4 ka = *reinterpret_cast<const float4*>(Ks + (kt * 16 + j) * 16 + 4 * g);
      f32x4 a = dot16(ka, qb);
      const float4 mk4 = *reinterpret_cast<const float4*>(Ms + kt * 16 + 4 * g);
      const float mk[4] = {mk4.x, mk4.y, mk4.z, mk4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        const float v = key < Tk ? a[r] * ATT_SCALE + (1.0f - mq * mk[r]) * HUAL_MASK_VALUE : -INFINITY;   // layers.py:82-84
        a[r] = v;
        mx = fmaxf(mx, v);
      }
      s[kt] = a;
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt)
    if (kt < nkt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[kt][r] - mx);
        s[kt][r] = e;
        sum += e;
      }
    }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
  if (job.stats && g == 0 && (q0 + j) < Tq) {
    const int si = (b * Tq + q0 + j) * 8 + h;
    job.stats[si] = mx;
    job.stats[job.B * Tq * 8 + si] = inv;
  }
  const bool dodrop = job.drop_site >= 0 && drop.enabled;
  const uint32_t drow = (job.drop_row0 + (uint32_t)qrow) * 8u + (uint32_t)h;
  uint8_t* mrow = (job.dmask && dodrop && (q0 + j) < Tq) ? job.dmask + ((size_t)(b * Tq + q0 + j) * 8 + h) * job.ldm : nullptr;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt) {
    if (kt < nkt) {
      f32x4 p = s[kt];
      float4 m = make_float4(inv, inv, inv, inv);
      if (dodrop) {
        const uint32_t bits = drop_bits4(drop, (uint32_t)job.drop_site, drow, (uint32_t)(kt * 4 + g));
        if (mrow) mrow[kt * 4 + g] = (uint8_t)bits;
        m = mask_from_bits4(bits, inv * drop.scale);
      }
      p[0] *= m.x; p[1] *= m.y; p[2] *= m.z; p[3] *= m.w;
#pragma unroll
      for (int r = 0; r < 4; ++r) o = mfma16_(p[r], Vs[(kt * 16 + 4 * g + r) * 16 + j], o);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + 4 * g + r;
    if (q < Tq) job.O[(size_t)(qbase + q) * job.ldo + 16 * h + j] = o[r];
  }
}

// ---- backward -------------------------------------------------------------------------------------------
// The forward leaves the softmax statistics (row max, 1/rowsum) per (query, head); delta = sum_k P~.dP~ = dO . O
// (row of the head's 16 output columns), so neither half needs a pass over the keys before its main loop and the
// two halves are independent: ONE launch, blockIdx.z = 2*job + half.
//   half 0 (dQ):      a wave owns 16 queries and sweeps the key tiles      (S^T orientation, like the forward)
//   half 1 (dK, dV):  a wave owns 16 keys and sweeps the query tiles       (S orientation)
__device__ __forceinline__ void attn_bwd_dq_part(const AttnJob& job, const DropCfg& drop, float* lds, int bx, int b, int h) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Tq = job.Tq, Tk = job.Tk;
  if (bx * 64 >= Tq) return;
  const int nkt = (Tk + 15) >> 4, Tkp = nkt * 16;
  const int qbase = job.qrow0 + b * Tq, kbase = job.krow0 + b * Tk;
  float* Ks = lds;
  float* Vs = lds + Tkp * 16;
  float* Ms = Vs + Tkp * 16;
  stage_panel(Ks, job.K + (size_t)kbase * job.ldkv + 16 * h, job.ldkv, Tk, Tkp);
  stage_panel(Vs, job.V + (size_t)kbase * job.ldkv + 16 * h, job.ldkv, Tk, Tkp);
  for (int k = threadIdx.x; k < Tkp; k += 256) Ms[k] = k < Tk ? job.kmask[kbase + k] : 0.f;
  __syncthreads();
  const int qt = bx * 4 + wave;
  if (qt * 16 >= Tq) return;
  const int q0 = qt * 16;
  const int ql = b * Tq + min(q0 + j, Tq - 1);       // job-local query index
  const int qrow = qbase + min(q0 + j, Tq - 1);
  const float4 qb = ld4(job.Q + (size_t)qrow * job.ldq + 16 * h + 4 * g);
  const float4 dob = ld4(job.dO + (size_t)qrow * job.lddo + 16 * h + 4 * g);
  const float4 ob = ld4(job.O + (size_t)qrow * job.ldo + 16 * h + 4 * g);
  const float mq = job.qmask[qrow];
  const int stat_n = job.B * Tq * 8;
  const float mx = job.stats[ql * 8 + h], inv = job.stats[stat_n + ql * 8 + h];
  float delta = (dob.x * ob.x + dob.y * ob.y) + (dob.z * ob.z + dob.w * ob.w);
  delta += __shfl_xor(delta, 16);
  delta += __shfl_xor(delta, 32);
  const bool dodrop = job.drop_site >= 0 && drop.enabled;
  const uint32_t drow = (job.drop_row0 + (uint32_t)qrow) * 8u + (uint32_t)h;
  const uint8_t* mrow = job.dmask ? job.dmask + ((size_t)ql * 8 + h) * job.ldm : nullptr;
  f32x4 dq = {0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt < nkt; ++kt) {
    const float4 ka = *reinterpret_cast<const float4*>(Ks + (kt * 16 + j) * 16 + 4 * g);
    const float4 va = *reinterpret_cast<const float4*>(Vs + (kt * 16 + j) * 16 + 4 * g);
    f32x4 a = dot16(ka, qb);          // S^T[key 4g+r][query j]
    f32x4 dp = dot16(va, dob);
    const float4 mk4 = *reinterpret_cast<const float4*>(Ms + kt * 16 + 4 * g);
    const float mk[4] = {mk4.x, mk4.y, mk4.z, mk4.w};
    float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
    if (dodrop)
      m = mrow ? mask_from_bits4(mrow[kt * 4 + g], drop.scale)
               : drop_mask4(drop, (uint32_t)job.drop_site, drow, (uint32_t)(kt * 4 + g));
    const float mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      float p = 0.f;
      if (key < Tk) p = __expf(a[r] * ATT_SCALE + (1.0f - mq * mk[r]) * HUAL_MASK_VALUE - mx) * inv;
      const float ds = p * (dp[r] * mm[r] - delta) * ATT_SCALE;
      dq = mfma16_(ds, Ks[(kt * 16 + 4 * g + r) * 16 + j], dq);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + 4 * g + r;
    if (q < Tq) job.dQ[(size_t)(qbase + q) * job.lddq + 16 * h + j] = dq[r];
  }
}

__device__ __forceinline__ void attn_bwd_dkv_part(const AttnJob& job, const DropCfg& drop, float* lds, int bx, int b, int h) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Tq = job.Tq, Tk = job.Tk;
  if (bx * 64 >= Tk) return;
  const int nqt = (Tq + 15) >> 4, Tqp = nqt * 16;
  const int qbase = job.qrow0 + b * Tq, kbase = job.krow0 + b * Tk;
  float* Qs = lds;
  float* Ds = lds + Tqp * 16;
  float* St = Ds + Tqp * 16;        // [4][Tqp]: row max, 1/rowsum, delta, query mask
  uint32_t* Mb = reinterpret_cast<uint32_t*>(St + 4 * Tqp);     // [Tqp][4] words = the keep bytes of this block's 64 keys
  stage_panel(Qs, job.Q + (size_t)qbase * job.ldq + 16 * h, job.ldq, Tq, Tqp);
  const bool dodrop = job.drop_site >= 0 && drop.enabled;
  const bool usemask = dodrop && job.dmask != nullptr;
  const int stat_n = job.B * Tq * 8;
  // dO panel + delta = dO . O: 4 consecutive threads hold the 16 columns of one query
  for (int idx = threadIdx.x; idx < Tqp * 4; idx += 256) {
    const int q = idx >> 2, c4 = idx & 3;
    float4 v = f4zero();
    float part = 0.f;
    uint32_t w = 0;
    if (q < Tq) {
      v = ld4(job.dO + (size_t)(qbase + q) * job.lddo + 16 * h + 4 * c4);
      const float4 o = ld4(job.O + (size_t)(qbase + q) * job.ldo + 16 * h + 4 * c4);
      part = (v.x * o.x + v.y * o.y) + (v.z * o.z + v.w * o.w);
      const int byte0 = bx * 16 + 4 * c4;
      if (usemask && byte0 < job.ldm)
        w = *reinterpret_cast<const uint32_t*>(job.dmask + ((size_t)(b * Tq + q) * 8 + h) * job.ldm + byte0);
    }
    *reinterpret_cast<float4*>(Ds + q * 16 + 4 * c4) = v;
    Mb[idx] = w;
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    if (c4 == 0) {
      const bool ok = q < Tq;
      const int si = (b * Tq + (ok ? q : 0)) * 8 + h;
      St[q] = ok ? job.stats[si] : 0.f;
      St[Tqp + q] = ok ? job.stats[stat_n + si] : 0.f;          // 1/rowsum = 0 for padding queries -> p = 0
      St[2 * Tqp + q] = part;
      St[3 * Tqp + q] = ok ? job.qmask[qbase + q] : 0.f;
    }
  }
  __syncthreads();
  const int kt = bx * 4 + wave;
  if (kt * 16 >= Tk) return;
  const int k0 = kt * 16;
  const int key = k0 + j;
  const bool keyok = key < Tk;
  const int krow = kbase + min(key, Tk - 1);
  const float4 kb = ld4(job.K + (size_t)krow * job.ldkv + 16 * h + 4 * g);
  const float4 vb = ld4(job.V + (size_t)krow * job.ldkv + 16 * h + 4 * g);
  const float mk = job.kmask[krow];
  f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
  for (int qt = 0; qt < nqt; ++qt) {
    const int q0 = qt * 16;
    const float4 qa = *reinterpret_cast<const float4*>(Qs + (q0 + j) * 16 + 4 * g);
    const float4 doa = *reinterpret_cast<const float4*>(Ds + (q0 + j) * 16 + 4 * g);
    f32x4 s = dot16(qa, kb);     // lane: S[query q0+4g+r][key k0+j]
    f32x4 dp = dot16(doa, vb);
    const float4 m4 = *reinterpret_cast<const float4*>(St + q0 + 4 * g);
    const float4 l4 = *reinterpret_cast<const float4*>(St + Tqp + q0 + 4 * g);
    const float4 d4 = *reinterpret_cast<const float4*>(St + 2 * Tqp + q0 + 4 * g);
    const float4 qm4 = *reinterpret_cast<const float4*>(St + 3 * Tqp + q0 + 4 * g);
    const float mxv[4] = {m4.x, m4.y, m4.z, m4.w}, liv[4] = {l4.x, l4.y, l4.z, l4.w};
    const float dlv[4] = {d4.x, d4.y, d4.z, d4.w}, qmv[4] = {qm4.x, qm4.y, qm4.z, qm4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + 4 * g + r;
      float p = 0.f;
      if (keyok) {
        const float v = s[r] * ATT_SCALE + (1.0f - qmv[r] * mk) * HUAL_MASK_VALUE;
        p = __expf(v - mxv[r]) * liv[r];
      }
      float m = 1.0f;
      if (usemask) {
        const uint32_t byte = (Mb[q * 4 + wave] >> (8 * (j >> 2))) & 0xFFu;
        m = ((byte >> (j & 3)) & 1u) ? drop.scale : 0.f;
      } else if (dodrop) {
        const uint32_t drow = (job.drop_row0 + (uint32_t)(qbase + min(q, Tq - 1))) * 8u + (uint32_t)h;
        float4 mm = drop_mask4(drop, (uint32_t)job.drop_site, drow, (uint32_t)(key >> 2));
        const int c = key & 3;
        m = c == 0 ? mm.x : (c == 1 ? mm.y : (c == 2 ? mm.z : mm.w));
      }
      const float ds = p * (dp[r] * m - dlv[r]) * ATT_SCALE;
      s[r] = p * m;     // dropped probability (A operand of dV)
      dp[r] = ds;       // A operand of dK
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dv = mfma16_(s[r], Ds[(q0 + 4 * g + r) * 16 + j], dv);
      dk = mfma16_(dp[r], Qs[(q0 + 4 * g + r) * 16 + j], dk);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kk = k0 + 4 * g + r;
    if (kk < Tk) {
      job.dK[(size_t)(kbase + kk) * job.lddkv + 16 * h + j] = dk[r];
      job.dV[(size_t)(kbase + kk) * job.lddkv + 16 * h + j] = dv[r];
    }
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnBatch batch, DropCfg drop) {
  extern __shared__ float lds[];
  // logical order: tile, half (dQ | dK/dV), head, job fastest, clip slowest - all blocks of one clip are neighbours
  int lid = xcd_logical_id();
  const int bx = lid % (int)gridDim.x; lid /= (int)gridDim.x;
  const int half = lid & 1; lid >>= 1;
  const int h = lid & 7; lid >>= 3;
  const int njobs = (int)gridDim.z >> 1;
  const AttnJob& job = batch.j[lid % njobs];       // clip-major: an XCD gets whole clips with ALL their jobs (balanced)
  const int b = lid / njobs;
  if (b >= job.B) return;   // block-uniform
  if (half) attn_bwd_dkv_part(job, drop, lds, bx, b, h);
  else attn_bwd_dq_part(job, drop, lds, bx, b, h);
}

namespace hual {

static int check_jobs(const AttnJob* jobs, int n, bool bwd, int& maxTq, int& maxTk, int& maxB) {
  HUAL_REQUIRE(n >= 1 && n <= HUAL_MAX_ATTN_JOBS, "attn: job count");
  maxTq = maxTk = maxB = 0;
  for (int i = 0; i < n; ++i) {
    const AttnJob& j = jobs[i];
    HUAL_REQUIRE(j.Q && j.K && j.V && j.qmask && j.kmask, "attn: null operand");
    HUAL_REQUIRE(j.B > 0 && j.Tq > 0 && j.Tk > 0 && j.Tk <= 256, "attn: need 0 < Tk <= 256, Tq > 0");
    HUAL_REQUIRE((j.ldq % 4) == 0 && (j.ldkv % 4) == 0, "attn: leading dims must be multiples of 4");
    HUAL_REQUIRE(!j.dmask || ((j.ldm % 4) == 0 && j.ldm >= 4 * cdiv(j.Tk, 16) && (reinterpret_cast<uintptr_t>(j.dmask) & 3) == 0),
                 "attn: dropout keep-byte rows need ldm >= 4*ceil(Tk/16), multiple of 4");
    if (bwd) HUAL_REQUIRE(j.dO && j.dQ && j.dK && j.dV && j.O && j.stats && (j.lddo % 4) == 0 && (j.ldo % 4) == 0,
                          "attn bwd: needs dO, dQ, dK, dV, the forward output O and the forward softmax statistics");
    else HUAL_REQUIRE(j.O != nullptr, "attn fwd: null output");
    maxTq = j.Tq > maxTq ? j.Tq : maxTq;
    maxTk = j.Tk > maxTk ? j.Tk : maxTk;
    maxB = j.B > maxB ? j.B : maxB;
  }
  return 0;
}

int launch_attn_fwd(const AttnJob* jobs, int n, const DropCfg& drop, hipStream_t s) {
  int maxTq, maxTk, maxB;
  int rc = check_jobs(jobs, n, false, maxTq, maxTk, maxB);
  if (rc) return rc;
  AttnBatch b;
  for (int i = 0; i < n; ++i) b.j[i] = jobs[i];
  dim3 grid(cdiv(cdiv(maxTq, 16), 4), maxB * 8, n), block(256);
  const int nkt = cdiv(maxTk, 16);
  double flops = 0.0;
  for (int i = 0; i < n; ++i) flops += 4.0 * jobs[i].B * 8.0 * jobs[i].Tq * jobs[i].Tk * 16.0;   // QK^T + PV
  const size_t lds = (size_t)nkt * 16 * 33 * sizeof(float);
  if (nkt <= 2) HUAL_LAUNCH(flops, 0.0, attn_fwd_kernel<2>, grid, block, lds, s, b, drop);
  else if (nkt <= 4) HUAL_LAUNCH(flops, 0.0, attn_fwd_kernel<4>, grid, block, lds, s, b, drop);
  else if (nkt <= 8) HUAL_LAUNCH(flops, 0.0, attn_fwd_kernel<8>, grid, block, lds, s, b, drop);
  else HUAL_LAUNCH(flops, 0.0, attn_fwd_kernel<16>, grid, block, lds, s, b, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_attn_bwd(const AttnJob* jobs, int n, const DropCfg& drop, hipStream_t s) {
  int maxTq, maxTk, maxB;
  int rc = check_jobs(jobs, n, true, maxTq, maxTk, maxB);
  if (rc) return rc;
  AttnBatch b;
  for (int i = 0; i < n; ++i) b.j[i] = jobs[i];
  const int nqt = cdiv(maxTq, 16), nkt = cdiv(maxTk, 16);
  dim3 grid(cdiv(nqt > nkt ? nqt : nkt, 4), maxB * 8, 2 * n), block(256);
  double flops = 0.0;
  for (int i = 0; i < n; ++i) flops += 2.0 * jobs[i].B * 8.0 * jobs[i].Tq * jobs[i].Tk * 16.0;
  // dQ half: S, dP, dQ ; dK/dV half: S, dP, dK, dV
  const size_t lds_dq = (size_t)nkt * 16 * 33 * sizeof(float), lds_dkv = (size_t)nqt * 16 * 40 * sizeof(float);
  HUAL_LAUNCH(7.0 * flops, 0.0, attn_bwd_kernel, grid, block, lds_dq > lds_dkv ? lds_dq : lds_dkv, s, b, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
