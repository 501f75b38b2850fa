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

#define ATT_SCALE 0.25f   // 1/sqrt(head_size=16)   layers.py:82

// scores -> normalised probabilities in place; keys >= Tk get probability 0.  Returns nothing; p[] holds softmax.
template <int MAXKT>
__device__ __forceinline__ void softmax_rows(f32x4 (&s)[MAXKT], int nkt, int Tk, int g, float mq, const float* kmask_clip) {
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt) {
    if (kt < nkt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        float v;
        if (key < Tk) {
          const float mk = kmask_clip[key];
          v = s[kt][r] * ATT_SCALE + (1.0f - mq * mk) * HUAL_MASK_VALUE;   // layers.py:82-84
        } else {
          v = -INFINITY;
        }
        s[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt) {
    if (kt < nkt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = __expf(s[kt][r] - mx);
        s[kt][r] = e;
        sum += e;
      }
    }
  }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt)
    if (kt < nkt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kt][r] *= inv;
    }
}

template <int MAXKT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnBatch batch, DropCfg drop) {
  const AttnJob& job = batch.j[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Tq = job.Tq, Tk = job.Tk;
  const int qt = blockIdx.x * 4 + wave;
  const int b = blockIdx.y >> 3, h = blockIdx.y & 7;
  if (b >= job.B || qt * 16 >= Tq) return;   // wave-uniform
  const int q0 = qt * 16;
  const int nkt = (Tk + 15) >> 4;
  const int qbase = job.qrow0 + b * Tq, kbase = job.krow0 + b * Tk;
  const int qrow = qbase + min(q0 + j, Tq - 1);
  const float4 qb = ld4(job.Q + (size_t)qrow * job.ldq + 16 * h + 4 * g);
  const float mq = job.qmask[qrow];
  f32x4 s[MAXKT];
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt) {
    if (kt < nkt) {
      const int krow = kbase + min(kt * 16 + j, Tk - 1);
      const float4 ka = ld4(job.K + (size_t)krow * job.ldkv + 16 * h + 4 * g);
      s[kt] = dot16(ka, qb);
    }
  }
  softmax_rows<MAXKT>(s, nkt, Tk, g, mq, job.kmask + kbase);
  const bool dodrop = job.drop_site >= 0 && drop.enabled;
  const uint32_t drow = (job.drop_row0 + (uint32_t)qrow) * 8u + (uint32_t)h;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt) {
    if (kt < nkt) {
      f32x4 p = s[kt];
      if (dodrop) {
        float4 m = drop_mask4(drop, (uint32_t)job.drop_site, drow, (uint32_t)(kt * 4 + g));
        p[0] *= m.x; p[1] *= m.y; p[2] *= m.z; p[3] *= m.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int krow = kbase + min(kt * 16 + 4 * g + r, Tk - 1);
        const float vb = job.V[(size_t)krow * job.ldkv + 16 * h + j];
        o = mfma16_(p[r], vb, o);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + 4 * g + r;
    if (q < Tq) job.O[(size_t)(qbase + q) * job.ldo + 16 * h + j] = o[r];
  }
}

// dQ: same decomposition as the forward; also leaves the softmax statistics (row max, 1/rowsum) and
// delta = sum_k P*dP in `stats` for the dK/dV kernel.
template <int MAXKT>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBatch batch, DropCfg drop, float* stats, int stat_n) {
  const AttnJob& job = batch.j[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Tq = job.Tq, Tk = job.Tk;
  const int qt = blockIdx.x * 4 + wave;
  const int b = blockIdx.y >> 3, h = blockIdx.y & 7;
  if (b >= job.B || qt * 16 >= Tq) return;
  const int q0 = qt * 16;
  const int nkt = (Tk + 15) >> 4;
  const int qbase = job.qrow0 + b * Tq, kbase = job.krow0 + b * Tk;
  const int qrow = qbase + min(q0 + j, Tq - 1);
  const float4 qb = ld4(job.Q + (size_t)qrow * job.ldq + 16 * h + 4 * g);
  const float4 dob = ld4(job.dO + (size_t)qrow * job.lddo + 16 * h + 4 * g);
  const float mq = job.qmask[qrow];
  f32x4 s[MAXKT], dp[MAXKT];
  // raw scores; keep the row max / sum to hand them to the dK/dV kernel
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt) {
    if (kt < nkt) {
      const int krow = kbase + min(kt * 16 + j, Tk - 1);
      const float4 ka = ld4(job.K + (size_t)krow * job.ldkv + 16 * h + 4 * g);
      const float4 va = ld4(job.V + (size_t)krow * job.ldkv + 16 * h + 4 * g);
      s[kt] = dot16(ka, qb);
      dp[kt] = dot16(va, dob);
    }
  }
  float mx = -INFINITY;
  const float* kmask_clip = job.kmask + kbase;
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt)
    if (kt < nkt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        float v = -INFINITY;
        if (key < Tk) v = s[kt][r] * ATT_SCALE + (1.0f - mq * kmask_clip[key]) * HUAL_MASK_VALUE;
        s[kt][r] = v;
        mx = fmaxf(mx, v);
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
        float e = __expf(s[kt][r] - mx);
        s[kt][r] = e;
        sum += e;
      }
    }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
  const bool dodrop = job.drop_site >= 0 && drop.enabled;
  const uint32_t drow = (job.drop_row0 + (uint32_t)qrow) * 8u + (uint32_t)h;
  float delta = 0.f;
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt)
    if (kt < nkt) {
      if (dodrop) {
        float4 m = drop_mask4(drop, (uint32_t)job.drop_site, drow, (uint32_t)(kt * 4 + g));
        dp[kt][0] *= m.x; dp[kt][1] *= m.y; dp[kt][2] *= m.z; dp[kt][3] *= m.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[kt][r] *= inv;
        delta = fmaf(s[kt][r], dp[kt][r], delta);
      }
    }
  delta += __shfl_xor(delta, 16);
  delta += __shfl_xor(delta, 32);
  if (g == 0 && (q0 + j) < Tq) {
    const int si = (b * Tq + q0 + j) * 8 + h;
    float* st = stats + (size_t)blockIdx.z * 3 * stat_n;
    st[si] = mx;
    st[stat_n + si] = inv;
    st[2 * stat_n + si] = delta;
  }
  f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < MAXKT; ++kt)
    if (kt < nkt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ds = s[kt][r] * (dp[kt][r] - delta) * ATT_SCALE;
        const int krow = kbase + min(kt * 16 + 4 * g + r, Tk - 1);
        const float kb = job.K[(size_t)krow * job.ldkv + 16 * h + j];
        dq = mfma16_(ds, kb, dq);
      }
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + 4 * g + r;
    if (q < Tq) job.dQ[(size_t)(qbase + q) * job.lddq + 16 * h + j] = dq[r];
  }
}

// dK, dV: a wave owns 16 keys of one (clip, head) and sweeps the query tiles.
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBatch batch, DropCfg drop, const float* stats, int stat_n) {
  const AttnJob& job = batch.j[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Tq = job.Tq, Tk = job.Tk;
  const int kt = blockIdx.x * 4 + wave;
  const int b = blockIdx.y >> 3, h = blockIdx.y & 7;
  if (b >= job.B || kt * 16 >= Tk) return;
  const int k0 = kt * 16;
  const int nqt = (Tq + 15) >> 4;
  const int qbase = job.qrow0 + b * Tq, kbase = job.krow0 + b * Tk;
  const int key = k0 + j;
  const bool keyok = key < Tk;
  const int krow = kbase + min(key, Tk - 1);
  const float4 kb = ld4(job.K + (size_t)krow * job.ldkv + 16 * h + 4 * g);
  const float4 vb = ld4(job.V + (size_t)krow * job.ldkv + 16 * h + 4 * g);
  const float mk = job.kmask[krow];
  const float* st = stats + (size_t)blockIdx.z * 3 * stat_n;
  const bool dodrop = job.drop_site >= 0 && drop.enabled;
  f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
  for (int qt = 0; qt < nqt; ++qt) {
    const int q0 = qt * 16;
    const int qrow_j = qbase + min(q0 + j, Tq - 1);
    const float4 qa = ld4(job.Q + (size_t)qrow_j * job.ldq + 16 * h + 4 * g);
    const float4 doa = ld4(job.dO + (size_t)qrow_j * job.lddo + 16 * h + 4 * g);
    f32x4 s = dot16(qa, kb);     // lane: S[query q0+4g+r][key k0+j]
    f32x4 dp = dot16(doa, vb);
    float qv[4], dov[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + 4 * g + r;
      const bool qok = q < Tq;
      const int qrow = qbase + min(q, Tq - 1);
      const int si = (b * Tq + min(q, Tq - 1)) * 8 + h;
      float p = 0.f;
      if (qok && keyok) {
        const float mqv = job.qmask[qrow];
        const float v = s[r] * ATT_SCALE + (1.0f - mqv * mk) * HUAL_MASK_VALUE;
        p = __expf(v - st[si]) * st[stat_n + si];
      }
      float m = 1.0f;
      if (dodrop) {
        const uint32_t drow = (job.drop_row0 + (uint32_t)qrow) * 8u + (uint32_t)h;
        float4 mm = drop_mask4(drop, (uint32_t)job.drop_site, drow, (uint32_t)(key >> 2));
        const int c = key & 3;
        m = c == 0 ? mm.x : (c == 1 ? mm.y : (c == 2 ? mm.z : mm.w));
      }
      const float ds = p * (dp[r] * m - st[2 * stat_n + si]) * ATT_SCALE;
      s[r] = p * m;     // dropped probability (A operand of dV)
      dp[r] = ds;       // A operand of dK
      qv[r] = job.Q[(size_t)qrow * job.ldq + 16 * h + j];
      dov[r] = job.dO[(size_t)qrow * job.lddo + 16 * h + j];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dv = mfma16_(s[r], dov[r], dv);
      dk = mfma16_(dp[r], qv[r], dk);
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

namespace hual {

static int check_jobs(const AttnJob* jobs, int n, bool bwd, int& maxTq, int& maxTk, int& maxB) {
  HUAL_REQUIRE(n >= 1 && n <= HUAL_MAX_ATTN_JOBS, "attn: job count");
  maxTq = maxTk = maxB = 0;
  for (int i = 0; i < n; ++i) {
    const AttnJob& j = jobs[i];
    HUAL_REQUIRE(j.Q && j.K && j.V && j.qmask && j.kmask, "attn: null operand");
    HUAL_REQUIRE(j.B > 0 && j.Tq > 0 && j.Tk > 0 && j.Tk <= 256, "attn: need 0 < Tk <= 256, Tq > 0");
    HUAL_REQUIRE((j.ldq % 4) == 0 && (j.ldkv % 4) == 0, "attn: leading dims must be multiples of 4");
    if (bwd) HUAL_REQUIRE(j.dO && j.dQ && j.dK && j.dV && (j.lddo % 4) == 0, "attn bwd: null gradient buffer");
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
  ProfScope ps(PK_ATTN_FWD, s, flops, 0.0);
  if (nkt <= 2) hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, block, 0, s, b, drop);
  else if (nkt <= 4) hipLaunchKernelGGL(attn_fwd_kernel<4>, grid, block, 0, s, b, drop);
  else if (nkt <= 8) hipLaunchKernelGGL(attn_fwd_kernel<8>, grid, block, 0, s, b, drop);
  else hipLaunchKernelGGL(attn_fwd_kernel<16>, grid, block, 0, s, b, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

// `stats` scratch: n jobs x 3 x stat_n floats, stat_n >= max_j(B*Tq*8)
int launch_attn_bwd_impl(const AttnJob* jobs, int n, const DropCfg& drop, float* stats, int stat_n, hipStream_t s) {
  int maxTq, maxTk, maxB;
  int rc = check_jobs(jobs, n, true, maxTq, maxTk, maxB);
  if (rc) return rc;
  HUAL_REQUIRE(stats != nullptr, "attn bwd: null stats scratch");
  for (int i = 0; i < n; ++i) HUAL_REQUIRE(jobs[i].B * jobs[i].Tq * 8 <= stat_n, "attn bwd: stats scratch too small");
  AttnBatch b;
  for (int i = 0; i < n; ++i) b.j[i] = jobs[i];
  dim3 grid(cdiv(cdiv(maxTq, 16), 4), maxB * 8, n), block(256);
  const int nkt = cdiv(maxTk, 16);
  double flops = 0.0;
  for (int i = 0; i < n; ++i) flops += 2.0 * jobs[i].B * 8.0 * jobs[i].Tq * jobs[i].Tk * 16.0;
  {
  ProfScope ps(PK_ATTN_BWD_DQ, s, 3.0 * flops, 0.0);   // S, dP, dQ
  if (nkt <= 2) hipLaunchKernelGGL(attn_bwd_dq_kernel<2>, grid, block, 0, s, b, drop, stats, stat_n);
  else if (nkt <= 4) hipLaunchKernelGGL(attn_bwd_dq_kernel<4>, grid, block, 0, s, b, drop, stats, stat_n);
  else if (nkt <= 8) hipLaunchKernelGGL(attn_bwd_dq_kernel<8>, grid, block, 0, s, b, drop, stats, stat_n);
  else hipLaunchKernelGGL(attn_bwd_dq_kernel<16>, grid, block, 0, s, b, drop, stats, stat_n);
  }
  HUAL_CHECK_HIP(hipGetLastError());
  dim3 grid2(cdiv(cdiv(maxTk, 16), 4), maxB * 8, n);
  ProfScope ps2(PK_ATTN_BWD_DKV, s, 4.0 * flops, 0.0);   // S, dP, dK, dV
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid2, block, 0, s, b, drop, (const float*)stats, stat_n);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
