// Attention forward at head size 64 (see attn.h `launch_attn_fwd_wide`): the structure of attn_fwd_kernel (attn.hip) -
// one workgroup per (clip, head), K / V split once into bf16 planes in LDS, a wave owns 16 queries and all keys,
// S^T = K.Q^T per 16-key tile with lane = query / registers = keys, in-register softmax, 8-bit Philox dropout, P.V with
// transposing LDS reads of the V panel - with the contraction over 64 head dims instead of 16.  SeqPAN itself runs 8 heads
// of 16 (configs: dim 128), where a 16 x 16 score tile is 2 + 1.5 MFMAs against ~100 vector instructions per lane; at
// head size 64 the same vector work stands against 6 + 6 MFMAs (three split passes over K = 64 for S, three over four
// 16-column tiles for P.V per key tile).  This file exists to measure the matrix-core utilisation of the attention
// products at a head size where they are not dwarfed by the softmax (north_star) - the model path
// does not use it.
#include "attn.h"
#include "bf16x3.h"
#include "philox.h"
#include "prof.h"

using namespace hual;

#define AW_DH 64
#define AW_LOG2E 1.4426950408889634f
#define AW_C1 (0.125f * AW_LOG2E)               // 1/sqrt(64), scores kept in the log2 domain
#define AW_NEGL (HUAL_MASK_VALUE * AW_LOG2E)

__device__ __forceinline__ f32x4 aw_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 aw_bf8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ void aw_split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
  bf16_split_pair(a.x, a.y, hi.x, lo.x);
  bf16_split_pair(a.z, a.w, hi.y, lo.y);
  bf16_split_pair(b.x, b.y, hi.z, lo.z);
  bf16_split_pair(b.z, b.w, hi.w, lo.w);
}
__device__ __forceinline__ uint32_t aw_t8(const DropCfg& d) {
  uint32_t t = (uint32_t)(((uint64_t)d.thresh + (1ull << 23)) >> 24);
  return t < 1u ? 1u : (t > 256u ? 256u : t);
}
// Panel row = one key: 16-byte chunks 0..7 = the 64 high parts, 8..15 = the 64 residuals (256 bytes).  The chunk index is
// XORed with ((row & 7) << 1) | ((row >> 3) & 1): 16 consecutive rows reading the same chunk (A operand of S^T) hit 16
// different chunks, and the transposing read of P.V - per 32-lane half 8 rows x 2 chunks x 2 halves of a chunk - hits 32
// different 8-byte slots.
__device__ __forceinline__ int aw_off(int row, int ch) {
  return row * 256 + 16 * (ch ^ (((row & 7) << 1) | ((row >> 3) & 1)));
}
__device__ __forceinline__ void aw_store(char* panel, int row, int c4, const float4& v) {     // dims 4 c4 .. 4 c4 + 3
  uint2 h, l;
  bf16_split4(v, h, l);
  *reinterpret_cast<uint2*>(panel + aw_off(row, c4 >> 1) + 8 * (c4 & 1)) = h;
  *reinterpret_cast<uint2*>(panel + aw_off(row, 8 + (c4 >> 1)) + 8 * (c4 & 1)) = l;
}
// B operand of P.V over the 32 keys row0 .. row0 + 31 for head dim 16 ct + (lane & 15): slots e < 4 <-> keys row0 + 4 g + e,
// e >= 4 <-> keys row0 + 16 + 4 g + (e - 4) (the order in which the accumulators of two key tiles hold them).  plane = 0 / 8.
__device__ __forceinline__ bf16x8 aw_tr(const char* panel, int plane, int row0, int ct, int lane) {
  const int row = row0 + 4 * (lane >> 4) + ((lane & 15) >> 2), p = lane & 3;
  const int ch = plane + 2 * ct + (p >> 1);
  return join_tr(lds_read_tr16(panel, aw_off(row, ch) + 8 * (p & 1)), lds_read_tr16(panel, aw_off(row + 16, ch) + 8 * (p & 1)));
}

struct AwArgs {
  const float* Q; int ldq;
  const float* K; const float* V; int ldkv;
  float* O; int ldo;
  int B, Tq, Tk, H;
  const float* qmask; const float* kmask;
  int drop_site;
};

template <int NKT, bool DROP>
__device__ __forceinline__ void aw_body(const AwArgs& a, int b, int h, char* lds, const DropCfg& drop) {
  constexpr int Tkp = 16 * NKT, NKQ = (NKT + 3) / 4, NST = (Tkp * 16 + 255) / 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Tq = a.Tq, Tk = a.Tk;
  const int qbase = b * Tq, kbase = b * Tk;
  char* Kp = lds;                                   // [Tkp][256]
  char* Vp = Kp + Tkp * 256;
  float* Bias = reinterpret_cast<float*>(Vp + Tkp * 256);      // [2][Tkp]
  const int nqt = (Tq + 15) >> 4;
  {
    const float* Kg = a.K + (size_t)kbase * a.ldkv + AW_DH * h;
    const float* Vg = a.V + (size_t)kbase * a.ldkv + AW_DH * h;
    // the panels in two halves of NST / 2 requests per thread (register budget)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 kv[(NST + 1) / 2], vv[(NST + 1) / 2];
#pragma unroll
      for (int it = 0; it < (NST + 1) / 2; ++it) {
        const int idx = threadIdx.x + 256 * (it + half * ((NST + 1) / 2)), row = min(idx >> 4, Tk - 1), c4 = idx & 15;
        kv[it] = ld4(Kg + (size_t)row * a.ldkv + 4 * c4);
        vv[it] = ld4(Vg + (size_t)row * a.ldkv + 4 * c4);
      }
#pragma unroll
      for (int it = 0; it < (NST + 1) / 2; ++it) {
        const int idx = threadIdx.x + 256 * (it + half * ((NST + 1) / 2)), row = idx >> 4, c4 = idx & 15;
        if (row < Tkp) {
          aw_store(Kp, row, c4, row < Tk ? kv[it] : f4zero());
          aw_store(Vp, row, c4, row < Tk ? vv[it] : f4zero());
        }
      }
    }
    if (threadIdx.x < Tkp) {
      const bool in = (int)threadIdx.x < Tk;
      const float km = in ? a.kmask[kbase + threadIdx.x] : 0.f;
      Bias[threadIdx.x] = in ? AW_NEGL : -INFINITY;                                   // (1 - mq mk) * -1e30 with mq = 0
      Bias[Tkp + threadIdx.x] = in ? (km != 0.f ? 0.f : AW_NEGL) : -INFINITY;
    }
  }
  __syncthreads();
  uint32_t k0 = 0, k1 = 0, off = 0;
  if (DROP) { k0 = drop_state(drop, 0); k1 = drop_state(drop, 1); off = drop_state(drop, 2); }      // scalar loads (philox.h)
  const uint32_t t8 = aw_t8(drop);
  const float scale8 = 256.0f / (float)t8;
  for (int qt = wave; qt < nqt; qt += 4) {
    const int q0 = qt * 16;
    const int qrow = qbase + min(q0 + j, Tq - 1);
    // B operands of S^T: lane (query j, g) holds head dims 32 c + 8 g .. + 7 of its query, c = 0, 1
    bf16x8 qh[2], ql[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* qp = a.Q + (size_t)qrow * a.ldq + AW_DH * h + 32 * c + 8 * g;
      uint4 hh, ll;
      aw_split8(ld4(qp), ld4(qp + 4), hh, ll);
      qh[c] = aw_bf8(hh); ql[c] = aw_bf8(ll);
    }
    const float mq = a.qmask[qrow];
    const float* bias = Bias + (mq != 0.f ? Tkp : 0);
    f32x4 s[NKT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Kp + aw_off(16 * kt + j, 4 * c + g));
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(Kp + aw_off(16 * kt + j, 8 + 4 * c + g));
        acc = aw_mfma(ah, qh[c], acc);
        acc = aw_mfma(al, qh[c], acc);
        acc = aw_mfma(ah, ql[c], acc);
      }
      const float4 b4 = *reinterpret_cast<const float4*>(bias + 16 * kt + 4 * g);
      acc[0] = fmaf(acc[0], AW_C1, b4.x); acc[1] = fmaf(acc[1], AW_C1, b4.y);
      acc[2] = fmaf(acc[2], AW_C1, b4.z); acc[3] = fmaf(acc[3], AW_C1, b4.w);
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
    const uint32_t drow = (uint32_t)qrow * (uint32_t)a.H + (uint32_t)h;
    const float keepv = DROP ? inv * scale8 : inv;
    f32x4 o[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq) {
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (DROP) {
        const uint4_ rnd = philox4x32((uint32_t)(g + 4 * kq), drow, (uint32_t)a.drop_site, off, k0, k1);
        w[0] = rnd.x; w[1] = rnd.y; w[2] = rnd.z; w[3] = rnd.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kt = 4 * kq + i;
        if (kt < NKT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool keep = !DROP || ((w[i] >> (8 * r)) & 0xffu) < t8;
            s[kt][r] *= keep ? keepv : 0.f;
          }
        }
      }
#pragma unroll
      for (int ip = 0; ip < 2; ++ip) {
        const int kp = 2 * kq + ip;                 // key tiles 2 kp, 2 kp + 1
        if (2 * kp < NKT) {
          uint4 ph, pl;
          aw_split8(make_float4(s[2 * kp][0], s[2 * kp][1], s[2 * kp][2], s[2 * kp][3]),
                    make_float4(s[2 * kp + 1][0], s[2 * kp + 1][1], s[2 * kp + 1][2], s[2 * kp + 1][3]), ph, pl);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            const bf16x8 vh = aw_tr(Vp, 0, 32 * kp, ct, lane), vl = aw_tr(Vp, 8, 32 * kp, ct, lane);
            o[ct] = aw_mfma(aw_bf8(ph), vh, o[ct]);
            o[ct] = aw_mfma(aw_bf8(ph), vl, o[ct]);
            o[ct] = aw_mfma(aw_bf8(pl), vh, o[ct]);
          }
        }
      }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + 4 * g + r;
        if (q < Tq) a.O[(size_t)(qbase + q) * a.ldo + AW_DH * h + 16 * ct + j] = o[ct][r];
      }
  }
}

template <int NKT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd_wide_kernel(AwArgs a, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
  if (a.drop_site >= 0 && drop.enabled) aw_body<NKT, true>(a, b, h, lds, drop);
  else aw_body<NKT, false>(a, b, h, lds, drop);
}

namespace hual {

int launch_attn_fwd_wide(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* O, int ldo, int B, int Tq,
                         int Tk, int heads, const float* qmask, const float* kmask, int drop_site, const DropCfg& drop,
                         hipStream_t s) {
  HUAL_REQUIRE(Q && K && V && O && qmask && kmask, "attn_fwd_wide: null pointer");
  HUAL_REQUIRE(B > 0 && Tq > 0 && Tk > 0 && Tk <= 128 && heads > 0, "attn_fwd_wide: need 0 < Tk <= 128");
  HUAL_REQUIRE((ldq % 4) == 0 && (ldkv % 4) == 0 && ldq >= 64 * heads && ldkv >= 64 * heads && ldo >= 64 * heads,
               "attn_fwd_wide: leading dims must be multiples of 4 and cover heads x 64 columns");
  AwArgs a{Q, ldq, K, V, ldkv, O, ldo, B, Tq, Tk, heads, qmask, kmask, drop_site};
  const int nkt = cdiv(Tk, 16);
  const int Tkp = 16 * (nkt <= 2 ? 2 : nkt <= 4 ? 4 : 8);
  const size_t lds = (size_t)2 * Tkp * 256 + 2 * Tkp * sizeof(float);
  const double flops = 4.0 * B * heads * (double)Tq * Tk * AW_DH;     // QK^T + PV
  HUAL_DYN_LDS(attn_fwd_wide_kernel<8>, 96 * 1024);
  const dim3 grid(B * heads), block(256);
  if (nkt <= 2) HUAL_LAUNCH(flops, 0.0, attn_fwd_wide_kernel<2>, grid, block, lds, s, a, drop);
  else if (nkt <= 4) HUAL_LAUNCH(flops, 0.0, attn_fwd_wide_kernel<4>, grid, block, lds, s, a, drop);
  else HUAL_LAUNCH(flops, 0.0, attn_fwd_wide_kernel<8>, grid, block, lds, s, a, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
