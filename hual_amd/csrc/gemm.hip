// Job-table GEMM kernels (see gemm.h).  Exact fp32 on the CDNA4 matrix cores.
//
// gemm_kernel: one wave computes a 16(row) x 64(col) output tile as four 16x16 accumulators with
// v_mfma_f32_16x16x4_f32.  Operand fragments come straight from global memory (weights are <= 512 KB and
// L2-resident; activations are [rows,128]) as 16-byte loads, WITHOUT an LDS stage or any barrier:
//   * lane (j = lane&15, g = lane>>4) loads A[row0+j][k0+4g .. +3]        -> k-step c uses component c
//   * and W[k0+4g+c][n0+4j .. +3] for c = 0..3                             -> accumulator t uses component t
// i.e. the MFMA's k index inside a 16-deep chunk is permuted (k = 4g + c) identically on A and B, and the
// accumulator's column index j of tile t is output column n0 + 4j + t, so the epilogue stores float4s.
// For dX = dY.W^T (transW) the same A pattern is used and lane (j,g) loads W[n0+4j+t][k0+4g .. +3].
//
// dw_kernel: dW += A^T.dY with v_mfma_f32_32x32x2_f32 (k index of the MFMA = row m of A/dY), 64x64 tile per
// wave, four waves of a block split the block's M-chunk, are summed through LDS and leave as ONE set of
// float atomics per block whose wave-instructions are two contiguous 128-B row segments (the full-rate
// atomic shape of MI355X_MICROARCH.md "Global float atomics").
#include "gemm.h"
#include "philox.h"
#include "bf16x3.h"
#include "prof.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

namespace hual {

void gemm_job_init(GemmJob& j) {
  ::memset((void*)&j, 0, sizeof(j));
  j.a_drop_site = -1;
  j.drop_site = -1;
  j.add_div = 1;
}
void dw_job_init(DwJob& j) {
  ::memset((void*)&j, 0, sizeof(j));
  j.a_drop_site = -1;
}

}  // namespace hual

using namespace hual;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float f4get(const float4& v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

struct Frag {
  float4 a;
  float4 b[4];
};

__device__ __forceinline__ void load_frag(Frag& f, const float* Ap, const float* A2p, const float* Wp, int ldw,
                                          int transW, int k0, int g, int j, int n0, int N, bool adrop,
                                          const DropCfg& drop, uint32_t site, uint32_t droprow) {
  f.a = ld4(Ap + k0 + 4 * g);
  if (A2p) f.a = f4mul(f.a, ld4(A2p + k0 + 4 * g));
  if (adrop) f.a = apply_drop4(drop, site, droprow, (uint32_t)((k0 + 4 * g) >> 2), f.a);
  if (!transW) {
#pragma unroll
    for (int c = 0; c < 4; ++c) f.b[c] = ld4(Wp + (size_t)(k0 + 4 * g + c) * ldw + min(n0 + 4 * j, N - 4));
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) f.b[t] = ld4(Wp + (size_t)min(n0 + 4 * j + t, N - 1) * ldw + k0 + 4 * g);
  }
}

__device__ __forceinline__ void mma_frag(f32x4 (&acc)[4], const float4& a, const float4 (&b)[4], int transW) {
  if (!transW) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float av = f4get(a, c);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(av, f4get(b[c], t), acc[t]);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float av = f4get(a, c);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(av, f4get(b[t], c), acc[t]);
    }
  }
}

template <bool DUAL>
__device__ __forceinline__ void gemm_epilogue(const GemmJob& job, const DropCfg& drop, f32x4 (&acc)[4], f32x4 (&acc2)[4],
                                              int rowbase, int n0, int j, int g) {
  const int M = job.M, N = job.N;
  // ---------------- epilogue: lane owns rows rowbase+4g+r (r=0..3), columns n0+4j .. n0+4j+3 -------------
  const int col = n0 + 4 * j;
  float4 bias = (job.bias && col < N) ? ld4(job.bias + col) : f4zero();
  float4 bias2 = f4zero();
  if (DUAL && job.bias2 && col < N) bias2 = ld4(job.bias2 + col);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = rowbase + 4 * g + r;
    if (row >= M || col >= N) continue;
    float4 v = make_float4(acc[0][r] + bias.x, acc[1][r] + bias.y, acc[2][r] + bias.z, acc[3][r] + bias.w);
    float rm = job.rowmask ? job.rowmask[row] : 1.0f;
    if (job.act == ACT_RELU) {
      v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    } else if (job.act == ACT_SIGMOID) {
      v = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
    } else if (job.act == ACT_SIGMOID_ROWMASK) {
      // sigmoid(mask_logits(x, m)): m=1 -> sigmoid(x); m=0 -> sigmoid(-1e30) == 0   (layers.py:110)
      v = rm != 0.f ? make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)) : f4zero();
    }
    if (DUAL) {
      float4 v2 = make_float4(acc2[0][r] + bias2.x, acc2[1][r] + bias2.y, acc2[2][r] + bias2.z, acc2[3][r] + bias2.w);
      if (job.comb == COMB_GATE_VAL) {
        if (job.save) st4(job.save + (size_t)row * job.ldsave + col, v);
        if (job.save2) st4(job.save2 + (size_t)row * job.ldsave2 + col, v2);
        v = f4mul(v, v2);
      } else if (job.comb == COMB_CROSSGATE) {
        v2 = make_float4(sigmoidf_(v2.x), sigmoidf_(v2.y), sigmoidf_(v2.z), sigmoidf_(v2.w));
        if (job.save) st4(job.save + (size_t)row * job.ldsave + col, v);
        if (job.save2) st4(job.save2 + (size_t)row * job.ldsave2 + col, v2);
        float4 x1 = ld4(job.aux1 + (size_t)row * job.ldaux + col);
        float4 x2 = ld4(job.aux2 + (size_t)row * job.ldaux + col);
        v = f4add(f4mul(v, x1), f4mul(v2, x2));
      }
    } else {
      if (job.save) st4(job.save + (size_t)row * job.ldsave + col, v);
    }
    if (job.mulmode != MUL_NONE) {
      float4 m = ld4(job.mul + (size_t)row * job.ldmul + col);
      if (job.mulmode == MUL_TENSOR) {
        v = f4mul(v, m);
      } else if (job.mulmode == MUL_DRELU) {
        v = make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
      } else {
        v = make_float4(v.x * m.x * (1.f - m.x), v.y * m.y * (1.f - m.y), v.z * m.z * (1.f - m.z), v.w * m.w * (1.f - m.w));
      }
    }
    if (job.drop_site >= 0 && drop.enabled)
      v = apply_drop4(drop, (uint32_t)job.drop_site, job.drop_row0 + (uint32_t)row, (uint32_t)(col >> 2), v);
    if (job.add) v = f4add(v, ld4(job.add + (size_t)(row / job.add_div) * job.ldadd + col));
    if (job.mask_out) v = make_float4(v.x * rm, v.y * rm, v.z * rm, v.w * rm);
    st4(job.Y + (size_t)row * job.ldy + col, v);
  }
}

// chunk index (16 k's each, over the concatenated pieces) -> piece and offset inside the piece (wave-uniform)
__device__ __forceinline__ void chunk_to_piece(const GemmJob& job, int ch, int& p, int& k0) {
  int k = ch * 16;
  p = 0;
  while (p + 1 < job.npieces && k >= job.kw[p]) {
    k -= job.kw[p];
    ++p;
  }
  k0 = k;
}

template <bool DUAL>
__device__ __forceinline__ void load_chunk(const GemmJob& job, int ch, int arow, int g, int j, int n0, bool adrop,
                                           const DropCfg& drop, Frag& f, Frag& fb) {
  int p, k0;
  chunk_to_piece(job, ch, p, k0);
  const float* Ap = job.A[p] + (size_t)arow * job.lda[p];
  const float* A2p = job.A2[p] ? job.A2[p] + (size_t)arow * job.lda2[p] : nullptr;
  load_frag(f, Ap, A2p, job.W[p], job.ldw, job.transW, k0, g, j, n0, job.N, adrop, drop, (uint32_t)job.a_drop_site,
            job.a_drop_row0 + (uint32_t)arow);
  if (DUAL) {
    const float* Abp = job.Ab[p] ? job.Ab[p] + (size_t)arow * job.ldab[p] : Ap;
    load_frag(fb, Abp, nullptr, job.W2[p], job.ldw, job.transW, k0, g, j, n0, job.N, false, drop, 0, 0);
  }
}

#define HUAL_PD 4   // prefetch depth in 16-k chunks: the whole K=128 panel of a wave is in flight after 2 groups

template <bool DUAL>
__global__ __launch_bounds__(256) void gemm_kernel(GemmBatch batch, DropCfg drop) {
  const GemmJob& job = batch.j[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int M = job.M, N = job.N;
  const int rowbase = blockIdx.x * 32 + (wave >> 1) * 16;
  const int n0 = blockIdx.y * 128 + (wave & 1) * 64;
  if (rowbase >= M || n0 >= N) return;   // wave-uniform
  const int arow = min(rowbase + j, M - 1);
  const int transW = job.transW;
  const bool adrop = job.a_drop_site >= 0 && drop.enabled;

  f32x4 acc[4], acc2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  int ktot = 0;
  for (int p = 0; p < job.npieces; ++p) ktot += job.kw[p];
  const int nch = ktot >> 4;
  const int ngroups = nch / HUAL_PD;

  Frag f[HUAL_PD], fb[HUAL_PD];
  if (ngroups > 0) {
#pragma unroll
    for (int u = 0; u < HUAL_PD; ++u) load_chunk<DUAL>(job, u, arow, g, j, n0, adrop, drop, f[u], fb[u]);
    for (int gi = 0; gi + 1 < ngroups; ++gi) {
#pragma unroll
      for (int u = 0; u < HUAL_PD; ++u) {
        mma_frag(acc, f[u].a, f[u].b, transW);
        if (DUAL) mma_frag(acc2, fb[u].a, fb[u].b, transW);
        load_chunk<DUAL>(job, (gi + 1) * HUAL_PD + u, arow, g, j, n0, adrop, drop, f[u], fb[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < HUAL_PD; ++u) {
      mma_frag(acc, f[u].a, f[u].b, transW);
      if (DUAL) mma_frag(acc2, fb[u].a, fb[u].b, transW);
    }
  }
  for (int ch = ngroups * HUAL_PD; ch < nch; ++ch) {   // K tail (< HUAL_PD chunks)
    load_chunk<DUAL>(job, ch, arow, g, j, n0, adrop, drop, f[0], fb[0]);
    mma_frag(acc, f[0].a, f[0].b, transW);
    if (DUAL) mma_frag(acc2, fb[0].a, fb[0].b, transW);
  }

  gemm_epilogue<DUAL>(job, drop, acc, acc2, rowbase, n0, j, g);
}

// ------------------------------------------------------------------------------------------------------
// gemm_lds_kernel: same job semantics and the same wave -> (16 rows x 64 cols) mapping as gemm_kernel, built to
// expose ONE memory round trip per block instead of four (PMC on gemm_kernel: 61 % of wave cycles in s_waitcnt,
// 26 k cycles of wave life for 4 k cycles of MFMA):
//   * the weight panel (all 128 output columns x 64 K rows per stage, 32 KB) goes global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4: no staging registers), once per BLOCK, shared by the four waves; two stages (= a
//     whole K=128 layer) are requested in the prologue, later stages are requested as soon as a buffer is free;
//   * each wave requests the A fragments of those stages and its epilogue operands (bias, residual rows) in the
//     same prologue, so they are in registers when the MFMAs / the epilogue need them;
//   * the panel image is the plain [64][128] row-major tile: for the B-fragment ds_read_b128 (lane (j,g) reads 16 B
//     at row 4g+c, column 4j) the hardware's 16-lane groups then cover disjoint bank ranges - conflict free.
// Weights must be stored [K,N] (transW = 0); dX jobs use the transposed copy made by transpose_weights_kernel.
#define GL_KS 64            // K rows per stage
#define GL_STAGE (GL_KS * 128)   // floats per staged panel

struct EpiRegs {
  float4 bias, bias2;
  float4 add[4];
  float rm[4];
};

template <bool DUAL, class J>
__device__ __forceinline__ void epi_prefetch(const J& job, EpiRegs& e, int rowbase, int n0, int j, int g) {
  const int M = job.M, N = job.N;
  const int col = n0 + 4 * j;
  const bool cok = col < N;
  e.bias = (job.bias && cok) ? ld4(job.bias + col) : f4zero();
  e.bias2 = (DUAL && job.bias2 && cok) ? ld4(job.bias2 + col) : f4zero();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(rowbase + 4 * g + r, M - 1);
    e.rm[r] = job.rowmask ? job.rowmask[row] : 1.0f;
    e.add[r] = (job.add && cok) ? ld4(job.add + (size_t)(row / job.add_div) * job.ldadd + col) : f4zero();
  }
}

template <bool DUAL, class J>
__device__ __forceinline__ void epi_apply(const J& job, const DropCfg& drop, const EpiRegs& e, f32x4 (&acc)[4],
                                          f32x4 (&acc2)[4], int rowbase, int n0, int j, int g) {
  const int M = job.M, N = job.N;
  const int col = n0 + 4 * j;
  if (col >= N) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = rowbase + 4 * g + r;
    if (row >= M) continue;
    float4 v = make_float4(acc[0][r] + e.bias.x, acc[1][r] + e.bias.y, acc[2][r] + e.bias.z, acc[3][r] + e.bias.w);
    const float rm = e.rm[r];
    if (job.act == ACT_RELU) {
      v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    } else if (job.act == ACT_SIGMOID) {
      v = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
    } else if (job.act == ACT_SIGMOID_ROWMASK) {
      v = rm != 0.f ? make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)) : f4zero();
    }
    if (DUAL) {
      float4 v2 = make_float4(acc2[0][r] + e.bias2.x, acc2[1][r] + e.bias2.y, acc2[2][r] + e.bias2.z, acc2[3][r] + e.bias2.w);
      if (job.comb == COMB_GATE_VAL) {
        if (job.save) st4(job.save + (size_t)row * job.ldsave + col, v);
        if (job.save2) st4(job.save2 + (size_t)row * job.ldsave2 + col, v2);
        v = f4mul(v, v2);
      } else if (job.comb == COMB_CROSSGATE) {
        v2 = make_float4(sigmoidf_(v2.x), sigmoidf_(v2.y), sigmoidf_(v2.z), sigmoidf_(v2.w));
        if (job.save) st4(job.save + (size_t)row * job.ldsave + col, v);
        if (job.save2) st4(job.save2 + (size_t)row * job.ldsave2 + col, v2);
        float4 x1 = ld4(job.aux1 + (size_t)row * job.ldaux + col);
        float4 x2 = ld4(job.aux2 + (size_t)row * job.ldaux + col);
        v = f4add(f4mul(v, x1), f4mul(v2, x2));
      }
    } else {
      if (job.save) st4(job.save + (size_t)row * job.ldsave + col, v);
    }
    if (job.mulmode != MUL_NONE) {
      const float4 m = ld4(job.mul + (size_t)row * job.ldmul + col);
      if (job.mulmode == MUL_TENSOR) {
        v = f4mul(v, m);
      } else if (job.mulmode == MUL_DRELU) {
        v = make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
      } else {
        v = make_float4(v.x * m.x * (1.f - m.x), v.y * m.y * (1.f - m.y), v.z * m.z * (1.f - m.z), v.w * m.w * (1.f - m.w));
      }
    }
    if (job.drop_site >= 0 && drop.enabled)
      v = apply_drop4(drop, (uint32_t)job.drop_site, job.drop_row0 + (uint32_t)row, (uint32_t)(col >> 2), v);
    if (job.add) v = f4add(v, e.add[r]);
    if (job.mask_out) v = make_float4(v.x * rm, v.y * rm, v.z * rm, v.w * rm);
    st4(job.Y + (size_t)row * job.ldy + col, v);
  }
}

// stage index (64 K rows each, over the concatenated pieces) -> piece and K offset inside it (block-uniform)
template <class J>
__device__ __forceinline__ void stage_to_piece(const J& job, int st, int& p, int& k0) {
  p = 0;
  while (p + 1 < job.npieces) {
    const int n = (job.kw[p] + GL_KS - 1) / GL_KS;
    if (st < n) break;
    st -= n;
    ++p;
  }
  k0 = st * GL_KS;
}

// RT = 16-row tiles per block (block = 2*RT waves = RT row tiles x 2 column halves).  RT = 3 (48 rows, 384 threads)
// is picked when it brings a launch down to one block per CU: the kernel is bound by what each CU can pull through
// its vector-memory path (64 KB of weights per block + the activations), so fewer, taller blocks win.
template <bool DUAL, int RT>
__global__ __launch_bounds__(RT * 128) void gemm_lds_kernel(GemmBatch batch, DropCfg drop) {
  extern __shared__ float lds[];     // Ws[2][GL_STAGE] (+ W2s[2][GL_STAGE] when DUAL)
  const GemmJob& job = batch.j[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int M = job.M, N = job.N;
  const int blockrow = blockIdx.x * (16 * RT);
  const int nblk = blockIdx.y * 128;
  if (blockrow >= M || nblk >= N) return;          // block-uniform
  const int rowbase = blockrow + (wave >> 1) * 16;
  const int n0 = nblk + (wave & 1) * 64;
  const bool wave_on = rowbase < M && n0 < N;       // idle waves still take part in staging and barriers
  const int arow = min(rowbase + j, M - 1);
  const int ldw = job.ldw;
  const bool adrop = job.a_drop_site >= 0 && drop.enabled;
  const uint32_t asite = (uint32_t)job.a_drop_site;
  const uint32_t adrow = job.a_drop_row0 + (uint32_t)arow;
  float* Ws = lds;
  float* W2s = lds + 2 * GL_STAGE;
  int nstages = 0;
  for (int p = 0; p < job.npieces; ++p) nstages += (job.kw[p] + GL_KS - 1) / GL_KS;

  f32x4 acc[4], acc2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // LDS-DMA of one stage: 32 pieces of 1 KB (2 panel rows each), 8 per wave; lane l -> row 2*pc + (l>>5), 16 B at
  // column 4*(l&31).  Out-of-range rows / columns are clamped to valid memory (never multiplied / never stored).
  auto dma_stage = [&](const float* Wp, int k0, int kw, float* dst) {
    const int c4 = lane & 31, rr = lane >> 5;
    const int n = min(nblk + 4 * c4, N - 4);
    for (int pc = wave; pc < 32; pc += 2 * RT) {
      const int kk = min(k0 + 2 * pc + rr, kw - 1);
      const float* src = Wp + (size_t)kk * ldw + n;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + pc * 256), 16, 0, 0);
    }
  };
  auto a_load = [&](const float* Ap, const float* A2p, int k0, int kw, float4 (&a)[4]) {
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const int kk = k0 + kc * 16;
      float4 v = f4zero();
      if (kk < kw) {
        v = ld4(Ap + kk + 4 * g);
        if (A2p) v = f4mul(v, ld4(A2p + kk + 4 * g));
        if (adrop) v = apply_drop4(drop, asite, adrow, (uint32_t)((kk + 4 * g) >> 2), v);
      }
      a[kc] = v;
    }
  };
  // request stage s: weight panel(s) by DMA into buffer (s & 1), A fragments into the given register slot
  auto issue = [&](int s, float4 (&a)[4], float4 (&a2)[4], int& kw_out, int& k0_out) {
    int p, k0;
    stage_to_piece(job, s, p, k0);
    const int kw = job.kw[p];
    dma_stage(job.W[p], k0, kw, Ws + (s & 1) * GL_STAGE);
    if (DUAL) dma_stage(job.W2[p], k0, kw, W2s + (s & 1) * GL_STAGE);
    const float* Ap = job.A[p] + (size_t)arow * job.lda[p];
    const float* A2p = job.A2[p] ? job.A2[p] + (size_t)arow * job.lda2[p] : nullptr;
    a_load(Ap, A2p, k0, kw, a);
    if (DUAL) {
      if (job.Ab[p]) a_load(job.Ab[p] + (size_t)arow * job.ldab[p], nullptr, k0, kw, a2);
      else {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) a2[kc] = a[kc];
      }
    }
    kw_out = kw;
    k0_out = k0;
  };
  auto compute = [&](int bufi, int k0, int kw, const float4 (&a)[4], const float4 (&a2)[4]) {
    const float* wsb = Ws + bufi * GL_STAGE + (wave & 1) * 64 + 4 * j;
    const float* w2b = W2s + bufi * GL_STAGE + (wave & 1) * 64 + 4 * j;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      if (k0 + kc * 16 < kw) {
        float4 b[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const float4*>(wsb + (kc * 16 + 4 * g + c) * 128);
        mma_frag(acc, a[kc], b, 0);
        if (DUAL) {
#pragma unroll
          for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const float4*>(w2b + (kc * 16 + 4 * g + c) * 128);
          mma_frag(acc2, a2[kc], b, 0);
        }
      }
    }
  };

  float4 a0[4], a1[4], c0[4], c1[4];
  int kw0 = 0, kw1 = 0, k00 = 0, k01 = 0;
  EpiRegs epi;
  // prologue: the first TWO stages (a whole K=128 layer) + everything the epilogue will read
  issue(0, a0, c0, kw0, k00);
  if (nstages > 1) issue(1, a1, c1, kw1, k01);
  epi_prefetch<DUAL>(job, epi, rowbase, n0, j, g);
  for (int s = 0; s < nstages; s += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // DMA of the resident stages has landed (this wave's part)
    __syncthreads();                                      // ... and everybody else's
    if (wave_on) compute(0, k00, kw0, a0, c0);
    if (s + 1 < nstages && wave_on) compute(1, k01, kw1, a1, c1);
    if (s + 2 < nstages) {
      __syncthreads();                                    // both buffers consumed by every wave
      issue(s + 2, a0, c0, kw0, k00);
      if (s + 3 < nstages) issue(s + 3, a1, c1, kw1, k01);
    }
  }
  if (!wave_on) return;
  epi_apply<DUAL>(job, drop, epi, acc, acc2, rowbase, n0, j, g);
}

// ------------------------------------------------------------------------------------------------------
// Split-bf16 dense kernel.  Same jobs, same wave -> (16 rows x 64 columns) mapping, same two-stage LDS-DMA pipeline and
// the same epilogue as gemm_lds_kernel; the products run as three v_mfma_f32_16x16x32_bf16 passes on split operands
// (bf16x3.h): 48 MFMAs x 16 cycles per 128-deep layer and wave instead of 128 x 32.
//   * weights come pre-split from pack_weights_kernel: per K row 256 B of bf16 high parts then 256 B of residuals, the
//     128 columns stored in the order that makes the transposed LDS read hand lane j of tile t column 4j + t (the
//     accumulator -> column map of the epilogue).  A stage (64 K rows) is two [64][256 B] tiles (hi, lo) in the
//     XOR-swizzled dual-use layout of bf16x3.h; the swizzle is applied on the GLOBAL side of the LDS-DMA (lane l of a
//     1 KB piece fetches chunk (l&15) ^ sw(row) of its row), the LDS side of global_load_lds being lane-linear.
//   * B fragments: ds_read_b64_tr_b16 (k is the strided direction of a [k][n] image), 2 reads per tile and plane;
//   * A fragments: lane (j, g) loads 8 consecutive k of row j as two float4 and splits them right before the MFMAs.
// dX = dY.W^T uses the same kernel on the image of the transposed weight (pack_weights_kernel writes both).
#define GB_TILE (GL_KS * 256)            // bytes of one [64][128 x bf16] tile
#define GB_STAGE (2 * GB_TILE)           // hi + lo
#define GB_COLBLOCK (128 * 512)          // bytes between the images of consecutive 128-column blocks (dX with N > 128)

// MODE 0: single accumulator; 1: dual (cross gating / bilinear); 2: decided per job at run time (chained launches)
template <int MODE, int RT, class J>
__device__ __forceinline__ void gemm_bf16_body(const J& job, const DropCfg& drop, char* ldsb) {
  const bool DUAL = MODE == 1 || (MODE == 2 && job.comb != COMB_NONE);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int M = job.M, N = job.N;
  const int blockrow = blockIdx.x * (16 * RT);
  const int nblk = blockIdx.y * 128;
  if (blockrow >= M || nblk >= N) return;          // block-uniform
  const int rowbase = blockrow + (wave >> 1) * 16;
  const int n0 = nblk + (wave & 1) * 64;
  const bool wave_on = rowbase < M && n0 < N;       // idle waves still take part in staging, barriers and LDS reads
  const int arow = min(rowbase + j, M - 1);
  const bool adrop = job.a_drop_site >= 0 && drop.enabled;
  const uint32_t asite = (uint32_t)job.a_drop_site;
  const uint32_t adrow = job.a_drop_row0 + (uint32_t)arow;
  char* Ws = ldsb;
  char* W2s = ldsb + 2 * GB_STAGE;
  int nstages = 0;
  for (int p = 0; p < job.npieces; ++p) nstages += (job.kw[p] + GL_KS - 1) / GL_KS;

  f32x4 acc[4], acc2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // LDS-DMA of one stage: 32 pieces of 1 KB = 4 tile rows each (pieces 0-15: hi tile, 16-31: lo tile)
  auto dma_stage = [&](const float* Wimg, int k0, int kw, char* dst) {
    const char* img = reinterpret_cast<const char*>(Wimg) + (size_t)blockIdx.y * GB_COLBLOCK;
    const int chp = lane & 15, rr = lane >> 4;
    for (int pc = wave; pc < 32; pc += 2 * RT) {
      const int r = 4 * (pc & 15) + rr;
      const int kk = min(k0 + r, kw - 1);
      const int ch = chp ^ (((r & 3) << 2) | ((r >> 2) & 3));
      const char* src = img + (size_t)kk * 512 + (pc >> 4) * 256 + 16 * ch;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
    }
  };
  // raw A fragments of one stage: a[2*ks + half] = row arow, k = k0 + 32 ks + 8 g + 4 half .. +3
  auto a_load = [&](const float* Ap, const float* A2p, int k0, int kw, float4 (&a)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k0 + 32 * (u >> 1) + 8 * g + 4 * (u & 1);
      float4 v = f4zero();
      if (kk < kw) {
        v = ld4(Ap + kk);
        if (A2p) v = f4mul(v, ld4(A2p + kk));
        if (adrop) v = apply_drop4(drop, asite, adrow, (uint32_t)(kk >> 2), v);
      }
      a[u] = v;
    }
  };
  auto issue = [&](int s, float4 (&a)[4], float4 (&a2)[4], int& kw_out, int& k0_out) {
    int p, k0;
    stage_to_piece(job, s, p, k0);
    const int kw = job.kw[p];
    dma_stage(job.W[p], k0, kw, Ws + (s & 1) * GB_STAGE);
    if (MODE && DUAL) dma_stage(job.W2[p], k0, kw, W2s + (s & 1) * GB_STAGE);
    const float* Ap = job.A[p] + (size_t)arow * job.lda[p];
    const float* A2p = job.A2[p] ? job.A2[p] + (size_t)arow * job.lda2[p] : nullptr;
    a_load(Ap, A2p, k0, kw, a);
    if (MODE && DUAL) {
      if (job.Ab[p]) a_load(job.Ab[p] + (size_t)arow * job.ldab[p], nullptr, k0, kw, a2);
      else {
#pragma unroll
        for (int u = 0; u < 4; ++u) a2[u] = a[u];
      }
    }
    kw_out = kw;
    k0_out = k0;
  };
  // transposed-read addressing: 16-lane group g takes rows 32 ks + 8 g + 4 rr + q, lane 4q+pp supplies stored columns
  // 64 (wave&1) + 16 t + 4 pp .. +3
  const int tq = (lane >> 2) & 3, tp = lane & 3;
  f32x4 accp[4], accp2[4];     // accumulators of the current 128-deep chunk (operands carry the chunk's row scale)
  // sc / sc2: this lane's A row scale of the chunk (bf16x3.h "f16x3")
  auto compute = [&](int bufi, int k0, int kw, const float4 (&a)[4], const float4 (&a2)[4], float sc, float sc2) {
    const char* hi = Ws + bufi * GB_STAGE;
    const char* hi2 = W2s + bufi * GB_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (k0 + 32 * ks < kw) {             // block-uniform
        uint2 h0, l0, h1, l1;
        f16_split4(f4scale1(a[2 * ks], sc), h0, l0);
        f16_split4(f4scale1(a[2 * ks + 1], sc), h1, l1);
        const f16x8 ah = __builtin_bit_cast(f16x8, (u32x4){h0.x, h0.y, h1.x, h1.y});
        const f16x8 al = __builtin_bit_cast(f16x8, (u32x4){l0.x, l0.y, l1.x, l1.y});
        f16x8 bh, bl, ch, cl;
        if (MODE && DUAL) {
          f16_split4(f4scale1(a2[2 * ks], sc2), h0, l0);
          f16_split4(f4scale1(a2[2 * ks + 1], sc2), h1, l1);
          bh = __builtin_bit_cast(f16x8, (u32x4){h0.x, h0.y, h1.x, h1.y});
          bl = __builtin_bit_cast(f16x8, (u32x4){l0.x, l0.y, l1.x, l1.y});
        }
        const int r0 = 32 * ks + 8 * g + tq, r1 = r0 + 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int chunk = 8 * (wave & 1) + 2 * t + (tp >> 1);
          const int o0 = tile256_off(r0, chunk) + 8 * (tp & 1), o1 = tile256_off(r1, chunk) + 8 * (tp & 1);
          const f16x8 wh = join_tr_f16(lds_read_tr16(hi, o0), lds_read_tr16(hi, o1));
          const f16x8 wl = join_tr_f16(lds_read_tr16(hi + GB_TILE, o0), lds_read_tr16(hi + GB_TILE, o1));
          accp[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, accp[t], 0, 0, 0);
          accp[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl, accp[t], 0, 0, 0);
          accp[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh, accp[t], 0, 0, 0);
          if (MODE && DUAL) {
            ch = join_tr_f16(lds_read_tr16(hi2, o0), lds_read_tr16(hi2, o1));
            cl = join_tr_f16(lds_read_tr16(hi2 + GB_TILE, o0), lds_read_tr16(hi2 + GB_TILE, o1));
            accp2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ch, accp2[t], 0, 0, 0);
            accp2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, cl, accp2[t], 0, 0, 0);
            accp2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ch, accp2[t], 0, 0, 0);
          }
        }
      }
    }
  };
  // largest magnitude of this lane's row over the fragments of the resident stages -> row scale (all 4 lanes of a row
  // agree after the two shuffles)
  auto row_scale = [&](const float4 (&x0)[4], const float4 (&x1)[4], bool two, float& inv) {
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      m = fmaxf(m, f4absmax(x0[u]));
      if (two) m = fmaxf(m, f4absmax(x1[u]));
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    return f16_row_scale(m, inv);
  };

  float4 a0[4], a1[4], c0[4], c1[4];
  int kw0 = 0, kw1 = 0, k00 = 0, k01 = 0;
  EpiRegs epi;
  issue(0, a0, c0, kw0, k00);
  if (nstages > 1) issue(1, a1, c1, kw1, k01);
  if (MODE == 0) epi_prefetch<false>(job, epi, rowbase, n0, j, g);
  else if (DUAL) epi_prefetch<true>(job, epi, rowbase, n0, j, g);
  else epi_prefetch<false>(job, epi, rowbase, n0, j, g);
  for (int s = 0; s < nstages; s += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // DMA of the resident stages has landed (this wave's part)
    __syncthreads();                                      // ... and everybody else's
    const bool two = s + 1 < nstages;
    float inv = 0.f, inv2 = 0.f, sc2 = 0.f;
    const float sc = row_scale(a0, a1, two, inv);
    if (MODE && DUAL) sc2 = row_scale(c0, c1, two, inv2);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      accp[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      accp2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    compute(0, k00, kw0, a0, c0, sc, sc2);                // every wave: the transposed reads need EXEC all ones
    if (two) compute(1, k01, kw1, a1, c1, sc, sc2);
    // fold the chunk into the total: accumulator register r is row 4g + r of the tile, whose scale lives in lanes j = 4g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ir = __shfl(inv, 4 * g + r);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][r] = fmaf(accp[t][r], ir, acc[t][r]);
      if (MODE && DUAL) {
        const float ir2 = __shfl(inv2, 4 * g + r);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc2[t][r] = fmaf(accp2[t][r], ir2, acc2[t][r]);
      }
    }
    if (s + 2 < nstages) {
      __syncthreads();                                    // both buffers consumed by every wave
      issue(s + 2, a0, c0, kw0, k00);
      if (s + 3 < nstages) issue(s + 3, a1, c1, kw1, k01);
    }
  }
  if (!wave_on) return;
  if (MODE == 0) epi_apply<false>(job, drop, epi, acc, acc2, rowbase, n0, j, g);
  else if (DUAL) epi_apply<true>(job, drop, epi, acc, acc2, rowbase, n0, j, g);
  else epi_apply<false>(job, drop, epi, acc, acc2, rowbase, n0, j, g);
}

template <bool DUAL, int RT>
__global__ __launch_bounds__(RT * 128) void gemm_bf16_kernel(GemmBatch batch, DropCfg drop) {
  extern __shared__ float lds[];     // Ws[2][GB_STAGE] (+ W2s[2][GB_STAGE] when DUAL)
  gemm_bf16_body<DUAL ? 1 : 0, RT>(batch.j[blockIdx.z], drop, reinterpret_cast<char*>(lds));
}

// The same kernel under its own symbol for the deep-K launch of a step - video_conv1d (+ query_conv1d), model.py:42,48: the
// FEATURE-LOAD phase that streams the [B,T,vdim] clip features from HBM - so that rocprofv3 and bench.py's per-kernel
// table show that phase separately from the 128-deep layers.
template <int RT>
__global__ __launch_bounds__(RT * 128) void feature_load_gemm_kernel(GemmBatch batch, DropCfg drop) {
  extern __shared__ float lds[];
  gemm_bf16_body<0, RT>(batch.j[blockIdx.z], drop, reinterpret_cast<char*>(lds));
}

// ------------------------------------------------------------------------------------------------------
// Feature-load kernel: partial products of video_conv1d (model.py:47-48), Y_q[rows, 128] = dropout(video)[rows, Kq] . W[Kq, :]
// for the K-quarter q = blockIdx.y.  The deep-K launch above is bound by what one CU pulls in (the whole 512 KB weight
// image next to 128 KB of clip features) and both column-half waves draw the same dropout decisions.  Here a block owns
// 128 rows x ONE quarter of K: its quarter of the weight image (KS <= 256 rows = 128 KB) is DMA'd into LDS once and stays
// there, every clip-feature element is loaded, dropped and split exactly once (a wave owns 16 rows x all 128 columns), and
// the four partial sums go to a [4][rows][128] slab that the layer-norm launch behind it adds up (ln_fwd_kernel, `part`).
// Per CU: 128 KB of features + 128 KB of weights in, 64 KB out - against 640 KB in for the same rows before.
#define FK_ROWS 128
__global__ __launch_bounds__(512) void feature_ksplit_kernel(FkBatch batch, DropCfg drop) {
  extern __shared__ float lds[];     // KS/64 stages of {hi tile, lo tile}
  char* ldsb = reinterpret_cast<char*>(lds);
  const FkJob& job = batch.j[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int q = blockIdx.y;
  const int M = job.M, K = job.K, KS = job.KS;
  if ((int)blockIdx.x * FK_ROWS >= M) return;          // block-uniform
  const int rowbase = blockIdx.x * FK_ROWS + wave * 16;
  const int arow = min(rowbase + j, M - 1);
  const bool adrop = job.drop_site >= 0 && drop.enabled;
  const uint32_t adrow = job.drop_row0 + (uint32_t)arow;
  const int nst = KS / 64;
  // the block's weight quarter: rows q*KS .. of the forward image (clamped to the last real row: the operand is zero
  // there), 32 one-KB pieces per stage
  {
    const char* img = reinterpret_cast<const char*>(job.Wimg);
    const int chp = lane & 15, rr = lane >> 4;
    for (int pc = wave; pc < 32 * nst; pc += 8) {
      const int st = pc >> 5, pl = pc & 31;
      const int r = 4 * (pl & 15) + rr;
      const int ch = chp ^ (((r & 3) << 2) | ((r >> 2) & 3));
      const int kk = min(q * KS + 64 * st + r, K - 1);
      const char* src = img + (size_t)kk * 512 + (pl >> 4) * 256 + 16 * ch;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(ldsb + st * GB_STAGE + pl * 1024), 16, 0, 0);
    }
  }
  f32x4 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* Ap = job.A + (size_t)arow * job.lda;
  const int nks = KS / 32;
  const uint16_t* Ap16 = reinterpret_cast<const uint16_t*>(job.A) + (size_t)arow * job.lda;
  const bool a16 = job.a_bf16 != 0;
  auto a_fetch = [&](int ks, float4& x0, float4& x1) {
    const int kk = q * KS + 32 * ks + 8 * g;
    if (a16) {                                          // bfloat16 features: 16 bytes = the lane's 8 values, widened exactly
      const uint4 raw = kk < K ? *reinterpret_cast<const uint4*>(Ap16 + kk) : make_uint4(0u, 0u, 0u, 0u);
      x0 = make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                       __uint_as_float(raw.y & 0xffff0000u));
      x1 = make_float4(__uint_as_float(raw.z << 16), __uint_as_float(raw.z & 0xffff0000u), __uint_as_float(raw.w << 16),
                       __uint_as_float(raw.w & 0xffff0000u));
      return;
    }
    x0 = kk < K ? ld4(Ap + kk) : f4zero();              // K is a multiple of 8: a lane's 8 values are in or out together
    x1 = kk < K ? ld4(Ap + kk + 4) : f4zero();
  };
  float4 c0, c1, n0 = f4zero(), n1 = f4zero();
  a_fetch(0, c0, c1);
  if (nks > 1) a_fetch(1, n0, n1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int tq = (lane >> 2) & 3, tp = lane & 3;
  for (int ks = 0; ks < nks; ++ks) {
    float4 f0 = f4zero(), f1 = f4zero();
    if (ks + 2 < nks) a_fetch(ks + 2, f0, f1);       // two k-steps ahead
    if (adrop) {
      const uint32_t cg = (uint32_t)((q * KS + 32 * ks + 8 * g) >> 2);
      if (job.keep_out) {                             // block-uniform: the bits go to the weight-gradient job (DwJob::a_keep)
        const uint32_t b0 = drop_bits4(drop, (uint32_t)job.drop_site, adrow, cg);
        const uint32_t b1 = drop_bits4(drop, (uint32_t)job.drop_site, adrow, cg + 1u);
        c0 = f4mul(c0, mask_from_bits4(b0, drop.scale));
        c1 = f4mul(c1, mask_from_bits4(b1, drop.scale));
        if (rowbase + j < M && (int)(4u * cg) < K)
          *reinterpret_cast<uint16_t*>(job.keep_out + (size_t)arow * job.ld_keep + cg) = (uint16_t)(b0 | (b1 << 8));
      } else {
        c0 = apply_drop4(drop, (uint32_t)job.drop_site, adrow, cg, c0);
        c1 = apply_drop4(drop, (uint32_t)job.drop_site, adrow, cg + 1u, c1);
      }
    }
    // f16x3 (bf16x3.h): the row scale is taken per 32-deep k-step (the features of a row arrive over the whole loop)
    float rmax = fmaxf(f4absmax(c0), f4absmax(c1));
    rmax = fmaxf(rmax, __shfl_xor(rmax, 16));
    rmax = fmaxf(rmax, __shfl_xor(rmax, 32));
    float inv;
    const float sc = f16_row_scale(rmax, inv);
    uint2 h0, l0, h1, l1;
    f16_split4(f4scale1(c0, sc), h0, l0);
    f16_split4(f4scale1(c1, sc), h1, l1);
    const f16x8 ah = __builtin_bit_cast(f16x8, (u32x4){h0.x, h0.y, h1.x, h1.y});
    const f16x8 al = __builtin_bit_cast(f16x8, (u32x4){l0.x, l0.y, l1.x, l1.y});
    const char* hi = ldsb + (ks >> 1) * GB_STAGE;
    const int r0 = 32 * (ks & 1) + 8 * g + tq, r1 = r0 + 4;
    float ir[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ir[r] = __shfl(inv, 4 * g + r);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int chunk = 2 * t + (tp >> 1);
      const int o0 = tile256_off(r0, chunk) + 8 * (tp & 1), o1 = tile256_off(r1, chunk) + 8 * (tp & 1);
      const f16x8 wh = join_tr_f16(lds_read_tr16(hi, o0), lds_read_tr16(hi, o1));
      const f16x8 wl = join_tr_f16(lds_read_tr16(hi + GB_TILE, o0), lds_read_tr16(hi + GB_TILE, o1));
      f32x4 p = (f32x4){0.f, 0.f, 0.f, 0.f};
      p = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, p, 0, 0, 0);
      p = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl, p, 0, 0, 0);
      p = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh, p, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = fmaf(p[r], ir[r], acc[t][r]);
    }
    c0 = n0; c1 = n1; n0 = f0; n1 = f1;
  }
  // accumulator tile t, lane (j, g), register r = row 4g + r, column 64 (t>>2) + 4j + (t&3)
  float* out = job.part + (size_t)q * job.part_stride;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = rowbase + 4 * g + r;
    if (row < M) {
      st4(out + (size_t)row * 128 + 4 * j, make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]));
      st4(out + (size_t)row * 128 + 64 + 4 * j, make_float4(acc[4][r], acc[5][r], acc[6][r], acc[7][r]));
    }
  }
}

// Chained launch: the jobs of the batch run ONE AFTER THE OTHER inside every block, on the block's own 16*RT rows.
// Valid when each job reads, of the tensors written earlier in the chain, only the rows of its own block (dense layers
// are row local), every job has the same M and N = 128: then a workgroup barrier between jobs is all the ordering
// needed (the block's stores are visible to its own waves).  Measured on MI355X (6 jobs of a dual attention block):
// 65 us chained vs 69 us as five launches - the jobs are bound by what a CU can pull in per clock (~12 B: the 64 KB
// weight image per layer and block dominates), not by the launch floor, so chaining buys little.
template <int RT>
__global__ __launch_bounds__(RT * 128) void gemm_chain_kernel(GemmBatch batch, DropCfg drop, int njobs) {
  extern __shared__ float lds[];
  // the job descriptors are read straight from the kernel-argument segment (constant address space, scalar loads):
  // indexing the by-value batch with the loop counter would make the compiler copy all of it to scratch memory
  typedef const __attribute__((address_space(4))) GemmJob CJob;
  CJob* jobs = (CJob*)__builtin_amdgcn_kernarg_segment_ptr();     // GemmBatch is the first argument
  for (int ji = 0; ji < njobs; ++ji) {
    if (ji) __syncthreads();
    gemm_bf16_body<2, RT>(jobs[ji], drop, reinterpret_cast<char*>(lds));
  }
}


// Pre-split weight images for gemm_bf16_kernel, made once per step (the weights are constant within a step).
// For every dense weight W [K,128] at float offset `off` of the flat parameter buffer:
//   forward image  at fwd + 4*off:  row k (512 B) = bf16 hi of W[k][perm(s)], s = 0..127 | the same for the residuals
//   backward image at bwd + boff:   block b (64 KB) = the forward-style image of the 128 x 128 matrix W[128b + c][kk]^T
//                                   (rows kk, columns c; zero where 128b + c >= K)
// perm(s) = 64 (s>>6) + 4 (s&15) + ((s>>4)&3): stored column 16 t + i of a 64-column half is original column 4 i + t.
struct PackJob { uint32_t off; int K; uint32_t boff; };
#define HUAL_MAX_PACK 96
struct PackBatch { PackJob j[HUAL_MAX_PACK]; };
__device__ __forceinline__ int pack_perm(int s) { return 64 * (s >> 6) + 4 * (s & 15) + ((s >> 4) & 3); }
__global__ __launch_bounds__(256) void pack_weights_kernel(PackBatch b, const float* P, char* fwd, char* bwd, int njobs, PackExtra ex) {
  __shared__ float tile[16][129];
  if ((int)blockIdx.y >= njobs) {      // the extra row: masks, loss accumulators, gradient zeroing (grid-stride)
    const int nt = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
    const int Nv = ex.B * ex.T, Nq = ex.B * ex.L;
    if (t0 < 8) ex.loss_acc[t0] = 0.f;
    for (int i = t0; i < Nv + Nq; i += nt)
      ex.rowmask[i] = i < Nv ? ((i % ex.T) < ex.lens[i / ex.T] ? 1.0f : 0.0f)       // tf.sequence_mask, model.py:31
                             : (ex.word_ids[i - Nv] != 0 ? 1.0f : 0.0f);            // model.py:32
    if (ex.zero_ptr)
      for (size_t i = t0; i < ex.zero_n / 4; i += nt) reinterpret_cast<float4*>(ex.zero_ptr)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const PackJob job = b.j[blockIdx.y];
  const int k0 = blockIdx.x * 16;
  if (k0 >= ((job.K + 127) & ~127)) return;          // block-uniform
  const float* W = P + job.off;
  for (int idx = threadIdx.x; idx < 16 * 128; idx += 256) {
    const int r = idx >> 7, n = idx & 127;
    tile[r][n] = (k0 + r) < job.K ? W[(size_t)(k0 + r) * 128 + n] : 0.f;
  }
  __syncthreads();
  // forward image: 16 rows x 64 column pairs
  if (fwd) {
    char* img = fwd + (size_t)job.off * 4;
    for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
      const int r = idx >> 6, sp = idx & 63;
      if (k0 + r < job.K) {
        uint32_t hi, lo;
        f16_split_pair(tile[r][pack_perm(2 * sp)] * HUAL_F16_WSCALE, tile[r][pack_perm(2 * sp + 1)] * HUAL_F16_WSCALE, hi, lo);
        *reinterpret_cast<uint32_t*>(img + (size_t)(k0 + r) * 512 + 4 * sp) = hi;
        *reinterpret_cast<uint32_t*>(img + (size_t)(k0 + r) * 512 + 256 + 4 * sp) = lo;
      }
    }
  }
  // backward image: the 16 original rows k0..k0+15 are columns c = k0 % 128 + (0..15) of block k0 / 128; for every kk
  // they sit at stored columns s with perm(s) = c, i.e. s = 64 (c>>6) + 16 t + i with 4 i + t = c & 63
  if (bwd) {
    char* img = bwd + job.boff + (size_t)(k0 >> 7) * GB_COLBLOCK;
    const int cbase = k0 & 127;
    for (int idx = threadIdx.x; idx < 128 * 8; idx += 256) {
      const int kk = idx >> 3, pr = idx & 7;          // pair pr: local rows r = 2*? -> stored columns come in pairs (s, s+1)
      // stored pair (s, s+1) with s even: perm(s) = c, perm(s+1) = c + 4  ->  local rows r and r + 4
      const int t = pr & 3, ii = pr >> 2;             // r = 4*(2*ii') ...: enumerate r in {0..15} with (r>>2)&1 == 0
      const int r = t + 8 * ii;                       // r in {0,1,2,3, 8,9,10,11}; partner r + 4
      const int c = cbase + r;
      const int s = 64 * (c >> 6) + 16 * (c & 3) + ((c & 63) >> 2);
      uint32_t hi, lo;
      f16_split_pair(tile[r][kk] * HUAL_F16_WSCALE, tile[r + 4][kk] * HUAL_F16_WSCALE, hi, lo);
      *reinterpret_cast<uint32_t*>(img + (size_t)kk * 512 + 2 * s) = hi;
      *reinterpret_cast<uint32_t*>(img + (size_t)kk * 512 + 256 + 2 * s) = lo;
    }
  }
}

// Variant of gemm_lds_kernel with the extended A prologue (only launched for jobs that use it: the extra operand
// registers and the longer dependency chain cost every launch about 1.5 us).
//
// A-operand prologue (what the MFMA consumes is f(A)), in this order:
//   A2 product -> layer norm over the 128 columns of piece 0 (ln_g) -> dropout (a_drop_site) -> relu mask (a_relu > 0)
//   -> optional store of the transformed operand (a_save: the dZ / LN output that later kernels need).
// This is how the elementwise kernels around the dense layers (LN, dropout, relu') disappear from the step.
template <bool DUAL, int RT>
__global__ __launch_bounds__(RT * 128) void gemm_lds_px_kernel(GemmBatch batch, DropCfg drop) {
  extern __shared__ float lds[];     // Ws[2][GL_STAGE] (+ W2s[2][GL_STAGE] when DUAL)
  const GemmJob& job = batch.j[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int M = job.M, N = job.N;
  const int blockrow = blockIdx.x * (16 * RT);
  const int nblk = blockIdx.y * 128;
  if (blockrow >= M || nblk >= N) return;          // block-uniform
  const int rowbase = blockrow + (wave >> 1) * 16;
  const int n0 = nblk + (wave & 1) * 64;
  const bool wave_on = rowbase < M && n0 < N;       // idle waves still take part in staging and barriers
  const int arow = min(rowbase + j, M - 1);
  const bool arow_ok = (rowbase + j) < M;
  const int ldw = job.ldw;
  const bool adrop = job.a_drop_site >= 0 && drop.enabled;
  const uint32_t asite = (uint32_t)job.a_drop_site;
  const uint32_t adrow = job.a_drop_row0 + (uint32_t)arow;
  const bool saver = job.a_save != nullptr && (wave & 1) == 0 && blockIdx.y == 0 && arow_ok;
  float* Ws = lds;
  float* W2s = lds + 2 * GL_STAGE;
  int nstages = 0;
  for (int p = 0; p < job.npieces; ++p) nstages += (job.kw[p] + GL_KS - 1) / GL_KS;

  f32x4 acc[4], acc2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // LDS-DMA of one stage: 32 pieces of 1 KB (2 panel rows each); lane l -> row 2*pc + (l>>5), 16 B at column 4*(l&31).
  // Out-of-range rows / columns are clamped to valid memory (never multiplied / never stored).
  auto dma_stage = [&](const float* Wp, int k0, int kw, float* dst) {
    const int c4 = lane & 31, rr = lane >> 5;
    const int n = min(nblk + 4 * c4, N - 4);
    for (int pc = wave; pc < 32; pc += 2 * RT) {
      const int kk = min(k0 + 2 * pc + rr, kw - 1);
      const float* src = Wp + (size_t)kk * ldw + n;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + pc * 256), 16, 0, 0);
    }
  };
  struct Stage { int p, k0, kw; };
  // request stage s: weight panel(s) by DMA into buffer (s & 1); raw A fragments (+ the multiplier / mask operand)
  auto issue = [&](int s, float4 (&a)[4], float4 (&x)[4], float4 (&a2)[4], Stage& st) {
    stage_to_piece(job, s, st.p, st.k0);
    const int p = st.p, k0 = st.k0;
    const int kw = job.kw[p];
    st.kw = kw;
    dma_stage(job.W[p], k0, kw, Ws + (s & 1) * GL_STAGE);
    if (DUAL) dma_stage(job.W2[p], k0, kw, W2s + (s & 1) * GL_STAGE);
    const float* Ap = job.A[p] + (size_t)arow * job.lda[p];
    const float* Xp = job.A2[p] ? job.A2[p] + (size_t)arow * job.lda2[p]
                                : ((p == 0 && job.a_relu) ? job.a_relu + (size_t)arow * job.lda_relu : nullptr);
    const float* Abp = (DUAL && job.Ab[p]) ? job.Ab[p] + (size_t)arow * job.ldab[p] : nullptr;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const int kk = k0 + kc * 16;
      const bool ok = kk < kw;
      a[kc] = ok ? ld4(Ap + kk + 4 * g) : f4zero();
      x[kc] = (ok && Xp) ? ld4(Xp + kk + 4 * g) : f4zero();
      if (DUAL) a2[kc] = (ok && Abp) ? ld4(Abp + kk + 4 * g) : f4zero();
    }
  };
  // everything of the prologue except the layer norm, for one stage of fragments
  auto finish = [&](float4 (&a)[4], const float4 (&x)[4], float4 (&a2)[4], const Stage& st) {
    const int p = st.p;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const int kk = st.k0 + kc * 16;
      if (kk >= st.kw) continue;
      float4 v = a[kc];
      if (job.A2[p]) v = f4mul(v, x[kc]);
      if (adrop) v = apply_drop4(drop, asite, adrow, (uint32_t)((kk + 4 * g) >> 2), v);
      if (p == 0 && job.a_relu) {
        const float4 y = x[kc];
        v = make_float4(y.x > 0.f ? v.x : 0.f, y.y > 0.f ? v.y : 0.f, y.z > 0.f ? v.z : 0.f, y.w > 0.f ? v.w : 0.f);
      }
      if (p == 0 && saver) st4(job.a_save + (size_t)arow * job.lda_save + kk + 4 * g, v);
      a[kc] = v;
      if (DUAL && !job.Ab[p]) a2[kc] = v;
    }
  };
  auto compute = [&](int bufi, const Stage& st, const float4 (&a)[4], const float4 (&a2)[4]) {
    const float* wsb = Ws + bufi * GL_STAGE + (wave & 1) * 64 + 4 * j;
    const float* w2b = W2s + bufi * GL_STAGE + (wave & 1) * 64 + 4 * j;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      if (st.k0 + kc * 16 < st.kw) {
        float4 b[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const float4*>(wsb + (kc * 16 + 4 * g + c) * 128);
        mma_frag(acc, a[kc], b, 0);
        if (DUAL) {
#pragma unroll
          for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const float4*>(w2b + (kc * 16 + 4 * g + c) * 128);
          mma_frag(acc2, a2[kc], b, 0);
        }
      }
    }
  };

  float4 a0[4], a1[4], x0[4], x1[4], c0[4], c1[4];
  Stage s0{0, 0, 0}, s1{0, 0, 0};
  EpiRegs epi;
  // prologue: the first TWO stages (a whole K=128 layer) + everything the epilogue will read
  issue(0, a0, x0, c0, s0);
  if (nstages > 1) issue(1, a1, x1, c1, s1);
  epi_prefetch<DUAL>(job, epi, rowbase, n0, j, g);
  if (job.ln_g) {
    // layer norm of piece 0 (kw[0] == 128 => exactly stages 0 and 1).  Row j of the tile is spread over the 4 lanes
    // (j, g = 0..3): in-lane sums + two shuffles give mean / biased variance (models/layers.py:13-15).
    float sum = 0.f;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) sum += (a0[kc].x + a0[kc].y) + (a0[kc].z + a0[kc].w) + (a1[kc].x + a1[kc].y) + (a1[kc].z + a1[kc].w);
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / 128.0f);
    float sq = 0.f;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      float4 d = make_float4(a0[kc].x - mean, a0[kc].y - mean, a0[kc].z - mean, a0[kc].w - mean);
      sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
      d = make_float4(a1[kc].x - mean, a1[kc].y - mean, a1[kc].z - mean, a1[kc].w - mean);
      sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq * (1.0f / 128.0f) + 1e-6f);
    if (job.ln_mean && g == 0 && (wave & 1) == 0 && blockIdx.y == 0 && arow_ok) {
      job.ln_mean[arow] = mean;
      job.ln_rstd[arow] = rstd;
    }
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const float4 g0 = ld4(job.ln_g + kc * 16 + 4 * g), b0 = ld4(job.ln_b + kc * 16 + 4 * g);
      const float4 g1 = ld4(job.ln_g + 64 + kc * 16 + 4 * g), b1 = ld4(job.ln_b + 64 + kc * 16 + 4 * g);
      a0[kc] = make_float4((a0[kc].x - mean) * rstd * g0.x + b0.x, (a0[kc].y - mean) * rstd * g0.y + b0.y,
                           (a0[kc].z - mean) * rstd * g0.z + b0.z, (a0[kc].w - mean) * rstd * g0.w + b0.w);
      a1[kc] = make_float4((a1[kc].x - mean) * rstd * g1.x + b1.x, (a1[kc].y - mean) * rstd * g1.y + b1.y,
                           (a1[kc].z - mean) * rstd * g1.z + b1.z, (a1[kc].w - mean) * rstd * g1.w + b1.w);
    }
  }
  finish(a0, x0, c0, s0);
  if (nstages > 1) finish(a1, x1, c1, s1);
  for (int s = 0; s < nstages; s += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // DMA of the resident stages has landed (this wave's part)
    __syncthreads();                                      // ... and everybody else's
    if (wave_on) compute(0, s0, a0, c0);
    if (s + 1 < nstages && wave_on) compute(1, s1, a1, c1);
    if (s + 2 < nstages) {
      __syncthreads();                                    // both buffers consumed by every wave
      issue(s + 2, a0, x0, c0, s0);
      if (s + 3 < nstages) issue(s + 3, a1, x1, c1, s1);
      finish(a0, x0, c0, s0);
      if (s + 3 < nstages) finish(a1, x1, c1, s1);
    }
  }
  if (!wave_on) return;
  epi_apply<DUAL>(job, drop, epi, acc, acc2, rowbase, n0, j, g);
}

// WT[n][k] = W[k][n] for a table of dense weights inside the flat parameter buffer (same offsets in `dst`).
struct TrJob { uint32_t off; int K, N; };
#define HUAL_MAX_TR 96
struct TrBatch { TrJob j[HUAL_MAX_TR]; };
__global__ __launch_bounds__(256) void transpose_weights_kernel(TrBatch b, const float* src, float* dst) {
  __shared__ float tile[32][33];
  const TrJob job = b.j[blockIdx.z];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  if (k0 >= job.K || n0 >= job.N) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* W = src + job.off;
  float* WT = dst + job.off;
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, n = n0 + tx;
    tile[r][tx] = (k < job.K && n < job.N) ? W[(size_t)k * job.N + n] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, k = k0 + tx;
    if (n < job.N && k < job.K) WT[(size_t)n * job.K + k] = tile[tx][r];
  }
}

// ------------------------------------------------------------------------------------------------------
// dW / db.  A block owns one 128(k) x 128(n) gradient tile of one job over a chunk of rows_per_block rows; its four
// waves own the four 64 x 64 quadrants.  Rows of A (with the job's prologue: elementwise product, dropout) and of dY
// stream through LDS in 32-row tiles, loaded ONCE with coalesced 16-byte loads and double buffered, so every
// activation / gradient element is read once per job from HBM.  The MFMA is v_mfma_f32_32x32x2_f32 with the row
// index m as its k dimension; accumulators leave as float atomics whose wave-instructions are two contiguous
// 128-byte row segments (the full-rate shape of MI355X_MICROARCH.md "Global float atomics").
#define DW_TM 32            // rows per LDS tile
#define DW_LD 132           // padded leading dimension (floats)
// copies a slice of job descriptors (passed by value, so graph-capture safe) into the device-resident job table
// and, behind the n descriptors, what the balanced launch needs per job: tiles in front of it, cost in front of it and
// the cost of one of its tiles (dw_plan)
struct DwPlanPart { int tiles[HUAL_MAX_DW_JOBS + 1], cost[HUAL_MAX_DW_JOBS + 1], w[HUAL_MAX_DW_JOBS + 1]; };
struct DwPlan { int* tiles; int* cost; int* w; };      // (n + 1) entries each
__device__ __host__ inline DwPlan dw_plan(const DwJob* table, int n) {
  int* base = const_cast<int*>(reinterpret_cast<const int*>(table + n));
  return DwPlan{base, base + (n + 1), base + 2 * (n + 1)};
}
__global__ void dw_table_write_kernel(DwBatch part, DwPlanPart pre, DwJob* table, int base, int cnt, int n) {
  const int t = threadIdx.x;
  if (t < cnt) table[base + t] = part.j[t];
  if (t <= cnt) {
    const DwPlan pl = dw_plan(table, n);
    pl.tiles[base + t] = pre.tiles[t];
    pl.cost[base + t] = pre.cost[t];
    pl.w[base + t] = pre.w[t];
  }
}

template <bool FROM_TABLE>
__global__ __launch_bounds__(256) void dw_kernel(DwBatch batch, const DwJob* __restrict__ table, DropCfg drop,
                                                 int rows_per_block) {
  extern __shared__ float lds[];     // As[2][DW_TM][DW_LD] | Ys[2][DW_TM][DW_LD]
  const DwJob& job = FROM_TABLE ? table[blockIdx.z] : batch.j[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int kq = wave >> 1, nq = wave & 1;
  const int M = job.M;
  // decode k-block -> (piece, offset inside the piece)
  int p = 0, kb = blockIdx.y;
  for (p = 0; p < job.npieces; ++p) {
    const int nkb = (job.kw[p] + 127) >> 7;
    if (kb < nkb) break;
    kb -= nkb;
  }
  if (p >= job.npieces) return;                 // block-uniform
  const int m_lo = blockIdx.x * rows_per_block;
  if (m_lo >= M) return;                        // block-uniform
  const int m_hi = min(m_lo + rows_per_block, M);
  const int k0 = kb * 128;
  const int kw = job.kw[p];
  const float* Ap = job.A[p];
  const float* A2p = job.A2[p];
  const int lda = job.lda[p], lda2 = job.lda2[p];
  const float* Yp = job.dY;
  const int ldy = job.ldy;
  const bool adrop = job.a_drop_site >= 0 && drop.enabled;
  const uint32_t asite = (uint32_t)job.a_drop_site;
  const uint32_t arow0 = job.a_drop_row0;
  float* As = lds;
  float* Ys = lds + 2 * DW_TM * DW_LD;

  f32x16 acc[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;
  float bsum0 = 0.f, bsum1 = 0.f;
  const bool dob = (job.db != nullptr) && p == 0 && kb == 0 && kq == 0;

  // staging: 32 rows x 32 float4 per matrix = 1024 float4 -> 4 per thread per matrix
  float4 ra[4], ry[4];
  auto stage_load = [&](int mt) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = threadIdx.x + 256 * u;
      const int row = idx >> 5, c4 = idx & 31;
      const int m = mt + row;
      float4 a = f4zero(), y = f4zero();
      if (m < m_hi) {
        const int k = k0 + 4 * c4;
        if (k < kw) {
          if (job.a_bf16) {
            const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(Ap) + (size_t)m * lda + k);
            a = make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                            __uint_as_float(raw.y & 0xffff0000u));
          } else {
            a = ld4(Ap + (size_t)m * lda + k);
          }
          if (A2p) a = f4mul(a, ld4(A2p + (size_t)m * lda2 + k));
          if (adrop) a = apply_drop4(drop, asite, arow0 + (uint32_t)m, (uint32_t)(k >> 2), a);
        }
        y = ld4(Yp + (size_t)m * ldy + 4 * c4);
      }
      ra[u] = a;
      ry[u] = y;
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = threadIdx.x + 256 * u;
      const int row = idx >> 5, c4 = idx & 31;
      *reinterpret_cast<float4*>(As + (buf * DW_TM + row) * DW_LD + 4 * c4) = ra[u];
      *reinterpret_cast<float4*>(Ys + (buf * DW_TM + row) * DW_LD + 4 * c4) = ry[u];
    }
  };

  stage_load(m_lo);
  stage_store(0);
  __syncthreads();
  int buf = 0;
  for (int mt = m_lo; mt < m_hi; mt += DW_TM) {
    const bool more = (mt + DW_TM) < m_hi;
    if (more) stage_load(mt + DW_TM);            // global loads in flight under the MFMAs below
    const float* At = As + buf * DW_TM * DW_LD + kq * 64 + 2 * i;
    const float* Yt = Ys + buf * DW_TM * DW_LD + nq * 64 + i;
#pragma unroll 4
    for (int r = 0; r < DW_TM; r += 2) {
      const float2 a = *reinterpret_cast<const float2*>(At + (r + h) * DW_LD);
      const float b0 = Yt[(r + h) * DW_LD];
      const float b1 = Yt[(r + h) * DW_LD + 32];
      acc[0][0] = mfma32(a.x, b0, acc[0][0]);
      acc[0][1] = mfma32(a.x, b1, acc[0][1]);
      acc[1][0] = mfma32(a.y, b0, acc[1][0]);
      acc[1][1] = mfma32(a.y, b1, acc[1][1]);
      bsum0 += b0;
      bsum1 += b1;
    }
    if (more) stage_store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // each wave owns its quadrant: no cross-wave reduction, straight to global atomics
  float* dWp = job.dW[p];
  const int kbase = k0 + kq * 64, nbase = nq * 64;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowi = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = kbase + 2 * rowi + c;
        const int n = nbase + 32 * t + i;
        if (k < kw) atomicAdd(dWp + (size_t)k * job.ldw + n, acc[c][t][r]);
      }
  if (dob) {
    bsum0 += __shfl_xor(bsum0, 32);
    bsum1 += __shfl_xor(bsum1, 32);
    if (h == 0) {
      atomicAdd(job.db + nbase + i, bsum0);
      atomicAdd(job.db + nbase + 32 + i, bsum1);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// dw_bf16_kernel: the same job semantics, tiling (128 x 128 gradient tile per block over a chunk of rows, one 64 x 64
// quadrant per wave) and atomics epilogue as dw_kernel, with the products on the bf16 matrix cores as three passes of
// split operands (bf16x3.h).  The reduction index of dW = A^T.dY is the ROW index m, which is the strided direction of
// both operands in memory: the 32-row tiles are staged row-major as bf16 (hi and lo planes, 256-byte rows, XOR swizzled)
// and read back with ds_read_b64_tr_b16, the LDS transpose read, so each lane receives 8 consecutive rows of its column.
// Per 32-row tile and wave: 24 MFMAs (768 cycles) instead of 64 (4096 cycles); the kernel is then bound by streaming
// the operands from HBM (every job reads its A and dY once).
#define DWB_PLANE (64 * 256)             // bytes of one [DWB_TM][128] bf16 plane
// one segment: rows [m_lo, m_hi) of k-block kb of piece p of a job -> atomics into its 128 x 128 gradient tile.
// 512 threads: wave (kq, nq) owns the 64 x 32 block of gradient rows 64kq.., columns 32nq.. (two 32x32 accumulators,
// <= 128 registers per lane, so two workgroups share a CU).  The launch is bound by how many bytes a CU keeps in flight
// (one workgroup per CU runs 1.7x longer than two), hence two register stages: the loads of tile t+2 are issued before
// the products of tile t and consumed (split, stored to LDS) at the end of iteration t+1.
#define DWB_THREADS 512
#define DWB_TM 64             // rows per LDS tile: one workgroup barrier per 64 rows
#define DWB_RU (DWB_TM / 16)  // rows a thread stages per tile
// MODE: DWB_PLAIN fp32 A without a prologue (all but a handful of jobs); DWB_PROD product prologue A[p] * A2[p] (its
// second operand is staged like the first: DEPTH 2 keeps the registers below 256); DWB_DROP dropout prologue (keep bits
// from the forward, or the Philox rounds) and / or bfloat16 A
#define DWB_PLAIN 0
#define DWB_PROD 1
#define DWB_DROP 2
template <int MODE, int DEPTH>
__device__ __forceinline__ void dw_bf16_segment(const DwJob& job, const int p, const int kb, const int m_lo, const int m_hi,
                                                const DropCfg& drop, char* ldsb, float4 (*bred)[32], const int dbg = 0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int kq = wave >> 2, nq = wave & 3;
  const int k0 = kb * 128;
  const int kw = job.kw[p];
  const char* Ap = reinterpret_cast<const char*>(job.A[p]);
  const char* A2p = MODE == DWB_PROD ? reinterpret_cast<const char*>(job.A2[p]) : nullptr;
  const int lda = job.lda[p], lda2 = job.lda2[p];
  const char* Yp = reinterpret_cast<const char*>(job.dY);
  const int ldy = job.ldy;
  constexpr bool PLAIN = MODE == DWB_PLAIN;
  const bool adrop = MODE == DWB_DROP && job.a_drop_site >= 0 && drop.enabled;
  const uint32_t asite = (uint32_t)job.a_drop_site;
  const uint32_t arow0 = job.a_drop_row0;
  const bool dob = (job.db != nullptr) && p == 0 && kb == 0;
  const bool abf = MODE == DWB_DROP && job.a_bf16 != 0;

  f32x16 acc[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float4 bsum = f4zero();

  // staging: thread -> rows (tid>>5) + 16u, columns 4*(tid&31)..+3 of both tiles; prologues (bf16 widening, product,
  // dropout) are applied at the store, so that a load is only waited for one iteration after its issue.  Addresses are
  // a uniform base per tile plus a 32-bit byte offset per thread (one register each for A and dY).
  const int srow = threadIdx.x >> 5, c4 = threadIdx.x & 31;
  const int kcol = k0 + 4 * c4;
  const bool kin = kcol < kw;
  const uint32_t esz = abf ? 2u : 4u;
  const uint32_t aoff = ((uint32_t)srow * (uint32_t)lda + (uint32_t)kcol) * esz;
  const uint32_t yoff = ((uint32_t)srow * (uint32_t)ldy + 4u * (uint32_t)c4) * 4u;
  struct Stage { float4 a[DWB_RU], y[DWB_RU], a2[MODE == DWB_PROD ? DWB_RU : 1]; uint32_t keep[DWB_RU]; };
  const uint32_t a2off = ((uint32_t)srow * (uint32_t)lda2 + (uint32_t)kcol) * 4u;
  const uint8_t* keepp = MODE == DWB_DROP ? job.a_keep : nullptr;
  const uint32_t koff = (uint32_t)srow * (uint32_t)job.ld_keep + (uint32_t)(kcol >> 2);
  Stage st[DEPTH - 1];      // tile i in LDS, tiles i+1 .. i+DEPTH-1 in (or on their way to) registers
  auto stage_load = [&](int mt, Stage& st) {
    const char* Ab = Ap + (size_t)mt * lda * esz;
    const char* Yb = Yp + (size_t)mt * ldy * 4;
#pragma unroll
    for (int u = 0; u < DWB_RU; ++u) {
      const int m = mt + srow + 16 * u;
      float4 a = f4zero(), y = f4zero();
      if (m < m_hi && !(dbg & 2)) {
        if (kin) {
          if (abf) {
            const uint2 raw = ld2_global(Ab + 16u * u * (uint32_t)lda * 2u + aoff);
            a.x = __uint_as_float(raw.x);
            a.y = __uint_as_float(raw.y);
          } else {
            a = ld4_global(Ab + 16u * u * (uint32_t)lda * 4u + aoff);
          }
          if (MODE == DWB_PROD && A2p) st.a2[u] = ld4_global(A2p + (size_t)mt * lda2 * 4 + 16u * u * (uint32_t)lda2 * 4u + a2off);
        }
        y = ld4_global(Yb + 16u * u * (uint32_t)ldy * 4u + yoff);
        if (MODE == DWB_DROP && keepp && kin) st.keep[u] = ld1_global(keepp + (size_t)mt * job.ld_keep + 16u * u * (uint32_t)job.ld_keep + koff);
      }
      st.a[u] = a;
      st.y[u] = y;
    }
  };
  auto stage_store = [&](int buf, int mt, const Stage& st) {
    char* base = ldsb + buf * 4 * DWB_PLANE;
#pragma unroll
    for (int u = 0; u < DWB_RU; ++u) {
      const int row = srow + 16 * u;
      const int off = tile256_off(row, c4 >> 1) + 8 * (c4 & 1);
      float4 a = st.a[u];
      if (!PLAIN) {
        const int m = mt + row;
        const bool live = m < m_hi && kin;
        if (abf) {
          const uint32_t r0 = __float_as_uint(a.x), r1 = __float_as_uint(a.y);
          a = make_float4(__uint_as_float(r0 << 16), __uint_as_float(r0 & 0xffff0000u), __uint_as_float(r1 << 16),
                          __uint_as_float(r1 & 0xffff0000u));
        }
        if (MODE == DWB_PROD && A2p && live) a = f4mul(a, st.a2[u]);
        if (adrop && live) {
          if (keepp) a = f4mul(a, mask_from_bits4(st.keep[u], drop.scale));
          else a = apply_drop4(drop, asite, arow0 + (uint32_t)m, (uint32_t)(kcol >> 2), a);
        }
      }
      uint2 hi, lo;
      bf16_split4(a, hi, lo);
      *reinterpret_cast<uint2*>(base + off) = hi;
      *reinterpret_cast<uint2*>(base + DWB_PLANE + off) = lo;
      bf16_split4(st.y[u], hi, lo);
      *reinterpret_cast<uint2*>(base + 2 * DWB_PLANE + off) = hi;
      *reinterpret_cast<uint2*>(base + 3 * DWB_PLANE + off) = lo;
      bsum = f4add(bsum, st.y[u]);
    }
  };
  // transposed-read addressing: 16-lane group g' = lane>>4 takes the 4-row x 16-column block of columns
  // colbase + 16*(g'&1), rows 16*ks + 8*(g'>>1) + 4*rr; lane 4q+pp of the group supplies row q, columns 4pp..4pp+3
  const int gq = (lane >> 2) & 3, gp = lane & 3, ghalf = (lane >> 4) & 1, gh = lane >> 5;
  auto tr_off = [&](int colbase, int ks, int rr) {
    const int row = 16 * ks + 8 * gh + 4 * rr + gq;
    const int ch = ((colbase + 16 * ghalf) >> 3) + (gp >> 1);
    return tile256_off(row, ch) + 8 * (gp & 1);
  };
  auto products = [&](int buf) {
    const char* base = ldsb + buf * 4 * DWB_PLANE;
#pragma unroll
    for (int ks = 0; ks < DWB_TM / 16; ++ks) {
      const int y0 = tr_off(nq * 32, ks, 0), y1 = tr_off(nq * 32, ks, 1);
      const bf16x8 yh = join_tr(lds_read_tr16(base + 2 * DWB_PLANE, y0), lds_read_tr16(base + 2 * DWB_PLANE, y1));
      const bf16x8 yl = join_tr(lds_read_tr16(base + 3 * DWB_PLANE, y0), lds_read_tr16(base + 3 * DWB_PLANE, y1));
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int o0 = tr_off(kq * 64 + 32 * c, ks, 0), o1 = tr_off(kq * 64 + 32 * c, ks, 1);
        const bf16x8 ah = join_tr(lds_read_tr16(base, o0), lds_read_tr16(base, o1));
        const bf16x8 al = join_tr(lds_read_tr16(base + DWB_PLANE, o0), lds_read_tr16(base + DWB_PLANE, o1));
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, yh, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, yl, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, yh, acc[c], 0, 0, 0);
      }
    }
  };

  // tile i sits in LDS buffer i&1 while the registers hold tiles i+1 .. i+DEPTH-1 (the last of them just issued)
#pragma unroll
  for (int q = 0; q < DEPTH - 1; ++q) stage_load(m_lo + q * DWB_TM, st[q]);     // rows at or beyond m_hi load nothing
  stage_store(0, m_lo, st[0]);
  __syncthreads();
  int buf = 0, mt = m_lo;
  // one tile: issue the loads of tile i+DEPTH-1 into `in` (the stage tile i left), multiply tile i, move tile i+1 from
  // `out` to the other buffer
  auto step = [&](Stage& in, const Stage& out) -> bool {
    stage_load(mt + (DEPTH - 1) * DWB_TM, in);
    if (!(dbg & 1)) products(buf);
    if (mt + DWB_TM >= m_hi) return true;      // block-uniform
    stage_store(buf ^ 1, mt + DWB_TM, out);
    if (!(dbg & 16)) __syncthreads();
    mt += DWB_TM;
    buf ^= 1;
    return false;
  };
  static_assert(DEPTH == 2 || DEPTH == 4, "the rotations below are written out for one and three stages");
  if (DEPTH == 2) {
    for (;;)
      if (step(st[0], st[0])) break;
  } else {
    for (;;) {       // st[0] held tile i (now in LDS) and receives tile i+3; st[1] holds tile i+1
      if (step(st[0], st[1])) break;
      if (step(st[1], st[2])) break;
      if (step(st[2], st[0])) break;
    }
  }

  // each wave owns its block: no cross-wave reduction, straight to global atomics.  Accumulator register r of lane
  // (i, h) is gradient row 32c + (r&3) + 8*(r>>2) + 4h, column i of the block.
  float* dWp = job.dW[p];
  const int kbase = k0 + kq * 64, n = nq * 32 + i;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = kbase + 32 * c + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (k < kw && !(dbg & 4)) atomic_add_global(dWp + (size_t)k * job.ldw + n, acc[c][r]);
    }
  if (dob) {                                   // block-uniform
    bred[srow][c4] = bsum;
    __syncthreads();
    if (threadIdx.x < 128) {
      const int cc = threadIdx.x;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += reinterpret_cast<const float*>(&bred[k][cc >> 2])[cc & 3];
      atomic_add_global(job.db + cc, s);
    }
  }
}

// block-uniform choice of the variant (launch_dw rejects a product prologue together with dropout / bfloat16)
__device__ __forceinline__ void dw_bf16_any_segment(const DwJob& job, const int p, const int kb, const int m_lo, const int m_hi,
                                                    const DropCfg& drop, char* ldsb, float4 (*bred)[32], const int dbg) {
  if (job.A2[p]) dw_bf16_segment<DWB_PROD, 2>(job, p, kb, m_lo, m_hi, drop, ldsb, bred, dbg);
  else if (job.a_bf16 || job.a_drop_site >= 0) dw_bf16_segment<DWB_DROP, 4>(job, p, kb, m_lo, m_hi, drop, ldsb, bred, dbg);
  else dw_bf16_segment<DWB_PLAIN, 4>(job, p, kb, m_lo, m_hi, drop, ldsb, bred, dbg);
}

// grid = (row chunks, k-blocks, jobs): one segment per block
template <bool FROM_TABLE>
__global__ __launch_bounds__(DWB_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void dw_bf16_kernel(DwBatch batch, const DwJob* __restrict__ table, DropCfg drop, int rows_per_block) {
  extern __shared__ float lds[];     // 2 buffers x {A_hi, A_lo, Y_hi, Y_lo} planes
  __shared__ float4 bred[16][32];
  const DwJob& job = FROM_TABLE ? table[blockIdx.z] : batch.j[blockIdx.z];
  int p = 0, kb = blockIdx.y;
  for (p = 0; p < job.npieces; ++p) {
    const int nkb = (job.kw[p] + 127) >> 7;
    if (kb < nkb) break;
    kb -= nkb;
  }
  if (p >= job.npieces) return;                 // block-uniform
  const int m_lo = blockIdx.x * rows_per_block;
  if (m_lo >= job.M) return;                    // block-uniform
  dw_bf16_any_segment(job, p, kb, m_lo, min(m_lo + rows_per_block, job.M), drop, reinterpret_cast<char*>(lds), bred, 0);
}

// Balanced launch: the 64-row tiles of all (job, piece, k-block) units form one list (job-major; plan.tiles[j] = tiles in
// front of job j) that is cut into gridDim.x runs of equal COST, one per workgroup (one workgroup per CU); a workgroup
// walks its run segment by segment and flushes its accumulators with atomics whenever the unit changes.  The cost of a
// tile depends on its job's prologue (strided clip features + keep bytes, product operands), and every unit carries a
// fixed cost in front of its first tile (DW_UNIT_COST: the flush, the pipeline refill): plan.cost[j] = cost in front of
// job j, plan.w[j] = cost per tile.  256 workgroups x 58 jobs: ~340 flushes of 64 KB instead of ~1500 with a fixed row
// split (float atomics run at 1.3 TB/s chip-wide), no round quantisation, no tail.
#ifdef HUAL_STAMPS
// debug: per-workgroup clock stamps of the balanced launch (scripts/exp/dw_stamps.py)
#define DW_STAMP_SLOTS 16
__device__ unsigned long long g_dw_stamps[2048 * DW_STAMP_SLOTS];
extern "C" int hual_debug_dw_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dw_stamps), sizeof(unsigned long long) * (size_t)n);
}
#define DW_STAMP(i, v) do { if (threadIdx.x == 0 && blockIdx.x < 2048 && (i) < DW_STAMP_SLOTS) g_dw_stamps[blockIdx.x * DW_STAMP_SLOTS + (i)] = (v); } while (0)
#else
#define DW_STAMP(i, v) do { } while (0)
#endif
#define DW_UNIT_COST 72
__global__ __launch_bounds__(DWB_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void dw_bf16_balanced_kernel(const DwJob* __restrict__ table, int n, DropCfg drop, int dbg) {
  extern __shared__ float lds[];
  __shared__ float4 bred[16][32];
  int* s_job = reinterpret_cast<int*>(&bred[0][0]);     // (no further static array: the planes must stay 16-byte aligned)
  DW_STAMP(0, __builtin_readcyclecounter());
  const DwPlan pl = dw_plan(table, n);
  const int total_cost = pl.cost[n];
  // run boundaries in cost space -> global tile index
  int cb[2];
  cb[0] = (int)(((long)blockIdx.x * total_cost) / gridDim.x);
  cb[1] = (int)(((long)(blockIdx.x + 1) * total_cost) / gridDim.x);
  for (int j = threadIdx.x; j < n; j += DWB_THREADS)
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (pl.cost[j] <= cb[e] && cb[e] < pl.cost[j + 1]) s_job[e] = j;
  __syncthreads();
  int tb[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    if (cb[e] >= total_cost) { tb[e] = pl.tiles[n]; continue; }
    const int j = s_job[e];
    const int per_unit = (table[j].M + DWB_TM - 1) / DWB_TM;
    const int w = pl.w[j];
    const int ucost = DW_UNIT_COST + per_unit * w;
    const int local = cb[e] - pl.cost[j];
    const int unit = local / ucost;
    const int r = local - unit * ucost - DW_UNIT_COST;
    tb[e] = pl.tiles[j] + unit * per_unit + (r > 0 ? r / w : 0);
  }
  int t = tb[0];
  const int t_end = tb[1];
  int j = s_job[0];
  __syncthreads();                              // s_job lives in bred
  if (t >= t_end) return;                       // block-uniform
  const int* prefix = pl.tiles;
  DW_STAMP(1, __builtin_readcyclecounter());
  DW_STAMP(3, (unsigned long long)(t_end - t));
  DW_STAMP(5, (unsigned long long)j);
  int nseg = 0, nnp = 0;
  while (t < t_end) {
    const DwJob& job = table[j];
    const int per_unit = (job.M + DWB_TM - 1) / DWB_TM;
    const int local = t - prefix[j];
    int unit = local / per_unit;
    const int tile = local - unit * per_unit;
    const int cnt = min(per_unit - tile, t_end - t);
    int p = 0;
    for (p = 0; p < job.npieces; ++p) {
      const int nkb = (job.kw[p] + 127) >> 7;
      if (unit < nkb) break;
      unit -= nkb;
    }
    const bool plain = !job.a_bf16 && !job.A2[p] && job.a_drop_site < 0;      // (stamps)
    dw_bf16_any_segment(job, p, unit, tile * DWB_TM, min((tile + cnt) * DWB_TM, job.M), drop, reinterpret_cast<char*>(lds), bred, dbg);
    t += cnt;
    if (t >= prefix[j + 1]) ++j;
    __syncthreads();                            // LDS planes and bred are reused by the next segment
    DW_STAMP(7 + nseg, __builtin_readcyclecounter());
    ++nseg;
    nnp += plain ? 0 : 1;
  }
  DW_STAMP(2, __builtin_readcyclecounter());
  DW_STAMP(4, (unsigned long long)nseg);
  DW_STAMP(6, (unsigned long long)nnp);
}

namespace hual {

int launch_transpose_weights(const uint32_t* offs, const int* Ks, const int* Ns, int n, const float* src, float* dst, hipStream_t stream) {
  HUAL_REQUIRE(n >= 0 && n <= HUAL_MAX_TR, "transpose: too many weights");
  if (n == 0) return 0;
  TrBatch b;
  int maxK = 0, maxN = 0;
  for (int i = 0; i < n; ++i) {
    b.j[i].off = offs[i]; b.j[i].K = Ks[i]; b.j[i].N = Ns[i];
    maxK = Ks[i] > maxK ? Ks[i] : maxK;
    maxN = Ns[i] > maxN ? Ns[i] : maxN;
  }
  HUAL_LAUNCH(0.0, 0.0, transpose_weights_kernel, dim3(cdiv(maxK, 32), cdiv(maxN, 32), n), dim3(256), 0, stream, b, src, dst);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_gemm(const GemmJob* jobs, int n, const DropCfg& drop, hipStream_t stream) {
  HUAL_REQUIRE(n >= 1 && n <= HUAL_MAX_JOBS, "launch_gemm: job count");
  GemmBatch b;
  int maxM = 0, maxN = 0;
  bool dual = false;
  for (int i = 0; i < n; ++i) {
    const GemmJob& j = jobs[i];
    HUAL_REQUIRE(j.M > 0 && j.N >= 4 && (j.N % 4) == 0, "launch_gemm: N must be a positive multiple of 4");
    HUAL_REQUIRE(j.npieces >= 1 && j.npieces <= HUAL_MAX_PIECES, "launch_gemm: pieces");
    for (int p = 0; p < j.npieces; ++p) {
      HUAL_REQUIRE(j.kw[p] > 0 && (j.kw[p] % 16) == 0, "launch_gemm: piece width must be a multiple of 16");
      HUAL_REQUIRE(j.A[p] && j.W[p], "launch_gemm: null operand");
      HUAL_REQUIRE((j.lda[p] % 4) == 0 && (j.ldw % 4) == 0, "launch_gemm: leading dims must be multiples of 4");
    }
    HUAL_REQUIRE(j.Y != nullptr, "launch_gemm: null output");
    HUAL_REQUIRE(j.add_div >= 1, "launch_gemm: add_div");
    HUAL_REQUIRE(!j.ln_g || (j.ln_b && j.kw[0] == 128 && !j.A2[0]), "launch_gemm: LN prologue needs kw[0] == 128");
    HUAL_REQUIRE(!(j.ln_g || j.a_relu || j.a_save) || j.transW == 0, "launch_gemm: A prologue extras need [K,N] weights");
    HUAL_REQUIRE(!(j.a_relu && j.A2[0]), "launch_gemm: a_relu and A2[0] are exclusive");
    if (j.comb != COMB_NONE) dual = true;
    b.j[i] = j;
    maxM = j.M > maxM ? j.M : maxM;
    maxN = j.N > maxN ? j.N : maxN;
  }
  for (int i = 0; i < n; ++i)
    HUAL_REQUIRE((jobs[i].comb != COMB_NONE) == dual, "launch_gemm: cannot mix dual and single jobs in one launch");
  dim3 grid(cdiv(maxM, 32), cdiv(maxN, 128), n), block(256);
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    double kt = 0.0;
    for (int p = 0; p < jobs[i].npieces; ++p) kt += jobs[i].kw[p];
    const double mult = dual ? 2.0 : 1.0;
    flops += 2.0 * jobs[i].M * kt * jobs[i].N * mult;
    bytes += 4.0 * ((double)jobs[i].M * kt + kt * jobs[i].N * mult + (double)jobs[i].M * jobs[i].N);
  }
  static const int impl = []() { const char* e = getenv("HUAL_GEMM_IMPL"); return e ? atoi(e) : 1; }();
  bool anytrans = false;
  for (int i = 0; i < n; ++i) anytrans = anytrans || jobs[i].transW != 0;
  bool extras = false;
  for (int i = 0; i < n; ++i) extras = extras || jobs[i].ln_g || jobs[i].a_relu || jobs[i].a_save;
  HUAL_REQUIRE(!(extras && anytrans), "launch_gemm: A prologue extras need [K,N] weights");
  if ((impl == 0 && !extras) || anytrans) {   // the LDS-DMA kernel needs [K,N] weights; transposed reads fall back to gemm_kernel
    if (dual)
      HUAL_LAUNCH(flops, bytes, gemm_kernel<true>, grid, block, 0, stream, b, drop);
    else
      HUAL_LAUNCH(flops, bytes, gemm_kernel<false>, grid, block, 0, stream, b, drop);
  } else {
    const size_t lds = (size_t)(dual ? 4 : 2) * GL_STAGE * sizeof(float);
    {
      const void* fns[] = {(const void*)gemm_lds_kernel<true, 2>,     (const void*)gemm_lds_kernel<false, 2>,
                           (const void*)gemm_lds_kernel<true, 3>,     (const void*)gemm_lds_kernel<false, 3>,
                           (const void*)gemm_lds_kernel<true, 4>,     (const void*)gemm_lds_kernel<false, 4>,
                           (const void*)gemm_lds_kernel<true, 6>,     (const void*)gemm_lds_kernel<false, 6>,
                           (const void*)gemm_lds_px_kernel<true, 2>,  (const void*)gemm_lds_px_kernel<false, 2>,
                           (const void*)gemm_lds_px_kernel<true, 3>,  (const void*)gemm_lds_px_kernel<false, 3>};
      for (const void* f : fns) HUAL_DYN_LDS(f, 160 * 1024);
    }
    static const int rt_env = []() { const char* e = getenv("HUAL_GEMM_RT"); return e ? atoi(e) : 0; }();
    const int ncol = cdiv(maxN, 128);
    int rt = (cdiv(maxM, 32) * ncol * n > 256 && maxM > 48) ? 3 : 2;
    if (rt_env == 2 || rt_env == 3) rt = rt_env;
    // launches with many blocks per CU (several jobs / column blocks): bigger blocks share one weight panel among more
    // row tiles (HUAL_GEMM_RTBIG = 4 or 6; experiment)
    static const int rt_big = []() { const char* e = getenv("HUAL_GEMM_RTBIG"); return e ? atoi(e) : 0; }();
    static const int big_min = []() { const char* e = getenv("HUAL_GEMM_BIGMIN"); return e ? atoi(e) : 512; }();
    if (!extras && (rt_big == 4 || rt_big == 6) && cdiv(maxM, 48) * ncol * n >= big_min) rt = rt_big;
    const dim3 g(cdiv(maxM, 16 * rt), ncol, n), blk(128 * rt);
#define HUAL_LAUNCH_LDS(KERN)                                                                      \
  do {                                                                                             \
    if (rt == 3) {                                                                                 \
      if (dual) HUAL_LAUNCH(flops, bytes, (KERN<true, 3>), g, blk, lds, stream, b, drop);          \
      else HUAL_LAUNCH(flops, bytes, (KERN<false, 3>), g, blk, lds, stream, b, drop);              \
    } else {                                                                                       \
      if (dual) HUAL_LAUNCH(flops, bytes, (KERN<true, 2>), g, blk, lds, stream, b, drop);          \
      else HUAL_LAUNCH(flops, bytes, (KERN<false, 2>), g, blk, lds, stream, b, drop);              \
    }                                                                                              \
  } while (0)
    if (extras) HUAL_LAUNCH_LDS(gemm_lds_px_kernel);
    else if (rt == 4) {
      if (dual) HUAL_LAUNCH(flops, bytes, (gemm_lds_kernel<true, 4>), g, blk, lds, stream, b, drop);
      else HUAL_LAUNCH(flops, bytes, (gemm_lds_kernel<false, 4>), g, blk, lds, stream, b, drop);
    } else if (rt == 6) {
      if (dual) HUAL_LAUNCH(flops, bytes, (gemm_lds_kernel<true, 6>), g, blk, lds, stream, b, drop);
      else HUAL_LAUNCH(flops, bytes, (gemm_lds_kernel<false, 6>), g, blk, lds, stream, b, drop);
    } else HUAL_LAUNCH_LDS(gemm_lds_kernel);
#undef HUAL_LAUNCH_LDS
  }
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_pack_weights(const uint32_t* offs, const int* Ks, const uint32_t* boffs, int n, const float* P, char* fwd, char* bwd,
                        hipStream_t stream, const PackExtra* extra) {
  HUAL_REQUIRE(!extra || (extra->lens && extra->word_ids && extra->rowmask && extra->loss_acc && (extra->zero_n % 4) == 0 &&
                          (reinterpret_cast<uintptr_t>(extra->zero_ptr) & 15) == 0), "pack: extra prologue work");
  PackExtra ex{};
  for (int base = 0; base < n; base += HUAL_MAX_PACK) {
    const int cnt = n - base < HUAL_MAX_PACK ? n - base : HUAL_MAX_PACK;
    PackBatch b;
    int maxK = 0;
    double elems = 0.0;
    for (int i = 0; i < cnt; ++i) {
      HUAL_REQUIRE(Ks[base + i] > 0 && (Ks[base + i] % 8) == 0, "pack: K must be a positive multiple of 8");
      b.j[i].off = offs[base + i]; b.j[i].K = Ks[base + i]; b.j[i].boff = boffs ? boffs[base + i] : 0;
      const int kp = (Ks[base + i] + 127) & ~127;
      maxK = kp > maxK ? kp : maxK;
      elems += (double)Ks[base + i] * 128;
    }
    const bool with_extra = extra && base == 0;
    if (with_extra) ex = *extra;
    HUAL_LAUNCH(0.0, elems * (4.0 + (fwd ? 4.0 : 0.0) + (bwd ? 4.0 : 0.0)), pack_weights_kernel, dim3(maxK / 16, cnt + (with_extra ? 1 : 0)),
                dim3(256), 0, stream, b, P, fwd, bwd, cnt, ex);
  }
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

// jobs whose W[p] / W2[p] already point at packed images (pack_weights_kernel); N multiple of 128 column blocks of the image
int launch_gemm_bf16(const GemmJob* jobs, int n, const DropCfg& drop, hipStream_t stream) {
  HUAL_REQUIRE(n >= 1 && n <= HUAL_MAX_JOBS, "launch_gemm_bf16: job count");
  GemmBatch b;
  int maxM = 0, maxN = 0;
  bool dual = false;
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const GemmJob& j = jobs[i];
    HUAL_REQUIRE(j.M > 0 && j.N >= 4 && (j.N % 4) == 0, "launch_gemm_bf16: N must be a positive multiple of 4");
    HUAL_REQUIRE(j.npieces >= 1 && j.npieces <= HUAL_MAX_PIECES, "launch_gemm_bf16: pieces");
    double kt = 0.0;
    for (int p = 0; p < j.npieces; ++p) {
      HUAL_REQUIRE(j.kw[p] > 0 && (j.kw[p] % 8) == 0, "launch_gemm_bf16: piece width must be a multiple of 8");
      HUAL_REQUIRE(j.A[p] && j.W[p], "launch_gemm_bf16: null operand");
      HUAL_REQUIRE((j.lda[p] % 4) == 0, "launch_gemm_bf16: leading dims must be multiples of 4");
      kt += j.kw[p];
    }
    HUAL_REQUIRE(j.Y != nullptr && j.add_div >= 1, "launch_gemm_bf16: output");
    HUAL_REQUIRE(!j.ln_g && !j.a_relu && !j.a_save && !j.transW, "launch_gemm_bf16: A prologue extras / transW are not supported");
    if (j.comb != COMB_NONE) dual = true;
    b.j[i] = j;
    maxM = j.M > maxM ? j.M : maxM;
    maxN = j.N > maxN ? j.N : maxN;
    const double mult = j.comb != COMB_NONE ? 2.0 : 1.0;
    flops += 2.0 * j.M * kt * j.N * mult;
    bytes += 4.0 * ((double)j.M * kt + kt * j.N * mult + (double)j.M * j.N);
  }
  for (int i = 0; i < n; ++i)
    HUAL_REQUIRE((jobs[i].comb != COMB_NONE) == dual, "launch_gemm_bf16: cannot mix dual and single jobs in one launch");
  {
    const void* fns[] = {(const void*)gemm_bf16_kernel<true, 2>, (const void*)gemm_bf16_kernel<false, 2>,
                         (const void*)gemm_bf16_kernel<true, 3>, (const void*)gemm_bf16_kernel<false, 3>};
    for (const void* f : fns) HUAL_DYN_LDS(f, 160 * 1024);
  }
  const size_t lds = (size_t)(dual ? 4 : 2) * GB_STAGE;
  static const int rt_env = []() { const char* e = getenv("HUAL_GEMM_RT"); return e ? atoi(e) : 0; }();
  const int ncol = cdiv(maxN, 128);
  // 32-row blocks (4 waves): with the launches that are left outside the fused kernels (CQ dense and its dX, cq_concat, heads' dX)
  // 1.6219 ms/step against 1.6317 with 48-row blocks for the large launches (HUAL_GEMM_RT=3 forces those)
  int rt = 2;
  // deep-K jobs (video_conv1d: K = vdim, with Philox dropout on its A operand) are bound by per-wave VALU / MFMA work, not by
  // the per-block weight traffic: four waves per block spread evenly over the four SIMDs, six do not
  int kmax = 0;
  for (int i = 0; i < n; ++i) { int kt = 0; for (int p = 0; p < jobs[i].npieces; ++p) kt += jobs[i].kw[p]; kmax = kt > kmax ? kt : kmax; }
  if (kmax >= 768) rt = 2;
  if (rt_env == 2 || rt_env == 3) rt = rt_env;
  const dim3 g(cdiv(maxM, 16 * rt), ncol, n), blk(128 * rt);
  if (kmax >= 768 && !dual && rt == 2) {
    HUAL_DYN_LDS(feature_load_gemm_kernel<2>, 160 * 1024);
    HUAL_LAUNCH(flops, bytes, feature_load_gemm_kernel<2>, g, blk, lds, stream, b, drop);
  } else if (rt == 3) {
    if (dual) HUAL_LAUNCH(flops, bytes, (gemm_bf16_kernel<true, 3>), g, blk, lds, stream, b, drop);
    else HUAL_LAUNCH(flops, bytes, (gemm_bf16_kernel<false, 3>), g, blk, lds, stream, b, drop);
  } else {
    if (dual) HUAL_LAUNCH(flops, bytes, (gemm_bf16_kernel<true, 2>), g, blk, lds, stream, b, drop);
    else HUAL_LAUNCH(flops, bytes, (gemm_bf16_kernel<false, 2>), g, blk, lds, stream, b, drop);
  }
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_feature_ksplit(const FkJob* jobs, int n, const DropCfg& drop, hipStream_t stream) {
  HUAL_REQUIRE(n >= 1 && n <= HUAL_MAX_FK_JOBS, "feature_ksplit: job count");
  FkBatch b;
  int maxM = 0, maxKS = 0;
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const FkJob& j = jobs[i];
    HUAL_REQUIRE(j.A && j.Wimg && j.part && j.M > 0, "feature_ksplit: null / empty");
    HUAL_REQUIRE(j.K > 0 && (j.K % 8) == 0 && (j.KS % 64) == 0 && j.KS >= 64 && j.KS <= 256 && 4 * j.KS >= j.K && (j.lda % 4) == 0,
                 "feature_ksplit: need K % 8 == 0 and a quarter size KS (multiple of 64, <= 256) with 4*KS >= K");
    b.j[i] = j;
    maxM = j.M > maxM ? j.M : maxM;
    maxKS = j.KS > maxKS ? j.KS : maxKS;
    flops += 2.0 * j.M * (double)j.K * 128.0;
    // ALGORITHMIC bytes (T.(V+D).e per clip + weights): features + weights in, ONE [M,128] projection out.  The
    // four K-quarter partial slabs this kernel actually writes (4.M.128 floats, summed by the layer-norm launch behind it)
    // are implementation traffic: they show up in the measured PMC bytes, not here.
    bytes += (j.a_bf16 ? 2.0 : 4.0) * (double)j.M * j.K + 4.0 * ((double)j.K * 128 + (double)j.M * 128);
  }
  HUAL_DYN_LDS(feature_ksplit_kernel, 160 * 1024);
  const size_t lds = (size_t)(maxKS / 64) * GB_STAGE;
  HUAL_LAUNCH(flops, bytes, feature_ksplit_kernel, dim3(cdiv(maxM, FK_ROWS), 4, n), dim3(512), lds, stream, b, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_gemm_chain(const GemmJob* jobs, int n, const DropCfg& drop, hipStream_t stream) {
  HUAL_REQUIRE(n >= 1 && n <= HUAL_MAX_JOBS, "launch_gemm_chain: job count");
  GemmBatch b;
  bool dual = false;
  double flops = 0.0, bytes = 0.0;
  const int M = jobs[0].M;
  for (int i = 0; i < n; ++i) {
    const GemmJob& j = jobs[i];
    HUAL_REQUIRE(j.M == M && j.M > 0 && j.N == 128, "launch_gemm_chain: every job needs the same M and N = 128");
    HUAL_REQUIRE(j.npieces >= 1 && j.npieces <= HUAL_MAX_PIECES, "launch_gemm_chain: pieces");
    double kt = 0.0;
    for (int p = 0; p < j.npieces; ++p) {
      HUAL_REQUIRE(j.kw[p] > 0 && (j.kw[p] % 8) == 0 && j.A[p] && j.W[p] && (j.lda[p] % 4) == 0, "launch_gemm_chain: operand");
      kt += j.kw[p];
    }
    HUAL_REQUIRE(j.Y != nullptr && j.add_div >= 1, "launch_gemm_chain: output");
    HUAL_REQUIRE(!j.ln_g && !j.a_relu && !j.a_save && !j.transW, "launch_gemm_chain: A prologue extras / transW are not supported");
    if (j.comb != COMB_NONE) dual = true;
    b.j[i] = j;
    const double mult = j.comb != COMB_NONE ? 2.0 : 1.0;
    flops += 2.0 * j.M * kt * j.N * mult;
    bytes += 4.0 * ((double)j.M * kt + kt * j.N * mult + (double)j.M * j.N);
  }
  HUAL_DYN_LDS(gemm_chain_kernel<2>, 160 * 1024);
  HUAL_DYN_LDS(gemm_chain_kernel<3>, 160 * 1024);
  const size_t lds = (size_t)(dual ? 4 : 2) * GB_STAGE;
  const int rt = (cdiv(M, 32) > 256 && M > 48) ? 3 : 2;
  const dim3 g(cdiv(M, 16 * rt), 1, 1), blk(128 * rt);
  if (rt == 3) HUAL_LAUNCH(flops, bytes, gemm_chain_kernel<3>, g, blk, lds, stream, b, drop, n);
  else HUAL_LAUNCH(flops, bytes, gemm_chain_kernel<2>, g, blk, lds, stream, b, drop, n);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

static int dw_check(const DwJob& j, int& kbs, double& flops, double& bytes) {
  HUAL_REQUIRE(j.M > 0 && j.N == 128, "launch_dw: N must be 128");
  HUAL_REQUIRE(j.npieces >= 1 && j.npieces <= HUAL_MAX_PIECES, "launch_dw: pieces");
  kbs = 0;
  double kt = 0.0;
  for (int p = 0; p < j.npieces; ++p) {
    HUAL_REQUIRE(j.kw[p] > 0 && (j.kw[p] % 16) == 0, "launch_dw: piece width must be a multiple of 16");
    HUAL_REQUIRE(j.A[p] && j.dW[p], "launch_dw: null operand");
    HUAL_REQUIRE((j.lda[p] % 4) == 0 && (j.ldy % 4) == 0, "launch_dw: leading dims must be multiples of 4");
    kbs += cdiv(j.kw[p], 128);
    kt += j.kw[p];
  }
  flops += 2.0 * j.M * kt * j.N;
  bytes += 4.0 * ((double)j.M * kt + (double)j.M * j.N + kt * j.N);
  return 0;
}

// `table`: optional device buffer of n DwJob entries.  With it ALL jobs run as ONE launch (the descriptors are first
// written to the table by tiny kernels that carry them by value); without it jobs go HUAL_MAX_DW_JOBS per launch.
// rows_per_block = 0: pick the split of M that minimises (rounds of co-resident blocks) x (block length).  Every block
// of the launch runs about equally long (rows + a fixed prologue / atomics epilogue) and 2 blocks fit a CU, so the launch
// time is quantised in rounds of 512 blocks; a fixed split can sit just past a round boundary (measured: 389 us at
// 512 rows, 311 us at 384 rows for the same 62 jobs).
static int dw_auto_rows(const DwJob* jobs, int n) {
  static const int dbg = []() { const char* e = getenv("HUAL_DEBUG_DW"); return e ? atoi(e) : 0; }();
  int best = 512;
  double best_cost = 1e30;
  for (int r = 192; r <= 1024; r += 32) {
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
      int kbs = 0;
      for (int p = 0; p < jobs[i].npieces; ++p) kbs += (jobs[i].kw[p] + 127) / 128;
      blocks += (long)kbs * cdiv(jobs[i].M, r);
    }
    static const int slots = []() { const char* e = getenv("HUAL_DW_SLOTS"); return e ? atoi(e) : 512; }();
    const long rounds = (blocks + slots - 1) / slots;
    static const double fixed = []() { const char* e = getenv("HUAL_DW_FIXED"); return e ? atof(e) : 64.0; }();
    const double cost = (double)rounds * (r + fixed);
    if (dbg) fprintf(stderr, "[dw] rows %4d blocks %5ld rounds %2ld cost %.0f\n", r, blocks, rounds, cost);
    if (cost < best_cost) { best_cost = cost; best = r; }
  }
  return best;
}

int launch_dw(const DwJob* jobs, int n, const DropCfg& drop, int rows_per_block, hipStream_t stream, DwJob* table,
              bool write_table, int balanced_blocks) {
  static const int rows_env = []() { const char* e = getenv("HUAL_DW_ROWS"); return e ? atoi(e) : 0; }();
  if (rows_per_block == 0) rows_per_block = rows_env > 0 ? rows_env : dw_auto_rows(jobs, n);
  HUAL_REQUIRE(rows_per_block >= DW_TM && (rows_per_block % DW_TM) == 0, "launch_dw: rows_per_block must be a multiple of 32");
  HUAL_DYN_LDS(dw_kernel<true>, 96 * 1024);
  HUAL_DYN_LDS(dw_kernel<false>, 96 * 1024);
  // HUAL_DW_IMPL=0: fp32 MFMA kernel (dw_kernel); default: split-bf16 kernel (dw_bf16_kernel)
  const int dw_impl = []() { const char* e = getenv("HUAL_DW_IMPL"); return e ? atoi(e) : 1; }();     // read per call (tests)
  HUAL_DYN_LDS(dw_bf16_kernel<true>, 144 * 1024);
  HUAL_DYN_LDS(dw_bf16_kernel<false>, 144 * 1024);
  size_t lds = dw_impl ? (size_t)8 * DWB_PLANE : (size_t)4 * DW_TM * DW_LD * sizeof(float);
  for (int i = 0; i < n; ++i)
    HUAL_REQUIRE(!jobs[i].a_bf16 || (dw_impl && jobs[i].npieces == 1 && !jobs[i].A2[0] && (jobs[i].lda[0] % 4) == 0),
                 "dw: a bfloat16 operand needs the split-bf16 kernel, one piece, no product prologue");
  // HUAL_DW_LDS_KB pads the LDS request: above 80 KB only ONE block fits a CU, which leaves room for the blocks of
  // other kernels when the launch runs on a side stream under the dX chain (experiment)
  static const int lds_kb = []() { const char* e = getenv("HUAL_DW_LDS_KB"); return e ? atoi(e) : 0; }();
  if ((size_t)lds_kb * 1024 > lds && lds_kb <= 96) lds = (size_t)lds_kb * 1024;
  if (table != nullptr) {
    int maxM = 0, maxKb = 0;
    double flops = 0.0, bytes = 0.0;
    int tiles = 0, cost = 0;                     // running tile count / cost of the balanced launch
    // cost of one 64-row tile by prologue (per-tile cycles of the workgroups, scripts/exp/dw_stamps.py: plain 6.5 k, product
    // and strided clip features + keep bytes 8-9 k; the launch time is flat from 44 to 52, scripts/exp/dw_weights.sh);
    // HUAL_DW_WEIGHTS="plain,prod,drop" overrides
    static int wts[3] = {32, 46, 46};
    static const bool wts_env = []() {
      const char* e = getenv("HUAL_DW_WEIGHTS");
      if (e) sscanf(e, "%d,%d,%d", &wts[0], &wts[1], &wts[2]);
      return e != nullptr;
    }();
    (void)wts_env;
    for (int base = 0; base < n; base += HUAL_MAX_DW_JOBS) {
      const int cnt = n - base < HUAL_MAX_DW_JOBS ? n - base : HUAL_MAX_DW_JOBS;
      DwBatch b;
      DwPlanPart pre;
      for (int i = 0; i < cnt; ++i) {
        const DwJob& jb = jobs[base + i];
        int kbs;
        int rc = dw_check(jb, kbs, flops, bytes);
        if (rc) return rc;
        b.j[i] = jb;
        maxM = jb.M > maxM ? jb.M : maxM;
        maxKb = kbs > maxKb ? kbs : maxKb;
        bool prod = false;
        for (int p = 0; p < jb.npieces; ++p) prod = prod || jb.A2[p] != nullptr;
        const int w = (jb.a_drop_site >= 0 || jb.a_bf16) ? wts[2] : prod ? wts[1] : wts[0];
        pre.tiles[i] = tiles;
        pre.cost[i] = cost;
        pre.w[i] = w;
        tiles += kbs * cdiv(jb.M, DWB_TM);
        cost += kbs * (DW_UNIT_COST + cdiv(jb.M, DWB_TM) * w);
      }
      for (int i = cnt; i <= HUAL_MAX_DW_JOBS; ++i) { pre.tiles[i] = tiles; pre.cost[i] = cost; pre.w[i] = 1; }
      if (write_table) HUAL_LAUNCH(0.0, 0.0, dw_table_write_kernel, dim3(1), dim3(64), 0, stream, b, pre, table, base, cnt, n);
    }
    // HUAL_DW_BALANCED=0: the fixed row split (grid of row chunks x k-blocks x jobs)
    const int balanced = []() { const char* e = getenv("HUAL_DW_BALANCED"); return e ? atoi(e) : 1; }();     // read per call (tests)
    if (dw_impl && balanced) {
      // one 512-thread workgroup per CU; HUAL_DW_BLOCKS overrides the count
      static const int blocks_env = []() { const char* e = getenv("HUAL_DW_BLOCKS"); return e ? atoi(e) : 0; }();
      int dev = 0, cus = 256;
      HUAL_CHECK_HIP(hipGetDevice(&dev));
      HUAL_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      static const int dbg_env = []() { const char* e = getenv("HUAL_DW_DBG"); return e ? atoi(e) : 0; }();
      int nblocks = balanced_blocks > 0 ? balanced_blocks : blocks_env > 0 ? blocks_env : cus;
      if (nblocks > tiles) nblocks = tiles;
      HUAL_DYN_LDS(dw_bf16_balanced_kernel, 144 * 1024);
      HUAL_LAUNCH(flops, bytes, dw_bf16_balanced_kernel, dim3(nblocks), dim3(DWB_THREADS), lds, stream, (const DwJob*)table, n, drop, dbg_env);
      HUAL_CHECK_HIP(hipGetLastError());
      return 0;
    }
    DwBatch dummy;
    dw_job_init(dummy.j[0]);
    dim3 grid(cdiv(maxM, rows_per_block), maxKb, n), block(256);
    if (dw_impl) HUAL_LAUNCH(flops, bytes, dw_bf16_kernel<true>, grid, dim3(DWB_THREADS), lds, stream, dummy, (const DwJob*)table, drop, rows_per_block);
    else HUAL_LAUNCH(flops, bytes, dw_kernel<true>, grid, block, lds, stream, dummy, (const DwJob*)table, drop, rows_per_block);
    HUAL_CHECK_HIP(hipGetLastError());
    return 0;
  }
  for (int base = 0; base < n; base += HUAL_MAX_DW_JOBS) {
    int cnt = n - base < HUAL_MAX_DW_JOBS ? n - base : HUAL_MAX_DW_JOBS;
    DwBatch b;
    int maxM = 0, maxKb = 0;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < cnt; ++i) {
      int kbs;
      int rc = dw_check(jobs[base + i], kbs, flops, bytes);
      if (rc) return rc;
      b.j[i] = jobs[base + i];
      maxM = jobs[base + i].M > maxM ? jobs[base + i].M : maxM;
      maxKb = kbs > maxKb ? kbs : maxKb;
    }
    dim3 grid(cdiv(maxM, rows_per_block), maxKb, cnt), block(256);
    if (dw_impl) HUAL_LAUNCH(flops, bytes, dw_bf16_kernel<false>, grid, dim3(DWB_THREADS), lds, stream, b, (const DwJob*)nullptr, drop, rows_per_block);
    else HUAL_LAUNCH(flops, bytes, dw_kernel<false>, grid, block, lds, stream, b, (const DwJob*)nullptr, drop, rows_per_block);
    HUAL_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

}  // namespace hual
