// Text-encoder front end: word lookup + char CNN (+ their backward).
//   word_embs  /root/reference/models/modules.py:8-16   table = [zeros; unk; frozen GloVe]
//   char_embs  /root/reference/models/modules.py:19-38  lookup ([zeros; char_table]) -> 4 x conv2d VALID
//              (widths 1..4 -> 10/20/30/40 channels) + bias -> relu -> max over chars (padding not masked)
// Output row = [word_emb(300) | char features(100)] (model.py:41), consumed by the query_conv1d GEMM.
//
// Work decomposition: a 256-thread block handles TWO words at a time (threads 0-127 / 128-255); inside a word
// thread c < 100 owns output channel c.  Filter taps are read from global memory with consecutive channels on
// consecutive lanes (coalesced, L2 resident: 60-120 KB), the word's char embeddings sit in LDS and are read as
// broadcasts.  The backward pass keeps ALL parameter gradients of the front end in LDS accumulators (ds_add_f32)
// while a block walks its words and flushes them with one set of global atomics per block.
#include "embed.h"
#include "philox.h"
#include "prof.h"

using namespace hual;

#define NCH 100          // 10+20+30+40 channels
#define MAXPOS 16        // conv positions kept in registers at once
__device__ __forceinline__ void chan_to_kernel(int ch, int& k, int& chk, int& nchk) {
  if (ch < 10) { k = 1; chk = ch; nchk = 10; }
  else if (ch < 30) { k = 2; chk = ch - 10; nchk = 20; }
  else if (ch < 60) { k = 3; chk = ch - 30; nchk = 30; }
  else { k = 4; chk = ch - 60; nchk = 40; }
}

__device__ __forceinline__ float drop1(const DropCfg& d, uint32_t site, uint32_t row, int col) {
  float4 m = drop_mask4(d, site, row, (uint32_t)(col >> 2));
  const int c = col & 3;
  return c == 0 ? m.x : (c == 1 ? m.y : (c == 2 ? m.z : m.w));
}

// stage the (dropped) char embeddings of `row` into emb[C*cd]; threads tid..(+=nthr)
__device__ __forceinline__ void load_char_emb(const EmbedArgs& a, const DropCfg& drop, int row, float* emb, int tid, int nthr) {
  const int cd = a.char_dim, C = a.C;
  for (int idx = tid; idx < C * cd; idx += nthr) {
    const int c = idx / cd, d = idx - c * cd;
    const int cid = a.char_ids[(size_t)row * C + c];
    float v = cid > 0 ? a.char_table[(size_t)(cid - 1) * cd + d] : 0.f;
    if (drop.enabled) v *= drop1(drop, HUAL_SITE_CHAR, (uint32_t)(row * C + c), d);
    emb[idx] = v;
  }
}

__global__ __launch_bounds__(256) void embed_fwd_kernel(EmbedArgs a, DropCfg drop, int nrows) {
  extern __shared__ float sm[];    // [2][C*cd]
  const int half = threadIdx.x >> 7, tid = threadIdx.x & 127;
  const int wd = a.word_dim, cd = a.char_dim, C = a.C;
  float* emb = sm + half * C * cd;
  for (int row0 = blockIdx.x * 2; row0 < nrows; row0 += gridDim.x * 2) {
    const int row = row0 + half;
    const bool act = row < nrows;
    __syncthreads();
    if (act) {
      float* out = a.cat + (size_t)row * a.ldcat;
      const int wid = a.word_ids[row];
      for (int c = tid; c < wd; c += 128) {
        float v = 0.f;
        if (wid == 1) v = a.unk[c];
        else if (wid >= 2) v = a.word_table[(size_t)(wid - 2) * wd + c];
        if (drop.enabled) v *= drop1(drop, HUAL_SITE_WORD, (uint32_t)row, c);
        out[c] = v;
      }
      load_char_emb(a, drop, row, emb, tid, 128);
    }
    __syncthreads();
    if (act && tid < NCH) {
      int k, chk, nchk;
      chan_to_kernel(tid, k, chk, nchk);
      const float* F = a.filt[k - 1] + chk;          // [k][cd][nchk]
      const float bias = a.fbias[k - 1][chk];
      const int npos = C - k + 1;
      float best = 0.f;                              // relu floor: max_c relu(o_c) = max(0, max_c o_c)
      int arg = -1;
      for (int p0 = 0; p0 < npos; p0 += MAXPOS) {
        float acc[MAXPOS];
#pragma unroll
        for (int p = 0; p < MAXPOS; ++p) acc[p] = bias;
        for (int dk = 0; dk < k; ++dk) {
          for (int d = 0; d < cd; ++d) {
            const float f = F[(size_t)(dk * cd + d) * nchk];
            const float* e = emb + (p0 + dk) * cd + d;
#pragma unroll
            for (int p = 0; p < MAXPOS; ++p)
              if (p0 + p < npos) acc[p] = fmaf(e[p * cd], f, acc[p]);
          }
        }
#pragma unroll
        for (int p = 0; p < MAXPOS; ++p)
          if (p0 + p < npos && acc[p] > best) { best = acc[p]; arg = p0 + p; }
      }
      a.cat[(size_t)row * a.ldcat + wd + tid] = best;
      a.char_arg[(size_t)row * NCH + tid] = arg;
    }
  }
}

// backward: LDS accumulators [dF | dB | dT | dU], two words in flight per block
__global__ __launch_bounds__(256) void embed_bwd_kernel(EmbedArgs a, EmbedGrads gr, DropCfg drop, int nrows) {
  extern __shared__ float sm[];
  const int wd = a.word_dim, cd = a.char_dim, C = a.C;
  const int nfil = cd * (1 * 10 + 2 * 20 + 3 * 30 + 4 * 40);
  const int ntab = (a.num_chars - 1) * cd;
  float* dF = sm;                        // [nfil]
  float* dB = dF + nfil;                 // [NCH]
  float* dT = dB + NCH;                  // [ntab]
  float* dU = dT + ntab;                 // [wd]
  float* slot = dU + wd;                 // per half: emb[C*cd], demb[C*cd]
  const int half = threadIdx.x >> 7, tid = threadIdx.x & 127;
  float* emb = slot + half * 2 * C * cd;
  float* demb = emb + C * cd;
  const int nacc = nfil + NCH + ntab + wd;
  for (int i = threadIdx.x; i < nacc; i += 256) sm[i] = 0.f;
  const int foff[4] = {0, cd * 10, cd * 10 + 2 * cd * 20, cd * 10 + 2 * cd * 20 + 3 * cd * 30};
  for (int row0 = blockIdx.x * 2; row0 < nrows; row0 += gridDim.x * 2) {
    const int row = row0 + half;
    const bool act = row < nrows;
    __syncthreads();
    if (act) {
      load_char_emb(a, drop, row, emb, tid, 128);
      for (int idx = tid; idx < C * cd; idx += 128) demb[idx] = 0.f;
      const float* dcat = gr.dcat + (size_t)row * gr.lddcat;
      if (a.word_ids[row] == 1)
        for (int c = tid; c < wd; c += 128) {
          float g = dcat[c];
          if (drop.enabled) g *= drop1(drop, HUAL_SITE_WORD, (uint32_t)row, c);
          atomicAdd(&dU[c], g);
        }
    }
    __syncthreads();
    if (act && tid < NCH) {
      const int arg = a.char_arg[(size_t)row * NCH + tid];
      if (arg >= 0) {
        int k, chk, nchk;
        chan_to_kernel(tid, k, chk, nchk);
        const float g = gr.dcat[(size_t)row * gr.lddcat + wd + tid];
        atomicAdd(&dB[tid], g);
        const float* F = a.filt[k - 1] + chk;
        float* f = dF + foff[k - 1] + chk;
        for (int dk = 0; dk < k; ++dk) {
          const float* e = emb + (arg + dk) * cd;
          float* de = demb + (arg + dk) * cd;
          for (int d = 0; d < cd; ++d) {
            const int fi = (dk * cd + d) * nchk;
            atomicAdd(&f[fi], g * e[d]);            // filter gradient (LDS)
            atomicAdd(&de[d], g * F[fi]);           // embedding gradient (LDS)
          }
        }
      }
    }
    __syncthreads();
    if (act) {
      for (int idx = tid; idx < C * cd; idx += 128) {
        const int c = idx / cd, d = idx - c * cd;
        const int cid = a.char_ids[(size_t)row * C + c];
        if (cid > 0) {
          float g = demb[idx];
          if (drop.enabled) g *= drop1(drop, HUAL_SITE_CHAR, (uint32_t)(row * C + c), d);
          atomicAdd(&dT[(size_t)(cid - 1) * cd + d], g);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nfil; i += 256) {
    const int k = i < foff[1] ? 0 : (i < foff[2] ? 1 : (i < foff[3] ? 2 : 3));
    const float v = dF[i];
    if (v != 0.f) atomicAdd(gr.dfilt[k] + (i - foff[k]), v);
  }
  for (int i = threadIdx.x; i < NCH; i += 256) {
    int k, chk, nchk;
    chan_to_kernel(i, k, chk, nchk);
    atomicAdd(gr.dfbias[k - 1] + chk, dB[i]);
  }
  for (int i = threadIdx.x; i < ntab; i += 256) {
    const float v = dT[i];
    if (v != 0.f) atomicAdd(gr.dchar_table + i, v);
  }
  for (int i = threadIdx.x; i < wd; i += 256) {
    const float v = dU[i];
    if (v != 0.f) atomicAdd(gr.dunk + i, v);
  }
}

namespace hual {

int launch_embed_fwd(const EmbedArgs& a, int nrows, const DropCfg& drop, hipStream_t s) {
  HUAL_REQUIRE(a.C >= 4, "char_ids need at least 4 chars per word (conv width 4, VALID) - modules.py:33");
  int grid = cdiv(nrows, 2);
  grid = grid < 1024 ? grid : 1024;
  ProfScope ps(PK_EMBED, s, 0.0, 0.0);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid), dim3(256), (size_t)2 * a.C * a.char_dim * sizeof(float), s, a, drop, nrows);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_embed_bwd(const EmbedArgs& a, const EmbedGrads& g, int nrows, const DropCfg& drop, hipStream_t s) {
  const int cd = a.char_dim;
  const int nfil = cd * (10 + 40 + 90 + 160);
  const size_t bytes = ((size_t)4 * a.C * cd + nfil + NCH + (size_t)(a.num_chars - 1) * cd + a.word_dim) * sizeof(float);
  HUAL_REQUIRE(bytes <= 160 * 1024, "embed_bwd: char filter gradients do not fit LDS");
  static bool attr = false;
  if (!attr) {
    HUAL_CHECK_HIP(hipFuncSetAttribute((const void*)embed_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  int grid = cdiv(nrows, 8);           // >= 4 word pairs per block before its accumulators are flushed
  grid = grid < 256 ? (grid > 0 ? grid : 1) : 256;
  ProfScope ps(PK_EMBED, s, 0.0, 0.0);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid), dim3(256), bytes, s, a, g, drop, nrows);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
