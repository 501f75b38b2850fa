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

// backward: LDS accumulators [dF | dB | dT | dU]; one word at a time per block, all 256 threads.
// Thread <-> filter tap (kernel width k, tap dk, embedding column d): it owns that row of dF (plain LDS
// read-modify-write, no conflicts) and walks the channels of width k; the char-embedding gradient goes through
// ds_add_f32 (taps of different threads can meet on the same (char, d) only across waves).
__global__ __launch_bounds__(256) void embed_bwd_kernel(EmbedArgs a, EmbedGrads gr, DropCfg drop, int nrows,
                                                        int words_per_block) {
  extern __shared__ float sm[];
  const int wd = a.word_dim, cd = a.char_dim, C = a.C;
  const int nfil = cd * (1 * 10 + 2 * 20 + 3 * 30 + 4 * 40);
  const int ntab = (a.num_chars - 1) * cd;
  float* dF = sm;                        // [nfil]
  float* dB = dF + nfil;                 // [NCH]
  float* dT = dB + NCH;                  // [ntab]
  float* dU = dT + ntab;                 // [wd]
  float* emb = dU + wd;                  // [C*cd] dropped char embeddings of the current word
  float* demb = emb + C * cd;            // [C*cd]
  float* gch = demb + C * cd;            // [NCH] upstream gradient per channel (0 where relu clipped)
  int* argc = reinterpret_cast<int*>(gch + NCH);   // [NCH]
  const int tid = threadIdx.x;
  const int nacc = nfil + NCH + ntab + wd;
  for (int i = tid; i < nacc; i += 256) sm[i] = 0.f;
  const int foff[4] = {0, cd * 10, cd * 10 + 2 * cd * 20, cd * 10 + 2 * cd * 20 + 3 * cd * 30};
  const int choff[4] = {0, 10, 30, 60};
  const int ntap = 10 * cd;              // (1+2+3+4) * cd taps
  const int row_lo = blockIdx.x * words_per_block;
  const int row_hi = min(row_lo + words_per_block, nrows);
  for (int row = row_lo; row < row_hi; ++row) {
    __syncthreads();
    load_char_emb(a, drop, row, emb, tid, 256);
    for (int idx = tid; idx < C * cd; idx += 256) demb[idx] = 0.f;
    const float* dcat = gr.dcat + (size_t)row * gr.lddcat;
    if (tid < NCH) {
      const int arg = a.char_arg[(size_t)row * NCH + tid];
      const float g = arg >= 0 ? dcat[wd + tid] : 0.f;
      argc[tid] = arg;
      gch[tid] = g;
      dB[tid] += g;
    }
    if (a.word_ids[row] == 1)
      for (int c = tid; c < wd; c += 256) {
        float g = dcat[c];
        if (drop.enabled) g *= drop1(drop, HUAL_SITE_WORD, (uint32_t)row, c);
        dU[c] += g;
      }
    __syncthreads();
    for (int tap = tid; tap < ntap; tap += 256) {
      // taps are ordered [k=1: cd][k=2: 2cd][k=3: 3cd][k=4: 4cd]
      int k, base;
      if (tap < cd) { k = 1; base = 0; }
      else if (tap < 3 * cd) { k = 2; base = cd; }
      else if (tap < 6 * cd) { k = 3; base = 3 * cd; }
      else { k = 4; base = 6 * cd; }
      const int rel = tap - base;            // dk*cd + d
      const int dk = rel / cd, d = rel - dk * cd;
      const int nchk = 10 * k;
      const float* F = a.filt[k - 1] + (size_t)rel * nchk;
      float* f = dF + foff[k - 1] + rel * nchk;
      const int c0 = choff[k - 1];
      for (int ch = 0; ch < nchk; ++ch) {
        const int arg = argc[c0 + ch];
        if (arg < 0) continue;
        const float g = gch[c0 + ch];
        const int ei = (arg + dk) * cd + d;
        f[ch] += g * emb[ei];
        atomicAdd(&demb[ei], g * F[ch]);
      }
    }
    __syncthreads();
    for (int idx = tid; idx < C * cd; idx += 256) {
      const int c = idx / cd, d = idx - c * cd;
      const int cid = a.char_ids[(size_t)row * C + c];
      if (cid > 0) {
        float g = demb[idx];
        if (drop.enabled) g *= drop1(drop, HUAL_SITE_CHAR, (uint32_t)(row * C + c), d);
        atomicAdd(&dT[(size_t)(cid - 1) * cd + d], g);
      }
    }
  }
  __syncthreads();
  // this block's partial sums -> scratch (plain coalesced stores; summed over blocks by embed_reduce_kernel)
  float* part = gr.partial + (size_t)blockIdx.x * nacc;
  for (int i = tid; i < nacc; i += 256) part[i] = sm[i];
}

// sums the per-block partials [nblocks][nacc] and adds them to the parameter gradients (one owner per element).
// 256 threads = 64 elements x 4 groups of blocks.
__global__ __launch_bounds__(256) void embed_reduce_kernel(EmbedArgs a, EmbedGrads gr, int nblocks) {
  __shared__ float part[4][64];
  const int cd = a.char_dim, wd = a.word_dim;
  const int nfil = cd * (1 * 10 + 2 * 20 + 3 * 30 + 4 * 40);
  const int ntab = (a.num_chars - 1) * cd;
  const int nacc = nfil + NCH + ntab + wd;
  const int foff[4] = {0, cd * 10, cd * 10 + 2 * cd * 20, cd * 10 + 2 * cd * 20 + 3 * cd * 30};
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  float s = 0.f;
  if (i < nacc)
    for (int b = grp; b < nblocks; b += 4) s += gr.partial[(size_t)b * nacc + i];
  part[grp][e] = s;
  __syncthreads();
  if (grp == 0 && i < nacc) {
    s = part[0][e] + part[1][e] + part[2][e] + part[3][e];
    if (i < nfil) {
      const int k = i < foff[1] ? 0 : (i < foff[2] ? 1 : (i < foff[3] ? 2 : 3));
      gr.dfilt[k][i - foff[k]] += s;
    } else if (i < nfil + NCH) {
      int k, chk, nchk;
      chan_to_kernel(i - nfil, k, chk, nchk);
      gr.dfbias[k - 1][chk] += s;
    } else if (i < nfil + NCH + ntab) {
      gr.dchar_table[i - nfil - NCH] += s;
    } else {
      gr.dunk[i - nfil - NCH - ntab] += s;
    }
  }
}

namespace hual {

int launch_embed_fwd(const EmbedArgs& a, int nrows, const DropCfg& drop, hipStream_t s) {
  HUAL_REQUIRE(a.C >= 4, "char_ids need at least 4 chars per word (conv width 4, VALID) - modules.py:33");
  int grid = cdiv(nrows, 2);
  grid = grid < 1024 ? grid : 1024;
  HUAL_LAUNCH(0.0, 0.0, embed_fwd_kernel, dim3(grid), dim3(256), (size_t)2 * a.C * a.char_dim * sizeof(float), s, a, drop, nrows);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

#define EMBED_WPB 5                      // words per block of the backward kernel
int embed_bwd_blocks(int nrows) { return cdiv(nrows, EMBED_WPB); }
size_t embed_bwd_partial_floats(int nrows, int word_dim, int char_dim, int num_chars) {
  const size_t nacc = (size_t)char_dim * 300 + NCH + (size_t)(num_chars - 1) * char_dim + word_dim;
  return nacc * embed_bwd_blocks(nrows);
}

int launch_embed_bwd(const EmbedArgs& a, const EmbedGrads& g, int nrows, const DropCfg& drop, hipStream_t s) {
  const int cd = a.char_dim;
  const int nfil = cd * (10 + 40 + 90 + 160);
  const size_t bytes = ((size_t)2 * a.C * cd + nfil + 3 * NCH + (size_t)(a.num_chars - 1) * cd + a.word_dim) * sizeof(float);
  HUAL_REQUIRE(bytes <= 160 * 1024, "embed_bwd: char filter gradients do not fit LDS");
  static bool attr = false;
  if (!attr) {
    HUAL_CHECK_HIP(hipFuncSetAttribute((const void*)embed_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  HUAL_REQUIRE(g.partial != nullptr, "embed_bwd: null partial-sum scratch");
  const int grid = embed_bwd_blocks(nrows);
  HUAL_LAUNCH(0.0, 0.0, embed_bwd_kernel, dim3(grid), dim3(256), bytes, s, a, g, drop, nrows, EMBED_WPB);
  const int nacc = nfil + NCH + (a.num_chars - 1) * cd + a.word_dim;
  HUAL_LAUNCH(0.0, 0.0, embed_reduce_kernel, dim3(cdiv(nacc, 64)), dim3(256), 0, s, a, g, grid);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
