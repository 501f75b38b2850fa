// Device pieces of the text-encoder front end that run inside the step's prologue launch (pack_weights_kernel, gemm.hip): the
// embedding gather (word lookup, char lookup, their dropout, the packed bias vector) and the element formula of the packed
// char-CNN filter bank "Wall" (embed.hip) from which the prologue writes the pre-split images.
#pragma once
#include "embed_args.h"
#include "philox.h"

#define NCH 100          // 10+20+30+40 channels
#define NALL 128         // padded channel count = GEMM N
__device__ __host__ __forceinline__ int bank_off(int k) { return k == 1 ? 0 : (k == 2 ? 10 : (k == 3 ? 30 : 60)); }
__device__ __forceinline__ void chan_to_kernel(int ch, int& k, int& chk) {
  if (ch < 10) { k = 1; chk = ch; }
  else if (ch < 30) { k = 2; chk = ch - 10; }
  else if (ch < 60) { k = 3; chk = ch - 30; }
  else { k = 4; chk = ch - 60; }
}

// Wall[r][n], r = dk * CP + d: filter_k[dk][d][chk] for channel n = (k, chk), zero for dk >= k, padding columns / dims
__device__ __forceinline__ float wall_value(const float* const (&filt)[4], int cd, int CP, int r, int n) {
  const int dk = r / CP, d = r - dk * CP;
  if (n >= NCH || d >= cd) return 0.f;
  int k, chk;
  chan_to_kernel(n, k, chk);
  return dk < k ? filt[k - 1][(size_t)(dk * cd + d) * (10 * k) + chk] : 0.f;
}

// one task = 4 consecutive columns of a word row or of a char slot row; tail tasks: guard rows of cemb, the packed biases.
// ntask_words = nrows * (word_dim / 4 + C * CP / 4); tasks gid >= embed_gather_tasks(...) do nothing
__device__ __forceinline__ void embed_gather_task(const hual::EmbedArgs& a, const hual::DropCfg& drop, int nrows, int CP, int ntask_words, int gid) {
  const int wd = a.word_dim, cd = a.char_dim, C = a.C;
  const int ngw = wd >> 2, ngc = CP >> 2;
  const int per_word = ngw + C * ngc;
  if (gid < ntask_words) {
    const int row = gid / per_word, t = gid - row * per_word;
    if (t < ngw) {
      const int wid = a.word_ids[row];
      float4 v = f4zero();
      if (wid == 1) v = ld4(a.unk + 4 * t);
      else if (wid >= 2) v = ld4(a.word_table + (size_t)(wid - 2) * wd + 4 * t);
      if (drop.enabled) v = apply_drop4(drop, HUAL_SITE_WORD, (uint32_t)row, (uint32_t)t, v);
      st4(a.cat + (size_t)row * a.ldcat + 4 * t, v);
    } else {
      const int u = t - ngw, c = u / ngc, g4 = u - c * ngc;
      const int cid = a.char_ids[(size_t)row * C + c];
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (cid > 0) {
        const float* src = a.char_table + (size_t)(cid - 1) * cd;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * g4 + q < cd) e[q] = src[4 * g4 + q];
      }
      float4 v = make_float4(e[0], e[1], e[2], e[3]);
      if (drop.enabled && 4 * g4 < cd) v = apply_drop4(drop, HUAL_SITE_CHAR, (uint32_t)(row * C + c), (uint32_t)g4, v);
      st4(a.cemb + ((size_t)row * C + c) * CP + 4 * g4, v);
    }
    return;
  }
  int x = gid - ntask_words;
  if (x < 4 * CP) { a.cemb[(size_t)nrows * C * CP + x] = 0.f; return; }      // guard rows behind the last window
  x -= 4 * CP;
  if (x < NALL) {
    float v = 0.f;
    if (x < NCH) {
      int k, chk;
      chan_to_kernel(x, k, chk);
      v = a.fbias[k - 1][chk];
    }
    a.ball[x] = v;
  }
}

// backward 3 (after the weight-gradient launch): packed dFall / dball -> filter and bias gradients; task gid of cd * 300 + NCH
__device__ __forceinline__ void embed_unpack_task(const hual::EmbedArgs& a, const hual::EmbedGrads& gr, int CP, int gid) {
  const int cd = a.char_dim;
  const int nfil = cd * (10 + 40 + 90 + 160);
  if (gid < nfil) {
    const int foff[4] = {0, cd * 10, cd * 10 + 2 * cd * 20, cd * 10 + 2 * cd * 20 + 3 * cd * 30};
    const int k = gid < foff[1] ? 1 : (gid < foff[2] ? 2 : (gid < foff[3] ? 3 : 4));
    const int rel = gid - foff[k - 1];             // (dk*cd + d) * 10k + n
    const int n = rel % (10 * k), tap = rel / (10 * k);
    const int dk = tap / cd, d = tap - dk * cd;
    gr.dfilt[k - 1][rel] += a.dfall[(size_t)(dk * CP + d) * NALL + bank_off(k) + n];
  } else if (gid < nfil + NCH) {
    const int ch = gid - nfil;
    int k, chk;
    chan_to_kernel(ch, k, chk);
    gr.dfbias[k - 1][chk] += a.dfall[(size_t)4 * CP * NALL + ch];
  }
}

// backward 2 of the text encoder front end (embed.hip): fold the window gradients back to char slots:
//   d cemb[r][d] = sum_dk dXall[r - dk][dk*CP + d]  (same word only), through the dropout mask, accumulated per char id.
// Workgroup `blk` (256 threads) = 64 slot rows, dT = its LDS accumulator [(num_chars - 1) * char_dim].  A launch of its own
// (embed_finish_kernel) or further workgroups of the launch that folds the per-workgroup partial sums (rowops.h launch_colsum).
#define EF_ROWS 64
__device__ __forceinline__ void embed_finish_block(const hual::EmbedArgs& a, const hual::EmbedGrads& gr, const hual::DropCfg& drop, int nrows, int CP,
                                                   int blk, float* dT) {
  const int cd = a.char_dim, C = a.C;
  const int ntab = (a.num_chars - 1) * cd;
  for (int i = threadIdx.x; i < ntab; i += 256) dT[i] = 0.f;
  __syncthreads();
  const int ngc = (cd + 3) >> 2;
  const int M = nrows * C;
  const int r0 = blk * EF_ROWS;
  for (int t = threadIdx.x; t < EF_ROWS * ngc; t += 256) {
    const int r = r0 + t / ngc, g4 = t % ngc;
    if (r >= M) break;
    const int cid = a.char_ids[r];
    if (cid <= 0) continue;
    const int c = r % C;
    float4 s = f4zero();
    for (int dk = 0; dk < 4 && dk <= c; ++dk) {
      const float4 v = ld4(a.dxall + (size_t)(r - dk) * 4 * CP + dk * CP + 4 * g4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (drop.enabled) s = apply_drop4(drop, HUAL_SITE_CHAR, (uint32_t)r, (uint32_t)g4, s);
    const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (4 * g4 + q < cd) atomicAdd(&dT[(size_t)(cid - 1) * cd + 4 * g4 + q], sv[q]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ntab; i += 256) {
    const float v = dT[i];
    if (v != 0.f) atomicAdd(gr.dchar_table + i, v);
  }
}
