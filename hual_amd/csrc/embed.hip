// Text-encoder front end: word lookup + char CNN (+ their backward).
//   word_embs  /root/reference/models/modules.py:8-16   table = [zeros; unk; frozen GloVe]
//   char_embs  /root/reference/models/modules.py:19-38  lookup ([zeros; char_table]) -> 4 x conv2d VALID
//              (widths 1..4 -> 10/20/30/40 channels) + bias -> relu -> max over chars (padding not masked)
// Output row = [word_emb(300) | char features(100)] (model.py:41), consumed by the query_conv1d GEMM.
#include "embed.h"
#include "philox.h"

using namespace hual;

#define NCH 100          // 10+20+30+40 channels
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

// one block per word (row of the [Nq] query rows)
__global__ __launch_bounds__(128) void embed_fwd_kernel(EmbedArgs a, DropCfg drop) {
  extern __shared__ float emb[];    // [C][cd]
  const int row = blockIdx.x;
  const int tid = threadIdx.x;
  const int wd = a.word_dim, cd = a.char_dim, C = a.C;
  float* out = a.cat + (size_t)row * a.ldcat;
  // ---- word embedding
  const int wid = a.word_ids[row];
  for (int c = tid; c < wd; c += 128) {
    float v = 0.f;
    if (wid == 1) v = a.unk[c];
    else if (wid >= 2) v = a.word_table[(size_t)(wid - 2) * wd + c];
    if (drop.enabled) v *= drop1(drop, HUAL_SITE_WORD, (uint32_t)row, c);
    out[c] = v;
  }
  // ---- char embeddings into LDS
  for (int idx = tid; idx < C * cd; idx += 128) {
    const int c = idx / cd, d = idx - c * cd;
    const int cid = a.char_ids[(size_t)row * C + c];
    float v = cid > 0 ? a.char_table[(size_t)(cid - 1) * cd + d] : 0.f;
    if (drop.enabled) v *= drop1(drop, HUAL_SITE_CHAR, (uint32_t)(row * C + c), d);
    emb[idx] = v;
  }
  __syncthreads();
  if (tid < NCH) {
    int k, chk, nchk;
    chan_to_kernel(tid, k, chk, nchk);
    const float* F = a.filt[k - 1];          // [k][cd][nchk]
    const float bias = a.fbias[k - 1][chk];
    float best = 0.f;                        // relu floor: max_c relu(o_c) = max(0, max_c o_c)
    int arg = -1;
    for (int c0 = 0; c0 + k <= C; ++c0) {
      float o = bias;
      for (int dk = 0; dk < k; ++dk) {
        const float* e = emb + (c0 + dk) * cd;
        const float* f = F + (size_t)dk * cd * nchk + chk;
        for (int d = 0; d < cd; ++d) o = fmaf(e[d], f[(size_t)d * nchk], o);
      }
      if (o > best) { best = o; arg = c0; }
    }
    out[wd + tid] = best;
    a.char_arg[(size_t)row * NCH + tid] = arg;
  }
}

// backward: a block walks `words_per_block` words and keeps the filter / table gradients in LDS
__global__ __launch_bounds__(128) void embed_bwd_kernel(EmbedArgs a, EmbedGrads gr, DropCfg drop, int nrows,
                                                        int words_per_block) {
  extern __shared__ float sm[];
  const int wd = a.word_dim, cd = a.char_dim, C = a.C;
  const int nfil = cd * (1 * 10 + 2 * 20 + 3 * 30 + 4 * 40);
  float* emb = sm;                       // [C][cd]  dropped char embeddings
  float* demb = emb + C * cd;            // [C][cd]
  float* dF = demb + C * cd;             // [nfil] concatenated filter grads (k=1..4)
  float* dB = dF + nfil;                 // [NCH]
  float* dT = dB + NCH;                  // [(num_chars-1)][cd]
  float* dU = dT + (a.num_chars - 1) * cd;   // [wd] unk
  const int tid = threadIdx.x;
  const int ntot = nfil + NCH + (a.num_chars - 1) * cd + wd;
  for (int i = tid; i < ntot; i += 128) dF[i] = 0.f;
  const int foff[4] = {0, cd * 10, cd * 10 + 2 * cd * 20, cd * 10 + 2 * cd * 20 + 3 * cd * 30};
  const int row0 = blockIdx.x * words_per_block;
  for (int row = row0; row < min(row0 + words_per_block, nrows); ++row) {
    __syncthreads();
    for (int idx = tid; idx < C * cd; idx += 128) {
      const int c = idx / cd, d = idx - c * cd;
      const int cid = a.char_ids[(size_t)row * C + c];
      float v = cid > 0 ? a.char_table[(size_t)(cid - 1) * cd + d] : 0.f;
      if (drop.enabled) v *= drop1(drop, HUAL_SITE_CHAR, (uint32_t)(row * C + c), d);
      emb[idx] = v;
      demb[idx] = 0.f;
    }
    const float* dcat = gr.dcat + (size_t)row * gr.lddcat;
    const int wid = a.word_ids[row];
    if (wid == 1)
      for (int c = tid; c < wd; c += 128) {
        float g = dcat[c];
        if (drop.enabled) g *= drop1(drop, HUAL_SITE_WORD, (uint32_t)row, c);
        dU[c] += g;
      }
    __syncthreads();
    // filter / bias gradients: thread per channel (each thread owns its channel's filter columns -> no race)
    if (tid < NCH) {
      const int arg = a.char_arg[(size_t)row * NCH + tid];
      if (arg >= 0) {
        int k, chk, nchk;
        chan_to_kernel(tid, k, chk, nchk);
        const float g = dcat[wd + tid];
        dB[tid] += g;
        float* f = dF + foff[k - 1];
        for (int dk = 0; dk < k; ++dk)
          for (int d = 0; d < cd; ++d) f[((size_t)dk * cd + d) * nchk + chk] += g * emb[(arg + dk) * cd + d];
      }
    }
    // embedding gradients: thread per (c, d) gathers over channels (no race)
    for (int idx = tid; idx < C * cd; idx += 128) {
      const int c = idx / cd, d = idx - c * cd;
      float s = 0.f;
      for (int ch = 0; ch < NCH; ++ch) {
        const int arg = a.char_arg[(size_t)row * NCH + ch];
        if (arg < 0) continue;
        int k, chk, nchk;
        chan_to_kernel(ch, k, chk, nchk);
        const int dk = c - arg;
        if (dk < 0 || dk >= k) continue;
        s += dcat[wd + ch] * a.filt[k - 1][((size_t)dk * cd + d) * nchk + chk];
      }
      demb[idx] = s;
    }
    __syncthreads();
    // scatter into the char table gradient (several chars of a word may share an id -> serialise per thread over c)
    for (int d = tid; d < cd; d += 128)
      for (int c = 0; c < C; ++c) {
        const int cid = a.char_ids[(size_t)row * C + c];
        if (cid > 0) {
          float g = demb[c * cd + d];
          if (drop.enabled) g *= drop1(drop, HUAL_SITE_CHAR, (uint32_t)(row * C + c), d);
          dT[(size_t)(cid - 1) * cd + d] += g;
        }
      }
  }
  __syncthreads();
  for (int i = tid; i < nfil; i += 128) {
    int k = i < foff[1] ? 0 : (i < foff[2] ? 1 : (i < foff[3] ? 2 : 3));
    atomicAdd(gr.dfilt[k] + (i - foff[k]), dF[i]);
  }
  for (int i = tid; i < NCH; i += 128) {
    int k, chk, nchk;
    chan_to_kernel(i, k, chk, nchk);
    atomicAdd(gr.dfbias[k - 1] + chk, dB[i]);
  }
  for (int i = tid; i < (a.num_chars - 1) * cd; i += 128) atomicAdd(gr.dchar_table + i, dT[i]);
  for (int i = tid; i < wd; i += 128) atomicAdd(gr.dunk + i, dU[i]);
}

namespace hual {

int launch_embed_fwd(const EmbedArgs& a, int nrows, const DropCfg& drop, hipStream_t s) {
  HUAL_REQUIRE(a.C >= 4, "char_ids need at least 4 chars per word (conv width 4, VALID) - modules.py:33");
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(nrows), dim3(128), (size_t)a.C * a.char_dim * sizeof(float), s, a, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_embed_bwd(const EmbedArgs& a, const EmbedGrads& g, int nrows, const DropCfg& drop, hipStream_t s) {
  const int cd = a.char_dim;
  const int nfil = cd * (10 + 40 + 90 + 160);
  const size_t bytes = ((size_t)2 * a.C * cd + nfil + NCH + (size_t)(a.num_chars - 1) * cd + a.word_dim) * sizeof(float);
  HUAL_REQUIRE(bytes <= 160 * 1024, "embed_bwd: char filter gradients do not fit LDS");
  static bool attr = false;
  if (!attr) {
    HUAL_CHECK_HIP(hipFuncSetAttribute((const void*)embed_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  const int wpb = 16;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(cdiv(nrows, wpb)), dim3(128), bytes, s, a, g, drop, nrows, wpb);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
