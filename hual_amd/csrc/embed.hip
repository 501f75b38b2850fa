// Text-encoder front end: word lookup + char CNN (+ their backward).
//   word_embs  /root/reference/models/modules.py:8-16   table = [zeros; unk; frozen GloVe]
//   char_embs  /root/reference/models/modules.py:19-38  lookup ([zeros; char_table]) -> 4 x conv2d VALID
//              (widths 1..4 -> 10/20/30/40 channels) + bias -> relu -> max over chars (padding not masked)
// Output row = [word_emb(300) | char features(100)] (model.py:41), consumed by the query_conv1d GEMM.
//
// The four convolutions are ONE dense product on the right layout.  cemb holds the (dropped) char embeddings, one row
// of CP floats (char_dim padded with zeros to a multiple of 16) per char slot, rows in (word, slot) order.  The window
// that starts at slot row r and is 4 slots wide is then simply the row of length 4*CP at cemb + r*CP (leading dimension
// CP: overlapping rows), and with
//     Wall[dk*CP + d][off_k + n] = filter_k[dk][d][n]   (zero for dk >= k, for the padding columns and d >= char_dim)
// Yall = windows . Wall + ball gives every channel of every width at every start position in one [M x 128] GEMM
// (M = words * C).  Windows that run past a word's last slot are computed and ignored.
//   forward : the gather (lookups, dropout, ball) and the pre-split images of Wall / Wall^T ride in the step's prologue launch
//             (pack_weights_kernel: embed_gather.h, gemm.h PackExtra) -> launch_gemm_bf16 -> char_pool_kernel (relu, max over
//             the valid start positions, arg-max)
//   backward: char_pool_bwd_kernel scatters the channel gradients to the arg-max rows of dYall [M x 128];
//             d filters / d bias = windows^T . dYall is one more job of the step's weight-gradient launch (dw_kernel),
//             into the packed scratch dFall which embed_unpack_kernel adds to the real gradients;
//             d windows = dYall . Wall^T (launch_gemm) and embed_finish_kernel folds the overlapping windows back to
//             char slots, applies the dropout mask and accumulates char-table rows (LDS, then global atomics).
#include "embed.h"
#include "mproj.h"
#include "philox.h"
#include "prof.h"

using namespace hual;

#include "embed_gather.h"

namespace hual {
static inline int cpad(int char_dim) { return (char_dim + 15) / 16 * 16; }
EmbedLayout embed_layout(int nrows, int C, int char_dim) {
  EmbedLayout l;
  const int CP = cpad(char_dim);
  const size_t M = (size_t)nrows * C;
  l.CP = CP;
  size_t off = 0;
  l.cemb = off; off += (M + 4) * CP;
  l.ball = off; off += NALL;
  l.yall = off; off += M * NALL;
  l.dxall = off; off += M * 4 * CP;
  l.dfall = off; off += (size_t)4 * CP * NALL + NALL;
  l.total = off;
  return l;
}
}  // namespace hual

// relu + max over the C - k + 1 valid window starts; thread = (word, channel)
__global__ __launch_bounds__(256) void char_pool_kernel(EmbedArgs a, int nrows) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= nrows * NCH) return;
  const int row = gid / NCH, ch = gid - row * NCH;
  int k, chk;
  chan_to_kernel(ch, k, chk);
  const float* y = a.yall + (size_t)row * a.C * NALL + ch;
  float best = 0.f;                              // relu floor: max_p relu(o_p) = max(0, max_p o_p)
  int arg = -1;
  for (int p = 0; p + k <= a.C; ++p) {
    const float v = y[(size_t)p * NALL];
    if (v > best) { best = v; arg = p; }
  }
  a.cat[(size_t)row * a.ldcat + a.word_dim + ch] = best;
  a.char_arg[(size_t)row * NCH + ch] = arg;
}

// ------------------------------------------------------------------------------------------------------
// backward 1: dYall (in place of Yall): channel gradient at the arg-max start row, zero elsewhere;
//             the unk row gradient; zero the packed filter-gradient scratch.
__global__ __launch_bounds__(256) void char_pool_bwd_kernel(EmbedArgs a, EmbedGrads gr, DropCfg drop, int nrows, int CP,
                                                            int ntask_pool) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int C = a.C, wd = a.word_dim;
  if (gid < ntask_pool) {            // (word, column of dYall)
    const int row = gid / NALL, col = gid - row * NALL;
    int arg = -1;
    float g = 0.f;
    if (col < NCH) {
      arg = a.char_arg[(size_t)row * NCH + col];
      if (arg >= 0) g = gr.dcat[(size_t)row * gr.lddcat + wd + col];
    }
    float* y = a.yall + (size_t)row * C * NALL + col;
    for (int p = 0; p < C; ++p) y[(size_t)p * NALL] = p == arg ? g : 0.f;
    return;
  }
  int x = gid - ntask_pool;
  const int ngw = wd >> 2;
  if (x < nrows * ngw) {             // unk row: d unk += dropout'(dcat[:, :wd]) over the words that ARE unk
    const int row = x / ngw, c4 = x - row * ngw;
    if (a.word_ids[row] == 1) {
      float4 g = ld4(gr.dcat + (size_t)row * gr.lddcat + 4 * c4);
      if (drop.enabled) g = apply_drop4(drop, HUAL_SITE_WORD, (uint32_t)row, (uint32_t)c4, g);
      atomicAdd(gr.dunk + 4 * c4, g.x); atomicAdd(gr.dunk + 4 * c4 + 1, g.y);
      atomicAdd(gr.dunk + 4 * c4 + 2, g.z); atomicAdd(gr.dunk + 4 * c4 + 3, g.w);
    }
    return;
  }
  x -= nrows * ngw;
  if (x < 4 * CP * NALL + NALL) a.dfall[x] = 0.f;
}

// backward 2 (embed_gather.h embed_finish_block) as a launch of its own
__global__ __launch_bounds__(256) void embed_finish_kernel(EmbedArgs a, EmbedGrads gr, DropCfg drop, int nrows, int CP) {
  extern __shared__ float dT[];      // [(num_chars-1) * cd]
  embed_finish_block(a, gr, drop, nrows, CP, blockIdx.x, dT);
}

// backward 3 (after the weight-gradient launch): packed dFall / dball -> filter and bias gradients (embed_gather.h embed_unpack_task;
// in the training step the tasks ride in the launch that folds the per-workgroup partial sums, rowops.h launch_colsum)
__global__ __launch_bounds__(256) void embed_unpack_kernel(EmbedArgs a, EmbedGrads gr, int CP) {
  embed_unpack_task(a, gr, CP, blockIdx.x * 256 + threadIdx.x);
}

namespace hual {

static int check_args(const EmbedArgs& a) {
  HUAL_REQUIRE(a.C >= 4, "char_ids need at least 4 chars per word (conv width 4, VALID) - modules.py:33");
  HUAL_REQUIRE((a.word_dim % 4) == 0 && a.char_dim >= 1, "embed: word_dim must be a multiple of 4");
  HUAL_REQUIRE(a.cemb && a.wall_img && a.ball && a.yall && a.dxall && a.dfall, "embed: null scratch");
  return 0;
}

int embed_cpad(int char_dim) { return cpad(char_dim); }
int embed_gather_tasks(const EmbedArgs& a, int nrows) {
  const int CP = cpad(a.char_dim);
  return nrows * (a.word_dim / 4 + a.C * (CP / 4)) + 4 * CP + NALL;
}

// (the gather ran in the step's prologue launch) char CNN as one split-operand product on the pre-split image of Wall, then the pooling
int launch_embed_fwd(const EmbedArgs& a, int nrows, const DropCfg& drop, hipStream_t s) {
  int rc = check_args(a);
  if (rc) return rc;
  const int CP = cpad(a.char_dim);
  // windows . Wall + ball: the 4 CP-deep windows (overlapping rows: window r = slot rows r .. r + 3, row stride CP) against the
  // pre-split image of the filter bank, ceil(4 CP / 128) weight steps (mproj.h)
  MProjArgs g{};
  g.R = nrows * a.C; g.MT = mproj_rows(g.R); g.nsteps = 1;
  MProjStep& st = g.s[0];
  st.A = a.cemb; st.lda = CP; st.rep = cdiv(4 * CP, 128); st.ktot = 4 * CP; st.kw = 4 * CP < 128 ? 4 * CP : 128; st.wimg = a.wall_img; st.wrows = 4 * CP;
  st.first = 1; st.last = 1; st.drop_site = -1; st.add_div = 1;
  st.bias = a.ball; st.out = a.yall; st.ldo = NALL; st.ncol = NALL;
  // relu + max over the window starts: in the same launch (the tile never leaves the registers: mproj.h pool epilogue) when the
  // chars per word are a power of two that divides the rows of a workgroup, else a launch of its own on the stored tile
  const int C = a.C;
  const int mtp = (g.MT + C - 1) / C * C;
  if ((C & (C - 1)) == 0 && C <= 16 && mtp <= 64) {
    g.MT = mtp;
    st.out = nullptr;
    g.pool_cat = a.cat; g.pool_ldcat = a.ldcat; g.pool_col0 = a.word_dim; g.pool_arg = a.char_arg; g.pool_C = C;
    return launch_mproj(&g, 1, drop, s);
  }
  rc = launch_mproj(&g, 1, drop, s);
  if (rc) return rc;
  HUAL_LAUNCH(0.0, 0.0, char_pool_kernel, dim3(cdiv(nrows * NCH, 256)), dim3(256), 0, s, a, nrows);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int embed_finish_blocks(const EmbedArgs& a, int nrows) { return cdiv(nrows * a.C, EF_ROWS); }
int embed_finish_lds(const EmbedArgs& a) { return (a.num_chars - 1) * a.char_dim * (int)sizeof(float); }
// finish = false: the caller lets the last step (embed_finish_block) ride in a later launch (rowops.h launch_colsum: EmbedUnpack)
int launch_embed_bwd(const EmbedArgs& a, const EmbedGrads& g, int nrows, const DropCfg& drop, hipStream_t s, DwJob* dwjob, bool finish) {
  int rc = check_args(a);
  if (rc) return rc;
  HUAL_REQUIRE(dwjob != nullptr, "embed_bwd: null weight-gradient job");
  const int cd = a.char_dim, CP = cpad(cd), M = nrows * a.C;
  const int ntask_pool = nrows * NALL;
  const int ntail = nrows * (a.word_dim / 4) + 4 * CP * NALL + NALL;
  HUAL_LAUNCH(0.0, 0.0, char_pool_bwd_kernel, dim3(cdiv(ntask_pool + ntail, 256)), dim3(256), 0, s, a, g, drop, nrows, CP,
              ntask_pool);
  HUAL_CHECK_HIP(hipGetLastError());
  // d filters / d bias: one more job for the step's weight-gradient launch
  embed_dw_job(a, nrows, dwjob);
  // d windows = dYall . Wall^T
  HUAL_REQUIRE(a.wall_img_t != nullptr && cdiv(4 * CP, 128) <= MP_MAX, "embed_bwd: image of the transposed filter bank");
  MProjArgs j{};
  j.R = M; j.MT = mproj_rows(M); j.nsteps = cdiv(4 * CP, 128);
  for (int p = 0; p < j.nsteps; ++p) {      // column block p of d windows: one operand, ceil(4 CP / 128) images of Wall^T
    MProjStep& st = j.s[p];
    st.A = a.yall; st.lda = NALL; st.kw = NALL; st.reuse = p > 0 ? 1 : 0; st.rep = 1;
    st.wimg = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.wall_img_t) + (size_t)p * HUAL_PACK_BLOCK_BYTES); st.wrows = NALL;
    st.first = 1; st.last = 1; st.drop_site = -1; st.add_div = 1;
    st.out = a.dxall + (size_t)p * 128; st.ldo = 4 * CP; st.ncol = 4 * CP - 128 * p < 128 ? 4 * CP - 128 * p : 128;
  }
  rc = launch_mproj(&j, 1, drop, s);
  if (rc) return rc;
  const size_t lds = (size_t)(a.num_chars - 1) * cd * sizeof(float);
  HUAL_REQUIRE(lds <= 64 * 1024, "embed_bwd: char table too large for the LDS accumulator");
  if (!finish) return 0;
  HUAL_LAUNCH(0.0, 0.0, embed_finish_kernel, dim3(cdiv(M, EF_ROWS)), dim3(256), lds, s, a, g, drop, nrows, CP);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

void embed_dw_job(const EmbedArgs& a, int nrows, DwJob* dwjob) {
  const int CP = cpad(a.char_dim);
  dw_job_init(*dwjob);
  dwjob->npieces = 1;
  dwjob->A[0] = a.cemb; dwjob->lda[0] = CP; dwjob->kw[0] = 4 * CP; dwjob->dW[0] = a.dfall; dwjob->ldw = NALL;
  dwjob->dY = a.yall; dwjob->ldy = NALL; dwjob->M = nrows * a.C; dwjob->N = NALL;
  dwjob->db = a.dfall + (size_t)4 * CP * NALL;
}

int embed_unpack_tasks(const EmbedArgs& a) { return a.char_dim * 300 + NCH; }
int embed_unpack_cpad(const EmbedArgs& a) { return cpad(a.char_dim); }
int launch_embed_unpack(const EmbedArgs& a, const EmbedGrads& g, hipStream_t s) {
  const int cd = a.char_dim;
  const int n = cd * 300 + NCH;
  HUAL_LAUNCH(0.0, 0.0, embed_unpack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, a, g, cpad(cd));
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
