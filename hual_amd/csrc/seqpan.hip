// Whole-graph orchestration of the SeqPAN hot path: the launch sequence that replaces
// `sess.run([...])` on /root/reference/models/model.py:29-122, plus its hand-derived backward pass.
// Everything here only ENQUEUES kernels on the caller's stream (no allocation, no sync) so that a training step
// can be captured into a hipGraph.  Buffers are carved out of the caller's workspace by a deterministic bump
// allocator; a "dry" pass of the very same code computes the workspace size and the name table.
#include "attn.h"
#include "prof.h"
#include "cq.h"
#include "convblock.h"
#include "dablock.h"
#include "embed.h"
#include "gemm.h"
#include "heads.h"
#include "mproj.h"
#include "optim.h"
#include "params.h"
#include "rowops.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <unordered_map>
#include <vector>

namespace hual {
const char* last_error_cstr();
}
using namespace hual;

namespace {

// stages of the graph (forward order; backward runs them in reverse)
enum { ST_ALWAYS = 0, ST_INPUT = 1, ST_CONV = 2, ST_DA = 3, ST_CQ = 4, ST_FUSE = 5, ST_PRED = 6 };

struct WsEntry { std::string name; size_t off, rows, cols; };

struct Ctx {
  const hual_cfg* cfg = nullptr;
  ParamMap pm;
  int B = 0, T = 0, L = 0, C = 0;
  RowSpace rs;      // unified rows (video + query)
  RowSpace rsv;     // video rows only (predictor)
  char* base = nullptr;
  size_t cap = 0, used = 0;
  bool dry = true;
  hipStream_t stream = nullptr;
  DropCfg drop;
  const float* P = nullptr;   // flat params
  float* G = nullptr;         // flat grads
  const float* word_table = nullptr;
  std::vector<WsEntry> entries;
  std::unordered_map<std::string, size_t> index;
  std::vector<DwJob> dwjobs;
  std::vector<ColsumJob> colsum;
  int rc = 0;
  bool static_tables = false;   // hual_run_opts.static_tables
  void* ext_table = nullptr; size_t ext_table_bytes = 0;      // hual_run_opts.dw_table
  int wall_K = 0; uint32_t wall_off = 0, wall_boff = 0;      // packed char-CNN filter bank inside PKT / PKN (setup_ctx)
  struct DenseW { size_t off; int K; size_t boff; uint8_t need; };      // need: HUAL_PACK_* images the kernels read of this weight
  bool ksplit = false;                // feature-load phase on the K-split kernel (plan)
  typedef DenseW DenseW_t;
  std::vector<DenseW> dense;          // every [K,128] weight of the graph, sorted by offset
  char* PKF = nullptr;                // pre-split LDS images (gemm.h launch_pack_weights) of the feature-load kernel's weights
  char* PKT = nullptr;                // images for the register-resident weights of the T-form kernels (tilecore.h): W^T blocks (forward) ...
  char* PKN = nullptr;                // ... and W blocks (dX), 64 KB per 128 contraction indices, at DenseW::boff
  const float* timg(size_t off, int blk = 0) const { return reinterpret_cast<const float*>(PKT + boff_of(off) + (size_t)blk * 65536); }
  const float* nimg(size_t off, int blk = 0) const { return reinterpret_cast<const float*>(PKN + boff_of(off) + (size_t)blk * 65536); }
  size_t boff_of(size_t off) const {
    for (const auto& d : dense) if (d.off == off) return d.boff;
    return off * 4;      // (virtual jobs: the char-CNN filter bank sits at wall_boff = wall_off * 4)
  }
  size_t pk_bytes = 0;

  float* buf(const std::string& name, size_t rows, size_t cols) {
    auto it = index.find(name);
    if (it != index.end()) return reinterpret_cast<float*>(base + entries[it->second].off);
    if (!dry) { if (rc == 0) rc = fail(HUAL_ERR_WORKSPACE, "internal: workspace buffer requested outside the planning pass"); return reinterpret_cast<float*>(base); }
    WsEntry e{name, used, rows, cols};
    size_t bytes = (rows * cols * sizeof(float) + 255) & ~(size_t)255;
    used += bytes + 256;     // 256 B guard: clamped fragment loads never leave the workspace
    index[name] = entries.size();
    entries.push_back(e);
    return reinterpret_cast<float*>(base + e.off);
  }
  float* act(const std::string& name) { return buf(name, (size_t)rs.R, HUAL_D); }     // [R,128]
  // bit plane of a [rows,128] tensor (csrc/tilecore.h): 16 bytes per row
  uint8_t* bits(const std::string& name, size_t rows) { return reinterpret_cast<uint8_t*>(buf(name, rows, 4)); }
  int novf = 0;               // words of "params.ovf" the loss launch reads
  bool debug_taps = false;    // hual_run_opts.debug_taps: also write the tensors only parity tests read (conv_block relu outputs)
  float* actv(const std::string& name) { return buf(name, (size_t)rs.Nv, HUAL_D); }   // [Nv,128]
  float* vec(const std::string& name) { return buf(name, (size_t)rs.R, 1); }
  const float* p(size_t off) const { return P + off; }
  float* g(size_t off) const { return G + off; }
  bool ok() const { return rc == 0; }
  void chk(int r) { if (rc == 0 && r != 0) rc = r; }
  // ---- stage selection (per-block entry points, include/hual_seqpan.h): the graph is cut into stages; with sel_stage >= 0
  // only the prologue (masks, weight images) and that stage enqueue work - buffer naming / allocation is unaffected
  int sel_stage = -1, sel_sub = 0;
  bool active = true;
  bool want_bwd = false;      // forward keeps what a backward pass needs (labels given, or a block entry point asks for it)
  int part_seq = 0;           // running id of the per-block partial-sum scratch buffers (stable whatever runs)
  void stage(int st, int sub = 0) { active = sel_stage < 0 || st == ST_ALWAYS || (st == sel_stage && (st != ST_DA || sub == sel_sub)); }
  bool stage_on(int st, int sub = 0) const { return sel_stage < 0 || (st == sel_stage && (st != ST_DA || sub == sel_sub)); }
  bool live() const { return !dry && active && rc == 0; }
  void push_dw(const DwJob& j) { if (active) dwjobs.push_back(j); }

  // ---- launch wrappers (skipped in the dry pass) ----
  // layer-norm backward; the per-block dgamma / dbeta sums go to scratch and are folded in by flush_colsum()
  void ln_bwd(const LnBwd& a0) {
    LnBwd a = a0;
    const int nblk = ln_bwd_blocks(a.R);
    a.part = buf("part." + std::to_string(part_seq++), (size_t)nblk * 4, HUAL_D);
    ColsumJob cj{};
    cj.src = a.part; cj.nblk = nblk; cj.nvec = 4;      // scratch layout [blk][4][128]; null dst = unused vector
    cj.dst[0] = a.dg1; cj.dst[1] = a.db1; cj.dst[2] = a.dg2; cj.dst[3] = a.db2;
    if (active) colsum.push_back(cj);
    if (live()) chk(launch_ln_bwd(a, drop, stream));
  }
  // the two input layer norms (rows below / from a.split on) in one launch: two reductions of the partial sums
  void ln_bwd_split(const LnBwd& a0, float* dg_lo, float* db_lo, float* dg_hi, float* db_hi, const PosBwdJob* pos = nullptr, int npos = 0) {
    LnBwd a = a0;
    const int nlo = ln_bwd_blocks(a.split), nhi = ln_bwd_blocks(a.R - a.split);
    a.part = buf("part." + std::to_string(part_seq++), (size_t)(nlo + nhi) * 4, HUAL_D);
    ColsumJob cj{};
    cj.src = a.part; cj.nblk = nlo; cj.nvec = 4;
    cj.dst[0] = dg_lo; cj.dst[1] = db_lo;
    if (active) colsum.push_back(cj);
    cj.src = a.part + (size_t)nlo * 4 * HUAL_D; cj.nblk = nhi;
    cj.dst[0] = dg_hi; cj.dst[1] = db_hi;
    if (active) colsum.push_back(cj);
    if (live()) chk(launch_ln_bwd(a, drop, stream, pos, npos, &rs));
  }
  // fused dX products + layer norm(s) backward (dablock.h); the per-workgroup parameter sums are folded in by flush_colsum()
  void ln_proj_bwd(const LnProjBwdArgs& a0, float* dg1, float* db1, float* dg2, float* db2) {
    LnProjBwdArgs a = a0;
    a.MT = ln_proj_bwd_rows(a.R, a.Nv);
    const int nblk = ln_proj_bwd_blocks(a.R, a.Nv);
    a.part = buf("part." + std::to_string(part_seq++), (size_t)nblk * 4, HUAL_D);
    ColsumJob cj{};
    cj.src = a.part; cj.nblk = nblk; cj.nvec = 4;
    cj.dst[0] = dg1; cj.dst[1] = db1; cj.dst[2] = dg2; cj.dst[3] = db2;
    if (active) colsum.push_back(cj);
    if (live()) chk(launch_ln_proj_bwd(a, drop, stream));
  }
  // (unpack: the char-CNN filter-gradient unpack rides in the last of these launches - it has to follow the weight-gradient launch)
  void flush_colsum(const EmbedUnpack* unpack = nullptr) {
    for (size_t i = 0; i < colsum.size() && !dry && ok(); i += HUAL_COLSUM_MAX_JOBS) {   // (whatever stages ran)
      const bool last = i + HUAL_COLSUM_MAX_JOBS >= colsum.size();
      chk(launch_colsum(colsum.data() + i, (int)std::min<size_t>(HUAL_COLSUM_MAX_JOBS, colsum.size() - i), stream, last ? unpack : nullptr));
    }
    if (colsum.empty() && unpack && !dry && ok()) chk(launch_colsum(nullptr, 0, stream, unpack));
    colsum.clear();
  }
  void attn_fwd(const AttnJob* j, int n) { if (live()) chk(launch_attn_fwd(j, n, drop, stream)); }
  void attn_bwd(const AttnJob* j, int n) {
    if (live()) chk(launch_attn_bwd(j, n, drop, stream));
  }
};

void set_embed_scratch(EmbedArgs& ea, float* base, int Nq, int C, int char_dim) {
  const EmbedLayout el = embed_layout(Nq, C, char_dim);
  ea.cemb = base + el.cemb; ea.ball = base + el.ball;
  ea.yall = base + el.yall; ea.dxall = base + el.dxall; ea.dfall = base + el.dfall;
}
// everything of EmbedArgs that does not depend on the gradients
void fill_embed_args(Ctx& c, EmbedArgs& ea, const hual_batch* bt, float* cat, int catw, int32_t* char_arg, float* scratch) {
  const ParamMap& pm = c.pm;
  ea.word_ids = bt->word_ids; ea.char_ids = bt->char_ids; ea.word_table = c.word_table; ea.unk = c.p(pm.unk);
  ea.char_table = c.p(pm.char_table);
  for (int i = 0; i < 4; ++i) { ea.filt[i] = c.p(pm.filt[i]); ea.fbias[i] = c.p(pm.fbias[i]); }
  ea.cat = cat; ea.ldcat = catw; ea.char_arg = char_arg;
  set_embed_scratch(ea, scratch, c.rs.Nq, c.C, c.cfg->char_dim);
  ea.word_dim = c.cfg->word_dim; ea.char_dim = c.cfg->char_dim; ea.C = c.C; ea.num_chars = c.cfg->num_chars;
  ea.wall_img = reinterpret_cast<const float*>(c.PKT + c.wall_boff);      // T image (forward), N image (d windows): tilecore.h
  ea.wall_img_t = reinterpret_cast<const float*>(c.PKN + c.wall_boff);
}
// keep-byte buffer of an attention job's probability dropout (attn.h): B*Tq*8 rows of ldm bytes
// + the softmax statistics the forward leaves for the backward
void set_dmask(Ctx& c, AttnJob& a, const std::string& name) {
  a.dmask = reinterpret_cast<uint8_t*>(c.buf(name, (attn_keep_bytes(a.B, a.Tq, a.Tk) + 3) / 4, 1));
  a.stats = c.buf(name + ".st", (size_t)2 * a.B * a.Tq * 8, 1);
}
DwJob mkdw(const float* A, int lda, int K, const float* dY, int ldy, int M, float* dW, float* db, int N = HUAL_D) {
  DwJob j;
  dw_job_init(j);
  j.npieces = 1;
  j.A[0] = A; j.lda[0] = lda; j.kw[0] = K; j.dW[0] = dW; j.ldw = N;
  j.dY = dY; j.ldy = ldy; j.M = M; j.N = N; j.db = db;
  return j;
}

// ---- steps of the multi-step dense kernel (mproj.h)
// one 128-deep (or shorter) operand block A[rows, kw] times the image `img` (wrows valid K rows)
MProjStep mstep(const float* A, int lda, int kw, const float* img, int wrows, bool first, bool last) {
  MProjStep s{};
  s.A = A; s.lda = lda; s.kw = kw; s.wimg = img; s.wrows = wrows; s.first = first ? 1 : 0; s.last = last ? 1 : 0;
  s.drop_site = -1; s.rep = 1; s.add_div = 1; s.ncol = HUAL_D;
  return s;
}
MProjStep mstep_reuse(const float* img, int wrows, bool first, bool last) {
  MProjStep s = mstep(nullptr, 0, 0, img, wrows, first, last);
  s.reuse = 1;
  return s;
}
void mstep_out(MProjStep& s, float* out, int ldo, const float* bias = nullptr, int ncol = HUAL_D) { s.out = out; s.ldo = ldo; s.bias = bias; s.ncol = ncol; }
MProjArgs margs(int R, int R_other = 0) {      // R_other: rows of the problem launched beside this one
  MProjArgs a{};
  a.R = R; a.MT = mproj_rows(R, R_other); a.drop_row0 = 0;
  return a;
}

// Launch every weight-gradient job of the step as ONE persistent launch (gemm.h launch_dw): long-row jobs first, so that the
// launch ends on the short ones.  (Measured: flushing earlier, under the dX chain, or on a side stream costs more than it
// hides - the two compete for the same CUs.)
void flush_dw(Ctx& c) {
  const size_t n = c.dwjobs.size();
  if (n == 0) return;
  std::stable_sort(c.dwjobs.begin(), c.dwjobs.end(), [](const DwJob& a, const DwJob& b) { return a.M > b.M; });
  DwJob* table = reinterpret_cast<DwJob*>(c.buf("dw.table.0", 1, dw_table_words(n)));
  if (!c.dry && c.ext_table) {       // the caller's own table (one per padded shape of an epoch loop) instead of the workspace copy
    if (4 * dw_table_words(n) > c.ext_table_bytes) { c.chk(fail(HUAL_ERR_INVALID, "hual_run_opts.dw_table too small: hual_seqpan_dw_table_bytes()")); return; }
    table = reinterpret_cast<DwJob*>(c.ext_table);
  }
  if (!c.dry && c.ok()) c.chk(launch_dw(c.dwjobs.data(), (int)n, c.drop, c.stream, table, !c.static_tables));
  c.dwjobs.clear();
}

int setup_ctx(Ctx& c, const hual_cfg* cfg, int B, int T, int L, int C) {
  HUAL_REQUIRE(cfg != nullptr, "null cfg");
  int rc = build_param_map(*cfg, c.pm);
  if (rc) return rc;
  HUAL_REQUIRE(B >= 1 && T >= 1 && L >= 1, "empty batch");
  HUAL_REQUIRE(T <= cfg->max_vlen && L <= cfg->max_vlen, "sequence longer than max_vlen (assert_less_equal, modules.py:44)");
  HUAL_REQUIRE(T <= 256 && L <= 256, "T, L <= 256");
  HUAL_REQUIRE(C >= 4, "char_ids need C >= 4 (conv width 4, VALID)");
  HUAL_REQUIRE((long long)B * (T + L) * 8 < (1ll << 28), "batch too large for 32-bit RNG row ids");
  HUAL_REQUIRE((long long)B * (T + L) < (1ll << 20), "batch too large: B (T + L) < 2^20 rows (row -> clip lookups, common.h small_div)");
  c.cfg = cfg; c.B = B; c.T = T; c.L = L; c.C = C;
  {
    const ParamMap& pm = c.pm;
    const int D = HUAL_D, catw = cfg->word_dim + 100;
    c.dense.clear();
    // which images of a weight the kernels read: F - the LDS-DMA image of the K-split feature load; T / N - forward and dX products
    // with register-resident weights: every other dense kernel (gemm.h HUAL_PACK_*, tilecore.h)
    const uint8_t TN = HUAL_PACK_T | HUAL_PACK_N;
    auto add = [&](size_t off, int K, uint8_t need) { if (K % 8 == 0) c.dense.push_back({off, K, off * 4, need}); };
    const int qks = ((catw + 3) / 4 + 63) & ~63;        // quarter size of query_conv1d's K (multiple of 64)
    c.ksplit = (cfg->vdim % 256) == 0 && cfg->vdim <= 1024 && (catw % 8) == 0 && qks <= 256;
    add(pm.vconv.k, cfg->vdim, c.ksplit ? HUAL_PACK_F : HUAL_PACK_T);
    add(pm.qconv.k, catw, (c.ksplit ? HUAL_PACK_F : HUAL_PACK_T) | HUAL_PACK_N);
    add(pm.shid.k, 2 * D, TN); add(pm.ehid.k, 2 * D, TN);
    add(pm.fe_dense.k, D, TN); add(pm.fe_q.k, D, TN); add(pm.fe_k.k, D, TN); add(pm.fe_v.k, D, TN);
    for (int i = 0; i < 4; ++i) { add(pm.fe_cb.pw[i], D, TN); add(pm.cb.pw[i], D, TN); }
    add(pm.cqcat.k, 2 * D, TN); add(pm.cq[0].dense, 4 * D, TN); add(pm.cq[1].dense, 4 * D, TN);
    for (int li = 0; li < cfg->attn_layer; ++li) {
      const DualAttnP& d = pm.da[li];
      const size_t w[] = {d.dense1.k, d.dense2.k, d.bl1_d1, d.bl1_d2, d.bl2_d1, d.bl2_d2, d.guided.k, d.s_gate.k, d.x_gate.k,
                          d.s_dense.k, d.x_dense.k, d.query.k, d.f_key.k, d.f_value.k, d.t_key.k, d.t_value.k};
      // all sixteen in the T-form kernels (ln_proj, da_post; ln_proj_bwd, da_mid_bwd): T / N images
      for (size_t o : w) add(o, D, TN);
    }
    std::sort(c.dense.begin(), c.dense.end(), [](const Ctx::DenseW& a, const Ctx::DenseW& b) { return a.off < b.off; });
    // the image of a transposed weight takes ceil(K/128) blocks of 64 KB: it fits the weight's own byte range when K is a
    // multiple of 128; the others (query_conv1d, K = word_dim + 100) go behind the end of the parameter range
    size_t extra = pm.total * 4;
    extra = (extra + 255) & ~(size_t)255;
    for (auto& d : c.dense)
      if (d.K % 128) { d.boff = extra; extra += (size_t)((d.K + 127) / 128) * HUAL_PACK_BLOCK_BYTES; }
    // the packed char-CNN filter bank Wall [4 CP, 128] (embed.hip): T image at PKT + wall_boff, N image at PKN + wall_boff
    c.wall_K = 4 * embed_cpad(cfg->char_dim);
    c.wall_off = (uint32_t)(extra / 4);
    c.wall_boff = (uint32_t)extra;
    const size_t wf = (size_t)c.wall_K * 512, wb = (size_t)((c.wall_K + 127) / 128) * HUAL_PACK_BLOCK_BYTES;
    extra += wf > wb ? wf : wb;
    c.pk_bytes = extra;
  }
  c.rs.B = B; c.rs.T = T; c.rs.L = L; c.rs.Nv = B * T; c.rs.Nq = B * L; c.rs.R = B * (T + L); c.rs.rowmask = nullptr;
  c.rsv = c.rs; c.rsv.Nq = 0; c.rsv.L = 0; c.rsv.R = c.rs.Nv;
  return 0;
}

// ======================================================================================================
// shared sub-graphs
// ======================================================================================================

// conv_block (modules.py:59-70) on rows described by `rs`; x0 -> returns x4.  tag prefixes the buffer names.
// pos_src / pos (predictor feature encoder, modules.py:124): x0 = pos_src + pos[t] is formed on the way in.
// layer norms (LN1, LN_t) + the five projections of dual attention layer li (dablock.h ln_proj; x is set by the caller)
void fill_da_ln_proj(Ctx& c, LnProjArgs& lp, int li, int R, int Nv) {
  const int D = HUAL_D;
  const DualAttnP& d = c.pm.da[li];
  const std::string t = "da" + std::to_string(li);
  lp = LnProjArgs{};
  lp.g1 = c.p(d.ln1.g); lp.b1 = c.p(d.ln1.b); lp.y1 = c.act(t + ".ln1"); lp.drop_site1 = -1; lp.pre_site = -1;
  lp.g2 = c.p(d.lnt.g); lp.b2 = c.p(d.lnt.b); lp.y2 = c.act(t + ".lnt"); lp.mean = c.vec(t + ".mean"); lp.rstd = c.vec(t + ".rstd");
  lp.nproj = 5; lp.R = R; lp.Nv = Nv; lp.MT = ln_proj_rows(R, Nv); lp.drop_row0 = 0;
  float* qkv = c.buf(t + ".qkv", R, 3 * D);
  float* ktvt = c.buf(t + ".ktvt", R, 2 * D);
  const DenseP* pr[5] = {&d.query, &d.f_key, &d.f_value, &d.t_key, &d.t_value};
  float* outs[5] = {qkv, qkv + D, qkv + 2 * D, ktvt, ktvt + D};
  for (int k = 0; k < 5; ++k) {
    lp.wimg[k] = c.timg(pr[k]->k); lp.bias[k] = c.p(pr[k]->b); lp.out[k] = outs[k]; lp.ldo[k] = k < 3 ? 3 * D : 2 * D; lp.src[k] = k < 3 ? 0 : 1;
    lp.out_site[k] = -1;
  }
}

// tail: the ln_proj launch that follows on the block output (its x is filled in here); it rides in the block's launch when the
// whole graph runs (the per-block entry points keep the two launches: the caller launches *tail itself when this returns false)
float* conv_block_fwd(Ctx& c, const std::string& tag, float* x, const ConvBlockP& cp, const RowSpace& rs, int site0,
                      const float* pos_src = nullptr, const float* pos = nullptr, LnProjArgs* tail = nullptr, bool* tail_done = nullptr) {
  const int R = rs.R;
  {                      // all four layers in one launch (convblock.h)
    CbFwdArgs a{};
    a.x0 = pos_src ? pos_src : x; a.pos = pos; a.x0_out = pos_src ? x : nullptr;
    a.MT = conv_block_fused_rows(R, rs.Nq > 0 ? rs.Nv : 0); a.drop_row0 = 0;
    float* xin = x;
    for (int i = 0; i < 4; ++i) {
      const std::string is = std::to_string(i);
      CbLayerFwd& L = a.l[i];
      L.c = c.buf(tag + ".c" + is, R, HUAL_D);
      L.y = c.buf(tag + ".y" + is, R, HUAL_D);
      if (!c.debug_taps) L.y = nullptr;                       // backward reads the bit planes, not y
      L.relu_bits = c.bits(tag + ".rb" + is, R);
      L.keep_bits = c.bits(tag + ".kb" + is, R);
      L.xout = c.buf(tag + ".x" + std::to_string(i + 1), R, HUAL_D);
      L.mean = c.buf(tag + ".mean" + is, R, 1);
      L.rstd = c.buf(tag + ".rstd" + is, R, 1);
      L.ln_g = c.p(cp.ln[i].g); L.ln_b = c.p(cp.ln[i].b); L.dw = c.p(cp.dw[i]);
      L.wimg = c.timg(cp.pw[i]); L.bias = c.p(cp.b[i]);      // T image: conv_block_fwd_kernel keeps its weights in registers
      L.drop_site = site0 + i;
      xin = L.xout;
    }
    static const bool no_tail = getenv("HUAL_CB_NO_TAIL") != nullptr && atoi(getenv("HUAL_CB_NO_TAIL")) != 0;      // (A/B timings)
    const bool fuse = tail && c.sel_stage < 0 && !no_tail;
    if (tail) tail->x = xin;
    if (tail_done) *tail_done = fuse;
    if (c.live()) c.chk(launch_conv_block_fwd(a, rs, c.drop, c.stream, fuse ? tail : nullptr));
    return xin;
  }
}

// backward of conv_block: d_out = gradient wrt x4 -> returns gradient wrt x0
float* conv_block_bwd(Ctx& c, const std::string& tag, float* x0, float* dx, const ConvBlockP& cp, const RowSpace& rs, int site0) {
  const int R = rs.R;
  {                      // all four layers in one launch (convblock.h); dZ_3 is formed inside from dx and y3
    CbBwdArgs a{};
    a.dx_in = dx; a.relu_bits3 = c.bits(tag + ".rb3", R); a.keep_bits3 = c.bits(tag + ".kb3", R);
    a.dx_out = c.buf("d." + tag + ".x0", R, HUAL_D);
    a.MT = conv_block_fused_rows_bwd(R, rs.Nq > 0 ? rs.Nv : 0); a.drop_row0 = 0;
    const int nblk = conv_block_bwd_blocks(R, rs.Nq > 0 ? rs.Nv : 0);
    for (int i = 3; i >= 0; --i) {
      const std::string is = std::to_string(i);
      CbLayerBwd& L = a.l[i];
      L.ln_g = c.p(cp.ln[i].g); L.ln_b = c.p(cp.ln[i].b); L.dw = c.p(cp.dw[i]);
      L.wimg_t = c.nimg(cp.pw[i]);      // N image: conv_block_bwd_kernel keeps its weights in registers
      L.x = i == 0 ? x0 : c.buf(tag + ".x" + is, R, HUAL_D);
      L.mean = c.buf(tag + ".mean" + is, R, 1); L.rstd = c.buf(tag + ".rstd" + is, R, 1);
      L.relu_prev = i > 0 ? c.bits(tag + ".rb" + std::to_string(i - 1), R) : nullptr;
      L.keep_prev = i > 0 ? c.bits(tag + ".kb" + std::to_string(i - 1), R) : nullptr;
      L.dz = c.buf("d." + tag + ".z" + is, R, HUAL_D);
      L.dz_prev = i > 0 ? c.buf("d." + tag + ".z" + std::to_string(i - 1), R, HUAL_D) : nullptr;
      L.part = c.buf("part." + std::to_string(c.part_seq++), (size_t)nblk * 9, HUAL_D);
      ColsumJob cj{};
      cj.src = L.part; cj.nblk = nblk; cj.nvec = 9;
      for (int k = 0; k < 7; ++k) cj.dst[k] = c.g(cp.dw[i]) + k * HUAL_D;
      cj.dst[7] = c.g(cp.ln[i].g); cj.dst[8] = c.g(cp.ln[i].b);
      if (c.active) c.colsum.push_back(cj);
      c.push_dw(mkdw(c.buf(tag + ".c" + is, R, HUAL_D), HUAL_D, HUAL_D, L.dz, HUAL_D, R, c.g(cp.pw[i]), c.g(cp.b[i])));
    }
    if (c.live()) c.chk(launch_conv_block_bwd(a, rs, c.drop, c.stream));
    return a.dx_out;
  }
}

// ======================================================================================================
// forward
// ======================================================================================================
int forward_graph(Ctx& c, const hual_batch* bt, const hual_labels* lab, const hual_outputs* out, const hual_run_opts* opt) {
  const ParamMap& pm = c.pm;
  const RowSpace& rs = c.rs;
  const int Nv = rs.Nv, Nq = rs.Nq, R = rs.R, B = c.B, T = c.T, L = c.L;
  const int D = HUAL_D;
  const int catw = c.cfg->word_dim + 100;
  float* rowmask = c.vec("rowmask");
  c.rs.rowmask = rowmask;
  c.rsv.rowmask = rowmask;
  float* loss_acc = c.buf("loss_acc", 8, 1);
  c.stage(ST_ALWAYS);
  // ---------------- prologue: row masks (model.py:31-32), cleared loss accumulators, pre-split images of every dense weight for
  // the split kernels (weights are constant within a step) and - hual_run_opts.grads_prezero - the gradient buffer zeroed: ONE launch
  c.PKF = reinterpret_cast<char*>(c.buf("params.pkf", (c.pk_bytes + 3) / 4, 1));
  c.PKT = reinterpret_cast<char*>(c.buf("params.pkt", (c.pk_bytes + 3) / 4, 1));
  c.PKN = reinterpret_cast<char*>(c.buf("params.pkn", (c.pk_bytes + 3) / 4, 1));
  float* ortho_dE = c.buf("ortho.dE", 4, HUAL_D);
  // one word per workgroup of the pack launch's job rows: set when a weight does not fit the scaled fp16 images (gemm.h PackExtra);
  // the loss launch reads them
  std::vector<int> pkK;
  for (const auto& d : c.dense) pkK.push_back(d.K);
  pkK.push_back(c.wall_K);
  const int ovf_words = pack_ovf_words(pkK.data(), (int)pkK.size());
  uint32_t* ovf = reinterpret_cast<uint32_t*>(c.buf("params.ovf", (size_t)ovf_words, 1));
  c.novf = (int)pkK.size() <= HUAL_MAX_PACK ? ovf_words : 0;
  // (the text encoder's gather - word / char lookups with their dropout, model.py:36-41 - rides in the same launch)
  float* cat = c.buf("cat", Nq, catw);
  int32_t* char_arg = reinterpret_cast<int32_t*>(c.buf("char_arg", Nq, 100));
  float* embed_scratch = c.buf("embed.scratch", embed_layout(Nq, c.C, c.cfg->char_dim).total, 1);
  EmbedArgs ea{};
  if (!c.dry && c.ok()) {
    std::vector<uint32_t> offs, boffs;
    std::vector<int> Ks;
    std::vector<uint8_t> needs;
    for (const auto& d : c.dense) { offs.push_back((uint32_t)d.off); Ks.push_back(d.K); boffs.push_back((uint32_t)d.boff); needs.push_back(d.need); }
    offs.push_back(c.wall_off); Ks.push_back(c.wall_K); boffs.push_back(c.wall_boff);      // the char-CNN filter bank (virtual source)
    needs.push_back(HUAL_PACK_T | HUAL_PACK_N);
    fill_embed_args(c, ea, bt, cat, catw, char_arg, embed_scratch);
    PackExtra ex{};
    ex.lens = bt->video_seq_len; ex.word_ids = bt->word_ids; ex.rowmask = rowmask; ex.loss_acc = loss_acc; ex.B = B; ex.T = T; ex.L = L;
    ex.zero_ptr = (lab && opt->grads_prezero && opt->prezero_token) ? opt->grads_prezero : nullptr;
    ex.zero_n = (size_t)((pm.total + 3) & ~(size_t)3);
    // the orthogonality term of the label embeddings depends on the parameters only: evaluated here (ortho.h)
    ex.E = lab ? c.p(pm.label_emb) : nullptr; ex.lambda = c.cfg->match_lambda; ex.dE_ortho = ortho_dE;
    // (per-block entry points of other stages run without the text encoder's inputs: no gather then)
    ex.gather_tasks = (c.stage_on(ST_INPUT) && c.word_table) ? embed_gather_tasks(ea, Nq) : 0;
    ex.gather_rows = Nq; ex.emb = ea; ex.drop = c.drop;
    ex.wall_K = c.wall_K; ex.wall_off = c.wall_off; ex.wall_boff = c.wall_boff;
    if (c.novf) { ex.ovf = ovf; ex.novf = c.novf; }
    c.chk(launch_pack_weights(offs.data(), Ks.data(), boffs.data(), (int)offs.size(), c.P, c.PKF, c.stream, &ex,
                              c.PKT, c.want_bwd ? c.PKN : nullptr, needs.data()));
    if (opt->prezero_token) *opt->prezero_token = (ex.zero_ptr && c.ok()) ? (uint64_t)(uintptr_t)ex.zero_ptr : 0;      // the receipt
  }
  // ---------------- text encoder front: word + char embeddings (model.py:36-41)
  c.stage(ST_INPUT);
  if (c.live()) c.chk(launch_embed_fwd(ea, Nq, c.drop, c.stream));      // char CNN on the gathered embeddings
  // ---------------- the two input projections (model.py:42,48) as one launch; LN + pos (model.py:43,49,53,56)
  float* lin = c.act("lin");
  // feature-load phase (video_conv1d / query_conv1d + v / q layer norm + position embeddings).  Two paths:
  //   * K-split kernel (gemm.h launch_feature_ksplit) when the weight quarters fit LDS (vdim a multiple of 256, <= 1024): four
  //     K-quarter partial slabs, summed by the layer-norm launch behind it - the faster one (34 vs 45 us at the bench shape);
  //   * any other width: K / 128 weight steps per workgroup of rows with the layer norm in the row phase behind the last step
  //     (mproj.h): ONE launch, one slab.
  // The keep bytes of the clip-feature dropout (bit plane, one byte per 8 features) go to the weight-gradient job of video_conv1d.
  uint8_t* vkeep = reinterpret_cast<uint8_t*>(c.buf("video.keep", (size_t)Nv, (size_t)(c.cfg->vdim + 31) / 32));
  float* x = c.act("cb.x0");
  float* lin_mean = c.vec("lin.mean");
  float* lin_rstd = c.vec("lin.rstd");
  const int qks = ((catw + 3) / 4 + 63) & ~63;        // quarter size of query_conv1d's K (multiple of 64)
  const bool ksplit = c.ksplit;
  float* vpart = c.buf("lin.part", (size_t)4 * R, D);
  if (ksplit) {
    if (c.live()) {
      FkJob fj[2];
      fj[0] = FkJob{reinterpret_cast<const float*>(bt->video), c.cfg->vdim, Nv, c.cfg->vdim, c.cfg->vdim / 4,
                    reinterpret_cast<const float*>(c.PKF + pm.vconv.k * 4), vpart, (size_t)R * D, HUAL_SITE_VIDEO, 0,
                    bt->video_dtype == HUAL_DTYPE_BF16 ? 1 : 0, c.want_bwd ? vkeep : nullptr, (c.cfg->vdim + 7) / 8};
      fj[1] = FkJob{cat, catw, Nq, catw, qks, reinterpret_cast<const float*>(c.PKF + pm.qconv.k * 4),
                    vpart + (size_t)Nv * D, (size_t)R * D, -1, 0, 0, nullptr, 0};
      c.chk(launch_feature_ksplit(fj, 2, c.drop, c.stream));
      // v_layer_norm on the video rows, q_layer_norm on the query rows (+ position embeddings): one launch over the unified rows
      LnFwd a{};
      a.R = R; a.g1 = c.p(pm.vln.g); a.b1 = c.p(pm.vln.b); a.y1 = x; a.mean = lin_mean; a.rstd = lin_rstd;
      a.pos = c.p(pm.pos); a.row0 = 0;
      a.split = Nv; a.g1_hi = c.p(pm.qln.g); a.b1_hi = c.p(pm.qln.b);
      a.x = nullptr; a.part = vpart; a.nparts = 4; a.part_stride = (size_t)R * D; a.part_bias = c.p(pm.vconv.b);
      a.part_bias_hi = c.p(pm.qconv.b); a.x_out = lin;
      if (c.ok()) c.chk(launch_ln_fwd(a, rs, c.drop, c.stream));
    }
  } else
  if (c.live()) {
    const int V = c.cfg->vdim;
    MProjArgs pr[2];
    pr[0] = margs(Nv, Nq);
    pr[0].nsteps = 1;
    pr[0].s[0] = mstep(reinterpret_cast<const float*>(bt->video), V, std::min(V, 128), c.timg(pm.vconv.k), V, true, true);
    pr[0].s[0].a_bf16 = bt->video_dtype == HUAL_DTYPE_BF16 ? 1 : 0;
    pr[0].s[0].rep = cdiv(V, 128); pr[0].s[0].ktot = V;
    pr[0].s[0].drop_site = HUAL_SITE_VIDEO; pr[0].s[0].col0 = 0;
    pr[0].s[0].keep_out = c.want_bwd ? vkeep : nullptr; pr[0].s[0].ld_keep = (V + 7) / 8;
    pr[0].s[0].bias = c.p(pm.vconv.b);
    pr[0].ln_g = c.p(pm.vln.g); pr[0].ln_b = c.p(pm.vln.b); pr[0].pos = c.p(pm.pos); pr[0].row_in_clip0 = 0; pr[0].Tc = T;
    pr[0].x_out = lin; pr[0].y_out = x; pr[0].mean = lin_mean; pr[0].rstd = lin_rstd;
    pr[1] = margs(Nq, Nv);
    pr[1].nsteps = 1;
    pr[1].s[0] = mstep(cat, catw, std::min(catw, 128), c.timg(pm.qconv.k), catw, true, true);
    pr[1].s[0].rep = cdiv(catw, 128); pr[1].s[0].ktot = catw;
    pr[1].s[0].bias = c.p(pm.qconv.b);
    pr[1].ln_g = c.p(pm.qln.g); pr[1].ln_b = c.p(pm.qln.b); pr[1].pos = c.p(pm.pos); pr[1].row_in_clip0 = 0; pr[1].Tc = L;
    pr[1].x_out = lin + (size_t)Nv * D; pr[1].y_out = x + (size_t)Nv * D; pr[1].mean = lin_mean + Nv; pr[1].rstd = lin_rstd + Nv;
    c.chk(launch_mproj(pr, 2, c.drop, c.stream));
  }
  // ---------------- shared conv block (model.py:54-58)
  c.stage(ST_CONV);
  LnProjArgs da0_lp{};      // layer 0's layer norms + projections: filled below, launched at the end of the conv block's launch
  bool da0_lp_done = false;
  const bool da0_tail = c.cfg->attn_layer > 0;
  if (da0_tail) fill_da_ln_proj(c, da0_lp, 0, R, Nv);
  x = conv_block_fwd(c, "cb", x, pm.cb, rs, HUAL_SITE_CONV, nullptr, nullptr, da0_tail ? &da0_lp : nullptr, &da0_lp_done);
  // ---------------- dual attention layers (model.py:60-68)
  bool da_lp_done = da0_lp_done;      // this layer's ln_proj went with the previous launch (the conv block's / the previous layer's da_post)
  for (int li = 0; li < c.cfg->attn_layer; ++li) {
    const DualAttnP& d = pm.da[li];
    const std::string t = "da" + std::to_string(li);
    const int site = HUAL_SITE_DA + 8 * li;
    c.stage(ST_DA, li);
    float* ln1 = c.act(t + ".ln1");
    float* lnt = c.act(t + ".lnt");
    float* mean = c.vec(t + ".mean");
    float* rstd = c.vec(t + ".rstd");
    float* qkv = c.buf(t + ".qkv", R, 3 * D);
    float* ktvt = c.buf(t + ".ktvt", R, 2 * D);
    if (!da_lp_done) {      // layer norms + the five projections in one launch (dablock.h) - unless they rode in the launch in front
      LnProjArgs lp{};
      fill_da_ln_proj(c, lp, li, R, Nv);
      lp.x = x;
      if (c.live()) c.chk(launch_ln_proj(lp, c.drop, c.stream));
    }
    float* s_att = c.act(t + ".s_att");
    float* x_att = c.act(t + ".x_att");
    {
      AttnJob a[4];
      for (int k = 0; k < 4; ++k) {
        attn_job_init(a[k]);
        a[k].Q = qkv; a[k].ldq = 3 * D; a[k].B = B; a[k].qmask = rowmask; a[k].kmask = rowmask; a[k].ldo = D;
      }
      // video side: self (keys = video) / cross (keys = query)
      a[0].K = qkv + D; a[0].V = qkv + 2 * D; a[0].ldkv = 3 * D; a[0].Tq = T; a[0].Tk = T; a[0].qrow0 = 0; a[0].krow0 = 0;
      a[0].O = s_att; a[0].drop_site = site + 0;
      a[1].K = ktvt; a[1].V = ktvt + D; a[1].ldkv = 2 * D; a[1].Tq = T; a[1].Tk = L; a[1].qrow0 = 0; a[1].krow0 = Nv;
      a[1].O = x_att; a[1].drop_site = site + 1;
      // query side
      a[2].K = qkv + D; a[2].V = qkv + 2 * D; a[2].ldkv = 3 * D; a[2].Tq = L; a[2].Tk = L; a[2].qrow0 = Nv; a[2].krow0 = Nv;
      a[2].O = s_att; a[2].drop_site = site + 0;
      a[3].K = ktvt; a[3].V = ktvt + D; a[3].ldkv = 2 * D; a[3].Tq = L; a[3].Tk = T; a[3].qrow0 = Nv; a[3].krow0 = 0;
      a[3].O = x_att; a[3].drop_site = site + 1;
      for (int k = 0; k < 4; ++k) set_dmask(c, a[k], t + ".dm" + std::to_string(k));
      c.attn_fwd(a, 4);
    }
    // s / x projections -> cross gating -> guided dense -> bilinear gate.value -> dense_1 + residual: six dense jobs that
    // only ever touch their own rows - one chained launch (layers.py:93-111, modules.py:82-83)
    float* sv = c.act(t + ".s");
    float* xv = c.act(t + ".x");
    float* sg = c.act(t + ".sg");
    float* xg = c.act(t + ".xg");
    float* o = c.act(t + ".o");
    float* gd = c.act(t + ".g");
    float* gate = c.act(t + ".gate");
    float* val = c.act(t + ".val");
    float* mha = c.act(t + ".mha");
    float* res = c.act(t + ".res");
    float* l2 = c.act(t + ".l2");
    float* mean2 = c.vec(t + ".mean2");
    float* rstd2 = c.vec(t + ".rstd2");
    float* xo = c.act(t + ".out");
    {                      // the whole chain behind the attention kernels in one launch (dablock.h)
      DaPostArgs pa{};
      pa.s_att = s_att; pa.x_att = x_att; pa.ln1 = ln1; pa.x = x; pa.rowmask = rowmask;
      const size_t wo[11] = {d.s_dense.k, d.x_dense.k, d.s_gate.k, d.x_gate.k, d.guided.k, d.bl1_d1, d.bl1_d2, d.bl2_d1, d.bl2_d2,
                             d.dense1.k, d.dense2.k};
      const size_t bo[9] = {d.s_dense.b, d.x_dense.b, d.s_gate.b, d.x_gate.b, d.guided.b, d.bl1_b, d.bl2_b, d.dense1.b, d.dense2.b};
      for (int k = 0; k < 11; ++k) pa.w[k] = c.timg(wo[k]);      // register-resident weights (T images)
      for (int k = 0; k < 9; ++k) pa.b[k] = c.p(bo[k]);
      pa.ln2_g = c.p(d.ln2.g); pa.ln2_b = c.p(d.ln2.b);
      pa.sv = sv; pa.xv = xv; pa.sg = sg; pa.xg = xg; pa.o = o; pa.gd = gd; pa.gate = gate; pa.val = val; pa.mha = mha; pa.res = res;
      pa.l2 = l2; pa.out = xo; pa.mean2 = mean2; pa.rstd2 = rstd2;
      pa.site = site; pa.R = R; pa.Nv = Nv; pa.MT = da_post_rows(R, Nv); pa.drop_row0 = 0;
      pa.bits2 = c.bits(t + ".kb2", R); pa.bits3 = c.bits(t + ".kb3", R); pa.bits4 = c.bits(t + ".kb4", R);
      // the next layer's layer norms + projections ride at the end of this launch when the whole graph runs
      static const bool no_tail = getenv("HUAL_CB_NO_TAIL") != nullptr && atoi(getenv("HUAL_CB_NO_TAIL")) != 0;      // (A/B timings)
      LnProjArgs nlp{};
      const bool fuse = li + 1 < c.cfg->attn_layer && c.sel_stage < 0 && !no_tail;
      if (fuse) { fill_da_ln_proj(c, nlp, li + 1, R, Nv); nlp.x = xo; }
      if (c.live()) c.chk(launch_da_post(pa, c.drop, c.stream, fuse ? &nlp : nullptr));
      da_lp_done = fuse;
    }
    x = xo;
  }
  // ---------------- context-query attention in both directions (model.py:70-73)
  c.stage(ST_CQ);
  CqBufs cq{};
  cq.X = x;
  cq.D1W = c.act("cq.d1w"); cq.D2 = c.act("cq.d2"); cq.S0 = c.vec("cq.s0"); cq.S1 = c.vec("cq.s1");
  cq.C2Q = c.act("cq.c2q"); cq.Q2C = c.act("cq.q2c");
  const size_t mat = cq_mat_elems_host(T, L);
  cq.SR = c.buf("cq.sr", (size_t)2 * B, mat);
  cq.SC = c.buf("cq.sc", (size_t)2 * B, mat);
  cq.M2 = c.buf("cq.m2", (size_t)2 * B * cq_m2_rows_host(T, L), D);
  cq.GS = cq_fwd_global(B, T, L) ? c.buf("cq.gs", (size_t)2 * B, mat) : nullptr;
  CqParams cqp{};
  for (int i = 0; i < 2; ++i) { cqp.w0[i] = c.p(pm.cq[i].w0); cqp.w1[i] = c.p(pm.cq[i].w1); cqp.wm[i] = c.p(pm.cq[i].wm); }
  if (c.live()) c.chk(launch_cq_fwd(cq, cqp, c.rs, c.drop, c.stream));      // (tri_prep inside)
  float* cqf = c.act("cq.feats");      // q2v_feats (video rows) | v2q_feats (query rows)
  if (c.live()) {      // dense over [x, c2q, x * c2q, x * q2c] (layers.py:127-130): four weight steps per direction, one launch
    MProjArgs pr[2];
    for (int sd = 0; sd < 2; ++sd) {
      const size_t ro = sd == 0 ? 0 : (size_t)Nv * D;
      pr[sd] = margs(sd == 0 ? Nv : Nq, sd == 0 ? Nq : Nv);
      pr[sd].nsteps = 4;
      const float* a1[4] = {x + ro, cq.C2Q + ro, x + ro, x + ro};
      const float* a2[4] = {nullptr, nullptr, cq.C2Q + ro, cq.Q2C + ro};
      for (int p = 0; p < 4; ++p) {
        pr[sd].s[p] = mstep(a1[p], D, D, c.timg(pm.cq[sd].dense, p), D, p == 0, p == 3);
        pr[sd].s[p].A2 = a2[p]; pr[sd].s[p].lda2 = D;
      }
      mstep_out(pr[sd].s[3], cqf + ro, D);
    }
    c.chk(launch_mproj(pr, 2, c.drop, c.stream));
  }
  // ---------------- cq_concat (layers.py:145-154)
  c.stage(ST_FUSE);
  PoolArgs pa{};
  pa.F2 = cqf; pa.wp = c.p(pm.pool_w); pa.Wbot = c.p(pm.cqcat.k) + (size_t)D * D;
  pa.alpha = c.buf("pool.alpha", B, L); pa.pooled = c.buf("pool.pooled", B, D); pa.PW = c.buf("pool.pw", B, D);
  // the per-sample part of the alignment loss (model.py:76) also only reads cq.feats: same launch
  AlignPool ap{};
  ap.F2 = cqf; ap.F1 = cqf; ap.inner = (lab && !c.dry) ? lab->inner_labels : nullptr;
  ap.tpre = c.buf("align.tpre", B, D); ap.vpre = c.buf("align.vpre", B, D);
  ap.that = c.buf("align.tv", B, 2 * D); ap.vhat = ap.that + D; ap.ld = 2 * D;      // [that | vhat]: one [B,256] buffer
  if (c.live()) c.chk(launch_pool_align_fwd(pa, lab ? &ap : nullptr, c.rs, c.stream));
  float* fuse = c.actv("fuse");
  if (c.live()) {      // fuse = q2v_feats . W_top + b + (pooled . W_bot)[clip]   (layers.py:150-153)
    MProjArgs pr = margs(Nv);
    pr.nsteps = 1;
    pr.s[0] = mstep(cqf, D, D, c.timg(pm.cqcat.k), D, true, true);
    mstep_out(pr.s[0], fuse, D, c.p(pm.cqcat.b));
    pr.s[0].add = pa.PW; pr.s[0].ldadd = D; pr.s[0].add_div = T;
    c.chk(launch_mproj(&pr, 1, c.drop, c.stream));
  }
  // ---------------- matching head + label embeddings (model.py:82-97)
  float* outputs = c.actv("outputs");
  MatchArgs ma{};
  ma.fuse = fuse; ma.Wm = c.p(pm.match.k); ma.bm = c.p(pm.match.b); ma.E = c.p(pm.label_emb);
  ma.labels = (lab && !c.dry) ? lab->match_labels : nullptr;
  ma.probs = c.dry ? nullptr : out->match_scores; ma.outputs = outputs; ma.loss_acc = loss_acc;
  // private copy for the backward pass (the caller owns match_scores and may overwrite it)
  float* probs_keep = c.buf("match.probs", Nv, 4);
  ma.probs2 = (lab && !c.dry) ? probs_keep : nullptr;
  float* match_part = c.buf("match.part", (size_t)match_fwd_blocks(Nv), 2);
  ma.part = match_part;
  if (!c.cfg->no_gumbel) { ma.rng = opt->rng_state; ma.inv_tau = 1.0f / c.cfg->tau; }      // layers.py:163-166
  // ---------------- alignment loss, cross-sample part (layers.py:232-247), rows of the [B,B] similarity in the same launch (its
  // column part is formed by the backward's pool_align launch); exact data parallel evaluates it outside
  float* d_that = c.buf("d.align.that", B, D);
  float* d_vhat = c.buf("d.align.vhat", B, D);
  float* align_scratch = c.buf("align.scratch", (size_t)2 * B, B);
  // the rows' loss terms: summed in row order by the loss tail (the same arithmetic as hual_align_loss_rows' column launch; float
  // atomics from B workgroups made the reported loss differ by an ulp from run to run)
  float* align_rows = c.buf("align.rowloss", B, 1);
  if (c.live()) {
    AlignSim as{ap.that, ap.vhat, B, align_scratch, align_scratch + (size_t)B * B, align_rows, d_that, d_vhat, loss_acc, 1.0f, ap.ld, 0, B};
    c.chk(launch_match_fwd(ma, c.rs, (lab && !opt->align_external) ? &as : nullptr, c.stream));
  }
  // ---------------- conditioned predictor (modules.py:143-160)
  c.stage(ST_PRED);
  float* fin = outputs;
  float* feo[2];
  for (int ps = 0; ps < 2; ++ps) {
    const std::string t = "fe" + std::to_string(ps);
    const int site = HUAL_SITE_FE + 16 * ps;
    float* x0 = c.actv(t + ".x0");
    float* a1 = c.actv(t + ".a");
    float* mean = c.buf(t + ".ln1.mean", Nv, 1);
    float* rstd = c.buf(t + ".ln1.rstd", Nv, 1);
    float* qkv = c.buf(t + ".qkv", Nv, 3 * D);
    // layer_norm_1 + dropout + query / key / value (dablock.h ln_proj): at the end of the conv block's launch when the whole graph runs
    LnProjArgs lp{};
    lp.g1 = c.p(pm.fe_ln1.g); lp.b1 = c.p(pm.fe_ln1.b); lp.y1 = a1; lp.drop_site1 = site + 4; lp.pre_site = -1;
    lp.y1_bits = c.bits(t + ".kb4", Nv);
    lp.mean = mean; lp.rstd = rstd; lp.nproj = 3; lp.R = Nv; lp.MT = ln_proj_rows(Nv); lp.drop_row0 = 0;
    {
      const DenseP* pr[3] = {&pm.fe_q, &pm.fe_k, &pm.fe_v};
      for (int k = 0; k < 3; ++k) {
        lp.wimg[k] = c.timg(pr[k]->k); lp.bias[k] = c.p(pr[k]->b); lp.out[k] = qkv + k * D; lp.ldo[k] = 3 * D; lp.out_site[k] = -1;
      }
    }
    bool lp_done = false;
    float* f = conv_block_fwd(c, t, x0, pm.fe_cb, c.rsv, site, fin, c.p(pm.fe_pos), &lp, &lp_done);
    if (!lp_done && c.live()) c.chk(launch_ln_proj(lp, c.drop, c.stream));
    float* att = c.actv(t + ".att");
    {
      AttnJob a;
      attn_job_init(a);
      a.Q = qkv; a.ldq = 3 * D; a.K = qkv + D; a.V = qkv + 2 * D; a.ldkv = 3 * D; a.O = att; a.ldo = D;
      a.B = B; a.Tq = T; a.Tk = T; a.qrow0 = 0; a.krow0 = 0; a.qmask = rowmask; a.kmask = rowmask;
      a.drop_site = site + 5;
      set_dmask(c, a, t + ".dm");
      c.attn_fwd(&a, 1);
    }
    float* res = c.actv(t + ".res");
    float* l2 = c.actv(t + ".l2");
    float* mean2 = c.buf(t + ".ln2.mean", Nv, 1);
    float* rstd2 = c.buf(t + ".ln2.rstd", Nv, 1);
    float* fo = c.actv(t + ".out");
    {                      // residual + layer_norm_2 + dropout + dense + dropout + residual in one launch (modules.py:132-139)
      LnProjArgs lp{};
      lp.x = f; lp.xa = att; lp.pre_site = site + 6; lp.x_out = res;
      lp.pre_bits = c.bits(t + ".kb6", Nv); lp.y1_bits = c.bits(t + ".kb7", Nv); lp.out_bits[0] = c.bits(t + ".kb8", Nv);
      lp.g1 = c.p(pm.fe_ln2.g); lp.b1 = c.p(pm.fe_ln2.b); lp.y1 = l2; lp.drop_site1 = site + 7; lp.mean = mean2; lp.rstd = rstd2;
      lp.nproj = 1; lp.R = Nv; lp.MT = ln_proj_rows(Nv); lp.drop_row0 = 0;
      lp.wimg[0] = c.timg(pm.fe_dense.k); lp.bias[0] = c.p(pm.fe_dense.b); lp.out[0] = fo; lp.ldo[0] = D; lp.out_site[0] = site + 8; lp.add_x[0] = 1;
      if (c.live()) c.chk(launch_ln_proj(lp, c.drop, c.stream));
    }
    feo[ps] = fo;
    fin = fo;
  }
  float* sfn = c.actv("head.sfn");
  float* efn = c.actv("head.efn");
  float* hmean = c.buf("head.mean", (size_t)2 * Nv, 1);
  float* hrstd = c.buf("head.rstd", (size_t)2 * Nv, 1);
  float* hs = c.actv("head.hs");
  float* he = c.actv("head.he");
  {                        // start / end layer norm + hidden layer ([LN(feats), outputs] . W + b, relu) (modules.py:152-157): one launch
    LnProjArgs lp2[2];
    for (int h = 0; h < 2; ++h) {
      const DenseP& hp = h == 0 ? pm.shid : pm.ehid;
      const LnP& lnp = h == 0 ? pm.sln : pm.eln;
      LnProjArgs lp{};
      lp.x = feo[h]; lp.g1 = c.p(lnp.g); lp.b1 = c.p(lnp.b); lp.y1 = h == 0 ? sfn : efn; lp.drop_site1 = -1; lp.pre_site = -1;
      lp.mean = hmean + (size_t)h * Nv; lp.rstd = hrstd + (size_t)h * Nv; lp.x2 = outputs;
      lp.nproj = 2; lp.R = Nv; lp.MT = ln_proj_pair_rows(Nv); lp.drop_row0 = 0;
      lp.wimg[0] = c.timg(hp.k, 0); lp.src[0] = 0; lp.accum[0] = 1; lp.out_site[0] = -1;
      lp.wimg[1] = c.timg(hp.k, 1); lp.src[1] = 1; lp.bias[1] = c.p(hp.b);
      lp.act[1] = 1; lp.out[1] = h == 0 ? hs : he; lp.ldo[1] = D; lp.out_site[1] = -1;
      lp2[h] = lp;
    }
    if (c.live()) c.chk(launch_ln_proj_pair(lp2[0], lp2[1], c.drop, c.stream));
  }
  float* d_s = c.buf("d.s_logit", B, T);
  float* d_e = c.buf("d.e_logit", B, T);
  // ---------------- logits, localizing loss, span argmax, the gradients of the two hidden layers' outputs and the loss tail:
  // one launch (heads.h)
  float* dz_hs = c.actv("d.head.zs");
  float* dz_he = c.actv("d.head.ze");
  float* hpart_s = c.buf("head.part.s", (size_t)B * 2, D);
  float* hpart_e = c.buf("head.part.e", (size_t)B * 2, D);
  float* loc_part = c.buf("loc.part", B, 1);
  if (c.live()) {
    HeadsArgs ha{};
    ha.h[0] = hs; ha.h[1] = he; ha.w[0] = c.p(pm.sdense.k); ha.w[1] = c.p(pm.edense.k);
    ha.b[0] = c.p(pm.sdense.b); ha.b[1] = c.p(pm.edense.b);
    ha.logit[0] = out->start_logits; ha.logit[1] = out->end_logits; ha.vmask = rowmask;
    ha.y1 = lab ? lab->y1 : nullptr; ha.y2 = lab ? lab->y2 : nullptr;
    ha.start_index = out->start_index; ha.end_index = out->end_index;
    ha.ds = lab ? d_s : nullptr; ha.de = lab ? d_e : nullptr;
    if (lab) { ha.dZ[0] = dz_hs; ha.dZ[1] = dz_he; ha.part[0] = hpart_s; ha.part[1] = hpart_e; }
    ha.loc_part = loc_part; ha.inv_batch = 1.0f / (float)B;
    if (!lab && c.novf) { ha.ovf = ovf; ha.novf = c.novf; }
    c.chk(launch_heads(ha, B, T, c.stream));
    if (lab && c.ok() && !opt->deferred_loss_terms) {      // (deferred: match_bwd_kernel closes the loss, backward_graph)
      LossTailArgs lt{loss_acc, match_part, match_fwd_blocks(Nv), loc_part, B, c.cfg->match_lambda, opt->match_denom_override,
                      opt->match_denom_dev, out->loss_terms, ovf, c.novf, opt->align_external ? nullptr : align_rows, B};
      c.chk(launch_loss_tail(lt, c.stream));
    }
  }
  return c.rc;
}

// ======================================================================================================
// backward  (hand-derived; mirrors forward_graph bottom-up).  Parameter gradients of the dense layers are
// queued as DwJobs and flushed at the very end as a few large launches.
// ======================================================================================================
int backward_graph(Ctx& c, const hual_batch* bt, const hual_labels* lab, const hual_run_opts* opt) {
  const ParamMap& pm = c.pm;
  const int Nv = c.rs.Nv, Nq = c.rs.Nq, R = c.rs.R, B = c.B, T = c.T, L = c.L;
  const int D = HUAL_D;
  const int catw = c.cfg->word_dim + 100;
  float* rowmask = c.vec("rowmask");
  float* loss_acc = c.buf("loss_acc", 8, 1);
  c.dwjobs.clear();
  c.colsum.clear();
  c.part_seq = 0;
  c.stage(ST_ALWAYS);
  // (a kernel, not hipMemsetAsync: memset nodes of a captured graph were seen to pick up the fill pattern of later eager
  //  memsets on this ROCm - every 4th gradient came back as the caller's learning rate)
  {      // the bucket was zeroed by the forward's first launch iff the caller's token says so (hual_run_opts.prezero_token)
    const bool prezeroed = !c.dry && opt->prezero_token && *opt->prezero_token == (uint64_t)(uintptr_t)c.G && c.sel_stage < 0;
    if (!c.dry && opt->prezero_token) *opt->prezero_token = 0;
    if (c.live() && !prezeroed) c.chk(launch_zero(c.G, pm.total, c.stream));
  }
  // every dX product reads the N image that forward's pack launch left in the workspace (PKN)
  c.PKF = reinterpret_cast<char*>(c.buf("params.pkf", (c.pk_bytes + 3) / 4, 1));
  c.PKT = reinterpret_cast<char*>(c.buf("params.pkt", (c.pk_bytes + 3) / 4, 1));
  c.PKN = reinterpret_cast<char*>(c.buf("params.pkn", (c.pk_bytes + 3) / 4, 1));
  float* outputs = c.actv("outputs");
  // ---------------- heads
  c.stage(ST_PRED);
  float* hs = c.actv("head.hs");
  float* he = c.actv("head.he");
  float* sfn = c.actv("head.sfn");
  float* efn = c.actv("head.efn");
  float* hmean = c.buf("head.mean", (size_t)2 * Nv, 1);
  float* hrstd = c.buf("head.rstd", (size_t)2 * Nv, 1);
  float* d_s = c.buf("d.s_logit", B, T);
  float* d_e = c.buf("d.e_logit", B, T);
  float* dz_hs = c.actv("d.head.zs");
  float* dz_he = c.actv("d.head.ze");
  // dZ of the two hidden layers and the per-clip sums of d w / d b of start_dense / end_dense were left by the forward's heads
  // launch; a per-block call (gradients of the logits supplied by the caller) forms them here
  float* hpart_s = c.buf("head.part.s", (size_t)B * 2, D);
  float* hpart_e = c.buf("head.part.e", (size_t)B * 2, D);
  if (c.live() && c.sel_stage >= 0) {
    HeadsArgs ha{};
    ha.grad_only = 1;
    ha.h[0] = hs; ha.h[1] = he; ha.w[0] = c.p(pm.sdense.k); ha.w[1] = c.p(pm.edense.k); ha.b[0] = c.p(pm.sdense.b); ha.b[1] = c.p(pm.edense.b);
    ha.ds = d_s; ha.de = d_e; ha.dZ[0] = dz_hs; ha.dZ[1] = dz_he; ha.part[0] = hpart_s; ha.part[1] = hpart_e;
    c.chk(launch_heads(ha, B, T, c.stream));
  }
  for (int h = 0; h < 2; ++h) {
    ColsumJob cj{};
    cj.src = h == 0 ? hpart_s : hpart_e; cj.nblk = B; cj.nvec = 2; cj.last_ncols = 1;
    cj.dst[0] = c.g(h == 0 ? pm.sdense.k : pm.edense.k); cj.dst[1] = c.g(h == 0 ? pm.sdense.b : pm.edense.b);
    if (c.active) c.colsum.push_back(cj);
  }
  for (int h = 0; h < 2; ++h) {
    const DenseP& hp = h == 0 ? pm.shid : pm.ehid;
    DwJob j = mkdw(h == 0 ? sfn : efn, D, D, h == 0 ? dz_hs : dz_he, D, Nv, c.g(hp.k), c.g(hp.b));
    j.npieces = 2; j.A[1] = outputs; j.lda[1] = D; j.kw[1] = D; j.dW[1] = c.g(hp.k) + (size_t)D * D;
    c.push_dw(j);
  }
  float* d_sfn = c.actv("d.head.sfn");
  float* d_efn = c.actv("d.head.efn");
  float* d_out_heads = c.actv("d.outputs.heads");
  if (c.live()) {      // the two hidden layers backward: d sfn, d efn and the part of d outputs that came through them - four weight steps
    auto imgt = [&](size_t off, int blk) { return c.nimg(off, blk); };
    MProjArgs pr = margs(Nv);
    pr.nsteps = 4;
    pr.s[0] = mstep(dz_hs, D, D, imgt(pm.shid.k, 0), D, true, true);      mstep_out(pr.s[0], d_sfn, D);
    pr.s[1] = mstep_reuse(imgt(pm.shid.k, 1), D, true, false);
    pr.s[2] = mstep(dz_he, D, D, imgt(pm.ehid.k, 1), D, false, true);     mstep_out(pr.s[2], d_out_heads, D);
    pr.s[3] = mstep_reuse(imgt(pm.ehid.k, 0), D, true, true);             mstep_out(pr.s[3], d_efn, D);
    c.chk(launch_mproj(&pr, 1, c.drop, c.stream));
  }
  // ---------------- feature encoders, pass 1 then pass 0
  // (start / end layer norm backward: the prologue of the launch that takes dense^T + layer_norm_2 backward of the encoder pass
  //  below - LnProjBwdArgs::pre_*; its input gradient never leaves the registers)
  float* d_in = nullptr;
  float* fe_dx0[2] = {nullptr, nullptr};
  for (int ps = 1; ps >= 0; --ps) {
    const std::string t = "fe" + std::to_string(ps);
    const int site = HUAL_SITE_FE + 16 * ps;
    float* x0 = c.actv(t + ".x0");
    float* f = c.buf(t + ".x4", Nv, D);
    float* a1 = c.actv(t + ".a");
    float* mean = c.buf(t + ".ln1.mean", Nv, 1);
    float* rstd = c.buf(t + ".ln1.rstd", Nv, 1);
    float* qkv = c.buf(t + ".qkv", Nv, 3 * D);
    float* res = c.actv(t + ".res");
    float* l2 = c.actv(t + ".l2");
    float* mean2 = c.buf(t + ".ln2.mean", Nv, 1);
    float* rstd2 = c.buf(t + ".ln2.rstd", Nv, 1);
    // out = dropout(l2 . Wd + b, s8) + res
    float* dzd = c.actv("d." + t + ".zd");
    c.push_dw(mkdw(l2, D, D, dzd, D, Nv, c.g(pm.fe_dense.k), c.g(pm.fe_dense.b)));
    float* d_res = c.actv("d." + t + ".res");
    auto imgt = [&](size_t off) { return c.nimg(off); };      // N images: ln_proj_bwd_kernel / da_mid_bwd_kernel keep their weights in registers
    {                      // dense^T + layer_norm_2 backward in one launch (dablock.h); dZ of the dense layer came from upstream
      LnProjBwdArgs lb{};
      // prologue: gradient wrt fe<ps>.out = end / start layer norm backward (+ what pass 1 sent back to its input, for pass 0);
      // dropout'(.) of it (site 8) is the dense layer's dZ: operand of the product below, saved for the weight-gradient job
      const LnP& hl = ps == 1 ? pm.eln : pm.sln;
      lb.pre_x = c.actv(t + ".out"); lb.pre_mean = hmean + (ps == 1 ? Nv : 0); lb.pre_rstd = hrstd + (ps == 1 ? Nv : 0);
      lb.pre_g = c.p(hl.g); lb.pre_dy = ps == 1 ? d_efn : d_sfn; lb.pre_add = ps == 1 ? nullptr : d_in;
      lb.a_bits[0] = c.bits(t + ".kb8", Nv); lb.a_save[0] = dzd;
      lb.nsteps = 1; lb.A[0] = nullptr; lb.lda[0] = D; lb.wimg_t[0] = imgt(pm.fe_dense.k); lb.dst[0] = 0;
      lb.dy1_bits = c.bits(t + ".kb7", Nv); lb.x = res; lb.mean = mean2; lb.rstd = rstd2; lb.g1 = c.p(pm.fe_ln2.g);
      lb.dx = d_res; lb.dz = c.actv("d." + t + ".att"); lb.dz_bits = c.bits(t + ".kb6", Nv); lb.R = Nv; lb.drop_row0 = 0;
      c.ln_proj_bwd(lb, c.g(pm.fe_ln2.g), c.g(pm.fe_ln2.b), c.g(hl.g), c.g(hl.b));
    }
    // res = dropout(att, s6) + f: dropout'(d res) was written by the launch above
    float* d_att = c.actv("d." + t + ".att");
    float* d_qkv = c.buf("d." + t + ".qkv", Nv, 3 * D);
    {
      AttnJob a;
      attn_job_init(a);
      a.Q = qkv; a.ldq = 3 * D; a.K = qkv + D; a.V = qkv + 2 * D; a.ldkv = 3 * D;
      a.B = B; a.Tq = T; a.Tk = T; a.qrow0 = 0; a.krow0 = 0; a.qmask = rowmask; a.kmask = rowmask;
      a.drop_site = site + 5;
      a.O = c.actv(t + ".att"); a.ldo = D;
      a.dO = d_att; a.lddo = D; a.dQ = d_qkv; a.lddq = 3 * D; a.dK = d_qkv + D; a.dV = d_qkv + 2 * D; a.lddkv = 3 * D;
      set_dmask(c, a, t + ".dm");
      c.attn_bwd(&a, 1);
    }
    c.push_dw(mkdw(a1, D, D, d_qkv, 3 * D, Nv, c.g(pm.fe_q.k), c.g(pm.fe_q.b)));
    c.push_dw(mkdw(a1, D, D, d_qkv + D, 3 * D, Nv, c.g(pm.fe_k.k), c.g(pm.fe_k.b)));
    c.push_dw(mkdw(a1, D, D, d_qkv + 2 * D, 3 * D, Nv, c.g(pm.fe_v.k), c.g(pm.fe_v.b)));
    float* d_f = c.actv("d." + t + ".x4");
    {                      // query / key / value ^T + layer_norm_1 backward in one launch
      LnProjBwdArgs lb{};
      lb.nsteps = 3;
      const size_t wo[3] = {pm.fe_q.k, pm.fe_k.k, pm.fe_v.k};
      for (int k = 0; k < 3; ++k) { lb.A[k] = d_qkv + k * D; lb.lda[k] = 3 * D; lb.wimg_t[k] = imgt(wo[k]); lb.dst[k] = 0; }
      lb.dy1_bits = c.bits(t + ".kb4", Nv); lb.x = f; lb.mean = mean; lb.rstd = rstd; lb.g1 = c.p(pm.fe_ln1.g); lb.add1 = d_res;
      lb.dx = d_f; lb.R = Nv; lb.drop_row0 = 0;
      c.ln_proj_bwd(lb, c.g(pm.fe_ln1.g), c.g(pm.fe_ln1.b), nullptr, nullptr);
    }
    float* d_x0 = conv_block_bwd(c, t, x0, d_f, pm.fe_cb, c.rsv, site);
    fe_dx0[ps] = d_x0;            // the position-table gradients of the step are summed in ONE launch at the end
    d_in = d_x0;
  }
  // ---------------- gradient wrt `outputs`, matching head
  c.stage(ST_FUSE);
  // (gradient wrt `outputs` = what the encoders sent back + what the two hidden layers of the heads sent back: summed by
  //  match_bwd_kernel on the way in)
  float* fuse = c.actv("fuse");
  float* d_fuse = c.actv("d.fuse");
  float* match_part_b = c.buf("part." + std::to_string(c.part_seq++), (size_t)match_bwd_blocks(Nv) * 9, D);
  {
    ColsumJob cj{};
    cj.src = match_part_b; cj.nblk = match_bwd_blocks(Nv); cj.nvec = 9; cj.last_ncols = 4;
    for (int k = 0; k < 4; ++k) { cj.dst[k] = c.g(pm.label_emb) + k * D; cj.dst[4 + k] = c.g(pm.match.k) + k * D; }
    cj.dst[8] = c.g(pm.match.b);
    if (c.active) c.colsum.push_back(cj);
  }
  if (c.live()) {
    MatchArgs ma{};
    ma.fuse = fuse; ma.Wm = c.p(pm.match.k); ma.bm = c.p(pm.match.b); ma.E = c.p(pm.label_emb);
    ma.labels = lab->match_labels; ma.probs = nullptr; ma.outputs = outputs; ma.loss_acc = loss_acc;
    // probs were written to the caller's match_scores buffer; keep a private copy for backward
    ma.probs = c.buf("match.probs", Nv, 4);
    if (!c.cfg->no_gumbel) { ma.rng = opt->rng_state; ma.inv_tau = 1.0f / c.cfg->tau; }
    MatchBwd mb{};
    mb.dOut = d_in; mb.dOut2 = d_out_heads; mb.dFuse = d_fuse; mb.dWm = c.g(pm.match.k); mb.dbm = c.g(pm.match.b); mb.dE = c.g(pm.label_emb);
    mb.lambda = c.cfg->match_lambda;
    mb.dE_ortho = c.buf("ortho.dE", 4, D);
    mb.part = match_part_b;
    if (opt->deferred_loss_terms && c.sel_stage < 0) {      // the forward left the loss open (hual_run_opts.deferred_loss_terms)
      uint32_t* ovf = reinterpret_cast<uint32_t*>(c.buf("params.ovf", 1, 1));
      mb.do_tail = 1;
      mb.tail = LossTailArgs{loss_acc, c.buf("match.part", 1, 1), match_fwd_blocks(Nv), c.buf("loc.part", 1, 1), B, c.cfg->match_lambda,
                             opt->match_denom_override, opt->match_denom_dev, opt->deferred_loss_terms, ovf, c.novf,
                             opt->align_external ? nullptr : c.buf("align.rowloss", 1, 1), B};
    }
    c.chk(launch_match_bwd(ma, mb, c.rs, c.stream));
  } else {
    c.buf("ortho.dE", 4, D);
    c.buf("match.probs", Nv, 4);
  }
  // ---------------- cq_concat
  float* cqf = c.act("cq.feats");
  float* d_cqf = c.act("d.cq.feats");
  c.push_dw(mkdw(cqf, D, D, d_fuse, D, Nv, c.g(pm.cqcat.k), c.g(pm.cqcat.b)));
  if (c.live()) {
    MProjArgs pr = margs(Nv);
    pr.nsteps = 1;
    pr.s[0] = mstep(d_fuse, D, D, c.nimg(pm.cqcat.k), D, true, true);
    mstep_out(pr.s[0], d_cqf, D);
    c.chk(launch_mproj(&pr, 1, c.drop, c.stream));
  }
  AlignPool ap{};
  ap.F2 = cqf; ap.F1 = cqf; ap.inner = (c.dry || !lab) ? nullptr : lab->inner_labels;
  ap.tpre = c.buf("align.tpre", B, D); ap.vpre = c.buf("align.vpre", B, D);
  ap.that = c.buf("align.tv", B, 2 * D); ap.vhat = ap.that + D; ap.ld = 2 * D;      // [that | vhat]: one [B,256] buffer
  float* d_that = c.buf("d.align.that", B, D);
  float* d_vhat = c.buf("d.align.vhat", B, D);
  PoolArgs pa{};
  pa.F2 = cqf; pa.wp = c.p(pm.pool_w); pa.Wbot = c.p(pm.cqcat.k) + (size_t)D * D;
  pa.alpha = c.buf("pool.alpha", B, L); pa.pooled = c.buf("pool.pooled", B, D); pa.PW = c.buf("pool.pw", B, D);
  float* d_pw = c.buf("d.pool.pw", B, D);
  if (c.live()) {
    AlignPoolBwd ab{d_that, d_vhat, d_cqf, d_cqf, nullptr, nullptr, 0};     // writes the query rows, accumulates the video rows
    if (!opt->align_external && c.sel_stage < 0) {      // column part of d vhat: from the similarity scratch of the forward
      float* asc = c.buf("align.scratch", (size_t)2 * B, B);
      ab.col_dq = asc; ab.col_da = asc + (size_t)B * B; ab.col_Bg = B;
    }
    PoolBwd pb{d_fuse, d_pw, d_cqf, c.g(pm.pool_w)};    // accumulates into the query rows
    c.chk(launch_pool_align_bwd(pa, pb, ap, ab, c.rs, c.stream));
  }
  c.push_dw(mkdw(pa.pooled, D, D, d_pw, D, B, c.g(pm.cqcat.k) + (size_t)D * D, nullptr));
  // ---------------- the two cq_attention dense layers
  c.stage(ST_CQ);
  float* xf = c.cfg->attn_layer > 0 ? c.act("da" + std::to_string(c.cfg->attn_layer - 1) + ".out") : nullptr;
  CqBufs cq{};
  cq.X = xf;
  cq.D1W = c.act("cq.d1w"); cq.D2 = c.act("cq.d2"); cq.S0 = c.vec("cq.s0"); cq.S1 = c.vec("cq.s1");
  cq.C2Q = c.act("cq.c2q"); cq.Q2C = c.act("cq.q2c");
  const size_t mat = cq_mat_elems_host(T, L);
  cq.SR = c.buf("cq.sr", (size_t)2 * B, mat);
  cq.SC = c.buf("cq.sc", (size_t)2 * B, mat);
  cq.M2 = c.buf("cq.m2", (size_t)2 * B * cq_m2_rows_host(T, L), D);
  cq.GS = cq_fwd_global(B, T, L) ? c.buf("cq.gs", (size_t)2 * B, mat) : nullptr;
  for (int s = 0; s < 2; ++s) {
    const size_t ro = s == 0 ? 0 : (size_t)Nv * D;
    DwJob j = mkdw(xf + ro, D, D, d_cqf + ro, D, s == 0 ? Nv : Nq, c.g(pm.cq[s].dense), nullptr);
    j.npieces = 4;
    const float* a1[4] = {xf + ro, cq.C2Q + ro, xf + ro, xf + ro};
    const float* a2[4] = {nullptr, nullptr, cq.C2Q + ro, cq.Q2C + ro};
    for (int p = 0; p < 4; ++p) {
      j.A[p] = a1[p]; j.A2[p] = a2[p]; j.lda[p] = D; j.lda2[p] = D; j.kw[p] = D;
      j.dW[p] = c.g(pm.cq[s].dense) + (size_t)p * D * D;
    }
    c.push_dw(j);
  }
  CqBwdBufs cg{};
  cg.dCat = nullptr; cg.ldcat = 4 * D;
  cg.dC2Q = c.act("d.cq.c2q"); cg.dQ2C = c.act("d.cq.q2c"); cg.dX = c.act("d.cq.x");
  if (c.live()) {      // d [x, c2q, x * c2q, x * q2c] = d feats . W^T: four column blocks of one operand per direction, one launch; the
    MProjArgs pr[2];   // four tiles of a row leave the kernel already split into d c2q, d q2c and the direct part of d x (mproj.h quad_*)
    for (int sd = 0; sd < 2; ++sd) {
      const size_t ro = sd == 0 ? 0 : (size_t)Nv;
      pr[sd] = margs(sd == 0 ? Nv : Nq, sd == 0 ? Nq : Nv);
      pr[sd].nsteps = 4;
      for (int p = 0; p < 4; ++p) {
        const float* img = c.nimg(pm.cq[sd].dense, p);
        pr[sd].s[p] = p == 0 ? mstep(d_cqf + ro * D, D, D, img, D, true, true) : mstep_reuse(img, D, true, true);
      }
      pr[sd].quad_x = xf + ro * D; pr[sd].quad_c2q = cq.C2Q + ro * D; pr[sd].quad_q2c = cq.Q2C + ro * D;
      pr[sd].quad_dc2q = cg.dC2Q + ro * D; pr[sd].quad_dq2c = cg.dQ2C + ro * D; pr[sd].quad_dx = cg.dX + ro * D;
    }
    c.chk(launch_mproj(pr, 2, c.drop, c.stream));
  }
  cg.dD1W = c.act("d.cq.d1w"); cg.dD2 = c.act("d.cq.d2"); cg.dS0 = c.vec("d.cq.s0"); cg.dS1 = c.vec("d.cq.s1");
  cg.dM2 = c.buf("d.cq.m2", (size_t)2 * B * cq_m2_rows_host(T, L), D);
  cg.GD = cq_bwd_global(B, T, L) ? c.buf("d.cq.gd", (size_t)4 * B, mat) : nullptr;
  float* dXa = c.act("d.cq.xa");
  float* dXb = c.act("d.cq.xb");
  CqParams cqp{};
  CqGrads cqg{};
  for (int i = 0; i < 2; ++i) {
    cqp.w0[i] = c.p(pm.cq[i].w0); cqp.w1[i] = c.p(pm.cq[i].w1); cqp.wm[i] = c.p(pm.cq[i].wm);
    cqg.w0[i] = c.g(pm.cq[i].w0); cqg.w1[i] = c.g(pm.cq[i].w1); cqg.wm[i] = c.g(pm.cq[i].wm);
  }
  const int tri_nv = tri_bwd_blocks_v(c.rs), tri_nq = tri_bwd_blocks_q(c.rs);
  float* tri_part = c.buf("part." + std::to_string(c.part_seq++), (size_t)(tri_nv + tri_nq) * 3, D);
  for (int side = 0; side < 2; ++side) {      // video-side workgroups: direction 0 plays x1, direction 1 x2; query side: the reverse
    ColsumJob cj{};
    cj.src = tri_part + (side == 0 ? 0 : (size_t)tri_nv * 3 * D); cj.nblk = side == 0 ? tri_nv : tri_nq; cj.nvec = 3;
    cj.dst[0] = cqg.wm[side]; cj.dst[1] = cqg.w0[side]; cj.dst[2] = cqg.w1[1 - side];
    if (c.active) c.colsum.push_back(cj);
  }
  if (c.live()) {
    c.chk(launch_cq_bwd_impl(cq, cg, c.rs, dXa, dXb, c.stream));
    if (c.ok()) c.chk(launch_tri_bwd_impl(cq, cg, cqp, tri_part, c.rs, c.drop, dXa, dXb, c.stream));
  }
  float* dx = cg.dX;
  // ---------------- dual attention layers, last to first
  for (int li = c.cfg->attn_layer - 1; li >= 0; --li) {
    const DualAttnP& d = pm.da[li];
    const std::string t = "da" + std::to_string(li);
    const int site = HUAL_SITE_DA + 8 * li;
    c.stage(ST_DA, li);
    float* xin = li == 0 ? c.act("cb.x4") : c.act("da" + std::to_string(li - 1) + ".out");
    float* ln1 = c.act(t + ".ln1");
    float* lnt = c.act(t + ".lnt");
    float* mean = c.vec(t + ".mean");
    float* rstd = c.vec(t + ".rstd");
    float* qkv = c.buf(t + ".qkv", R, 3 * D);
    float* ktvt = c.buf(t + ".ktvt", R, 2 * D);
    float* s_att = c.act(t + ".s_att");
    float* x_att = c.act(t + ".x_att");
    float* sv = c.act(t + ".s");
    float* xv = c.act(t + ".x");
    float* sg = c.act(t + ".sg");
    float* xg = c.act(t + ".xg");
    float* o = c.act(t + ".o");
    float* gd = c.act(t + ".g");
    float* gate = c.act(t + ".gate");
    float* val = c.act(t + ".val");
    float* mha = c.act(t + ".mha");
    float* res = c.act(t + ".res");
    float* l2 = c.act(t + ".l2");
    float* mean2 = c.vec(t + ".mean2");
    float* rstd2 = c.vec(t + ".rstd2");
    const std::string dt = "d." + t;
    // out = dropout(l2 . Wd2 + b, s4) + res
    float* dz2 = c.act(dt + ".z2");
    c.push_dw(mkdw(l2, D, D, dz2, D, R, c.g(d.dense2.k), c.g(d.dense2.b)));
    float* d_res = c.act(dt + ".res");
    float* dz1 = c.act(dt + ".z1");
    float* d_sc = c.act(dt + ".sc");
    float* d_val = c.act(dt + ".val");
    float* d_ln1a = c.act(dt + ".ln1a");
    float* d_g = c.act(dt + ".g");
    float* dz_sg = c.act(dt + ".zsg");
    float* dz_xg = c.act(dt + ".zxg");
    float* d_sv = c.act(dt + ".s");
    float* d_xv = c.act(dt + ".x");
    float* d_satt = c.act(dt + ".s_att");
    float* d_xatt = c.act(dt + ".x_att");
    auto imgt = [&](size_t off) { return c.nimg(off); };      // N images: ln_proj_bwd_kernel / da_mid_bwd_kernel keep their weights in registers
    const bool dz2_ready = li < c.cfg->attn_layer - 1 && c.sel_stage < 0;     // the layer above left dropout'(dx) in dz2
    // weight-gradient jobs of this half of the block (operands are written by whichever path runs below)
    c.push_dw(mkdw(mha, D, D, dz1, D, R, c.g(d.dense1.k), c.g(d.dense1.b)));
    for (int k = 0; k < 2; ++k) {
      DwJob j = mkdw(ln1, D, D, k == 0 ? d_sc : d_val, D, R, c.g(k == 0 ? d.bl1_d1 : d.bl2_d1), c.g(k == 0 ? d.bl1_b : d.bl2_b));
      j.npieces = 2; j.A[1] = gd; j.lda[1] = D; j.kw[1] = D; j.dW[1] = c.g(k == 0 ? d.bl1_d2 : d.bl2_d2);
      c.push_dw(j);
    }
    c.push_dw(mkdw(o, D, D, d_g, D, R, c.g(d.guided.k), c.g(d.guided.b)));
    c.push_dw(mkdw(sv, D, D, dz_sg, D, R, c.g(d.s_gate.k), c.g(d.s_gate.b)));
    c.push_dw(mkdw(xv, D, D, dz_xg, D, R, c.g(d.x_gate.k), c.g(d.x_gate.b)));
    c.push_dw(mkdw(s_att, D, D, d_sv, D, R, c.g(d.s_dense.k), c.g(d.s_dense.b)));
    c.push_dw(mkdw(x_att, D, D, d_xv, D, R, c.g(d.x_dense.k), c.g(d.x_dense.b)));
    {
      // (1) dense_2^T + layer_norm_2 backward -> d res, dZ1     (2) the gated middle, ten weight steps     (dablock.h)
      LnProjBwdArgs lb{};
      lb.nsteps = 1; lb.lda[0] = D; lb.wimg_t[0] = imgt(d.dense2.k); lb.dst[0] = 0;
      if (dz2_ready) { lb.A[0] = dz2; }
      else { lb.A[0] = dx; lb.a_bits[0] = c.bits(t + ".kb4", R); lb.a_save[0] = dz2; }
      lb.dy1_bits = c.bits(t + ".kb3", R); lb.x = res; lb.mean = mean2; lb.rstd = rstd2; lb.g1 = c.p(d.ln2.g); lb.add1 = dx;
      lb.dx = d_res; lb.dz = dz1; lb.dz_bits = c.bits(t + ".kb2", R); lb.R = R; lb.Nv = Nv; lb.drop_row0 = 0;
      c.ln_proj_bwd(lb, c.g(d.ln2.g), c.g(d.ln2.b), nullptr, nullptr);
      DaMidBwdArgs mb{};
      mb.dz1 = dz1; mb.gate = gate; mb.val = val; mb.sg = sg; mb.xg = xg; mb.sv = sv; mb.xv = xv;
      const size_t wo[10] = {d.dense1.k, d.bl1_d1, d.bl2_d1, d.bl1_d2, d.bl2_d2, d.guided.k, d.s_gate.k, d.x_gate.k, d.s_dense.k, d.x_dense.k};
      for (int k = 0; k < 10; ++k) mb.w[k] = c.nimg(wo[k]);      // register-resident weights (N images)
      mb.d_sc = d_sc; mb.d_val = d_val; mb.d_ln1a = d_ln1a; mb.d_g = d_g; mb.dz_sg = dz_sg; mb.dz_xg = dz_xg; mb.d_sv = d_sv; mb.d_xv = d_xv;
      mb.d_satt = d_satt; mb.d_xatt = d_xatt; mb.R = R; mb.Nv = Nv; mb.MT = da_post_rows(R, Nv);
      if (c.live()) c.chk(launch_da_mid_bwd(mb, c.stream));
    }
    // the four attentions
    float* dq_self = c.act(dt + ".q_self");
    float* dq_cross = c.act(dt + ".q_cross");
    float* d_qkv = c.buf(dt + ".qkv", R, 3 * D);     // only the Kf / Vf column blocks are used
    float* d_ktvt = c.buf(dt + ".ktvt", R, 2 * D);
    {
      AttnJob a[4];
      for (int k = 0; k < 4; ++k) {
        attn_job_init(a[k]);
        a[k].Q = qkv; a[k].ldq = 3 * D; a[k].B = B; a[k].qmask = rowmask; a[k].kmask = rowmask; a[k].lddo = D; a[k].lddq = D;
        a[k].O = (k & 1) ? x_att : s_att; a[k].ldo = D;
      }
      a[0].K = qkv + D; a[0].V = qkv + 2 * D; a[0].ldkv = 3 * D; a[0].Tq = T; a[0].Tk = T; a[0].qrow0 = 0; a[0].krow0 = 0;
      a[0].dO = d_satt; a[0].dQ = dq_self; a[0].dK = d_qkv + D; a[0].dV = d_qkv + 2 * D; a[0].lddkv = 3 * D; a[0].drop_site = site + 0;
      a[1].K = ktvt; a[1].V = ktvt + D; a[1].ldkv = 2 * D; a[1].Tq = T; a[1].Tk = L; a[1].qrow0 = 0; a[1].krow0 = Nv;
      a[1].dO = d_xatt; a[1].dQ = dq_cross; a[1].dK = d_ktvt; a[1].dV = d_ktvt + D; a[1].lddkv = 2 * D; a[1].drop_site = site + 1;
      a[2].K = qkv + D; a[2].V = qkv + 2 * D; a[2].ldkv = 3 * D; a[2].Tq = L; a[2].Tk = L; a[2].qrow0 = Nv; a[2].krow0 = Nv;
      a[2].dO = d_satt; a[2].dQ = dq_self; a[2].dK = d_qkv + D; a[2].dV = d_qkv + 2 * D; a[2].lddkv = 3 * D; a[2].drop_site = site + 0;
      a[3].K = ktvt; a[3].V = ktvt + D; a[3].ldkv = 2 * D; a[3].Tq = L; a[3].Tk = T; a[3].qrow0 = Nv; a[3].krow0 = 0;
      a[3].dO = d_xatt; a[3].dQ = dq_cross; a[3].dK = d_ktvt; a[3].dV = d_ktvt + D; a[3].lddkv = 2 * D; a[3].drop_site = site + 1;
      for (int k = 0; k < 4; ++k) set_dmask(c, a[k], t + ".dm" + std::to_string(k));
      c.attn_bwd(a, 4);
    }
    c.push_dw(mkdw(ln1, D, D, dq_self, D, R, c.g(d.query.k), c.g(d.query.b)));
    c.push_dw(mkdw(ln1, D, D, dq_cross, D, R, c.g(d.query.k), c.g(d.query.b)));
    c.push_dw(mkdw(ln1, D, D, d_qkv + D, 3 * D, R, c.g(d.f_key.k), c.g(d.f_key.b)));
    c.push_dw(mkdw(ln1, D, D, d_qkv + 2 * D, 3 * D, R, c.g(d.f_value.k), c.g(d.f_value.b)));
    c.push_dw(mkdw(lnt, D, D, d_ktvt, 2 * D, R, c.g(d.t_key.k), c.g(d.t_key.b)));
    c.push_dw(mkdw(lnt, D, D, d_ktvt + D, 2 * D, R, c.g(d.t_value.k), c.g(d.t_value.b)));
    float* d_xin = c.act(dt + ".in");
    // operand of the next dX product down the stack: previous layer's dense_2, or (unfused conv block) its layer 3
    float* nz = nullptr; const uint8_t* nz_bits = nullptr;
    if (li > 0) { nz = c.act("d.da" + std::to_string(li - 1) + ".z2"); nz_bits = c.bits("da" + std::to_string(li - 1) + ".kb4", R); }
    {                      // the six projection^T products + layer_norm_1 / layer_norm_t backward in one launch (dablock.h)
      LnProjBwdArgs lb{};
      lb.nsteps = 6;
      const float* As[6] = {dq_self, dq_cross, d_qkv + D, d_qkv + 2 * D, d_ktvt, d_ktvt + D};
      const int lds_[6] = {D, D, 3 * D, 3 * D, 2 * D, 2 * D};
      const size_t wo[6] = {d.query.k, d.query.k, d.f_key.k, d.f_value.k, d.t_key.k, d.t_value.k};
      for (int k = 0; k < 6; ++k) { lb.A[k] = As[k]; lb.lda[k] = lds_[k]; lb.wimg_t[k] = imgt(wo[k]); lb.dst[k] = k < 4 ? 0 : 1; }
      lb.add_dy1 = d_ln1a; lb.x = xin; lb.mean = mean; lb.rstd = rstd; lb.g1 = c.p(d.ln1.g); lb.g2 = c.p(d.lnt.g);
      lb.add1 = d_res; lb.dx = d_xin; lb.dz = nz; lb.dz_bits = nz_bits; lb.R = R; lb.Nv = Nv; lb.drop_row0 = 0;
      c.ln_proj_bwd(lb, c.g(d.ln1.g), c.g(d.ln1.b), c.g(d.lnt.g), c.g(d.lnt.b));
    }
    dx = d_xin;
  }
  // ---------------- shared conv block, position table, input layer norms, projections
  float* x0 = c.act("cb.x0");
  c.stage(ST_CONV);
  float* d_x0 = conv_block_bwd(c, "cb", x0, dx, pm.cb, c.rs, HUAL_SITE_CONV);
  c.stage(ST_ALWAYS);
  // position-table gradients of the stages that ran: in the whole model they ride in the launch of the input layer norms' backward
  // (same gradient tensor, independent work); a per-block call launches them on their own
  PosBwdJob pj[2];
  int npj = 0;
  if (!c.dry && c.stage_on(ST_INPUT)) pj[npj++] = PosBwdJob{{d_x0, nullptr}, c.g(pm.pos), 1, 1};     // the table is added in the input stage
  if (!c.dry && c.stage_on(ST_PRED)) pj[npj++] = PosBwdJob{{fe_dx0[1], fe_dx0[0]}, c.g(pm.fe_pos), 1, 0};
  const bool pos_rides = c.sel_stage < 0;
  if (c.live() && npj && !pos_rides) c.chk(launch_pos_bwd(pj, npj, c.rs, c.stream));
  c.stage(ST_INPUT);
  float* lin = c.act("lin");
  float* lin_mean = c.vec("lin.mean");
  float* lin_rstd = c.vec("lin.rstd");
  float* d_lin = c.act("d.lin");
  {
    // v_layer_norm (video rows) and q_layer_norm (query rows) backward: one launch over the unified rows
    LnBwd a{};
    a.x = lin; a.mean = lin_mean; a.rstd = lin_rstd; a.R = R; a.dy1 = d_x0; a.g1 = c.p(pm.vln.g);
    a.dx = d_lin; a.split = Nv; a.g1_hi = c.p(pm.qln.g);
    c.ln_bwd_split(a, c.g(pm.vln.g), c.g(pm.vln.b), c.g(pm.qln.g), c.g(pm.qln.b), pj, pos_rides ? npj : 0);
  }
  float* cat = c.buf("cat", Nq, catw);
  {
    DwJob j = mkdw(c.dry ? nullptr : reinterpret_cast<const float*>(bt->video), c.cfg->vdim, c.cfg->vdim, d_lin, D, Nv, c.g(pm.vconv.k), c.g(pm.vconv.b));
    j.a_drop_site = HUAL_SITE_VIDEO; j.a_drop_row0 = 0;
    j.a_bf16 = (!c.dry && bt->video_dtype == HUAL_DTYPE_BF16) ? 1 : 0;
    // written by the K-split feature-load kernel
    j.a_keep = reinterpret_cast<const uint8_t*>(c.buf("video.keep", 0, 0));      // written by the feature-load launch
    j.ld_keep = (c.cfg->vdim + 7) / 8;
    c.push_dw(j);
    c.push_dw(mkdw(cat, catw, catw, d_lin + (size_t)Nv * D, D, Nq, c.g(pm.qconv.k), c.g(pm.qconv.b)));
  }
  float* d_cat = c.buf("d.cat", Nq, catw);
  if (c.live()) {      // d cat = d lin[q rows] . W_q^T: ceil(catw / 128) column blocks of one operand
    HUAL_REQUIRE(cdiv(catw, 128) <= MP_MAX, "internal: query_conv1d image");
    MProjArgs pr = margs(Nq);
    pr.nsteps = cdiv(catw, 128);
    for (int p = 0; p < pr.nsteps; ++p) {
      const float* img = c.nimg(pm.qconv.k, p);
      pr.s[p] = p == 0 ? mstep(d_lin + (size_t)Nv * D, D, D, img, D, true, true) : mstep_reuse(img, D, true, true);
      mstep_out(pr.s[p], d_cat + (size_t)p * D, catw, nullptr, std::min(D, catw - p * D));
    }
    c.chk(launch_mproj(&pr, 1, c.drop, c.stream));
  }
  // ---------------- text encoder front end (embed.hip); its filter gradients ride in the weight-gradient launch
  int32_t* char_arg = reinterpret_cast<int32_t*>(c.buf("char_arg", Nq, 100));
  float* embed_scratch = c.buf("embed.scratch", embed_layout(Nq, c.C, c.cfg->char_dim).total, 1);
  EmbedArgs ea{};
  EmbedGrads eg{};
  set_embed_scratch(ea, embed_scratch, Nq, c.C, c.cfg->char_dim);
  ea.word_dim = c.cfg->word_dim; ea.char_dim = c.cfg->char_dim; ea.C = c.C; ea.num_chars = c.cfg->num_chars;
  DwJob embed_dw;
  embed_dw_job(ea, Nq, &embed_dw);
  if (c.live()) {
    fill_embed_args(c, ea, bt, cat, catw, char_arg, embed_scratch);
    eg.dcat = d_cat; eg.lddcat = catw; eg.dunk = c.g(pm.unk); eg.dchar_table = c.g(pm.char_table);
    for (int i = 0; i < 4; ++i) { eg.dfilt[i] = c.g(pm.filt[i]); eg.dfbias[i] = c.g(pm.fbias[i]); }
    // (its last step - folding the window gradients into the char table's - rides in the launch that folds the partial sums, below)
    c.chk(launch_embed_bwd(ea, eg, Nq, c.drop, c.stream, &embed_dw, false));
  }
  c.push_dw(embed_dw);
  const bool input_ran = c.active;
  c.stage(ST_ALWAYS);
  flush_dw(c);            // every dense / conv weight gradient of the step: one launch
  // layer-norm / depthwise-conv parameter gradients: one reduction of the per-block partial sums, with the unpack of the char-CNN
  // filter gradients (behind the weight-gradient launch) as further workgroups of the same launch
  EmbedUnpack eu{};
  const bool unpack = !c.dry && input_ran && c.active;
  if (unpack) {
    eu.a = ea; eu.g = eg; eu.CP = embed_unpack_cpad(ea); eu.ntasks = embed_unpack_tasks(ea);
    eu.finish_blocks = embed_finish_blocks(ea, Nq); eu.finish_lds = embed_finish_lds(ea); eu.nrows = Nq; eu.drop = c.drop;
  }
  c.flush_colsum(unpack ? &eu : nullptr);
  (void)opt;
  return c.rc;
}

int plan(Ctx& c, const hual_cfg* cfg, int B, int T, int L, int C) {
  int rc = setup_ctx(c, cfg, B, T, L, C);
  if (rc) return rc;
  c.dry = true;
  c.base = nullptr;
  hual_run_opts o{};
  rc = forward_graph(c, nullptr, nullptr, nullptr, &o);
  if (rc) return rc;
  return backward_graph(c, nullptr, nullptr, &o);
}

}  // namespace

// ======================================================================================================
// C ABI
// ======================================================================================================
extern "C" {

int hual_seqpan_validate(const hual_cfg* cfg) {
  HUAL_REQUIRE(cfg != nullptr, "null cfg");
  return validate_cfg(*cfg);
}

// room for the weight-gradient jobs of any configuration (attn_layer 2: 62 jobs; every further dual-attention layer adds 16)
#define HUAL_DW_TABLE_JOBS 256
uint64_t hual_seqpan_dw_table_bytes(void) { return 4 * dw_table_words(HUAL_DW_TABLE_JOBS); }

int hual_seqpan_param_count(const hual_cfg* cfg, uint64_t* padded_floats, uint64_t* count) {
  HUAL_REQUIRE(cfg != nullptr, "null cfg");
  ParamMap pm;
  int rc = build_param_map(*cfg, pm);
  if (rc) return rc;
  if (padded_floats) *padded_floats = pm.total;
  if (count) *count = pm.count;
  return 0;
}

int hual_seqpan_param_table(const hual_cfg* cfg, hual_param_entry* out, int max_entries) {
  HUAL_REQUIRE(cfg != nullptr, "null cfg");
  ParamMap pm;
  int rc = build_param_map(*cfg, pm);
  if (rc) return rc;
  const int n = (int)pm.entries.size();
  for (int i = 0; i < n && i < max_entries && out; ++i) {
    const ParamEntry& e = pm.entries[i];
    memset(&out[i], 0, sizeof(out[i]));
    strncpy(out[i].name, e.name.c_str(), sizeof(out[i].name) - 1);
    out[i].offset = e.off; out[i].size = e.size; out[i].ndim = e.ndim; out[i].decay = e.decay;
    for (int k = 0; k < 4; ++k) out[i].shape[k] = e.shape[k];
  }
  return n;
}

int hual_seqpan_query_workspace(const hual_cfg* cfg, int B, int T, int L, int C, uint64_t* bytes) {
  HUAL_REQUIRE(bytes != nullptr, "null bytes");
  Ctx c;
  int rc = plan(c, cfg, B, T, L, C);
  if (rc) return rc;
  *bytes = c.used + 4096;
  return 0;
}

int hual_seqpan_ws_table(const hual_cfg* cfg, int B, int T, int L, int C, hual_ws_entry* out, int max_entries) {
  Ctx c;
  int rc = plan(c, cfg, B, T, L, C);
  if (rc) return rc;
  const int n = (int)c.entries.size();
  for (int i = 0; i < n && i < max_entries && out; ++i) {
    memset(&out[i], 0, sizeof(out[i]));
    strncpy(out[i].name, c.entries[i].name.c_str(), sizeof(out[i].name) - 1);
    out[i].offset = c.entries[i].off; out[i].rows = c.entries[i].rows; out[i].cols = c.entries[i].cols;
  }
  return n;
}

static int check_common(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts,
                        void* workspace) {
  HUAL_REQUIRE(cfg && params && batch && opts && workspace, "null argument");
  HUAL_REQUIRE(batch->video && batch->video_seq_len && batch->word_ids && batch->char_ids, "null batch tensor");
  HUAL_REQUIRE(batch->video_dtype == HUAL_DTYPE_F32 || batch->video_dtype == HUAL_DTYPE_BF16, "video_dtype: HUAL_DTYPE_F32 or HUAL_DTYPE_BF16");
  HUAL_REQUIRE(((uintptr_t)batch->video & 15) == 0, "video features must be 16-byte aligned");
  HUAL_REQUIRE(opts->drop_rate >= 0.f && opts->drop_rate < 1.f, "drop_rate in [0,1)");
  HUAL_REQUIRE(opts->drop_rate == 0.f || opts->rng_state != nullptr, "rng_state required when drop_rate > 0");
  HUAL_REQUIRE(cfg->no_gumbel || opts->rng_state != nullptr, "rng_state required when loss.no_gumbel is false");
  HUAL_REQUIRE(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)params & 15) == 0, "workspace/params alignment");
  return 0;
}

int hual_seqpan_forward(const hual_cfg* cfg, const float* params, const float* word_table, const hual_batch* batch,
                        const hual_labels* labels, const hual_outputs* out, const hual_run_opts* opts, void* workspace,
                        uint64_t ws_bytes, void* stream) {
  int rc = check_common(cfg, params, batch, opts, workspace);
  if (rc) return rc;
  HUAL_REQUIRE(word_table != nullptr, "null word_table");
  HUAL_REQUIRE(out && out->start_logits && out->end_logits && out->match_scores && out->start_index && out->end_index,
               "null output tensor");
  if (labels) HUAL_REQUIRE(labels->y1 && labels->y2 && labels->match_labels && labels->inner_labels, "null label tensor");
  Ctx c;
  rc = plan(c, cfg, batch->B, batch->T, batch->L, batch->C);
  if (rc) return rc;
  if (c.used + 4096 > ws_bytes) return fail(HUAL_ERR_WORKSPACE, "workspace too small: call hual_seqpan_query_workspace");
  c.dry = false;
  c.base = (char*)workspace;
  c.stream = (hipStream_t)stream;
  c.P = params;
  c.word_table = word_table;
  c.drop = make_dropcfg(opts->rng_state, opts->drop_rate);
  c.want_bwd = labels != nullptr;
  c.debug_taps = opts->debug_taps != 0;
  rc = forward_graph(c, batch, labels, out, opts);
  if (rc) return rc;
  return 0;
}

int hual_seqpan_backward(const hual_cfg* cfg, const float* params, const float* word_table, const hual_batch* batch,
                         const hual_labels* labels, const hual_run_opts* opts, float* grads, void* workspace,
                         uint64_t ws_bytes, void* stream) {
  int rc = check_common(cfg, params, batch, opts, workspace);
  if (rc) return rc;
  HUAL_REQUIRE(grads != nullptr && ((uintptr_t)grads & 15) == 0, "null/unaligned grads");
  HUAL_REQUIRE(labels && labels->y1 && labels->y2 && labels->match_labels && labels->inner_labels, "null label tensor");
  Ctx c;
  rc = plan(c, cfg, batch->B, batch->T, batch->L, batch->C);
  if (rc) return rc;
  if (c.used + 4096 > ws_bytes) return fail(HUAL_ERR_WORKSPACE, "workspace too small: call hual_seqpan_query_workspace");
  c.dry = false;
  c.base = (char*)workspace;
  c.stream = (hipStream_t)stream;
  c.P = params;
  c.G = grads;
  c.static_tables = opts->static_tables != 0;
  HUAL_REQUIRE(!opts->dw_table || ((uintptr_t)opts->dw_table & 15) == 0, "unaligned hual_run_opts.dw_table");
  c.ext_table = opts->dw_table;
  c.ext_table_bytes = (size_t)opts->dw_table_bytes;
  c.word_table = word_table;
  c.drop = make_dropcfg(opts->rng_state, opts->drop_rate);
  c.rs.rowmask = c.vec("rowmask");
  c.rsv.rowmask = c.rs.rowmask;
  return backward_graph(c, batch, labels, opts);
}

int hual_adamw_clip_step(float* params, const float* grads, float* adam_m, float* adam_v, const float* decay,
                         uint64_t n_padded, const float* lr, float clip_norm, float grad_prescale, float* sqnorm,
                         void* stream) {
  AdamArgs a{params, const_cast<float*>(grads), adam_m, adam_v, decay, (size_t)n_padded, lr, clip_norm, grad_prescale, sqnorm,
             nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
  return launch_adamw(a, (hipStream_t)stream);
}

int hual_adamw_clip_step_rng(float* params, const float* grads, float* adam_m, float* adam_v, const float* decay,
                             uint64_t n_padded, const float* lr, float clip_norm, float grad_prescale, float* sqnorm,
                             uint32_t* rng_state, void* stream) {
  AdamArgs a{params, const_cast<float*>(grads), adam_m, adam_v, decay, (size_t)n_padded, lr, clip_norm, grad_prescale, sqnorm,
             rng_state, nullptr, nullptr, nullptr, 0, 0, 0};
  return launch_adamw(a, (hipStream_t)stream);
}

int hual_adamw_clip_step_loop(float* params, const float* grads, float* adam_m, float* adam_v, const float* decay,
                              uint64_t n_padded, const float* lr, float clip_norm, float grad_prescale, float* sqnorm,
                              uint32_t* rng_state, int64_t* cursor, const int64_t* spans, int64_t* bank, int span_words, int sel_inc,
                              int bank_inc, void* stream) {
  HUAL_REQUIRE(cursor != nullptr && span_words >= 0 && sel_inc >= 0 && bank_inc >= 0, "hual_adamw_clip_step_loop: cursor / increments");
  AdamArgs a{params, const_cast<float*>(grads), adam_m, adam_v, decay, (size_t)n_padded, lr, clip_norm, grad_prescale, sqnorm,
             rng_state, cursor, spans, bank, span_words, sel_inc, bank_inc};
  return launch_adamw(a, (hipStream_t)stream);
}

int hual_align_loss(const float* that, const float* vhat, int Bg, float* scratch, float* d_that, float* d_vhat,
                    float* loss, float grad_scale, void* stream) {
  HUAL_REQUIRE(that && vhat && scratch && d_that && d_vhat && loss, "hual_align_loss: null pointer");
  // loss[0] is presented as the LA_ALIGN slot of an accumulator array; the row terms go through the tail of the scratch
  AlignSim as{that, vhat, Bg, scratch, scratch + (size_t)Bg * Bg, scratch + (size_t)2 * Bg * Bg, d_that, d_vhat, loss - LA_ALIGN, grad_scale, HUAL_D, 0, Bg};
  return launch_align_sim(as, (hipStream_t)stream);
}

int hual_align_loss_rows(const float* that, const float* vhat, int ld, int Bg, int row0, int nrows, float* scratch, float* d_that,
                         float* d_vhat, float* loss, float grad_scale, void* stream) {
  HUAL_REQUIRE(that && vhat && scratch && d_that && d_vhat && loss, "hual_align_loss_rows: null pointer");
  AlignSim as{that, vhat, Bg, scratch, scratch + (size_t)Bg * Bg, scratch + (size_t)2 * Bg * Bg, d_that, d_vhat, loss - LA_ALIGN, grad_scale, ld, row0, nrows};
  return launch_align_sim(as, (hipStream_t)stream);
}

}  // extern "C"

// ======================================================================================================
// per-block entry points: ONE stage of the graph on caller-supplied inputs, through the very launch
// sequence the whole model uses (Ctx::sel_stage).  Inputs are copied into the stage's input buffer in the workspace,
// results copied out of it; a backward call must follow the forward call of the same block on the same workspace.
// ======================================================================================================
namespace {
struct BlkCopy { std::string name; const float* src; float* dst; size_t row0, rows; };   // src: caller -> ws ; dst: ws -> caller

int run_block(const hual_cfg* cfg, const float* params, const float* word_table, const hual_batch* batch, const hual_run_opts* opts,
              float* grads, void* workspace, uint64_t ws_bytes, void* stream, int stage, int sub, bool backward,
              const std::vector<BlkCopy>& pre, const std::vector<BlkCopy>& post, const hual_outputs* out, const float** sum2_out = nullptr) {
  int rc = check_common(cfg, params, batch, opts, workspace);
  if (rc) return rc;
  HUAL_REQUIRE(!backward || (grads != nullptr && ((uintptr_t)grads & 15) == 0), "null/unaligned grads");
  Ctx c;
  rc = plan(c, cfg, batch->B, batch->T, batch->L, batch->C);
  if (rc) return rc;
  if (c.used + 4096 > ws_bytes) return fail(HUAL_ERR_WORKSPACE, "workspace too small: call hual_seqpan_query_workspace");
  c.dry = false;
  c.base = (char*)workspace;
  c.stream = (hipStream_t)stream;
  c.P = params;
  c.G = grads;
  c.word_table = word_table;
  c.drop = make_dropcfg(opts->rng_state, opts->drop_rate);
  c.sel_stage = stage; c.sel_sub = sub; c.want_bwd = true;
  c.debug_taps = opts->debug_taps != 0;
  c.part_seq = 0;
  c.rs.rowmask = c.vec("rowmask");
  c.rsv.rowmask = c.rs.rowmask;
  auto locate = [&](const BlkCopy& k, float*& ptr, size_t& cols) -> int {
    auto it = c.index.find(k.name);
    HUAL_REQUIRE(it != c.index.end(), "block entry point: unknown workspace buffer");
    const WsEntry& e = c.entries[it->second];
    HUAL_REQUIRE(k.row0 + k.rows <= e.rows, "block entry point: row window");
    ptr = reinterpret_cast<float*>(c.base + e.off) + k.row0 * e.cols;
    cols = e.cols;
    return 0;
  };
  for (const BlkCopy& k : pre) {
    float* ptr; size_t cols;
    if ((rc = locate(k, ptr, cols))) return rc;
    HUAL_CHECK_HIP(hipMemcpyAsync(ptr, k.src, k.rows * cols * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
  }
  hual_outputs dummy{};
  static const hual_run_opts o0{};
  (void)o0;
  rc = backward ? backward_graph(c, batch, nullptr, opts) : forward_graph(c, batch, nullptr, out ? out : &dummy, opts);
  if (rc) return rc;
  for (const BlkCopy& k : post) {
    float* ptr; size_t cols;
    if ((rc = locate(k, ptr, cols))) return rc;
    if (sum2_out && &k == &post.back() && post.size() == 2) {      // last two entries: dst = first + second (predictor backward)
      float* a; size_t ca;
      if ((rc = locate(post[0], a, ca))) return rc;
      return launch_add_rows(a, ptr, k.dst, (int)k.rows, c.stream);
    }
    if (sum2_out && post.size() == 2) continue;
    HUAL_CHECK_HIP(hipMemcpyAsync(k.dst, ptr, k.rows * cols * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
  }
  return 0;
}
std::string da_out_name(const hual_cfg* cfg, int layer) { return layer < 0 ? "cb.x4" : "da" + std::to_string(layer) + ".out"; }
std::string da_din_name(const hual_cfg* cfg, int layer) { return layer >= cfg->attn_layer ? "d.cq.x" : "d.da" + std::to_string(layer) + ".in"; }
}  // namespace

extern "C" {

int hual_video_proj_ln_fwd(const hual_cfg* cfg, const float* params, const float* word_table, const hual_batch* batch,
                           const hual_run_opts* opts, float* x0, void* workspace, uint64_t ws_bytes, void* stream) {
  HUAL_REQUIRE(cfg && batch && x0 && word_table, "hual_video_proj_ln_fwd: null argument");
  const size_t R = (size_t)batch->B * (batch->T + batch->L);
  return run_block(cfg, params, word_table, batch, opts, nullptr, workspace, ws_bytes, stream, ST_INPUT, 0, false, {},
                   {BlkCopy{"cb.x0", nullptr, x0, 0, R}}, nullptr);
}

int hual_conv_block_fwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, const float* x,
                        float* y, void* workspace, uint64_t ws_bytes, void* stream) {
  HUAL_REQUIRE(cfg && batch && x && y, "hual_conv_block_fwd: null argument");
  const size_t R = (size_t)batch->B * (batch->T + batch->L);
  return run_block(cfg, params, nullptr, batch, opts, nullptr, workspace, ws_bytes, stream, ST_CONV, 0, false,
                   {BlkCopy{"cb.x0", x, nullptr, 0, R}}, {BlkCopy{"cb.x4", nullptr, y, 0, R}}, nullptr);
}
int hual_conv_block_bwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, const float* dy,
                        float* dx, float* grads, void* workspace, uint64_t ws_bytes, void* stream) {
  HUAL_REQUIRE(cfg && batch && dy && dx, "hual_conv_block_bwd: null argument");
  const size_t R = (size_t)batch->B * (batch->T + batch->L);
  return run_block(cfg, params, nullptr, batch, opts, grads, workspace, ws_bytes, stream, ST_CONV, 0, true,
                   {BlkCopy{da_din_name(cfg, 0), dy, nullptr, 0, R}}, {BlkCopy{"d.cb.x0", nullptr, dx, 0, R}}, nullptr);
}

int hual_dual_attn_fwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, int layer,
                       const float* x, float* y, void* workspace, uint64_t ws_bytes, void* stream) {
  HUAL_REQUIRE(cfg && batch && x && y && layer >= 0 && layer < cfg->attn_layer, "hual_dual_attn_fwd: null argument / layer");
  const size_t R = (size_t)batch->B * (batch->T + batch->L);
  return run_block(cfg, params, nullptr, batch, opts, nullptr, workspace, ws_bytes, stream, ST_DA, layer, false,
                   {BlkCopy{da_out_name(cfg, layer - 1), x, nullptr, 0, R}}, {BlkCopy{da_out_name(cfg, layer), nullptr, y, 0, R}}, nullptr);
}
int hual_dual_attn_bwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, int layer,
                       const float* dy, float* dx, float* grads, void* workspace, uint64_t ws_bytes, void* stream) {
  HUAL_REQUIRE(cfg && batch && dy && dx && layer >= 0 && layer < cfg->attn_layer, "hual_dual_attn_bwd: null argument / layer");
  const size_t R = (size_t)batch->B * (batch->T + batch->L);
  return run_block(cfg, params, nullptr, batch, opts, grads, workspace, ws_bytes, stream, ST_DA, layer, true,
                   {BlkCopy{da_din_name(cfg, layer + 1), dy, nullptr, 0, R}}, {BlkCopy{da_din_name(cfg, layer), nullptr, dx, 0, R}}, nullptr);
}

int hual_cq_attn_fwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, const float* x,
                     float* feats, void* workspace, uint64_t ws_bytes, void* stream) {
  HUAL_REQUIRE(cfg && batch && x && feats, "hual_cq_attn_fwd: null argument");
  const size_t R = (size_t)batch->B * (batch->T + batch->L);
  return run_block(cfg, params, nullptr, batch, opts, nullptr, workspace, ws_bytes, stream, ST_CQ, 0, false,
                   {BlkCopy{da_out_name(cfg, cfg->attn_layer - 1), x, nullptr, 0, R}}, {BlkCopy{"cq.feats", nullptr, feats, 0, R}}, nullptr);
}
int hual_cq_attn_bwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, const float* dfeats,
                     float* dx, float* grads, void* workspace, uint64_t ws_bytes, void* stream) {
  HUAL_REQUIRE(cfg && batch && dfeats && dx, "hual_cq_attn_bwd: null argument");
  const size_t R = (size_t)batch->B * (batch->T + batch->L);
  return run_block(cfg, params, nullptr, batch, opts, grads, workspace, ws_bytes, stream, ST_CQ, 0, true,
                   {BlkCopy{"d.cq.feats", dfeats, nullptr, 0, R}}, {BlkCopy{"d.cq.x", nullptr, dx, 0, R}}, nullptr);
}

int hual_predictor_fwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, const float* outputs,
                       float* start_logits, float* end_logits, int64_t* start_index, int64_t* end_index, void* workspace,
                       uint64_t ws_bytes, void* stream) {
  HUAL_REQUIRE(cfg && batch && outputs && start_logits && end_logits && start_index && end_index, "hual_predictor_fwd: null argument");
  const size_t Nv = (size_t)batch->B * batch->T;
  hual_outputs o{};
  o.start_logits = start_logits; o.end_logits = end_logits; o.start_index = start_index; o.end_index = end_index;
  return run_block(cfg, params, nullptr, batch, opts, nullptr, workspace, ws_bytes, stream, ST_PRED, 0, false,
                   {BlkCopy{"outputs", outputs, nullptr, 0, Nv}}, {}, &o);
}
int hual_predictor_bwd(const hual_cfg* cfg, const float* params, const hual_batch* batch, const hual_run_opts* opts, const float* d_start,
                       const float* d_end, float* d_outputs, float* grads, void* workspace, uint64_t ws_bytes, void* stream) {
  HUAL_REQUIRE(cfg && batch && d_start && d_end && d_outputs, "hual_predictor_bwd: null argument");
  const size_t Nv = (size_t)batch->B * batch->T;
  const float* marker = nullptr;
  return run_block(cfg, params, nullptr, batch, opts, grads, workspace, ws_bytes, stream, ST_PRED, 0, true,
                   {BlkCopy{"d.s_logit", d_start, nullptr, 0, (size_t)batch->B}, BlkCopy{"d.e_logit", d_end, nullptr, 0, (size_t)batch->B}},
                   {BlkCopy{"d.fe0.x0", nullptr, nullptr, 0, Nv}, BlkCopy{"d.outputs.heads", nullptr, d_outputs, 0, Nv}}, nullptr, &marker);
}

}  // extern "C"
