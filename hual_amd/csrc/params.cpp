// Flat parameter layout (see params.h).
#include "params.h"
#include <initializer_list>

namespace hual {

int validate_cfg(const hual_cfg& c) {
  HUAL_REQUIRE(c.dim == HUAL_D, "model.dim must be 128 (kernels are specialised for it)");
  HUAL_REQUIRE(c.num_heads == HUAL_H, "model.num_heads must be 8 (head size 16)");
  HUAL_REQUIRE(c.vdim >= 16 && c.vdim % 16 == 0, "model.vdim must be a positive multiple of 16");
  HUAL_REQUIRE(c.word_dim >= 4 && (c.word_dim + 100) % 16 == 0, "word_dim + 100 must be a multiple of 16");
  HUAL_REQUIRE(c.char_dim >= 1 && c.char_dim <= 128, "char_dim in [1,128]");
  HUAL_REQUIRE(c.max_vlen >= 1 && c.max_vlen <= 256, "max_vlen in [1,256]");
  HUAL_REQUIRE(c.attn_layer >= 1 && c.attn_layer <= HUAL_MAX_ATTN_LAYERS, "attn_layer in [1,8]");
  HUAL_REQUIRE(c.num_chars >= 2 && c.num_words >= 2, "num_chars / num_words too small");
  HUAL_REQUIRE(c.no_gumbel || c.tau > 0.f, "loss.tau must be positive when loss.no_gumbel is false");
  return 0;
}

namespace {
struct Builder {
  ParamMap& m;
  size_t cur = 0;
  size_t add(const std::string& name, std::initializer_list<int> shape) {
    ParamEntry e;
    e.name = name;
    e.ndim = (int)shape.size();
    size_t n = 1;
    int i = 0;
    for (int s : shape) { e.shape[i++] = s; n *= (size_t)s; }
    for (; i < 4; ++i) e.shape[i] = 1;
    e.size = n;
    e.off = cur;
    e.decay = (name.find("LayerNorm") == std::string::npos && name.find("layer_norm") == std::string::npos &&
               name.find("bias") == std::string::npos) ? 1 : 0;
    m.entries.push_back(e);
    m.count += n;
    cur += (n + 3) & ~(size_t)3;
    return e.off;
  }
  LnP ln(const std::string& p) {
    LnP r;
    r.g = add(p + "/layer_norm_scale", {HUAL_D});
    r.b = add(p + "/layer_norm_bias", {HUAL_D});
    return r;
  }
  DenseP dense(const std::string& p, int cin, int cout, bool bias = true) {
    DenseP r;
    r.k = add(p + "/kernel", {1, cin, cout});
    r.b = bias ? add(p + "/bias", {1, 1, cout}) : (size_t)-1;
    return r;
  }
  ConvBlockP conv_block(const std::string& p) {
    ConvBlockP r;
    for (int i = 0; i < 4; ++i) {
      const std::string is = std::to_string(i);
      r.ln[i] = ln(p + "/layer_norm_" + is);
      const std::string d = p + "/depthwise_conv_layers_" + is;
      r.dw[i] = add(d + "/depthwise_filter", {7, 1, HUAL_D, 1});
      r.pw[i] = add(d + "/pointwise_filter", {1, 1, HUAL_D, HUAL_D});
      r.b[i] = add(d + "/bias", {HUAL_D});
    }
    return r;
  }
};
}  // namespace

int build_param_map(const hual_cfg& c, ParamMap& m) {
  int rc = validate_cfg(c);
  if (rc) return rc;
  m.entries.clear();
  m.count = 0;
  Builder b{m};
  const int D = HUAL_D;
  static const int KS[4] = {1, 2, 3, 4}, CH[4] = {10, 20, 30, 40};
  m.unk = b.add("word_embs/unk", {1, c.word_dim});
  m.char_table = b.add("char_embs/char_table", {c.num_chars - 1, c.char_dim});
  for (int i = 0; i < 4; ++i) {
    m.filt[i] = b.add("char_embs/filter_" + std::to_string(i), {1, KS[i], c.char_dim, CH[i]});
    m.fbias[i] = b.add("char_embs/bias_" + std::to_string(i), {CH[i]});
  }
  m.qconv = b.dense("query_conv1d", c.word_dim + 100, D);
  m.qln = b.ln("q_layer_norm");
  m.vconv = b.dense("video_conv1d", c.vdim, D);
  m.vln = b.ln("v_layer_norm");
  m.pos = b.add("pos_emb/position_embeddings", {c.max_vlen, D});
  m.cb = b.conv_block("conv_block");
  for (int li = 0; li < c.attn_layer; ++li) {
    const std::string p = "d_attn_" + std::to_string(li);
    DualAttnP& d = m.da[li];
    d.ln1 = b.ln(p + "/layer_norm_1");
    d.lnt = b.ln(p + "/layer_norm_t");
    const std::string a = p + "/dual_multihead_attention";
    d.query = b.dense(a + "/query", D, D);
    d.f_key = b.dense(a + "/f_key", D, D);
    d.f_value = b.dense(a + "/f_value", D, D);
    d.t_key = b.dense(a + "/t_key", D, D);
    d.t_value = b.dense(a + "/t_value", D, D);
    d.s_dense = b.dense(a + "/s_dense", D, D);
    d.x_dense = b.dense(a + "/x_dense", D, D);
    d.s_gate = b.dense(a + "/s_gate", D, D);
    d.x_gate = b.dense(a + "/x_gate", D, D);
    d.guided = b.dense(a + "/guided_dense", D, D);
    d.bl1_d1 = b.add(a + "/bilinear_1/dense_1/kernel", {1, D, D});
    d.bl1_d2 = b.add(a + "/bilinear_1/dense_2/kernel", {1, D, D});
    d.bl1_b = b.add(a + "/bilinear_1/bias", {D});
    d.bl2_d1 = b.add(a + "/bilinear_2/dense_1/kernel", {1, D, D});
    d.bl2_d2 = b.add(a + "/bilinear_2/dense_2/kernel", {1, D, D});
    d.bl2_b = b.add(a + "/bilinear_2/bias", {D});
    d.dense1 = b.dense(p + "/dense_1", D, D);
    d.ln2 = b.ln(p + "/layer_norm_2");
    d.dense2 = b.dense(p + "/dense_2", D, D);
  }
  const char* cqn[2] = {"q2v_attn", "v2q_attn"};
  for (int i = 0; i < 2; ++i) {
    const std::string n = cqn[i];
    m.cq[i].w0 = b.add(n + "/efficient_trilinear/linear_kernel4arg0", {D, 1});
    m.cq[i].w1 = b.add(n + "/efficient_trilinear/linear_kernel4arg1", {D, 1});
    m.cq[i].wm = b.add(n + "/efficient_trilinear/linear_kernel4mul", {1, 1, D});
    m.cq[i].dense = b.add(n + "/dense/kernel", {1, 4 * D, D});
  }
  m.pool_w = b.add("cq_cat/weighted_pooling/weight", {D, 1});
  m.cqcat = b.dense("cq_cat/dense", 2 * D, D);
  m.match = b.dense("matching_loss/dense", D, 4);
  m.label_emb = b.add("label_emb", {4, D});
  const std::string fe = "predictor/feature_encoder";
  m.fe_pos = b.add(fe + "/pos_emb/position_embeddings", {c.max_vlen, D});
  m.fe_cb = b.conv_block(fe + "/conv_block");
  const std::string mb = fe + "/multihead_attention_block";
  m.fe_ln1 = b.ln(mb + "/layer_norm_1");
  m.fe_q = b.dense(mb + "/top_self_attention/query", D, D);
  m.fe_k = b.dense(mb + "/top_self_attention/key", D, D);
  m.fe_v = b.dense(mb + "/top_self_attention/value", D, D);
  m.fe_ln2 = b.ln(mb + "/layer_norm_2");
  m.fe_dense = b.dense(mb + "/dense", D, D);
  m.sln = b.ln("predictor/start_layer_norm");
  m.eln = b.ln("predictor/end_layer_norm");
  m.shid = b.dense("predictor/start_hidden", 2 * D, D);
  m.ehid = b.dense("predictor/end_hidden", 2 * D, D);
  m.sdense = b.dense("predictor/start_dense", D, 1);
  m.edense = b.dense("predictor/end_dense", D, 1);
  m.total = b.cur;
  return 0;
}

}  // namespace hual
