// Flat parameter layout.  One fp32 buffer holds every trainable variable of the SeqPAN graph in the
// graph-construction order of /root/reference/models/model.py:29-122, each tensor 16-byte aligned, under its
// TensorFlow variable-scope name and TF shape so a checkpoint importer is rename-free.
// Gradients and the two Adam slots (`<var>/adam_m`, `<var>/adam_v`, ops.py:156-165) use the same layout, which
// is also the single RCCL all-reduce bucket in data-parallel training.
#pragma once
#include "common.h"
#include <string>
#include <vector>

namespace hual {

struct ParamEntry {
  std::string name;
  size_t off;        // in floats
  size_t size;       // in floats (unpadded)
  int ndim;
  int shape[4];
  int decay;         // 1: weight decay applies (name matches none of LayerNorm|layer_norm|bias, ops.py:123,176-184)
};

struct LnP { size_t g, b; };
struct DenseP { size_t k, b; };
struct ConvBlockP { LnP ln[4]; size_t dw[4], pw[4], b[4]; };
struct DualAttnP {
  LnP ln1, lnt, ln2;
  DenseP query, f_key, f_value, t_key, t_value, s_dense, x_dense, s_gate, x_gate, guided;
  size_t bl1_d1, bl1_d2, bl1_b, bl2_d1, bl2_d2, bl2_b;
  DenseP dense1, dense2;
};
struct CqP { size_t w0, w1, wm, dense; };

#define HUAL_MAX_ATTN_LAYERS 8

struct ParamMap {
  size_t unk, char_table, filt[4], fbias[4];
  DenseP qconv; LnP qln; DenseP vconv; LnP vln;
  size_t pos;
  ConvBlockP cb;
  DualAttnP da[HUAL_MAX_ATTN_LAYERS];
  CqP cq[2];
  size_t pool_w; DenseP cqcat; DenseP match; size_t label_emb;
  size_t fe_pos; ConvBlockP fe_cb; LnP fe_ln1, fe_ln2; DenseP fe_q, fe_k, fe_v, fe_dense;
  LnP sln, eln; DenseP shid, ehid, sdense, edense;
  size_t total;      // padded flat size in floats (multiple of 4)
  size_t count;      // true number of trainable scalars
  std::vector<ParamEntry> entries;
};

int validate_cfg(const hual_cfg& c);
int build_param_map(const hual_cfg& c, ParamMap& m);

}  // namespace hual
