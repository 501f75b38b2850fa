// Multi-head scaled-dot attention with the reference's additive outer-product mask, softmax and dropout on the
// probabilities: /root/reference/models/layers.py:80-96 (self + cross attention of dual_multihead_attention) and
// /root/reference/models/modules.py:104-119 (top_self_attention).  Heads stay merged: head h = columns [16h,16h+16).
#pragma once
#include "common.h"

namespace hual {

// One attention problem = B clips; queries of clip b are rows qrow0 + b*Tq + [0,Tq) of Q/dO/O,
// keys/values are rows krow0 + b*Tk + [0,Tk) of K/V.  Masks are indexed by the same absolute rows.
struct AttnJob {
  const float* Q; int ldq;
  const float* K; const float* V; int ldkv;
  float* O; int ldo;                       // forward output, merged heads [rows,128] (an input of the backward)
  float* stats;                            // [2][B*Tq*8]: row max (of the scores in the log2 domain), 1/rowsum per
                                           // (query, head); written by the forward when non-null, required by the backward
  int B, Tq, Tk, qrow0, krow0;
  const float* qmask; const float* kmask;  // [rows] floats
  int drop_site; uint32_t drop_row0;       // Philox row = (drop_row0 + qrow) * 8 + head ; 16-bit decisions, see attn.hip
  // keep words of the dropout on the probabilities: per (clip b, head h) a block of nqt * nkt * 4 words of 8 bytes (nqt, nkt = query /
  // key tiles of 16) at dmask + (b * 8 + h) * nqt * nkt * 32; word (qt * nkt + kt) * 4 + r, bit 16 g + jq = (query 16 qt + jq,
  // key 16 kt + 4 g + r) is kept.  Written by the forward when non-null; the backward needs it whenever dropout is on.
  uint8_t* dmask;                          // attn_keep_bytes(B, Tq, Tk) bytes, 8-byte aligned
  // backward
  const float* dO; int lddo;
  float* dQ; int lddq;                     // written (not accumulated)
  float* dK; float* dV; int lddkv;         // written (not accumulated)
};

#define HUAL_MAX_ATTN_JOBS 4
struct AttnBatch {
  AttnJob j[HUAL_MAX_ATTN_JOBS];
};
// forward: the (job, 16-query tile) units of a launch, sorted by cost on the host: code = job << 4 | tile
#define HUAL_MAX_ATTN_UNITS 64
struct AttnUnits {
  int n;
  uint8_t u[HUAL_MAX_ATTN_UNITS];
};

void attn_job_init(AttnJob& j);
int attn_ldm(int Tk);                     // public sizing rule (hual_attention_keep_row_bytes): B * Tq * 8 rows of this many bytes hold the keep words
size_t attn_keep_bytes(int B, int Tq, int Tk);      // exact size of a job's keep words
int launch_attn_fwd(const AttnJob* jobs, int n, const DropCfg& drop, hipStream_t s);
// dQ, dK, dV in one launch (needs O and stats of the forward)
int launch_attn_bwd(const AttnJob* jobs, int n, const DropCfg& drop, hipStream_t s);

}  // namespace hual
