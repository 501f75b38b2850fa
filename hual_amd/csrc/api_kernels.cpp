// C-ABI wrappers of the per-kernel entry points declared in include/hual_seqpan.h.
#include "../../include/hual_seqpan.h"
#include "common.h"
#include "gemm.h"
#include "mproj.h"
#include "attn.h"
#include "heads.h"
#include "rowops.h"

namespace hual {
const char* last_error_cstr();
}

using namespace hual;

extern "C" {

int hual_abi_version(void) { return HUAL_ABI_VERSION; }
const char* hual_last_error(void) { return hual::last_error_cstr(); }

int hual_linear_dw(const float* A, int lda, const float* dY, int ldy, float* dW, int ldw, float* db, int M, int K,
                   int N, int workgroups, void* scratch, uint64_t scratch_bytes, void* stream) {
  HUAL_REQUIRE(A && dY && dW, "hual_linear_dw: null pointer");
  HUAL_REQUIRE(M > 0 && K % 16 == 0 && N == 128, "hual_linear_dw: K % 16 == 0 and N == 128 required");
  HUAL_REQUIRE(workgroups >= 0, "hual_linear_dw: workgroups >= 0 (0: one per CU)");
  HUAL_REQUIRE(scratch && scratch_bytes >= 4 * dw_table_words(1), "hual_linear_dw: scratch table too small");
  DwJob j;
  dw_job_init(j);
  j.npieces = 1;
  j.A[0] = A; j.lda[0] = lda; j.kw[0] = K; j.dW[0] = dW; j.ldw = ldw;
  j.dY = dY; j.ldy = ldy; j.M = M; j.N = N; j.db = db;
  DropCfg d = make_dropcfg(nullptr, 0.f);
  return launch_dw(&j, 1, d, (hipStream_t)stream, reinterpret_cast<DwJob*>(scratch), true, workgroups);
}

int hual_linear_bf16x3(const float* A, int lda, const float* W, int trans_w, const float* bias, float* Y, int ldy, int M,
                       int K, int N, int act, void* scratch, uint64_t scratch_bytes, void* stream) {
  HUAL_REQUIRE(A && W && Y && scratch, "hual_linear_bf16x3: null pointer");
  HUAL_REQUIRE(M > 0 && K > 0 && N > 0 && act >= 0 && act <= 1, "hual_linear_bf16x3: bad shape / act (0 none, 1 relu)");
  HUAL_REQUIRE((lda % 4) == 0 && (ldy % 4) == 0, "hual_linear_bf16x3: leading dims must be multiples of 4");
  const uint32_t off = 0, boff = 0;
  MProjArgs g{};
  g.R = M; g.MT = mproj_rows(M);
  if (!trans_w) {        // Y = act(A[M,K] . W[K,128] + bias): ceil(K / 128) weight steps over one deep operand
    HUAL_REQUIRE(N == 128 && K % 8 == 0, "hual_linear_bf16x3: W must be [K,128] with K % 8 == 0");
    HUAL_REQUIRE(scratch_bytes >= (uint64_t)((K + 127) / 128) * 65536, "hual_linear_bf16x3: scratch too small (ceil(K/128) * 65536 bytes)");
    int rc = launch_pack_weights(&off, &K, &boff, 1, W, nullptr, (hipStream_t)stream, nullptr, (char*)scratch, nullptr);
    if (rc) return rc;
    g.nsteps = 1;
    MProjStep& st = g.s[0];
    st.A = A; st.lda = lda; st.rep = (K + 127) / 128; st.ktot = K; st.kw = K < 128 ? K : 128; st.wimg = reinterpret_cast<const float*>(scratch); st.wrows = K;
    st.first = 1; st.last = 1; st.drop_site = -1; st.add_div = 1; st.bias = bias; st.act = act; st.out = Y; st.ldo = ldy; st.ncol = 128;
  } else {               // Y[M,N] = A[M,128] . W^T, W stored [N,128]  (dX of a dense layer with weight W): ceil(N / 128) column blocks
    HUAL_REQUIRE(K == 128 && N % 8 == 0 && (N + 127) / 128 <= MP_MAX, "hual_linear_bf16x3: transposed use needs K == 128, N % 8 == 0, N <= 1024");
    HUAL_REQUIRE(scratch_bytes >= (uint64_t)((N + 127) / 128) * HUAL_PACK_BLOCK_BYTES, "hual_linear_bf16x3: scratch too small");
    int rc = launch_pack_weights(&off, &N, &boff, 1, W, nullptr, (hipStream_t)stream, nullptr, nullptr, (char*)scratch);
    if (rc) return rc;
    g.nsteps = (N + 127) / 128;
    for (int p = 0; p < g.nsteps; ++p) {
      MProjStep& st = g.s[p];
      st.A = A; st.lda = lda; st.kw = 128; st.reuse = p > 0 ? 1 : 0; st.rep = 1;
      st.wimg = reinterpret_cast<const float*>((const char*)scratch + (size_t)p * HUAL_PACK_BLOCK_BYTES); st.wrows = 128;
      st.first = 1; st.last = 1; st.drop_site = -1; st.add_div = 1; st.act = act;
      st.out = Y + (size_t)p * 128; st.ldo = ldy; st.ncol = N - 128 * p < 128 ? N - 128 * p : 128;
    }
  }
  DropCfg d = make_dropcfg(nullptr, 0.f);
  return launch_mproj(&g, 1, d, (hipStream_t)stream);
}

int hual_layer_norm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int R,
                        void* stream) {
  HUAL_REQUIRE(x && gamma && beta && y && R > 0, "hual_layer_norm_fwd: null / empty");
  LnFwd a{};
  a.x = x; a.R = R; a.g1 = gamma; a.b1 = beta; a.y1 = y; a.mean = mean; a.rstd = rstd;
  RowSpace rs{};
  rs.R = R; rs.Nv = R; rs.T = R; rs.B = 1;
  DropCfg d = make_dropcfg(nullptr, 0.f);
  return launch_ln_fwd(a, rs, d, (hipStream_t)stream);
}

int hual_attention_fwd(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* O, int ldo, int B, int Tq,
                       int Tk, const float* qmask, const float* kmask, void* stream) {
  HUAL_REQUIRE(Q && K && V && O && qmask && kmask, "hual_attention_fwd: null pointer");
  AttnJob j;
  attn_job_init(j);
  j.Q = Q; j.ldq = ldq; j.K = K; j.V = V; j.ldkv = ldkv; j.O = O; j.ldo = ldo;
  j.B = B; j.Tq = Tq; j.Tk = Tk; j.qrow0 = 0; j.krow0 = 0; j.qmask = qmask; j.kmask = kmask;
  DropCfg d = make_dropcfg(nullptr, 0.f);
  return launch_attn_fwd(&j, 1, d, (hipStream_t)stream);
}

int hual_attention_fwd_save(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* O, int ldo, int B, int Tq,
                            int Tk, const float* qmask, const float* kmask, float* stats, uint8_t* keep_bytes, int ldm,
                            const uint32_t* rng_state, float drop_rate, int drop_site, void* stream) {
  HUAL_REQUIRE(Q && K && V && O && qmask && kmask && stats, "hual_attention_fwd_save: null pointer");
  HUAL_REQUIRE(drop_rate >= 0.f && drop_rate < 1.f && (drop_rate == 0.f || (rng_state && keep_bytes)), "hual_attention_fwd_save: dropout needs rng_state and keep_bytes");
  AttnJob j;
  attn_job_init(j);
  j.Q = Q; j.ldq = ldq; j.K = K; j.V = V; j.ldkv = ldkv; j.O = O; j.ldo = ldo;
  j.B = B; j.Tq = Tq; j.Tk = Tk; j.qrow0 = 0; j.krow0 = 0; j.qmask = qmask; j.kmask = kmask;
  HUAL_REQUIRE(!keep_bytes || ldm >= attn_ldm(Tk), "hual_attention_fwd_save: ldm < hual_attention_keep_row_bytes(Tk)");
  j.stats = stats; j.dmask = keep_bytes; j.drop_site = drop_rate > 0.f ? drop_site : -1; j.drop_row0 = 0;
  DropCfg d = make_dropcfg(rng_state, drop_rate);
  return launch_attn_fwd(&j, 1, d, (hipStream_t)stream);
}

int hual_attention_bwd(const float* Q, int ldq, const float* K, const float* V, int ldkv, const float* O, int ldo,
                       const float* stats, const uint8_t* keep_bytes, int ldm, const float* dO, int lddo, float* dQ, int lddq,
                       float* dK, float* dV, int lddkv, int B, int Tq, int Tk, const float* qmask, const float* kmask,
                       const uint32_t* rng_state, float drop_rate, int drop_site, void* stream) {
  HUAL_REQUIRE(Q && K && V && O && stats && dO && dQ && dK && dV && qmask && kmask, "hual_attention_bwd: null pointer");
  HUAL_REQUIRE(drop_rate >= 0.f && drop_rate < 1.f && (drop_rate == 0.f || (rng_state && keep_bytes)), "hual_attention_bwd: dropout needs rng_state and keep_bytes");
  AttnJob j;
  attn_job_init(j);
  j.Q = Q; j.ldq = ldq; j.K = K; j.V = V; j.ldkv = ldkv; j.O = const_cast<float*>(O); j.ldo = ldo;
  j.B = B; j.Tq = Tq; j.Tk = Tk; j.qrow0 = 0; j.krow0 = 0; j.qmask = qmask; j.kmask = kmask;
  HUAL_REQUIRE(!keep_bytes || ldm >= attn_ldm(Tk), "hual_attention_bwd: ldm < hual_attention_keep_row_bytes(Tk)");
  j.stats = const_cast<float*>(stats); j.dmask = const_cast<uint8_t*>(keep_bytes);
  j.drop_site = drop_rate > 0.f ? drop_site : -1; j.drop_row0 = 0;
  j.dO = dO; j.lddo = lddo; j.dQ = dQ; j.lddq = lddq; j.dK = dK; j.dV = dV; j.lddkv = lddkv;
  DropCfg d = make_dropcfg(rng_state, drop_rate);
  return launch_attn_bwd(&j, 1, d, (hipStream_t)stream);
}

int hual_attention_keep_row_bytes(int Tk) { return attn_ldm(Tk); }

int hual_span_argmax(const float* start_logits, const float* end_logits, const float* vmask, int64_t* start_index,
                     int64_t* end_index, int B, int T, void* stream) {
  HUAL_REQUIRE(start_logits && end_logits && vmask && start_index && end_index && B > 0 && T > 0, "hual_span_argmax: null / empty");
  HeadsArgs a{};
  a.logit[0] = const_cast<float*>(start_logits); a.logit[1] = const_cast<float*>(end_logits); a.vmask = vmask;
  a.start_index = start_index; a.end_index = end_index; a.inv_batch = 1.0f / (float)B;
  return launch_heads(a, B, T, (hipStream_t)stream);
}

}  // extern "C"
