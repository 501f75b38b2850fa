// Job-table GEMM family on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32 / 32x32x2_f32).
// Every dense contraction of the SeqPAN graph - the 46 conv1d(kernel_size=1) weight sets of
// /root/reference/models/layers.py:20-29 and their gradients - goes through these two kernels.
#pragma once
#include "common.h"

namespace hual {

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_SIGMOID_ROWMASK = 3 };
enum MulMode { MUL_NONE = 0, MUL_TENSOR = 1, MUL_DRELU = 2, MUL_DSIGMOID = 3 };
enum Comb { COMB_NONE = 0, COMB_GATE_VAL = 1, COMB_CROSSGATE = 2 };

#define HUAL_MAX_PIECES 4
#define HUAL_MAX_JOBS 6

// Y[M,N] = epilogue( sum_p prologue(A[p])[M,kw[p]] . W[p] )   (N multiple of 64, kw multiple of 16)
struct GemmJob {
  // ---- A operand: up to 4 column pieces (concatenation along K), optional elementwise multiplier
  const float* A[HUAL_MAX_PIECES];
  const float* A2[HUAL_MAX_PIECES];
  int lda[HUAL_MAX_PIECES];
  int lda2[HUAL_MAX_PIECES];
  int kw[HUAL_MAX_PIECES];
  int npieces;
  // prologue extras on piece 0, applied in this order after the A2 product (gemm_lds_kernel only):
  const float* ln_g; const float* ln_b;    // layer norm over the 128 columns of piece 0 (kw[0] must be 128)
  float* ln_mean; float* ln_rstd;          // optional per-row statistics out
  int a_drop_site;            // >=0: A = dropout(A) (tf.nn.dropout on the GEMM input, model.py:47)
  uint32_t a_drop_row0;
  const float* a_relu; int lda_relu;       // A *= (a_relu > 0)          (relu' of a saved activation)
  float* a_save; int lda_save;             // store the transformed operand (LN output / dZ) for later kernels
  // ---- B operand: one weight block per piece. transW=0: W[p][k][n] (ldw); transW=1: W[p][n][k] (used for dX)
  const float* W[HUAL_MAX_PIECES];
  int ldw;
  int transW;
  const float* bias;          // [N] or null
  int M, N;
  // ---- epilogue pipeline: +bias -> act -> save -> mul -> dropout -> +add -> *rowmask -> Y
  int act;
  const float* rowmask;       // [M] floats (0/1)
  float* save; int ldsave;
  int mulmode; const float* mul; int ldmul;
  int drop_site; uint32_t drop_row0;
  const float* add; int ldadd; int add_div;
  int mask_out;
  float* Y; int ldy;
  // ---- dual mode (second accumulator): A_b pieces (null => same as A), W2/bias2, combine
  const float* Ab[HUAL_MAX_PIECES];
  int ldab[HUAL_MAX_PIECES];
  const float* W2[HUAL_MAX_PIECES];
  const float* bias2;
  int comb;
  float* save2; int ldsave2;
  const float* aux1; const float* aux2; int ldaux;   // COMB_CROSSGATE: Y = sig1*aux1 + sig2*aux2
};

struct GemmBatch {
  GemmJob j[HUAL_MAX_JOBS];
};

// dW[p][k][n] += sum_m prologue(A[p])[m][k] * dY[m][n] ;  db[n] += sum_m dY[m][n]     (N multiple of 64)
struct DwJob {
  const float* A[HUAL_MAX_PIECES];
  const float* A2[HUAL_MAX_PIECES];
  int lda[HUAL_MAX_PIECES];
  int lda2[HUAL_MAX_PIECES];
  int kw[HUAL_MAX_PIECES];
  int npieces;
  int a_drop_site;
  uint32_t a_drop_row0;
  float* dW[HUAL_MAX_PIECES];
  int ldw;
  const float* dY; int ldy;
  int M, N;
  float* db;
  int a_bf16;                 // 1: A[0] points at bfloat16 elements (lda in elements); single piece, no A2
  // keep bits of the a_drop_site dropout as the forward stored them (one byte per 4 columns, ld_keep bytes per row;
  // FkJob::keep_out): the kernel then skips the Philox rounds.  Single piece.
  const uint8_t* a_keep; int ld_keep;
};

#define HUAL_MAX_DW_JOBS 12
struct DwBatch {
  DwJob j[HUAL_MAX_DW_JOBS];
};

// dst[off + n*K + k] = src[off + k*N + n] for n dense weights [K,N] inside the flat parameter buffer
int launch_transpose_weights(const uint32_t* offs, const int* Ks, const int* Ns, int n, const float* src, float* dst,
                             hipStream_t stream);
// Split-bf16 path (gemm_bf16_kernel, bf16x3.h).  launch_pack_weights writes, for n dense weights [K,128] at float offsets
// offs[] of P, the forward image at fwd + 4*off and the image of the transposed weight (for dX) at bwd + boffs[]
// (ceil(K/128) blocks of 64 KB); either destination may be null.  launch_gemm_bf16 takes jobs whose W[p] / W2[p] point at
// such images (reinterpreted) - N > 128 walks the 64 KB column blocks of a backward image.
#define HUAL_PACK_BLOCK_BYTES (128 * 512)
// `extra` (optional): work of the step's prologue that rides in the same launch as one more row of workgroups - the row
// masks of model.py:31-32 (+ the loss accumulators cleared) and, when zero_ptr is set, the flat gradient buffer zeroed
struct PackExtra {
  const int32_t* lens; const int32_t* word_ids; float* rowmask; float* loss_acc; int B, T, L;
  float* zero_ptr; size_t zero_n;      // zero_n floats (multiple of 4), 16-byte aligned
};
int launch_pack_weights(const uint32_t* offs, const int* Ks, const uint32_t* boffs, int n, const float* P, char* fwd, char* bwd,
                        hipStream_t stream, const PackExtra* extra = nullptr);
int launch_gemm_bf16(const GemmJob* jobs, int n, const DropCfg& drop, hipStream_t stream);
// the same jobs executed one after the other inside each block (row-local dependent layers; same M, N = 128)
int launch_gemm_chain(const GemmJob* jobs, int n, const DropCfg& drop, hipStream_t stream);
// Feature-load kernel: part[q][M][128] = dropout(A)[M, q*KS .. q*KS+KS) . W[q*KS .., :] for the four K-quarters q (Wimg =
// forward image of the [K,128] weight, K <= 4*KS); the sum over q (+ bias) is taken by the consumer (ln_fwd_kernel's `part`).
struct FkJob {
  const float* A; int lda; int M; int K; int KS;
  const float* Wimg;
  float* part; size_t part_stride;
  int drop_site; uint32_t drop_row0;
  int a_bf16;                 // 1: A points at bfloat16 elements (lda in elements)
  uint8_t* keep_out; int ld_keep;   // optional: the dropout keep bits, one byte per 4 columns (bit c = column 4g + c)
};
#define HUAL_MAX_FK_JOBS 2
struct FkBatch { FkJob j[HUAL_MAX_FK_JOBS]; };
int launch_feature_ksplit(const FkJob* jobs, int n, const DropCfg& drop, hipStream_t stream);
void gemm_job_init(GemmJob& j);
void dw_job_init(DwJob& j);
// enqueue `n` jobs (n <= HUAL_MAX_JOBS) as ONE launch on `stream`
int launch_gemm(const GemmJob* jobs, int n, const DropCfg& drop, hipStream_t stream);
// enqueue `n` gradient jobs (any n; split into launches of HUAL_MAX_DW_JOBS); rows_per_block tunes split-M
// table != null: jobs are read from that device-resident table (any job count, one launch); write_table = false skips
// filling it (the caller vouches that it still holds exactly these jobs)
// words of a device job table for n jobs: the descriptors followed by 3 (n + 1) integers (plan of the balanced launch)
inline size_t dw_table_words(size_t n) { return (n * sizeof(DwJob) + 3 * (n + 1) * sizeof(int) + 3) / 4; }
int launch_dw(const DwJob* jobs, int n, const DropCfg& drop, int rows_per_block, hipStream_t stream,
              DwJob* table = nullptr, bool write_table = true, int balanced_blocks = 0);   // table: device buffer of n entries -> all jobs in ONE launch

}  // namespace hual
