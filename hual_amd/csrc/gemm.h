// Job-table dense kernels.  Every dense contraction of the SeqPAN graph that is not inside a fused row-local kernel
// (convblock.h, dablock.h) - the conv1d(kernel_size=1) layers of /root/reference/models/layers.py:20-29 and all
// weight gradients - goes through these launches.
#pragma once
#include "common.h"
#include "embed_args.h"

namespace hual {

#define HUAL_MAX_PIECES 4

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

// Pre-split weight images (bf16x3.h "f16x3").  launch_pack_weights writes, for n dense weights [K,128] at float offsets offs[] of
// P, the forward LDS image at fwd + 4*off (feature-load kernel) and the T / N images of the register-resident weights (tilecore.h)
// at timg / nimg + boffs[] (ceil(K/128) blocks of 64 KB); any destination may be null, `needs` selects per weight.
#define HUAL_PACK_BLOCK_BYTES (128 * 512)
// `extra` (optional): work of the step's prologue that rides in the same launch as one more row of workgroups - the row
// masks of model.py:31-32 (+ the loss accumulators cleared) and, when zero_ptr is set, the flat gradient buffer zeroed
struct PackExtra {
  const int32_t* lens; const int32_t* word_ids; float* rowmask; float* loss_acc; int B, T, L;
  float* zero_ptr; size_t zero_n;      // zero_n floats (multiple of 4), 16-byte aligned
  const float* E; float lambda; float* dE_ortho;   // optional: label_emb [4,128] -> loss_acc[LA_ORTHO] and lambda * d ortho / dE (ortho.h)
  // optional: the embedding gather of the text encoder (embed_gather.h) as further rows of workgroups, and the packed char-CNN
  // filter bank Wall [4 CP, 128] (embed.hip) as one more pack job whose elements come from the four filters: its forward
  // image goes to fwd + wall_off * 4, the image of its transpose to bwd + wall_boff
  int gather_tasks; int gather_rows; EmbedArgs emb; DropCfg drop;
  int wall_K; uint32_t wall_off, wall_boff;
  // optional: one word per workgroup of the job rows (novf of them: pack_ovf_words): 1 when a weight of its 16 K rows has a
  // magnitude the scaled fp16 image cannot hold (|w| >= HUAL_F16_WMAX, or NaN) - every word is written, no zeroing needed;
  // the loss launch turns any set word into a NaN loss (heads.h LossTailArgs)
  uint32_t* ovf; int novf;
};
#define HUAL_MAX_PACK 96                 // jobs of one pack launch
#define HUAL_PACK_GX 64                  // workgroups per row of its grid
// words the pack launch writes for jobs of these K: its job rows hold ceil(K / 128) * 8 workgroups per job, back to back
inline int pack_ovf_words(const int* Ks, int n) {
  int nblk = 0;
  for (int i = 0; i < n; ++i) nblk += ((Ks[i] + 127) & ~127) / 16;
  return (nblk + HUAL_PACK_GX - 1) / HUAL_PACK_GX * HUAL_PACK_GX;
}
// timg / nimg (optional): images for the register-resident weights of the T-form kernels (tilecore.h), blocks of 64 KB at boffs[]
// needs (optional, one byte per job): which of the images a weight is wanted in - the forward / transposed images of the LDS-DMA
// kernels (F, B), the T / N images; default all that have a destination
enum { HUAL_PACK_F = 1, HUAL_PACK_T = 4, HUAL_PACK_N = 8 };
int launch_pack_weights(const uint32_t* offs, const int* Ks, const uint32_t* boffs, int n, const float* P, char* fwd,
                        hipStream_t stream, const PackExtra* extra = nullptr, char* timg = nullptr, char* nimg = nullptr,
                        const uint8_t* needs = nullptr);
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
void dw_job_init(DwJob& j);
// words of a device job table for n jobs: the descriptors followed by 3 (n + 1) integers (plan of the balanced launch)
inline size_t dw_table_words(size_t n) { return (n * sizeof(DwJob) + 3 * (n + 1) * sizeof(int) + 3) / 4; }
// enqueue `n` gradient jobs (any n) as ONE persistent launch that reads them from the device-resident `table`;
// write_table = false skips filling it (the caller vouches that it still holds exactly these jobs); blocks = 0: one per CU
int launch_dw(const DwJob* jobs, int n, const DropCfg& drop, hipStream_t stream, DwJob* table, bool write_table = true,
              int blocks = 0);

}  // namespace hual
