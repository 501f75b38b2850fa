// Dense kernels outside the fused row-local kernels and the multi-step kernel (see gemm.h):
//   feature_ksplit_kernel   the feature-load phase (video_conv1d + query_conv1d) when the weight quarters fit LDS
//   pack_weights_kernel     once per step: pre-split images of every dense weight (+ the step prologue: masks, gradient zeroing,
//                           orthogonality term, embedding gather)
//   dw_f16_balanced_kernel every weight / bias gradient of a step in one persistent launch
#include "gemm.h"
#include <type_traits>
#include "philox.h"
#include "bf16x3.h"
#include "tilecore.h"
#include "prof.h"
#include "ortho.h"
#include "embed_gather.h"
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

namespace hual {

void dw_job_init(DwJob& j) {
  ::memset((void*)&j, 0, sizeof(j));
  j.a_drop_site = -1;
}

}  // namespace hual

using namespace hual;

__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

#define GL_KS 64                         // K rows per stage of a weight image
#define GB_TILE (GL_KS * 256)            // bytes of one [64][128 x 16 bit] tile
#define GB_STAGE (2 * GB_TILE)           // hi + lo

// ------------------------------------------------------------------------------------------------------
// Feature-load kernel: partial products of video_conv1d (model.py:47-48), Y_q[rows, 128] = dropout(video)[rows, Kq] . W[Kq, :]
// for the K-quarter q = blockIdx.y.  The deep-K launch above is bound by what one CU pulls in (the whole 512 KB weight
// image next to 128 KB of clip features) and both column-half waves draw the same dropout decisions.  Here a block owns
// 128 rows x ONE quarter of K: its quarter of the weight image (KS <= 256 rows = 128 KB) is DMA'd into LDS once and stays
// there, every clip-feature element is loaded, dropped and split exactly once (a wave owns 16 rows x all 128 columns), and
// the four partial sums go to a [4][rows][128] slab that the layer-norm launch behind it adds up (ln_fwd_kernel, `part`).
// Per CU: 128 KB of features + 128 KB of weights in, 64 KB out - against 640 KB in for the same rows before.
#define FK_ROWS 128
__global__ __launch_bounds__(512) void feature_ksplit_kernel(FkBatch batch, DropCfg drop) {
  extern __shared__ float lds[];     // KS/64 stages of {hi tile, lo tile}
  char* ldsb = reinterpret_cast<char*>(lds);
  const FkJob& job = batch.j[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int q = blockIdx.y;
  const int M = job.M, K = job.K, KS = job.KS;
  if ((int)blockIdx.x * FK_ROWS >= M) return;          // block-uniform
  const int rowbase = blockIdx.x * FK_ROWS + wave * 16;
  const int arow = min(rowbase + j, M - 1);
  const bool adrop = job.drop_site >= 0 && drop.enabled;
  const uint32_t adrow = job.drop_row0 + (uint32_t)arow;
  const DropRegs drk = drop_load(drop);
  const int nst = KS / 64;
  // the block's weight quarter: rows q*KS .. of the forward image (clamped to the last real row: the operand is zero
  // there), 32 one-KB pieces per stage
  {
    const char* img = reinterpret_cast<const char*>(job.Wimg);
    const int chp = lane & 15, rr = lane >> 4;
    for (int pc = wave; pc < 32 * nst; pc += 8) {
      const int st = pc >> 5, pl = pc & 31;
      const int r = 4 * (pl & 15) + rr;
      const int ch = chp ^ (((r & 3) << 2) | ((r >> 2) & 3));
      const int kk = min(q * KS + 64 * st + r, K - 1);
      const char* src = img + (size_t)kk * 512 + (pl >> 4) * 256 + 16 * ch;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(ldsb + st * GB_STAGE + pl * 1024), 16, 0, 0);
    }
  }
  f32x4 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* Ap = job.A + (size_t)arow * job.lda;
  const int nks = KS / 32;
  const uint16_t* Ap16 = reinterpret_cast<const uint16_t*>(job.A) + (size_t)arow * job.lda;
  const bool a16 = job.a_bf16 != 0;
  auto a_fetch = [&](int ks, float4& x0, float4& x1) {
    const int kk = q * KS + 32 * ks + 8 * g;
    if (a16) {      // bfloat16 features: 16 bytes = the lane's 8 values; they travel RAW in x0 and are widened where they are used
      // (widening here would wait for the load just issued: a memory round trip per k-step - the bfloat16 feed was no faster)
      const uint4 raw = *reinterpret_cast<const uint4*>(Ap16 + min(kk, K - 8));
      x0 = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w));
      return;
    }
    // K is a multiple of 8: a lane's 8 values are in or out together.  Unconditional loads on a clamped column (a load behind
    // the lane-dependent branch is waited for inside it), selected to zero beyond K at the top of their k-step
    // (the select happens where the values are used: at the load it would be a wait for the load just issued)
    const int kc = min(kk, K - 8);
    x0 = ld4(Ap + kc);
    x1 = ld4(Ap + kc + 4);
  };
  float4 c0, c1, n0 = f4zero(), n1 = f4zero();
  a_fetch(0, c0, c1);
  if (nks > 1) a_fetch(1, n0, n1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int tq = (lane >> 2) & 3, tp = lane & 3;
  // A16 (bfloat16 features): 8 significant bits x the dropout scale 1.25 x a power-of-two row scale fit fp16's 11 - the residual
  // plane is identically zero, so its split and the third MFMA pass are not issued (compile time: the loop body carries no test)
  auto kloop = [&](auto a16c) {
  constexpr bool A16 = decltype(a16c)::value;      // bfloat16 features AND a dropout scale of at most 4 significant bits (a_bf16 == 2)
  for (int ks = 0; ks < nks; ++ks) {
    float4 f0 = f4zero(), f1 = f4zero();
    if (ks + 2 < nks) a_fetch(ks + 2, f0, f1);       // two k-steps ahead
    if (!(q * KS + 32 * ks + 8 * g < K)) { c0 = f4zero(); c1 = f4zero(); }      // beyond K: zeros (raw bf16 zeros widen to zeros)
    if (a16) {                                       // widen this k-step's raw bfloat16 pairs (exact)
      const uint32_t w0 = __float_as_uint(c0.x), w1 = __float_as_uint(c0.y), w2 = __float_as_uint(c0.z), w3 = __float_as_uint(c0.w);
      c0 = make_float4(__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u), __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u));
      c1 = make_float4(__uint_as_float(w2 << 16), __uint_as_float(w2 & 0xffff0000u), __uint_as_float(w3 << 16), __uint_as_float(w3 & 0xffff0000u));
    }
    if (adrop) {
      // the lane's 8 consecutive features are ONE call of the 16-bit decision scheme (tilecore.h drop_bits8_r, oracle/philox.py
      // mask16); the keep byte goes to the bit plane the weight-gradient job reads (DwJob::a_keep)
      const uint32_t c8 = (uint32_t)((q * KS + 32 * ks + 8 * g) >> 3);
      const uint32_t b = drop_bits8_r(drk, (uint32_t)job.drop_site, adrow, c8);
      c0 = f4mul(c0, mask_from_bits4(b & 15u, drop.scale));
      c1 = f4mul(c1, mask_from_bits4(b >> 4, drop.scale));
      if (job.keep_out && rowbase + j < M && (int)(8u * c8) < K) job.keep_out[(size_t)arow * job.ld_keep + c8] = (uint8_t)b;
    }
    // f16x3 (bf16x3.h): the row scale is taken per 32-deep k-step (the features of a row arrive over the whole loop)
    float rmax = fmaxf(f4absmax(c0), f4absmax(c1));
    rmax = fmaxf(rmax, __shfl_xor(rmax, 16));
    rmax = fmaxf(rmax, __shfl_xor(rmax, 32));
    float inv;
    const float sc = f16_row_scale(rmax, inv);
    uint2 h0, l0 = make_uint2(0u, 0u), h1, l1 = make_uint2(0u, 0u);
    if (A16) {
      const float4 s0 = f4scale1(c0, sc), s1 = f4scale1(c1, sc);
      h0 = make_uint2(f16_pack2(s0.x, s0.y), f16_pack2(s0.z, s0.w));
      h1 = make_uint2(f16_pack2(s1.x, s1.y), f16_pack2(s1.z, s1.w));
    } else {
      f16_split4(f4scale1(c0, sc), h0, l0);
      f16_split4(f4scale1(c1, sc), h1, l1);
    }
    const f16x8 ah = __builtin_bit_cast(f16x8, (u32x4){h0.x, h0.y, h1.x, h1.y});
    const f16x8 al = __builtin_bit_cast(f16x8, (u32x4){l0.x, l0.y, l1.x, l1.y});
    const char* hi = ldsb + (ks >> 1) * GB_STAGE;
    const int r0 = 32 * (ks & 1) + 8 * g + tq, r1 = r0 + 4;
    float ir[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ir[r] = __shfl(inv, 4 * g + r);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int chunk = 2 * t + (tp >> 1);
      const int o0 = tile256_off(r0, chunk) + 8 * (tp & 1), o1 = tile256_off(r1, chunk) + 8 * (tp & 1);
      const f16x8 wh = join_tr_f16(lds_read_tr16(hi, o0), lds_read_tr16(hi, o1));
      const f16x8 wl = join_tr_f16(lds_read_tr16(hi + GB_TILE, o0), lds_read_tr16(hi + GB_TILE, o1));
      f32x4 p = (f32x4){0.f, 0.f, 0.f, 0.f};
      p = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, p, 0, 0, 0);
      p = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl, p, 0, 0, 0);
      if (!A16) p = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh, p, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = fmaf(p[r], ir[r], acc[t][r]);
    }
    c0 = n0; c1 = n1; n0 = f0; n1 = f1;
  }
  };
  if (job.a_bf16 == 2) kloop(std::true_type{}); else kloop(std::false_type{});
  // accumulator tile t, lane (j, g), register r = row 4g + r, column 64 (t>>2) + 4j + (t&3)
  float* out = job.part + (size_t)q * job.part_stride;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = rowbase + 4 * g + r;
    if (row < M) {
      st4(out + (size_t)row * 128 + 4 * j, make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]));
      st4(out + (size_t)row * 128 + 64 + 4 * j, make_float4(acc[4][r], acc[5][r], acc[6][r], acc[7][r]));
    }
  }
}

// Pre-split weight images for gemm_bf16_kernel, made once per step (the weights are constant within a step).
// For every dense weight W [K,128] at float offset `off` of the flat parameter buffer:
//   forward image  at fwd + 4*off:  row k (512 B) = fp16 hi of W[k][perm(s)], s = 0..127 | the same for the residuals (LDS-DMA
//                                   operand of the feature-load kernel)
//   T / N images   at timg / nimg + boff: one 64 KB block per 128 contraction indices, fragment-major (tilecore.h tf_img_off): the
//                                   register-resident weights of every other dense kernel, forward (W^T) and dX (W)
// perm(s) = 64 (s>>6) + 4 (s&15) + ((s>>4)&3): stored column 16 t + i of a 64-column half is original column 4 i + t.
struct PackJob { uint32_t off; int K; uint32_t boff; uint32_t need; };      // need: HUAL_PACK_* images wanted of this weight
// start[j] = first 16-row block of job j in the packed block order (start[njobs] = total): the job rows of the launch hold the
// blocks of all jobs back to back (a [K,128] weight has ceil(K / 128) * 8 of them) instead of one padded row per job
struct PackBatch { PackJob j[HUAL_MAX_PACK]; uint16_t start[HUAL_MAX_PACK + 1]; };
__device__ __forceinline__ int pack_perm(int s) { return 64 * (s >> 6) + 4 * (s & 15) + ((s >> 4) & 3); }
__global__ __launch_bounds__(256) void pack_weights_kernel(PackBatch b, const float* P, char* fwd, char* timg, char* nimg, int njobs, int jrows, PackExtra ex) {
  __shared__ float tile[16][129];
  if ((int)blockIdx.y > jrows) {       // further rows: the embedding gather of the text encoder, one task per thread
    const int gid = (((int)blockIdx.y - jrows - 1) * (int)gridDim.x + (int)blockIdx.x) * 256 + (int)threadIdx.x;
    if (gid < ex.gather_tasks)
      embed_gather_task(ex.emb, ex.drop, ex.gather_rows, ex.wall_K / 4, ex.gather_tasks - ex.wall_K - NALL, gid);
    return;
  }
  if ((int)blockIdx.y == jrows) {      // the extra row: masks, loss accumulators, gradient zeroing (grid-stride)
    const int nt = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
    const int Nv = ex.B * ex.T, Nq = ex.B * ex.L;
    if (t0 < 8) ex.loss_acc[t0] = 0.f;
    for (int i = t0; i < Nv + Nq; i += nt)
      ex.rowmask[i] = i < Nv ? ((i % ex.T) < ex.lens[i / ex.T] ? 1.0f : 0.0f)       // tf.sequence_mask, model.py:31
                             : (ex.word_ids[i - Nv] != 0 ? 1.0f : 0.0f);            // model.py:32
    if (ex.zero_ptr)
      for (size_t i = t0; i < ex.zero_n / 4; i += nt) reinterpret_cast<float4*>(ex.zero_ptr)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ex.E && blockIdx.x == 0) {      // orthogonality term of the label embeddings (parameters only, ortho.h)
      __shared__ float osm[20];
      __syncthreads();                  // behind the zeroing of the loss accumulators above
      ortho_body(ex.E, ex.loss_acc + 2 /* LA_ORTHO */, ex.lambda, ex.dE_ortho, osm);
    }
    return;
  }
  const int lb = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;      // packed block index
  uint32_t* ovf = ex.ovf ? ex.ovf + lb : nullptr;
  if (lb >= (int)b.start[njobs]) {                   // block-uniform: the tail of the last job row
    if (ovf && threadIdx.x == 0) *ovf = 0u;
    return;
  }
  int jlo = 0, jhi = njobs;                          // job of the block: start[jlo] <= lb < start[jlo + 1]
  while (jhi - jlo > 1) { const int mid = (jlo + jhi) >> 1; if (lb >= (int)b.start[mid]) jlo = mid; else jhi = mid; }
  const PackJob job = b.j[jlo];
  const int k0 = (lb - (int)b.start[jlo]) * 16;
  const float* W = P + job.off;
  const bool virt = ex.wall_K > 0 && job.off == ex.wall_off;      // the char-CNN filter bank: elements from the four filters
  bool bad = false;
  for (int idx = threadIdx.x; idx < 16 * 128; idx += 256) {
    const int r = idx >> 7, n = idx & 127;
    float v = 0.f;
    if ((k0 + r) < job.K) v = virt ? wall_value(ex.emb.filt, ex.emb.char_dim, ex.wall_K / 4, k0 + r, n) : W[(size_t)(k0 + r) * 128 + n];
    bad |= !(fabsf(v) < HUAL_F16_WMAX);              // (NaN included)
    tile[r][n] = v;
  }
  const int anybad = __syncthreads_or(bad ? 1 : 0);
  if (ovf && threadIdx.x == 0) *ovf = anybad ? 1u : 0u;
  // forward image: 16 rows x 64 column pairs
  if (fwd && (job.need & HUAL_PACK_F)) {
    char* img = fwd + (size_t)job.off * 4;
    for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
      const int r = idx >> 6, sp = idx & 63;
      if (k0 + r < job.K) {
        uint32_t hi, lo;
        f16_split_pair(tile[r][pack_perm(2 * sp)] * HUAL_F16_WSCALE, tile[r][pack_perm(2 * sp + 1)] * HUAL_F16_WSCALE, hi, lo);
        *reinterpret_cast<uint32_t*>(img + (size_t)(k0 + r) * 512 + 4 * sp) = hi;
        *reinterpret_cast<uint32_t*>(img + (size_t)(k0 + r) * 512 + 256 + 4 * sp) = lo;
      }
    }
  }
  // register-resident weights (tilecore.h "T-form"): one 64 KB block per 128 K rows at boff, element [column][index] at tf_img_off
  if (timg && (job.need & HUAL_PACK_T)) {        // T image: W^T - column n, index k
    char* img = timg + job.boff + (size_t)(k0 >> 7) * TF_BLOCK;
    const int kb = k0 & 127;
    for (int idx = threadIdx.x; idx < 128 * 8; idx += 256) {
      const int n = idx >> 3, pr = idx & 7;             // k pair (k0 + 2 pr, + 1) of column n
      uint32_t hi, lo;
      f16_split_pair(tile[2 * pr][n] * HUAL_F16_WSCALE, tile[2 * pr + 1][n] * HUAL_F16_WSCALE, hi, lo);
      *reinterpret_cast<uint32_t*>(img + tf_img_off(n, kb + 2 * pr)) = hi;
      *reinterpret_cast<uint32_t*>(img + tf_img_off(n, kb + 2 * pr) + TF_LO_OFF) = lo;
    }
  }
  if (nimg && (job.need & HUAL_PACK_N)) {        // N image: W itself - "column" k, index n
    char* img = nimg + job.boff + (size_t)(k0 >> 7) * TF_BLOCK;
    const int kb = k0 & 127;
    for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
      const int r = idx >> 6, np = idx & 63;            // n pair (2 np, 2 np + 1) of row k0 + r
      uint32_t hi, lo;
      f16_split_pair(tile[r][2 * np] * HUAL_F16_WSCALE, tile[r][2 * np + 1] * HUAL_F16_WSCALE, hi, lo);
      *reinterpret_cast<uint32_t*>(img + tf_img_off(kb + r, 2 * np)) = hi;
      *reinterpret_cast<uint32_t*>(img + tf_img_off(kb + r, 2 * np) + TF_LO_OFF) = lo;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// dW / db: job table of the persistent weight-gradient launch.
// copies a slice of job descriptors (passed by value, so graph-capture safe) into the device-resident job table
// and, behind the n descriptors, what the balanced launch needs per job: tiles in front of it, cost in front of it and
// the cost of one of its tiles (dw_plan)
struct DwPlanPart { int tiles[HUAL_MAX_DW_JOBS + 1], cost[HUAL_MAX_DW_JOBS + 1], w[HUAL_MAX_DW_JOBS + 1]; };
struct DwPlan { int* tiles; int* cost; int* w; };      // (n + 1) entries each
__device__ __host__ inline DwPlan dw_plan(const DwJob* table, int n) {
  int* base = const_cast<int*>(reinterpret_cast<const int*>(table + n));
  return DwPlan{base, base + (n + 1), base + 2 * (n + 1)};
}
__global__ void dw_table_write_kernel(DwBatch part, DwPlanPart pre, DwJob* table, int base, int cnt, int n) {
  const int t = threadIdx.x;
  if (t < cnt) table[base + t] = part.j[t];
  if (t <= cnt) {
    const DwPlan pl = dw_plan(table, n);
    pl.tiles[base + t] = pre.tiles[t];
    pl.cost[base + t] = pre.cost[t];
    pl.w[base + t] = pre.w[t];
  }
}

// ------------------------------------------------------------------------------------------------------
// dw_f16_balanced_kernel (rounds 1-4: dw_bf16_*): tiling (128 x 128 gradient tile per unit over a run of rows, one 64 x 32 block per
// wave) and atomics epilogue of the first fp32 kernel, with the products on the 16-bit matrix cores as three passes of split operands
// (bf16x3.h) - since round 5 fp16 pairs (22-bit operands; the scales are described in dw_f16_segment), before that bf16 pairs.  The
// reduction index of dW = A^T.dY is the ROW index m, which is the strided direction of both operands in memory: the 64-row tiles are
// staged row-major as 16-bit hi and lo planes (256-byte rows, XOR swizzled) and read back with ds_read_b64_tr_b16, the LDS transpose
// read, so each lane receives 8 consecutive rows of its column.  The kernel is bound by streaming the operands from HBM (every job
// reads its A and dY once).
#define DWB_PLANE (64 * 256)             // bytes of one [DWB_TM][128] 16-bit plane
// (the A side had the fixed 2^4 of the other kernels until the end of round 5: the PRODUCT pieces of the context-query dense layer -
//  x * c2q, x * q2c - and the unnormalised inputs of the predictor's hidden layers passed 4094 in a fast-learning toy run every ~20th
//  training run of tests/test_gpu_runner.py: NaN gradients for tensors whose forward was finite.  Now a running scale like dY's.)
// one segment: rows [m_lo, m_hi) of k-block kb of piece p of a job -> atomics into its 128 x 128 gradient tile.
// 512 threads: wave (kq, nq) owns the 64 x 32 block of gradient rows 64kq.., columns 32nq.. (two 32x32 accumulators,
// <= 128 registers per lane, so two workgroups share a CU).  The launch is bound by how many bytes a CU keeps in flight
// (one workgroup per CU runs 1.7x longer than two), hence two register stages: the loads of tile t+2 are issued before
// the products of tile t and consumed (split, stored to LDS) at the end of iteration t+1.
#define DWB_THREADS 512
#define DWB_TM 64             // rows per LDS tile: one workgroup barrier per 64 rows
#define DWB_RU (DWB_TM / 16)  // rows a thread stages per tile
// MODE: DWB_PLAIN fp32 A without a prologue (all but a handful of jobs); DWB_PROD product prologue A[p] * A2[p] (its
// second operand is staged like the first: DEPTH 2 keeps the registers below 256); DWB_DROP dropout prologue (keep bits
// from the forward, or the Philox rounds) and / or bfloat16 A
#define DWB_PLAIN 0
#define DWB_PROD 1
#define DWB_DROP 2
template <int MODE, int DEPTH>
__device__ __forceinline__ void dw_f16_segment(const DwJob& job, const int p, const int kb, const int m_lo, const int m_hi,
                                                const DropCfg& drop, char* ldsb, float4 (*bred)[32]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int kq = wave >> 2, nq = wave & 3;
  const int k0 = kb * 128;
  const int kw = job.kw[p];
  const char* Ap = reinterpret_cast<const char*>(job.A[p]);
  const char* A2p = MODE == DWB_PROD ? reinterpret_cast<const char*>(job.A2[p]) : nullptr;
  const int lda = job.lda[p], lda2 = job.lda2[p];
  const char* Yp = reinterpret_cast<const char*>(job.dY);
  const int ldy = job.ldy;
  constexpr bool PLAIN = MODE == DWB_PLAIN;
  const bool adrop = MODE == DWB_DROP && job.a_drop_site >= 0 && drop.enabled;
  const uint32_t asite = (uint32_t)job.a_drop_site;
  const uint32_t arow0 = job.a_drop_row0;
  const bool dob = (job.db != nullptr) && p == 0 && kb == 0;
  const bool abf = MODE == DWB_DROP && job.a_bf16 != 0;

  f32x16 acc[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float4 bsum = f4zero();
  // Round 5: fp16 pairs (22-bit operands) instead of bf16 pairs.  dY is a GRADIENT and the contraction runs over all rows of the
  // segment, so its scale has to hold for the whole segment: a RUNNING power-of-two
  // scale s_y - set from the first tile's largest |dY| with three binades of headroom, and whenever a later tile would exceed 2^14 the
  // accumulators are multiplied down with it (32 multiplications per wave, rare: gradient magnitudes vary slowly along the rows).  A
  // tile's maximum crosses the workgroup through eight LDS slots one barrier ahead of its split (the barriers were there).
  // A gets the same treatment (s_a): its magnitude is NOT bounded by construction - products of two activations, unnormalised block
  // outputs - and an operand beyond fp16's range turns a finite forward pass into NaN gradients.
  float* mxs = reinterpret_cast<float*>(&bred[0][0]);      // [2][16] slots: |dY| maxima, then |A| maxima (bred is free until the bias reduction at the end)
  float s_y = 0.f, s_y_inv = 0.f;                           // (wave-uniform)
  float s_a = 0.f, s_a_inv = 0.f;

  // staging: thread -> rows (tid>>5) + 16u, columns 4*(tid&31)..+3 of both tiles; prologues (bf16 widening, product,
  // dropout) are applied at the store, so that a load is only waited for one iteration after its issue.  Addresses are
  // a uniform base per tile plus a 32-bit byte offset per thread (one register each for A and dY).
  const int srow = threadIdx.x >> 5, c4 = threadIdx.x & 31;
  const int kcol = k0 + 4 * c4;
  const bool kin = kcol < kw;
  const uint32_t esz = abf ? 2u : 4u;
  const uint32_t aoff = ((uint32_t)srow * (uint32_t)lda + (uint32_t)kcol) * esz;
  const uint32_t yoff = ((uint32_t)srow * (uint32_t)ldy + 4u * (uint32_t)c4) * 4u;
  struct Stage { float4 a[DWB_RU], y[DWB_RU], a2[MODE == DWB_PROD ? DWB_RU : 1]; uint32_t keep[DWB_RU]; };
  const uint32_t a2off = ((uint32_t)srow * (uint32_t)lda2 + (uint32_t)kcol) * 4u;
  const uint8_t* keepp = MODE == DWB_DROP ? job.a_keep : nullptr;
  const uint32_t koff = (uint32_t)srow * (uint32_t)job.ld_keep + (uint32_t)(kcol >> 3);      // bit plane: one byte per 8 columns
  Stage st[DEPTH - 1];      // tile i in LDS, tiles i+1 .. i+DEPTH-1 in (or on their way to) registers
  auto stage_load = [&](int mt, Stage& st) {
    const char* Ab = Ap + (size_t)mt * lda * esz;
    const char* Yb = Yp + (size_t)mt * ldy * 4;
#pragma unroll
    for (int u = 0; u < DWB_RU; ++u) {
      const int m = mt + srow + 16 * u;
      float4 a = f4zero(), y = f4zero();
      if (m < m_hi) {
        if (kin) {
          if (abf) {
            const uint2 raw = ld2_global(Ab + 16u * u * (uint32_t)lda * 2u + aoff);
            a.x = __uint_as_float(raw.x);
            a.y = __uint_as_float(raw.y);
          } else {
            a = ld4_global(Ab + 16u * u * (uint32_t)lda * 4u + aoff);
          }
          if (MODE == DWB_PROD && A2p) st.a2[u] = ld4_global(A2p + (size_t)mt * lda2 * 4 + 16u * u * (uint32_t)lda2 * 4u + a2off);
        }
        y = ld4_global(Yb + 16u * u * (uint32_t)ldy * 4u + yoff);
        if (MODE == DWB_DROP && keepp && kin) st.keep[u] = ld1_global(keepp + (size_t)mt * job.ld_keep + 16u * u * (uint32_t)job.ld_keep + koff);
      }
      st.a[u] = a;
      st.y[u] = y;
    }
  };
  auto stage_store = [&](int buf, int mt, const Stage& st) {
    char* base = ldsb + buf * 4 * DWB_PLANE;
#pragma unroll
    for (int u = 0; u < DWB_RU; ++u) {
      const int row = srow + 16 * u;
      const int off = tile256_off(row, c4 >> 1) + 8 * (c4 & 1);
      float4 a = st.a[u];
      if (!PLAIN) {
        const int m = mt + row;
        const bool live = m < m_hi && kin;
        if (abf) {
          const uint32_t r0 = __float_as_uint(a.x), r1 = __float_as_uint(a.y);
          a = make_float4(__uint_as_float(r0 << 16), __uint_as_float(r0 & 0xffff0000u), __uint_as_float(r1 << 16),
                          __uint_as_float(r1 & 0xffff0000u));
        }
        if (MODE == DWB_PROD && A2p && live) a = f4mul(a, st.a2[u]);
        if (adrop && live) {
          a = f4mul(a, mask_from_bits4((st.keep[u] >> (4 * (c4 & 1))) & 15u, drop.scale));
        }
      }
      uint2 hi, lo;
      f16_split4_s(a, s_a, hi, lo);                           // (scale folded into the mixed-precision FMA: bf16x3.h)
      *reinterpret_cast<uint2*>(base + off) = hi;
      *reinterpret_cast<uint2*>(base + DWB_PLANE + off) = lo;
      f16_split4_s(st.y[u], s_y, hi, lo);
      *reinterpret_cast<uint2*>(base + 2 * DWB_PLANE + off) = hi;
      *reinterpret_cast<uint2*>(base + 3 * DWB_PLANE + off) = lo;
      bsum = f4add(bsum, st.y[u]);
    }
  };
  // transposed-read addressing: 16-lane group g' = lane>>4 takes the 4-row x 16-column block of columns
  // colbase + 16*(g'&1), rows 16*ks + 8*(g'>>1) + 4*rr; lane 4q+pp of the group supplies row q, columns 4pp..4pp+3
  const int gq = (lane >> 2) & 3, gp = lane & 3, ghalf = (lane >> 4) & 1, gh = lane >> 5;
  auto tr_off = [&](int colbase, int ks, int rr) {
    const int row = 16 * ks + 8 * gh + 4 * rr + gq;
    const int ch = ((colbase + 16 * ghalf) >> 3) + (gp >> 1);
    return tile256_off(row, ch) + 8 * (gp & 1);
  };
  auto products = [&](int buf) {
    const char* base = ldsb + buf * 4 * DWB_PLANE;
#pragma unroll
    for (int ks = 0; ks < DWB_TM / 16; ++ks) {
      const int y0 = tr_off(nq * 32, ks, 0), y1 = tr_off(nq * 32, ks, 1);
      const f16x8 yh = join_tr_f16(lds_read_tr16(base + 2 * DWB_PLANE, y0), lds_read_tr16(base + 2 * DWB_PLANE, y1));
      const f16x8 yl = join_tr_f16(lds_read_tr16(base + 3 * DWB_PLANE, y0), lds_read_tr16(base + 3 * DWB_PLANE, y1));
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int o0 = tr_off(kq * 64 + 32 * c, ks, 0), o1 = tr_off(kq * 64 + 32 * c, ks, 1);
        const f16x8 ah = join_tr_f16(lds_read_tr16(base, o0), lds_read_tr16(base, o1));
        const f16x8 al = join_tr_f16(lds_read_tr16(base + DWB_PLANE, o0), lds_read_tr16(base + DWB_PLANE, o1));
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, yh, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, yl, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, yh, acc[c], 0, 0, 0);
      }
    }
  };

  // largest |dY| of a staged tile: this wave's part into its slot of row `par` (published by the next workgroup barrier) ...
  auto put_max = [&](const Stage& st, int par) {
#ifdef DWB_EXP_FIXSCALE     // timing experiment: no maxima, a fixed dY scale (numerically wrong for small gradients)
    return;
#endif
    float m = 0.f, ma = 0.f;
#pragma unroll
    for (int u = 0; u < DWB_RU; ++u) {
      m = fmaxf(m, f4absmax(st.y[u]));
      float4 a = st.a[u];                      // the operand as stage_store will split it (an upper bound under dropout: x 1 / (1 - rate))
      if (abf) {
        const uint32_t r0 = __float_as_uint(a.x), r1 = __float_as_uint(a.y);
        a = make_float4(__uint_as_float(r0 << 16), __uint_as_float(r0 & 0xffff0000u), __uint_as_float(r1 << 16), __uint_as_float(r1 & 0xffff0000u));
      }
      if (MODE == DWB_PROD && A2p) a = f4mul(a, st.a2[u]);
      ma = fmaxf(ma, f4absmax(a));
    }
    if (adrop) ma *= drop.scale;
    m = fast_max32(m);
    m = fmaxf(m, lane_xor32_partner(m));
    ma = fast_max32(ma);
    ma = fmaxf(ma, lane_xor32_partner(ma));
    if (lane == 0) { mxs[par * 16 + wave] = m; mxs[par * 16 + 8 + wave] = ma; }
  };
  // ... and the scale for the tile whose maxima sit in row `par`: the first tile sets it, a later one only lowers it
  auto take_scale = [&](int par) {
#ifdef DWB_EXP_FIXSCALE
    s_y = 1024.0f; s_y_inv = 1.0f / 1024.0f; s_a = 16.0f; s_a_inv = 1.0f / 16.0f;
    return;
#endif
    auto wgmax = [&](const float* q) {
      const float4 m0 = *reinterpret_cast<const float4*>(q), m1 = *reinterpret_cast<const float4*>(q + 4);
      const float g = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
      return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, g)));
    };
    // one running scale: set by the first tile (its maximum x scale in [2^11, 2^12)), lowered - and the accumulators with it - when a later
    // tile would pass 2^14
    auto running = [&](float g, float& sc, float& sc_inv) {
      if (sc == 0.f || g * sc > 16384.0f) {      // (uniform)
        uint32_t eb = (__float_as_uint(g) >> 23) & 0xffu;
        eb = eb < 27u ? 27u : (eb > 240u ? 240u : eb);
        const float s_new = __uint_as_float((265u - eb) << 23);
        if (sc != 0.f) {
          const float f = s_new * sc_inv;                              // < 1, exact
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] *= f;
        }
        sc = s_new;
        sc_inv = __uint_as_float((eb - 11u) << 23);
      }
    };
    running(wgmax(mxs + par * 16), s_y, s_y_inv);
    running(wgmax(mxs + par * 16 + 8), s_a, s_a_inv);
  };
  // tile i sits in LDS buffer i&1 while the registers hold tiles i+1 .. i+DEPTH-1 (the last of them just issued)
#pragma unroll
  for (int q = 0; q < DEPTH - 1; ++q) stage_load(m_lo + q * DWB_TM, st[q]);     // rows at or beyond m_hi load nothing
  put_max(st[0], 0);
  __syncthreads();
  take_scale(0);
  stage_store(0, m_lo, st[0]);
#ifndef DWB_EXP_LATEMAX
  if (DEPTH == 4) put_max(st[1], 1);
#endif
  __syncthreads();
  int buf = 0, mt = m_lo, par = 1;
  // one tile: issue the loads of tile i+DEPTH-1 into `in` (the stage tile i left), multiply tile i, move tile i+1 from
  // `out` to the other buffer (its maxima were published one barrier ago: row `par`), leave the maxima of tile i+2 (`nxt`)
  auto step = [&](Stage& in, const Stage& out, const Stage& nxt) -> bool {
    stage_load(mt + (DEPTH - 1) * DWB_TM, in);
    products(buf);
    if (mt + DWB_TM >= m_hi) return true;      // block-uniform
#ifdef DWB_EXP_LATEMAX      // (measured: the maximum taken when the tile is due, on a barrier of its own: 149 us against 144 for the form below)
    put_max(out, par);
    __syncthreads();
    take_scale(par);
    stage_store(buf ^ 1, mt + DWB_TM, out);
#else
    if (DEPTH == 2) { put_max(out, par); __syncthreads(); }      // (one register stage: the tile has only just been requested - its own barrier)
    take_scale(par);
    stage_store(buf ^ 1, mt + DWB_TM, out);
    if (DEPTH == 4) put_max(nxt, par ^ 1);     // the maxima of tile i+2 ride on the barrier below: known when its split comes up
#endif
    __syncthreads();
    mt += DWB_TM;
    buf ^= 1;
    par ^= 1;
    return false;
  };
  static_assert(DEPTH == 2 || DEPTH == 4, "the rotations below are written out for one and three stages");
  if (DEPTH == 2) {
    for (;;)
      if (step(st[0], st[0], st[0])) break;
  } else {
    for (;;) {       // st[0] held tile i (now in LDS) and receives tile i+3; st[1] holds tile i+1, st[2] tile i+2
      if (step(st[0], st[1], st[2])) break;
      if (step(st[1], st[2], st[0])) break;
      if (step(st[2], st[0], st[1])) break;
    }
  }

  // each wave owns its block: no cross-wave reduction, straight to global atomics.  Accumulator register r of lane
  // (i, h) is gradient row 32c + (r&3) + 8*(r>>2) + 4h, column i of the block.
  float* dWp = job.dW[p];
  const int kbase = k0 + kq * 64, n = nq * 32 + i;
  const float flush_scale = s_y_inv * s_a_inv;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = kbase + 32 * c + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (k < kw) atomic_add_global(dWp + (size_t)k * job.ldw + n, acc[c][r] * flush_scale);
    }
  if (dob) {                                   // block-uniform
    bred[srow][c4] = bsum;
    __syncthreads();
    if (threadIdx.x < 128) {
      const int cc = threadIdx.x;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += reinterpret_cast<const float*>(&bred[k][cc >> 2])[cc & 3];
      atomic_add_global(job.db + cc, s);
    }
  }
}

// block-uniform choice of the variant (launch_dw rejects a product prologue together with dropout / bfloat16)
__device__ __forceinline__ void dw_f16_any_segment(const DwJob& job, const int p, const int kb, const int m_lo, const int m_hi,
                                                    const DropCfg& drop, char* ldsb, float4 (*bred)[32]) {
  if (job.A2[p]) dw_f16_segment<DWB_PROD, 2>(job, p, kb, m_lo, m_hi, drop, ldsb, bred);
  else if (job.a_bf16 || job.a_drop_site >= 0) dw_f16_segment<DWB_DROP, 4>(job, p, kb, m_lo, m_hi, drop, ldsb, bred);
  else dw_f16_segment<DWB_PLAIN, 4>(job, p, kb, m_lo, m_hi, drop, ldsb, bred);
}

// Balanced launch: the 64-row tiles of all (job, piece, k-block) units form one list (job-major; plan.tiles[j] = tiles in
// front of job j) that is cut into gridDim.x runs of equal COST, one per workgroup (one workgroup per CU); a workgroup
// walks its run segment by segment and flushes its accumulators with atomics whenever the unit changes.  The cost of a
// tile depends on its job's prologue (strided clip features + keep bytes, product operands), and every unit carries a
// fixed cost in front of its first tile (DW_UNIT_COST: the flush, the pipeline refill): plan.cost[j] = cost in front of
// job j, plan.w[j] = cost per tile.  256 workgroups x 58 jobs: ~340 flushes of 64 KB instead of ~1500 with a fixed row
// split (float atomics run at 1.3 TB/s chip-wide), no round quantisation, no tail.
#ifdef HUAL_STAMPS
// debug: per-workgroup clock stamps of the balanced launch (scripts/exp/dw_stamps.py)
#define DW_STAMP_SLOTS 16
__device__ unsigned long long g_dw_stamps[2048 * DW_STAMP_SLOTS];
extern "C" int hual_debug_dw_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dw_stamps), sizeof(unsigned long long) * (size_t)n);
}
#define DW_STAMP(i, v) do { if (threadIdx.x == 0 && blockIdx.x < 2048 && (i) < DW_STAMP_SLOTS) g_dw_stamps[blockIdx.x * DW_STAMP_SLOTS + (i)] = (v); } while (0)
#else
#define DW_STAMP(i, v) do { } while (0)
#endif
#define DW_UNIT_COST 72
__global__ __launch_bounds__(DWB_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void dw_f16_balanced_kernel(const DwJob* __restrict__ table, int n, DropCfg drop) {
  extern __shared__ float lds[];
  __shared__ float4 bred[16][32];
  int* s_job = reinterpret_cast<int*>(&bred[0][0]);     // (no further static array: the planes must stay 16-byte aligned)
  DW_STAMP(0, __builtin_readcyclecounter());
  const DwPlan pl = dw_plan(table, n);
  const int total_cost = pl.cost[n];
  // run boundaries in cost space -> global tile index
  int cb[2];
  cb[0] = (int)(((long)blockIdx.x * total_cost) / gridDim.x);
  cb[1] = (int)(((long)(blockIdx.x + 1) * total_cost) / gridDim.x);
  for (int j = threadIdx.x; j < n; j += DWB_THREADS)
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (pl.cost[j] <= cb[e] && cb[e] < pl.cost[j + 1]) s_job[e] = j;
  __syncthreads();
  int tb[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    if (cb[e] >= total_cost) { tb[e] = pl.tiles[n]; continue; }
    const int j = s_job[e];
    const int per_unit = (table[j].M + DWB_TM - 1) / DWB_TM;
    const int w = pl.w[j];
    const int ucost = DW_UNIT_COST + per_unit * w;
    const int local = cb[e] - pl.cost[j];
    const int unit = local / ucost;
    const int r = local - unit * ucost - DW_UNIT_COST;
    tb[e] = pl.tiles[j] + unit * per_unit + (r > 0 ? r / w : 0);
  }
  int t = tb[0];
  const int t_end = tb[1];
  int j = s_job[0];
  __syncthreads();                              // s_job lives in bred
  if (t >= t_end) return;                       // block-uniform
  const int* prefix = pl.tiles;
  DW_STAMP(1, __builtin_readcyclecounter());
  DW_STAMP(3, (unsigned long long)(t_end - t));
  DW_STAMP(5, (unsigned long long)j);
  int nseg = 0, nnp = 0;
  while (t < t_end) {
    const DwJob& job = table[j];
    const int per_unit = (job.M + DWB_TM - 1) / DWB_TM;
    const int local = t - prefix[j];
    int unit = local / per_unit;
    const int tile = local - unit * per_unit;
    const int cnt = min(per_unit - tile, t_end - t);
    int p = 0;
    for (p = 0; p < job.npieces; ++p) {
      const int nkb = (job.kw[p] + 127) >> 7;
      if (unit < nkb) break;
      unit -= nkb;
    }
    const bool plain = !job.a_bf16 && !job.A2[p] && job.a_drop_site < 0;      // (stamps)
    dw_f16_any_segment(job, p, unit, tile * DWB_TM, min((tile + cnt) * DWB_TM, job.M), drop, reinterpret_cast<char*>(lds), bred);
    t += cnt;
    if (t >= prefix[j + 1]) ++j;
    __syncthreads();                            // LDS planes and bred are reused by the next segment
    DW_STAMP(7 + nseg, __builtin_readcyclecounter());
    ++nseg;
    nnp += plain ? 0 : 1;
  }
  DW_STAMP(2, __builtin_readcyclecounter());
  DW_STAMP(4, (unsigned long long)nseg);
  DW_STAMP(6, (unsigned long long)nnp);
}

namespace hual {

int launch_pack_weights(const uint32_t* offs, const int* Ks, const uint32_t* boffs, int n, const float* P, char* fwd,
                        hipStream_t stream, const PackExtra* extra, char* timg, char* nimg, const uint8_t* needs) {
  HUAL_REQUIRE(!extra || (extra->lens && extra->word_ids && extra->rowmask && extra->loss_acc && (extra->zero_n % 4) == 0 &&
                          (reinterpret_cast<uintptr_t>(extra->zero_ptr) & 15) == 0), "pack: extra prologue work");
  PackExtra ex{};
  for (int base = 0; base < n; base += HUAL_MAX_PACK) {
    const int cnt = n - base < HUAL_MAX_PACK ? n - base : HUAL_MAX_PACK;
    PackBatch b;
    int maxK = 0, nblk = 0;
    double elems = 0.0;
    for (int i = 0; i < cnt; ++i) {
      b.start[i] = (uint16_t)nblk;
      nblk += ((Ks[base + i] + 127) & ~127) / 16;
      HUAL_REQUIRE(Ks[base + i] > 0 && (Ks[base + i] % 8) == 0, "pack: K must be a positive multiple of 8");
      b.j[i].off = offs[base + i]; b.j[i].K = Ks[base + i]; b.j[i].boff = boffs ? boffs[base + i] : 0;
      b.j[i].need = needs ? needs[base + i] : (HUAL_PACK_F | HUAL_PACK_T | HUAL_PACK_N);
      const int kp = (Ks[base + i] + 127) & ~127;
      maxK = kp > maxK ? kp : maxK;
      elems += (double)Ks[base + i] * 128;
    }
    b.start[cnt] = (uint16_t)nblk;
    HUAL_REQUIRE(nblk < 65536, "pack: too many blocks");
    const bool with_extra = extra && base == 0;
    if (with_extra) ex = *extra;
    const int gx = HUAL_PACK_GX;
    const int jrows = cdiv(nblk, gx);
    HUAL_REQUIRE(!(ex.ovf && n > HUAL_MAX_PACK), "pack: the overflow words cover one batch of jobs");
    HUAL_REQUIRE(!ex.ovf || ex.novf >= jrows * gx, "pack: overflow words");
    const int grows = (with_extra && ex.gather_tasks > 0) ? cdiv(ex.gather_tasks, gx * 256) : 0;      // rows of workgroups of the gather
    HUAL_LAUNCH(0.0, elems * (4.0 + (fwd ? 4.0 : 0.0) + (timg ? 4.0 : 0.0) + (nimg ? 4.0 : 0.0)), pack_weights_kernel, dim3(gx, jrows + (with_extra ? 1 + grows : 0)),
                dim3(256), 0, stream, b, P, fwd, timg, nimg, cnt, jrows, ex);
  }
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_feature_ksplit(const FkJob* jobs, int n, const DropCfg& drop, hipStream_t stream) {
  HUAL_REQUIRE(n >= 1 && n <= HUAL_MAX_FK_JOBS, "feature_ksplit: job count");
  FkBatch b;
  int maxM = 0, maxKS = 0;
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const FkJob& j = jobs[i];
    HUAL_REQUIRE(j.A && j.Wimg && j.part && j.M > 0, "feature_ksplit: null / empty");
    HUAL_REQUIRE(j.K > 0 && (j.K % 8) == 0 && (j.KS % 64) == 0 && j.KS >= 64 && j.KS <= 256 && 4 * j.KS >= j.K && (j.lda % 4) == 0,
                 "feature_ksplit: need K % 8 == 0 and a quarter size KS (multiple of 64, <= 256) with 4*KS >= K");
    b.j[i] = j;
    if (j.a_bf16) {
      // bfloat16 features (8 significant bits) x the dropout scale x a power-of-two row scale are EXACT in fp16 (11 bits) when the scale
      // has at most 4 significant bits (1.25 for rate 0.2, 2 for 0.5, 1 without dropout): the residual plane is zero and the kernel
      // skips it (a_bf16 = 2); any other rate keeps the full split
      int e = 0;
      const double m16 = frexp((double)((j.drop_site >= 0 && drop.enabled) ? drop.scale : 1.0f), &e) * 16.0;
      b.j[i].a_bf16 = (m16 == floor(m16)) ? 2 : 1;
    }
    maxM = j.M > maxM ? j.M : maxM;
    maxKS = j.KS > maxKS ? j.KS : maxKS;
    flops += 2.0 * j.M * (double)j.K * 128.0;
    // ALGORITHMIC bytes (T.(V+D).e per clip + weights): features + weights in, ONE [M,128] projection out.  The
    // four K-quarter partial slabs this kernel actually writes (4.M.128 floats, summed by the layer-norm launch behind it)
    // are implementation traffic: they show up in the measured PMC bytes, not here.
    bytes += (j.a_bf16 ? 2.0 : 4.0) * (double)j.M * j.K + 4.0 * ((double)j.K * 128 + (double)j.M * 128);
  }
  HUAL_DYN_LDS(feature_ksplit_kernel, 160 * 1024);
  const size_t lds = (size_t)(maxKS / 64) * GB_STAGE;
  HUAL_LAUNCH(flops, bytes, feature_ksplit_kernel, dim3(cdiv(maxM, FK_ROWS), 4, n), dim3(512), lds, stream, b, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

static int dw_check(const DwJob& j, int& kbs, double& flops, double& bytes) {
  HUAL_REQUIRE(j.M > 0 && j.N == 128, "launch_dw: N must be 128");
  HUAL_REQUIRE(j.npieces >= 1 && j.npieces <= HUAL_MAX_PIECES, "launch_dw: pieces");
  kbs = 0;
  double kt = 0.0;
  for (int p = 0; p < j.npieces; ++p) {
    HUAL_REQUIRE(j.kw[p] > 0 && (j.kw[p] % 16) == 0, "launch_dw: piece width must be a multiple of 16");
    HUAL_REQUIRE(j.A[p] && j.dW[p], "launch_dw: null operand");
    HUAL_REQUIRE((j.lda[p] % 4) == 0 && (j.ldy % 4) == 0, "launch_dw: leading dims must be multiples of 4");
    kbs += cdiv(j.kw[p], 128);
    kt += j.kw[p];
  }
  flops += 2.0 * j.M * kt * j.N;
  bytes += 4.0 * ((double)j.M * kt + (double)j.M * j.N + kt * j.N);
  return 0;
}

// All jobs run as ONE persistent launch from the device-resident job table `table` (dw_table_words(n) words; the
// descriptors are first written there by tiny kernels that carry them by value, unless write_table is false: the caller
// vouches that the table still holds exactly these jobs).  blocks = 0: one workgroup per CU.
int launch_dw(const DwJob* jobs, int n, const DropCfg& drop, hipStream_t stream, DwJob* table, bool write_table, int blocks) {
  HUAL_REQUIRE(table != nullptr && n >= 1, "launch_dw: job table");
  for (int i = 0; i < n; ++i)
    HUAL_REQUIRE(!jobs[i].a_bf16 || (jobs[i].npieces == 1 && !jobs[i].A2[0] && (jobs[i].lda[0] % 4) == 0),
                 "dw: a bfloat16 operand needs one piece, no product prologue");
  double flops = 0.0, bytes = 0.0;
  int tiles = 0, cost = 0;                     // running tile count / cost
  // cost of one 64-row tile by prologue (per-tile cycles of the workgroups: plain 6.5 k, product and strided clip
  // features + keep bytes 8-9 k; the launch time is flat from 44 to 52)
  static const int wts[3] = {32, 46, 46};
  for (int base = 0; base < n; base += HUAL_MAX_DW_JOBS) {
    const int cnt = n - base < HUAL_MAX_DW_JOBS ? n - base : HUAL_MAX_DW_JOBS;
    DwBatch b;
    DwPlanPart pre;
    for (int i = 0; i < cnt; ++i) {
      const DwJob& jb = jobs[base + i];
      int kbs;
      int rc = dw_check(jb, kbs, flops, bytes);
      if (rc) return rc;
      b.j[i] = jb;
      bool prod = false;
      for (int p = 0; p < jb.npieces; ++p) prod = prod || jb.A2[p] != nullptr;
      const int w = (jb.a_drop_site >= 0 || jb.a_bf16) ? wts[2] : prod ? wts[1] : wts[0];
      pre.tiles[i] = tiles;
      pre.cost[i] = cost;
      pre.w[i] = w;
      tiles += kbs * cdiv(jb.M, DWB_TM);
      cost += kbs * (DW_UNIT_COST + cdiv(jb.M, DWB_TM) * w);
    }
    for (int i = cnt; i <= HUAL_MAX_DW_JOBS; ++i) { pre.tiles[i] = tiles; pre.cost[i] = cost; pre.w[i] = 1; }
    if (write_table) HUAL_LAUNCH(0.0, 0.0, dw_table_write_kernel, dim3(1), dim3(64), 0, stream, b, pre, table, base, cnt, n);
  }
  int dev = 0, cus = 256;
  HUAL_CHECK_HIP(hipGetDevice(&dev));
  HUAL_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  int nblocks = blocks > 0 ? blocks : cus;
  if (nblocks > tiles) nblocks = tiles;
  HUAL_DYN_LDS(dw_f16_balanced_kernel, 144 * 1024);
  HUAL_LAUNCH(flops, bytes, dw_f16_balanced_kernel, dim3(nblocks), dim3(DWB_THREADS), (size_t)8 * DWB_PLANE, stream, (const DwJob*)table, n, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual
