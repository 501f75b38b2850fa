// Job-table GEMM kernels (see gemm.h).  Exact fp32 on the CDNA4 matrix cores.
//
// gemm_kernel: one wave computes a 16(row) x 64(col) output tile as four 16x16 accumulators with
// v_mfma_f32_16x16x4_f32.  Operand fragments come straight from global memory (weights are <= 512 KB and
// L2-resident; activations are [rows,128]) as 16-byte loads, WITHOUT an LDS stage or any barrier:
//   * lane (j = lane&15, g = lane>>4) loads A[row0+j][k0+4g .. +3]        -> k-step c uses component c
//   * and W[k0+4g+c][n0+4j .. +3] for c = 0..3                             -> accumulator t uses component t
// i.e. the MFMA's k index inside a 16-deep chunk is permuted (k = 4g + c) identically on A and B, and the
// accumulator's column index j of tile t is output column n0 + 4j + t, so the epilogue stores float4s.
// For dX = dY.W^T (transW) the same A pattern is used and lane (j,g) loads W[n0+4j+t][k0+4g .. +3].
//
// dw_kernel: dW += A^T.dY with v_mfma_f32_32x32x2_f32 (k index of the MFMA = row m of A/dY), 64x64 tile per
// wave, four waves of a block split the block's M-chunk, are summed through LDS and leave as ONE set of
// float atomics per block whose wave-instructions are two contiguous 128-B row segments (the full-rate
// atomic shape of MI355X_MICROARCH.md "Global float atomics").
#include "gemm.h"
#include "philox.h"
#include "prof.h"
#include <string.h>

namespace hual {

void gemm_job_init(GemmJob& j) {
  ::memset((void*)&j, 0, sizeof(j));
  j.a_drop_site = -1;
  j.drop_site = -1;
  j.add_div = 1;
}
void dw_job_init(DwJob& j) {
  ::memset((void*)&j, 0, sizeof(j));
  j.a_drop_site = -1;
}

}  // namespace hual

using namespace hual;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float f4get(const float4& v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

struct Frag {
  float4 a;
  float4 b[4];
};

__device__ __forceinline__ void load_frag(Frag& f, const float* Ap, const float* A2p, const float* Wp, int ldw,
                                          int transW, int k0, int g, int j, int n0, int N, bool adrop,
                                          const DropCfg& drop, uint32_t site, uint32_t droprow) {
  f.a = ld4(Ap + k0 + 4 * g);
  if (A2p) f.a = f4mul(f.a, ld4(A2p + k0 + 4 * g));
  if (adrop) f.a = apply_drop4(drop, site, droprow, (uint32_t)((k0 + 4 * g) >> 2), f.a);
  if (!transW) {
#pragma unroll
    for (int c = 0; c < 4; ++c) f.b[c] = ld4(Wp + (size_t)(k0 + 4 * g + c) * ldw + min(n0 + 4 * j, N - 4));
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) f.b[t] = ld4(Wp + (size_t)min(n0 + 4 * j + t, N - 1) * ldw + k0 + 4 * g);
  }
}

__device__ __forceinline__ void mma_frag(f32x4 (&acc)[4], const float4& a, const float4 (&b)[4], int transW) {
  if (!transW) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float av = f4get(a, c);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(av, f4get(b[c], t), acc[t]);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float av = f4get(a, c);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(av, f4get(b[t], c), acc[t]);
    }
  }
}

// chunk index (16 k's each, over the concatenated pieces) -> piece and offset inside the piece (wave-uniform)
__device__ __forceinline__ void chunk_to_piece(const GemmJob& job, int ch, int& p, int& k0) {
  int k = ch * 16;
  p = 0;
  while (p + 1 < job.npieces && k >= job.kw[p]) {
    k -= job.kw[p];
    ++p;
  }
  k0 = k;
}

template <bool DUAL>
__device__ __forceinline__ void load_chunk(const GemmJob& job, int ch, int arow, int g, int j, int n0, bool adrop,
                                           const DropCfg& drop, Frag& f, Frag& fb) {
  int p, k0;
  chunk_to_piece(job, ch, p, k0);
  const float* Ap = job.A[p] + (size_t)arow * job.lda[p];
  const float* A2p = job.A2[p] ? job.A2[p] + (size_t)arow * job.lda2[p] : nullptr;
  load_frag(f, Ap, A2p, job.W[p], job.ldw, job.transW, k0, g, j, n0, job.N, adrop, drop, (uint32_t)job.a_drop_site,
            job.a_drop_row0 + (uint32_t)arow);
  if (DUAL) {
    const float* Abp = job.Ab[p] ? job.Ab[p] + (size_t)arow * job.ldab[p] : Ap;
    load_frag(fb, Abp, nullptr, job.W2[p], job.ldw, job.transW, k0, g, j, n0, job.N, false, drop, 0, 0);
  }
}

#define HUAL_PD 4   // prefetch depth in 16-k chunks: the whole K=128 panel of a wave is in flight after 2 groups

template <bool DUAL>
__global__ __launch_bounds__(256) void gemm_kernel(GemmBatch batch, DropCfg drop) {
  const GemmJob& job = batch.j[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int M = job.M, N = job.N;
  const int rowbase = blockIdx.x * 32 + (wave >> 1) * 16;
  const int n0 = blockIdx.y * 128 + (wave & 1) * 64;
  if (rowbase >= M || n0 >= N) return;   // wave-uniform
  const int arow = min(rowbase + j, M - 1);
  const int transW = job.transW;
  const bool adrop = job.a_drop_site >= 0 && drop.enabled;

  f32x4 acc[4], acc2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  int ktot = 0;
  for (int p = 0; p < job.npieces; ++p) ktot += job.kw[p];
  const int nch = ktot >> 4;
  const int ngroups = nch / HUAL_PD;

  Frag f[HUAL_PD], fb[HUAL_PD];
  if (ngroups > 0) {
#pragma unroll
    for (int u = 0; u < HUAL_PD; ++u) load_chunk<DUAL>(job, u, arow, g, j, n0, adrop, drop, f[u], fb[u]);
    for (int gi = 0; gi + 1 < ngroups; ++gi) {
#pragma unroll
      for (int u = 0; u < HUAL_PD; ++u) {
        mma_frag(acc, f[u].a, f[u].b, transW);
        if (DUAL) mma_frag(acc2, fb[u].a, fb[u].b, transW);
        load_chunk<DUAL>(job, (gi + 1) * HUAL_PD + u, arow, g, j, n0, adrop, drop, f[u], fb[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < HUAL_PD; ++u) {
      mma_frag(acc, f[u].a, f[u].b, transW);
      if (DUAL) mma_frag(acc2, fb[u].a, fb[u].b, transW);
    }
  }
  for (int ch = ngroups * HUAL_PD; ch < nch; ++ch) {   // K tail (< HUAL_PD chunks)
    load_chunk<DUAL>(job, ch, arow, g, j, n0, adrop, drop, f[0], fb[0]);
    mma_frag(acc, f[0].a, f[0].b, transW);
    if (DUAL) mma_frag(acc2, fb[0].a, fb[0].b, transW);
  }

  // ---------------- epilogue: lane owns rows rowbase+4g+r (r=0..3), columns n0+4j .. n0+4j+3 -------------
  const int col = n0 + 4 * j;
  float4 bias = (job.bias && col < N) ? ld4(job.bias + col) : f4zero();
  float4 bias2 = f4zero();
  if (DUAL && job.bias2 && col < N) bias2 = ld4(job.bias2 + col);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = rowbase + 4 * g + r;
    if (row >= M || col >= N) continue;
    float4 v = make_float4(acc[0][r] + bias.x, acc[1][r] + bias.y, acc[2][r] + bias.z, acc[3][r] + bias.w);
    float rm = job.rowmask ? job.rowmask[row] : 1.0f;
    if (job.act == ACT_RELU) {
      v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    } else if (job.act == ACT_SIGMOID) {
      v = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
    } else if (job.act == ACT_SIGMOID_ROWMASK) {
      // sigmoid(mask_logits(x, m)): m=1 -> sigmoid(x); m=0 -> sigmoid(-1e30) == 0   (layers.py:110)
      v = rm != 0.f ? make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w)) : f4zero();
    }
    if (DUAL) {
      float4 v2 = make_float4(acc2[0][r] + bias2.x, acc2[1][r] + bias2.y, acc2[2][r] + bias2.z, acc2[3][r] + bias2.w);
      if (job.comb == COMB_GATE_VAL) {
        if (job.save) st4(job.save + (size_t)row * job.ldsave + col, v);
        if (job.save2) st4(job.save2 + (size_t)row * job.ldsave2 + col, v2);
        v = f4mul(v, v2);
      } else if (job.comb == COMB_CROSSGATE) {
        v2 = make_float4(sigmoidf_(v2.x), sigmoidf_(v2.y), sigmoidf_(v2.z), sigmoidf_(v2.w));
        if (job.save) st4(job.save + (size_t)row * job.ldsave + col, v);
        if (job.save2) st4(job.save2 + (size_t)row * job.ldsave2 + col, v2);
        float4 x1 = ld4(job.aux1 + (size_t)row * job.ldaux + col);
        float4 x2 = ld4(job.aux2 + (size_t)row * job.ldaux + col);
        v = f4add(f4mul(v, x1), f4mul(v2, x2));
      }
    } else {
      if (job.save) st4(job.save + (size_t)row * job.ldsave + col, v);
    }
    if (job.mulmode != MUL_NONE) {
      float4 m = ld4(job.mul + (size_t)row * job.ldmul + col);
      if (job.mulmode == MUL_TENSOR) {
        v = f4mul(v, m);
      } else if (job.mulmode == MUL_DRELU) {
        v = make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
      } else {
        v = make_float4(v.x * m.x * (1.f - m.x), v.y * m.y * (1.f - m.y), v.z * m.z * (1.f - m.z), v.w * m.w * (1.f - m.w));
      }
    }
    if (job.drop_site >= 0 && drop.enabled)
      v = apply_drop4(drop, (uint32_t)job.drop_site, job.drop_row0 + (uint32_t)row, (uint32_t)(col >> 2), v);
    if (job.add) v = f4add(v, ld4(job.add + (size_t)(row / job.add_div) * job.ldadd + col));
    if (job.mask_out) v = make_float4(v.x * rm, v.y * rm, v.z * rm, v.w * rm);
    st4(job.Y + (size_t)row * job.ldy + col, v);
  }
}

// ------------------------------------------------------------------------------------------------------
// dW / db.  A block owns one 128(k) x 128(n) gradient tile of one job over a chunk of rows_per_block rows; its four
// waves own the four 64 x 64 quadrants.  Rows of A (with the job's prologue: elementwise product, dropout) and of dY
// stream through LDS in 32-row tiles, loaded ONCE with coalesced 16-byte loads and double buffered, so every
// activation / gradient element is read once per job from HBM.  The MFMA is v_mfma_f32_32x32x2_f32 with the row
// index m as its k dimension; accumulators leave as float atomics whose wave-instructions are two contiguous
// 128-byte row segments (the full-rate shape of MI355X_MICROARCH.md "Global float atomics").
#define DW_TM 32            // rows per LDS tile
#define DW_LD 132           // padded leading dimension (floats)
// copies a slice of job descriptors (passed by value, so graph-capture safe) into the device-resident job table
__global__ void dw_table_write_kernel(DwBatch part, DwJob* table, int base, int cnt) {
  const int t = threadIdx.x;
  if (t < cnt) table[base + t] = part.j[t];
}

template <bool FROM_TABLE>
__global__ __launch_bounds__(256) void dw_kernel(DwBatch batch, const DwJob* __restrict__ table, DropCfg drop,
                                                 int rows_per_block) {
  extern __shared__ float lds[];     // As[2][DW_TM][DW_LD] | Ys[2][DW_TM][DW_LD]
  const DwJob& job = FROM_TABLE ? table[blockIdx.z] : batch.j[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int kq = wave >> 1, nq = wave & 1;
  const int M = job.M;
  // decode k-block -> (piece, offset inside the piece)
  int p = 0, kb = blockIdx.y;
  for (p = 0; p < job.npieces; ++p) {
    const int nkb = (job.kw[p] + 127) >> 7;
    if (kb < nkb) break;
    kb -= nkb;
  }
  if (p >= job.npieces) return;                 // block-uniform
  const int m_lo = blockIdx.x * rows_per_block;
  if (m_lo >= M) return;                        // block-uniform
  const int m_hi = min(m_lo + rows_per_block, M);
  const int k0 = kb * 128;
  const int kw = job.kw[p];
  const float* Ap = job.A[p];
  const float* A2p = job.A2[p];
  const int lda = job.lda[p], lda2 = job.lda2[p];
  const float* Yp = job.dY;
  const int ldy = job.ldy;
  const bool adrop = job.a_drop_site >= 0 && drop.enabled;
  const uint32_t asite = (uint32_t)job.a_drop_site;
  const uint32_t arow0 = job.a_drop_row0;
  float* As = lds;
  float* Ys = lds + 2 * DW_TM * DW_LD;

  f32x16 acc[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;
  float bsum0 = 0.f, bsum1 = 0.f;
  const bool dob = (job.db != nullptr) && p == 0 && kb == 0 && kq == 0;

  // staging: 32 rows x 32 float4 per matrix = 1024 float4 -> 4 per thread per matrix
  float4 ra[4], ry[4];
  auto stage_load = [&](int mt) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = threadIdx.x + 256 * u;
      const int row = idx >> 5, c4 = idx & 31;
      const int m = mt + row;
      float4 a = f4zero(), y = f4zero();
      if (m < m_hi) {
        const int k = k0 + 4 * c4;
        if (k < kw) {
          a = ld4(Ap + (size_t)m * lda + k);
          if (A2p) a = f4mul(a, ld4(A2p + (size_t)m * lda2 + k));
          if (adrop) a = apply_drop4(drop, asite, arow0 + (uint32_t)m, (uint32_t)(k >> 2), a);
        }
        y = ld4(Yp + (size_t)m * ldy + 4 * c4);
      }
      ra[u] = a;
      ry[u] = y;
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = threadIdx.x + 256 * u;
      const int row = idx >> 5, c4 = idx & 31;
      *reinterpret_cast<float4*>(As + (buf * DW_TM + row) * DW_LD + 4 * c4) = ra[u];
      *reinterpret_cast<float4*>(Ys + (buf * DW_TM + row) * DW_LD + 4 * c4) = ry[u];
    }
  };

  stage_load(m_lo);
  stage_store(0);
  __syncthreads();
  int buf = 0;
  for (int mt = m_lo; mt < m_hi; mt += DW_TM) {
    const bool more = (mt + DW_TM) < m_hi;
    if (more) stage_load(mt + DW_TM);            // global loads in flight under the MFMAs below
    const float* At = As + buf * DW_TM * DW_LD + kq * 64 + 2 * i;
    const float* Yt = Ys + buf * DW_TM * DW_LD + nq * 64 + i;
#pragma unroll 4
    for (int r = 0; r < DW_TM; r += 2) {
      const float2 a = *reinterpret_cast<const float2*>(At + (r + h) * DW_LD);
      const float b0 = Yt[(r + h) * DW_LD];
      const float b1 = Yt[(r + h) * DW_LD + 32];
      acc[0][0] = mfma32(a.x, b0, acc[0][0]);
      acc[0][1] = mfma32(a.x, b1, acc[0][1]);
      acc[1][0] = mfma32(a.y, b0, acc[1][0]);
      acc[1][1] = mfma32(a.y, b1, acc[1][1]);
      bsum0 += b0;
      bsum1 += b1;
    }
    if (more) stage_store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // each wave owns its quadrant: no cross-wave reduction, straight to global atomics
  float* dWp = job.dW[p];
  const int kbase = k0 + kq * 64, nbase = nq * 64;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowi = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = kbase + 2 * rowi + c;
        const int n = nbase + 32 * t + i;
        if (k < kw) atomicAdd(dWp + (size_t)k * job.ldw + n, acc[c][t][r]);
      }
  if (dob) {
    bsum0 += __shfl_xor(bsum0, 32);
    bsum1 += __shfl_xor(bsum1, 32);
    if (h == 0) {
      atomicAdd(job.db + nbase + i, bsum0);
      atomicAdd(job.db + nbase + 32 + i, bsum1);
    }
  }
}

namespace hual {

int launch_gemm(const GemmJob* jobs, int n, const DropCfg& drop, hipStream_t stream) {
  HUAL_REQUIRE(n >= 1 && n <= HUAL_MAX_JOBS, "launch_gemm: job count");
  GemmBatch b;
  int maxM = 0, maxN = 0;
  bool dual = false;
  for (int i = 0; i < n; ++i) {
    const GemmJob& j = jobs[i];
    HUAL_REQUIRE(j.M > 0 && j.N >= 4 && (j.N % 4) == 0, "launch_gemm: N must be a positive multiple of 4");
    HUAL_REQUIRE(j.npieces >= 1 && j.npieces <= HUAL_MAX_PIECES, "launch_gemm: pieces");
    for (int p = 0; p < j.npieces; ++p) {
      HUAL_REQUIRE(j.kw[p] > 0 && (j.kw[p] % 16) == 0, "launch_gemm: piece width must be a multiple of 16");
      HUAL_REQUIRE(j.A[p] && j.W[p], "launch_gemm: null operand");
      HUAL_REQUIRE((j.lda[p] % 4) == 0 && (j.ldw % 4) == 0, "launch_gemm: leading dims must be multiples of 4");
    }
    HUAL_REQUIRE(j.Y != nullptr, "launch_gemm: null output");
    HUAL_REQUIRE(j.add_div >= 1, "launch_gemm: add_div");
    if (j.comb != COMB_NONE) dual = true;
    b.j[i] = j;
    maxM = j.M > maxM ? j.M : maxM;
    maxN = j.N > maxN ? j.N : maxN;
  }
  for (int i = 0; i < n; ++i)
    HUAL_REQUIRE((jobs[i].comb != COMB_NONE) == dual, "launch_gemm: cannot mix dual and single jobs in one launch");
  dim3 grid(cdiv(maxM, 32), cdiv(maxN, 128), n), block(256);
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    double kt = 0.0;
    for (int p = 0; p < jobs[i].npieces; ++p) kt += jobs[i].kw[p];
    const double mult = dual ? 2.0 : 1.0;
    flops += 2.0 * jobs[i].M * kt * jobs[i].N * mult;
    bytes += 4.0 * ((double)jobs[i].M * kt + kt * jobs[i].N * mult + (double)jobs[i].M * jobs[i].N);
  }
  ProfScope ps(dual ? PK_GEMM_DUAL : PK_GEMM, stream, flops, bytes);
  if (dual)
    hipLaunchKernelGGL(gemm_kernel<true>, grid, block, 0, stream, b, drop);
  else
    hipLaunchKernelGGL(gemm_kernel<false>, grid, block, 0, stream, b, drop);
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

// `table`: optional device buffer of n DwJob entries.  With it ALL jobs run as ONE launch (the descriptors are first
// written to the table by tiny kernels that carry them by value); without it jobs go HUAL_MAX_DW_JOBS per launch.
int launch_dw(const DwJob* jobs, int n, const DropCfg& drop, int rows_per_block, hipStream_t stream, DwJob* table) {
  HUAL_REQUIRE(rows_per_block >= DW_TM && (rows_per_block % DW_TM) == 0, "launch_dw: rows_per_block must be a multiple of 32");
  static bool attr = false;
  if (!attr) {
    HUAL_CHECK_HIP(hipFuncSetAttribute((const void*)dw_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    HUAL_CHECK_HIP(hipFuncSetAttribute((const void*)dw_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr = true;
  }
  const size_t lds = (size_t)4 * DW_TM * DW_LD * sizeof(float);
  if (table != nullptr) {
    int maxM = 0, maxKb = 0;
    double flops = 0.0, bytes = 0.0;
    for (int base = 0; base < n; base += HUAL_MAX_DW_JOBS) {
      const int cnt = n - base < HUAL_MAX_DW_JOBS ? n - base : HUAL_MAX_DW_JOBS;
      DwBatch b;
      for (int i = 0; i < cnt; ++i) {
        int kbs;
        int rc = dw_check(jobs[base + i], kbs, flops, bytes);
        if (rc) return rc;
        b.j[i] = jobs[base + i];
        maxM = jobs[base + i].M > maxM ? jobs[base + i].M : maxM;
        maxKb = kbs > maxKb ? kbs : maxKb;
      }
      hipLaunchKernelGGL(dw_table_write_kernel, dim3(1), dim3(64), 0, stream, b, table, base, cnt);
    }
    DwBatch dummy;
    dw_job_init(dummy.j[0]);
    dim3 grid(cdiv(maxM, rows_per_block), maxKb, n), block(256);
    ProfScope ps(PK_DW, stream, flops, bytes);
    hipLaunchKernelGGL(dw_kernel<true>, grid, block, lds, stream, dummy, (const DwJob*)table, drop, rows_per_block);
    HUAL_CHECK_HIP(hipGetLastError());
    return 0;
  }
  for (int base = 0; base < n; base += HUAL_MAX_DW_JOBS) {
    int cnt = n - base < HUAL_MAX_DW_JOBS ? n - base : HUAL_MAX_DW_JOBS;
    DwBatch b;
    int maxM = 0, maxKb = 0;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < cnt; ++i) {
      int kbs;
      int rc = dw_check(jobs[base + i], kbs, flops, bytes);
      if (rc) return rc;
      b.j[i] = jobs[base + i];
      maxM = jobs[base + i].M > maxM ? jobs[base + i].M : maxM;
      maxKb = kbs > maxKb ? kbs : maxKb;
    }
    dim3 grid(cdiv(maxM, rows_per_block), maxKb, cnt), block(256);
    ProfScope ps(PK_DW, stream, flops, bytes);
    hipLaunchKernelGGL(dw_kernel<false>, grid, block, lds, stream, b, (const DwJob*)nullptr, drop, rows_per_block);
    HUAL_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

}  // namespace hual
