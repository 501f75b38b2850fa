// Row-wise kernels (see rowops.h).  One activation row = 128 floats = 32 lanes x float4; a 256-thread block
// works on 8 rows at a time; row reductions are wave shuffles inside a 32-lane half (no LDS, no barrier).
#include <string.h>
#include "rowops.h"
#include "embed_gather.h"
#include "philox.h"
#include "prof.h"

using namespace hual;

#define LN_EPS 1e-6f   // models/layers.py:15

__device__ __forceinline__ float4 f4mul_(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4add_(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub_(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4scale_(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float f4hsum_(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float4 f4fma_(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// mean / rstd of one row held as one float4 per lane of a 32-lane half (biased variance, layers.py:13-15)
__device__ __forceinline__ void row_stats(float4 v, float& mean, float& rstd) {
  mean = half_sum32(f4hsum_(v)) * (1.0f / HUAL_D);
  float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
  float var = half_sum32(f4hsum_(f4mul_(d, d))) * (1.0f / HUAL_D);
  rstd = rsqrtf(var + LN_EPS);
}

// dz = dropout'(dx) with the keep bits the forward left (bit plane: byte [row * 16 + (col >> 3)], bit col & 7; csrc/tilecore.h):
// the operand of the next dX product, written next to dx
__device__ __forceinline__ void store_dz(float* dz, const uint8_t* bits, float scale, int row, size_t off, int l32, float4 dx) {
  if (bits) {
    const uint32_t nib = ((uint32_t)bits[(size_t)row * 16 + (l32 >> 1)] >> (4 * (l32 & 1))) & 15u;
    dx = make_float4((nib & 1u) ? dx.x * scale : 0.f, (nib & 2u) ? dx.y * scale : 0.f, (nib & 4u) ? dx.z * scale : 0.f, (nib & 8u) ? dx.w * scale : 0.f);
  }
  st4(dz + off, dx);
}

__device__ __forceinline__ void row_to_clip(int row, int Nv, int T, int L, int& t, int& n, int& base) {
  const bool v = row < Nv;
  n = v ? T : L;
  const int first = v ? 0 : Nv;
  base = first + __mul24(small_div(row - first, n), n);
  t = row - base;
}

// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwd a, RowSpace rs, DropCfg drop) {
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  for (int row = blockIdx.x * 8 + grp; row < a.R; row += gridDim.x * 8) {
    float4 v;
    const bool hi = a.split > 0 && row >= a.split;
    if (a.part) {       // K-split partial sums of the producing dense layer (+ its bias)
      v = ld4((hi ? a.part_bias_hi : a.part_bias) + col);
      for (int q = 0; q < a.nparts; ++q) v = f4add_(v, ld4(a.part + (size_t)q * a.part_stride + (size_t)row * HUAL_D + col));
      st4(a.x_out + (size_t)row * HUAL_D + col, v);
    } else {
      v = ld4(a.x + (size_t)row * HUAL_D + col);
    }
    float mean, rstd;
    row_stats(v, mean, rstd);
    float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
    float4 y = f4fma_(xh, ld4((hi ? a.g1_hi : a.g1) + col), ld4((hi ? a.b1_hi : a.b1) + col));
    if (a.pos) {
      int t, n, base;
      row_to_clip(row + a.row0, rs.Nv, rs.T, rs.L, t, n, base);
      y = f4add_(y, ld4(a.pos + (size_t)t * HUAL_D + col));
    }
    st4(a.y1 + (size_t)row * HUAL_D + col, y);
    if (a.y2) st4(a.y2 + (size_t)row * HUAL_D + col, f4fma_(xh, ld4(a.g2 + col), ld4(a.b2 + col)));
    if (a.mean && l32 == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
  }
}

// ------------------------------------------------------------------------------------------------------
// LN backward.  For y = xhat*g + b:  gv = dy*g ;  dx = rstd * (gv - mean(gv) - xhat * mean(gv*xhat))
// (ngrid = workgroups of the layer-norm part of the launch)
__device__ __forceinline__ void ln_bwd_body(const LnBwd& a, const DropCfg& drop, int ngrid) {
  __shared__ float4 red[4][8][32];
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = 4 * l32;
  float4 sg1 = f4zero(), sb1 = f4zero(), sg2 = f4zero(), sb2 = f4zero();
  const int nlo = a.split > 0 ? (a.split + 31) / 32 : 0;
  const bool hi = a.split > 0 && (int)blockIdx.x >= nlo;
  const int row_lo = hi ? a.split : 0, row_hi = (a.split > 0 && !hi) ? a.split : a.R;
  const int bid = hi ? (int)blockIdx.x - nlo : (int)blockIdx.x, nblk = a.split > 0 ? (hi ? ngrid - nlo : nlo) : ngrid;
  const float4 g1 = ld4((hi ? a.g1_hi : a.g1) + col);
  const float4 g2 = a.dy2 ? ld4(a.g2 + col) : f4zero();
  for (int row = row_lo + bid * 8 + grp; row < row_hi; row += nblk * 8) {
    const size_t off = (size_t)row * HUAL_D + col;
    float4 v = ld4(a.x + off);
    float mean = a.mean[row], rstd = a.rstd[row];
    float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
    float4 dy = ld4(a.dy1 + off);
    sb1 = f4add_(sb1, dy);
    sg1 = f4fma_(dy, xh, sg1);
    float4 gv = f4mul_(dy, g1);
    if (a.dy2) {
      float4 dy2 = ld4(a.dy2 + off);
      sb2 = f4add_(sb2, dy2);
      sg2 = f4fma_(dy2, xh, sg2);
      gv = f4fma_(dy2, g2, gv);
    }
    float m1 = half_sum32(f4hsum_(gv)) * (1.0f / HUAL_D);
    float m2 = half_sum32(f4hsum_(f4mul_(gv, xh))) * (1.0f / HUAL_D);
    float4 dx = make_float4(rstd * (gv.x - m1 - xh.x * m2), rstd * (gv.y - m1 - xh.y * m2),
                            rstd * (gv.z - m1 - xh.z * m2), rstd * (gv.w - m1 - xh.w * m2));
    if (a.add1) dx = f4add_(dx, ld4(a.add1 + off));
    if (a.add2) dx = f4add_(dx, ld4(a.add2 + off));
    st4(a.dx + off, dx);
    if (a.dz) store_dz(a.dz, drop.enabled ? a.dz_bits : nullptr, drop.scale, row, off, l32, dx);
  }
  red[0][grp][l32] = sg1; red[1][grp][l32] = sb1; red[2][grp][l32] = sg2; red[3][grp][l32] = sb2;
  __syncthreads();
  // 4 vectors x 128 columns = 512 sums of 8 partials; thread t handles (vec = t>>7 .. ) two passes
  for (int idx = threadIdx.x; idx < 512; idx += 256) {
    const int vec = idx >> 7, c = idx & 127;
    if (vec >= 2 && !a.dy2) continue;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += reinterpret_cast<const float*>(&red[vec][k][c >> 2])[c & 3];
    if (a.part) {
      a.part[((size_t)blockIdx.x * 4 + vec) * HUAL_D + c] = s;
      continue;
    }
    float* dst = vec == 0 ? a.dg1 : (vec == 1 ? a.db1 : (vec == 2 ? a.dg2 : a.db2));
    if (dst) atomicAdd(dst + c, s);
  }
}
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwd a, DropCfg drop) { ln_bwd_body(a, drop, (int)gridDim.x); }

// ------------------------------------------------------------------------------------------------------
struct ColsumBatch { ColsumJob j[HUAL_COLSUM_MAX_JOBS]; };
#define COLSUM_SPLIT 16   // independent row slices per (job, vector): the kernel is a chain of dependent-latency loads (18.5 us at 4, see profiles)
// One grid row of workgroups: the riding embed_finish blocks FIRST (they are the long ones of the launch: dispatched behind the
// thousands of short fold workgroups they were its tail - 16.2 us against 9.0 without them), then the fold / unpack workgroups in
// the order (job or unpack column x, vector y, slice z), x fastest, of the former three-dimensional grid (rx, ry, rz = its dimensions).
__global__ __launch_bounds__(256) void colsum_kernel(ColsumBatch batch, int njobs, EmbedUnpack eu, int rx, int ry) {
  __shared__ float part[HUAL_D];
  extern __shared__ float colsum_dyn[];      // (embed_finish_block's accumulator, when that step rides here)
  if ((int)blockIdx.x < eu.finish_blocks) {      // (block-uniform)
    embed_finish_block(eu.a, eu.g, eu.drop, eu.nrows, eu.CP, (int)blockIdx.x, colsum_dyn);
    return;
  }
  const int role = (int)blockIdx.x - eu.finish_blocks;
  const int bx = role % rx, byz = role / rx, by = byz % ry, bz = byz / ry;
  if (bx >= njobs) {      // the workgroups behind the jobs: unpack tasks, one per thread
    const int lid = ((bx - njobs) * ry + by) * COLSUM_SPLIT + bz;
    const int gid = lid * 256 + (int)threadIdx.x;
    if (gid < eu.ntasks) embed_unpack_task(eu.a, eu.g, eu.CP, gid);
    return;
  }
  const ColsumJob& job = batch.j[bx];
  const int vec = by;
  if (vec >= job.nvec || job.dst[vec] == nullptr) return;     // block-uniform
  const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
  float s = 0.f;
#pragma unroll 8
  for (int r = bz * 2 + half; r < job.nblk; r += 2 * COLSUM_SPLIT) s += job.src[((size_t)r * job.nvec + vec) * HUAL_D + c];
  if (half) part[c] = s;
  __syncthreads();
  if (!half && !(job.last_ncols > 0 && vec == job.nvec - 1 && c >= job.last_ncols)) atomicAdd(job.dst[vec] + c, s + part[c]);
}

// ------------------------------------------------------------------------------------------------------
// NG = groups of 128 threads; (t, jb) = position, job of the workgroup
template <int NG>
__device__ __forceinline__ void pos_bwd_body(const PosBwdBatch& batch, const RowSpace& rs, int t, int jb) {
  __shared__ float part[NG][HUAL_D];
  const PosBwdJob& job = batch.j[jb];
  const int c = threadIdx.x & 127, grp = threadIdx.x >> 7;
  float s = 0.f;
  for (int k = 0; k < 2; ++k) {
    const float* dx = job.dx[k];
    if (!dx) continue;
    if (job.do_v && t < rs.T) {
#pragma unroll 8
      for (int b = grp; b < rs.B; b += NG) s += dx[(size_t)(b * rs.T + t) * HUAL_D + c];
    }
    if (job.do_q && t < rs.L) {
#pragma unroll 8
      for (int b = grp; b < rs.B; b += NG) s += dx[(size_t)(rs.Nv + b * rs.L + t) * HUAL_D + c];
    }
  }
  part[grp][c] = s;
  __syncthreads();
  if (grp == 0 && ((job.do_v && t < rs.T) || (job.do_q && t < rs.L))) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < NG; ++k) tot += part[k][c];
    job.dpos[(size_t)t * HUAL_D + c] += tot;
  }
}
__global__ __launch_bounds__(512) void pos_bwd_kernel(PosBwdBatch batch, RowSpace rs) { pos_bwd_body<4>(batch, rs, blockIdx.x, blockIdx.y); }
// a layer-norm backward and position-table jobs that read the same gradient tensor, one launch: workgroups [0, nln) the former,
// the rest the latter (npos positions per job)
__global__ __launch_bounds__(256) void ln_pos_bwd_kernel(LnBwd a, DropCfg drop, PosBwdBatch batch, RowSpace rs, int nln, int npos) {
  if ((int)blockIdx.x < nln) ln_bwd_body(a, drop, nln);
  else pos_bwd_body<2>(batch, rs, ((int)blockIdx.x - nln) % npos, ((int)blockIdx.x - nln) / npos);
}

// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_rows_kernel(const float* a, const float* b, float* out, int R) {
  const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  for (int row = blockIdx.x * 8 + grp; row < R; row += gridDim.x * 8) {
    const size_t off = (size_t)row * HUAL_D + 4 * l32;
    st4(out + off, f4add_(ld4(a + off), ld4(b + off)));
  }
}

namespace hual {

static inline int row_grid(int R) {
  int g = cdiv(R, 8);
  return g < 2048 ? (g > 0 ? g : 1) : 2048;
}

int launch_ln_fwd(const LnFwd& a, const RowSpace& rs, const DropCfg& drop, hipStream_t s) {
  HUAL_REQUIRE((a.x || (a.part && a.part_bias && a.x_out && a.nparts > 0)) && a.g1 && a.b1 && a.y1 && a.R > 0, "ln_fwd: null/empty");
  HUAL_LAUNCH(0.0, (8.0 + (a.part ? 4.0 * a.nparts + 4.0 : 0.0)) * a.R * HUAL_D, ln_fwd_kernel, dim3(row_grid(a.R)), dim3(256), 0, s, a, rs, drop);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int ln_bwd_blocks(int R) { return cdiv(R, 32); }     // 4 rows per 32-lane group

int launch_ln_bwd(const LnBwd& a, const DropCfg& drop, hipStream_t s, const PosBwdJob* pos, int npos, const RowSpace* rs) {
  HUAL_REQUIRE(a.x && a.mean && a.rstd && a.dy1 && a.g1 && a.dx && a.R > 0, "ln_bwd: null/empty");
  int g = cdiv(a.R, 8);
  g = g < 128 ? g : 128;     // every block ends with 256-512 same-address float atomics: keep the count low
  if (a.part) g = ln_bwd_blocks(a.R);
  if (a.split > 0) {
    HUAL_REQUIRE(a.part && a.g1_hi && a.split < a.R && !a.dy2, "ln_bwd: split needs part, g1_hi, one layer norm");
    g = ln_bwd_blocks(a.split) + ln_bwd_blocks(a.R - a.split);
  }
  if (npos > 0) {
    HUAL_REQUIRE(pos && rs && npos <= HUAL_POS_MAX_JOBS, "ln_bwd: position-table jobs");
    PosBwdBatch b;
    ::memset((void*)&b, 0, sizeof(b));
    int n = 0;
    for (int i = 0; i < npos; ++i) {
      HUAL_REQUIRE(pos[i].dx[0] && pos[i].dpos, "pos_bwd: null tensor");
      b.j[i] = pos[i];
      if (pos[i].do_v && rs->T > n) n = rs->T;
      if (pos[i].do_q && rs->L > n) n = rs->L;
    }
    HUAL_LAUNCH(0.0, 12.0 * a.R * HUAL_D, ln_pos_bwd_kernel, dim3(g + n * npos), dim3(256), 0, s, a, drop, b, *rs, g, n);
  } else {
    HUAL_LAUNCH(0.0, 12.0 * a.R * HUAL_D, ln_bwd_kernel, dim3(g), dim3(256), 0, s, a, drop);
  }
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_colsum(const ColsumJob* jobs, int n, hipStream_t s, const EmbedUnpack* unpack) {
  if (n == 0 && !unpack) return 0;
  HUAL_REQUIRE(n >= 0 && n <= HUAL_COLSUM_MAX_JOBS, "colsum: job count");
  ColsumBatch b;
  ::memset((void*)&b, 0, sizeof(b));
  int maxvec = 0;
  for (int i = 0; i < n; ++i) {
    HUAL_REQUIRE(jobs[i].src && jobs[i].nblk > 0 && jobs[i].nvec >= 1 && jobs[i].nvec <= HUAL_COLSUM_MAX_VEC, "colsum: bad job");
    b.j[i] = jobs[i];
    maxvec = jobs[i].nvec > maxvec ? jobs[i].nvec : maxvec;
  }
  if (maxvec == 0) maxvec = 1;
  EmbedUnpack eu{};
  int xu = 0;
  int lds = 0;
  if (unpack) {
    eu = *unpack;
    xu = cdiv(cdiv(eu.ntasks, 256), maxvec * COLSUM_SPLIT);
    lds = eu.finish_blocks > 0 ? eu.finish_lds : 0;
    HUAL_REQUIRE(lds <= 64 * 1024, "colsum: LDS accumulator of the riding embed_finish blocks");
  }
  if (lds > 32 * 1024) HUAL_DYN_LDS(colsum_kernel, 64 * 1024);
  const int rx = n + xu;
  HUAL_LAUNCH(0.0, 0.0, colsum_kernel, dim3(eu.finish_blocks + rx * maxvec * COLSUM_SPLIT), dim3(256), lds, s, b, n, eu, rx, maxvec);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_pos_bwd(const PosBwdJob* jobs, int njobs, const RowSpace& rs, hipStream_t s) {
  HUAL_REQUIRE(njobs >= 1 && njobs <= HUAL_POS_MAX_JOBS, "pos_bwd: job count");
  PosBwdBatch b;
  ::memset((void*)&b, 0, sizeof(b));
  int n = 0;
  for (int i = 0; i < njobs; ++i) {
    HUAL_REQUIRE(jobs[i].dx[0] && jobs[i].dpos, "pos_bwd: null tensor");
    b.j[i] = jobs[i];
    if (jobs[i].do_v && rs.T > n) n = rs.T;
    if (jobs[i].do_q && rs.L > n) n = rs.L;
  }
  if (n == 0) return 0;
  HUAL_LAUNCH(0.0, 0.0, pos_bwd_kernel, dim3(n, njobs), dim3(512), 0, s, b, rs);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_add_rows(const float* a, const float* b, float* out, int R, hipStream_t s) {
  HUAL_REQUIRE(R > 0 && a && b && out, "add_rows: null/empty");
  HUAL_LAUNCH(0.0, 12.0 * R * HUAL_D, add_rows_kernel, dim3(row_grid(R)), dim3(256), 0, s, a, b, out, R);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hual


// ---- debug: NaN patterns over the whole LDS of every CU (prof.h lds_poison; HUAL_DEBUG_LDS_POISON=1).  One 160 KB workgroup fits a CU at a time,
// so a grid of twice the CU count reaches every CU whatever else is resident; plain launch (it must not poison in front of itself).
__global__ __launch_bounds__(256) void lds_poison_kernel() {
  extern __shared__ uint32_t lds_all[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) lds_all[i] = 0xFFFFFFFFu;
  __syncthreads();
  if (lds_all[threadIdx.x] == 0u) __builtin_trap();      // (keeps the stores alive)
}
namespace hual {
void lds_poison(hipStream_t stream) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL(lds_poison_kernel, dim3(2 * cus), dim3(256), 160 * 1024, stream);
}
}  // namespace hual
