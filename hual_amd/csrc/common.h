// Shared device/host helpers for the SeqPAN HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/hual_seqpan.h"

#define HUAL_D 128          // hidden size the kernels are specialised for (configs/*/SeqPAN.yaml model.dim)
#define HUAL_H 8            // heads
#define HUAL_DH 16          // head size
#define HUAL_MASK_VALUE (-1e30f)   // models/ops.py:89

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace hual {

// ---- error plumbing: no exception crosses the C ABI; message via hual_last_error() -------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define HUAL_CHECK_HIP(expr)                                                                   \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) return ::hual::fail(-2, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

#define HUAL_REQUIRE(cond, msg)                                                                \
  do {                                                                                         \
    if (!(cond)) return ::hual::fail(-1, std::string("invalid argument: ") + (msg));           \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// raise a kernel's dynamic-LDS limit once per (device, kernel); thread safe (core.cpp)
int ensure_dyn_lds(const void* fn, int bytes);
#define HUAL_DYN_LDS(fn, bytes)                                          \
  do {                                                                   \
    int _rc = ::hual::ensure_dyn_lds((const void*)(fn), (bytes));        \
    if (_rc) return _rc;                                                 \
  } while (0)

// ---- XCD-aware order of row tiles / clips.  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs, each with
// a private 4 MB L2, and what one kernel of a graph wrote is still in the writer's L2 when the next kernel starts (measured: a
// 4.8 MB tensor re-read from the XCD that wrote it costs a launch floor, from another XCD 2.1 us more - scripts/exp/xcd_locality.hip).
// Every kernel that owns rows or clips therefore hands XCD x the x-th contiguous eighth of its tiles (grids are rounded up to a
// multiple of 8, tiles beyond the end return at once), the same eighth of the row space the attention kernels give it.
__host__ __device__ inline int xcd_round8(int n) { return (n + 7) & ~7; }
__device__ __forceinline__ int xcd_tile(int b, int nblk) { return (b & 7) * (nblk >> 3) + (b >> 3); }      // nblk: multiple of 8
// The same for tiles of MT rows over the UNIFIED row space [Nv video rows | R - Nv query rows]: XCD x takes the tiles that start in
// video-row eighth x and those that start in query-row eighth x - the rows of clips [B x / 8, B (x + 1) / 8), which is what the
// attention kernels and the per-clip kernels give XCD x too.  (Plain eighths of the unified rows put clip b's video rows on XCD
// 128 b / 1184 but its attention on XCD b / 8 at the bench shape, and every query row on XCDs 6 - 7: the attention launches read their
// 24 MB of projections from another XCD's write-back instead of their own L2.)  Returns the tile id - every real tile exactly once
// over a grid of xcd_clip_grid() workgroups - or -1.
__host__ __device__ inline int xcd_clip_t(long rows, int MT) { return (int)((rows + MT - 1) / MT); }
__device__ __forceinline__ int xcd_tile_clip(int bid, int R, int Nv, int MT) {
  const int x = bid & 7, slot = bid >> 3;
  const long nv = Nv > 0 ? Nv : R, nq = R - nv;
  const int tv0 = xcd_clip_t(nv * x / 8, MT), tv1 = xcd_clip_t(nv * (x + 1) / 8, MT);
  if (slot < tv1 - tv0) return tv0 + slot;
  const int tq0 = xcd_clip_t(nv + nq * x / 8, MT), tq1 = xcd_clip_t(nv + nq * (x + 1) / 8, MT);
  const int s2 = slot - (tv1 - tv0);
  return s2 < tq1 - tq0 ? tq0 + s2 : -1;
}
static inline int xcd_clip_grid(int R, int Nv, int MT);
// rows per workgroup for R rows (video rows Nv, 0: one row space): the smallest MT >= min_rows that needs no more than one workgroup
// per CU under the clip order (no XCD gets more than 32 tiles), capped at max_rows
static inline int xcd_clip_rows(int R, int Nv, int min_rows, int max_rows);
static inline int xcd_clip_grid(int R, int Nv, int MT) {
  const long nv = Nv > 0 ? Nv : R, nq = R - nv;
  int m = 0;
  for (int x = 0; x < 8; ++x) {
    const int n = (xcd_clip_t(nv * (x + 1) / 8, MT) - xcd_clip_t(nv * x / 8, MT)) + (xcd_clip_t(nv + nq * (x + 1) / 8, MT) - xcd_clip_t(nv + nq * x / 8, MT));
    m = n > m ? n : m;
  }
  return 8 * m;
}
static inline int xcd_clip_rows(int R, int Nv, int min_rows, int max_rows) {
  int mt = (R + 255) / 256;
  if (mt < min_rows) mt = min_rows;
  while (mt < max_rows && xcd_clip_grid(R, Nv, mt) > 256) ++mt;
  return mt > max_rows ? max_rows : mt;
}

// ---- dropout parameters shared by every kernel (derived from drop_rate on the host) -----------
// The Philox key/offset live in DEVICE memory (state[0]=seed lo, [1]=seed hi, [2]=offset) so that a captured
// hipGraph can be replayed with a fresh offset every step without re-capturing.
struct DropCfg {
  const uint32_t* state;  // device: {k0, k1, offset}
  uint32_t thresh;        // keep iff rnd < thresh
  float scale;            // 1/(1-rate)
  int enabled;            // rate > 0
};

DropCfg make_dropcfg(const uint32_t* state, float rate);

}  // namespace hual

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
// x / n for 0 <= x < 2^20, 1 <= n <= 2^12 (row -> clip lookups: n = T or L) in 4 instructions instead of the ~30 of the integer
// division sequence (two quarter-rate multiplies among them; 17 % of the VALU issue cycles of conv_block_fwd_kernel's listing,
// scripts/exp/isa_by_line.py).  (x + 0.5) / n is >= 0.5 / n away from the nearest integer; the float product is within
// (x / n) * 2^-22 <= 0.25 / n of it (v_rcp_f32: 1 ulp), so the truncation is exact.  The callers' R = B (T + L) < 2^20 is checked
// on the host (seqpan.hip setup_ctx).
__device__ __forceinline__ int small_div(int x, int n) { return (int)(((float)x + 0.5f) * __builtin_amdgcn_rcpf((float)n)); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// loads through a pointer that was itself read from device memory (job tables): without the address-space cast they are
// FLAT loads, which count in lgkmcnt as well - every wait for an LDS read then waits for the global loads in flight
typedef float gvec_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int gvec_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld4_global(const void* p) {
  const gvec_f4 v = *(const __attribute__((address_space(1))) gvec_f4*)(uintptr_t)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
// float atomic add through a global-address-space pointer (a FLAT atomic would also count in lgkmcnt)
__device__ __forceinline__ void atomic_add_global(float* p, float v) {
  __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(uintptr_t)p, v);
}
__device__ __forceinline__ uint32_t ld1_global(const void* p) { return *(const __attribute__((address_space(1))) uint8_t*)(uintptr_t)p; }
__device__ __forceinline__ uint2 ld2_global(const void* p) {
  const gvec_u2 v = *(const __attribute__((address_space(1))) gvec_u2*)(uintptr_t)p;
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// componentwise select (a ternary on the struct goes through a stack slot: the compiler selects the ADDRESS)
__device__ __forceinline__ float4 f4_pick(bool c, float4 a, float4 b) { return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }
// ---- range-checked row stores: buffer instructions drop a store whose offset lies outside the resource, so "store this row only if
// it belongs to the tile" needs no branch - and a branch around a vector-memory operation makes the compiler's wait-count pass give
// up (it drains EVERY outstanding load and store at the next wait: a memory round trip per guarded store, scripts/exp/isa_vmcnt0.py).
// A resource covers `bytes` from `base` (< 4 GB); a lane whose row is not to be written passes an offset of ROW_SKIP.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#define ROW_SKIP 0xfffffff0u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);      // raw buffer, 32-bit data format (gfx9 family)
}
__device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, float4 v) {
  const u32x4_t d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(d, r, byte_off, 0, 0);
}
__device__ __forceinline__ void bst1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, byte_off, 0, 0);
}
// stores through a pointer whose address space the compiler cannot see (read from LDS / a device table): as FLAT stores they
// would count in lgkmcnt too
__device__ __forceinline__ void st4_global(float* p, float4 v) {
  const gvec_f4 vv = {v.x, v.y, v.z, v.w};
  *(__attribute__((address_space(1))) gvec_f4*)(uintptr_t)p = vv;
}
__device__ __forceinline__ void st1f_global(float* p, float v) { *(__attribute__((address_space(1))) float*)(uintptr_t)p = v; }
__device__ __forceinline__ void st1b_global(uint8_t* p, uint8_t v) { *(__attribute__((address_space(1))) uint8_t*)(uintptr_t)p = v; }
typedef int gvec_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4i_global(int32_t* p, int a, int b, int c, int d) {
  const gvec_i4 vv = {a, b, c, d};
  *(__attribute__((address_space(1))) gvec_i4*)(uintptr_t)p = vv;
}
// streaming store: tensors that are written once and only read again much later (saved for the backward pass, operands of the
// weight-gradient launch at the end of the step) - measured -4 % on da_post_kernel against plain stores
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  const f32x4 vv = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// relu that lets a NaN through, like tf.nn.relu (fmaxf(NaN, 0) is 0: an activation beyond the fp16 operand range - Inf - Inf = NaN in a
// product - would come out of the next relu as a plausible zero instead of reaching the loss / the logits)
__device__ __forceinline__ float relu_nan(float x) { return x < 0.f ? 0.f : x; }
__device__ __forceinline__ float4 relu_nan4(const float4& v) { return make_float4(relu_nan(v.x), relu_nan(v.y), relu_nan(v.z), relu_nan(v.w)); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ float wave_sum16(float v) {   // sum across the 16 lanes sharing lane>>4
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}
// ---- 32-lane butterfly reductions on the VALU (DPP + v_permlane16_swap) --------------------------------------------------
// Same pairs, same order as common.h half_sum32 / half_max32 (xor 1, 2, 4, 8, 16: bit-identical results), but without the
// five dependent ds_bpermute round trips (~100+ cycles each) those cost: at one workgroup per CU the fused kernels are
// latency bound and a row reduction per ds_bpermute chain was measured to dominate them (15 us per conv_block layer).
__device__ __forceinline__ float dpp_xor_partner(float v, int step) {
  const int x = __builtin_bit_cast(int, v);
  int r;
  if (step == 1) r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);          // quad_perm [1,0,3,2]
  else if (step == 2) r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
  else if (step == 4) {
    r = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);                       // row_shl:4 -> lanes 0-3, 8-11 read lane + 4
    r = __builtin_amdgcn_update_dpp(r, x, 0x114, 0xF, 0xA, false);                       // row_shr:4 -> lanes 4-7, 12-15 read lane - 4
  } else {
    r = __builtin_amdgcn_update_dpp(x, x, 0x108, 0xF, 0x3, false);                       // row_shl:8 -> lanes 0-7 read lane + 8
    r = __builtin_amdgcn_update_dpp(r, x, 0x118, 0xF, 0xC, false);                       // row_shr:8 -> lanes 8-15 read lane - 8
  }
  return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float lane_xor16_partner(float v) {
  const unsigned x = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);    // r[0]: odd rows <- even rows, r[1]: even rows <- odd rows
  const unsigned p = (threadIdx.x & 16) ? r[0] : r[1];
  return __builtin_bit_cast(float, p);
}
__device__ __forceinline__ float fast_sum32(float v) {
  v += dpp_xor_partner(v, 1);
  v += dpp_xor_partner(v, 2);
  v += dpp_xor_partner(v, 4);
  v += dpp_xor_partner(v, 8);
  v += lane_xor16_partner(v);
  return v;
}
__device__ __forceinline__ float fast_max32(float v) {
  v = fmaxf(v, dpp_xor_partner(v, 1));
  v = fmaxf(v, dpp_xor_partner(v, 2));
  v = fmaxf(v, dpp_xor_partner(v, 4));
  v = fmaxf(v, dpp_xor_partner(v, 8));
  v = fmaxf(v, lane_xor16_partner(v));
  return v;
}

// sum / max across the 32 lanes sharing lane >> 5 (the row of 128 floats a half wave holds as one float4 per lane)
__device__ __forceinline__ float half_sum32(float v) { return fast_sum32(v); }
__device__ __forceinline__ float half_max32(float v) { return fast_max32(v); }
__device__ __forceinline__ float lane_xor32_partner(float v) {
  const unsigned x = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float wave_sum64(float v) {
  v = half_sum32(v);
  v += lane_xor32_partner(v);
  return v;
}
__device__ __forceinline__ float wave_max64(float v) {
  v = half_max32(v);
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}
