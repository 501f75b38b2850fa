// One-shot all-reduce of the flat gradient bucket over peer mappings (absent in the reference, which pins one GPU:
// /root/reference/utils/runner_utils.py:11).  The step's one collective is a 4.75 MB sum - far below the size where a ring's bandwidth
// matters, so RCCL's cost there is hop latency x (world - 1) steps x 2 phases.  xGMI is point to point: every GPU can read every peer's
// memory directly, so the whole exchange is ONE launch per rank with two flag barriers:
//
//   barrier A   every rank's bucket is final (its backward pass has ended: stream order in front of this launch)
//   reduce-scatter   rank r sums slice r of ALL buckets, read straight from the peers' memory, in rank order 0 .. world-1
//                    (one rank sums a slice, everybody copies it: all ranks end with identical bits) -> its own scratch slice
//   barrier B   every scratch slice is written, nobody reads a bucket any more
//   all-gather       rank r copies every peer's scratch slice into its own bucket
//
// The scratch slices are not touched again before the NEXT call's barrier A, which a peer only reaches after this call's launch has
// ended on its stream - so two barriers per call are enough.  Flags: one 32-bit word per (phase, peer) in UNCACHED device memory of
// every rank (hual_xgmi_flags_alloc: PyTorch's allocator only hands out cached, coarse-grained memory, which a running kernel may not
// see a peer's write to), written by the peer with a system-scope release store, polled with system-scope acquire loads; the value is
// the call's sequence number, kept in a device word so that a captured hipGraph advances it on every replay.  Every spin is bounded:
// a rank that waits longer than ~2 s of polls sets status[0] and ALL its waves leave the kernel (the result is then garbage and the
// host raises when it looks at the status word) - a lost peer must not wedge the GPU.
//
// The grid is small on purpose (HUAL_XGMI_BLOCKS workgroups): all of them must be resident at once, because they spin together.
#include <string.h>
#include "common.h"
#include "prof.h"

using namespace hual;

#define HUAL_XGMI_BLOCKS 64
#define HUAL_XGMI_THREADS 256
#define HUAL_XGMI_MAX_WORLD 16
#define HUAL_XGMI_SPIN_LIMIT (1u << 22)      // polls per wait (each a system-scope load + s_sleep, ~1 us): seconds, then give up

struct XgmiArgs {
  int rank, world;
  float* flat[HUAL_XGMI_MAX_WORLD];          // every rank's bucket (own entry: the local pointer)
  float* scratch[HUAL_XGMI_MAX_WORLD];       // every rank's reduced slice [chunk]
  uint32_t* flags[HUAL_XGMI_MAX_WORLD];      // every rank's flag words [2][HUAL_XGMI_MAX_WORLD] (+ [32]: done counter of the local grid)
  uint32_t* seq;                             // local device word: sequence number of the previous call
  uint32_t* status;                          // local device word: 0 ok, 1 a wait timed out
  size_t n, chunk;                           // floats in the bucket (multiple of 4), floats per slice (multiple of 4)
};

__device__ __forceinline__ void xgmi_signal(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ bool xgmi_wait(const uint32_t* p, uint32_t v) {
  for (uint32_t it = 0; it < HUAL_XGMI_SPIN_LIMIT; ++it) {
    // (signed distance: the sequence number may wrap)
    if ((int32_t)(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - v) >= 0) return true;
    __builtin_amdgcn_s_sleep(8);
  }
  return false;
}
// all threads of the block: wait until every peer has signalled `v` in phase `ph` of the LOCAL flag words
__device__ __forceinline__ bool xgmi_block_wait(const XgmiArgs& a, int ph, uint32_t v) {
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  if ((int)threadIdx.x < a.world && (int)threadIdx.x != a.rank)
    if (!xgmi_wait(a.flags[a.rank] + ph * HUAL_XGMI_MAX_WORLD + threadIdx.x, v)) { ok = 0; a.status[0] = 1u; }
  __syncthreads();
  const bool r = ok != 0;
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // what the peers wrote before their signal is visible to every load behind this point
  return r;
}

__global__ __launch_bounds__(HUAL_XGMI_THREADS) void xgmi_allreduce_kernel(XgmiArgs a) {
  const uint32_t v = a.seq[0] + 1u;                  // (advanced by xgmi_seq_kernel, the next launch on the stream)
  uint32_t* done = a.flags[a.rank] + 2 * HUAL_XGMI_MAX_WORLD;
  // ---- barrier A
  if (blockIdx.x == 0 && (int)threadIdx.x < a.world && (int)threadIdx.x != a.rank)
    xgmi_signal(a.flags[threadIdx.x] + 0 * HUAL_XGMI_MAX_WORLD + a.rank, v);
  bool ok = xgmi_block_wait(a, 0, v);
  // ---- reduce-scatter: my slice of every bucket, summed in rank order
  const size_t s0 = (size_t)a.rank * a.chunk, s1 = s0 + a.chunk < a.n ? s0 + a.chunk : a.n;
  if (ok) {
    for (size_t i = s0 + 4 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x); i < s1; i += 4 * (size_t)gridDim.x * blockDim.x) {
      float4 acc = *reinterpret_cast<const float4*>(a.flat[0] + i);
      for (int p = 1; p < a.world; ++p) {
        const float4 x = *reinterpret_cast<const float4*>(a.flat[p] + i);
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
      }
      *reinterpret_cast<float4*>(a.scratch[a.rank] + (i - s0)) = acc;
    }
  }
  // ---- barrier B: the LAST workgroup of this rank to finish the phase signals the peers
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __syncthreads();
  __shared__ int last;
  if (threadIdx.x == 0) last = (atomicAdd(done, 1u) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (last) {
    if (threadIdx.x == 0) *done = 0u;                // (nobody adds again before the next call)
    if ((int)threadIdx.x < a.world && (int)threadIdx.x != a.rank)
      xgmi_signal(a.flags[threadIdx.x] + 1 * HUAL_XGMI_MAX_WORLD + a.rank, v);
  }
  ok = xgmi_block_wait(a, 1, v) && ok;
  // ---- all-gather: every reduced slice into my bucket
  if (ok) {
    for (int p = 0; p < a.world; ++p) {
      const size_t p0 = (size_t)p * a.chunk, p1 = p0 + a.chunk < a.n ? p0 + a.chunk : a.n;
      for (size_t i = p0 + 4 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x); i < p1; i += 4 * (size_t)gridDim.x * blockDim.x)
        *reinterpret_cast<float4*>(a.flat[a.rank] + i) = *reinterpret_cast<const float4*>(a.scratch[p] + (i - p0));
    }
  }
}
__global__ void xgmi_seq_kernel(uint32_t* seq) { seq[0] += 1u; }

extern "C" {

// ---- setup helpers (NOT the hot path: they own the flag words because the caller's allocator cannot provide uncached device memory)
uint64_t hual_xgmi_flags_bytes(void) { return 4 * (2 * HUAL_XGMI_MAX_WORLD + 32); }
int hual_xgmi_flags_alloc(void** p) {
  HUAL_REQUIRE(p != nullptr, "hual_xgmi_flags_alloc: null");
  HUAL_CHECK_HIP(hipExtMallocWithFlags(p, hual_xgmi_flags_bytes(), hipDeviceMallocUncached));
  HUAL_CHECK_HIP(hipMemset(*p, 0, hual_xgmi_flags_bytes()));
  HUAL_CHECK_HIP(hipDeviceSynchronize());
  return 0;
}
int hual_xgmi_flags_free(void* p) {
  if (p) HUAL_CHECK_HIP(hipFree(p));
  return 0;
}
// p may lie INSIDE an allocation (a tensor of a caching allocator's segment): the handle is taken for the allocation's base and
// *offset is p's distance from it - the opener adds it to the pointer hual_xgmi_ipc_open returns
int hual_xgmi_ipc_export(void* p, void* handle64, uint64_t* offset) {
  HUAL_REQUIRE(p && handle64 && offset, "hual_xgmi_ipc_export: null");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  HUAL_CHECK_HIP(hipMemGetAddressRange(&base, &size, p));
  HUAL_CHECK_HIP(hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), base));
  *offset = (uint64_t)((char*)p - (char*)base);
  return 0;
}
int hual_xgmi_ipc_open(const void* handle64, void** p) {
  HUAL_REQUIRE(p && handle64, "hual_xgmi_ipc_open: null");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  HUAL_CHECK_HIP(hipIpcOpenMemHandle(p, h, hipIpcMemLazyEnablePeerAccess));
  return 0;
}
int hual_xgmi_ipc_close(void* p) {
  if (p) HUAL_CHECK_HIP(hipIpcCloseMemHandle(p));
  return 0;
}

// ---- the collective.  flat / scratch / flags: `world` DEVICE pointers each (entry `rank` = this rank's own memory, the others peer
// mappings); seq, status: local device words (seq zero-initialised once, the same on every rank).  n floats per bucket, n % 4 == 0.
int hual_xgmi_allreduce(int rank, int world, void* const* flat, void* const* scratch, void* const* flags, uint32_t* seq, uint32_t* status,
                        uint64_t n, uint64_t scratch_floats, void* stream) {
  HUAL_REQUIRE(world >= 1 && world <= HUAL_XGMI_MAX_WORLD && rank >= 0 && rank < world, "hual_xgmi_allreduce: rank / world");
  HUAL_REQUIRE(flat && scratch && flags && seq && status && n > 0 && (n % 4) == 0, "hual_xgmi_allreduce: null pointer or n % 4 != 0");
  XgmiArgs a{};
  a.rank = rank; a.world = world; a.seq = seq; a.status = status; a.n = (size_t)n;
  a.chunk = ((((size_t)n + world - 1) / world) + 3) & ~(size_t)3;
  HUAL_REQUIRE(scratch_floats >= a.chunk, "hual_xgmi_allreduce: scratch smaller than a slice (ceil(n / world) rounded up to 4 floats)");
  for (int p = 0; p < world; ++p) {
    HUAL_REQUIRE(flat[p] && scratch[p] && flags[p] && ((uintptr_t)flat[p] & 15) == 0 && ((uintptr_t)scratch[p] & 15) == 0,
                 "hual_xgmi_allreduce: null / unaligned peer pointer");
    a.flat[p] = (float*)flat[p]; a.scratch[p] = (float*)scratch[p]; a.flags[p] = (uint32_t*)flags[p];
  }
  const double bytes = 4.0 * ((double)a.chunk * (world + 1) + 2.0 * (double)n);
  HUAL_LAUNCH(0.0, bytes, xgmi_allreduce_kernel, dim3(HUAL_XGMI_BLOCKS), dim3(HUAL_XGMI_THREADS), 0, (hipStream_t)stream, a);
  HUAL_LAUNCH(0.0, 0.0, xgmi_seq_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, seq);
  HUAL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
