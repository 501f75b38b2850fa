// Device Philox4x32-7: the dropout generator shared with oracle/philox.py (same counter layout).  SEVEN rounds (round 4; the first
// three rounds of the build used ten): Philox4x32-7 is the fewest-round variant Salmon et al. (SC'11) report as passing BigCrush
// ("Crush-resistant"); Random123 ships known-answer vectors for it, which oracle/philox.py is checked against (tests/test_oracle.py).
// The rounds are the cost of the generator - two quarter-rate 32 x 32 -> 64 multiplies each - and 30 % fewer of them pay for the 16-bit
// keep decisions (exact 1 / (1 - rate) scale) the attention-probability sites now use.
//   c0 = col >> 2, c1 = row, c2 = site, c3 = offset ; key = seed.  Output w belongs to column 4*c0 + w.
// Replaces TensorFlow's stateful RNG behind tf.nn.dropout (/root/reference/models/modules.py:15,27,69,83-88,
// 131-139; layers.py:86,91; ops.py:104; model.py:47) - see DESIGN.md "Dropout".
#pragma once
#include "common.h"

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

#define HUAL_PHILOX_ROUNDS 7

struct uint4_ { uint32_t x, y, z, w; };

__device__ __forceinline__ uint4_ philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                uint32_t k1) {
#pragma unroll
  for (int i = 0; i < HUAL_PHILOX_ROUNDS; ++i) {
    // one 32x32->64 multiply (v_mad_u64_u32) per product instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: integer
    // multiplies are quarter rate on CDNA4 and dominate the cost of the generator
    const uint64_t p0 = (uint64_t)PHILOX_M0 * c0, p1 = (uint64_t)PHILOX_M1 * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += PHILOX_W0;
    k1 += PHILOX_W1;
  }
  uint4_ r = {c0, c1, c2, c3};
  return r;
}

// multiplicative keep-mask {0, scale} for the 4 columns [4*c0, 4*c0+3] of `row` at call-site `site`
// word i of the dropout stream's state (key lo, key hi, step): through the constant address space - scalar loads out of the scalar
// cache (the state is only written by the optimizer launch); as vector loads every call waited for a memory access of its own
__device__ __forceinline__ uint32_t drop_state(const hual::DropCfg& d, int i) {
  return ((const __attribute__((address_space(4))) uint32_t*)(uintptr_t)d.state)[i];
}
__device__ __forceinline__ float4 drop_mask4(const hual::DropCfg& d, uint32_t site, uint32_t row, uint32_t col4) {
  uint4_ r = philox4x32(col4, row, site, drop_state(d, 2), drop_state(d, 0), drop_state(d, 1));
  float4 m;
  m.x = r.x < d.thresh ? d.scale : 0.f;
  m.y = r.y < d.thresh ? d.scale : 0.f;
  m.z = r.z < d.thresh ? d.scale : 0.f;
  m.w = r.w < d.thresh ? d.scale : 0.f;
  return m;
}

// the same draw as 4 keep bits (bit c = column 4*col4 + c is kept).  Forward kernels store this byte so that backward
// kernels do not have to repeat the Philox rounds (40 quarter-rate integer multiplies per call on CDNA4).
__device__ __forceinline__ uint32_t drop_bits4(const hual::DropCfg& d, uint32_t site, uint32_t row, uint32_t col4) {
  uint4_ r = philox4x32(col4, row, site, drop_state(d, 2), drop_state(d, 0), drop_state(d, 1));
  return (r.x < d.thresh ? 1u : 0u) | (r.y < d.thresh ? 2u : 0u) | (r.z < d.thresh ? 4u : 0u) | (r.w < d.thresh ? 8u : 0u);
}
__device__ __forceinline__ float4 mask_from_bits4(uint32_t bits, float scale) {
  return make_float4((bits & 1u) ? scale : 0.f, (bits & 2u) ? scale : 0.f, (bits & 4u) ? scale : 0.f, (bits & 8u) ? scale : 0.f);
}

__device__ __forceinline__ float4 apply_drop4(const hual::DropCfg& d, uint32_t site, uint32_t row, uint32_t col4,
                                              float4 v) {
  float4 m = drop_mask4(d, site, row, col4);
  v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
  return v;
}
