// fp32 products on the bf16 matrix cores: x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (both round-to-nearest), and
//   a * b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi            (three v_mfma_*_bf16, fp32 accumulate)
// The dropped terms are <= 2^-16 |a b| per product (lo*lo and the rounding of lo), unbiased; sums over K = 128..9472
// land within ~1e-6 relative of the fp32 result.  One v_mfma_f32_32x32x16_bf16 does the work of 8 v_mfma_f32_32x32x2_f32
// in half their time, so the three passes cost 3/16 of the fp32 matrix-core time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// (x0, x1) -> packed bf16 pairs {lo16 = element 0, hi16 = element 1}: the high parts and the residuals
__device__ __forceinline__ void bf16_split_pair(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  const bf16x2_t h = __builtin_convertvector((f32x2_t){x0, x1}, bf16x2_t);        // v_cvt_pk_bf16_f32 (RNE)
  hi = __builtin_bit_cast(uint32_t, h);
  const float h0 = __builtin_bit_cast(float, hi << 16);
  const float h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
  const bf16x2_t l = __builtin_convertvector((f32x2_t){x0 - h0, x1 - h1}, bf16x2_t);
  lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ void bf16_split4(const float4& x, uint2& hi, uint2& lo) {
  bf16_split_pair(x.x, x.y, hi.x, lo.x);
  bf16_split_pair(x.z, x.w, hi.y, lo.y);
}

// ------------------------------------------------------------------------------------------------------
// fp16 variant of the same three-pass scheme, used by the forward / dX dense kernels ("f16x3"):
//   x' = x * 2^s (exact),  hi = f16(x'),  lo = f16(x' - hi)      ->  x' = hi + lo up to 2^-22 |x'|
// fp16 carries 11 significant bits per term (bf16: 8), so two terms represent a float32 to 22 bits and the three-pass
// product a_hi*b_hi + a_hi*b_lo + a_lo*b_hi is as accurate as an fp32 FMA chain (measured: 7e-8 rms relative at K = 128
// against 4.5e-6 for the bf16 split; end to end the whole-model gradients move from ~5e-4 to ~3e-5 of the fp64 oracle).
// fp16's narrow exponent range is handled by exact power-of-two scaling: every A row (per 128-deep K chunk) is scaled so
// that its largest magnitude lies in [2^13, 2^14) - no overflow (max 65504), and every element within 2^16 of the row
// maximum keeps a normal-range `lo` - and the weights by the fixed 2^HUAL_F16_WSCALE_LOG2 (|w| < 2^5 assumed; dense weights
// are O(0.1)).  The fp32 accumulator of the chunk is multiplied by the exact inverse when it is folded into the total.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
#define HUAL_F16_WSCALE_LOG2 10
#define HUAL_F16_WSCALE 1024.0f
#define HUAL_F16_WMAX 63.0f              // |w| beyond this does not fit the scaled fp16 image (65504 / 1024): flagged by the pack launch

__device__ __forceinline__ void f16_split_pair(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  const f16x2_t h = __builtin_convertvector((f32x2_t){x0, x1}, f16x2_t);          // round to nearest even
  hi = __builtin_bit_cast(uint32_t, h);
  const f32x2_t hf = __builtin_convertvector(h, f32x2_t);
  const f16x2_t l = __builtin_convertvector((f32x2_t){x0 - hf[0], x1 - hf[1]}, f16x2_t);   // x - hi is exact in fp32
  lo = __builtin_bit_cast(uint32_t, l);
}
// The same split of (x0 s, x1 s) with the scale folded in, on the mixed-precision FMA: v_fma_mixlo_f16 / v_fma_mixhi_f16 evaluate
// f16(a * b + c) with every source either float32 or one half of a register - hi = f16(x s + 0), lo = f16(x s - hi), the fma exact in
// float32 (s is a power of two), ONE rounding each: bit-identical to f16_split_pair(x0 * s, x1 * s, ..) (scripts/exp/mix_split_test.hip)
// in FOUR vector instructions per pair instead of seven (v_pk_mul, v_cvt_pk_f16_f32, 2 v_cvt_f32_f16, v_pk_fma, v_cvt_pk_f16_f32), none
// of them packed-f32 (an anti-lever beside MFMAs, MI355X_MICROARCH.md: the scale multiplications alone cost the weight-gradient launch 18 us).
// ONLY where the halves go to memory (LDS / global stores): as MFMA operands they need the wait states of a VALU write in front of a matrix
// instruction, which the compiler's hazard recogniser cannot place behind inline asm (the attention forward computed wrong products at
// Tk = 256 that way).  Measured per kernel (same-box A/B): weight gradients 159 -> 144 us, attention forward panels -1.8 us; the fused
// chains, conv_block and the context-query kernels were 1-2 % SLOWER with it (their splits sit in phases that are not issue bound) and
// keep the conversion sequence.
__device__ __forceinline__ void f16_split_pair_s(float x0, float x1, float s, uint32_t& hi, uint32_t& lo) {
  uint32_t h, l;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(x0), "v"(s));      // (the upper half is written next)
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
  hi = h; lo = l;
}
__device__ __forceinline__ void f16_split4_s(const float4& x, float s, uint2& hi, uint2& lo) {
  f16_split_pair_s(x.x, x.y, s, hi.x, lo.x);
  f16_split_pair_s(x.z, x.w, s, hi.y, lo.y);
}
__device__ __forceinline__ uint32_t f16_pack2(float x0, float x1) {      // v_cvt_pk_f16_f32, round to nearest even
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){x0, x1}, f16x2_t));
}
__device__ __forceinline__ void f16_split4(const float4& x, uint2& hi, uint2& lo) {
  f16_split_pair(x.x, x.y, hi.x, lo.x);
  f16_split_pair(x.z, x.w, hi.y, lo.y);
}
__device__ __forceinline__ float f4absmax(const float4& v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
// power-of-two scale that brings a row maximum rmax >= 0 into [2^13, 2^14) (rmax below 2^-100, incl. 0: scale 2^113, still
// finite); inv = 1 / (scale * 2^HUAL_F16_WSCALE_LOG2), both exact powers of two
__device__ __forceinline__ float f16_row_scale(float rmax, float& inv) {
  uint32_t eb = (__float_as_uint(rmax) >> 23) & 0xffu;
  eb = eb < 27u ? 27u : (eb > 254u ? 254u : eb);
  inv = __uint_as_float((eb - 13u - (uint32_t)HUAL_F16_WSCALE_LOG2) << 23);
  return __uint_as_float((267u - eb) << 23);
}
__device__ __forceinline__ float4 f4scale1(const float4& v, float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }
__device__ __forceinline__ f16x8 join_tr_f16(s16x4 a, s16x4 b);

// Byte offset of 16-byte chunk `ch` (0..15) of row `row` in a [rows][128 x bf16] LDS tile with plain 256-byte rows and the
// XOR swizzle that keeps BOTH ds_read_b128 row reads and ds_read_b64_tr_b16 transposed reads conflict free
// (cdna_hip_programming.md T10, image (b)).
__device__ __forceinline__ int tile256_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// ds_read_b64_tr_b16: per group of 16 lanes, a block of 4 rows x 16 columns of 16-bit elements is delivered column-major:
// lane 4q+p of the group supplies the address of row q, columns 4p..4p+3; lane i receives column i, rows 0..3.
// EXEC must be all ones.
__device__ __forceinline__ s16x4 lds_read_tr16(const char* lds_base, int byte_off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_base + byte_off));
}
__device__ __forceinline__ bf16x8 join_tr(s16x4 a, s16x4 b) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f16x8 join_tr_f16(s16x4 a, s16x4 b) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(f16x8, v);
}
