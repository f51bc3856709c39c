// Shared pieces of the split-f16 kernels (sr3d_hconv.hip, sr3d_hconv_s2.hip, sr3d_hwgrad.hip): fp32 operands as two fp16
// halves on v_mfma_f32_32x32x16_f16.  gfx950 only.
#pragma once
#include "sr3d_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// store one result as the activation element type (BF: bfloat16, round to nearest even)
template <bool BF>
__device__ __forceinline__ void st_act(float* base, long long idx, float v) {
  if constexpr (BF)
    reinterpret_cast<__bf16*>(base)[idx] = (__bf16)v;
  else
    base[idx] = v;
}
typedef __attribute__((address_space(3))) void* lds_p;

// Exponent s with amax * 2^s in [2^13, 2^14): the largest element then splits into hi = fp16(a), lo = fp16(a - hi) with
// |a - hi - lo| <= 2^-22 |a| and cannot overflow.  kSplitScaleNone for amax = 0 ("no opinion": running minima keep
// their value), 0 for inf / NaN.
constexpr int kSplitScaleNone = 120;
__host__ __device__ inline int split_scale_exp(float amax) {
  if (amax != amax || amax > 3.0e38f) return 0;
  if (!(amax > 0.f)) return kSplitScaleNone;
  int e;
  frexpf(amax, &e);   // amax = m * 2^e, m in [0.5, 1)
  const int s = 14 - e;
  return s > kSplitScaleNone ? kSplitScaleNone : s;
}

// 16 bytes per lane, global -> LDS, buffer form.  (The buffer form on purpose: global_load_lds is a FLAT instruction
// touching two address spaces, and while one is pending hipcc turns every LDS wait of the wave into lgkmcnt(0).  Device
// pass only: the host pass of hipcc rejects the 16-byte size of this builtin -- it checks it against the host's
// feature set -- and then drops the calling kernel's stub without a message.)
__device__ __forceinline__ void split_lds_dma16(__amdgpu_buffer_rsrc_t rs, lds_p dst, int voffset) {
#if __HIP_DEVICE_COMPILE__
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voffset, 0, 0, 0);
#endif
}

// keeps a wave-uniform value in a scalar register and opaque to the optimiser
template <typename T>
__device__ __forceinline__ void split_pin_scalar(T& x) {
  asm volatile("" : "+s"(x));
}

__device__ __forceinline__ float split_act(float v, int act) {
  if (act == SR3D_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == SR3D_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
  return v;
}

// The split of TWO elements: hi = fp16(v * mult), lo = fp16(v * mult - hi), packed (a in the low halves): two v_fma_mix
// per element, written straight into the halves of the packed dwords.  (hipcc turns the plain C form -- multiply, convert,
// convert back, subtract, convert -- into 3 instructions per element, a third of them packed fp32 (v_pk_mul_f32 /
// v_pk_fma_f32): each of those costs ~20 cycles of vector issue beside MFMAs, and an MFMA 16x16x32 leaves the SIMD's
// vector issue free for only 8 of its 16 cycles -- MI355X_MICROARCH.md, cycle constants.)  Same values as the C form: the
// product by a power of two is exact, so each half is rounded once.
__device__ __forceinline__ void split_pair(const float a, const float b, const float mult, unsigned& hi, unsigned& lo) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(a), "v"(mult));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(b), "v"(mult));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(a), "v"(mult), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(b), "v"(mult), "v"(hi));
}
// ... of 8 elements v[0], v[st], ..., v[7 st] into one 16-byte piece per part
__device__ __forceinline__ void split_piece(const float* v, const int st, const float mult, h8& hi, h8& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int k = 0; k < 4; k++) split_pair(v[2 * k * st], v[(2 * k + 1) * st], mult, h[k], l[k]);
  hi = __builtin_bit_cast(h8, u32x4{h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(h8, u32x4{l[0], l[1], l[2], l[3]});
}

// Largest value of a non-negative float over the 64 lanes of the wave, returned wave-uniform.  Four DPP steps inside the
// rows of 16 lanes (quad swaps, half-row mirror, row mirror) and four lane reads: vector-ALU only.  (__shfl_xor is a
// ds_bpermute per step: six dependent trips through the LDS pipe, ~700 cycles in front of a barrier.)
__device__ __forceinline__ float split_wave_max(float m) {
// (old = the value itself: every lane has a valid source in these four patterns, and a constant would cost a v_mov per step)
#define SR3D_DPP_MAX(ctrl) m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m), __builtin_bit_cast(int, m), ctrl, 0xf, 0xf, false)))
  SR3D_DPP_MAX(0xB1);    // quad_perm [1, 0, 3, 2]
  SR3D_DPP_MAX(0x4E);    // quad_perm [2, 3, 0, 1]
  SR3D_DPP_MAX(0x141);   // row_half_mirror
  SR3D_DPP_MAX(0x140);   // row_mirror
#undef SR3D_DPP_MAX
  const int mi = __builtin_bit_cast(int, m);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(mi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(mi, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(mi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(mi, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// max |x| of a tensor into *slot with atomicMax (bits of a non-negative float order like unsigned integers; the caller
// zeroes the slot).  Defined in sr3d_hconv.hip.
int sr3d_absmax_launch(const float* x, long long n, unsigned* slot, hipStream_t st);
