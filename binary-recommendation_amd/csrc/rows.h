// Row-granular access helpers: a *row group* of LPR lanes (power of two <= 64) owns one
// embedding row; each lane moves VEC floats (16/8/4 bytes) per access so a 256-B row
// (dim 64) is one coalesced 16-lane x 16-B request and a wave covers 4 rows per instruction.
#pragma once
#include "common.h"

namespace br {

template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<1> { using type = float; };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vload(const float* p) {
  return *reinterpret_cast<const typename VecT<VEC>::type*>(p);
}
template <int VEC>
__device__ __forceinline__ void vstore(float* p, typename VecT<VEC>::type v) {
  *reinterpret_cast<typename VecT<VEC>::type*>(p) = v;
}
template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vzero();
template <> __device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float2 vzero<2>() { return make_float2(0.f, 0.f); }
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }

__device__ __forceinline__ float4 vmul(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float2 vmul(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
__device__ __forceinline__ float vmul(float a, float s) { return a * s; }
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float2 vadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vsub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float2 vsub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float vsub(float a, float b) { return a - b; }
// the order is spelled out (one product, then fused multiply-adds in column order): kernels with different lane layouts reproduce
// it and give the same bits (neumf_embed_fwd_deferred_wave_kernel chains it across two lanes)
__device__ __forceinline__ float vdot(float4 a, float4 b) {
  return __builtin_fmaf(a.w, b.w, __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ float vdot(float2 a, float2 b) { return __builtin_fmaf(a.y, b.y, a.x * b.x); }
__device__ __forceinline__ float vdot(float a, float b) { return a * b; }

// Register pins: an empty asm the compiler cannot move or drop.  pin(v) = "the value has to be in its registers here" - put behind
// a batch of independent loads it makes them one batch (all issued, then waited for at once) where the compiler would otherwise
// sink each load to its first use behind a branch.
#if defined(__HIP_DEVICE_COMPILE__)
#define BR_PIN_S(x) asm volatile("" : "+s"(x))
#define BR_PIN_V(x) asm volatile("" : "+v"(x))
#else          // the host pass parses device functions too and knows no "s" / "v" registers
#define BR_PIN_S(x) (void)(x)
#define BR_PIN_V(x) (void)(x)
#endif
__device__ __forceinline__ void pin(float& a) { BR_PIN_V(a); }
__device__ __forceinline__ void pin(float2& a) { BR_PIN_V(a.x); BR_PIN_V(a.y); }
__device__ __forceinline__ void pin(float4& a) { BR_PIN_V(a.x); BR_PIN_V(a.y); BR_PIN_V(a.z); BR_PIN_V(a.w); }

// Host-side geometry of a row group for a given embedding dim.
struct RowGeom {
  int vec;       // floats per lane access: 4, 2 or 1
  int chunks;    // dim / vec
  int lpr_log2;  // log2(lanes per row), lanes per row = min(64, pow2ceil(chunks))
};
static inline RowGeom row_geom(int dim) {
  RowGeom g;
  g.vec = (dim % 4 == 0) ? 4 : (dim % 2 == 0) ? 2 : 1;
  g.chunks = dim / g.vec;
  int l = 0;
  while ((1 << l) < g.chunks && l < 6) ++l;
  g.lpr_log2 = l;
  return g;
}

// same, when rows are read with a row stride `ld` (floats): the vector width must divide both.
static inline RowGeom row_geom_ld(int dim, int64_t ld) {
  RowGeom g;
  g.vec = (dim % 4 == 0 && ld % 4 == 0) ? 4 : (dim % 2 == 0 && ld % 2 == 0) ? 2 : 1;
  g.chunks = dim / g.vec;
  int l = 0;
  while ((1 << l) < g.chunks && l < 6) ++l;
  g.lpr_log2 = l;
  return g;
}

// sum over the LPR lanes of a row group (LPR runtime power of two <= 64)
__device__ __forceinline__ float rowgroup_sum(float v, int lpr) {
  for (int off = lpr >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

#define BR_DISPATCH_VEC(vec, ...)                                        \
  do {                                                                   \
    if ((vec) == 4) { constexpr int VEC = 4; __VA_ARGS__; }              \
    else if ((vec) == 2) { constexpr int VEC = 2; __VA_ARGS__; }         \
    else { constexpr int VEC = 1; __VA_ARGS__; }                         \
  } while (0)

}  // namespace br
