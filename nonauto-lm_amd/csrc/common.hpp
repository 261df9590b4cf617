// Shared device/host helpers for libmivae (gfx950 / CDNA4 only, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

#define MIVAE_OK 0
#define MIVAE_EINVAL (-1)
#define MIVAE_ELAUNCH (-2)
#define MIVAE_EUNSUPPORTED (-3)

#define MIVAE_F32 0
#define MIVAE_BF16 1

void mivae_set_error(const char* fmt, ...);

#define MIVAE_REQUIRE(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      mivae_set_error(__VA_ARGS__);              \
      return MIVAE_EINVAL;                       \
    }                                            \
  } while (0)

#define MIVAE_CHECK_LAUNCH(name)                                                  \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) {                                                       \
      mivae_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
      return MIVAE_ELAUNCH;                                                       \
    }                                                                             \
  } while (0)

namespace mivae {

constexpr float kLog2Pi = 1.8378770664093453f;  // log(2*pi)
constexpr float kInvSqrt2 = 0.70710678118654752f;
constexpr float kInvSqrt2Pi = 0.3989422804014327f;

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_to_f32(unsigned short u) {
  return __uint_as_float(((unsigned)u) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  bf16_t b = (bf16_t)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(unsigned short, b);
}

// ---- typed row access: VEC contiguous elements <-> float registers -----------------------------
template <typename T, int VEC>
struct RowIO;

template <>
struct RowIO<float, 4> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    f32x4 x = *reinterpret_cast<const f32x4*>(p);
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    f32x4 x = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = x;
  }
};
template <>
struct RowIO<float, 1> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
  static __device__ __forceinline__ void store(float* p, const float (&v)[1]) { *p = v[0]; }
};
template <>
struct RowIO<unsigned short, 8> {
  static __device__ __forceinline__ void load(const unsigned short* p, float (&v)[8]) {
    u16x8 x = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = bf16_to_f32(x[i]);
  }
  static __device__ __forceinline__ void store(unsigned short* p, const float (&v)[8]) {
    u16x8 x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = f32_to_bf16(v[i]);
    *reinterpret_cast<u16x8*>(p) = x;
  }
};
template <>
struct RowIO<unsigned short, 4> {
  static __device__ __forceinline__ void load(const unsigned short* p, float (&v)[4]) {
    u16x4 x = *reinterpret_cast<const u16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = bf16_to_f32(x[i]);
  }
  static __device__ __forceinline__ void store(unsigned short* p, const float (&v)[4]) {
    u16x4 x;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = f32_to_bf16(v[i]);
    *reinterpret_cast<u16x4*>(p) = x;
  }
};
template <>
struct RowIO<unsigned short, 1> {
  static __device__ __forceinline__ void load(const unsigned short* p, float (&v)[1]) {
    v[0] = bf16_to_f32(*p);
  }
  static __device__ __forceinline__ void store(unsigned short* p, const float (&v)[1]) {
    *p = f32_to_bf16(v[0]);
  }
};

// Sum over the `width` consecutive lanes of a lane group (width = power of two <= 64).
__device__ __forceinline__ float group_sum(float v, int width) {
  for (int off = width >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Lanes cooperating on one token row: smallest power of two covering D/VEC chunks, capped at 64.
static inline int lanes_per_row(int64_t D, int vec) {
  int64_t chunks = (D + vec - 1) / vec;
  int l = 1;
  while (l < chunks && l < 64) l <<= 1;
  return l;
}

// ---- transposing LDS read (gfx950 ds_read_b64_tr_b16) ------------------------------------------
// MFMA fragment whose 8 elements per lane run along the SLOW index of a row-major bf16 LDS tile: f[j] =
// tile[k0 + 8 hh + j][fbase + r32] (A or B operand of v_mfma_f32_32x32x16_bf16 contracting over tile rows).  Per
// 16-lane group the instruction delivers a 4-row x 16-column block column-major.  EXEC must be all ones.
typedef __bf16 bf16x4_tr __attribute__((__vector_size__(4 * sizeof(__bf16))));
__device__ __forceinline__ bf16x8 lds_tr_frag(const unsigned short* tile, int stride, int fbase, int k0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const unsigned short* a0 = tile + (k0 + 8 * (g >> 1) + q) * stride + fbase + 16 * (g & 1) + 4 * p;
  typedef __attribute__((address_space(3))) bf16x4_tr* lds_ptr;
  const bf16x4_tr v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(a0));
  const bf16x4_tr v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(a0 + 4 * stride));
  return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace mivae
