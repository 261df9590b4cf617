// Shared device helpers of the NICE stack kernels (nice_stack.hip: latency-bound small-D kernels;
// nice_big.hip: MFMA-bound kernels for D >= 256): activations, MFMA policies, fragment / accumulator I/O.
#pragma once
#include "common.hpp"

namespace mivae {

enum Act { ACT_LINEAR = 0, ACT_GELU = 1, ACT_SWISH = 2, ACT_RELU = 3, ACT_TANH = 4, ACT_SIGMOID = 5 };

// ---- fast transcendental helpers (v_exp_f32 based) -------------------------------------------------
// exp(x) = 2^(x log2 e) on the hardware exp2 unit: relative error ~1e-7 (1 + |x|) -- well inside the
// 1e-5 parity budget for |s| < ~30, far cheaper than the ocml range-reduced expf.
static __device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// erf(x) = sign(x) (1 - 2^{a q(a)}), a = min(|x|, A), q = weighted least-squares fit of log2(erfc(a)) / a
// on [0, A].  Branch free, one v_exp_f32.  Reference: x * 0.5 * (1 + erf(x / sqrt 2)), nn/activation.py:9-16.
//   LOWP = false (fp32 latents): degree 7, A = 4: max abs error 1.1e-7 in fp32 arithmetic (GELU <= 1.0e-7)
//   LOWP = true  (bf16 latents): degree 3, A = 3.2: max abs error 1.6e-5 (GELU <= 3e-5), two orders of
//     magnitude under the bf16 rounding the hidden activations get before the second MFMA.
template <bool LOWP>
__device__ __forceinline__ float fast_erf(float x) {
  float q, a;
  if constexpr (LOWP) {
    a = fminf(fabsf(x), 3.2f);
    q = 0.019193846f;
    q = fmaf(q, a, -0.136759108f);
    q = fmaf(q, a, -0.923531482f);
    q = fmaf(q, a, -1.6272781f);
  } else {
    a = fminf(fabsf(x), 4.0f);
    q = -4.471107996e-05f;
    q = fmaf(q, a, 4.402441469e-04f);
    q = fmaf(q, a, -1.472712344e-03f);
    q = fmaf(q, a, -8.012262794e-04f);
    q = fmaf(q, a, 2.827611896e-02f);
    q = fmaf(q, a, -1.484914114e-01f);
    q = fmaf(q, a, -9.184144181e-01f);
    q = fmaf(q, a, -1.627908727e+00f);
  }
  const float r = 1.0f - __builtin_amdgcn_exp2f(q * a);
  return copysignf(r, x);
}

// activation applied to a whole accumulator tile.  ACT = ACT_GELU is the specialised default of the
// reference (couplings.py:32); ACT = -1 dispatches on the wave-uniform runtime id.
template <int ACT, bool LOWP>
__device__ __forceinline__ void act_tile(int act, f32x16& a) {
  if constexpr (ACT == ACT_GELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float hx = 0.5f * a[r];
      a[r] = fmaf(hx, fast_erf<LOWP>(a[r] * kInvSqrt2), hx);
    }
  } else {
    switch (act) {
      case ACT_GELU:
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = a[r] * 0.5f * (1.0f + fast_erf<LOWP>(a[r] * kInvSqrt2));
        break;
      case ACT_SWISH:
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = a[r] / (1.0f + fast_exp(-a[r]));
        break;
      case ACT_RELU:
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = a[r] > 0.f ? a[r] : 0.f;
        break;
      case ACT_TANH:
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = tanhf(a[r]);
        break;
      case ACT_SIGMOID:
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 1.0f / (1.0f + fast_exp(-a[r]));
        break;
      default:
        break;
    }
  }
}
// value (in place) and derivative
template <int ACT, bool LOWP>
__device__ __forceinline__ void act_tile_fd(int act, f32x16& a, f32x16& d) {
  if constexpr (ACT == ACT_GELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float x = a[r];
      const float cdf = fmaf(0.5f, fast_erf<LOWP>(x * kInvSqrt2), 0.5f);
      a[r] = x * cdf;
      d[r] = fmaf(x * kInvSqrt2Pi, __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f), cdf);
    }
  } else {
    switch (act) {
      case ACT_GELU:
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float x = a[r];
          const float cdf = 0.5f * (1.0f + fast_erf<LOWP>(x * kInvSqrt2));
          a[r] = x * cdf;
          d[r] = cdf + x * kInvSqrt2Pi * fast_exp(-0.5f * x * x);
        }
        break;
      case ACT_SWISH:
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float x = a[r];
          const float sg = 1.0f / (1.0f + fast_exp(-x));
          a[r] = x * sg;
          d[r] = sg + x * sg * (1.0f - sg);
        }
        break;
      case ACT_RELU:
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float x = a[r];
          a[r] = x > 0.f ? x : 0.f;
          d[r] = x > 0.f ? 1.f : 0.f;
        }
        break;
      case ACT_TANH:
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float y = tanhf(a[r]);
          a[r] = y;
          d[r] = 1.0f - y * y;
        }
        break;
      case ACT_SIGMOID:
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float y = 1.0f / (1.0f + fast_exp(-a[r]));
          a[r] = y;
          d[r] = y * (1.0f - y);
        }
        break;
      default:
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 1.f;
        break;
    }
  }
}

// ---- MFMA policies ---------------------------------------------------------------------------------
// C/D layout of both (guide section 3): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
__host__ __device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

struct MmaBF16 {
  typedef bf16x8 AF;  // A fragment: row = lane & 31, k = 8 * (lane >> 5) + j
  typedef bf16x8 BF;  // B fragment: col = lane & 31, k = 8 * (lane >> 5) + j
  typedef unsigned short E;
  static constexpr int KS = 16;   // k per MFMA
  static constexpr int NS = 2;    // MFMAs per 32-deep k block
  static constexpr int EPF = 8;   // elements per lane per fragment
  static __device__ __forceinline__ f32x16 mma(AF a, BF b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  // accumulator tile X (rows in registers) -> B fragment of k-step s of a product summing over X's rows.
  // element j of lane half hh is row 16 s + 8 (j >> 2) + 4 hh + (j & 3) = acc_row(8 s + j, hh).
  template <int s>
  static __device__ __forceinline__ BF from_acc(const f32x16& x) {
    BF f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16_t)x[8 * s + j];
    return f;
  }
  static __host__ __device__ __forceinline__ int nat_k(int ks, int hh, int j) { return 16 * ks + 8 * hh + j; }
  static __host__ __device__ __forceinline__ int perm_k(int s, int hh, int j) { return acc_row(8 * s + j, hh); }
  static __device__ __forceinline__ E enc(float v) { return f32_to_bf16(v); }
};

struct MmaF32 {
  typedef float AF;  // A: row = lane & 31, k = lane >> 5
  typedef float BF;  // B: col = lane & 31, k = lane >> 5
  typedef float E;
  static constexpr int KS = 2;
  static constexpr int NS = 16;
  static constexpr int EPF = 1;
  static __device__ __forceinline__ f32x16 mma(AF a, BF b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  template <int s>
  static __device__ __forceinline__ BF from_acc(const f32x16& x) { return x[s]; }
  static __host__ __device__ __forceinline__ int nat_k(int ks, int hh, int) { return 2 * ks + hh; }
  static __host__ __device__ __forceinline__ int perm_k(int s, int hh, int) { return acc_row(s, hh); }
  static __device__ __forceinline__ E enc(float v) { return v; }
};

// ---- per-lane I/O in fragment / accumulator layouts ------------------------------------------------
// x0 as B fragments (natural k order): lane (tok = lane & 31, hh) holds x0[tok][nat_k(ks, hh, j)].
template <typename T, bool VEC, int NKS>
__device__ __forceinline__ void load_x0_f32(const T* __restrict__ zrow, bool valid, int h, int hh,
                                            float (&x0f)[NKS]) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int f = 2 * ks + hh;
    float v = 0.f;
    if (valid && f < h) {
      float t1[1];
      RowIO<T, 1>::load(zrow + f, t1);
      v = t1[0];
    }
    x0f[ks] = v;
  }
}
template <typename T, bool VEC, int NKS>
__device__ __forceinline__ void load_x0_bf16(const T* __restrict__ zrow, bool valid, int h, int hh,
                                             bf16x8 (&x0f)[NKS]) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int f = 16 * ks + 8 * hh;
    u16x8 raw = {0, 0, 0, 0, 0, 0, 0, 0};
    if (VEC) {
      if (valid && f < h) raw = *reinterpret_cast<const u16x8*>(zrow + f);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (valid && f + j < h) raw[j] = ((const unsigned short*)zrow)[f + j];
    }
    x0f[ks] = __builtin_bit_cast(bf16x8, raw);
  }
}

// accumulator layout tile set: regs[t][r] <-> row[32 t + acc_row(r, hh)], groups of 4 contiguous.
template <typename T, bool VEC, int HT>
__device__ __forceinline__ void load_acc(const T* __restrict__ row, bool valid, int h, int hh,
                                         f32x16 (&x)[HT]) {
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = 32 * t + 8 * g + 4 * hh;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (VEC) {
        if (valid && f < h) RowIO<T, 4>::load(row + f, v);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (valid && f + i < h) {
            float t1[1];
            RowIO<T, 1>::load(row + f + i, t1);
            v[i] = t1[0];
          }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) x[t][4 * g + i] = v[i];
    }
}
template <typename T, bool VEC, int HT>
__device__ __forceinline__ void store_acc(T* __restrict__ row, bool valid, int h, int hh,
                                          const f32x16 (&x)[HT]) {
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = 32 * t + 8 * g + 4 * hh;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = x[t][4 * g + i];
      if (VEC) {
        if (valid && f < h) RowIO<T, 4>::store(row + f, v);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (valid && f + i < h) {
            float t1[1] = {v[i]};
            RowIO<T, 1>::store(row + f + i, t1);
          }
      }
    }
}

// Side outputs of the backward (consumed by the weight-gradient kernel) are stored "token-blocked":
// element (feature row f, token tau) of a section with F rows lives at ((tau / TB) * F + f) * TB + tau % TB,
// TB = 16 bytes worth of tokens (8 bf16 / 4 fp32).  The reader's MFMA fragment (row on the lane, TB
// consecutive tokens per lane) is then ONE fully coalesced 16-byte-per-lane load per operand.
// Store of one accumulator tile (rows row0 + acc_row(r, hh), token on the lane): `base`, `row0` are
// wave-uniform, `loff` = (tau / TB) * F * TB + tau % TB + 4 * hh * TB is the per-lane part (computed
// once per kernel): no per-store 64-bit VALU address arithmetic.
template <typename T>
__device__ __forceinline__ void store_tile_t(T* __restrict__ base, int row0, unsigned loff, const f32x16& x) {
  constexpr int TB = 16 / sizeof(T);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    T* rowp = base + (row0 + (r & 3) + 8 * (r >> 2)) * TB;
    float t1[1] = {x[r]};
    RowIO<T, 1>::store(rowp + loff, t1);
  }
}


// ==== shared by nice_stack.hip and nice_small.hip ==================================================
// ---- packed weight streams -------------------------------------------------------------------------
// The packed image is two linear streams of equally sized "task" regions laid out in exactly the
// order the kernels consume them, so staging is a pure sequential copy:
//   F stream (forward; steps reversed when packed with inverse = 1): task (kk, c, net), c = 32-wide hidden
//     chunk, net 0 = s, 1 = t (a pair per iteration: the s and t nets run on sibling waves):
//     [W1n: NKS frags][W2p: HT*NS frags][tail]                                                  (tbf bytes)
//   B stream (backward, steps K-1 .. 0): task (kk, phase, c); phase 0 = s-net forward recompute
//     ([W1n][W2p][pad]), phase 1 / 2 = s / t net backward ([W1n][W2Tp][W1Tp]); + [tail]       (tbb bytes)
//   tail (1 KiB): b1[32c .. 32c+31] (32 floats) | b2[0 .. hp) -- biases travel with the weights.
// a fragment = 64 lanes x EPF elements (1 KiB bf16 / 256 B fp32); every task is a multiple of 1 KiB.
struct Geom {
  int h, H, hp, Hp, HT, HC, K;
  int fe, nks, ns, esize;
  int64_t nfe_f, nfe_b;   // fragment elements per task
  int64_t tbf, tbb;       // bytes per task
  int64_t off_b, total;
};
constexpr int kTailBytes = 1024;
static inline Geom make_geom(int h, int H, int K, bool bf16) {
  Geom g;
  g.h = h; g.H = H; g.K = K;
  g.hp = (h + 31) / 32 * 32; g.Hp = (H + 31) / 32 * 32;
  g.HT = g.hp / 32; g.HC = g.Hp / 32;
  const int epf = bf16 ? 8 : 1, ks = bf16 ? 16 : 2;
  g.esize = bf16 ? 2 : 4;
  g.fe = 64 * epf; g.nks = g.hp / ks; g.ns = 32 / ks;
  g.nfe_f = (int64_t)(g.nks + g.HT * g.ns) * g.fe;
  g.nfe_b = (int64_t)(g.nks + 2 * g.HT * g.ns) * g.fe;
  g.tbf = g.nfe_f * g.esize + kTailBytes;
  g.tbb = g.nfe_b * g.esize + kTailBytes;
  g.off_b = (int64_t)K * 2 * g.HC * g.tbf;
  g.total = g.off_b + (int64_t)K * 3 * g.HC * g.tbb;
  return g;
}

struct StackArgs {
  const void* z_in;     // (N, D)
  void* z_out;          // (N, D)
  void* x1_hist;        // (K, N, h) or null            [fwd: written, bwd: read]
  const int64_t* mask;  // (Bn / mask_div, S) or null
  const char* packed;
  float* tok_val;       // (N) per-token masked log-det (+ base log-prob)
  int64_t N;
  int S, mask_div, D, K;
  int inverse, base, act;
  int G;                // tasks per staged block
  float mu, inv2var, cst;  // base dist: -(z-mu)^2 * inv2var - cst
  // backward only
  const void* dz_out;   // (N, D) or null
  const float* dlogp;   // (Bn) or null: d loss / d (per-sentence sum of tok_val)
  void* dz_in;          // (N, D)
  void* Gt;             // (K, 2, Hp, Npad) d pre-activation, token-contiguous
  void* Ht;             // (K, 2, Hp, Npad) hidden activations
  void* dSt;            // (K, 2, hp, Npad) ds / dt
  void* x0t;            // (hp, Npad) x0 * mask
  int64_t Npad;
  Geom g;
};

// 32 consecutive biases from the staged tail -> accumulator layout (broadcast ds_read_b128)
__device__ __forceinline__ f32x16 bias_tile(const float* B, int hh) {
  f32x16 a;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(B + 8 * q + 4 * hh);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[4 * q + i] = v[i];
  }
  return a;
}

// accumulator tiles <-> per-wave private LDS area (64 B per lane per tile, lane-linear)
template <int HT>
__device__ __forceinline__ void stash_put(float* st, const f32x16 (&x)[HT]) {
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {x[t][4 * q], x[t][4 * q + 1], x[t][4 * q + 2], x[t][4 * q + 3]};
      *reinterpret_cast<f32x4*>(st + t * 1024 + q * 256) = v;
    }
}
template <int HT>
__device__ __forceinline__ void stash_get(const float* st, f32x16 (&x)[HT]) {
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(st + t * 1024 + q * 256);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[t][4 * q + e] = v[e];
    }
}

struct DwArgs {
  const void *Gt, *Ht, *dSt, *x0t;
  float* part;    // [items][HT][64][16]
  float* partb1;  // [items][32]        (type 0 items only are meaningful)
  float* partb2;  // [K*2][nslab][hp]
  float *dW1, *db1, *dW2, *db2;
  const int64_t* sink;   // optional device table [K*2][4] of fp32 gradient addresses {W1, b1, W2, b2}: accumulate there
  int64_t Npad;
  int K, h, H, hp, Hp, HT, HC, nslab, slab_len;
  int x0_per_step;   // 0: one x0t section shared by all steps (a coupling-only stack never changes x0); 1: one per step
  int64_t nitems;
};

// fixed-order fold of the dumped partials (k_dw_reduce, nice_stack.hip)
int nice_dw_reduce_launch(const DwArgs& a, hipStream_t st);
// backward of small bf16 stacks with the nets of all steps side by side (k_stack_bwd_par, nice_small.hip)
bool nice_small_bwd_takes(const StackArgs& a, bool bf16, bool vec);
int nice_small_bwd_launch(const StackArgs& a, hipStream_t st);

}  // namespace mivae
