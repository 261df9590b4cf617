// Pairwise squared distances between the rows of two matrices, and their gradient: the O(B^2 F) core of the InfoVAE
// MMD term (vae_lm/nn/kl_loss.py:100-149: `(x1.unsqueeze(1) - x2).pow(2)` reduced over the features).  The reference
// materialises the (B, B, F) difference tensor (1 GiB at B = 256, F = 64 x 64); here a block owns a 16 x 16 tile of the
// (B1, B2) result and streams both row panels through LDS in 64-feature chunks, summing (a - b)^2 in the reference's
// own arithmetic (no Gram-identity cancellation) in a fixed order.  HBM traffic: the two inputs once per tile row /
// column (L2 resident: 2 B F e bytes), B1 B2 floats out.  VALU bound: 3 B1 B2 F lane-ops.
#include "common.hpp"

namespace {
using namespace mivae;

constexpr int kTile = 16;     // output tile edge
constexpr int kChunk = 64;    // features per LDS chunk

template <typename T>
__device__ __forceinline__ float ld1(const T* p) {
  float t[1];
  RowIO<T, 1>::load(p, t);
  return t[0];
}

template <typename T>
__global__ __launch_bounds__(256) void k_pair_sqdist(const T* __restrict__ a, const T* __restrict__ b,
                                                     float* __restrict__ out, int B1, int B2, int64_t F) {
  __shared__ float sa[kTile][kChunk + 1], sb[kTile][kChunk + 1];
  const int ti = threadIdx.x / kTile, tj = threadIdx.x % kTile;
  const int i0 = blockIdx.y * kTile, j0 = blockIdx.x * kTile;
  float acc = 0.f;
  for (int64_t f0 = 0; f0 < F; f0 += kChunk) {
    for (int e = threadIdx.x; e < kTile * kChunk; e += 256) {      // consecutive lanes read consecutive features
      const int r = e / kChunk, c = e % kChunk;
      const bool fin = f0 + c < F;
      sa[r][c] = (fin && i0 + r < B1) ? ld1(a + (int64_t)(i0 + r) * F + f0 + c) : 0.f;
      sb[r][c] = (fin && j0 + r < B2) ? ld1(b + (int64_t)(j0 + r) * F + f0 + c) : 0.f;
    }
    __syncthreads();
    float part = 0.f;                  // two-level sum (per chunk, then across chunks): error grows with F / 64, not F
#pragma unroll 8
    for (int c = 0; c < kChunk; ++c) {
      const float d = sa[ti][c] - sb[tj][c];
      part = fmaf(d, d, part);
    }
    acc += part;
    __syncthreads();
  }
  if (i0 + ti < B1 && j0 + tj < B2) out[(int64_t)(i0 + ti) * B2 + j0 + tj] = acc;
}

// da[r][f] = scale * sum_c G[r * gs_r + c * gs_c] * (a[r][f] - b[c][f])   for r < R, f < F, c < C
// (scale = 2, G row major: d/da of sum_ij G_ij |a_i - b_j|^2;  scale = -2 with the strides swapped: d/db)
template <typename T>
__global__ __launch_bounds__(256) void k_pair_sqdist_bwd(const T* __restrict__ a, const T* __restrict__ b,
                                                         const float* __restrict__ G, float* __restrict__ da, int R, int C,
                                                         int64_t F, int64_t gs_r, int64_t gs_c, float scale, int accumulate) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (f >= F || r >= R) return;
  const float av = ld1(a + (int64_t)r * F + f);
  float acc = 0.f;
  for (int c = 0; c < C; ++c) {
    const float g = G[r * gs_r + c * gs_c];                         // wave-uniform address: one broadcast load
    acc = fmaf(g, av - ld1(b + (int64_t)c * F + f), acc);
  }
  float* dst = da + (int64_t)r * F + f;
  *dst = accumulate ? *dst + scale * acc : scale * acc;
}

}  // namespace

// out (B1, B2) fp32, out[i][j] = sum_f (a[i][f] - b[j][f])^2 for a (B1, F), b (B2, F) in `dtype`: the pairwise term of
// InfoVAEKLLoss._compute_rbf / _compute_inv_mult_quad (nn/kl_loss.py:124-149) before the kernel function.
extern "C" int mivae_pair_sqdist(const void* a, const void* b, float* out, int64_t B1, int64_t B2, int64_t F, int dtype,
                                 void* stream) {
  MIVAE_REQUIRE(B1 >= 0 && B2 >= 0 && F >= 1 && B1 < (1 << 20) && B2 < (1 << 20), "pair_sqdist: bad sizes");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "pair_sqdist: dtype must be f32 or bf16");
  if (B1 == 0 || B2 == 0) return MIVAE_OK;
  MIVAE_REQUIRE(a && b && out, "pair_sqdist: null pointer");
  dim3 grid((unsigned)((B2 + kTile - 1) / kTile), (unsigned)((B1 + kTile - 1) / kTile));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MIVAE_F32) k_pair_sqdist<float><<<grid, 256, 0, s>>>((const float*)a, (const float*)b, out, (int)B1, (int)B2, F);
  else k_pair_sqdist<unsigned short><<<grid, 256, 0, s>>>((const unsigned short*)a, (const unsigned short*)b, out, (int)B1, (int)B2, F);
  MIVAE_CHECK_LAUNCH("pair_sqdist");
  return MIVAE_OK;
}

// Gradients of sum_ij G[i][j] out[i][j]: da (B1, F) and db (B2, F) fp32 (each may be NULL); accumulate != 0 adds into
// them (the z-z kernel matrix has both arguments equal: call twice into one buffer).
extern "C" int mivae_pair_sqdist_bwd(const void* a, const void* b, const float* G, float* da, float* db, int64_t B1,
                                     int64_t B2, int64_t F, int dtype, int accumulate, void* stream) {
  MIVAE_REQUIRE(B1 >= 0 && B2 >= 0 && F >= 1 && B1 < (1 << 20) && B2 < (1 << 20), "pair_sqdist_bwd: bad sizes");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "pair_sqdist_bwd: dtype must be f32 or bf16");
  if (B1 == 0 || B2 == 0) return MIVAE_OK;
  MIVAE_REQUIRE(a && b && G, "pair_sqdist_bwd: null pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned fb = (unsigned)((F + 255) / 256);
#define PAIR_BWD(TT)                                                                                                   \
  do {                                                                                                                 \
    if (da) k_pair_sqdist_bwd<TT><<<dim3(fb, (unsigned)B1), 256, 0, s>>>((const TT*)a, (const TT*)b, G, da, (int)B1,    \
                                                                         (int)B2, F, B2, 1, 2.0f, accumulate);         \
    if (db) k_pair_sqdist_bwd<TT><<<dim3(fb, (unsigned)B2), 256, 0, s>>>((const TT*)b, (const TT*)a, G, db, (int)B2,    \
                                                                         (int)B1, F, 1, B2, 2.0f, accumulate);         \
  } while (0)
  if (dtype == MIVAE_F32) PAIR_BWD(float); else PAIR_BWD(unsigned short);
#undef PAIR_BWD
  MIVAE_CHECK_LAUNCH("pair_sqdist_bwd");
  return MIVAE_OK;
}
