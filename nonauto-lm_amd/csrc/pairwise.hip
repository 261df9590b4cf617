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

// dens[i][j] = sum over (s, d) of mask[i][s] * -0.5 * ((z[i] - mu[j / n])^2 / sigma[j / n]^2 + 2 log sigma[j / n] + log 2 pi):
// the log-density of latent code i under the posterior of sentence j / n, every code against every sentence -- the
// (B n, B n, S, D) tensor of Posterior.calc_mutual_info (posterior.py:174-222) reduced on the fly.  Same tiling as
// k_pair_sqdist; 1 / sigma^2 and the per-feature constant are formed once per staged element.
template <typename T>
__global__ __launch_bounds__(256) void k_pair_gauss_logdens(const T* __restrict__ z, const T* __restrict__ mu,
                                                            const T* __restrict__ sigma, const int64_t* __restrict__ mask,
                                                            float* __restrict__ out, int Bn, int n, int S, int D) {
  __shared__ float sz[kTile][kChunk + 1], sm[kTile][kChunk + 1], su[kTile][kChunk + 1], sw[kTile][kChunk + 1],
      sc[kTile][kChunk + 1];
  const int ti = threadIdx.x / kTile, tj = threadIdx.x % kTile;
  const int i0 = blockIdx.y * kTile, j0 = blockIdx.x * kTile;
  const int64_t F = (int64_t)S * D;
  float acc = 0.f;
  for (int64_t f0 = 0; f0 < F; f0 += kChunk) {
    for (int e = threadIdx.x; e < kTile * kChunk; e += 256) {
      const int r = e / kChunk, c = e % kChunk;
      const int64_t f = f0 + c;
      const bool fin = f < F;
      const bool iok = fin && i0 + r < Bn, jok = fin && j0 + r < Bn;
      sz[r][c] = iok ? ld1(z + (int64_t)(i0 + r) * F + f) : 0.f;
      sm[r][c] = iok ? (mask ? (float)mask[(int64_t)(i0 + r) * S + f / D] : 1.f) : 0.f;
      float u = 0.f, w = 0.f, k = 0.f;
      if (jok) {
        const int64_t row = (int64_t)((j0 + r) / n) * F + f;
        const float sg = ld1(sigma + row);
        u = ld1(mu + row);
        w = 1.0f / (sg * sg);
        k = 2.0f * logf(sg) + kLog2Pi;
      }
      su[r][c] = u; sw[r][c] = w; sc[r][c] = k;
    }
    __syncthreads();
    float part = 0.f;
#pragma unroll 8
    for (int c = 0; c < kChunk; ++c) {
      const float d = sz[ti][c] - su[tj][c];
      part += sm[ti][c] * (-0.5f * (d * d * sw[tj][c] + sc[tj][c]));
    }
    acc += part;
    __syncthreads();
  }
  if (i0 + ti < Bn && j0 + tj < Bn) out[(int64_t)(i0 + ti) * Bn + j0 + tj] = acc;
}

}  // namespace

// dens (Bn, Bn) fp32 for z (Bn, S, D), mu / sigma (Bn / n, S, D) in `dtype`, mask (Bn, S) int64 or NULL: the pairwise
// Gaussian log-densities of Posterior.calc_mutual_info (nonauto posterior.py:174-222, auto posterior.py:158-200 with
// S = 1 and no mask) without the (Bn, Bn, S, D) intermediate; the caller takes logsumexp over j.
extern "C" int mivae_pair_gauss_logdens(const void* z, const void* mu, const void* sigma, const int64_t* mask, float* out,
                                        int64_t Bn, int64_t n, int64_t S, int64_t D, int dtype, void* stream) {
  MIVAE_REQUIRE(Bn >= 0 && Bn < (1 << 20) && n >= 1 && S >= 1 && D >= 1 && S * D < (1ll << 31) && Bn % n == 0,
                "pair_gauss_logdens: bad sizes");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "pair_gauss_logdens: dtype must be f32 or bf16");
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(z && mu && sigma && out, "pair_gauss_logdens: null pointer");
  dim3 grid((unsigned)((Bn + kTile - 1) / kTile), (unsigned)((Bn + kTile - 1) / kTile));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIVAE_F32)
    k_pair_gauss_logdens<float><<<grid, 256, 0, st>>>((const float*)z, (const float*)mu, (const float*)sigma, mask, out,
                                                      (int)Bn, (int)n, (int)S, (int)D);
  else
    k_pair_gauss_logdens<unsigned short><<<grid, 256, 0, st>>>((const unsigned short*)z, (const unsigned short*)mu,
                                                               (const unsigned short*)sigma, mask, out, (int)Bn, (int)n,
                                                               (int)S, (int)D);
  MIVAE_CHECK_LAUNCH("pair_gauss_logdens");
  return MIVAE_OK;
}

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
