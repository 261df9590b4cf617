// HBM-bound kernels of the latent path: masks, posterior reparameterisation + log q, base log p,
// NICE affine tail, ActNorm, per-sentence (segment) sums, KL / ELBO tail.
//
// Common shape: a latent tensor is `rows x D` with rows = (B*n)*S tokens, row-major.  L lanes
// (power of two <= 64) cooperate on one token row with 16-byte accesses; per-token sums are reduced
// with wavefront shuffles (__shfl_xor over the L-lane group) and written as one fp32 per token;
// per-sentence sums are a second, fixed-order pass (segment_sum) -- no float atomics anywhere, so
// results are run-to-run deterministic.  All accumulation is fp32 regardless of the I/O dtype.
#include "common.hpp"
#include <stdarg.h>

using namespace mivae;

// ------------------------------------------------------------------------------------------------
// error plumbing (thread-local last error, SURVEY 8b "Error convention")
static thread_local char g_err[512] = "";
void mivae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* mivae_last_error(void) { return g_err; }
extern "C" int mivae_version(void) { return 100; }

namespace {

constexpr int kBlock = 256;

// Single-instruction transcendentals (v_log_f32 / v_exp_f32 / v_rcp_f32: ~1 ulp): the elementwise kernels of the latent
// path draw noise (Philox + Box-Muller), take a softplus, a log and a reciprocal PER ELEMENT -- with the IEEE-exact libm
// forms (logf, expf, log1pf, sincospif, division: 20-60 instructions each) they were VALU-bound at 2-3x their HBM time.
// Relative error <= ~3e-7 on the values used here, two orders inside the 1e-5 parity bound.
__device__ __forceinline__ float f_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }
__device__ __forceinline__ float f_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float f_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// softplus with beta = 1, threshold = 20 (torch.nn.Softplus defaults, posterior.py:23-27): log1p(e^x) as a 4-term series
// below e^x = 2^-6 (truncation 1e-8 relative) and log(1 + e^x) above (1 + e^x is then exact to 6e-8 of a value >= 0.0155)
__device__ __forceinline__ float softplus20(float x) {
  const float e = f_exp(fminf(x, 20.f));
  const float series = e * (1.0f - e * (0.5f - e * (0.33333334f - 0.25f * e)));
  const float lg = f_log(1.0f + e);
  return x > 20.f ? x : (e < 0.015625f ? series : lg);
}
__device__ __forceinline__ float sigmoid_f(float x) { return f_rcp(1.0f + f_exp(-x)); }

struct RowCfg {
  int vec;
  int L;
  unsigned grid;
};

// vec_max = 4 (f32) or 8 (bf16); `width` is the row width the lanes tile (D, or h for half rows).
static RowCfg row_cfg(int64_t rows, int64_t width, int dtype, bool aligned) {
  RowCfg c;
  int vmax = dtype == MIVAE_BF16 ? 8 : 4;
  c.vec = (aligned && width % vmax == 0) ? vmax : 1;
  c.L = lanes_per_row(width, c.vec);
  int rows_per_block = kBlock / c.L;
  c.grid = (unsigned)((rows + rows_per_block - 1) / rows_per_block);
  if (c.grid == 0) c.grid = 1;
  return c;
}

#define DISPATCH(dtype, vec, FN, ...)                                  \
  do {                                                                   \
    if ((dtype) == MIVAE_F32) {                                          \
      if ((vec) == 4) FN<float, 4>(__VA_ARGS__);                         \
      else FN<float, 1>(__VA_ARGS__);                                    \
    } else {                                                             \
      if ((vec) == 8) FN<unsigned short, 8>(__VA_ARGS__);                \
      else FN<unsigned short, 1>(__VA_ARGS__);                           \
    }                                                                    \
  } while (0)

// ------------------------------------------------------------------------------------------------
// a1: get_tokens_mask  (vae_lm/nn/utils.py:12-14)  -- integer, bit exact
__global__ void k_tokens_mask(const int64_t* __restrict__ tok, int64_t* __restrict__ mask, float* __restrict__ maskf,
                              int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const bool keep = tok[i] != 0;
    mask[i] = keep ? 1 : 0;
    if (maskf) maskf[i] = keep ? 1.f : 0.f;        // the same mask as fp32 token weights (the loss takes it as such)
  }
}

// a2: last-token gather (encoders/encoder.py:35-37).  One block per batch row.
template <typename T>
__global__ void k_gather_last(const T* __restrict__ out, const int64_t* __restrict__ mask,
                              T* __restrict__ ctx, int64_t* __restrict__ last_idx, int S, int E) {
  __shared__ long long s_cnt[kBlock / 64];
  __shared__ long long s_total;
  int b = blockIdx.x;
  long long c = 0;
  for (int s = threadIdx.x; s < S; s += blockDim.x) c += mask[(int64_t)b * S + s];
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_cnt[w];
    s_total = t;
  }
  __syncthreads();
  long long idx = s_total - 1;
  if (threadIdx.x == 0 && last_idx) last_idx[b] = idx;
  if (idx < 0) idx += S;  // torch advanced indexing wraps -1 to the last position
  if (idx >= S) idx = S - 1;
  const T* src = out + ((int64_t)b * S + idx) * E;
  for (int e = threadIdx.x; e < E; e += blockDim.x) ctx[(int64_t)b * E + e] = src[e];
}

// ------------------------------------------------------------------------------------------------
// Counter-based standard-normal noise for the reparameterisation (posterior.py:115-118 draws eps with
// base_dist.sample): Philox4x32-10 (Salmon et al., SC'11) keyed by the seed, counter = (element quad, draw offset);
// the four 32-bit outputs of one counter give four normals through two Box-Muller pairs.  Element e of draw number
// `offset` is a pure function of (seed, offset, e): the backward REGENERATES eps instead of reading it back.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
  c[0] = n0; c[1] = (unsigned)p1; c[2] = n2; c[3] = (unsigned)p0;
}
__device__ __forceinline__ void philox_normal4v(unsigned long long seed, unsigned long long off, int64_t quad, float (&out)[4]) {
  unsigned c[4] = {(unsigned)quad, (unsigned)((unsigned long long)quad >> 32), (unsigned)off, (unsigned)(off >> 32)};
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)c[2 * h] + 0.5f) * 2.3283064365386963e-10f;        // (0, 1]
    const float u2 = ((float)c[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
    // Box-Muller on the hardware transcendentals: r = sqrt(-2 ln u1) from v_log_f32 (log2), v_sin / v_cos take the angle
    // in revolutions
    const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    out[2 * h] = r * __builtin_amdgcn_cosf(u2); out[2 * h + 1] = r * __builtin_amdgcn_sinf(u2);
  }
}
__device__ __forceinline__ void philox_normal4(const int64_t* __restrict__ rng, int64_t quad, float (&out)[4]) {
  philox_normal4v((unsigned long long)rng[0], (unsigned long long)rng[1], quad, out);
}
// eps of the VEC elements starting at flat element index e0 (VEC = 1: any e0; VEC = 4, 8: e0 a multiple of 4)
template <int VEC>
__device__ __forceinline__ void philox_eps(const int64_t* __restrict__ rng, int64_t e0, float (&ve)[VEC]) {
  if constexpr (VEC == 1) {
    float q[4];
    philox_normal4(rng, e0 >> 2, q);
    ve[0] = q[e0 & 3];
  } else {
#pragma unroll
    for (int i = 0; i < VEC; i += 4) {
      float q[4];
      philox_normal4(rng, (e0 + i) >> 2, q);
#pragma unroll
      for (int j = 0; j < 4; ++j) ve[i + j] = q[j];
    }
  }
}

// take one draw number from the generator: used = state; state.offset += 1 (one thread; stream ordered)
__global__ void k_rng_take(int64_t* __restrict__ state, int64_t* __restrict__ used) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int64_t seed = state[0], off = state[1];
    if (used) { used[0] = seed; used[1] = off; }
    state[1] = off + 1;
  }
}
__global__ void k_philox_normal(const int64_t* __restrict__ rng, float* __restrict__ out, int64_t count) {
  const int64_t quad = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (quad * 4 >= count) return;
  float q[4];
  philox_normal4(rng, quad, q);
  for (int j = 0; j < 4; ++j)
    if (quad * 4 + j < count) out[quad * 4 + j] = q[j];
}

// ------------------------------------------------------------------------------------------------
// a4 + a5: z = (mu + eps * sigma) * mask ; tok_logq = mask * sum_d -1/2 ((z-mu)^2 / sigma^2 + 2 log sigma + log 2pi)
// rows over (b, j, s); mu/sigma are (B, S, D), eps/z are ((B n), S, D).  posterior.py:119-133,157-172.
template <typename T, int VEC>
__global__ void k_reparam_fwd(const T* __restrict__ mu, const T* __restrict__ sigma,
                              const T* __restrict__ eps, const int64_t* __restrict__ mask,
                              T* __restrict__ z, float* __restrict__ tok_logq, int64_t rows, int n,
                              int S, int D, int L, const int64_t* __restrict__ rng) {
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = tid / L;
  int lig = (int)(tid % L);
  float acc = 0.f;
  if (row < rows) {
    int64_t bn = row / S;
    int s = (int)(row - bn * S);
    int64_t b = bn / n;
    int64_t prow = b * S + s;
    float m = mask ? (float)mask[prow] : 1.f;
    for (int c = lig * VEC; c < D; c += L * VEC) {
      float vm[VEC], vs[VEC], ve[VEC], vz[VEC];
      RowIO<T, VEC>::load(mu + prow * D + c, vm);
      RowIO<T, VEC>::load(sigma + prow * D + c, vs);
      if (eps) RowIO<T, VEC>::load(eps + row * D + c, ve);
      else if (rng) philox_eps<VEC>(rng, row * D + c, ve);          // in-kernel draw: eps never touches HBM
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float e = (eps || rng) ? ve[i] : 0.f;
        float zz = (vm[i] + e * vs[i]) * m;
        float d = zz - vm[i];
        float sg = vs[i];
        acc += -0.5f * (d * d * f_rcp(sg * sg) + 2.0f * f_log(sg) + kLog2Pi);
        vz[i] = zz;
      }
      RowIO<T, VEC>::store(z + row * D + c, vz);
    }
    acc *= m;
  }
  acc = group_sum(acc, L);
  if (row < rows && lig == 0) tok_logq[row] = acc;
}

// backward: rows over (b, s); sums the n samples in fixed order.
//   dmu = sum_j dz_j * m ; dsigma = sum_j (dz_j * eps_j - dlogq_j / sigma) * m
template <typename T, int VEC>
__global__ void k_reparam_bwd(const T* __restrict__ dz, const float* __restrict__ dlogq,
                              const T* __restrict__ sigma, const T* __restrict__ eps,
                              const int64_t* __restrict__ mask, T* __restrict__ dmu,
                              T* __restrict__ dsigma, int64_t prows, int n, int S, int D, int L,
                              const int64_t* __restrict__ rng) {
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t prow = tid / L;
  int lig = (int)(tid % L);
  if (prow >= prows) return;
  int64_t b = prow / S;
  int s = (int)(prow - b * S);
  float m = mask ? (float)mask[prow] : 1.f;
  for (int c = lig * VEC; c < D; c += L * VEC) {
    float vs[VEC], am[VEC], as[VEC];
    RowIO<T, VEC>::load(sigma + prow * D + c, vs);
#pragma unroll
    for (int i = 0; i < VEC; ++i) { am[i] = 0.f; as[i] = 0.f; }
    for (int j = 0; j < n; ++j) {
      int64_t bn = b * n + j;
      int64_t row = bn * S + s;
      float g[VEC], e[VEC];
      if (dz) RowIO<T, VEC>::load(dz + row * D + c, g);
      if (eps) RowIO<T, VEC>::load(eps + row * D + c, e);
      else if (rng) philox_eps<VEC>(rng, row * D + c, e);           // the same draw as the forward
      float dq = dlogq ? dlogq[bn] : 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float gi = dz ? g[i] : 0.f;
        float ei = (eps || rng) ? e[i] : 0.f;
        am[i] += gi;
        as[i] += gi * ei - dq * f_rcp(vs[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) { am[i] *= m; as[i] *= m; }
    RowIO<T, VEC>::store(dmu + prow * D + c, am);
    RowIO<T, VEC>::store(dsigma + prow * D + c, as);
  }
}

// ------------------------------------------------------------------------------------------------
// a3 tail + a4 + a5 in ONE launch each way (launch-count pass over the latent path): the sigma head
// Hardtanh(Softplus(pre), lo, hi) (posterior.py:23-27) is evaluated inside the sample / log q kernel (PRE = true: the
// second operand is the head's pre-activation) and sigma is written out once for the callers that want it.  With
// in-kernel noise the launch READS the generator state {seed, offset} and records it in rng_used; advancing the state
// is left to a later single-thread step on the same stream -- mivae_rng_advance, or the ELBO tail kernel of the same
// forward (mivae_latent_tail_fwd's rng_state argument), so that the training step pays no launch for it.  (A version in
// which the last block to finish advanced the state through an atomic ticket was measured: thousands of blocks
// serialise on the one ticket word, 53 us instead of 23 us at the C3 shape.)  sigma is rounded to the latent dtype
// before use, exactly what the two-launch form (sigma head, then k_reparam_fwd) computes with.
template <int VEC>
__device__ __forceinline__ void philox_epsv(unsigned long long seed, unsigned long long off, int64_t e0, float (&ve)[VEC]) {
  if constexpr (VEC == 1) {
    float q[4];
    philox_normal4v(seed, off, e0 >> 2, q);
    ve[0] = q[e0 & 3];
  } else {
#pragma unroll
    for (int i = 0; i < VEC; i += 4) {
      float q[4];
      philox_normal4v(seed, off, (e0 + i) >> 2, q);
#pragma unroll
      for (int j = 0; j < 4; ++j) ve[i + j] = q[j];
    }
  }
}
template <typename T>
__device__ __forceinline__ float round_to(float x) {
  if constexpr (sizeof(T) == 2) return bf16_to_f32(f32_to_bf16(x));
  else return x;
}

template <typename T, int VEC, bool PRE>
__global__ void k_post_fwd(const T* __restrict__ mu, const T* __restrict__ sg_in, const T* __restrict__ eps,
                           const int64_t* __restrict__ mask, T* __restrict__ z, T* __restrict__ sigma_out,
                           float* __restrict__ tok_logq, int64_t rows, int n, int S, int D, int L, float lo, float hi,
                           const int64_t* __restrict__ rng_state, int64_t* __restrict__ rng_used) {
  unsigned long long seed = 0, off = 0;
  if (rng_state) { seed = (unsigned long long)rng_state[0]; off = (unsigned long long)rng_state[1]; }
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = tid / L;
  int lig = (int)(tid % L);
  float acc = 0.f;
  if (row < rows) {
    int64_t bn = row / S;
    int s = (int)(row - bn * S);
    int64_t b = bn / n;
    int64_t prow = b * S + s;
    const bool first = bn - b * n == 0;          // the sample that writes sigma (one copy per posterior row)
    float m = mask ? (float)mask[prow] : 1.f;
    for (int c = lig * VEC; c < D; c += L * VEC) {
      float vm[VEC], vs[VEC], ve[VEC], vz[VEC];
      RowIO<T, VEC>::load(mu + prow * D + c, vm);
      RowIO<T, VEC>::load(sg_in + prow * D + c, vs);
      if (PRE) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) vs[i] = round_to<T>(fminf(fmaxf(softplus20(vs[i]), lo), hi));
        if (sigma_out && first) RowIO<T, VEC>::store(sigma_out + prow * D + c, vs);
      }
      if (eps) RowIO<T, VEC>::load(eps + row * D + c, ve);
      else if (rng_state) philox_epsv<VEC>(seed, off, row * D + c, ve);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float e = (eps || rng_state) ? ve[i] : 0.f;
        float zz = (vm[i] + e * vs[i]) * m;
        float d = zz - vm[i];
        float sg = vs[i];
        acc += -0.5f * (d * d * f_rcp(sg * sg) + 2.0f * f_log(sg) + kLog2Pi);
        vz[i] = zz;
      }
      RowIO<T, VEC>::store(z + row * D + c, vz);
    }
    acc *= m;
  }
  acc = group_sum(acc, L);
  if (row < rows && lig == 0) tok_logq[row] = acc;
  if (rng_state && blockIdx.x == 0 && threadIdx.x == 0) {   // the draw this launch used, for the backward
    rng_used[0] = (int64_t)seed; rng_used[1] = (int64_t)off;
  }
}

// backward: rows over (b, s); dq_tok = 0: dlogq holds one cotangent per sentence (B n); 1: one per token ((B n) S).
//   dmu = sum_j dz_j * m ; dsigma = sum_j (dz_j * eps_j - dlogq_j / sigma) * m ; PRE: d pre = dsigma * sigmoid(pre) inside
//   the open interval (lo, hi), 0 elsewhere (k_sigma_head_bwd), dsigma rounded to the latent dtype first like the
//   two-launch form
template <typename T, int VEC, bool PRE>
__global__ void k_post_bwd(const T* __restrict__ dz, const float* __restrict__ dlogq, int dq_tok,
                           const T* __restrict__ sg_in, const T* __restrict__ eps, const int64_t* __restrict__ mask,
                           T* __restrict__ dmu, T* __restrict__ dsg, int64_t prows, int n, int S, int D, int L,
                           float lo, float hi, const int64_t* __restrict__ rng_used) {
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t prow = tid / L;
  int lig = (int)(tid % L);
  if (prow >= prows) return;
  unsigned long long seed = 0, off = 0;
  if (rng_used) { seed = (unsigned long long)rng_used[0]; off = (unsigned long long)rng_used[1]; }
  int64_t b = prow / S;
  int s = (int)(prow - b * S);
  float m = mask ? (float)mask[prow] : 1.f;
  for (int c = lig * VEC; c < D; c += L * VEC) {
    float vp[VEC], vs[VEC], am[VEC], as[VEC];
    RowIO<T, VEC>::load(sg_in + prow * D + c, vp);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      vs[i] = PRE ? round_to<T>(fminf(fmaxf(softplus20(vp[i]), lo), hi)) : vp[i];
      am[i] = 0.f; as[i] = 0.f;
    }
    for (int j = 0; j < n; ++j) {
      int64_t bn = b * n + j;
      int64_t row = bn * S + s;
      float g[VEC], e[VEC];
      if (dz) RowIO<T, VEC>::load(dz + row * D + c, g);
      if (eps) RowIO<T, VEC>::load(eps + row * D + c, e);
      else if (rng_used) philox_epsv<VEC>(seed, off, row * D + c, e);
      float dq = dlogq ? dlogq[dq_tok ? row : bn] : 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float gi = dz ? g[i] : 0.f;
        float ei = (eps || rng_used) ? e[i] : 0.f;
        am[i] += gi;
        as[i] += gi * ei - dq * f_rcp(vs[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      am[i] *= m; as[i] *= m;
      if (PRE) {
        const float sp = softplus20(vp[i]);
        const float d = vp[i] > 20.f ? 1.f : sigmoid_f(vp[i]);
        as[i] = (sp > lo && sp < hi) ? round_to<T>(as[i]) * d : 0.f;
      }
    }
    RowIO<T, VEC>::store(dmu + prow * D + c, am);
    RowIO<T, VEC>::store(dsg + prow * D + c, as);
  }
}

// ------------------------------------------------------------------------------------------------
// a15: base Gaussian log prob per token: tok = m * sum_d ( -(z-mu)^2/(2 std^2) - log std - 1/2 log 2pi )
// (torch.distributions.Normal.log_prob; nonauto priors/prior.py:148-158).  If tok_add != null the
// per-token value is added to it (used to fold in per-token log-dets).
template <typename T, int VEC>
__global__ void k_gauss_logprob_fwd(const T* __restrict__ z, const int64_t* __restrict__ mask,
                                    float* __restrict__ tok, int64_t rows, int mask_div, int S,
                                    int D, int L, float mu, float inv2var, float cst) {
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = tid / L;
  int lig = (int)(tid % L);
  float acc = 0.f;
  float m = 1.f;
  if (row < rows) {
    if (mask) {
      int64_t bn = row / S;
      m = (float)mask[(bn / mask_div) * S + (row - bn * S)];
    }
    for (int c = lig * VEC; c < D; c += L * VEC) {
      float v[VEC];
      RowIO<T, VEC>::load(z + row * D + c, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float d = v[i] - mu;
        acc += -(d * d) * inv2var - cst;
      }
    }
    acc *= m;
  }
  acc = group_sum(acc, L);
  if (row < rows && lig == 0) tok[row] = acc;
}

// dz = dlogp[bn] * m * (-(z - mu) / std^2)
template <typename T, int VEC>
__global__ void k_gauss_logprob_bwd(const T* __restrict__ z, const int64_t* __restrict__ mask,
                                    const float* __restrict__ dlogp, T* __restrict__ dz,
                                    int64_t rows, int mask_div, int S, int D, int L, float mu,
                                    float invvar) {
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = tid / L;
  int lig = (int)(tid % L);
  if (row >= rows) return;
  int64_t bn = row / S;
  float m = mask ? (float)mask[(bn / mask_div) * S + (row - bn * S)] : 1.f;
  float g = dlogp[bn] * m;
  for (int c = lig * VEC; c < D; c += L * VEC) {
    float v[VEC], o[VEC];
    RowIO<T, VEC>::load(z + row * D + c, v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = -g * (v[i] - mu) * invvar;
    RowIO<T, VEC>::store(dz + row * D + c, o);
  }
}

// ------------------------------------------------------------------------------------------------
// per-sentence sums: out[b] = scale * sum_s tok[b, s] (+ add[b]).  One wave per sentence, lanes
// stride over s, shuffle tree at the end: fixed order for a given S.
__global__ void k_segment_sum(const float* __restrict__ tok, const float* __restrict__ add,
                              float* __restrict__ out, int64_t nseg, int S) {
  int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  float acc = 0.f;
  if (wave < nseg)
    for (int s = lane; s < S; s += 64) acc += tok[wave * S + s];
  acc = group_sum(acc, 64);
  if (wave < nseg && lane == 0) out[wave] = acc + (add ? add[wave] : 0.f);
}

// integer token counts per (repeated) sentence: count[bn] = sum_s mask[bn / mask_div, s]
__global__ void k_mask_count(const int64_t* __restrict__ mask, int64_t* __restrict__ count,
                             int64_t nseg, int mask_div, int S) {
  int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  long long acc = 0;
  if (wave < nseg)
    for (int s = lane; s < S; s += 64) acc += mask[(wave / mask_div) * S + s];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (wave < nseg && lane == 0) count[wave] = acc;
}

// ------------------------------------------------------------------------------------------------
// a8/a9 affine tail of NICECoupling given s,t (rows x h): couplings.py:151-162 / :172-183.
template <typename T, int VEC>
__global__ void k_nice_affine_fwd(const T* __restrict__ z, const T* __restrict__ s,
                                  const T* __restrict__ t, const int64_t* __restrict__ mask,
                                  T* __restrict__ out, float* __restrict__ tok_ld, int64_t rows,
                                  int mask_div, int S, int h, int L, int inverse) {
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = tid / L;
  int lig = (int)(tid % L);
  float acc = 0.f;
  if (row < rows) {
    float m = 1.f;
    if (mask) {
      int64_t bn = row / S;
      m = (float)mask[(bn / mask_div) * S + (row - bn * S)];
    }
    const int D = 2 * h;
    for (int c = lig * VEC; c < h; c += L * VEC) {
      float x0[VEC], x1[VEC], vs[VEC], vt[VEC], y1[VEC];
      RowIO<T, VEC>::load(z + row * D + c, x0);
      RowIO<T, VEC>::load(z + row * D + h + c, x1);
      RowIO<T, VEC>::load(s + row * h + c, vs);
      RowIO<T, VEC>::load(t + row * h + c, vt);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        if (!inverse) {
          y1[i] = (expf(vs[i]) * x1[i] + vt[i]) * m;
          acc += vs[i];
        } else {
          y1[i] = ((x1[i] - vt[i]) * expf(-vs[i])) * m;
          acc -= vs[i];
        }
        x0[i] *= m;
      }
      RowIO<T, VEC>::store(out + row * D + c, x0);
      RowIO<T, VEC>::store(out + row * D + h + c, y1);
    }
    acc *= m;
  }
  acc = group_sum(acc, L);
  if (row < rows && lig == 0) tok_ld[row] = acc;
}

template <typename T, int VEC>
__global__ void k_nice_affine_bwd(const T* __restrict__ dout, const float* __restrict__ dld,
                                  const T* __restrict__ z, const T* __restrict__ s,
                                  const T* __restrict__ t, const int64_t* __restrict__ mask,
                                  T* __restrict__ dz, T* __restrict__ ds, T* __restrict__ dt,
                                  int64_t rows, int mask_div, int S, int h, int L, int inverse) {
  int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = tid / L;
  int lig = (int)(tid % L);
  if (row >= rows) return;
  int64_t bn = row / S;
  float m = mask ? (float)mask[(bn / mask_div) * S + (row - bn * S)] : 1.f;
  float gl = dld ? dld[bn] * m : 0.f;
  const int D = 2 * h;
  for (int c = lig * VEC; c < h; c += L * VEC) {
    float g0[VEC], g1[VEC], x1[VEC], vs[VEC], vt[VEC], o1[VEC], os[VEC], ot[VEC];
    RowIO<T, VEC>::load(dout + row * D + c, g0);
    RowIO<T, VEC>::load(dout + row * D + h + c, g1);
    RowIO<T, VEC>::load(z + row * D + h + c, x1);
    RowIO<T, VEC>::load(s + row * h + c, vs);
    if (inverse) RowIO<T, VEC>::load(t + row * h + c, vt);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float g = g1[i] * m;
      if (!inverse) {
        float e = expf(vs[i]);
        o1[i] = g * e;
        os[i] = g * e * x1[i] + gl;
        ot[i] = g;
      } else {
        float e = expf(-vs[i]);
        o1[i] = g * e;
        ot[i] = -g * e;
        os[i] = -g * (x1[i] - vt[i]) * e - gl;
      }
      g0[i] *= m;
    }
    RowIO<T, VEC>::store(dz + row * D + c, g0);
    RowIO<T, VEC>::store(dz + row * D + h + c, o1);
    RowIO<T, VEC>::store(ds + row * h + c, os);
    RowIO<T, VEC>::store(dt + row * h + c, ot);
  }
}

// ------------------------------------------------------------------------------------------------
// a12/a13 ActNorm (actnorm.py:31-33 forward, :45-48 inverse).  Parameters are fp32.
// log_det[bn] = sign * (sum_d log_scale[d]) * count[bn]   (count = integer token count, or 1 when there is no mask /
// 2-D input, in which case the caller passes nseg = 1).  One wave per output.
__device__ __forceinline__ void actnorm_logdet_wave(const float* __restrict__ log_scale, const int64_t* __restrict__ mask,
                                                    float* __restrict__ out, int64_t wave, int64_t nseg, int mask_div,
                                                    int S, int D, float sign) {
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) acc += log_scale[d];
  acc = group_sum(acc, 64);
  long long cnt = 0;
  if (mask && wave < nseg)
    for (int s = lane; s < S; s += 64) cnt += mask[(wave / mask_div) * S + s];
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  if (wave < nseg && lane == 0) out[wave] = sign * acc * (mask ? (float)cnt : 1.f);
}
__global__ void k_actnorm_logdet(const float* __restrict__ log_scale, const int64_t* __restrict__ mask,
                                 float* __restrict__ out, int64_t nseg, int mask_div, int S, int D,
                                 float sign) {
  actnorm_logdet_wave(log_scale, mask, out, ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nseg, mask_div, S, D, sign);
}

// A thread owns one VEC-wide column chunk of kActRows rows (the block's row groups interleaved): exp(log_scale) and the
// bias are formed once per thread instead of once per element (the IEEE expf per element made the kernel VALU-bound at
// 4x its HBM time: 36.6 us for 32 MiB at the C3 shape), and the kActRows independent 16-byte loads are in flight
// together.  The first waves of the grid also write the log-determinants (was a launch of its own, 4 us).
constexpr int kActRows = 4;
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_actnorm_fwd(const T* __restrict__ z, const float* __restrict__ log_scale,
                                                     const float* __restrict__ bias, const int64_t* __restrict__ mask,
                                                     T* __restrict__ out, int64_t rows, int mask_div, int S, int D, int L,
                                                     int inverse, float* __restrict__ logdet, int64_t n_logdet,
                                                     const int64_t* __restrict__ ld_mask) {
  const int rif = blockDim.x / L, rg = threadIdx.x / L, lig = threadIdx.x % L;
  const int64_t row0 = (int64_t)blockIdx.x * (rif * kActRows) + rg;
  float m[kActRows];
#pragma unroll
  for (int k = 0; k < kActRows; ++k) {
    const int64_t row = row0 + (int64_t)k * rif;
    m[k] = 1.f;
    if (mask && row < rows) {
      const int64_t bn = row / S;
      m[k] = (float)mask[(bn / mask_div) * S + (row - bn * S)];
    }
  }
  if (row0 < rows)
    for (int c = lig * VEC; c < D; c += L * VEC) {
      float sc[VEC], bi[VEC], v[kActRows][VEC];
#pragma unroll
      for (int k = 0; k < kActRows; ++k) {
        const int64_t row = row0 + (int64_t)k * rif;
        if (row < rows) RowIO<T, VEC>::load(z + row * D + c, v[k]);
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float e = expf(log_scale[c + i]);
        sc[i] = inverse ? e + 1e-13f : e;
        bi[i] = bias[c + i];
      }
#pragma unroll
      for (int k = 0; k < kActRows; ++k) {
        const int64_t row = row0 + (int64_t)k * rif;
        if (row >= rows) continue;
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          if (!inverse) o[i] = (v[k][i] * sc[i] + bi[i]) * m[k];
          else o[i] = ((v[k][i] - bi[i]) * m[k]) / sc[i];
        }
        RowIO<T, VEC>::store(out + row * D + c, o);
      }
    }
  if (logdet) {
    const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w < n_logdet) actnorm_logdet_wave(log_scale, ld_mask, logdet, w, n_logdet, mask_div, S, D, inverse ? -1.f : 1.f);
  }
}

// Column reductions over token rows (two stage, fixed order).  Each block owns a contiguous chunk
// of rows; lanes own VEC columns; partial[block][NV][D] then k_col_finish sums blocks in order.
//   MODE 0: v0 = x                       (mean pass of the ActNorm init)
//   MODE 1: v0 = (x - aux[c])^2          (variance pass)
//   MODE 2: ActNorm backward: g = dout*m ; dz = g*exp(ls) ; v0 = g*x*exp(ls) ; v1 = g
template <typename T, int VEC, int MODE>
__global__ void __launch_bounds__(256) k_col_partial(const T* __restrict__ x, const T* __restrict__ dout,
                              const float* __restrict__ aux, const int64_t* __restrict__ mask,
                              T* __restrict__ dz, float* __restrict__ partial, int64_t rows,
                              int rows_per_block, int mask_div, int S, int D, int L) {
  constexpr int NV = MODE == 2 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [rows_in_flight][NV][L*VEC]
  const int rif = blockDim.x / L;                               // rows in flight
  const int rg = threadIdx.x / L;
  const int lig = threadIdx.x % L;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (int cbase = 0; cbase < D; cbase += L * VEC) {
    const int c = cbase + lig * VEC;
    float a0[VEC], a1[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
    if (c < D) {
      float ax[VEC];
      if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) ax[i] = MODE == 2 ? expf(aux[c + i]) : aux[c + i];
      }
      int64_t row = r0 + rg;
      if (MODE == 2) {
        // four rows per trip: eight independent 16-byte loads in flight, accumulated in the same row order as the
        // one-row loop below (which takes the remainder)
        for (; row + 3 * (int64_t)rif < r1; row += 4 * (int64_t)rif) {
          float v[4][VEC], g[4][VEC], m[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int64_t rk = row + (int64_t)k * rif;
            RowIO<T, VEC>::load(x + rk * D + c, v[k]);
            RowIO<T, VEC>::load(dout + rk * D + c, g[k]);
            m[k] = 1.f;
            if (mask) {
              const int64_t bn = rk / S;
              m[k] = (float)mask[(bn / mask_div) * S + (rk - bn * S)];
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float o[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              const float gi = g[k][i] * m[k];
              o[i] = gi * ax[i];
              a0[i] += gi * v[k][i] * ax[i];
              a1[i] += gi;
            }
            RowIO<T, VEC>::store(dz + (row + (int64_t)k * rif) * D + c, o);
          }
        }
      }
      for (; row < r1; row += rif) {
        float v[VEC];
        RowIO<T, VEC>::load(x + row * D + c, v);
        if (MODE == 0) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) a0[i] += v[i];
        } else if (MODE == 1) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) { float d = v[i] - ax[i]; a0[i] += d * d; }
        } else {
          float m = 1.f;
          if (mask) {
            int64_t bn = row / S;
            m = (float)mask[(bn / mask_div) * S + (row - bn * S)];
          }
          float g[VEC], o[VEC];
          RowIO<T, VEC>::load(dout + row * D + c, g);
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            float gi = g[i] * m;
            o[i] = gi * ax[i];
            a0[i] += gi * v[i] * ax[i];
            a1[i] += gi;
          }
          RowIO<T, VEC>::store(dz + row * D + c, o);
        }
      }
    }
    // combine the row groups of this block in fixed order
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      smem[(rg * NV + 0) * (L * VEC) + lig * VEC + i] = a0[i];
      if (NV == 2) smem[(rg * NV + 1) * (L * VEC) + lig * VEC + i] = a1[i];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NV * L * VEC; idx += blockDim.x) {
      int v = idx / (L * VEC);
      int cc = idx % (L * VEC);
      if (cbase + cc < D) {
        float sum = 0.f;
        for (int g = 0; g < rif; ++g) sum += smem[(g * NV + v) * (L * VEC) + cc];
        partial[((int64_t)blockIdx.x * NV + v) * D + cbase + cc] = sum;
      }
    }
  }
}

// out[v][c] = scale * sum_blocks partial[blk][v][c]; take_sqrt: std = sqrt(sum / (rows - 1)).
// A block of 4 waves owns 64 consecutive (v, c) outputs; wave g sums blocks g, g+4, ... (independent loads,
// 16 in flight) and the four partial sums are combined through LDS in wave order (fixed order).
__global__ void __launch_bounds__(256) k_col_finish(const float* __restrict__ partial, float* __restrict__ out,
                                                    int nblk, int nv, int D, float scale, int take_sqrt,
                                                    float* __restrict__ out1 = nullptr, int split = 0, int accumulate = 0) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;
  const int total = nv * D;
  float sum = 0.f;
  if (idx < total) {
    const float* p = partial + idx;                 // partial[(b * nv + v) * D + c] = partial[b * total + idx]
#pragma unroll 16
    for (int b = grp; b < nblk; b += 4) sum += p[(int64_t)b * total];
  }
  red[grp][lane] = sum;
  __syncthreads();
  if (grp != 0 || idx >= total) return;
  sum = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
  sum *= scale;
  sum = take_sqrt ? sqrtf(sum) : sum;
  float* dst = (out1 && idx >= split) ? out1 + (idx - split) : out + idx;   // two destinations: [0, split) | [split, ..)
  *dst = accumulate ? *dst + sum : sum;
}

// Column sums of a short, wide matrix (rows <= a few hundred, e.g. the G split-K partials of a weight
// gradient): 4 waves split the rows of 64 consecutive columns, combined through LDS in wave order.
template <typename T>
__global__ void __launch_bounds__(256) k_colsum_short(const T* __restrict__ x, float* __restrict__ out, int rows,
                                                      int64_t D, float* __restrict__ out1 = nullptr, int64_t split = 0,
                                                      int accumulate = 0) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  float sum = 0.f;
  if (c < D) {
#pragma unroll 16
    for (int r = grp; r < rows; r += 4) {
      float v[1];
      RowIO<T, 1>::load(x + (int64_t)r * D + c, v);
      sum += v[0];
    }
  }
  red[grp][lane] = sum;
  __syncthreads();
  if (grp == 0 && c < D) {
    const float v = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    float* dst = (out1 && c >= split) ? out1 + (c - split) : out + c;
    *dst = accumulate ? *dst + v : v;
  }
}

// Many short column sums in ONE launch (blockIdx.y = job), each ADDED into its two destinations exactly like
// k_colsum_short<float> with accumulate = 1 (same row split, same combine order -> bit-identical results): the
// [dW | db] partials of every small Linear of a training step, folded once at the end of backward instead of by
// one 3.7 us launch per layer.
struct ColJob { const float* x; float* out0; float* out1; int64_t rows, D, split, stride; };   // stride: floats per row of x
__global__ void __launch_bounds__(256) k_colsum_short_multi(const ColJob* __restrict__ jobs) {
  __shared__ float red[4][64];
  const ColJob j = jobs[blockIdx.y];
  if ((int64_t)blockIdx.x * 64 >= j.D) return;              // block-uniform
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  float sum = 0.f;
  if (c < j.D) {
#pragma unroll 16
    for (int r = grp; r < (int)j.rows; r += 4) sum += j.x[(int64_t)r * j.stride + c];
  }
  red[grp][lane] = sum;
  __syncthreads();
  if (grp == 0 && c < j.D) {
    const float v = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    float* dst = c >= j.split ? j.out1 + (c - j.split) : j.out0 + c;
    *dst = *dst + v;
  }
}

// Fold of the vocabulary projection's weight-gradient partials: part (G, V, E + A) fp32, G token-slab partials of
// dl^T [h | 1 ... 1] (A ones columns appended to h so that d b rides on the same product).  dW[v][e] += sum_g
// part[g][v][e], db[v] += sum_g part[g][v][E], fixed order: one launch instead of a column sum + two accumulations.
__global__ void __launch_bounds__(256) k_fold_aug(const float* __restrict__ part, int G, int64_t V, int E, int A,
                                                  float* __restrict__ dW, float* __restrict__ db) {
  const int64_t W = E + A, n = V * W;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int64_t v = idx / W;
  const int e = (int)(idx - v * W);
  if (e > E) return;                      // the other ones columns repeat d b
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += part[(int64_t)g * n + idx];
  if (e < E) dW[v * E + e] += s;
  else db[v] += s;
}

// Embedding weight gradient (torch.nn.Embedding inside the reference's Embedder, the first op of VAELmModel.encode):
// dW[v] = sum over tokens with id v of dy[token], deterministic without a sort and without atomics.  A block owns
// kEmbRows consecutive vocabulary ids; its 4 waves scan one contiguous quarter of the token ids each (all blocks read
// the same id array: it stays in L2), keep the matching tokens in a wave-private LDS list IN TOKEN ORDER, and add
// their dy rows into a wave-private LDS accumulator in that order; the 4 accumulators are combined in wave order.
// The library path sorts the 16 k ids first (radix sort + merges + segmented sums: ~80 us at the bench shape).
constexpr int kEmbCap = 2048;
struct EmbArgs { const float* dy; const int64_t* ids; float* dW; int64_t N; int V, E, pad, accumulate; };
// ROWS vocabulary ids per block (32 for E <= 128, 16 for E <= 256, 8 for E <= 512: the accumulators fill the LDS);
// HC = 64-column chunks per row.
template <int ROWS, int HC>
__global__ void __launch_bounds__(256) k_embed_bwd(EmbArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int E = a.E;
  float* acc = smem + wave * ROWS * E;                                           // [ROWS][E]
  int* list = reinterpret_cast<int*>(smem + 4 * ROWS * E) + wave * kEmbCap;      // token * ROWS + row
  const int base = blockIdx.x * ROWS;
  for (int i = lane; i < ROWS * E; i += 64) acc[i] = 0.f;
  for (int64_t p0 = 0; p0 < a.N; p0 += 4 * kEmbCap) {
    const int64_t lo = p0 + (int64_t)wave * kEmbCap;
    int64_t hi = lo + kEmbCap;
    if (hi > a.N) hi = a.N;
    int cnt = 0;
    for (int64_t t0 = lo; t0 < hi; t0 += 64 * 8) {                               // wave-uniform trip count
      int64_t id[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {                                              // 8 independent loads in flight
        const int64_t tok = t0 + 64 * u + lane;
        id[u] = tok < hi ? a.ids[tok] : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t tok = t0 + 64 * u + lane;
        const bool match = id[u] >= base && id[u] < base + ROWS && id[u] != a.pad && id[u] < a.V;
        const unsigned long long bal = __ballot(match);
        if (match) list[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = (int)(tok * ROWS + (id[u] - base));
        cnt += __popcll(bal);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int RIF = HC <= 2 ? 8 : 4;                                         // dy rows in flight, added in order
    for (int j = 0; j < cnt; j += RIF) {
      int e[RIF];
      float v[RIF][HC];
#pragma unroll
      for (int u = 0; u < RIF; ++u) {
        e[u] = j + u < cnt ? list[j + u] : -1;
#pragma unroll
        for (int h = 0; h < HC; ++h) {
          const int c = lane + 64 * h;
          v[u][h] = (e[u] >= 0 && c < E) ? a.dy[(int64_t)(e[u] / ROWS) * E + c] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < RIF; ++u)
        if (e[u] >= 0) {
#pragma unroll
          for (int h = 0; h < HC; ++h) {
            const int c = lane + 64 * h;
            if (c < E) acc[(e[u] % ROWS) * E + c] += v[u][h];
          }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  const int RE = ROWS * E;
  for (int i = threadIdx.x; i < RE; i += 256) {
    const int v = base + i / E;
    if (v >= a.V) continue;
    const float s4 = ((smem[i] + smem[RE + i]) + smem[2 * RE + i]) + smem[3 * RE + i];
    float* dst = a.dW + (int64_t)v * E + (i % E);
    *dst = a.accumulate ? *dst + s4 : s4;
  }
}

// ------------------------------------------------------------------------------------------------
// a16: DefaultKLLoss (kl_loss.py:38-49): kl = clamp(logq - logp, min=free_bits)
__global__ void k_kl_fwd(const float* __restrict__ logq, const float* __restrict__ logp,
                         float* __restrict__ kl, int64_t n, int use_fb, float fb) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = logq[i] - logp[i];
  if (use_fb) v = v < fb ? fb : v;   // NaN propagates like torch.clamp
  kl[i] = v;
}
__global__ void k_kl_bwd(const float* __restrict__ logq, const float* __restrict__ logp,
                         const float* __restrict__ dkl, float* __restrict__ dq,
                         float* __restrict__ dp, int64_t n, int use_fb, float fb) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = logq[i] - logp[i];
  float g = (use_fb && !(v >= fb)) ? 0.f : dkl[i];
  dq[i] = g;
  dp[i] = -g;
}

// a17: ELBO tail (models/base/model.py:171-193).  Single block; out = {batch-loss, batch-kl, batch-nll}.
__device__ float block_sum_256(float v, float* red) {
  v = group_sum(v, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return t;
}

// wdev (optional): {recon weight, kl weight} in device memory, read instead of the by-value arguments -- a captured
// HIP graph replays with whatever the host wrote there before the replay (annealed weight schedulers)
__global__ void k_elbo_fwd(const float* __restrict__ recon, const float* __restrict__ kl,
                           float* __restrict__ out, int B, int n, float wr, float wk, int iwae,
                           const float* __restrict__ wdev) {
  __shared__ float red[8];
  if (wdev) { wr = wdev[0]; wk = wdev[1]; }
  float a_loss = 0.f, a_kl = 0.f, a_nll = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float mx = -INFINITY;
    for (int j = 0; j < n; ++j) {
      float r = recon[b * n + j], k = kl[b * n + j];
      a_kl += k;
      a_nll += r;
      float l = wr * r + wk * k;
      if (iwae) mx = fmaxf(mx, l);
      else a_loss += l;
    }
    if (iwae) {
      float den = 0.f, num = 0.f;
      for (int j = 0; j < n; ++j) {
        float l = wr * recon[b * n + j] + wk * kl[b * n + j];
        float e = expf(l - mx);
        den += e;
        num += e * l;
      }
      a_loss += num / den;
    }
  }
  float t_loss = block_sum_256(a_loss, red);
  float t_kl = block_sum_256(a_kl, red);
  float t_nll = block_sum_256(a_nll, red);
  if (threadIdx.x == 0) {
    out[0] = t_loss / (iwae ? (float)B : (float)B * n);
    out[1] = t_kl / ((float)B * n);
    out[2] = t_nll / ((float)B * n);
  }
}

// gout = {d batch-loss, d batch-kl, d batch-nll}
__global__ void k_elbo_bwd(const float* __restrict__ recon, const float* __restrict__ kl,
                           const float* __restrict__ gout, float* __restrict__ drecon,
                           float* __restrict__ dkl, int B, int n, float wr, float wk, int iwae,
                           const float* __restrict__ wdev) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (wdev) { wr = wdev[0]; wk = wdev[1]; }
  float gl = gout[0], gk = gout[1] / ((float)B * n), gn = gout[2] / ((float)B * n);
  if (!iwae) {
    float c = gl / ((float)B * n);
    for (int j = 0; j < n; ++j) {
      drecon[b * n + j] = wr * c + gn;
      dkl[b * n + j] = wk * c + gk;
    }
    return;
  }
  float mx = -INFINITY;
  for (int j = 0; j < n; ++j) mx = fmaxf(mx, wr * recon[b * n + j] + wk * kl[b * n + j]);
  float den = 0.f, num = 0.f;
  for (int j = 0; j < n; ++j) {
    float l = wr * recon[b * n + j] + wk * kl[b * n + j];
    float e = expf(l - mx);
    den += e;
    num += e * l;
  }
  float mean = num / den;
  for (int j = 0; j < n; ++j) {
    float l = wr * recon[b * n + j] + wk * kl[b * n + j];
    float w = expf(l - mx) / den;
    float c = gl / (float)B * w * (1.f + l - mean);
    drecon[b * n + j] = wr * c + gn;
    dkl[b * n + j] = wk * c + gk;
  }
}


// a5 / a15 reductions + a16 + a17 in ONE launch each way (launch-count pass): per-sentence sums of the
// per-token log q and log p (the k_segment_sum order: lanes stride over s, shuffle tree), DefaultKLLoss
// (kl_loss.py:38-49) and the ELBO tail (models/base/model.py:171-193, the k_elbo_fwd arithmetic on the first 256
// threads) by one block of 1024 threads; kl of every sentence is staged in LDS (dynamic, (B n) floats).
__global__ void __launch_bounds__(1024) k_tail_fwd(const float* __restrict__ tok_q, const float* __restrict__ tok_p,
                                                  const float* __restrict__ recon, float* __restrict__ logq,
                                                  float* __restrict__ logp, float* __restrict__ kl_out,
                                                  float* __restrict__ out, int B, int n, int S, int use_fb, float fb,
                                                  float wr, float wk, int iwae, const float* __restrict__ wdev,
                                                  int64_t* __restrict__ rng_state) {
  extern __shared__ float s_kl[];
  __shared__ float red[8];
  const int Bn = B * n;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  // four sentences of a wave in flight at a time (eight independent load streams): the single block is latency bound
  for (int bn0 = 4 * wave; bn0 < Bn; bn0 += 4 * nw) {
    float aq[4] = {0.f, 0.f, 0.f, 0.f}, ap[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += 64) {
      // unconditional loads from clamped addresses (all eight in flight), the out-of-range ones masked afterwards: a
      // load under a branch is followed by its own wait, and eight serialised round trips per iteration made this
      // single-block kernel take 13 us
      const int s = s0 + lane, sc = s < S ? s : S - 1;
      float vq[4], vp[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t row = bn0 + u < Bn ? bn0 + u : Bn - 1;
        vq[u] = tok_q[row * S + sc];
        vp[u] = tok_p[row * S + sc];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        aq[u] += s < S ? vq[u] : 0.f;
        ap[u] += s < S ? vp[u] : 0.f;
      }
    }
    // The eight wave sums (q and p of four sentences) as ONE transposing butterfly: at offsets 32, 16, 8 a lane keeps half
    // of its values and hands the other half to its partner, then three plain steps -- 10 shuffles instead of 48, every
    // addition pairing exactly the lanes group_sum() pairs (32, 16, ..., 1), so the sums are bit-identical to
    // k_segment_sum's.  (Eight separate reductions with a store branch between them ran as serial dependent chains: the
    // single block took 13 us.)  Value j = 2 u + {0: q, 1: p} ends up in the lanes with bits (5, 4, 3) = (j & 1, j >> 1 & 1, j >> 2).
    float v8[8] = {aq[0], ap[0], aq[1], ap[1], aq[2], ap[2], aq[3], ap[3]};
    float v4[4], v2[2];
    const bool h32 = (lane & 32) != 0, h16 = (lane & 16) != 0, h8 = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float got = __shfl_xor(h32 ? v8[2 * i] : v8[2 * i + 1], 32, 64);
      v4[i] = (h32 ? v8[2 * i + 1] : v8[2 * i]) + got;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float got = __shfl_xor(h16 ? v4[2 * i] : v4[2 * i + 1], 16, 64);
      v2[i] = (h16 ? v4[2 * i + 1] : v4[2 * i]) + got;
    }
    float y = (h8 ? v2[1] : v2[0]) + __shfl_xor(h8 ? v2[0] : v2[1], 8, 64);
    y += __shfl_xor(y, 4, 64);
    y += __shfl_xor(y, 2, 64);
    y += __shfl_xor(y, 1, 64);
    const float other = __shfl_xor(y, 32, 64);            // the p sum sits opposite the q sum of the same sentence
    const int u = (h16 ? 1 : 0) + (h8 ? 2 : 0);
    if ((lane & 7) == 0 && bn0 + u < Bn) {
      if (!h32) {
        float v = y - other;
        if (use_fb) v = v < fb ? fb : v;
        logq[bn0 + u] = y; kl_out[bn0 + u] = v;
        s_kl[bn0 + u] = v;
      } else {
        logp[bn0 + u] = y;
      }
    }
  }
  __syncthreads();
  if (wdev) { wr = wdev[0]; wk = wdev[1]; }
  float a_loss = 0.f, a_kl = 0.f, a_nll = 0.f;
  if (threadIdx.x < 256) {
    for (int b = threadIdx.x; b < B; b += 256) {
      float mx = -INFINITY;
      for (int j = 0; j < n; ++j) {
        float r = recon[b * n + j], k = s_kl[b * n + j];
        a_kl += k;
        a_nll += r;
        float l = wr * r + wk * k;
        if (iwae) mx = fmaxf(mx, l);
        else a_loss += l;
      }
      if (iwae) {
        float den = 0.f, num = 0.f;
        for (int j = 0; j < n; ++j) {
          float l = wr * recon[b * n + j] + wk * s_kl[b * n + j];
          float e = expf(l - mx);
          den += e;
          num += e * l;
        }
        a_loss += num / den;
      }
    }
  }
  // the three sums over the first four waves, in k_elbo_fwd's order
  float sums[3] = {a_loss, a_kl, a_nll};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    float v = group_sum(sums[q], 64);
    __syncthreads();
    if (lane == 0 && wave < 4) red[wave] = v;
    __syncthreads();
    sums[q] = ((red[0] + red[1]) + red[2]) + red[3];
  }
  if (threadIdx.x == 0) {
    if (rng_state) rng_state[1] += 1;       // the posterior kernel of this forward drew from the state: one draw taken
    out[0] = sums[0] / (iwae ? (float)B : (float)B * n);
    out[1] = sums[1] / ((float)B * n);
    out[2] = sums[2] / ((float)B * n);
  }
}

// cotangents of batch-loss, batch-kl, batch-nll (one device scalar each, NULL = zero) -> d recon, d log q, d log p per
// sentence (k_elbo_bwd + k_kl_bwd)
__global__ void k_tail_bwd(const float* __restrict__ recon, const float* __restrict__ logq, const float* __restrict__ logp,
                           const float* __restrict__ g_loss, const float* __restrict__ g_kl,
                           const float* __restrict__ g_nll, float* __restrict__ drecon, float* __restrict__ dq,
                           float* __restrict__ dp, int B, int n, int use_fb, float fb, float wr, float wk, int iwae,
                           const float* __restrict__ wdev) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (wdev) { wr = wdev[0]; wk = wdev[1]; }
  float gl = g_loss ? g_loss[0] : 0.f, gk = (g_kl ? g_kl[0] : 0.f) / ((float)B * n),
        gn = (g_nll ? g_nll[0] : 0.f) / ((float)B * n);
  auto kl_of = [&](int i) { float v = logq[i] - logp[i]; if (use_fb) v = v < fb ? fb : v; return v; };
  auto put = [&](int i, float dr, float dk) {
    const float v = logq[i] - logp[i];
    const float g = (use_fb && !(v >= fb)) ? 0.f : dk;
    drecon[i] = dr; dq[i] = g; dp[i] = -g;
  };
  if (!iwae) {
    float c = gl / ((float)B * n);
    for (int j = 0; j < n; ++j) put(b * n + j, wr * c + gn, wk * c + gk);
    return;
  }
  float mx = -INFINITY;
  for (int j = 0; j < n; ++j) mx = fmaxf(mx, wr * recon[b * n + j] + wk * kl_of(b * n + j));
  float den = 0.f, num = 0.f;
  for (int j = 0; j < n; ++j) {
    float l = wr * recon[b * n + j] + wk * kl_of(b * n + j);
    float e = expf(l - mx);
    den += e;
    num += e * l;
  }
  float mean = num / den;
  for (int j = 0; j < n; ++j) {
    float l = wr * recon[b * n + j] + wk * kl_of(b * n + j);
    float w = expf(l - mx) / den;
    float c = gl / (float)B * w * (1.f + l - mean);
    put(b * n + j, wr * c + gn, wk * c + gk);
  }
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
#define S_(x) ((hipStream_t)(x))

extern "C" int mivae_tokens_mask(const int64_t* tokens, int64_t* mask, int64_t n, void* stream) {
  MIVAE_REQUIRE(n >= 0, "tokens_mask: n < 0");
  if (n == 0) return MIVAE_OK;
  MIVAE_REQUIRE(tokens && mask, "tokens_mask: null pointer");
  unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
  if (grid > 4096) grid = 4096;
  k_tokens_mask<<<grid, kBlock, 0, S_(stream)>>>(tokens, mask, nullptr, n);
  MIVAE_CHECK_LAUNCH("tokens_mask");
  return MIVAE_OK;
}

// mivae_tokens_mask that also writes the mask as fp32 (the per-token weights LabelSmoothingNLL multiplies by,
// models/base/losses.py:128-164): one launch instead of the mask kernel + a cast.
extern "C" int mivae_tokens_mask_f32(const int64_t* tokens, int64_t* mask, float* mask_f32, int64_t n, void* stream) {
  MIVAE_REQUIRE(n >= 0, "tokens_mask_f32: bad size");
  if (n == 0) return MIVAE_OK;
  MIVAE_REQUIRE(tokens && mask && mask_f32, "tokens_mask_f32: null pointer");
  unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
  if (grid > 4096) grid = 4096;
  k_tokens_mask<<<grid, kBlock, 0, S_(stream)>>>(tokens, mask, mask_f32, n);
  MIVAE_CHECK_LAUNCH("tokens_mask_f32");
  return MIVAE_OK;
}

// Additive attention bias of the transformer bracket from the token mask: 0 where the key may be attended, -inf
// elsewhere (key padding, and k > q when causal).  out (B, 1, 1, S), or (B, 1, S, S) when causal.  Replaces the
// mask.ne(0) / tril / & / where chain that MultiHeadAttention.forward (nn/transformer.py:75-89) rebuilds per layer.
template <typename T>
__global__ void k_attn_bias(const int64_t* __restrict__ mask, T* __restrict__ out, int64_t total, int S, int causal) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % S);
    bool keep;
    if (causal) {
      const int64_t bq = i / S;
      const int q = (int)(bq % S);
      keep = mask[(bq / S) * S + k] != 0 && k <= q;
    } else {
      keep = mask[i] != 0;
    }
    float v[1] = {keep ? 0.f : -INFINITY};
    RowIO<T, 1>::store(out + i, v);
  }
}

extern "C" int mivae_attn_bias(const int64_t* mask, void* out, int64_t B, int64_t S, int causal, int dtype, void* stream) {
  MIVAE_REQUIRE(B >= 0 && S >= 1 && S < (1 << 20), "attn_bias: bad shape B=%ld S=%ld", (long)B, (long)S);
  if (B == 0) return MIVAE_OK;
  MIVAE_REQUIRE(mask && out, "attn_bias: null pointer");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "attn_bias: bad dtype %d", dtype);
  const int64_t total = causal ? B * S * S : B * S;
  unsigned grid = (unsigned)((total + kBlock - 1) / kBlock);
  if (grid > 8192) grid = 8192;
  if (dtype == MIVAE_F32) k_attn_bias<float><<<grid, kBlock, 0, S_(stream)>>>(mask, (float*)out, total, (int)S, causal ? 1 : 0);
  else k_attn_bias<unsigned short><<<grid, kBlock, 0, S_(stream)>>>(mask, (unsigned short*)out, total, (int)S, causal ? 1 : 0);
  MIVAE_CHECK_LAUNCH("attn_bias");
  return MIVAE_OK;
}

// sigma head of the posterior: sigma = Hardtanh(Softplus(pre), lo, hi) (nonauto posterior.py:23-27: Softplus with
// beta = 1, threshold = 20, then Hardtanh(1e-4, 5)) and its gradient d pre = d sigma * sigmoid(pre) inside the open
// interval (lo, hi), 0 elsewhere -- one launch each instead of softplus / hardtanh / dtype-cast kernels.
template <typename T>
__global__ void k_sigma_head_fwd(const T* __restrict__ pre, T* __restrict__ sigma, int64_t n, float lo, float hi) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v[1];
    RowIO<T, 1>::load(pre + i, v);
    v[0] = fminf(fmaxf(softplus20(v[0]), lo), hi);
    RowIO<T, 1>::store(sigma + i, v);
  }
}
template <typename T>
__global__ void k_sigma_head_bwd(const T* __restrict__ pre, const T* __restrict__ dsigma, T* __restrict__ dpre, int64_t n,
                                 float lo, float hi) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v[1], g[1];
    RowIO<T, 1>::load(pre + i, v);
    RowIO<T, 1>::load(dsigma + i, g);
    const float sp = softplus20(v[0]);
    const float d = v[0] > 20.f ? 1.f : sigmoid_f(v[0]);
    g[0] = (sp > lo && sp < hi) ? g[0] * d : 0.f;
    RowIO<T, 1>::store(dpre + i, g);
  }
}

extern "C" int mivae_sigma_head_fwd(const void* pre, void* sigma, int64_t n, float lo, float hi, int dtype, void* stream) {
  MIVAE_REQUIRE(n >= 0 && lo <= hi, "sigma_head_fwd: bad arguments");
  if (n == 0) return MIVAE_OK;
  MIVAE_REQUIRE(pre && sigma, "sigma_head_fwd: null pointer");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "sigma_head_fwd: bad dtype %d", dtype);
  unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
  if (grid > 8192) grid = 8192;
  if (dtype == MIVAE_F32) k_sigma_head_fwd<float><<<grid, kBlock, 0, S_(stream)>>>((const float*)pre, (float*)sigma, n, lo, hi);
  else k_sigma_head_fwd<unsigned short><<<grid, kBlock, 0, S_(stream)>>>((const unsigned short*)pre, (unsigned short*)sigma, n, lo, hi);
  MIVAE_CHECK_LAUNCH("sigma_head_fwd");
  return MIVAE_OK;
}

extern "C" int mivae_sigma_head_bwd(const void* pre, const void* dsigma, void* dpre, int64_t n, float lo, float hi,
                                    int dtype, void* stream) {
  MIVAE_REQUIRE(n >= 0 && lo <= hi, "sigma_head_bwd: bad arguments");
  if (n == 0) return MIVAE_OK;
  MIVAE_REQUIRE(pre && dsigma && dpre, "sigma_head_bwd: null pointer");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "sigma_head_bwd: bad dtype %d", dtype);
  unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
  if (grid > 8192) grid = 8192;
  if (dtype == MIVAE_F32)
    k_sigma_head_bwd<float><<<grid, kBlock, 0, S_(stream)>>>((const float*)pre, (const float*)dsigma, (float*)dpre, n, lo, hi);
  else
    k_sigma_head_bwd<unsigned short><<<grid, kBlock, 0, S_(stream)>>>((const unsigned short*)pre, (const unsigned short*)dsigma,
                                                                      (unsigned short*)dpre, n, lo, hi);
  MIVAE_CHECK_LAUNCH("sigma_head_bwd");
  return MIVAE_OK;
}

// Entry of the transformer encoder / decoder: y = (x + pe[s]) * mask[b, s] (nn/transformer.py:199-202 then the mask
// multiply of encoders/transformer.py / decoders), and its input gradient dx = dy * mask.  pe is the fp32 sinusoid
// table (>= S rows of D); it is rounded to the latent dtype before the add, like `pe.to(x.dtype)` in the reference.
template <typename T, bool BWD>
__global__ void k_posenc_mask(const T* __restrict__ x, const float* __restrict__ pe, const int64_t* __restrict__ mask,
                              T* __restrict__ y, int64_t rows, int S, int D) {
  const int64_t total = rows * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / D;
    const int d = (int)(i - row * D);
    const float m = (float)mask[row];
    float v[1];
    RowIO<T, 1>::load(x + i, v);
    if (!BWD) {
      float p[1] = {pe[(row % S) * D + d]};
      if (sizeof(T) == 2) p[0] = bf16_to_f32(f32_to_bf16(p[0]));
      float t[1] = {v[0] + p[0]};
      if (sizeof(T) == 2) t[0] = bf16_to_f32(f32_to_bf16(t[0]));   // the reference's add rounds before the multiply
      v[0] = t[0] * m;
    } else {
      v[0] = v[0] * m;
    }
    RowIO<T, 1>::store(y + i, v);
  }
}

// backward = 1: x is dy, y is dx, pe unused
extern "C" int mivae_posenc_mask(const void* x, const float* pe, const int64_t* mask, void* y, int64_t B, int64_t S,
                                 int64_t D, int backward, int dtype, void* stream) {
  MIVAE_REQUIRE(B >= 0 && S >= 1 && D >= 1, "posenc_mask: bad shape");
  if (B == 0) return MIVAE_OK;
  MIVAE_REQUIRE(x && mask && y && (backward || pe), "posenc_mask: null pointer");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "posenc_mask: bad dtype %d", dtype);
  const int64_t rows = B * S, total = rows * D;
  unsigned grid = (unsigned)((total + kBlock - 1) / kBlock);
  if (grid > 16384) grid = 16384;
  hipStream_t st = S_(stream);
  if (dtype == MIVAE_F32) {
    if (backward) k_posenc_mask<float, true><<<grid, kBlock, 0, st>>>((const float*)x, pe, mask, (float*)y, rows, (int)S, (int)D);
    else k_posenc_mask<float, false><<<grid, kBlock, 0, st>>>((const float*)x, pe, mask, (float*)y, rows, (int)S, (int)D);
  } else {
    if (backward) k_posenc_mask<unsigned short, true><<<grid, kBlock, 0, st>>>((const unsigned short*)x, pe, mask, (unsigned short*)y, rows, (int)S, (int)D);
    else k_posenc_mask<unsigned short, false><<<grid, kBlock, 0, st>>>((const unsigned short*)x, pe, mask, (unsigned short*)y, rows, (int)S, (int)D);
  }
  MIVAE_CHECK_LAUNCH("posenc_mask");
  return MIVAE_OK;
}

extern "C" int mivae_gather_last(const void* out, const int64_t* mask, void* ctx, int64_t* last_idx,
                                 int64_t B, int64_t S, int64_t E, int dtype, void* stream) {
  MIVAE_REQUIRE(B >= 0 && S > 0 && E > 0, "gather_last: bad shape B=%ld S=%ld E=%ld", (long)B, (long)S, (long)E);
  if (B == 0) return MIVAE_OK;
  MIVAE_REQUIRE(out && mask && ctx, "gather_last: null pointer");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "gather_last: bad dtype %d", dtype);
  if (dtype == MIVAE_F32)
    k_gather_last<float><<<(unsigned)B, kBlock, 0, S_(stream)>>>((const float*)out, mask, (float*)ctx, last_idx, (int)S, (int)E);
  else
    k_gather_last<unsigned short><<<(unsigned)B, kBlock, 0, S_(stream)>>>((const unsigned short*)out, mask, (unsigned short*)ctx, last_idx, (int)S, (int)E);
  MIVAE_CHECK_LAUNCH("gather_last");
  return MIVAE_OK;
}

static int check_latent(const char* name, int64_t Bn, int64_t S, int64_t D, int dtype) {
  MIVAE_REQUIRE(Bn >= 0 && S > 0 && D > 0, "%s: bad shape Bn=%ld S=%ld D=%ld", name, (long)Bn, (long)S, (long)D);
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "%s: bad dtype %d", name, dtype);
  MIVAE_REQUIRE(Bn * S < (1ll << 40) && D < (1 << 24) && S < (1ll << 31), "%s: shape too large", name);
  return MIVAE_OK;
}


namespace {
// ---- typed launchers (void* -> T*) ---------------------------------------------------------------
template <typename T, int VEC>
void l_reparam_fwd(const void* mu, const void* sigma, const void* eps, const int64_t* mask, void* z,
                   float* tok, int64_t rows, int n, int S, int D, RowCfg c, hipStream_t st, const int64_t* rng) {
  k_reparam_fwd<T, VEC><<<c.grid, kBlock, 0, st>>>((const T*)mu, (const T*)sigma, (const T*)eps, mask,
                                                  (T*)z, tok, rows, n, S, D, c.L, rng);
}
template <typename T, int VEC>
void l_reparam_bwd(const void* dz, const float* dlogq, const void* sigma, const void* eps,
                   const int64_t* mask, void* dmu, void* dsigma, int64_t prows, int n, int S, int D,
                   RowCfg c, hipStream_t st, const int64_t* rng) {
  k_reparam_bwd<T, VEC><<<c.grid, kBlock, 0, st>>>((const T*)dz, dlogq, (const T*)sigma, (const T*)eps,
                                                  mask, (T*)dmu, (T*)dsigma, prows, n, S, D, c.L, rng);
}
template <typename T, int VEC>
void l_gauss_fwd(const void* z, const int64_t* mask, float* tok, int64_t rows, int mask_div, int S,
                 int D, float mu, float inv2var, float cst, RowCfg c, hipStream_t st) {
  k_gauss_logprob_fwd<T, VEC><<<c.grid, kBlock, 0, st>>>((const T*)z, mask, tok, rows, mask_div, S, D,
                                                        c.L, mu, inv2var, cst);
}
template <typename T, int VEC>
void l_gauss_bwd(const void* z, const int64_t* mask, const float* dlogp, void* dz, int64_t rows,
                 int mask_div, int S, int D, float mu, float invvar, RowCfg c, hipStream_t st) {
  k_gauss_logprob_bwd<T, VEC><<<c.grid, kBlock, 0, st>>>((const T*)z, mask, dlogp, (T*)dz, rows,
                                                        mask_div, S, D, c.L, mu, invvar);
}
template <typename T, int VEC>
void l_affine_fwd(const void* z, const void* s, const void* t, const int64_t* mask, void* out,
                  float* tok, int64_t rows, int mask_div, int S, int h, int inverse, RowCfg c,
                  hipStream_t st) {
  k_nice_affine_fwd<T, VEC><<<c.grid, kBlock, 0, st>>>((const T*)z, (const T*)s, (const T*)t, mask,
                                                      (T*)out, tok, rows, mask_div, S, h, c.L, inverse);
}
template <typename T, int VEC>
void l_affine_bwd(const void* dout, const float* dld, const void* z, const void* s, const void* t,
                  const int64_t* mask, void* dz, void* ds, void* dt, int64_t rows, int mask_div,
                  int S, int h, int inverse, RowCfg c, hipStream_t st) {
  k_nice_affine_bwd<T, VEC><<<c.grid, kBlock, 0, st>>>((const T*)dout, dld, (const T*)z, (const T*)s,
                                                      (const T*)t, mask, (T*)dz, (T*)ds, (T*)dt, rows,
                                                      mask_div, S, h, c.L, inverse);
}
template <typename T, int VEC>
void l_actnorm_fwd(const void* z, const float* ls, const float* bias, const int64_t* mask, void* out,
                   int64_t rows, int mask_div, int S, int D, int inverse, RowCfg c, hipStream_t st, float* logdet,
                   int64_t n_logdet, const int64_t* ld_mask) {
  const int rows_per_block = (kBlock / c.L) * kActRows;
  int64_t grid = (rows + rows_per_block - 1) / rows_per_block;
  if (logdet && (n_logdet + 3) / 4 > grid) grid = (n_logdet + 3) / 4;
  k_actnorm_fwd<T, VEC><<<(unsigned)grid, kBlock, 0, st>>>((const T*)z, ls, bias, mask, (T*)out, rows, mask_div, S, D,
                                                          c.L, inverse, logdet, n_logdet, ld_mask);
}
template <int MODE>
struct ColPartial {
  template <typename T, int VEC>
  static void run(const void* x, const void* dout, const float* aux, const int64_t* mask, void* dz,
                  float* partial, int64_t rows, int rows_per_block, int nblk, int mask_div, int S,
                  int D, int L, hipStream_t st) {
    constexpr int NV = MODE == 2 ? 2 : 1;
    size_t smem = (size_t)(kBlock / L) * NV * L * VEC * sizeof(float);
    k_col_partial<T, VEC, MODE><<<nblk, kBlock, smem, st>>>((const T*)x, (const T*)dout, aux, mask,
                                                           (T*)dz, partial, rows, rows_per_block,
                                                           mask_div, S, D, L);
  }
};
#define DISPATCH_COL(dtype, vec, MODE, ...)                                             \
  do {                                                                                  \
    if ((dtype) == MIVAE_F32) {                                                         \
      if ((vec) == 4) ColPartial<MODE>::run<float, 4>(__VA_ARGS__);                     \
      else ColPartial<MODE>::run<float, 1>(__VA_ARGS__);                                \
    } else {                                                                            \
      if ((vec) == 8) ColPartial<MODE>::run<unsigned short, 8>(__VA_ARGS__);            \
      else ColPartial<MODE>::run<unsigned short, 1>(__VA_ARGS__);                       \
    }                                                                                   \
  } while (0)

// column-reduction geometry shared by the size query and the launches
struct ColCfg { int vec; int L; int nblk; int rows_per_block; };
static ColCfg col_cfg(int64_t rows, int64_t D, int dtype, bool aligned) {
  ColCfg c;
  int vmax = dtype == MIVAE_BF16 ? 8 : 4;
  c.vec = (aligned && D % vmax == 0) ? vmax : 1;
  c.L = lanes_per_row(D, c.vec);
  int64_t nblk = (rows + 63) / 64;        // >= 64 rows per block: 256 blocks at 16k tokens
  if (nblk > 1024) nblk = 1024;
  if (nblk < 1) nblk = 1;
  c.rows_per_block = (int)((rows + nblk - 1) / nblk);
  if (c.rows_per_block < 1) c.rows_per_block = 1;
  c.nblk = (int)((rows + c.rows_per_block - 1) / c.rows_per_block);
  if (c.nblk < 1) c.nblk = 1;
  return c;
}
}  // namespace

static int reparam_fwd_impl(const char* name, const void* mu, const void* sigma, const void* eps, const int64_t* rng,
                            const int64_t* mask, void* z, float* tok_logq, int64_t B, int64_t n, int64_t S, int64_t D,
                            int dtype, void* stream) {
  int rc = check_latent(name, B * n, S, D, dtype);
  if (rc) return rc;
  MIVAE_REQUIRE(n >= 1, "%s: samples must be >= 1", name);
  if (B == 0) return MIVAE_OK;
  MIVAE_REQUIRE(mu && sigma && z && tok_logq, "%s: null pointer", name);
  int64_t rows = B * n * S;
  bool al = aligned16(mu) && aligned16(sigma) && aligned16(z) && (!eps || aligned16(eps));
  RowCfg c = row_cfg(rows, D, dtype, al);
  DISPATCH(dtype, c.vec, l_reparam_fwd, mu, sigma, eps, mask, z, tok_logq, rows, (int)n, (int)S,
           (int)D, c, S_(stream), rng);
  MIVAE_CHECK_LAUNCH(name);
  return MIVAE_OK;
}

static int reparam_bwd_impl(const char* name, const void* dz, const float* dlogq, const void* sigma, const void* eps,
                            const int64_t* rng, const int64_t* mask, void* dmu, void* dsigma, int64_t B, int64_t n,
                            int64_t S, int64_t D, int dtype, void* stream) {
  int rc = check_latent(name, B * n, S, D, dtype);
  if (rc) return rc;
  MIVAE_REQUIRE(n >= 1, "%s: samples must be >= 1", name);
  if (B == 0) return MIVAE_OK;
  MIVAE_REQUIRE(sigma && dmu && dsigma, "%s: null pointer", name);
  int64_t prows = B * S;
  bool al = aligned16(sigma) && aligned16(dmu) && aligned16(dsigma) && (!dz || aligned16(dz)) &&
            (!eps || aligned16(eps));
  RowCfg c = row_cfg(prows, D, dtype, al);
  DISPATCH(dtype, c.vec, l_reparam_bwd, dz, dlogq, sigma, eps, mask, dmu, dsigma, prows, (int)n,
           (int)S, (int)D, c, S_(stream), rng);
  MIVAE_CHECK_LAUNCH(name);
  return MIVAE_OK;
}

extern "C" int mivae_reparam_logq_fwd(const void* mu, const void* sigma, const void* eps,
                                      const int64_t* mask, void* z, float* tok_logq, int64_t B,
                                      int64_t n, int64_t S, int64_t D, int dtype, void* stream) {
  return reparam_fwd_impl("reparam_logq_fwd", mu, sigma, eps, nullptr, mask, z, tok_logq, B, n, S, D, dtype, stream);
}

extern "C" int mivae_reparam_logq_bwd(const void* dz, const float* dlogq, const void* sigma,
                                      const void* eps, const int64_t* mask, void* dmu, void* dsigma,
                                      int64_t B, int64_t n, int64_t S, int64_t D, int dtype,
                                      void* stream) {
  return reparam_bwd_impl("reparam_logq_bwd", dz, dlogq, sigma, eps, nullptr, mask, dmu, dsigma, B, n, S, D, dtype,
                          stream);
}

// The same pair with eps drawn inside the kernels.  rng_state: device int64 {seed, offset} of the stream of draws
// (the caller's generator); the forward COPIES it to rng_used (device int64[2], kept with the saved tensors), draws
// eps = Normal(0, 1)(seed, offset, element) and then advances rng_state's offset by one, all on `stream`, so that a
// captured HIP graph draws fresh noise at every replay.  The backward regenerates the draw from rng_used.
extern "C" int mivae_reparam_logq_fwd_rng(const void* mu, const void* sigma, int64_t* rng_state, int64_t* rng_used,
                                          const int64_t* mask, void* z, float* tok_logq, int64_t B, int64_t n,
                                          int64_t S, int64_t D, int dtype, void* stream) {
  MIVAE_REQUIRE(rng_state && rng_used, "reparam_logq_fwd_rng: null generator state");
  k_rng_take<<<1, 64, 0, S_(stream)>>>(rng_state, rng_used);
  MIVAE_CHECK_LAUNCH("reparam_logq_fwd_rng(take)");
  return reparam_fwd_impl("reparam_logq_fwd_rng", mu, sigma, nullptr, rng_used, mask, z, tok_logq, B, n, S, D, dtype,
                          stream);
}

extern "C" int mivae_reparam_logq_bwd_rng(const void* dz, const float* dlogq, const void* sigma,
                                          const int64_t* rng_used, const int64_t* mask, void* dmu, void* dsigma,
                                          int64_t B, int64_t n, int64_t S, int64_t D, int dtype, void* stream) {
  MIVAE_REQUIRE(rng_used, "reparam_logq_bwd_rng: null generator state");
  return reparam_bwd_impl("reparam_logq_bwd_rng", dz, dlogq, sigma, nullptr, rng_used, mask, dmu, dsigma, B, n, S, D,
                          dtype, stream);
}

// ---- fused posterior sample (sigma head + sample + log q + generator bookkeeping): one launch each way --------------
namespace {
template <typename T, int VEC>
void l_post_fwd(const void* mu, const void* sg, int pre, const void* eps, const int64_t* mask, void* z, void* sigma_out,
                float* tok, int64_t rows, int n, int S, int D, float lo, float hi, const int64_t* rng_state, int64_t* rng_used,
                RowCfg c, hipStream_t st) {
  if (pre)
    k_post_fwd<T, VEC, true><<<c.grid, kBlock, 0, st>>>((const T*)mu, (const T*)sg, (const T*)eps, mask, (T*)z,
                                                       (T*)sigma_out, tok, rows, n, S, D, c.L, lo, hi, rng_state, rng_used);
  else
    k_post_fwd<T, VEC, false><<<c.grid, kBlock, 0, st>>>((const T*)mu, (const T*)sg, (const T*)eps, mask, (T*)z,
                                                        (T*)sigma_out, tok, rows, n, S, D, c.L, lo, hi, rng_state, rng_used);
}
template <typename T, int VEC>
void l_post_bwd(const void* dz, const float* dlogq, int dq_tok, const void* sg, int pre, const void* eps,
                const int64_t* mask, void* dmu, void* dsg, int64_t prows, int n, int S, int D, float lo, float hi,
                const int64_t* rng_used, RowCfg c, hipStream_t st) {
  if (pre)
    k_post_bwd<T, VEC, true><<<c.grid, kBlock, 0, st>>>((const T*)dz, dlogq, dq_tok, (const T*)sg, (const T*)eps, mask,
                                                       (T*)dmu, (T*)dsg, prows, n, S, D, c.L, lo, hi, rng_used);
  else
    k_post_bwd<T, VEC, false><<<c.grid, kBlock, 0, st>>>((const T*)dz, dlogq, dq_tok, (const T*)sg, (const T*)eps, mask,
                                                        (T*)dmu, (T*)dsg, prows, n, S, D, c.L, lo, hi, rng_used);
}
}  // namespace

// Posterior sample + log q with the sigma head inside (a3 tail + a4 + a5, posterior.py:23-27, 86-172) in ONE launch.
//   sigma_or_pre (B, S, D): sigma itself (sigma_is_pre = 0) or the sigma head's pre-activation (1: sigma =
//   Hardtanh(Softplus(pre), lo, hi), written to sigma_out (B, S, D) when that is not NULL);
//   noise: eps ((B n), S, D) injected, or drawn in-kernel when rng_state (device int64[2] {seed, offset}) is given -- the
//   launch copies the state to rng_used (int64[2], kept for the backward) and does NOT advance it: follow up with
//   mivae_rng_advance(rng_state) or hand rng_state to mivae_latent_tail_fwd of the same forward; both NULL: eps = 0.
//   tok_logq ((B n) S) fp32 per-token log q; sum it per sentence with mivae_segment_sum or mivae_latent_tail_fwd.
extern "C" int mivae_posterior_fwd(const void* mu, const void* sigma_or_pre, int sigma_is_pre, float lo, float hi,
                                   const void* eps, const int64_t* rng_state, int64_t* rng_used, const int64_t* mask, void* z,
                                   void* sigma_out, float* tok_logq, int64_t B, int64_t n, int64_t S, int64_t D,
                                   int dtype, void* stream) {
  int rc = check_latent("posterior_fwd", B * n, S, D, dtype);
  if (rc) return rc;
  MIVAE_REQUIRE(n >= 1 && lo <= hi, "posterior_fwd: bad arguments");
  MIVAE_REQUIRE(!(eps && rng_state) && (!rng_state || rng_used), "posterior_fwd: eps and rng_state exclude each other; "
                "rng_state needs rng_used");
  if (B == 0) return MIVAE_OK;
  MIVAE_REQUIRE(mu && sigma_or_pre && z && tok_logq, "posterior_fwd: null pointer");
  int64_t rows = B * n * S;
  bool al = aligned16(mu) && aligned16(sigma_or_pre) && aligned16(z) && (!eps || aligned16(eps)) &&
            (!sigma_out || aligned16(sigma_out));
  RowCfg c = row_cfg(rows, D, dtype, al);
  DISPATCH(dtype, c.vec, l_post_fwd, mu, sigma_or_pre, sigma_is_pre, eps, mask, z, sigma_out, tok_logq, rows, (int)n,
           (int)S, (int)D, lo, hi, rng_state, rng_used, c, S_(stream));
  MIVAE_CHECK_LAUNCH("posterior_fwd");
  return MIVAE_OK;
}

// Its backward: d mu and d sigma (or d pre when sigma_is_pre) (B, S, D).  dlogq: per sentence (B n) (dlogq_per_token =
// 0) or per token ((B n) S) (1), or NULL; dz ((B n), S, D) or NULL; noise as in the forward (eps, or rng_used).
extern "C" int mivae_posterior_bwd(const void* dz, const float* dlogq, int dlogq_per_token, const void* sigma_or_pre,
                                   int sigma_is_pre, float lo, float hi, const void* eps, const int64_t* rng_used,
                                   const int64_t* mask, void* dmu, void* dsigma, int64_t B, int64_t n, int64_t S,
                                   int64_t D, int dtype, void* stream) {
  int rc = check_latent("posterior_bwd", B * n, S, D, dtype);
  if (rc) return rc;
  MIVAE_REQUIRE(n >= 1 && lo <= hi && !(eps && rng_used), "posterior_bwd: bad arguments");
  if (B == 0) return MIVAE_OK;
  MIVAE_REQUIRE(sigma_or_pre && dmu && dsigma, "posterior_bwd: null pointer");
  int64_t prows = B * S;
  bool al = aligned16(sigma_or_pre) && aligned16(dmu) && aligned16(dsigma) && (!dz || aligned16(dz)) &&
            (!eps || aligned16(eps));
  RowCfg c = row_cfg(prows, D, dtype, al);
  DISPATCH(dtype, c.vec, l_post_bwd, dz, dlogq, dlogq_per_token ? 1 : 0, sigma_or_pre, sigma_is_pre, eps, mask, dmu,
           dsigma, prows, (int)n, (int)S, (int)D, lo, hi, rng_used, c, S_(stream));
  MIVAE_CHECK_LAUNCH("posterior_bwd");
  return MIVAE_OK;
}

// Per-sentence log q / log p (sums of the per-token values), DefaultKLLoss and the ELBO tail in ONE launch
// (posterior.py:157-172 reduction, prior.py:148-158 reduction, nn/kl_loss.py:38-49, models/base/model.py:171-193).
//   tok_logq, tok_logp ((B n), S) fp32; recon (B n); outputs logq, logp, kl (B n) and out3 = {batch-loss, batch-kl,
//   batch-nll}.  weights_dev as mivae_elbo_fwd.  rng_state (or NULL): generator state whose offset this launch advances by
//   one (the draw mivae_posterior_fwd of the same forward took).  Limits (one block): B n <= 16384, (B n) S <= 2^21.
extern "C" int mivae_latent_tail_supported(int64_t B, int64_t n, int64_t S) {
  return B > 0 && n > 0 && S > 0 && B * n <= 16384 && B * n * S <= (1ll << 21) ? 1 : 0;
}
extern "C" int mivae_latent_tail_fwd(const float* tok_logq, const float* tok_logp, const float* recon, float* logq,
                                     float* logp, float* kl, float* out3, int64_t B, int64_t n, int64_t S, int use_fb,
                                     float free_bits, float recon_weight, float kl_weight, int iwae,
                                     const float* weights_dev, int64_t* rng_state, void* stream) {
  MIVAE_REQUIRE(mivae_latent_tail_supported(B, n, S), "latent_tail_fwd: unsupported shape B=%ld n=%ld S=%ld", (long)B,
                (long)n, (long)S);
  MIVAE_REQUIRE(tok_logq && tok_logp && recon && logq && logp && kl && out3, "latent_tail_fwd: null pointer");
  const size_t lds = (size_t)(B * n) * sizeof(float);
  k_tail_fwd<<<1, 1024, lds, S_(stream)>>>(tok_logq, tok_logp, recon, logq, logp, kl, out3, (int)B, (int)n, (int)S,
                                           use_fb, free_bits, recon_weight, kl_weight, iwae, weights_dev, rng_state);
  MIVAE_CHECK_LAUNCH("latent_tail_fwd");
  return MIVAE_OK;
}
extern "C" int mivae_latent_tail_bwd(const float* recon, const float* logq, const float* logp, const float* g_loss,
                                     const float* g_kl, const float* g_nll, float* drecon, float* dlogq, float* dlogp,
                                     int64_t B, int64_t n, int use_fb, float free_bits, float recon_weight,
                                     float kl_weight, int iwae, const float* weights_dev, void* stream) {
  MIVAE_REQUIRE(B > 0 && n > 0 && B * n < (1ll << 30), "latent_tail_bwd: bad shape");
  MIVAE_REQUIRE(recon && logq && logp && drecon && dlogq && dlogp, "latent_tail_bwd: null pointer");
  k_tail_bwd<<<(unsigned)((B + kBlock - 1) / kBlock), kBlock, 0, S_(stream)>>>(recon, logq, logp, g_loss, g_kl, g_nll, drecon, dlogq,
                                                                              dlogp, (int)B, (int)n, use_fb, free_bits,
                                                                              recon_weight, kl_weight, iwae, weights_dev);
  MIVAE_CHECK_LAUNCH("latent_tail_bwd");
  return MIVAE_OK;
}

// state.offset += 1: one draw taken from the generator (stream ordered, one thread)
extern "C" int mivae_rng_advance(int64_t* rng_state, void* stream) {
  MIVAE_REQUIRE(rng_state, "rng_advance: null generator state");
  k_rng_take<<<1, 64, 0, S_(stream)>>>(rng_state, nullptr);
  MIVAE_CHECK_LAUNCH("rng_advance");
  return MIVAE_OK;
}

// eps ((B n), S, D) fp32 of the draw recorded in rng_used: what the two kernels above used (tests, debugging)
extern "C" int mivae_philox_normal(const int64_t* rng_used, float* out, int64_t count, void* stream) {
  MIVAE_REQUIRE(count >= 0 && (count == 0 || (rng_used && out)), "philox_normal: bad arguments");
  if (count == 0) return MIVAE_OK;
  k_philox_normal<<<(unsigned)((count + 4 * kBlock - 1) / (4 * kBlock)), kBlock, 0, S_(stream)>>>(rng_used, out, count);
  MIVAE_CHECK_LAUNCH("philox_normal");
  return MIVAE_OK;
}

extern "C" int mivae_gauss_logprob_fwd(const void* z, const int64_t* mask, float* tok, int64_t Bn,
                                       int64_t mask_div, int64_t S, int64_t D, float mu, float std,
                                       int dtype, void* stream) {
  int rc = check_latent("gauss_logprob_fwd", Bn, S, D, dtype);
  if (rc) return rc;
  MIVAE_REQUIRE(std > 0.f && mask_div >= 1, "gauss_logprob_fwd: std must be > 0, mask_div >= 1");
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(z && tok, "gauss_logprob_fwd: null pointer");
  int64_t rows = Bn * S;
  RowCfg c = row_cfg(rows, D, dtype, aligned16(z));
  float inv2var = 1.0f / (2.0f * std * std);
  float cst = logf(std) + 0.5f * kLog2Pi;
  DISPATCH(dtype, c.vec, l_gauss_fwd, z, mask, tok, rows, (int)mask_div, (int)S, (int)D, mu, inv2var,
           cst, c, S_(stream));
  MIVAE_CHECK_LAUNCH("gauss_logprob_fwd");
  return MIVAE_OK;
}

extern "C" int mivae_gauss_logprob_bwd(const void* z, const int64_t* mask, const float* dlogp,
                                       void* dz, int64_t Bn, int64_t mask_div, int64_t S, int64_t D,
                                       float mu, float std, int dtype, void* stream) {
  int rc = check_latent("gauss_logprob_bwd", Bn, S, D, dtype);
  if (rc) return rc;
  MIVAE_REQUIRE(std > 0.f && mask_div >= 1, "gauss_logprob_bwd: std must be > 0, mask_div >= 1");
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(z && dlogp && dz, "gauss_logprob_bwd: null pointer");
  int64_t rows = Bn * S;
  RowCfg c = row_cfg(rows, D, dtype, aligned16(z) && aligned16(dz));
  DISPATCH(dtype, c.vec, l_gauss_bwd, z, mask, dlogp, dz, rows, (int)mask_div, (int)S, (int)D, mu,
           1.0f / (std * std), c, S_(stream));
  MIVAE_CHECK_LAUNCH("gauss_logprob_bwd");
  return MIVAE_OK;
}

extern "C" int mivae_segment_sum(const float* tok, const float* add, float* out, int64_t nseg,
                                 int64_t S, void* stream) {
  MIVAE_REQUIRE(nseg >= 0 && S > 0 && S < (1ll << 31), "segment_sum: bad shape");
  if (nseg == 0) return MIVAE_OK;
  MIVAE_REQUIRE(tok && out, "segment_sum: null pointer");
  unsigned grid = (unsigned)((nseg + 3) / 4);
  k_segment_sum<<<grid, kBlock, 0, S_(stream)>>>(tok, add, out, nseg, (int)S);
  MIVAE_CHECK_LAUNCH("segment_sum");
  return MIVAE_OK;
}

extern "C" int mivae_mask_count(const int64_t* mask, int64_t* count, int64_t Bn, int64_t mask_div,
                                int64_t S, void* stream) {
  MIVAE_REQUIRE(Bn >= 0 && S > 0 && mask_div >= 1, "mask_count: bad shape");
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(mask && count, "mask_count: null pointer");
  unsigned grid = (unsigned)((Bn + 3) / 4);
  k_mask_count<<<grid, kBlock, 0, S_(stream)>>>(mask, count, Bn, (int)mask_div, (int)S);
  MIVAE_CHECK_LAUNCH("mask_count");
  return MIVAE_OK;
}

extern "C" int mivae_nice_affine_fwd(const void* z, const void* s, const void* t,
                                     const int64_t* mask, void* out, float* tok_ld, int64_t Bn,
                                     int64_t mask_div, int64_t S, int64_t D, int dtype, int inverse,
                                     void* stream) {
  int rc = check_latent("nice_affine_fwd", Bn, S, D, dtype);
  if (rc) return rc;
  MIVAE_REQUIRE(D % 2 == 0 && mask_div >= 1, "nice_affine_fwd: D must be even");
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(z && s && t && out && tok_ld, "nice_affine_fwd: null pointer");
  int64_t rows = Bn * S, h = D / 2;
  bool al = aligned16(z) && aligned16(s) && aligned16(t) && aligned16(out);
  RowCfg c = row_cfg(rows, h, dtype, al);
  DISPATCH(dtype, c.vec, l_affine_fwd, z, s, t, mask, out, tok_ld, rows, (int)mask_div, (int)S,
           (int)h, inverse, c, S_(stream));
  MIVAE_CHECK_LAUNCH("nice_affine_fwd");
  return MIVAE_OK;
}

extern "C" int mivae_nice_affine_bwd(const void* dout, const float* dld, const void* z,
                                     const void* s, const void* t, const int64_t* mask, void* dz,
                                     void* ds, void* dt, int64_t Bn, int64_t mask_div, int64_t S,
                                     int64_t D, int dtype, int inverse, void* stream) {
  int rc = check_latent("nice_affine_bwd", Bn, S, D, dtype);
  if (rc) return rc;
  MIVAE_REQUIRE(D % 2 == 0 && mask_div >= 1, "nice_affine_bwd: D must be even");
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(dout && z && s && t && dz && ds && dt, "nice_affine_bwd: null pointer");
  int64_t rows = Bn * S, h = D / 2;
  bool al = aligned16(dout) && aligned16(z) && aligned16(s) && aligned16(t) && aligned16(dz) &&
            aligned16(ds) && aligned16(dt);
  RowCfg c = row_cfg(rows, h, dtype, al);
  DISPATCH(dtype, c.vec, l_affine_bwd, dout, dld, z, s, t, mask, dz, ds, dt, rows, (int)mask_div,
           (int)S, (int)h, inverse, c, S_(stream));
  MIVAE_CHECK_LAUNCH("nice_affine_bwd");
  return MIVAE_OK;
}

extern "C" int mivae_actnorm_fwd(const void* z, const float* log_scale, const float* bias,
                                 const int64_t* mask, void* out, float* logdet, int64_t Bn,
                                 int64_t mask_div, int64_t S, int64_t D, int64_t n_logdet,
                                 int count_weighted, int dtype, int inverse, void* stream) {
  int rc = check_latent("actnorm_fwd", Bn, S, D, dtype);
  if (rc) return rc;
  MIVAE_REQUIRE(mask_div >= 1, "actnorm_fwd: mask_div >= 1");
  MIVAE_REQUIRE(log_scale && bias, "actnorm_fwd: null parameter pointer");
  const bool want_ld = logdet && n_logdet > 0;
  // n_logdet = Bn and count_weighted = 1 for 3-D input with a mask, else 1 / 0 (actnorm.py:35-38)
  MIVAE_REQUIRE(!want_ld || !count_weighted || mask, "actnorm_fwd: count_weighted needs a mask");
  if (Bn > 0) {
    MIVAE_REQUIRE(z && out, "actnorm_fwd: null pointer");
    int64_t rows = Bn * S;
    RowCfg c = row_cfg(rows, D, dtype, aligned16(z) && aligned16(out));
    DISPATCH(dtype, c.vec, l_actnorm_fwd, z, log_scale, bias, mask, out, rows, (int)mask_div, (int)S,
             (int)D, inverse, c, S_(stream), want_ld ? logdet : nullptr, n_logdet, count_weighted ? mask : nullptr);
    MIVAE_CHECK_LAUNCH("actnorm_fwd");
  } else if (want_ld) {
    unsigned grid = (unsigned)((n_logdet + 3) / 4);
    k_actnorm_logdet<<<grid, kBlock, 0, S_(stream)>>>(log_scale, count_weighted ? mask : nullptr,
                                                    logdet, n_logdet, (int)mask_div, (int)S, (int)D,
                                                    inverse ? -1.f : 1.f);
    MIVAE_CHECK_LAUNCH("actnorm_logdet");
  }
  return MIVAE_OK;
}

extern "C" int64_t mivae_col_workspace_floats(int64_t rows, int64_t D, int dtype) {
  ColCfg c = col_cfg(rows, D, dtype, true);
  return (int64_t)c.nblk * 2 * D;
}

// ActNorm backward (forward direction): dz = dout*m*exp(ls); dlog_scale[d] = sum g*z*exp(ls);
// dbias[d] = sum g.  `dparams` receives [dlog_scale (D) | dbias (D)]; the log-det term
// (sum_b dld[b]*count[b]) is added by the caller on the D-vector (it is the same for every d).
extern "C" int mivae_actnorm_bwd(const void* dout, const void* z, const float* log_scale,
                                 const int64_t* mask, void* dz, float* dparams, float* workspace,
                                 int64_t Bn, int64_t mask_div, int64_t S, int64_t D, int dtype,
                                 void* stream) {
  int rc = check_latent("actnorm_bwd", Bn, S, D, dtype);
  if (rc) return rc;
  MIVAE_REQUIRE(Bn > 0, "actnorm_bwd: empty batch");
  MIVAE_REQUIRE(dout && z && log_scale && dz && dparams && workspace, "actnorm_bwd: null pointer");
  int64_t rows = Bn * S;
  ColCfg c = col_cfg(rows, D, dtype, aligned16(dout) && aligned16(z) && aligned16(dz));
  DISPATCH_COL(dtype, c.vec, 2, z, dout, log_scale, mask, dz, workspace, rows, c.rows_per_block,
               c.nblk, (int)mask_div, (int)S, (int)D, c.L, S_(stream));
  MIVAE_CHECK_LAUNCH("actnorm_bwd");
  unsigned grid = (unsigned)((2 * D + 63) / 64);
  k_col_finish<<<grid, kBlock, 0, S_(stream)>>>(workspace, dparams, c.nblk, 2, (int)D, 1.f, 0);
  MIVAE_CHECK_LAUNCH("actnorm_bwd_finish");
  return MIVAE_OK;
}

// Column mean and unbiased std over all rows (padding included), two passes (actnorm.py:23-24).
extern "C" int mivae_actnorm_stats(const void* z, float* mean, float* std, float* workspace,
                                   int64_t rows, int64_t D, int dtype, void* stream) {
  MIVAE_REQUIRE(rows >= 2 && D > 0, "actnorm_stats: need >= 2 rows (unbiased std)");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "actnorm_stats: bad dtype");
  MIVAE_REQUIRE(z && mean && std && workspace, "actnorm_stats: null pointer");
  ColCfg c = col_cfg(rows, D, dtype, aligned16(z));
  unsigned grid = (unsigned)((D + 63) / 64);
  DISPATCH_COL(dtype, c.vec, 0, z, nullptr, nullptr, nullptr, nullptr, workspace, rows,
               c.rows_per_block, c.nblk, 1, 1, (int)D, c.L, S_(stream));
  MIVAE_CHECK_LAUNCH("actnorm_stats_mean");
  k_col_finish<<<grid, kBlock, 0, S_(stream)>>>(workspace, mean, c.nblk, 1, (int)D, 1.f / (float)rows, 0);
  DISPATCH_COL(dtype, c.vec, 1, z, nullptr, mean, nullptr, nullptr, workspace, rows,
               c.rows_per_block, c.nblk, 1, 1, (int)D, c.L, S_(stream));
  MIVAE_CHECK_LAUNCH("actnorm_stats_var");
  k_col_finish<<<grid, kBlock, 0, S_(stream)>>>(workspace, std, c.nblk, 1, (int)D,
                                              1.f / (float)(rows - 1), 1);
  MIVAE_CHECK_LAUNCH("actnorm_stats_finish");
  return MIVAE_OK;
}

// out[c] = sum over rows of x[row, c] (fp32 accumulation, fixed order): the bias gradient of a Linear layer
// over many tokens (posterior heads a3, bracket projections).  workspace: mivae_col_workspace_floats.
static int colsum_impl(const void* x, float* out, float* out1, int64_t split, int accumulate, float* workspace,
                       int64_t rows, int64_t D, int dtype, void* stream) {
  MIVAE_REQUIRE(rows >= 1 && D > 0 && D < (1ll << 31), "colsum: bad shape");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "colsum: bad dtype");
  MIVAE_REQUIRE(x && out && workspace, "colsum: null pointer");
  MIVAE_REQUIRE(!out1 || (split >= 0 && split <= D), "colsum: bad split");
  if (rows <= 512) {   // short and wide: one pass, no workspace traffic
    unsigned g = (unsigned)((D + 63) / 64);
    if (dtype == MIVAE_BF16)
      k_colsum_short<unsigned short><<<g, kBlock, 0, S_(stream)>>>((const unsigned short*)x, out, (int)rows, D, out1, split, accumulate);
    else
      k_colsum_short<float><<<g, kBlock, 0, S_(stream)>>>((const float*)x, out, (int)rows, D, out1, split, accumulate);
    MIVAE_CHECK_LAUNCH("colsum_short");
    return MIVAE_OK;
  }
  ColCfg c = col_cfg(rows, D, dtype, aligned16(x));
  DISPATCH_COL(dtype, c.vec, 0, x, nullptr, nullptr, nullptr, nullptr, workspace, rows, c.rows_per_block,
               c.nblk, 1, 1, (int)D, c.L, S_(stream));
  MIVAE_CHECK_LAUNCH("colsum_partial");
  unsigned grid = (unsigned)((D + 63) / 64);
  k_col_finish<<<grid, kBlock, 0, S_(stream)>>>(workspace, out, c.nblk, 1, (int)D, 1.f, 0, out1, (int)split, accumulate);
  MIVAE_CHECK_LAUNCH("colsum_finish");
  return MIVAE_OK;
}

// out[c] = sum over rows of x[row, c] (fp32 accumulation, fixed order): the bias gradient of a Linear layer
// over many tokens (posterior heads a3, bracket projections).  workspace: mivae_col_workspace_floats.
extern "C" int mivae_colsum(const void* x, float* out, float* workspace, int64_t rows, int64_t D, int dtype,
                            void* stream) {
  return colsum_impl(x, out, nullptr, 0, 0, workspace, rows, D, dtype, stream);
}

// First stage only: per-block partial sums of the columns of x (rows, D) into workspace (count, D) fp32, count
// RETURNED (> 0; negative = error status); the caller folds them later (mivae_colsum_sink_multi, e.g. once per gradient
// bucket for every bias gradient of a training step).  workspace: mivae_col_workspace_floats.
extern "C" int64_t mivae_colsum_partial(const void* x, float* workspace, int64_t rows, int64_t D, int dtype, void* stream) {
  if (!(rows >= 1 && D > 0 && D < (1ll << 31)) || !(dtype == MIVAE_F32 || dtype == MIVAE_BF16) || !x || !workspace) {
    mivae_set_error("colsum_partial: bad arguments");
    return MIVAE_EINVAL;
  }
  ColCfg c = col_cfg(rows, D, dtype, aligned16(x));
  DISPATCH_COL(dtype, c.vec, 0, x, nullptr, nullptr, nullptr, nullptr, workspace, rows, c.rows_per_block,
               c.nblk, 1, 1, (int)D, c.L, S_(stream));
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) {
    mivae_set_error("colsum_partial: launch failed: %s", hipGetErrorString(e_));
    return MIVAE_ELAUNCH;
  }
  return c.nblk;
}

// ---- allocation-free variant of mivae_colsum ---------------------------------------------------------------------
// For callers that cannot allocate (HIP-graph capture, serving loops): the split-row partial sums go to a file-scope
// device array instead of a caller-supplied workspace.  Every GPU the library runs on holds its own copy of the array
// (4 MiB) until the process ends; its address on a device is resolved once with hipGetSymbolAddress and kept per
// device -- the array's NAME is only a host-side stand-in and is never handed to a kernel.  Calls on one GPU share the
// array: two calls that overlap in time on the same GPU (different streams) race.  The partials are fully overwritten
// before they are read, so nothing needs clearing between calls.
constexpr int64_t kStaticScratchFloats = 1 << 20;
static __device__ __align__(256) float scratch[kStaticScratchFloats];
constexpr int kMaxDevices = 64;
static float* g_scratch_ptr[kMaxDevices] = {};

extern "C" int64_t mivae_colsum_static_max_floats(void) { return kStaticScratchFloats; }

// Address of this device's copy of the array, resolved with `dev` current (restored afterwards).
static int static_scratch_resolve(int dev, float** out) {
  float* ws = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (!ws) {   // first use on this device (racing first uses resolve the same address)
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) cur = -1;
    if (cur != dev && hipSetDevice(dev) != hipSuccess) {
      mivae_set_error("colsum_static: cannot make device %d current", dev);
      return MIVAE_ELAUNCH;
    }
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (cur != dev && cur >= 0) (void)hipSetDevice(cur);
    if (e != hipSuccess || !p) {
      mivae_set_error("colsum_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      return MIVAE_ELAUNCH;
    }
    ws = static_cast<float*>(p);
    __atomic_store_n(&g_scratch_ptr[dev], ws, __ATOMIC_RELEASE);
  }
  *out = ws;
  return MIVAE_OK;
}

// Set-up call: resolves the scratch array of the CURRENT device (module load + hipGetSymbolAddress, neither of which is
// safe inside a stream capture).  Call it once per device before the first mivae_colsum_static under HIP-graph capture;
// eager callers may skip it (the first eager call resolves lazily).
extern "C" int mivae_colsum_static_init(void) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    mivae_set_error("colsum_static_init: cannot identify the current device");
    return MIVAE_ELAUNCH;
  }
  float* ws = nullptr;
  return static_scratch_resolve(dev, &ws);
}

extern "C" int mivae_colsum_static(const void* x, float* out, int64_t rows, int64_t D, int dtype, void* stream) {
  MIVAE_REQUIRE(rows >= 1 && D > 0 && D < (1ll << 31), "colsum_static: bad shape");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "colsum_static: bad dtype");
  const int64_t need = rows <= 512 ? 0 : mivae_col_workspace_floats(rows, D, dtype);
  if (need > kStaticScratchFloats) {
    mivae_set_error("colsum_static: needs %lld scratch floats, the static buffer holds %lld (rows=%lld D=%lld): use "
                    "mivae_colsum with a caller workspace", (long long)need, (long long)kStaticScratchFloats,
                    (long long)rows, (long long)D);
    return MIVAE_EUNSUPPORTED;
  }
  // the device whose copy of the array the kernels will see is the STREAM's device (the current device only for the
  // null stream); launching on another device's stream is rejected rather than handed the wrong copy
  int dev = -1, cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur < 0 || cur >= kMaxDevices) {
    mivae_set_error("colsum_static: cannot identify the current device");
    return MIVAE_ELAUNCH;
  }
  dev = cur;
  if (stream) {
    hipDevice_t sd = -1;
    if (hipStreamGetDevice(S_(stream), &sd) != hipSuccess || sd < 0 || sd >= kMaxDevices) {
      mivae_set_error("colsum_static: cannot identify the stream's device");
      return MIVAE_ELAUNCH;
    }
    dev = (int)sd;
    if (dev != cur) {
      mivae_set_error("colsum_static: the stream belongs to device %d but device %d is current", dev, cur);
      return MIVAE_EINVAL;
    }
  }
  float* ws = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (!ws) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(S_(stream), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      mivae_set_error("colsum_static: first use on device %d inside a stream capture: call mivae_colsum_static_init() "
                      "(or one eager mivae_colsum_static) on this device before capturing", dev);
      return MIVAE_EINVAL;
    }
    const int rc = static_scratch_resolve(dev, &ws);
    if (rc) return rc;
  }
  return colsum_impl(x, out, nullptr, 0, 0, ws, rows, D, dtype, stream);
}

// The same sums ADDED into two destinations: columns [0, split) += into out0, columns [split, D) += into out1
// (e.g. the [dW | db] row of mivae_linear_wgrad straight into weight.grad and bias.grad).
extern "C" int mivae_colsum_sink(const void* x, float* out0, float* out1, int64_t split, float* workspace,
                                 int64_t rows, int64_t D, int dtype, void* stream) {
  MIVAE_REQUIRE(out1, "colsum_sink: null pointer");
  return colsum_impl(x, out0, out1, split, 1, workspace, rows, D, dtype, stream);
}

// njobs x {x (rows, D) fp32 with `stride` floats between rows, out0, out1, rows <= 512, D, split, stride} as 7 x int64
// per job in DEVICE memory (stride > D: a column block of a wider partial matrix, e.g. one layer of a fused q|k|v
// weight-gradient product): for every
// job, columns [0, split) of the column sums are added into out0 and columns [split, D) into out1 (mivae_colsum_sink
// for many matrices in one launch, bit-identical to it).  max_D = the largest D in the table.
extern "C" int mivae_colsum_sink_multi(const void* jobs, int64_t njobs, int64_t max_D, void* stream) {
  MIVAE_REQUIRE(njobs >= 0 && njobs < 65536 && max_D >= 0, "colsum_sink_multi: bad sizes");
  if (njobs == 0 || max_D == 0) return MIVAE_OK;
  MIVAE_REQUIRE(jobs, "colsum_sink_multi: null pointer");
  static_assert(sizeof(ColJob) == 56, "ColJob must be 7 x 8 bytes");
  dim3 grid((unsigned)((max_D + 63) / 64), (unsigned)njobs);
  k_colsum_short_multi<<<grid, kBlock, 0, S_(stream)>>>(static_cast<const ColJob*>(jobs));
  MIVAE_CHECK_LAUNCH("colsum_sink_multi");
  return MIVAE_OK;
}

extern "C" int mivae_fold_aug_sink(const float* part, int64_t G, int64_t V, int64_t E, int64_t A, float* dW, float* db,
                                   void* stream) {
  MIVAE_REQUIRE(G >= 1 && G <= 4096 && V >= 1 && E >= 1 && A >= 1 && V * (E + A) < (1ll << 40), "fold_aug_sink: bad sizes");
  MIVAE_REQUIRE(part && dW && db, "fold_aug_sink: null pointer");
  const int64_t n = V * (E + A);
  k_fold_aug<<<(unsigned)((n + 255) / 256), 256, 0, S_(stream)>>>(part, (int)G, V, (int)E, (int)A, dW, db);
  MIVAE_CHECK_LAUNCH("fold_aug_sink");
  return MIVAE_OK;
}

// dW (V, E) fp32 (+)= scatter-add of dy (N, E) fp32 rows by ids (N) int64, rows of `padding_idx` (< 0: none) skipped:
// the weight gradient of torch.nn.Embedding (Embedder of VAELmModel.encode, models/base/model.py:213-222), run-to-run
// deterministic.  accumulate != 0 adds into dW (a gradient sink), else every row of dW is written.  E <= 512.
extern "C" int mivae_embedding_bwd(const float* dy, const int64_t* ids, float* dW, int64_t N, int64_t V, int64_t E,
                                   int64_t padding_idx, int accumulate, void* stream) {
  MIVAE_REQUIRE(N >= 0 && N < (1ll << 26) && V >= 1 && V < (1ll << 31), "embedding_bwd: bad sizes");
  if (E < 1 || E > 512) {
    mivae_set_error("embedding_bwd: unsupported width E=%lld (1..512)", (long long)E);
    return MIVAE_EUNSUPPORTED;
  }
  MIVAE_REQUIRE(dW && (N == 0 || (dy && ids)), "embedding_bwd: null pointer");
  EmbArgs a = {dy, ids, dW, N, (int)V, (int)E, (int)padding_idx, accumulate};
  const int rows = E <= 128 ? 32 : (E <= 256 ? 16 : 8);
  const size_t lds = (size_t)4 * rows * E * sizeof(float) + (size_t)4 * kEmbCap * sizeof(int);
  const unsigned grid = (unsigned)((V + rows - 1) / rows);
#define EMB_LAUNCH(R, H)                                                                                            \
  do {                                                                                                              \
    if (lds > 64 * 1024)                                                                                            \
      (void)hipFuncSetAttribute((const void*)k_embed_bwd<R, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    k_embed_bwd<R, H><<<grid, kBlock, lds, S_(stream)>>>(a);                                                        \
  } while (0)
  if (E <= 128) EMB_LAUNCH(32, 2);
  else if (E <= 256) EMB_LAUNCH(16, 4);
  else EMB_LAUNCH(8, 8);
#undef EMB_LAUNCH
  MIVAE_CHECK_LAUNCH("embedding_bwd");
  return MIVAE_OK;
}

extern "C" int mivae_kl_fwd(const float* logq, const float* logp, float* kl, int64_t n, int use_fb,
                            float free_bits, void* stream) {
  MIVAE_REQUIRE(n >= 0, "kl_fwd: n < 0");
  if (n == 0) return MIVAE_OK;
  MIVAE_REQUIRE(logq && logp && kl, "kl_fwd: null pointer");
  k_kl_fwd<<<(unsigned)((n + kBlock - 1) / kBlock), kBlock, 0, S_(stream)>>>(logq, logp, kl, n, use_fb, free_bits);
  MIVAE_CHECK_LAUNCH("kl_fwd");
  return MIVAE_OK;
}

extern "C" int mivae_kl_bwd(const float* logq, const float* logp, const float* dkl, float* dq,
                            float* dp, int64_t n, int use_fb, float free_bits, void* stream) {
  MIVAE_REQUIRE(n >= 0, "kl_bwd: n < 0");
  if (n == 0) return MIVAE_OK;
  MIVAE_REQUIRE(logq && logp && dkl && dq && dp, "kl_bwd: null pointer");
  k_kl_bwd<<<(unsigned)((n + kBlock - 1) / kBlock), kBlock, 0, S_(stream)>>>(logq, logp, dkl, dq, dp, n, use_fb, free_bits);
  MIVAE_CHECK_LAUNCH("kl_bwd");
  return MIVAE_OK;
}

extern "C" int mivae_elbo_fwd(const float* recon, const float* kl, float* out3, int64_t B, int64_t n,
                              float recon_weight, float kl_weight, int iwae, const float* weights_dev,
                              void* stream) {
  MIVAE_REQUIRE(B > 0 && n > 0 && B * n < (1ll << 30), "elbo_fwd: bad shape B=%ld n=%ld", (long)B, (long)n);
  MIVAE_REQUIRE(recon && kl && out3, "elbo_fwd: null pointer");
  k_elbo_fwd<<<1, kBlock, 0, S_(stream)>>>(recon, kl, out3, (int)B, (int)n, recon_weight, kl_weight, iwae, weights_dev);
  MIVAE_CHECK_LAUNCH("elbo_fwd");
  return MIVAE_OK;
}

extern "C" int mivae_elbo_bwd(const float* recon, const float* kl, const float* gout3, float* drecon,
                              float* dkl, int64_t B, int64_t n, float recon_weight, float kl_weight,
                              int iwae, const float* weights_dev, void* stream) {
  MIVAE_REQUIRE(B > 0 && n > 0 && B * n < (1ll << 30), "elbo_bwd: bad shape");
  MIVAE_REQUIRE(recon && kl && gout3 && drecon && dkl, "elbo_bwd: null pointer");
  k_elbo_bwd<<<(unsigned)((B + kBlock - 1) / kBlock), kBlock, 0, S_(stream)>>>(recon, kl, gout3, drecon, dkl, (int)B, (int)n, recon_weight, kl_weight, iwae, weights_dev);
  MIVAE_CHECK_LAUNCH("elbo_bwd");
  return MIVAE_OK;
}
