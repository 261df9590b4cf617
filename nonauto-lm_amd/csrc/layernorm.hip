// LayerNorm over the feature axis for the pre-norm sublayers (vae_lm/nn/transformer.py:147-162,
// SublayerConnection._norm = torch.nn.LayerNorm) used by the transformer bracket and by the attention coupling
// (couplings.py:54-81).  HBM-bound row kernels: the forward reads x once and writes y once
// (y may be bf16 while x is the fp32 residual stream, which removes the separate cast autocast inserts before the
// next projection); the backward reads dy and x once, writes dx once, and reduces d gamma / d beta in fixed order
// (per-block partial column sums, then a second pass): no atomics.
#include "common.hpp"

using namespace mivae;

namespace {

constexpr int kBlock = 256;
constexpr int kMaxChunks = 4;   // D <= 64 lanes * VEC * kMaxChunks

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <typename TI, typename TO, int VEC>
__global__ __launch_bounds__(kBlock) void k_ln_fwd(const TI* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, TO* __restrict__ y,
                                                    float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                    int64_t rows, int D, int L, float eps) {
  const int rif = kBlock / L;
  const int64_t row = (int64_t)blockIdx.x * rif + threadIdx.x / L;
  const int lig = threadIdx.x % L;
  const bool ok = row < rows;
  const int64_t r = ok ? row : 0;
  float v[kMaxChunks][VEC];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxChunks; ++j) {
    const int c = (lig + j * L) * VEC;
    if (c < D) {
      RowIO<TI, VEC>::load(x + r * D + c, v[j]);
#pragma unroll
      for (int i = 0; i < VEC; ++i) s += v[j][i];
    }
  }
  const float mean = group_sum(s, L) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxChunks; ++j) {
    const int c = (lig + j * L) * VEC;
    if (c < D) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { const float d = v[j][i] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(group_sum(q, L) / (float)D + eps);
  if (!ok) return;
  if (lig == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
  for (int j = 0; j < kMaxChunks; ++j) {
    const int c = (lig + j * L) * VEC;
    if (c < D) {
      float o[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = fmaf((v[j][i] - mean) * rstd, gamma[c + i], beta[c + i]);
      RowIO<TO, VEC>::store(y + row * D + c, o);
    }
  }
}

// dx = rstd * (a - mean_D(a) - xhat * mean_D(a * xhat)), a = dy * gamma; partial[blk][0][c] = sum dy * xhat,
// partial[blk][1][c] = sum dy over the block's rows (row groups combined in LDS in fixed order).
template <typename TI, typename TO, int VEC>
__global__ __launch_bounds__(kBlock) void k_ln_bwd(const TO* __restrict__ dy, const TI* __restrict__ x,
                                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, TI* __restrict__ dx,
                                                    float* __restrict__ partial, int64_t rows, int rows_per_block,
                                                    int D, int L, const TI* __restrict__ add,
                                                    unsigned short* __restrict__ dx16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [rif][2][L * VEC * kMaxChunks]
  const int rif = kBlock / L;
  const int rg = threadIdx.x / L, lig = threadIdx.x % L;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float gm[kMaxChunks][VEC], dg[kMaxChunks][VEC], db[kMaxChunks][VEC];
#pragma unroll
  for (int j = 0; j < kMaxChunks; ++j) {
    const int c = (lig + j * L) * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      gm[j][i] = c < D ? gamma[c + i] : 0.f;
      dg[j][i] = 0.f;
      db[j][i] = 0.f;
    }
  }
  const int64_t iters = (rows_per_block + rif - 1) / rif;
  for (int64_t it = 0; it < iters; ++it) {         // every lane of a row group runs every iteration (shuffles)
    const int64_t row = r0 + rg + it * rif;
    const bool ok = row < r1;
    const int64_t r = ok ? row : r0;
    const float mu = mean[r], rs = rstd[r];
    float xh[kMaxChunks][VEC], a[kMaxChunks][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxChunks; ++j) {
      const int c = (lig + j * L) * VEC;
      if (c < D) {
        float xv[VEC], gv[VEC];
        RowIO<TI, VEC>::load(x + r * D + c, xv);
        RowIO<TO, VEC>::load(dy + r * D + c, gv);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          xh[j][i] = (xv[i] - mu) * rs;
          a[j][i] = gv[i] * gm[j][i];
          s1 += a[j][i];
          s2 += a[j][i] * xh[j][i];
          if (ok) { dg[j][i] += gv[i] * xh[j][i]; db[j][i] += gv[i]; }
        }
      }
    }
    const float c1 = group_sum(s1, L) / (float)D, c2 = group_sum(s2, L) / (float)D;
    if (ok) {
#pragma unroll
      for (int j = 0; j < kMaxChunks; ++j) {
        const int c = (lig + j * L) * VEC;
        if (c < D) {
          float o[VEC];
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[i] = rs * (a[j][i] - c1 - xh[j][i] * c2);
          if (add) {   // residual branch of a pre-norm sublayer: d x = d(residual) + LN backward, one store
            float r[VEC];
            RowIO<TI, VEC>::load(add + row * D + c, r);
            if (sizeof(TI) == 2) {   // the stock path rounds the LN gradient to bf16 before the add
#pragma unroll
              for (int i = 0; i < VEC; ++i) o[i] = bf16_to_f32(f32_to_bf16(o[i]));
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) o[i] += r[i];
          }
          RowIO<TI, VEC>::store(dx + row * D + c, o);
          if (dx16) RowIO<unsigned short, VEC>::store(dx16 + row * D + c, o);   // bf16 twin for the next projection's backward
        }
      }
    }
  }
  const int W = L * VEC * kMaxChunks;               // padded row width in the LDS image
#pragma unroll
  for (int j = 0; j < kMaxChunks; ++j)
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int cc = (lig + j * L) * VEC + i;
      smem[(rg * 2 + 0) * W + cc] = dg[j][i];
      smem[(rg * 2 + 1) * W + cc] = db[j][i];
    }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * D; idx += kBlock) {
    const int v = idx / D, c = idx % D;
    float sum = 0.f;
    for (int g = 0; g < rif; ++g) sum += smem[(g * 2 + v) * W + c];
    partial[((int64_t)blockIdx.x * 2 + v) * D + c] = sum;
  }
}

// out[idx] = sum_b partial[b][idx], idx < total: 4 waves split the blocks, combined in wave order
__global__ void __launch_bounds__(256) k_ln_finish(const float* __restrict__ partial, float* __restrict__ out,
                                                   int nblk, int total, float* __restrict__ out1, int split,
                                                   int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;
  float sum = 0.f;
  if (idx < total) {
#pragma unroll 16
    for (int b = grp; b < nblk; b += 4) sum += partial[(int64_t)b * total + idx];
  }
  red[grp][lane] = sum;
  __syncthreads();
  if (grp == 0 && idx < total) {
    const float v = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    float* dst = (out1 && idx >= split) ? out1 + (idx - split) : out + idx;
    *dst = accumulate ? *dst + v : v;
  }
}

struct LnCfg { int vec, L, nblk, rows_per_block; };
static bool ln_cfg(int64_t rows, int64_t D, bool vec4, LnCfg& c) {
  c.vec = vec4 ? 4 : 1;
  c.L = lanes_per_row(D, c.vec);
  if ((int64_t)c.L * c.vec * kMaxChunks < D) return false;
  const int rif = kBlock / c.L;
  int64_t nblk = (rows + rif - 1) / rif;           // one pass of row groups per block while that fills <= 1024 blocks
  if (nblk > 1024) nblk = 1024;
  if (nblk < 1) nblk = 1;
  int64_t rpb = (rows + nblk - 1) / nblk;
  rpb = (rpb + rif - 1) / rif * rif;
  c.rows_per_block = (int)rpb;
  c.nblk = (int)((rows + rpb - 1) / rpb);
  return true;
}

template <typename TI, typename TO>
void launch_fwd(int vec, const void* x, const float* g, const float* b, void* y, float* mean, float* rstd,
                int64_t rows, int D, int L, float eps, hipStream_t st) {
  const int rif = kBlock / L;
  const unsigned grid = (unsigned)((rows + rif - 1) / rif);
  if (vec == 4) k_ln_fwd<TI, TO, 4><<<grid, kBlock, 0, st>>>((const TI*)x, g, b, (TO*)y, mean, rstd, rows, D, L, eps);
  else k_ln_fwd<TI, TO, 1><<<grid, kBlock, 0, st>>>((const TI*)x, g, b, (TO*)y, mean, rstd, rows, D, L, eps);
}
template <typename TI, typename TO>
void launch_bwd(const LnCfg& c, const void* dy, const void* x, const float* g, const float* mean, const float* rstd,
                void* dx, float* partial, int64_t rows, int D, hipStream_t st, const void* add, void* dx16) {
  const size_t smem = (size_t)(kBlock / c.L) * 2 * c.L * c.vec * kMaxChunks * sizeof(float);
  if (c.vec == 4)
    k_ln_bwd<TI, TO, 4><<<c.nblk, kBlock, smem, st>>>((const TO*)dy, (const TI*)x, g, mean, rstd, (TI*)dx, partial,
                                                      rows, c.rows_per_block, D, c.L, (const TI*)add, (unsigned short*)dx16);
  else
    k_ln_bwd<TI, TO, 1><<<c.nblk, kBlock, smem, st>>>((const TO*)dy, (const TI*)x, g, mean, rstd, (TI*)dx, partial,
                                                      rows, c.rows_per_block, D, c.L, (const TI*)add, (unsigned short*)dx16);
}

}  // namespace

extern "C" int64_t mivae_layernorm_workspace_floats(int64_t rows, int64_t D) {
  LnCfg c;
  if (rows < 1 || D < 1) return 0;
  int64_t nblk = 0;                    // the vectorised and the scalar geometry may split the rows differently
  if (D % 4 == 0 && ln_cfg(rows, D, true, c)) nblk = c.nblk;
  if (ln_cfg(rows, D, false, c) && c.nblk > nblk) nblk = c.nblk;
  return nblk * 2 * D;
}

extern "C" int mivae_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                   float* rstd, int64_t rows, int64_t D, float eps, int in_dtype, int out_dtype,
                                   void* stream) {
  MIVAE_REQUIRE(rows >= 0 && D >= 1, "layernorm_fwd: bad shape");
  MIVAE_REQUIRE((in_dtype == MIVAE_F32 || in_dtype == MIVAE_BF16) && (out_dtype == MIVAE_F32 || out_dtype == MIVAE_BF16),
                "layernorm_fwd: dtype must be 0 (f32) or 1 (bf16)");
  if (rows == 0) return MIVAE_OK;
  MIVAE_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
  LnCfg c;
  MIVAE_REQUIRE(ln_cfg(rows, D, D % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta), c),
                "layernorm_fwd: D = %lld too wide for this kernel", (long long)D);
  hipStream_t st = S_(stream);
  if (in_dtype == MIVAE_F32 && out_dtype == MIVAE_F32) launch_fwd<float, float>(c.vec, x, gamma, beta, y, mean, rstd, rows, (int)D, c.L, eps, st);
  else if (in_dtype == MIVAE_F32) launch_fwd<float, unsigned short>(c.vec, x, gamma, beta, y, mean, rstd, rows, (int)D, c.L, eps, st);
  else if (out_dtype == MIVAE_F32) launch_fwd<unsigned short, float>(c.vec, x, gamma, beta, y, mean, rstd, rows, (int)D, c.L, eps, st);
  else launch_fwd<unsigned short, unsigned short>(c.vec, x, gamma, beta, y, mean, rstd, rows, (int)D, c.L, eps, st);
  MIVAE_CHECK_LAUNCH("layernorm_fwd");
  return MIVAE_OK;
}

static int layernorm_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean,
                              const float* rstd, void* dx, float* dgamma_dbeta, float* dbeta_sink, float* workspace,
                              int64_t rows, int64_t D, int in_dtype, int out_dtype, void* stream,
                              int* nblk_out = nullptr, const void* add = nullptr, void* dx16 = nullptr) {
  MIVAE_REQUIRE(rows >= 1 && D >= 1, "layernorm_bwd: bad shape");
  MIVAE_REQUIRE((in_dtype == MIVAE_F32 || in_dtype == MIVAE_BF16) && (out_dtype == MIVAE_F32 || out_dtype == MIVAE_BF16),
                "layernorm_bwd: dtype must be 0 (f32) or 1 (bf16)");
  MIVAE_REQUIRE(dy && x && gamma && mean && rstd && dx && (dgamma_dbeta || nblk_out) && workspace,
                "layernorm_bwd: null pointer");
  LnCfg c;
  MIVAE_REQUIRE(ln_cfg(rows, D, D % 4 == 0 && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(add), c),
                "layernorm_bwd: D = %lld too wide for this kernel", (long long)D);
  hipStream_t st = S_(stream);
  if (in_dtype == MIVAE_F32 && out_dtype == MIVAE_F32) launch_bwd<float, float>(c, dy, x, gamma, mean, rstd, dx, workspace, rows, (int)D, st, add, dx16);
  else if (in_dtype == MIVAE_F32) launch_bwd<float, unsigned short>(c, dy, x, gamma, mean, rstd, dx, workspace, rows, (int)D, st, add, dx16);
  else if (out_dtype == MIVAE_F32) launch_bwd<unsigned short, float>(c, dy, x, gamma, mean, rstd, dx, workspace, rows, (int)D, st, add, dx16);
  else launch_bwd<unsigned short, unsigned short>(c, dy, x, gamma, mean, rstd, dx, workspace, rows, (int)D, st, add, dx16);
  MIVAE_CHECK_LAUNCH("layernorm_bwd");
  if (nblk_out) { *nblk_out = c.nblk; return MIVAE_OK; }    // partials only: the caller folds them
  const int total = 2 * (int)D;
  k_ln_finish<<<(unsigned)((total + 63) / 64), 256, 0, st>>>(workspace, dgamma_dbeta, c.nblk, total, dbeta_sink, (int)D,
                                                              dbeta_sink ? 1 : 0);
  MIVAE_CHECK_LAUNCH("layernorm_bwd_finish");
  return MIVAE_OK;
}

extern "C" int mivae_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                   const float* rstd, void* dx, float* dgamma_dbeta, float* workspace, int64_t rows,
                                   int64_t D, int in_dtype, int out_dtype, void* stream) {
  return layernorm_bwd_impl(dy, x, gamma, mean, rstd, dx, dgamma_dbeta, nullptr, workspace, rows, D, in_dtype, out_dtype, stream);
}

// Same, with d gamma ADDED into dgamma (D) and d beta ADDED into dbeta (D): the parameters' gradient buffers.
extern "C" int mivae_layernorm_bwd_sink(const void* dy, const void* x, const float* gamma, const float* mean,
                                        const float* rstd, void* dx, float* dgamma, float* dbeta, float* workspace,
                                        int64_t rows, int64_t D, int in_dtype, int out_dtype, void* stream) {
  MIVAE_REQUIRE(dbeta, "layernorm_bwd_sink: null pointer");
  return layernorm_bwd_impl(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, D, in_dtype, out_dtype, stream);
}

// dx as mivae_layernorm_bwd, but the parameter gradients are left as per-block partials in workspace:
// workspace[b][0][c] = sum dy * xhat, workspace[b][1][c] = sum dy over block b's rows, b < the RETURNED count (> 0;
// negative = error status).  Column sums of that (count, 2 D) matrix are [d gamma | d beta]: fold them with
// mivae_colsum_sink / mivae_colsum_sink_multi (split = D), e.g. once per step for all LayerNorms.
extern "C" int64_t mivae_layernorm_bwd_partial(const void* dy, const void* x, const float* gamma, const float* mean,
                                               const float* rstd, void* dx, float* workspace, int64_t rows, int64_t D,
                                               int in_dtype, int out_dtype, void* stream) {
  int nblk = 0;
  const int rc = layernorm_bwd_impl(dy, x, gamma, mean, rstd, dx, nullptr, nullptr, workspace, rows, D, in_dtype,
                                    out_dtype, stream, &nblk);
  return rc == MIVAE_OK ? (int64_t)nblk : (int64_t)rc;
}

// mivae_layernorm_bwd_partial with the residual branch folded in: dx = add + (LayerNorm input gradient), `add` (rows, D)
// in x's dtype -- the gradient of y = x + f(LN(x)) with respect to x in one store (pre-norm sublayers).
extern "C" int64_t mivae_layernorm_bwd_partial_add(const void* dy, const void* x, const float* gamma, const float* mean,
                                                   const float* rstd, const void* add, void* dx, float* workspace,
                                                   int64_t rows, int64_t D, int in_dtype, int out_dtype, void* stream) {
  if (!add) { mivae_set_error("layernorm_bwd_partial_add: null residual gradient"); return MIVAE_EINVAL; }
  int nblk = 0;
  const int rc = layernorm_bwd_impl(dy, x, gamma, mean, rstd, dx, nullptr, nullptr, workspace, rows, D, in_dtype,
                                    out_dtype, stream, &nblk, add);
  return rc == MIVAE_OK ? (int64_t)nblk : (int64_t)rc;
}

// mivae_layernorm_bwd_partial_add that also writes dx rounded to bf16 into dx16 (rows, D): the fp32 residual stream's
// gradient feeds the previous sublayer's projections, whose backward works on bf16 (one cast launch less per sublayer).
extern "C" int64_t mivae_layernorm_bwd_partial_add_twin(const void* dy, const void* x, const float* gamma,
                                                        const float* mean, const float* rstd, const void* add, void* dx,
                                                        void* dx16, float* workspace, int64_t rows, int64_t D,
                                                        int in_dtype, int out_dtype, void* stream) {
  if (!add || !dx16) { mivae_set_error("layernorm_bwd_partial_add_twin: null pointer"); return MIVAE_EINVAL; }
  if (D % 4 == 0 && (reinterpret_cast<uintptr_t>(dx16) & 7)) {
    mivae_set_error("layernorm_bwd_partial_add_twin: dx16 must be 8-byte aligned");
    return MIVAE_EINVAL;
  }
  int nblk = 0;
  const int rc = layernorm_bwd_impl(dy, x, gamma, mean, rstd, dx, nullptr, nullptr, workspace, rows, D, in_dtype,
                                    out_dtype, stream, &nblk, add, dx16);
  return rc == MIVAE_OK ? (int64_t)nblk : (int64_t)rc;
}
