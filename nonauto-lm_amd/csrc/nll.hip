// Reconstruction term of the ELBO: label-smoothing NLL over the vocabulary.
//
// Replaces LabelSmoothingNLL.forward (vae_lm/models/base/losses.py:128-169): log_softmax over V,
// gather (or smoothed sum), x weights, per-sequence weighted token mean.  The eager chain reads
// and writes the (N, V) logits several times (fp32 upcast, log_softmax, one-hot, mul, sum); here
// the forward reads the logits ONCE (online max / sum-exp, 16 B per lane), the backward reads them
// once and writes d logits once.  HBM roofline: forward N*V*e bytes, backward 2*N*V*e.
//
// One wave per token row: a 64-lane wave covers 1 KiB of a row per load instruction, rows whose
// weight is zero (padding) are not read at all (their loss contribution is exactly 0 and their
// gradient row is written as zeros).  Reductions are fixed-order (lane-local, then xor butterfly,
// then one wave per sentence walking its tokens in order): bit-reproducible run to run.
#include "common.hpp"

using namespace mivae;

namespace {

constexpr int kBlock = 256;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float wave_max(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

template <typename T>
__device__ __forceinline__ float load1(const T* p);
template <>
__device__ __forceinline__ float load1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load1<unsigned short>(const unsigned short* p) { return bf16_to_f32(*p); }

// tok_nll[r] = w_r * nll_r, lse[r] = logsumexp(logits[r, :]).
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_nll_fwd(const T* __restrict__ logits,
                                                     const int64_t* __restrict__ target,
                                                     const float* __restrict__ weights,
                                                     float* __restrict__ tok_nll, float* __restrict__ lse_out,
                                                     int64_t N, int V, float smoothing) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (r >= N) return;
  const float w = weights[r];
  if (w == 0.0f) {                      // padding token: contributes exactly 0, row is not read
    if (lane == 0) { tok_nll[r] = 0.0f; lse_out[r] = 0.0f; }
    return;
  }
  const T* row = logits + r * (int64_t)V;
  // online softmax in the base-2 domain: m = running max of x*log2e, s = sum 2^(x*log2e - m)
  float m = -INFINITY, s = 0.0f, sx = 0.0f;
  const int chunks = V / VEC;
#pragma unroll 4
  for (int c = lane; c < chunks; c += 64) {
    float v[VEC];
    RowIO<T, VEC>::load(row + (int64_t)c * VEC, v);
    float cm = v[0];
#pragma unroll
    for (int i = 1; i < VEC; ++i) cm = fmaxf(cm, v[i]);
    const float mn = fmaxf(m, cm * kLog2e);
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      acc += ex2(fmaf(v[i], kLog2e, -mn));
      sx += v[i];
    }
    s = fmaf(s, ex2(m - mn), acc);
    m = mn;
  }
  const float M = wave_max(m);
  s = (m == -INFINITY) ? 0.0f : s * ex2(m - M);
  s = wave_sum(s);
  sx = wave_sum(sx);
  if (lane == 0) {
    const float lse = (M + __builtin_amdgcn_logf(s)) * kLn2;   // v_log_f32 is log2
    const int64_t t = target[r];
    const float xt = (t >= 0 && t < V) ? load1<T>(row + t) : NAN;
    float nll = lse - xt;
    if (smoothing > 0.0f) nll = (1.0f - smoothing) * nll + smoothing * lse - (smoothing / (float)V) * sx;
    tok_nll[r] = nll * w;
    lse_out[r] = lse;
  }
}

// per_seq[b] = sum_s tok_nll[b,s] / clamp(sum_s w[b,s], 1e-13);  tokscale[b,s] = w[b,s] / clamp(...)
__global__ __launch_bounds__(kBlock) void k_nll_seq(const float* __restrict__ tok_nll,
                                                     const float* __restrict__ weights,
                                                     float* __restrict__ per_seq, float* __restrict__ tokscale,
                                                     int64_t Bn, int S) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (b >= Bn) return;
  float a = 0.0f, ws = 0.0f;
  for (int s = lane; s < S; s += 64) {
    a += tok_nll[b * S + s];
    ws += weights[b * S + s];
  }
  a = wave_sum(a);
  ws = wave_sum(ws);
  const float inv = 1.0f / fmaxf(ws, 1e-13f);
  if (lane == 0) per_seq[b] = a * inv;
  for (int s = lane; s < S; s += 64) tokscale[b * S + s] = weights[b * S + s] * inv;
}

// dlogits[r, v] = dper[b] * tokscale[r] * (softmax(logits[r])_v - q_v),  q = (1-sm) onehot(t) + sm/V
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_nll_bwd(const T* __restrict__ logits,
                                                     const int64_t* __restrict__ target,
                                                     const float* __restrict__ lse,
                                                     const float* __restrict__ tokscale,
                                                     const float* __restrict__ dper, T* __restrict__ dlogits,
                                                     int64_t N, int S, int V, float smoothing) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (r >= N) return;
  const float coef = dper[r / S] * tokscale[r];
  const T* row = logits + r * (int64_t)V;
  T* drow = dlogits + r * (int64_t)V;
  const int chunks = V / VEC;
  if (coef == 0.0f) {
    float z[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) z[i] = 0.0f;
    for (int c = lane; c < chunks; c += 64) RowIO<T, VEC>::store(drow + (int64_t)c * VEC, z);
    return;
  }
  const float l2 = lse[r] * kLog2e;
  const int t = (int)target[r];
  const float uni = coef * (smoothing > 0.0f ? smoothing / (float)V : 0.0f);
  const float hot = coef * (smoothing > 0.0f ? 1.0f - smoothing : 1.0f);
#pragma unroll 4
  for (int c = lane; c < chunks; c += 64) {
    float v[VEC];
    RowIO<T, VEC>::load(row + (int64_t)c * VEC, v);
    const int base = c * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float g = fmaf(coef, ex2(fmaf(v[i], kLog2e, -l2)), -uni);
      v[i] = (base + i == t) ? g - hot : g;
    }
    RowIO<T, VEC>::store(drow + (int64_t)c * VEC, v);
  }
}

template <typename T, int VEC>
void launch_fwd(const void* logits, const int64_t* target, const float* weights, float* tok_nll, float* lse,
                int64_t N, int V, float sm, hipStream_t st) {
  unsigned grid = (unsigned)((N + kBlock / 64 - 1) / (kBlock / 64));
  k_nll_fwd<T, VEC><<<grid, kBlock, 0, st>>>(static_cast<const T*>(logits), target, weights, tok_nll, lse, N, V, sm);
}
template <typename T, int VEC>
void launch_bwd(const void* logits, const int64_t* target, const float* lse, const float* tokscale,
                const float* dper, void* dlogits, int64_t N, int S, int V, float sm, hipStream_t st) {
  unsigned grid = (unsigned)((N + kBlock / 64 - 1) / (kBlock / 64));
  k_nll_bwd<T, VEC><<<grid, kBlock, 0, st>>>(static_cast<const T*>(logits), target, lse, tokscale, dper,
                                             static_cast<T*>(dlogits), N, S, V, sm);
}

int check(const char* name, int64_t Bn, int64_t S, int64_t V, float smoothing, int dtype) {
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "%s: dtype must be 0 (f32) or 1 (bf16)", name);
  MIVAE_REQUIRE(Bn >= 0 && S > 0 && V > 0 && S < (1ll << 31) && V < (1ll << 31) && Bn * S < (1ll << 33),
                "%s: bad shape Bn=%lld S=%lld V=%lld", name, (long long)Bn, (long long)S, (long long)V);
  MIVAE_REQUIRE(smoothing >= 0.0f && smoothing < 1.0f, "%s: smoothing must be in [0, 1)", name);
  return MIVAE_OK;
}

}  // namespace

extern "C" int mivae_nll_fwd(const void* logits, const int64_t* target, const float* weights, float* per_seq,
                             float* lse, float* tokscale, float* tok_nll, int64_t Bn, int64_t S, int64_t V,
                             float smoothing, int dtype, void* stream) {
  int rc = check("nll_fwd", Bn, S, V, smoothing, dtype);
  if (rc) return rc;
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(logits && target && weights && per_seq && lse && tokscale && tok_nll, "nll_fwd: null pointer");
  const int64_t N = Bn * S;
  const bool al = aligned16(logits);
  if (dtype == MIVAE_BF16) {
    if (al && V % 8 == 0) launch_fwd<unsigned short, 8>(logits, target, weights, tok_nll, lse, N, (int)V, smoothing, S_(stream));
    else launch_fwd<unsigned short, 1>(logits, target, weights, tok_nll, lse, N, (int)V, smoothing, S_(stream));
  } else {
    if (al && V % 4 == 0) launch_fwd<float, 4>(logits, target, weights, tok_nll, lse, N, (int)V, smoothing, S_(stream));
    else launch_fwd<float, 1>(logits, target, weights, tok_nll, lse, N, (int)V, smoothing, S_(stream));
  }
  MIVAE_CHECK_LAUNCH("nll_fwd");
  k_nll_seq<<<(unsigned)((Bn + 3) / 4), kBlock, 0, S_(stream)>>>(tok_nll, weights, per_seq, tokscale, Bn, (int)S);
  MIVAE_CHECK_LAUNCH("nll_seq");
  return MIVAE_OK;
}

extern "C" int mivae_nll_bwd(const void* logits, const int64_t* target, const float* lse, const float* tokscale,
                             const float* dper_seq, void* dlogits, int64_t Bn, int64_t S, int64_t V,
                             float smoothing, int dtype, void* stream) {
  int rc = check("nll_bwd", Bn, S, V, smoothing, dtype);
  if (rc) return rc;
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(logits && target && lse && tokscale && dper_seq && dlogits, "nll_bwd: null pointer");
  const int64_t N = Bn * S;
  const bool al = aligned16(logits) && aligned16(dlogits);
  if (dtype == MIVAE_BF16) {
    if (al && V % 8 == 0) launch_bwd<unsigned short, 8>(logits, target, lse, tokscale, dper_seq, dlogits, N, (int)S, (int)V, smoothing, S_(stream));
    else launch_bwd<unsigned short, 1>(logits, target, lse, tokscale, dper_seq, dlogits, N, (int)S, (int)V, smoothing, S_(stream));
  } else {
    if (al && V % 4 == 0) launch_bwd<float, 4>(logits, target, lse, tokscale, dper_seq, dlogits, N, (int)S, (int)V, smoothing, S_(stream));
    else launch_bwd<float, 1>(logits, target, lse, tokscale, dper_seq, dlogits, N, (int)S, (int)V, smoothing, S_(stream));
  }
  MIVAE_CHECK_LAUNCH("nll_bwd");
  return MIVAE_OK;
}
