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
#include <stdlib.h>

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

// =====================================================================================================
// Fused vocabulary projection + NLL ("lce"): logits = h W^T + b are produced tile by tile on the
// matrix cores and consumed in registers -- the (N, V) logits are never written to HBM.
//   NonAutoModel.decode: logits = self._vocab_projection(decoded)      (nonauto_lm/model.py:105-112)
//   LabelSmoothingNLL.forward on those logits                           (models/base/losses.py:128-169)
// bf16 operands, fp32 accumulation (v_mfma_f32_32x32x16_bf16): products of bf16 values are exact in fp32, so the
// logits equal an fp32 GEMM of the same rounded operands up to summation order.
//   forward : a wave owns 32 tokens x a slice of the vocabulary; per 32-row vocabulary chunk one accumulator
//             tile (rows = vocabulary in registers, columns = tokens on lanes), online max / sum-exp per lane,
//             target logit picked by row compare; partial (max, sum, sum x, x_t) per token and slice, combined
//             by k_lce_combine.  VALU (v_exp_f32) bound: N V exponentials.
//   backward: recomputes the tile, forms coef (softmax - q) and writes d logits (bf16) through a per-wave LDS
//             transpose as 128-byte row segments; the two dense products with d logits (d h, d W) stay library
//             GEMMs.  HBM: N V e written, nothing of that size read.
namespace {

typedef __bf16 lce_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int lce_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// NF = E / 16 + 1 fragments per chunk: the last k-step carries the bias (hi + lo bf16 halves) against a
// [1, 1, 0, ...] column of h, so the bias add rides on the matrix pipe and the accumulator starts from zero.
template <int NF>
__device__ __forceinline__ void lce_load_w(const unsigned short* __restrict__ W, int c, int r32, int hh,
                                           lce_bf16x8 (&f)[NF]) {
  // W is packed in fragment order [chunk][k-step][lane][8]: one contiguous 1 KiB per wave load
  const unsigned short* p = W + ((int64_t)c * NF * 64 + (hh * 32 + r32)) * 8;
#pragma unroll
  for (int ks = 0; ks < NF; ++ks) f[ks] = *reinterpret_cast<const lce_bf16x8*>(p + ks * 512);
}

template <int NF>
__device__ __forceinline__ f32x16 lce_tile(const lce_bf16x8 (&wf)[NF], const lce_bf16x8 (&hf)[NF]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < NF; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], hf[ks], acc, 0, 0, 0);
  return acc;
}

// h fragments of one 32-token tile (+ the [1, 1, 0, ...] bias column)
template <int NF>
__device__ __forceinline__ void lce_load_h(const unsigned short* __restrict__ h, int64_t tok, bool valid, int hh,
                                           lce_bf16x8 (&hf)[NF]) {
  const unsigned short* hp = h + (valid ? tok : 0) * (16 * (NF - 1)) + 8 * hh;
#pragma unroll
  for (int ks = 0; ks < NF - 1; ++ks) hf[ks] = *reinterpret_cast<const lce_bf16x8*>(hp + 16 * ks);
  const u16x8 one = {0x3F80, 0x3F80, 0, 0, 0, 0, 0, 0}, zero = {0, 0, 0, 0, 0, 0, 0, 0};
  hf[NF - 1] = __builtin_bit_cast(lce_bf16x8, hh == 0 ? one : zero);
}

struct LceArgs {
  const unsigned short* h;      // (N, E) bf16
  const unsigned short* W;      // (V, E + 16) bf16 in fragment order incl. the bias k-step (k_lce_pack)
  const int64_t* target;        // (N)
  const float* weights;         // (N)           [fwd]
  f32x4* partial;               // (VS, N) {max2, sum, sum x, x_t}   [fwd]
  const float* lse;             // (N)           [bwd]
  const float* tokscale;        // (N)           [bwd]
  const float* dper;            // (Bn)          [bwd]
  unsigned short* dlogits;      // (N, V) bf16   [bwd]
  const unsigned short* Wt;     // W^T fragments for d h (k_lce_pack), or null   [bwd with d h]
  float* dhp;                   // (VSB, N, E) fp32 d h partials                [bwd with d h]
  int64_t N;
  int S, V, per;                // per: chunks (fwd) or chunk pairs (bwd) per vocabulary slice
  float smoothing;
};

// Wp[chunk c][k-step ks][lane (hh, r32)][j] = W[32 c + r32][16 ks + 8 hh + j] for ks < E / 16 (the A fragments of
// the logits tiles); the extra k-step ks = E / 16 holds {bf16 hi, bf16 lo} of bias[32 c + r32] in slots 0, 1 of the
// hh = 0 lanes (hi + lo carries the fp32 bias to ~2^-17 relative) and zeros elsewhere.
//
// Wtp (optional, for the fused d h product): the A fragments of d h^T (E x tokens) += W^T (E x 32 rows) dl (32 x tokens)
// with the contraction index in ACCUMULATOR order, so that the d logits tile a lane already holds is its B fragment:
// Wtp[chunk c][k-step s][E block eb][lane (hh, r32)][j] = W[32 c + 8 (2 s + (j >> 2)) + 4 hh + (j & 3)][32 eb + r32].
__global__ __launch_bounds__(256) void k_lce_pack(const unsigned short* __restrict__ W, const float* __restrict__ bias,
                                                   unsigned short* __restrict__ Wp, unsigned short* __restrict__ Wtp,
                                                   int64_t V, int E) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;       // one 16-byte group per thread
  const int nf = (E >> 4) + 1;
  const int64_t g1 = (V >> 5) * nf * 64;
  if (g >= g1) {
    const int EB = E >> 5;
    const int64_t g2 = g - g1;
    if (!Wtp || g2 >= (V >> 5) * 2 * EB * 64) return;
    const int lane = (int)(g2 & 63), r32 = lane & 31, hh = lane >> 5;
    const int64_t q = g2 >> 6;
    const int eb = (int)(q % EB), s2 = (int)((q / EB) & 1);
    const int64_t c = q / (2 * EB);
    u16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = W[(32 * c + 8 * (2 * s2 + (j >> 2)) + 4 * hh + (j & 3)) * E + 32 * eb + r32];
    *reinterpret_cast<u16x8*>(Wtp + g2 * 8) = v;
    return;
  }
  const int lane = (int)(g & 63);
  const int64_t q = g >> 6;
  const int ks = (int)(q % nf);
  const int64_t c = q / nf;
  const int r32 = lane & 31, hh = lane >> 5;
  u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (ks < nf - 1) {
    v = *reinterpret_cast<const u16x8*>(W + (32 * c + r32) * E + 16 * ks + 8 * hh);
  } else if (hh == 0 && bias) {
    const float b = bias[32 * c + r32];
    const unsigned short hi = f32_to_bf16(b);
    v[0] = hi;
    v[1] = f32_to_bf16(b - bf16_to_f32(hi));
    v[2] = v[3] = v[4] = 0x3F80;                   // against a per-token constant in slots 2..4 of h's column (k_lce_dwp)
  }
  *reinterpret_cast<u16x8*>(Wp + g * 8) = v;
}

template <int NKS, bool SMOOTH>
__global__ __launch_bounds__(256) void k_lce_fwd(LceArgs a) {
  constexpr int NF = NKS + 1;
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t tok = tile * 32 + r32;
  const bool valid = tok < a.N;
  const float w = valid ? a.weights[tok] : 0.f;
  f32x4* out = a.partial + (int64_t)blockIdx.y * a.N + tok;
  if (!__any(w != 0.f)) {                     // tile of padding tokens: nothing is read (wave-uniform)
    if (valid && hh == 0) { const f32x4 z = {-INFINITY, 0.f, 0.f, 0.f}; *out = z; }
    return;
  }
  lce_bf16x8 hf[NF];
  lce_load_h<NF>(a.h, tok, valid, hh, hf);
  const int t = valid ? (int)a.target[tok] : -1;
  const int nch = a.V >> 5;
  const int c0 = blockIdx.y * a.per;
  int c1 = c0 + a.per;
  if (c1 > nch) c1 = nch;
  float m = -INFINITY, s = 0.f, sx = 0.f, xt = 0.f;
  lce_bf16x8 wf[NF], wn[NF];
  if (c0 < c1) lce_load_w<NF>(a.W, c0, r32, hh, wf);
  for (int c = c0; c < c1; ++c) {
    if (c + 1 < c1) lce_load_w<NF>(a.W, c + 1, r32, hh, wn);      // next chunk in flight under this one's math
    const f32x16 x = lce_tile<NF>(wf, hf);
    float cm = x[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) cm = fmaxf(cm, x[r]);
    const float mn = fmaxf(m, cm * kLog2e);
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc += ex2(fmaf(x[r], kLog2e, -mn));
      if (SMOOTH) sx += x[r];
    }
    s = fmaf(s, ex2(m - mn), acc);
    m = mn;
    if (__any((t >> 5) == c)) {               // some token of this tile has its target in this chunk (wave-uniform)
      const int rel = t - 32 * c - 4 * hh;    // row of the target inside this lane's 16 rows, if any
#pragma unroll
      for (int r = 0; r < 16; ++r) xt += (rel == (r & 3) + 8 * (r >> 2)) ? x[r] : 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < NF; ++ks) wf[ks] = wn[ks];
  }
  // the two half-waves hold the two row halves of the same token column
  const float mo = __shfl_xor(m, 32, 64), so = __shfl_xor(s, 32, 64);
  const float M = fmaxf(m, mo);
  s = (m == -INFINITY ? 0.f : s * ex2(m - M)) + (mo == -INFINITY ? 0.f : so * ex2(mo - M));
  sx += __shfl_xor(sx, 32, 64);
  xt += __shfl_xor(xt, 32, 64);
  if (valid && hh == 0) { const f32x4 o = {M, s, sx, xt}; *out = o; }
}

// LDS-staged forward for E <= 64 (see k_lce_bwd_s): the block's 4 waves share one vocabulary slice, so the W fragments of
// every chunk pair are fetched once per block (register-staged into a double LDS buffer, one barrier per pair) instead
// of once per wave through the CU's L1.
template <int NKS, bool SMOOTH, int CPS = 2>
__global__ __launch_bounds__(256) void k_lce_fwd_s(LceArgs a) {
  constexpr int NF = NKS + 1;
  constexpr int WB = CPS * NF * 1024;                // bytes of W fragments per stage (CPS chunks: a pair, or one chunk at E = 256)
  constexpr int NV = (WB / 16 + 255) / 256;
  __shared__ __attribute__((aligned(16))) char wbuf[2][WB];
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t tok = tile * 32 + r32;
  const bool valid = tok < a.N;
  const float w = valid ? a.weights[tok] : 0.f;
  f32x4* out = a.partial + (int64_t)blockIdx.y * a.N + tok;
  const bool live = __any(w != 0.f) != 0;
  if (!live && valid && hh == 0) { const f32x4 z = {-INFINITY, 0.f, 0.f, 0.f}; *out = z; }
  if (!__syncthreads_or(live ? 1 : 0)) return;
  lce_bf16x8 hf[NF];
  lce_load_h<NF>(a.h, live ? tok : 0, live && valid, hh, hf);
  const int t = valid ? (int)a.target[tok] : -1;
  const int npair = (a.V >> 5) / CPS;                // stages; V is a multiple of 64
  const int per2 = (a.per + CPS - 1) / CPS;          // stages per slice (per = chunks per slice, a multiple of CPS here)
  const int p0 = blockIdx.y * per2;
  int p1 = p0 + per2;
  if (p1 > npair) p1 = npair;
  float m = -INFINITY, s = 0.f, sx = 0.f, xt = 0.f;
  const char* wsrc = reinterpret_cast<const char*>(a.W);
  u16x8 pre[NV];
  auto fetch = [&](int p) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int off = (v * 256 + (int)threadIdx.x) * 16;
      if (off < WB) pre[v] = *reinterpret_cast<const u16x8*>(wsrc + (int64_t)p * WB + off);
    }
  };
  auto commit = [&](int b) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int off = (v * 256 + (int)threadIdx.x) * 16;
      if (off < WB) *reinterpret_cast<u16x8*>(wbuf[b] + off) = pre[v];
    }
  };
  if (p0 < p1) { fetch(p0); commit(0); }
  __syncthreads();
  for (int p = p0; p < p1; ++p) {
    const int b = (p - p0) & 1;
    if (p + 1 < p1) fetch(p + 1);
    if (live) {
#pragma unroll
      for (int u = 0; u < CPS; ++u) {
        const int c = CPS * p + u;
        lce_bf16x8 wf[NF];
#pragma unroll
        for (int ks = 0; ks < NF; ++ks)
          wf[ks] = *reinterpret_cast<const lce_bf16x8*>(wbuf[b] + ((u * NF + ks) * 64 + lane) * 16);
        const f32x16 x = lce_tile<NF>(wf, hf);
        float cm = x[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) cm = fmaxf(cm, x[r]);
        const float mn = fmaxf(m, cm * kLog2e);
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc += ex2(fmaf(x[r], kLog2e, -mn));
          if (SMOOTH) sx += x[r];
        }
        s = fmaf(s, ex2(m - mn), acc);
        m = mn;
        if (__any((t >> 5) == c)) {
          const int rel = t - 32 * c - 4 * hh;
#pragma unroll
          for (int r = 0; r < 16; ++r) xt += (rel == (r & 3) + 8 * (r >> 2)) ? x[r] : 0.f;
        }
      }
    }
    if (p + 1 < p1) commit(b ^ 1);
    __syncthreads();
  }
  if (!live) return;
  const float mo = __shfl_xor(m, 32, 64), so = __shfl_xor(s, 32, 64);
  const float M = fmaxf(m, mo);
  s = (m == -INFINITY ? 0.f : s * ex2(m - M)) + (mo == -INFINITY ? 0.f : so * ex2(mo - M));
  sx += __shfl_xor(sx, 32, 64);
  xt += __shfl_xor(xt, 32, 64);
  if (valid && hh == 0) { const f32x4 o = {M, s, sx, xt}; *out = o; }
}

// tok_nll[r] = w_r * nll_r and lse[r] from the per-slice partials (slices folded in slice order)
__global__ __launch_bounds__(256) void k_lce_combine(const f32x4* __restrict__ partial, const float* __restrict__ weights,
                                                      float* __restrict__ tok_nll, float* __restrict__ lse_out,
                                                      int64_t N, int VS, int V, float smoothing) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= N) return;
  const float w = weights[r];
  if (w == 0.f) { tok_nll[r] = 0.f; lse_out[r] = 0.f; return; }
  float M = -INFINITY, s = 0.f, sx = 0.f, xt = 0.f;
  for (int v = 0; v < VS; ++v) {
    const f32x4 p = partial[(int64_t)v * N + r];
    const float Mn = fmaxf(M, p[0]);
    s = (M == -INFINITY ? 0.f : s * ex2(M - Mn)) + (p[0] == -INFINITY ? 0.f : p[1] * ex2(p[0] - Mn));
    M = Mn;
    sx += p[2];
    xt += p[3];
  }
  const float lse = (M + __builtin_amdgcn_logf(s)) * kLn2;
  float nll = lse - xt;
  if (smoothing > 0.f) nll = (1.f - smoothing) * nll + smoothing * lse - (smoothing / (float)V) * sx;
  tok_nll[r] = nll * w;
  lse_out[r] = lse;
}

// k_lce_combine and k_nll_seq in one launch: a wave per sentence folds the slice partials of its tokens (same arithmetic
// as k_lce_combine), then reduces them exactly as k_nll_seq does (lane-strided sums, wave_sum): bit-identical outputs.
__global__ __launch_bounds__(kBlock) void k_lce_combine_seq(const f32x4* __restrict__ partial, const float* __restrict__ weights,
                                                             float* __restrict__ tok_nll, float* __restrict__ lse_out,
                                                             float* __restrict__ per_seq, float* __restrict__ tokscale,
                                                             int64_t N, int64_t Bn, int S, int VS, int V, float smoothing) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (b >= Bn) return;
  float a = 0.0f, ws = 0.0f;
  for (int sp = lane; sp < S; sp += 64) {
    const int64_t r = b * S + sp;
    const float w = weights[r];
    float tn = 0.f, lse = 0.f;
    if (w != 0.f) {
      float M = -INFINITY, s = 0.f, sx = 0.f, xt = 0.f;
      for (int v = 0; v < VS; ++v) {
        const f32x4 p = partial[(int64_t)v * N + r];
        const float Mn = fmaxf(M, p[0]);
        s = (M == -INFINITY ? 0.f : s * ex2(M - Mn)) + (p[0] == -INFINITY ? 0.f : p[1] * ex2(p[0] - Mn));
        M = Mn;
        sx += p[2];
        xt += p[3];
      }
      lse = (M + __builtin_amdgcn_logf(s)) * kLn2;
      float nll = lse - xt;
      if (smoothing > 0.f) nll = (1.f - smoothing) * nll + smoothing * lse - (smoothing / (float)V) * sx;
      tn = nll * w;
    }
    tok_nll[r] = tn;
    lse_out[r] = lse;
    a += tn;
    ws += w;
  }
  a = wave_sum(a);
  ws = wave_sum(ws);
  const float inv = 1.0f / fmaxf(ws, 1e-13f);
  if (lane == 0) per_seq[b] = a * inv;
  for (int sp = lane; sp < S; sp += 64) tokscale[b * S + sp] = weights[b * S + sp] * inv;
}

constexpr int kLceStride = 72;   // bf16 per staged token row (64 + pad): 144 B keeps 16-byte alignment

// DH: every wave also accumulates d h^T (E x 32 tokens) += W^T dl on the matrix cores from the bf16 d logits it has
// just formed (the accumulator tile IS the B fragment, see k_lce_pack) and writes one fp32 partial tile per
// vocabulary slice (k_lce_dh_combine sums the slices in order): the (N, V) x (V, E) library GEMM and its read of
// d logits disappear.  The 4 waves of a block walk the same vocabulary slice, so W / W^T fragments hit in L1.
template <int NKS, bool DH>
__global__ __launch_bounds__(256) void k_lce_bwd(LceArgs a) {
  constexpr int NF = NKS + 1;
  constexpr int E = 16 * NKS, EB = (E + 31) / 32;
  __shared__ __attribute__((aligned(16))) unsigned short stage_all[4][32 * kLceStride];
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int wave = threadIdx.x >> 6;
  unsigned short* stage = stage_all[wave];
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int slice = (int)blockIdx.y;
  const int64_t tok0 = tile * 32, tok = tok0 + r32;
  if (tok0 >= a.N) return;                                   // wave-uniform
  const bool valid = tok < a.N;
  const float coef = valid ? a.dper[tok / a.S] * a.tokscale[tok] : 0.f;
  const int npair = a.V >> 6;
  const int p0 = slice * a.per;
  int p1 = p0 + a.per;
  if (p1 > npair) p1 = npair;
  // copy-out geometry: lane -> (row = lane >> 3 (+ 8 i), 16-byte column lane & 7)
  const int orow = lane >> 3, ocol = (lane & 7) * 8;
  if (!__any(coef != 0.f)) {                                 // padding tile: zero gradient rows, nothing read
    const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = p0; p < p1; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = tok0 + orow + 8 * i;
        if (row < a.N) *reinterpret_cast<u16x8*>(a.dlogits + row * a.V + 64 * p + ocol) = z;
      }
    if (DH) {                                                // zero d h partial rows of this tile
      const f32x4 zf = {0.f, 0.f, 0.f, 0.f};
      for (int idx = lane; idx < 32 * (E / 4); idx += 64) {
        const int row = idx / (E / 4), c4 = idx % (E / 4);
        if (tok0 + row < a.N)
          *reinterpret_cast<f32x4*>(a.dhp + ((int64_t)slice * a.N + tok0 + row) * E + 4 * c4) = zf;
      }
    }
    return;
  }
  lce_bf16x8 hf[NF];
  lce_load_h<NF>(a.h, tok, valid, hh, hf);
  const int t = valid ? (int)a.target[tok] : -1;
  f32x16 dh[EB];
#pragma unroll
  for (int eb = 0; eb < EB; ++eb)
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[eb][r] = 0.f;
  // coef * softmax = sign(coef) * 2^(x log2e - lse log2e + log2|coef|): one fma + one v_exp_f32 per element, the sign
  // goes on with an integer xor (coef = 0 -> 2^-inf = 0: padding tokens inside a live tile get exact zeros)
  const float c0 = valid ? __builtin_amdgcn_logf(fabsf(coef)) - a.lse[tok] * kLog2e : -INFINITY;
  const unsigned sgn = __float_as_uint(coef) & 0x80000000u;
  const float uni = coef * (a.smoothing > 0.f ? a.smoothing / (float)a.V : 0.f);
  const float hot = coef * (a.smoothing > 0.f ? 1.f - a.smoothing : 1.f);
  const bool smooth = a.smoothing > 0.f;
  lce_bf16x8 wf[2][NF];
  for (int p = p0; p < p1; ++p) {
    lce_load_w<NF>(a.W, 2 * p, r32, hh, wf[0]);
    lce_load_w<NF>(a.W, 2 * p + 1, r32, hh, wf[1]);
    // a target of this tile falls into this pair of chunks, or label smoothing: the slow element path (wave-uniform)
    const bool slow = smooth || (__any((t >> 6) == p) != 0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = 2 * p + u;
      lce_bf16x8 af[2][EB];
      if (DH) {                                              // in flight under this chunk's MFMA + exp work
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int eb = 0; eb < EB; ++eb)
            af[s2][eb] = *reinterpret_cast<const lce_bf16x8*>(a.Wt + ((((int64_t)c * 2 + s2) * EB + eb) * 64 + lane) * 8);
      }
      const f32x16 x = lce_tile<NF>(wf[u], hf);
      u16x4 ov[4];
      if (!slow) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u16x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            o[i] = f32_to_bf16(__uint_as_float(__float_as_uint(ex2(fmaf(x[4 * q + i], kLog2e, c0))) ^ sgn));
          *reinterpret_cast<u16x4*>(stage + r32 * kLceStride + 32 * u + 8 * q + 4 * hh) = o;
          ov[q] = o;
        }
      } else {
        const int rel = t - 32 * c - 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u16x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float g = __uint_as_float(__float_as_uint(ex2(fmaf(x[4 * q + i], kLog2e, c0))) ^ sgn) - uni;
            if (rel == i + 8 * q) g -= hot;
            o[i] = f32_to_bf16(g);
          }
          *reinterpret_cast<u16x4*>(stage + r32 * kLceStride + 32 * u + 8 * q + 4 * hh) = o;
          ov[q] = o;
        }
      }
      if (DH) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const u16x8 braw = {ov[2 * s2][0], ov[2 * s2][1], ov[2 * s2][2], ov[2 * s2][3],
                              ov[2 * s2 + 1][0], ov[2 * s2 + 1][1], ov[2 * s2 + 1][2], ov[2 * s2 + 1][3]};
          const lce_bf16x8 bfr = __builtin_bit_cast(lce_bf16x8, braw);
#pragma unroll
          for (int eb = 0; eb < EB; ++eb)
            dh[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s2][eb], bfr, dh[eb], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = orow + 8 * i;
      const u16x8 v = *reinterpret_cast<const u16x8*>(stage + row * kLceStride + ocol);
      if (tok0 + row < a.N) *reinterpret_cast<u16x8*>(a.dlogits + (tok0 + row) * a.V + 64 * p + ocol) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  if (DH && valid) {                                         // lane = token column; 4 consecutive E rows per store
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {dh[eb][4 * q], dh[eb][4 * q + 1], dh[eb][4 * q + 2], dh[eb][4 * q + 3]};
        *reinterpret_cast<f32x4*>(a.dhp + ((int64_t)slice * a.N + tok) * E + 32 * eb + 8 * q + 4 * hh) = v;
      }
  }
}

// LDS-staged variant for E <= 64: the 4 waves of a block (4 token tiles, one vocabulary slice) need the same W / W^T
// fragments, and with per-wave global loads the 20-36 KB per chunk pair and block go through the CU's L1 at more
// than its 64 B/clk -- the kernel was L1-bound, not HBM-bound.  Here the block fetches each pair's fragment image
// ONCE (register-staged: loads for pair p+1 are in flight under pair p's math, committed to the other LDS buffer at
// the end of the iteration, one barrier per pair) and every wave reads its fragments with conflict-free ds_read_b128.
template <int NKS, bool DH, int CPS = 2>
__global__ __launch_bounds__(256) void k_lce_bwd_s(LceArgs a) {
  constexpr int NF = NKS + 1;
  constexpr int E = 16 * NKS, EB = (E + 31) / 32;
  // a stage = a chunk PAIR (one 64-column piece of d logits), or, CPS = 1 (E >= 128: a pair's W + W^T fragments would
  // not leave room for two buffers), one chunk: a pair is then walked as two stages
  constexpr int WB = CPS * NF * 1024;                // bytes of W fragments per stage
  constexpr int TB = DH ? CPS * 2 * EB * 1024 : 0;   // bytes of W^T fragments per stage
  constexpr int SB = WB + TB;
  constexpr int NV = (SB / 16 + 255) / 256;          // 16-byte vectors per thread and stage
  __shared__ __attribute__((aligned(16))) unsigned short stage_all[4][32 * kLceStride];
  __shared__ __attribute__((aligned(16))) char wbuf[2][SB];
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int wave = threadIdx.x >> 6;
  unsigned short* stage = stage_all[wave];
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int slice = (int)blockIdx.y;
  const int64_t tok0 = tile * 32, tok = tok0 + r32;
  const bool valid = tok < a.N;
  const float coef = valid ? a.dper[tok / a.S] * a.tokscale[tok] : 0.f;
  const int npair = a.V >> 6;
  const int p0 = slice * a.per;
  int p1 = p0 + a.per;
  if (p1 > npair) p1 = npair;
  const int orow = lane >> 3, ocol = (lane & 7) * 8;
  const bool live = __any(coef != 0.f) != 0;                 // wave-uniform; tiles past N are not live either
  const bool store_dl = a.dlogits != nullptr;                // null: d h only (the d W product recomputes, k_lce_dw)
  if (!live) {                                               // padding tile: zero gradient rows, nothing read
    const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (store_dl)
      for (int p = p0; p < p1; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t row = tok0 + orow + 8 * i;
          if (row < a.N) *reinterpret_cast<u16x8*>(a.dlogits + row * a.V + 64 * p + ocol) = z;
        }
    if (DH) {
      const f32x4 zf = {0.f, 0.f, 0.f, 0.f};
      for (int idx = lane; idx < 32 * (E / 4); idx += 64) {
        const int row = idx / (E / 4), c4 = idx % (E / 4);
        if (tok0 + row < a.N)
          *reinterpret_cast<f32x4*>(a.dhp + ((int64_t)slice * a.N + tok0 + row) * E + 4 * c4) = zf;
      }
    }
  }
  if (!__syncthreads_or(live ? 1 : 0)) return;               // no live tile in this block: no weights needed
  lce_bf16x8 hf[NF];
  lce_load_h<NF>(a.h, live ? tok : 0, live && valid, hh, hf);
  const int t = valid ? (int)a.target[tok] : -1;
  f32x16 dh[EB];
#pragma unroll
  for (int eb = 0; eb < EB; ++eb)
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[eb][r] = 0.f;
  const float c0 = valid ? __builtin_amdgcn_logf(fabsf(coef)) - a.lse[tok] * kLog2e : -INFINITY;
  const unsigned sgn = __float_as_uint(coef) & 0x80000000u;
  const float uni = coef * (a.smoothing > 0.f ? a.smoothing / (float)a.V : 0.f);
  const float hot = coef * (a.smoothing > 0.f ? 1.f - a.smoothing : 1.f);
  const bool smooth = a.smoothing > 0.f;
  const char* wsrc = reinterpret_cast<const char*>(a.W);
  const char* tsrc = reinterpret_cast<const char*>(a.Wt);
  u16x8 pre[NV];
  auto fetch = [&](int p) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int off = (v * 256 + (int)threadIdx.x) * 16;
      if (off < WB) pre[v] = *reinterpret_cast<const u16x8*>(wsrc + (int64_t)p * WB + off);
      else if (DH && off < SB) pre[v] = *reinterpret_cast<const u16x8*>(tsrc + (int64_t)p * TB + (off - WB));
    }
  };
  auto commit = [&](int b) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int off = (v * 256 + (int)threadIdx.x) * 16;
      if (off < SB) *reinterpret_cast<u16x8*>(wbuf[b] + off) = pre[v];
    }
  };
  constexpr int SPP = 2 / CPS;                       // stages per pair
  const int s0 = p0 * SPP, s1 = p1 * SPP;
  if (s0 < s1) { fetch(s0); commit(0); }
  __syncthreads();
  for (int sg = s0; sg < s1; ++sg) {
    const int b = (sg - s0) & 1;
    const int p = sg / SPP;
    if (sg + 1 < s1) fetch(sg + 1);                          // in flight under this stage's math
    if (live) {
      const bool slow = smooth || (__any((t >> 6) == p) != 0);
#pragma unroll
      for (int uu = 0; uu < CPS; ++uu) {
        const int u = CPS == 2 ? uu : (sg & 1);              // chunk of the pair
        const int c = 2 * p + u;
        lce_bf16x8 wf[NF];
#pragma unroll
        for (int ks = 0; ks < NF; ++ks)
          wf[ks] = *reinterpret_cast<const lce_bf16x8*>(wbuf[b] + ((uu * NF + ks) * 64 + lane) * 16);
        const f32x16 x = lce_tile<NF>(wf, hf);
        u16x4 ov[4];
        if (!slow) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            u16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              o[i] = f32_to_bf16(__uint_as_float(__float_as_uint(ex2(fmaf(x[4 * q + i], kLog2e, c0))) ^ sgn));
            if (store_dl) *reinterpret_cast<u16x4*>(stage + r32 * kLceStride + 32 * u + 8 * q + 4 * hh) = o;
            ov[q] = o;
          }
        } else {
          const int rel = t - 32 * c - 4 * hh;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            u16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float g = __uint_as_float(__float_as_uint(ex2(fmaf(x[4 * q + i], kLog2e, c0))) ^ sgn) - uni;
              if (rel == i + 8 * q) g -= hot;
              o[i] = f32_to_bf16(g);
            }
            if (store_dl) *reinterpret_cast<u16x4*>(stage + r32 * kLceStride + 32 * u + 8 * q + 4 * hh) = o;
            ov[q] = o;
          }
        }
        if (DH) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const u16x8 braw = {ov[2 * s2][0], ov[2 * s2][1], ov[2 * s2][2], ov[2 * s2][3],
                                ov[2 * s2 + 1][0], ov[2 * s2 + 1][1], ov[2 * s2 + 1][2], ov[2 * s2 + 1][3]};
            const lce_bf16x8 bfr = __builtin_bit_cast(lce_bf16x8, braw);
#pragma unroll
            for (int eb = 0; eb < EB; ++eb) {
              const lce_bf16x8 af = *reinterpret_cast<const lce_bf16x8*>(
                  wbuf[b] + WB + (((uu * 2 + s2) * EB + eb) * 64 + lane) * 16);
              dh[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, dh[eb], 0, 0, 0);
            }
          }
        }
      }
      if (store_dl && (CPS == 2 || (sg & 1))) {              // both chunks of the pair are in the wave's transpose stage
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = orow + 8 * i;
          const u16x8 v = *reinterpret_cast<const u16x8*>(stage + row * kLceStride + ocol);
          if (tok0 + row < a.N) *reinterpret_cast<u16x8*>(a.dlogits + (tok0 + row) * a.V + 64 * p + ocol) = v;
        }
      }
    }
    if (sg + 1 < s1) commit(b ^ 1);
    __syncthreads();       // stage sg+1 visible; every wave is done with buffer b and with its transpose stage
  }
  if (DH && live && valid) {
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {dh[eb][4 * q], dh[eb][4 * q + 1], dh[eb][4 * q + 2], dh[eb][4 * q + 3]};
        *reinterpret_cast<f32x4*>(a.dhp + ((int64_t)slice * a.N + tok) * E + 32 * eb + 8 * q + 4 * hh) = v;
      }
  }
}

// d h with NO stored d logits (the d W product recomputes, k_lce_dw): the same arithmetic as k_lce_bwd_s<NKS, true> with
// dlogits = NULL, restructured so that the matrix pipe works under the exponentials instead of beside them.  Measured on
// k_lce_bwd_s at config 2 (tools/bench_lce.py, variants with one part removed): exponentials + conversions 29 us, logits
// MFMAs + their LDS reads 16, d h MFMAs 13, the per-pair barrier with its register-staged copy 14, the target-row branch
// 8 -- the parts ADD UP to the kernel time, i.e. nothing overlapped: a wave issued its five dependent logits MFMAs
// back to back (stalled 32 cycles each), then 130 VALU instructions, then four more MFMAs, and the partner wave on the
// SIMD sat in the same phase behind the same barrier.  Here, per 32 x 32 tile i of the wave's stream:
//   L(i+1)  logits MFMAs of the NEXT tile                      (matrix pipe)
//   V(i)    p = exp2(x log2e + c0), sign, bf16 pairs           (VALU)
//   D(i-1)  d h += W^T fragments x bf16 d logits of the PREVIOUS tile, fragments preloaded   (matrix pipe)
// are independent and sit in one branch-free region which the scheduler interleaves; the W / W^T fragment images of a
// chunk pair arrive by LDS-DMA (no registers, no ds_write) into a ring of three slots, requested two pairs ahead, one
// barrier per pair.  A target row inside the tile (one tile in eight at V = 8192) is PATCHED after V(i): the one element
// is recomputed with the `- hot` term and inserted into its packed word (wave-uniform register index), instead of a
// second copy of the whole tile loop with a compare per element.
typedef __attribute__((address_space(3))) void lce_lds_void_t;
typedef const __attribute__((address_space(1))) void lce_glb_cvoid_t;
typedef unsigned lce_u32x8 __attribute__((ext_vector_type(8)));
typedef unsigned lce_u32x4 __attribute__((ext_vector_type(4)));
typedef float lce_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 lce_bf16x2 __attribute__((ext_vector_type(2)));

template <int NKS, bool SMOOTH, int DBG = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lce_dh(LceArgs a) {
  constexpr int NF = NKS + 1;
  constexpr int E = 16 * NKS, EB = E / 32;
  constexpr int WB = 2 * NF * 1024;                  // bytes of W fragments per chunk pair
  constexpr int TB = 2 * 2 * EB * 1024;              // bytes of W^T fragments per chunk pair
  constexpr int SB = WB + TB, NP = SB / 1024;        // ring slot: NP pieces of 1 KiB
  constexpr int PW = (NP + 3) / 4;                   // LDS-DMA pieces per wave and pair
  // three ring slots as three OBJECTS: the compiler's wait insertion then knows that the fragment reads of slots A, B do
  // not touch the slot an LDS-DMA is filling (with one array and a computed slot it waits vmcnt(0) before every read)
  __shared__ __attribute__((aligned(1024))) char ringA[SB];
  __shared__ __attribute__((aligned(1024))) char ringB[SB];
  __shared__ __attribute__((aligned(1024))) char ringC[SB];
  __shared__ __attribute__((aligned(1024))) char ringD[SB];
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int slice = (int)blockIdx.y;
  const int64_t tok0 = tile * 32, tok = tok0 + r32;
  const bool valid = tok < a.N;
  const float coef = valid ? a.dper[tok / a.S] * a.tokscale[tok] : 0.f;
  const int npair = a.V >> 6;
  const int p0 = slice * a.per;
  int p1 = p0 + a.per;
  if (p1 > npair) p1 = npair;
  const bool live = __any(coef != 0.f) != 0;                 // wave-uniform; tiles past N are not live either
  if (!live || p0 >= p1) {                                   // padding tile, or a slice past the vocabulary: zero rows
    const f32x4 zf = {0.f, 0.f, 0.f, 0.f};
    for (int idx = lane; idx < 32 * (E / 4); idx += 64) {
      const int row = idx / (E / 4), c4 = idx % (E / 4);
      if (tok0 + row < a.N)
        *reinterpret_cast<f32x4*>(a.dhp + ((int64_t)slice * a.N + tok0 + row) * E + 4 * c4) = zf;
    }
  }
  if (!__syncthreads_or(live ? 1 : 0) || p0 >= p1) return;   // no live tile in this block: no weights needed
  lce_bf16x8 hf[NF];
  lce_load_h<NF>(a.h, live ? tok : 0, live && valid, hh, hf);
  const int t = valid ? (int)a.target[tok] : -1;
  f32x16 dh[EB];
#pragma unroll
  for (int eb = 0; eb < EB; ++eb)
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[eb][r] = 0.f;
  const float c0 = valid ? __builtin_amdgcn_logf(fabsf(coef)) - a.lse[tok] * kLog2e : -INFINITY;
  // d logits = sign(coef) (|coef| softmax - |coef| q): the sign of a token's cotangent is applied ONCE, to its finished
  // d h column (rounding to bf16 and the fp32 accumulation are symmetric in the sign: bit-identical to applying it to
  // every element), which takes a v_xor per element out of the loop
  const unsigned sgn = __float_as_uint(coef) & 0x80000000u;
  const float uni = fabsf(coef) * (SMOOTH ? a.smoothing / (float)a.V : 0.f);
  const float hot = fabsf(coef) * (SMOOTH ? 1.f - a.smoothing : 1.f);
  const char* wsrc = reinterpret_cast<const char*>(a.W);
  const char* tsrc = reinterpret_cast<const char*>(a.Wt);
  // LDS-DMA pieces round-robin over the four waves; every wave issues PW pieces per pair (the last may repeat one), from
  // wave-uniform bases: piece q of pair p is at sbase[k] + p * sstride[k]
  const char* sbase[PW];
  int sstride[PW], soff[PW];
#pragma unroll
  for (int k = 0; k < PW; ++k) {
    const int q = wave + 4 * k, piece = q < NP ? q : NP - 1;
    const bool isw = piece < WB / 1024;
    sbase[k] = isw ? wsrc + piece * 1024 : tsrc + (piece - WB / 1024) * 1024;
    sstride[k] = isw ? WB : TB;
    soff[k] = piece * 1024;
  }
  auto stage = [&](int p, char* slot) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PW; ++k)
      __builtin_amdgcn_global_load_lds((lce_glb_cvoid_t*)(sbase[k] + (int64_t)p * sstride[k] + (unsigned)(lane * 16)),
                                       (lce_lds_void_t*)(slot + soff[k]), 16, 0, 0);
  };
  lce_bf16x8 wfn[NF];                                        // W fragments of the tile after the current one
  auto load_w = [&](const char* slot, int u) {
#pragma unroll
    for (int ks = 0; ks < NF; ++ks) wfn[ks] = *reinterpret_cast<const lce_bf16x8*>(slot + ((u * NF + ks) * 64 + lane) * 16);
  };
  auto coefs = [&](const f32x16& x) -> lce_u32x8 {           // V: word w = bf16 pair of accumulator registers 2 w, 2 w + 1
    lce_u32x8 o;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      if (DBG & 8) { o[w] = (__float_as_uint(x[2 * w]) ^ __float_as_uint(x[2 * w + 1])) & 0x3f803f80u; continue; }
      float g0 = ex2(fmaf(x[2 * w], kLog2e, c0));
      float g1 = ex2(fmaf(x[2 * w + 1], kLog2e, c0));
      if (SMOOTH) { g0 -= uni; g1 -= uni; }
      const lce_f32x2 pr = {g0, g1};
      o[w] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, lce_bf16x2));   // v_cvt_pk_bf16_f32
    }
    return o;
  };
  auto fix_targets = [&](lce_u32x8& o, const f32x16& x, int c) {   // rows that are some token's target in chunk c
    unsigned long long m = __ballot(hh == 0 && (t >> 5) == c);
    while (m) {
      const int r = __builtin_ctzll(m);
      m &= m - 1;
      const int row = __builtin_amdgcn_readlane(t, r) - 32 * c;     // wave-uniform, 0 .. 31 = 8 q + 4 hh' + i
      const int e = 4 * (row >> 3) + (row & 3), owner = r + 32 * ((row >> 2) & 1);
      float g = ex2(fmaf(x[e], kLog2e, c0));
      if (SMOOTH) g -= uni;
      g -= hot;
      const unsigned g16 = f32_to_bf16(g), old = o[e >> 1];
      const unsigned nw = (e & 1) ? (old & 0xffffu) | (g16 << 16) : (old & 0xffff0000u) | g16;
      o[e >> 1] = lane == owner ? nw : old;
    }
  };
  lce_bf16x8 afp[2][EB];                                     // W^T fragments of the pending tile
  auto load_wt = [&](const char* slot, int u) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int eb = 0; eb < EB; ++eb)
        afp[s2][eb] = *reinterpret_cast<const lce_bf16x8*>(slot + WB + (((u * 2 + s2) * EB + eb) * 64 + lane) * 16);
  };
  auto product = [&](const lce_u32x8& o) {                   // D
    if (DBG & 16) { dh[0][0] += __uint_as_float((o[0] ^ o[1] ^ o[2] ^ o[3] ^ o[4] ^ o[5] ^ o[6] ^ o[7]) & 0x3f803f80u); return; }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const lce_u32x4 b4 = {o[4 * s2], o[4 * s2 + 1], o[4 * s2 + 2], o[4 * s2 + 3]};
      const lce_bf16x8 bfr = __builtin_bit_cast(lce_bf16x8, b4);
#pragma unroll
      for (int eb = 0; eb < EB; ++eb) dh[eb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afp[s2][eb], bfr, dh[eb], 0, 0, 0);
    }
  };
  // order of one tile's region: the fragment reads of the NEXT region first (a full region to land), then one MFMA per
  // five VALU / transcendental instructions (16 x (fma, exp) + 8 conversions beside NF + 2 EB MFMAs)
  auto pipe = [&](bool dma) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_group_barrier(0x100, NF + 2 * EB, 0);
#pragma unroll
    for (int i = 0; i < NF + 2 * EB; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x402, (SMOOTH ? 56 : 40) / (NF + 2 * EB) + 1, 0);
      if (dma && i >= NF + 2 * EB - PW) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // one LDS-DMA piece per late gap
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the reads landed long ago; the next region starts without a wait
  };
  stage(p0, ringA);
  if (p0 + 1 < p1) stage(p0 + 1, ringB);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (p0 + 2 < p1) stage(p0 + 2, ringC);
  if (DBG & 1) { stage(p0, ringD); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  f32x16 xa, xb;
  lce_u32x8 ovp = {0, 0, 0, 0, 0, 0, 0, 0};                  // no pending tile yet: a zero product on finite fragments
  load_w(ringA, 0);
  xa = lce_tile<NF>(wfn, hf);
  load_w(ringA, 1);
  load_wt(ringA, 0);
  __builtin_amdgcn_s_waitcnt(0xC07F);                        // lgkmcnt(0): the loop is entered with nothing in flight
  // one chunk pair: slot `cur` holds pair p, `nxt` pair p + 1 (past the last pair: stale or unwritten data, whose tile
  // is computed and discarded), pair p + 2 is in flight and `stg` receives pair p + 3: a request has two pair times to
  // land (every block walks the pairs in step, so each pair's first touch is an L2 miss for all of them at once)
  auto pair = [&](const char* cur, const char* nxt, char* stg, int p) __attribute__((always_inline)) {
    const int pn = p + 3 < p1 ? p + 3 : p1 - 1;              // (past the end: a pair that is there already, into a free slot)
    {                                                        // (a padding tile in a live block computes zeros: c0 = -inf)
      {
        xb = lce_tile<NF>(wfn, hf);                          // L(p, 1)
        load_w(nxt, 0);
        lce_u32x8 ov = coefs(xa);
        product(ovp);
        load_wt(cur, 0);
        if (!(DBG & 1)) stage(pn, stg);
        pipe(!(DBG & 1));
        if (!(DBG & 4) && __any((t >> 5) == 2 * p)) fix_targets(ov, xa, 2 * p);
        ovp = ov;
      }
      {
        xa = lce_tile<NF>(wfn, hf);                          // L(p + 1, 0)
        load_w(nxt, 1);
        lce_u32x8 ov = coefs(xb);
        product(ovp);
        load_wt(cur, 1);
        pipe(false);
        if (!(DBG & 4) && __any((t >> 5) == 2 * p + 1)) fix_targets(ov, xb, 2 * p + 1);
        ovp = ov;
      }
    }
    if (!(DBG & 2)) {
      // this wave's pieces of pair p + 2 have landed (all but the PW just issued), its fragment reads are complete; then
      // the barrier: pair p + 2 is there for every wave and nobody reads slot `cur` any more.  (A raw barrier:
      // __syncthreads() carries a fence that waits for EVERY outstanding request, the new ones included.)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt((DBG & 1) ? 0x0070 : (0x0070 | PW));               // vmcnt(PW) lgkmcnt(0), visible to the compiler's own counters
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };
  for (int p = p0;;) {
    pair(ringA, ringB, ringD, p);
    if (++p >= p1) break;
    pair(ringB, ringC, ringA, p);
    if (++p >= p1) break;
    pair(ringC, ringD, ringB, p);
    if (++p >= p1) break;
    pair(ringD, ringA, ringC, p);
    if (++p >= p1) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA in flight when the wave ends
  if (!live) return;
  product(ovp);
  if (valid) {
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(__float_as_uint(dh[eb][4 * q + i]) ^ sgn);
        *reinterpret_cast<f32x4*>(a.dhp + ((int64_t)slice * a.N + tok) * E + 32 * eb + 8 * q + 4 * hh) = v;
      }
  }
}

// d h (N, E) bf16 = sum over the VSB vocabulary-slice partials, in slice order
__global__ __launch_bounds__(256) void k_lce_dh_combine(const float* __restrict__ partial, unsigned short* __restrict__ dh,
                                                         int64_t NE, int VSB) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= NE) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(partial + i);
  for (int v = 1; v < VSB; ++v) s += *reinterpret_cast<const f32x4*>(partial + (int64_t)v * NE + i);
  const u16x4 o = {f32_to_bf16(s[0]), f32_to_bf16(s[1]), f32_to_bf16(s[2]), f32_to_bf16(s[3])};
  *reinterpret_cast<u16x4*>(dh + i) = o;
}

// ---- d W, d b of the fused projection + NLL without the (N, V) d logits ------------------------------------------
// Vocabulary-major: a block owns one chunk pair (64 vocabulary rows; its W fragments stay in registers) and a share of
// the token tiles; every wave walks its tiles, RECOMPUTES the logits tile transposed (rows = tokens in the accumulator
// registers, columns = vocabulary on the lanes: the same two operands of k_lce_bwd_s with their roles swapped), forms
// the bf16 d logits exactly as that kernel does, and contracts them with h over the tokens -- the accumulator tile is
// the B fragment (contraction index = accumulator row), h^T comes out of the wave's LDS copy of the h tile through the
// transposing read in the same permuted token order.  d W^T (E x 64) and d b accumulate in registers over all tiles; the
// four waves are combined through LDS in wave order and the block writes one partial per token split: part (TS, V, E + 8)
// with d b in column E (the layout mivae_fold_aug_sink folds).  The 268 MB (C2) d logits tensor is neither written by
// the backward pass nor read by a GEMM.
struct LceDwArgs {
  const unsigned short* h;
  const unsigned short* W;      // packed (k_lce_pack)
  const int64_t* target;
  const float* lse;
  const float* tokscale;
  const float* dper;
  float* part;                  // (TS, V, E + 8)
  int64_t N;
  int S, V, tiles_per_split;
  float smoothing;
};

// f[j] = tile[16 s + 8 (j >> 2) + 4 hh + (j & 3)][fbase + r32]: the accumulator-row order of MmaBF16::from_acc<s>
__device__ __forceinline__ lce_bf16x8 lce_tr_frag_perm(const unsigned short* tile, int stride, int fbase, int s, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const unsigned short* a0 = tile + (16 * s + 4 * (g >> 1) + q) * stride + fbase + 16 * (g & 1) + 4 * p;
  typedef __attribute__((address_space(3))) bf16x4_tr* lds_ptr;
  const bf16x4_tr v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(a0));
  const bf16x4_tr v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(a0 + 8 * stride));
  return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int NKS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lce_dw(LceDwArgs a) {
  constexpr int NF = NKS + 1;
  constexpr int E = 16 * NKS, EB = (E + 31) / 32;
  constexpr int HST = E + 8;                                 // h tile row stride (elements)
  constexpr int WB = 2 * NF * 1024;                          // bytes of W fragments per chunk pair
  constexpr int ACC = EB * 2 * 16;                           // accumulator registers per lane
  __shared__ __attribute__((aligned(16))) float red[2 * ACC * 64 + 2 * 2 * 64];   // wave-pair reduction (32 KiB at E = 64)
  __shared__ __attribute__((aligned(16))) float tokc[4][3][32];
  unsigned short* htile = reinterpret_cast<unsigned short*>(red) + (threadIdx.x >> 6) * (32 * HST);   // aliases `red`
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int p = blockIdx.x, ts = blockIdx.y;
  lce_bf16x8 wf[2][NF];
  {
    const char* wsrc = reinterpret_cast<const char*>(a.W) + (int64_t)p * WB;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ks = 0; ks < NF; ++ks)
        wf[u][ks] = *reinterpret_cast<const lce_bf16x8*>(wsrc + ((u * NF + ks) * 64 + lane) * 16);
  }
  f32x16 dW[EB][2];
  float db[2] = {0.f, 0.f};
#pragma unroll
  for (int eb = 0; eb < EB; ++eb)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW[eb][u][r] = 0.f;
  const bool smooth = a.smoothing > 0.f;
  const float usc = smooth ? a.smoothing / (float)a.V : 0.f, hsc = smooth ? 1.f - a.smoothing : 1.f;
  const int64_t ntiles = (a.N + 31) / 32;
  const int64_t t0 = (int64_t)ts * a.tiles_per_split;
  int64_t t1 = t0 + a.tiles_per_split;
  if (t1 > ntiles) t1 = ntiles;
  // the next tile's operands are requested while the current tile is being processed (two waves per SIMD do not hide
  // a global round trip per tile on their own)
  lce_bf16x8 hn[NKS];
  float coef_n = 0.f, lse_n = 0.f;
  int t_n = -1;
  const u16x8 one_raw = {0x3F80, 0x3F80, 0, 0, 0, 0, 0, 0}, zero_raw = {0, 0, 0, 0, 0, 0, 0, 0};
  const lce_bf16x8 hbias = __builtin_bit_cast(lce_bf16x8, hh == 0 ? one_raw : zero_raw);   // the [1, 1, 0, ...] bias column
  auto request = [&](int64_t tile) {
    const int64_t tok = tile * 32 + r32;
    const bool valid = tile < t1 && tok < a.N;
    const unsigned short* hp = a.h + (valid ? tok : 0) * E + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) hn[ks] = *reinterpret_cast<const lce_bf16x8*>(hp + 16 * ks);
    coef_n = valid ? a.dper[tok / a.S] * a.tokscale[tok] : 0.f;
    lse_n = valid ? a.lse[tok] : 0.f;
    t_n = valid ? (int)a.target[tok] : -1;
  };
  request(t0 + wave);
  for (int64_t tile = t0 + wave; tile < t1; tile += 4) {
    lce_bf16x8 hf[NF];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) hf[ks] = hn[ks];
    hf[NKS] = hbias;
    const float coef = coef_n, lse = lse_n;
    const int t = t_n;
    request(tile + 4);
    if (__any(coef != 0.f) == 0) continue;                   // padding tile: contributes nothing (wave-uniform)
    // per-token scalars and the h tile through the wave's LDS area
    if (hh == 0) {
      tokc[wave][0][r32] = coef != 0.f ? __builtin_amdgcn_logf(fabsf(coef)) - lse * kLog2e : -INFINITY;
      tokc[wave][1][r32] = coef;
      tokc[wave][2][r32] = __int_as_float(t);
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      *reinterpret_cast<lce_bf16x8*>(htile + r32 * HST + 16 * ks + 8 * hh) = hf[ks];
    const bool slow = smooth || (__any((t >> 6) == p) != 0);
    const bool negs = __any(coef < 0.f) != 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    lce_bf16x8 hT[EB][2];
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) hT[eb][s2] = lce_tr_frag_perm(htile, HST, 32 * eb, s2, lane);
    f32x4 c0v[4];                                            // rows 8 q + 4 hh + {0..3} of this lane's accumulator
    unsigned sgn = 0;                                        // bit r: coef of accumulator row r is negative
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      c0v[q] = *reinterpret_cast<const f32x4*>(&tokc[wave][0][8 * q + 4 * hh]);
      const f32x4 cf = *reinterpret_cast<const f32x4*>(&tokc[wave][1][8 * q + 4 * hh]);
#pragma unroll
      for (int i = 0; i < 4; ++i) sgn |= (__float_as_uint(cf[i]) >> 31) << (4 * q + i);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x16 x;
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < NF; ++ks) x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[ks], wf[u][ks], x, 0, 0, 0);
      const int v = 64 * p + 32 * u + r32;                   // this lane's vocabulary row
      typedef float f32x2g __attribute__((ext_vector_type(2)));
      typedef __bf16 bf16x2g __attribute__((ext_vector_type(2)));
      typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
      u32x4g bw[2];
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float g2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float g = ex2(fmaf(x[r + i], kLog2e, c0v[(r + i) >> 2][(r + i) & 3]));
          if (negs)                                          // (wave-uniform) a negative cotangent in this tile: rare
            g = __uint_as_float(__float_as_uint(g) ^ ((sgn >> (r + i)) << 31));
          if (slow) {                                        // (wave-uniform) smoothing, or a target inside this pair
            const int row = 8 * ((r + i) >> 2) + 4 * hh + ((r + i) & 3);
            const float cf = tokc[wave][1][row];
            g -= cf * usc;
            if (__float_as_int(tokc[wave][2][row]) == v) g -= cf * hsc;
          }
          sum += g;                                          // d b from the unrounded values (fp32)
          g2[i] = g;
        }
        const f32x2g pr = {g2[0], g2[1]};
        bw[r >> 3][(r & 7) >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2g));   // v_cvt_pk_bf16_f32
      }
      db[u] += sum;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const lce_bf16x8 bfr = __builtin_bit_cast(lce_bf16x8, bw[s2]);
#pragma unroll
        for (int eb = 0; eb < EB; ++eb) dW[eb][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hT[eb][s2], bfr, dW[eb][u], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();                         // the tile area is rewritten by the next iteration
  }
  // d b: the two lane halves hold the rows 4 hh + ... of the same column
#pragma unroll
  for (int u = 0; u < 2; ++u) db[u] += __shfl_xor(db[u], 32, 64);
  // combine the four waves in fixed order: (0 + 2), (1 + 3), then (0 + 1)
  __syncthreads();                                           // every wave is done with its h tile (aliases red)
  float* slot = red + (wave & 1) * (ACC * 64) + lane;
  float* dslot = red + 2 * ACC * 64 + (wave & 1) * 128 + lane;
  if (wave >= 2) {
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[((eb * 2 + u) * 16 + r) * 64] = dW[eb][u][r];
    dslot[0] = db[0]; dslot[64] = db[1];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[eb][u][r] += slot[((eb * 2 + u) * 16 + r) * 64];
    db[0] += dslot[0]; db[1] += dslot[64];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((eb * 2 + u) * 16 + r) * 64 + lane] = dW[eb][u][r];
    red[2 * ACC * 64 + lane] = db[0]; red[2 * ACC * 64 + 64 + lane] = db[1];
  }
  __syncthreads();
  if (wave != 0) return;
  float* P = a.part + ((int64_t)ts * a.V) * (E + 8);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float* row = P + (int64_t)(64 * p + 32 * u + r32) * (E + 8);
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = dW[eb][u][4 * q + i] + red[((eb * 2 + u) * 16 + 4 * q + i) * 64 + lane];
        if (32 * eb + 8 * q + 4 * hh < E) *reinterpret_cast<f32x4*>(row + 32 * eb + 8 * q + 4 * hh) = o;
      }
    if (hh == 0) row[E] = db[u] + red[2 * ACC * 64 + 64 * u + lane];
  }
}

// k_lce_dw for smoothing = 0, software-pipelined like k_lce_dh: per wave the stream of units (token tile i, chunk u of
// the block's vocabulary pair) runs L(next unit) / V(current) / D(previous) in one scheduled region, the W fragments of
// the pair stay in registers and the h tile goes through a private double-buffered LDS tile (written once per tile;
// read back as the A fragments of L and, transposed by ds_read_b64_tr_b16, as the A fragments of D): no block barrier
// and no shared staging inside the loop.  What made the un-pipelined loop slow is taken out of the per-element work:
//  * the per-token constant c = ln|coef| - lse rides on the matrix pipe: it sits (as three bf16 pieces, ~24 bits) in
//    slots 2..4 of the token side's bias column against ones on the vocabulary side, so the accumulator already holds
//    x + c and a d logits element is exp2(acc log2e): one multiply, one exponential, one add (d b), half a conversion;
//  * a target inside the unit (one unit in four at V = 8192) is patched afterwards: the one element is recomputed
//    with its `- |coef|` term and inserted into the packed word (wave-uniform register index), d b corrected;
//  * rows with a negative cotangent (none in training: the loss weights are positive) are sign-flipped the same way.
// Padding tiles are computed (c = -1e30 gives exact zeros) rather than skipped.
template <int NKS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lce_dwp(LceDwArgs a) {
  constexpr int NF = NKS + 1;
  constexpr int E = 16 * NKS, EB = (E + 31) / 32;
  constexpr int HST = E + 8;                                 // h tile row stride (elements): conflict-free b128 and tr reads
  constexpr int TILE = 32 * HST;                             // elements per tile buffer
  constexpr int WB = 2 * NF * 1024;
  constexpr int ACC = EB * 2 * 16;
  constexpr int RED = 2 * ACC * 64 + 2 * 2 * 64;             // floats of the end reduction
  constexpr int LDSF = RED > 4 * 2 * TILE / 2 ? RED : 4 * 2 * TILE / 2;
  __shared__ __attribute__((aligned(16))) float red[LDSF];   // the tile buffers alias the reduction area
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned short* tbuf = reinterpret_cast<unsigned short*>(red) + wave * (2 * TILE);
  const int p = blockIdx.x, ts = blockIdx.y;
  lce_bf16x8 wf[2][NF];
  {
    const char* wsrc = reinterpret_cast<const char*>(a.W) + (int64_t)p * WB;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ks = 0; ks < NF; ++ks)
        wf[u][ks] = *reinterpret_cast<const lce_bf16x8*>(wsrc + ((u * NF + ks) * 64 + lane) * 16);
  }
  f32x16 dW[EB][2];
  float db[2] = {0.f, 0.f};
#pragma unroll
  for (int eb = 0; eb < EB; ++eb)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW[eb][u][r] = 0.f;
  const int64_t ntiles = (a.N + 31) / 32;
  const int64_t t0 = (int64_t)ts * a.tiles_per_split;
  int64_t t1 = t0 + a.tiles_per_split;
  if (t1 > ntiles) t1 = ntiles;
  const int niter = t1 > t0 + wave ? (int)((t1 - t0 - wave + 3) / 4) : 0;
  // ---- stages ----
  // G: the global -> register leg of one tile (tile i of this wave's sequence; past the end: zeros).  Raw loads only:
  // any arithmetic on them here would wait for the round trip on the spot.
  struct Req {
    lce_bf16x8 h[NKS];
    float dp, tsc, lse;
    int t;
  };
  auto is_valid = [&](int i) __attribute__((always_inline)) -> bool {
    const int64_t tile = t0 + wave + 4 * (int64_t)i;
    return tile < t1 && tile * 32 + r32 < a.N;
  };
  auto request = [&](int i, Req& r) __attribute__((always_inline)) {   // unconditional loads (token 0 when out of range)
    const int64_t tok = is_valid(i) ? (t0 + wave + 4 * (int64_t)i) * 32 + r32 : 0;
    const unsigned short* hp = a.h + tok * E + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) r.h[ks] = *reinterpret_cast<const lce_bf16x8*>(hp + 16 * ks);
    r.dp = a.dper[(unsigned)tok / (unsigned)a.S];            // N < 2^31 (lce_check): 32-bit division
    r.tsc = a.tokscale[tok];
    r.lse = a.lse[tok];
    r.t = reinterpret_cast<const int*>(a.target)[2 * tok];   // low half of the int64 (0 <= target < V)
  };
  auto store_tile = [&](unsigned short* buf, const Req& r) __attribute__((always_inline)) {   // S
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) *reinterpret_cast<lce_bf16x8*>(buf + r32 * HST + 16 * ks + 8 * hh) = r.h[ks];
  };
  // the token side's last k-step: [1, 1, c_hi, c_lo, c_lo2, 0, 0, 0] in the hh = 0 lanes, c = ln|coef| - lse
  auto bias_col = [&](float coef, float lse) __attribute__((always_inline)) -> lce_bf16x8 {
    const float c = coef != 0.f ? (__builtin_amdgcn_logf(fabsf(coef)) - lse * kLog2e) * 0.6931471805599453f : -1e30f;
    const unsigned short c1 = f32_to_bf16(c);
    const float r1 = c - bf16_to_f32(c1);
    const unsigned short c2 = f32_to_bf16(r1);
    const unsigned short c3 = f32_to_bf16(r1 - bf16_to_f32(c2));
    const u16x8 v = {0x3F80, 0x3F80, c1, c2, c3, 0, 0, 0}, z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(lce_bf16x8, hh == 0 ? v : z);
  };
  auto logits = [&](const unsigned short* buf, const lce_bf16x8& hb, int u) __attribute__((always_inline)) -> f32x16 {   // L
    lce_bf16x8 hf[NF];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) hf[ks] = *reinterpret_cast<const lce_bf16x8*>(buf + r32 * HST + 16 * ks + 8 * hh);
    hf[NKS] = hb;
    f32x16 x;
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NF; ++ks) x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[ks], wf[u][ks], x, 0, 0, 0);
    return x;
  };
  auto coefs = [&](const f32x16& x, float& sum) __attribute__((always_inline)) -> lce_u32x8 {   // V
    lce_u32x8 o;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float g0 = ex2(x[2 * w] * kLog2e), g1 = ex2(x[2 * w + 1] * kLog2e);
      sum += g0;
      sum += g1;
      const lce_f32x2 pr = {g0, g1};
      o[w] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, lce_bf16x2));
    }
    return o;
  };
  lce_bf16x8 hT[2][EB];                                      // transposed fragments of the tile whose products are pending
  auto load_ht = [&](const unsigned short* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int eb = 0; eb < EB; ++eb) hT[s2][eb] = lce_tr_frag_perm(buf, HST, 32 * eb, s2, lane);
  };
  auto product = [&](const lce_u32x8& o, int u) __attribute__((always_inline)) {   // D
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const lce_u32x4 b4 = {o[4 * s2], o[4 * s2 + 1], o[4 * s2 + 2], o[4 * s2 + 3]};
      const lce_bf16x8 bfr = __builtin_bit_cast(lce_bf16x8, b4);
#pragma unroll
      for (int eb = 0; eb < EB; ++eb) dW[eb][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hT[s2][eb], bfr, dW[eb][u], 0, 0, 0);
    }
  };
  // rare fixes of one unit: rows (tokens) with a negative cotangent, then rows whose target lies in chunk c
  auto fix = [&](lce_u32x8& o, const f32x16& x, float& sum, int c, int t, float coef) {
    unsigned long long m = __ballot(hh == 0 && coef < 0.f);
    while (m) {
      const int row = __builtin_ctzll(m);
      m &= m - 1;
      const int e = 4 * (row >> 3) + (row & 3);
      const bool mine = hh == ((row >> 2) & 1);
      const float g = ex2(x[e] * kLog2e);
      const unsigned old = o[e >> 1];
      o[e >> 1] = mine ? old ^ ((e & 1) ? 0x80000000u : 0x00008000u) : old;
      if (mine) sum -= 2.f * g;
    }
    m = __ballot(hh == 0 && (t >> 5) == c);
    while (m) {
      const int row = __builtin_ctzll(m);
      m &= m - 1;
      const int col = __builtin_amdgcn_readlane(t, row) & 31;
      const float cf = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(coef), row));
      const int e = 4 * (row >> 3) + (row & 3), owner = col + 32 * ((row >> 2) & 1);
      float g = ex2(x[e] * kLog2e) - fabsf(cf);
      if (cf < 0.f) g = -g;
      const unsigned g16 = f32_to_bf16(g);
      const unsigned old = o[e >> 1];
      const unsigned nw = (e & 1) ? (old & 0xffffu) | (g16 << 16) : (old & 0xffff0000u) | g16;
      o[e >> 1] = lane == owner ? nw : old;
      if (lane == owner) sum -= cf;                          // sign(cf) |cf|
    }
  };
  auto pipe = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_group_barrier(0x100, NKS, 0);     // the A fragments of L first
#pragma unroll
    for (int i = 0; i < NF + 2 * EB; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x402, 56 / (NF + 2 * EB) + 1, 0);
    }
  };
  // ---- prologue: tile 0 in buffer 0, tile 1 requested ----
  unsigned short* bufc = tbuf;                               // tile i
  unsigned short* bufn = tbuf + TILE;                        // tile i + 1
  Req rq;
  request(0, rq);
  store_tile(bufc, rq);
  float coef_c = is_valid(0) ? rq.dp * rq.tsc : 0.f;
  int t_c = is_valid(0) ? rq.t : -1;
  lce_bf16x8 hb_c = bias_col(coef_c, rq.lse);
  request(1, rq);
  load_ht(bufc);
  f32x16 x0 = logits(bufc, hb_c, 0), x1;
  lce_u32x8 ovp = {0, 0, 0, 0, 0, 0, 0, 0};                  // no pending unit yet: a zero product
  for (int i = 0; i < niter; ++i) {                          // rq holds tile i + 1 (requested a tile ago)
    lce_bf16x8 hb_n;
    lce_u32x8 ov0, ov1;
    float coef_x;
    int t_x;
    {                                                        // region A: unit (i, 0)
      x1 = logits(bufc, hb_c, 1);                            // L(i, 1)
      ov0 = coefs(x0, db[0]);                                // V(i, 0)
      product(ovp, 1);                                       // D(i - 1, 1) on the previous tile's transposed fragments
      store_tile(bufn, rq);                                  // S(i + 1)
      coef_x = is_valid(i + 1) ? rq.dp * rq.tsc : 0.f;
      t_x = is_valid(i + 1) ? rq.t : -1;
      hb_n = bias_col(coef_x, rq.lse);
      load_ht(bufc);                                         // for D(i, 0), D(i, 1)
      pipe();
      if (__any(coef_c < 0.f || (t_c >> 5) == 2 * p)) fix(ov0, x0, db[0], 2 * p, t_c, coef_c);
    }
    request(i + 2, rq);                                      // G(i + 2): a region and a half to arrive, nothing of it used before
    {                                                        // region B: unit (i, 1)
      x0 = logits(bufn, hb_n, 0);                            // L(i + 1, 0)
      ov1 = coefs(x1, db[1]);                                // V(i, 1)
      product(ov0, 0);                                       // D(i, 0)
      pipe();
      if (__any(coef_c < 0.f || (t_c >> 5) == 2 * p + 1)) fix(ov1, x1, db[1], 2 * p + 1, t_c, coef_c);
    }
    ovp = ov1;
    hb_c = hb_n;
    coef_c = coef_x;
    t_c = t_x;
    unsigned short* sw = bufc;
    bufc = bufn;
    bufn = sw;
  }
  product(ovp, 1);
  (void)x0;
  // d b: the two lane halves hold the rows 4 hh + ... of the same column
#pragma unroll
  for (int u = 0; u < 2; ++u) db[u] += __shfl_xor(db[u], 32, 64);
  // combine the four waves in fixed order: (0 + 2), (1 + 3), then (0 + 1)
  __syncthreads();                                           // every wave is done with its h tile (aliases red)
  float* slot = red + (wave & 1) * (ACC * 64) + lane;
  float* dslot = red + 2 * ACC * 64 + (wave & 1) * 128 + lane;
  if (wave >= 2) {
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[((eb * 2 + u) * 16 + r) * 64] = dW[eb][u][r];
    dslot[0] = db[0]; dslot[64] = db[1];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[eb][u][r] += slot[((eb * 2 + u) * 16 + r) * 64];
    db[0] += dslot[0]; db[1] += dslot[64];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((eb * 2 + u) * 16 + r) * 64 + lane] = dW[eb][u][r];
    red[2 * ACC * 64 + lane] = db[0]; red[2 * ACC * 64 + 64 + lane] = db[1];
  }
  __syncthreads();
  if (wave != 0) return;
  float* P = a.part + ((int64_t)ts * a.V) * (E + 8);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float* row = P + (int64_t)(64 * p + 32 * u + r32) * (E + 8);
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = dW[eb][u][4 * q + i] + red[((eb * 2 + u) * 16 + 4 * q + i) * 64 + lane];
        if (32 * eb + 8 * q + 4 * hh < E) *reinterpret_cast<f32x4*>(row + 32 * eb + 8 * q + 4 * hh) = o;
      }
    if (hh == 0) row[E] = db[u] + red[2 * ACC * 64 + 64 * u + lane];
  }
}


// token splits of mivae_lce_dw: ~512 blocks of 4 waves (two resident per CU)
static int lce_dw_splits(int64_t N, int64_t V) {
  const int64_t pairs = V / 64, tiles = (N + 31) / 32;
  int64_t tsn = (512 + pairs - 1) / pairs;
  if (tsn > tiles / 4) tsn = tiles / 4;
  if (tsn > 16) tsn = 16;
  return tsn < 1 ? 1 : (int)tsn;
}

int lce_check(const char* name, int64_t Bn, int64_t S, int64_t E, int64_t V, float smoothing) {
  MIVAE_REQUIRE(Bn >= 0 && S > 0 && S < (1ll << 31) && Bn * S < (1ll << 31), "%s: bad shape Bn=%lld S=%lld", name,
                (long long)Bn, (long long)S);
  MIVAE_REQUIRE(E == 32 || E == 64 || E == 128 || E == 256, "%s: E must be 32, 64, 128 or 256 (got %lld)", name, (long long)E);
  MIVAE_REQUIRE(V >= 64 && V % 64 == 0 && V < (1ll << 31), "%s: V must be a multiple of 64 (got %lld)", name, (long long)V);
  MIVAE_REQUIRE(smoothing >= 0.f && smoothing < 1.f, "%s: smoothing must be in [0, 1)", name);
  return MIVAE_OK;
}

// vocabulary slices so that ~8192 waves are in flight (8 per SIMD: exp latency and the MFMA chain overlap), at most 64
int lce_slices(int64_t N, int64_t units) {
  const int64_t tiles = (N + 31) / 32;
  static const int64_t target = getenv("MIVAE_LCE_WAVES") ? atoll(getenv("MIVAE_LCE_WAVES")) : 8192;
  int64_t vs = (target + tiles - 1) / tiles;
  if (vs > 64) vs = 64;
  if (vs > units) vs = units;
  if (vs < 1) vs = 1;
  return (int)vs;
}

// E = 128 / 256: LDS-staged W fragments as for E <= 64 (MIVAE_LCE_WIDE_STAGED=0: one fetch per wave through L1)
static bool lce_wide_staged() {
  const char* e = getenv("MIVAE_LCE_WIDE_STAGED");
  return !(e && atoi(e) == 0);
}

}  // namespace

extern "C" int64_t mivae_lce_workspace_floats(int64_t N, int64_t V) {
  if (N <= 0 || V <= 0) return 0;
  return (int64_t)lce_slices(N, V / 32) * N * 4 + N;
}

// W (V, E) bf16 row-major + bias (V) fp32 (or NULL) -> Wp (V * (E + 16) bf16): what mivae_lce_fwd / _bwd read
extern "C" int64_t mivae_lce_pack_elems(int64_t V, int64_t E) { return V > 0 && E > 0 ? V * (E + 16) : 0; }

// Wtp (V * E bf16, may be NULL): the W^T fragment image mivae_lce_bwd_dh reads, written by the same launch.
extern "C" int mivae_lce_pack2(const void* W, const float* bias, void* Wp, void* Wtp, int64_t V, int64_t E,
                               void* stream) {
  MIVAE_REQUIRE(W && Wp, "lce_pack: null pointer");
  MIVAE_REQUIRE((E == 32 || E == 64 || E == 128 || E == 256) && V >= 32 && V % 32 == 0 && V < (1ll << 31),
                "lce_pack: bad shape V=%lld E=%lld", (long long)V, (long long)E);
  MIVAE_REQUIRE(aligned16(W) && aligned16(Wp) && aligned16(Wtp), "lce_pack: operands must be 16-byte aligned");
  int64_t groups = (V / 32) * (E / 16 + 1) * 64;
  if (Wtp) groups += (V / 32) * 2 * (E / 32) * 64;
  k_lce_pack<<<(unsigned)((groups + 255) / 256), 256, 0, S_(stream)>>>((const unsigned short*)W, bias, (unsigned short*)Wp,
                                                                         (unsigned short*)Wtp, V, (int)E);
  MIVAE_CHECK_LAUNCH("lce_pack");
  return MIVAE_OK;
}

extern "C" int mivae_lce_pack(const void* W, const float* bias, void* Wp, int64_t V, int64_t E, void* stream) {
  return mivae_lce_pack2(W, bias, Wp, nullptr, V, E, stream);
}

// per_seq (Bn), lse (N), tokscale (N) as mivae_nll_fwd; h (N, E) bf16, W = packed weight + bias (mivae_lce_pack).
extern "C" int mivae_lce_fwd(const void* h, const void* W, const int64_t* target,
                             const float* weights, float* per_seq, float* lse, float* tokscale, float* workspace,
                             int64_t Bn, int64_t S, int64_t E, int64_t V, float smoothing, void* stream) {
  int rc = lce_check("lce_fwd", Bn, S, E, V, smoothing);
  if (rc) return rc;
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(h && W && target && weights && per_seq && lse && tokscale && workspace, "lce_fwd: null pointer");
  MIVAE_REQUIRE(aligned16(h) && aligned16(W) && aligned16(workspace), "lce_fwd: operands must be 16-byte aligned");
  const int64_t N = Bn * S;
  LceArgs a = {};
  a.h = (const unsigned short*)h; a.W = (const unsigned short*)W; a.target = target; a.weights = weights;
  a.partial = reinterpret_cast<f32x4*>(workspace);
  a.N = N; a.S = (int)S; a.V = (int)V; a.smoothing = smoothing;
  const int VS = lce_slices(N, V / 32);
  a.per = (int)((V / 32 + VS - 1) / VS);
  if (E <= 128) a.per = (a.per + 1) / 2 * 2;         // the staged kernel walks chunk PAIRS: whole pairs per slice
  float* tok_nll = workspace + (int64_t)VS * N * 4;
  dim3 grid((unsigned)(((N + 31) / 32 + 3) / 4), (unsigned)VS);
  hipStream_t st = S_(stream);
#define LCE_FWD(NKS)                                                         \
  do {                                                                       \
    if (smoothing > 0.f) k_lce_fwd<NKS, true><<<grid, 256, 0, st>>>(a);      \
    else k_lce_fwd<NKS, false><<<grid, 256, 0, st>>>(a);                     \
  } while (0)
#define LCE_FWD_S(NKS)                                                       \
  do {                                                                       \
    if (smoothing > 0.f) k_lce_fwd_s<NKS, true><<<grid, 256, 0, st>>>(a);    \
    else k_lce_fwd_s<NKS, false><<<grid, 256, 0, st>>>(a);                   \
  } while (0)
  switch (E) {
    case 32: LCE_FWD_S(2); break;
    case 64: LCE_FWD_S(4); break;
    case 128: if (lce_wide_staged()) LCE_FWD_S(8); else LCE_FWD(8); break;
    default:
      if (lce_wide_staged()) {
        if (smoothing > 0.f) k_lce_fwd_s<16, true, 1><<<grid, 256, 0, st>>>(a);
        else k_lce_fwd_s<16, false, 1><<<grid, 256, 0, st>>>(a);
      } else LCE_FWD(16);
      break;
  }
#undef LCE_FWD
#undef LCE_FWD_S
  MIVAE_CHECK_LAUNCH("lce_fwd");
  if (S >= 16 && S <= 128) {   // one launch: a wave per sentence folds its tokens' partials (at most two per lane) and reduces them
    k_lce_combine_seq<<<(unsigned)((Bn + kBlock / 64 - 1) / (kBlock / 64)), kBlock, 0, st>>>(a.partial, weights, tok_nll, lse, per_seq,
                                                                                     tokscale, N, Bn, (int)S, VS, (int)V, smoothing);
    MIVAE_CHECK_LAUNCH("lce_combine_seq");
    return MIVAE_OK;
  }                    // very short or long sentences: a thread per token, then a wave per sentence
  k_lce_combine<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(a.partial, weights, tok_nll, lse, N, VS, (int)V, smoothing);
  MIVAE_CHECK_LAUNCH("lce_combine");
  k_nll_seq<<<(unsigned)((Bn + 3) / 4), kBlock, 0, st>>>(tok_nll, weights, per_seq, tokscale, Bn, (int)S);
  MIVAE_CHECK_LAUNCH("lce_seq");
  return MIVAE_OK;
}

// dlogits (N, V) bf16 = dper_seq[b] * tokscale * (softmax(h W^T + b) - q); the caller forms d h, d W, d b from it.
extern "C" int mivae_lce_bwd(const void* h, const void* W, const int64_t* target, const float* lse,
                             const float* tokscale, const float* dper_seq, void* dlogits, int64_t Bn, int64_t S,
                             int64_t E, int64_t V, float smoothing, void* stream) {
  int rc = lce_check("lce_bwd", Bn, S, E, V, smoothing);
  if (rc) return rc;
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(h && W && target && lse && tokscale && dper_seq && dlogits, "lce_bwd: null pointer");
  MIVAE_REQUIRE(aligned16(h) && aligned16(W) && aligned16(dlogits), "lce_bwd: operands must be 16-byte aligned");
  const int64_t N = Bn * S;
  LceArgs a = {};
  a.h = (const unsigned short*)h; a.W = (const unsigned short*)W; a.target = target;
  a.lse = lse; a.tokscale = tokscale; a.dper = dper_seq; a.dlogits = (unsigned short*)dlogits;
  a.N = N; a.S = (int)S; a.V = (int)V; a.smoothing = smoothing;
  const int VS = lce_slices(N, V / 64);
  a.per = (int)((V / 64 + VS - 1) / VS);
  dim3 grid((unsigned)(((N + 31) / 32 + 3) / 4), (unsigned)VS);
  hipStream_t st = S_(stream);
  switch (E) {
    case 32: k_lce_bwd_s<2, false><<<grid, 256, 0, st>>>(a); break;
    case 64: k_lce_bwd_s<4, false><<<grid, 256, 0, st>>>(a); break;
    case 128: if (lce_wide_staged()) k_lce_bwd_s<8, false, 1><<<grid, 256, 0, st>>>(a); else k_lce_bwd<8, false><<<grid, 256, 0, st>>>(a); break;
    default: if (lce_wide_staged()) k_lce_bwd_s<16, false, 1><<<grid, 256, 0, st>>>(a); else k_lce_bwd<16, false><<<grid, 256, 0, st>>>(a); break;
  }
  MIVAE_CHECK_LAUNCH("lce_bwd");
  return MIVAE_OK;
}

// vocabulary slices of mivae_lce_bwd_dh: ~2048 waves (one resident set at 2 waves / SIMD) keeps the fp32 d h
// partials (slices x N x E) small
static int lce_dh_blocks(int64_t N, int64_t V) {
  const int64_t tiles = (N + 31) / 32;
  static const int64_t target = getenv("MIVAE_LCE_DH_WAVES") ? atoll(getenv("MIVAE_LCE_DH_WAVES")) : 2048;
  int64_t vs = (target + tiles - 1) / tiles;
  if (vs > 64) vs = 64;
  if (vs > V / 64) vs = V / 64;
  return vs < 1 ? 1 : (int)vs;
}

extern "C" int64_t mivae_lce_dh_workspace_floats(int64_t N, int64_t E, int64_t V) {
  if (N <= 0 || E <= 0 || V <= 0) return 0;
  return (int64_t)lce_dh_blocks(N, V) * N * E;
}

// mivae_lce_bwd that also produces d h (N, E) bf16 = dlogits W in the same pass (Wtp from mivae_lce_pack2,
// workspace of mivae_lce_dh_workspace_floats(N, E, V) floats); dlogits is written for a d W product on it, or NULL
// (E <= 64) when d W comes from mivae_lce_dw.
extern "C" int mivae_lce_bwd_dh(const void* h, const void* W, const void* Wtp, const int64_t* target, const float* lse,
                                const float* tokscale, const float* dper_seq, void* dlogits, float* workspace, void* dh,
                                int64_t Bn, int64_t S, int64_t E, int64_t V, float smoothing, void* stream) {
  int rc = lce_check("lce_bwd_dh", Bn, S, E, V, smoothing);
  if (rc) return rc;
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(h && W && Wtp && target && lse && tokscale && dper_seq && workspace && dh, "lce_bwd_dh: null pointer");
  MIVAE_REQUIRE(dlogits || E <= 64, "lce_bwd_dh: d logits may be omitted for E <= 64 only");
  MIVAE_REQUIRE(aligned16(h) && aligned16(W) && aligned16(Wtp) && aligned16(dlogits) && aligned16(workspace) &&
                aligned16(dh), "lce_bwd_dh: operands must be 16-byte aligned");
  const int64_t N = Bn * S;
  LceArgs a = {};
  a.h = (const unsigned short*)h; a.W = (const unsigned short*)W; a.Wt = (const unsigned short*)Wtp; a.target = target;
  a.lse = lse; a.tokscale = tokscale; a.dper = dper_seq; a.dlogits = (unsigned short*)dlogits; a.dhp = workspace;
  a.N = N; a.S = (int)S; a.V = (int)V; a.smoothing = smoothing;
  const int VSB = lce_dh_blocks(N, V);
  a.per = (int)((V / 64 + VSB - 1) / VSB);
  dim3 grid((unsigned)(((N + 31) / 32 + 3) / 4), (unsigned)VSB);
  hipStream_t st = S_(stream);
  static const bool pipelined = !(getenv("MIVAE_LCE_DH_PIPE") && atoi(getenv("MIVAE_LCE_DH_PIPE")) == 0);
  if (!dlogits && pipelined) {
    if (E == 32) { if (smoothing > 0.f) k_lce_dh<2, true><<<grid, 256, 0, st>>>(a); else k_lce_dh<2, false><<<grid, 256, 0, st>>>(a); }
    else { if (smoothing > 0.f) k_lce_dh<4, true><<<grid, 256, 0, st>>>(a); else {
#ifdef MIVAE_LCE_ABLATION   // timing ablations of DESIGN.md section 6 (wrong results by construction): a build of their own only
      static const int dbg = getenv("MIVAE_LCE_DBG") ? atoi(getenv("MIVAE_LCE_DBG")) : 0;
      switch (dbg) {
#define DBGCASE(n) case n: k_lce_dh<4, false, n><<<grid, 256, 0, st>>>(a); break;
        DBGCASE(1) DBGCASE(3) DBGCASE(4) DBGCASE(7) DBGCASE(15) DBGCASE(23) DBGCASE(31) DBGCASE(12) DBGCASE(20)
        default: k_lce_dh<4, false><<<grid, 256, 0, st>>>(a); break;
      }
#else
      k_lce_dh<4, false><<<grid, 256, 0, st>>>(a);
#endif
    } }
  } else
  switch (E) {
    case 32: k_lce_bwd_s<2, true><<<grid, 256, 0, st>>>(a); break;
    case 64: k_lce_bwd_s<4, true><<<grid, 256, 0, st>>>(a); break;
    case 128: if (lce_wide_staged()) k_lce_bwd_s<8, true, 1><<<grid, 256, 0, st>>>(a); else k_lce_bwd<8, true><<<grid, 256, 0, st>>>(a); break;
    default: if (lce_wide_staged()) k_lce_bwd_s<16, true, 1><<<grid, 256, 0, st>>>(a); else k_lce_bwd<16, true><<<grid, 256, 0, st>>>(a); break;
  }
  MIVAE_CHECK_LAUNCH("lce_bwd_dh");
  k_lce_dh_combine<<<(unsigned)((N * E / 4 + 255) / 256), 256, 0, st>>>(workspace, (unsigned short*)dh, N * E, VSB);
  MIVAE_CHECK_LAUNCH("lce_dh_combine");
  return MIVAE_OK;
}

// number of partial slabs mivae_lce_dw writes (the G of mivae_fold_aug_sink); 0 when the shape is not supported
extern "C" int64_t mivae_lce_dw_slabs(int64_t N, int64_t E, int64_t V) {
  if (N <= 0 || V < 64 || V % 64 || (E != 32 && E != 64)) return 0;
  return lce_dw_splits(N, V);
}

// part (slabs, V, E + 8) fp32: per token split the partial [d W | d b | 7 unused columns] of the fused projection + NLL,
// d W = d logits^T h, d b = d logits^T 1 with the bf16 d logits of mivae_lce_bwd recomputed tile by tile (never stored).
// Sum over the slab axis (mivae_fold_aug_sink / mivae_colsum) = the gradients.  E in {32, 64}.
extern "C" int mivae_lce_dw(const void* h, const void* W, const int64_t* target, const float* lse, const float* tokscale,
                            const float* dper_seq, float* part, int64_t Bn, int64_t S, int64_t E, int64_t V,
                            float smoothing, void* stream) {
  int rc = lce_check("lce_dw", Bn, S, E, V, smoothing);
  if (rc) return rc;
  if (E != 32 && E != 64) {
    mivae_set_error("lce_dw: E must be 32 or 64 (got %lld)", (long long)E);
    return MIVAE_EUNSUPPORTED;
  }
  MIVAE_REQUIRE(Bn >= 1, "lce_dw: empty batch");
  MIVAE_REQUIRE(h && W && target && lse && tokscale && dper_seq && part, "lce_dw: null pointer");
  MIVAE_REQUIRE(aligned16(h) && aligned16(W) && aligned16(part), "lce_dw: operands must be 16-byte aligned");
  const int64_t N = Bn * S;
  LceDwArgs a = {};
  a.h = (const unsigned short*)h; a.W = (const unsigned short*)W; a.target = target; a.lse = lse; a.tokscale = tokscale;
  a.dper = dper_seq; a.part = part; a.N = N; a.S = (int)S; a.V = (int)V; a.smoothing = smoothing;
  const int TS = lce_dw_splits(N, V);
  a.tiles_per_split = (int)(((N + 31) / 32 + TS - 1) / TS);
  dim3 grid((unsigned)(V / 64), (unsigned)TS);
  static const bool pipelined = !(getenv("MIVAE_LCE_DW_PIPE") && atoi(getenv("MIVAE_LCE_DW_PIPE")) == 0);
  if (pipelined && smoothing == 0.f) {
    if (E == 32) k_lce_dwp<2><<<grid, 256, 0, S_(stream)>>>(a);
    else k_lce_dwp<4><<<grid, 256, 0, S_(stream)>>>(a);
  } else if (E == 32) k_lce_dw<2><<<grid, 256, 0, S_(stream)>>>(a);
  else k_lce_dw<4><<<grid, 256, 0, S_(stream)>>>(a);
  MIVAE_CHECK_LAUNCH("lce_dw");
  return MIVAE_OK;
}
