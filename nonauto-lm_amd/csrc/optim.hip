// Step-harness tail on flat buffers: gradient-norm clip, value clip, Adam and
// gradient zeroing in ONE pass over the packed parameter / gradient buckets.
//
// Replaces, from DefaultTrainer._train_batch (training/trainers/trainer.py:286-289, 359-367):
//   clip_grad_norm_(params, grad_norm); clip_grad_value_(params, grad_clip); optimizer.step();
//   optimizer.zero_grad()                      with optimizer = torch.optim.Adam (nn/optimizer.py:11-33)
// which is a multi-tensor walk of ~120 parameters, four times, plus per-parameter scalar kernels for the
// bias corrections.  Here: k_sumsq_partial (one read of the gradients, fixed-order partial sums) and
// k_adam_clip (every block folds the partials in the same order -> identical clip coefficient, then
// p, g, m, v are read once and p, m, v written once, g written as zero).  HBM-bound: 16 B read + 16 B
// written per parameter, + 4 B for the norm pass.  No atomics: bit-reproducible.
#include "common.hpp"

using namespace mivae;

namespace {

constexpr int kBlock = 256;

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// partial[b] = sum of g[i]^2 over the contiguous slab of block b (slab = ceil(n / nblocks) rounded to 4)
__global__ __launch_bounds__(kBlock) void k_sumsq_partial(const float* __restrict__ g, float* __restrict__ partial,
                                                           int64_t n, int64_t slab) {
  __shared__ float red[4];
  const int64_t lo = (int64_t)blockIdx.x * slab;
  int64_t hi = lo + slab;
  if (hi > n) hi = n;
  float s = 0.f;
  for (int64_t i = lo + 4 * (int64_t)threadIdx.x; i < hi; i += 4 * kBlock) {
    if (i + 4 <= hi) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
      s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    } else {
      for (int64_t j = i; j < hi; ++j) s += g[j] * g[j];
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

struct AdamArgs {
  float *p, *g, *m, *v;
  unsigned short* p16;    // optional bf16 shadow of p (the copy the bf16 forward reads), or null
  const float* partial;   // sums of squares over ALL buckets
  const float* step;      // device scalar: number of steps taken so far
  float* norm_out;        // optional: total gradient norm (written by block 0)
  int64_t n;
  int npartial;
  double lr, b1, b2;
  float eps, wd, max_norm, max_value;
  float gscale;           // every gradient (and so the norm) is multiplied by this first: 1 / world after a SUM all-reduce
};

__global__ __launch_bounds__(kBlock) void k_adam_clip(AdamArgs a) {
  __shared__ float red[4];
  // total norm: every block sums the same partials in the same order -> the same coefficient everywhere
  float s = 0.f;
  for (int i = threadIdx.x; i < a.npartial; i += kBlock) s += a.partial[i];
  const float norm = sqrtf(block_sum(s, red)) * a.gscale;
  if (a.norm_out && blockIdx.x == 0 && threadIdx.x == 0) *a.norm_out = norm;
  float coef = 1.0f;
  if (a.max_norm > 0.f) coef = fminf(a.max_norm / (norm + 1e-6f), 1.0f);   // clip_grad_norm_
  if (a.gscale != 1.0f) coef *= a.gscale;
  // bias corrections in double like the host arithmetic of torch.optim.Adam (1 - beta ** step): two double pow() and a
  // double sqrt, computed by ONE thread per block and handed round through LDS (every thread used to run them, which
  // cost more than the thread's share of the update)
  __shared__ float bcs[2];
  if (threadIdx.x == 0) {
    const double t = (double)*a.step + 1.0;
    const double bc1 = 1.0 - pow(a.b1, t), bc2 = 1.0 - pow(a.b2, t);
    bcs[0] = (float)(a.lr / bc1);
    bcs[1] = (float)sqrt(bc2);
  }
  __syncthreads();
  const float step_size = bcs[0], bc2_sqrt = bcs[1];
  const float omb1 = (float)(1.0 - a.b1), b2f = (float)a.b2, omb2 = (float)(1.0 - a.b2);
  const bool vclip = a.max_value > 0.f;
  const int64_t stride = (int64_t)gridDim.x * kBlock * 4;
  for (int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4; i < a.n; i += stride) {
    // buckets are padded to multiples of 64 floats: whole 16-byte vectors
    f32x4 g = *reinterpret_cast<const f32x4*>(a.g + i);
    f32x4 p = *reinterpret_cast<const f32x4*>(a.p + i);
    f32x4 m = *reinterpret_cast<const f32x4*>(a.m + i);
    f32x4 v = *reinterpret_cast<const f32x4*>(a.v + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ge = g[e] * coef;
      if (vclip) ge = fminf(fmaxf(ge, -a.max_value), a.max_value);           // clip_grad_value_
      if (a.wd != 0.f) ge = fmaf(a.wd, p[e], ge);
      m[e] = m[e] + (ge - m[e]) * omb1;                                      // exp_avg.lerp_(grad, 1 - beta1)
      v[e] = fmaf(ge * ge, omb2, v[e] * b2f);                                // mul_(beta2).addcmul_(g, g, 1 - beta2)
      const float denom = sqrtf(v[e]) / bc2_sqrt + a.eps;
      p[e] = p[e] - step_size * (m[e] / denom);
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(a.p + i) = p;
    if (a.p16) {
      u16x4 q = {f32_to_bf16(p[0]), f32_to_bf16(p[1]), f32_to_bf16(p[2]), f32_to_bf16(p[3])};
      *reinterpret_cast<u16x4*>(a.p16 + i) = q;
    }
    *reinterpret_cast<f32x4*>(a.m + i) = m;
    *reinterpret_cast<f32x4*>(a.v + i) = v;
    *reinterpret_cast<f32x4*>(a.g + i) = zero;                               // optimizer.zero_grad()
  }
}

__global__ void k_scalar_add(float* x, float d) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *x += d;
}

}  // namespace

extern "C" int64_t mivae_sumsq_blocks(int64_t n) {
  if (n <= 0) return 0;
  int64_t b = (n + 4095) / 4096;          // >= 4k floats per block: a 5 MB gradient already fills the chip
  return b > 1024 ? 1024 : b;
}

extern "C" int mivae_sumsq_partial(const float* g, float* partial, int64_t n, void* stream) {
  MIVAE_REQUIRE(n >= 0, "sumsq_partial: n < 0");
  if (n == 0) return MIVAE_OK;
  MIVAE_REQUIRE(g && partial, "sumsq_partial: null pointer");
  MIVAE_REQUIRE(aligned16(g), "sumsq_partial: gradients must be 16-byte aligned");
  const int64_t nb = mivae_sumsq_blocks(n);
  const int64_t slab = ((n + nb - 1) / nb + 3) / 4 * 4;
  k_sumsq_partial<<<(unsigned)nb, kBlock, 0, S_(stream)>>>(g, partial, n, slab);
  MIVAE_CHECK_LAUNCH("sumsq_partial");
  return MIVAE_OK;
}

extern "C" int mivae_adam_clip_step(float* p, void* p_bf16, float* g, float* m, float* v, const float* partial,
                                    int64_t npartial, const float* step, float* norm_out, int64_t n, double lr,
                                    double beta1, double beta2, float eps, float weight_decay, float max_norm,
                                    float max_value, float grad_scale, void* stream) {
  MIVAE_REQUIRE(n >= 0 && npartial >= 0 && npartial < (1 << 20), "adam_clip_step: bad sizes");
  if (n == 0) return MIVAE_OK;
  MIVAE_REQUIRE(p && g && m && v && step && (partial || npartial == 0), "adam_clip_step: null pointer");
  MIVAE_REQUIRE(n % 4 == 0 && aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v),
                "adam_clip_step: buffers must be 16-byte aligned and padded to a multiple of 4 floats");
  MIVAE_REQUIRE(grad_scale > 0.f, "adam_clip_step: grad_scale must be positive");
  MIVAE_REQUIRE(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0.f,
                "adam_clip_step: bad hyper-parameters");
  AdamArgs a = {p, g, m, v, static_cast<unsigned short*>(p_bf16), partial, step, norm_out, n, (int)npartial, lr, beta1, beta2, eps, weight_decay,
                max_norm, max_value, grad_scale};
  int64_t blocks = (n / 4 + kBlock - 1) / kBlock;
  if (blocks > 2048) blocks = 2048;
  k_adam_clip<<<(unsigned)blocks, kBlock, 0, S_(stream)>>>(a);
  MIVAE_CHECK_LAUNCH("adam_clip_step");
  return MIVAE_OK;
}

extern "C" int mivae_scalar_add(float* x, float delta, void* stream) {
  MIVAE_REQUIRE(x, "scalar_add: null pointer");
  k_scalar_add<<<1, 64, 0, S_(stream)>>>(x, delta);
  MIVAE_CHECK_LAUNCH("scalar_add");
  return MIVAE_OK;
}

// dst[i] += (src[i] ? *src[i] : inc[i]) for up to four device scalars in one launch: the running metric sums of the step
// (models/base/model.py:200-210) and the optimizer's step counter, which a replayed graph advances together.
struct Acc4 { float* dst[4]; const float* src[4]; float inc[4]; int n; };
__global__ void k_accumulate4(Acc4 a) {
  const int i = threadIdx.x;
  if (i < a.n && a.dst[i]) *a.dst[i] += a.src[i] ? *a.src[i] : a.inc[i];
}
extern "C" int mivae_accumulate_scalars(float* const* dst, const float* const* src, const float* inc, int64_t n, void* stream) {
  MIVAE_REQUIRE(dst && src && inc && n >= 0 && n <= 4, "accumulate_scalars: up to four scalars");
  if (n == 0) return MIVAE_OK;
  Acc4 a = {};
  a.n = (int)n;
  for (int i = 0; i < (int)n; ++i) { a.dst[i] = dst[i]; a.src[i] = src[i]; a.inc[i] = inc[i]; }
  k_accumulate4<<<1, 64, 0, S_(stream)>>>(a);
  MIVAE_CHECK_LAUNCH("accumulate_scalars");
  return MIVAE_OK;
}

// ---- measurement aid: a plain streaming copy (16 bytes per lane, 4 loads in flight per lane) ----------------------
// bench.py times it on a 256 MiB buffer to report the copy bandwidth THIS box reaches next to the 8 TB/s datasheet
// figure (SURVEY 8d asks for both denominators).  Not part of the product path.  A block walks a contiguous segment,
// 16 KiB per iteration (4 x 256 lanes x 16 B); mode 1 uses non-temporal loads / stores.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(256) k_hbm_copy(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, int64_t n16,
                                                  int64_t per_block) {
  const int64_t lo = (int64_t)blockIdx.x * per_block;
  int64_t hi = lo + per_block;
  if (hi > n16) hi = n16;
  int64_t i = lo + threadIdx.x;
  for (; i + 768 < hi; i += 1024) {
    u32x4_t a, b, c, d;
    if (NT) {
      a = __builtin_nontemporal_load(src + i); b = __builtin_nontemporal_load(src + i + 256);
      c = __builtin_nontemporal_load(src + i + 512); d = __builtin_nontemporal_load(src + i + 768);
      __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + 256);
      __builtin_nontemporal_store(c, dst + i + 512); __builtin_nontemporal_store(d, dst + i + 768);
    } else {
      a = src[i]; b = src[i + 256]; c = src[i + 512]; d = src[i + 768];
      dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
    }
  }
  for (; i < hi; i += 256) dst[i] = src[i];
}

extern "C" int mivae_hbm_copy(const void* src, void* dst, int64_t bytes, int mode, void* stream) {
  MIVAE_REQUIRE(src && dst && bytes >= 0 && bytes % 16 == 0, "hbm_copy: bad arguments");
  MIVAE_REQUIRE(aligned16(src) && aligned16(dst), "hbm_copy: buffers must be 16-byte aligned");
  if (bytes == 0) return MIVAE_OK;
  const int64_t n16 = bytes / 16;
  int64_t blocks = (n16 + 1023) / 1024;
  const int64_t cap = mode >= 16 ? (int64_t)256 * (mode / 16) : 256 * 16;   // mode / 16 = blocks per CU (default 16)
  if (blocks > cap) blocks = cap;
  const int64_t per_block = ((n16 + blocks - 1) / blocks + 1023) / 1024 * 1024;
  if (mode & 1) k_hbm_copy<true><<<(unsigned)blocks, 256, 0, S_(stream)>>>((const u32x4_t*)src, (u32x4_t*)dst, n16, per_block);
  else k_hbm_copy<false><<<(unsigned)blocks, 256, 0, S_(stream)>>>((const u32x4_t*)src, (u32x4_t*)dst, n16, per_block);
  MIVAE_CHECK_LAUNCH("hbm_copy");
  return MIVAE_OK;
}
