// Mixed flow stack in ONE launch for narrow latents (D <= 128): any sequence of NICE feed-forward coupling steps,
// ActNorm steps and invertible-linear steps (single matrix or one matrix shared by all heads), in the density
// direction or the sampling direction.  Evaluation / sampling path (no gradients): training keeps one launch set per
// flow, because the parameter gradients of ActNorm and of the linear flows are reductions over all tokens.
//
// Reference semantics, per step: NICECoupling.forward / .backward (vae_lm/models/base/flows/couplings.py:141-183) with
// FeedForwardCoupling nets (:50-51), ActNorm.forward / .backward (flows/actnorm.py:30-54), InvertibleLinearFlow and
// InvertibleMultiHeadFlow (flows/invertible_flows.py:26-44, :58-83), iterated like FlowPrior.forward_pass /
// backward_pass (nonauto_lm/priors/flow_prior.py:37-70).  The parameter-only log-determinants (sum log_scale x tokens,
// slogdet x tokens x heads) are added by the caller; the kernel returns the data-dependent part (the NICE scales).
//
// Design: a wave64 owns 32 consecutive tokens for the whole step list; the tile lives transposed (feature x token,
// fp32) in the wave's LDS area, so z is read once and written once whatever the number of steps.  All products run
// on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation) for both latent dtypes: the step matrices are
// staged per step into LDS by the whole block (odd row pitch: conflict-free fragment reads), the hidden layer of a
// coupling net goes from the first product's accumulator straight into the second product's B operand in registers
// (the permuted-k trick of nice_stack.hip), and an ActNorm step is elementwise on the resident tile.
#include "nice_common.hpp"

using namespace mivae;

namespace {

constexpr int kMaxSteps = 24;
constexpr int kMaxD = 128;
constexpr int kMaxH = 128;
enum { STEP_NICE = 0, STEP_ACTNORM = 1, STEP_LINEAR = 2 };

struct MixStep {
  int kind, H, act, hs;
  const float* p[8];   // NICE: W1s b1s W2s b2s W1t b1t W2t b2t | ActNorm: log_scale bias | linear: M (hs, hs)
};
struct MixArgs {
  const void* z_in;
  void* z_out;
  const int64_t* mask;
  float* tok_val;
  int64_t N;
  int S, mask_div, D, K, inverse;
  MixStep steps[kMaxSteps];
};

__host__ __device__ inline int even_up(int x) { return (x + 1) & ~1; }

// LDS floats of the staging region for one NICE net / one linear step
__host__ __device__ inline int nice_region(int h, int H, int HT) {
  const int Hp = (H + 31) / 32 * 32;
  return Hp * (even_up(h) + 1) + 32 * HT * (Hp + 1) + Hp + 32 * HT;
}
__host__ __device__ inline int linear_region(int D, int HT) { return 64 * HT * (even_up(D) + 1); }

// Block-cooperative copy of a (rows x cols) fp32 matrix (row pitch src_ld) into an LDS image of prow x pld floats, zero
// outside the matrix.  A thread owns 32-column-strided elements of 8 rows per batch; the 8 global loads of a batch are
// unconditional (clamped addresses) so they are all in flight before the first LDS write waits.
__device__ __forceinline__ void stage_matrix(const float* __restrict__ src, int rows, int cols, int src_ld,
                                             float* __restrict__ dst, int prow, int pld) {
  const int c0 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  for (int rb = r0; rb < prow; rb += 64)
    for (int c = c0; c < pld; c += 32) {
      const int cc = c < cols ? c : cols - 1;
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + 8 * u;
        v[u] = src[(r < rows ? r : rows - 1) * src_ld + cc];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + 8 * u;
        if (r < prow) dst[r * pld + c] = (r < rows && c < cols) ? v[u] : 0.f;
      }
    }
}

template <typename T, int HT>
__global__ void __launch_bounds__(256) k_mixed_stack(MixArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int DT = 2 * HT;
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int D = a.D, h = D / 2;
  float* zt = sm + wave * (D * 32);   // this wave's tile: zt[f * 32 + token]
  float* wr = sm + 4 * D * 32;        // step matrices, staged by the block
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  const int64_t tok = tile0 + r32;
  const bool valid = tok < a.N;
  float m = 0.f;
  if (valid) {
    m = 1.f;
    if (a.mask) {
      const int64_t bn = tok / a.S;
      m = (float)a.mask[(bn / a.mask_div) * a.S + (tok - bn * a.S)];
    }
  }
  // the tile is 32 * D contiguous elements of z: linear, coalesced copy into the transposed LDS image
  const T* zin = reinterpret_cast<const T*>(a.z_in) + tile0 * D;
  {
    const int64_t left = (a.N - tile0) * D;                      // elements of z from the tile's start to the end
    const int nel = left <= 0 ? 0 : (left < 32 * D ? (int)left : 32 * D);
    for (int base = 0; base < 32 * D; base += 8 * 64) {         // D is even: 32 D is a multiple of 64
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + 64 * u + lane;
        float t1[1] = {0.f};
        if (nel > 0) RowIO<T, 1>::load(zin + (idx < nel ? idx : nel - 1), t1);   // wave-uniform branch
        v[u] = idx < nel ? t1[0] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + 64 * u + lane;
        if (idx < 32 * D) {
          const int tl = idx / D, f = idx - tl * D;
          zt[f * 32 + tl] = v[u];
        }
      }
    }
  }
  float ld = 0.f;

  for (int si = 0; si < a.K; ++si) {
    const MixStep& s = a.steps[si];
    if (s.kind == STEP_ACTNORM) {
      // actnorm.py:30-54: forward (z e^ls + b) m, backward ((z - b) m) / (e^ls + 1e-13)
      // lane l holds the parameters of features l and l + 64 (two coalesced loads), handed round with shuffles
      const float* __restrict__ ls = s.p[0];
      const float* __restrict__ bias = s.p[1];
      const int f0 = lane < D ? lane : D - 1, f1 = lane + 64 < D ? lane + 64 : D - 1;
      const float e0 = expf(ls[f0]), e1 = expf(ls[f1]), b0 = bias[f0], b1v = bias[f1];
      for (int f = hh; f < D; f += 2) {
        const float e = f < 64 ? __shfl(e0, f, 64) : __shfl(e1, f - 64, 64);
        const float b = f < 64 ? __shfl(b0, f, 64) : __shfl(b1v, f - 64, 64);
        const float v = zt[f * 32 + r32];
        zt[f * 32 + r32] = a.inverse ? ((v - b) * m) / (e + 1e-13f) : (v * e + b) * m;
      }
    } else if (s.kind == STEP_LINEAR) {
      // invertible_flows.py:29 / :61-63: every head's slice times M^T, no mask multiply.  The block-diagonal D x D
      // image of M is staged with an odd pitch; out^T = Mfull . z^T
      const int Dk = even_up(D), LD = Dk + 1, hs = s.hs;
      __syncthreads();
      {
        const float* __restrict__ Mh = s.p[0];
        const int c0 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
        for (int rb = r0; rb < 64 * HT; rb += 64)
          for (int c = c0; c < LD; c += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int r = rb + 8 * u;
              const int ri = r % hs, ci = c - (r - ri);              // column inside this row's head block
              const bool in = r < D && ci >= 0 && ci < hs;
              v[u] = Mh[in ? ri * hs + ci : 0];
              v[u] = in ? v[u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) wr[(rb + 8 * u) * LD + c] = v[u];
          }
      }
      __syncthreads();
      f32x16 acc[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 4
      for (int ks = 0; ks < Dk / 2; ++ks) {
        const int k = 2 * ks + hh;
        const float b = zt[k * 32 + r32];               // D is even: k < D
#pragma unroll
        for (int t = 0; t < DT; ++t) acc[t] = MmaF32::mma(wr[(32 * t + r32) * LD + k], b, acc[t]);
      }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = 32 * t + acc_row(r, hh);
          if (f < D) zt[f * 32 + r32] = acc[t][r];
        }
    } else {
      // couplings.py:141-183.  Masked tokens run on zeros (their outputs are multiplied by the mask anyway), which
      // keeps exp() finite at padding positions.
      const int H = s.H, Hp = (H + 31) / 32 * 32, HC = Hp / 32;
      const int hk = even_up(h), LD1 = hk + 1, LD2 = Hp + 1;
      float* w1 = wr;                       // [Hp][LD1]
      float* w2 = w1 + Hp * LD1;            // [32 HT][LD2]
      float* b1 = w2 + 32 * HT * LD2;       // [Hp]
      float* b2 = b1 + Hp;                  // [32 HT]
      f32x16 out[2][HT];
#pragma unroll
      for (int net = 0; net < 2; ++net) {
        const float* __restrict__ W1 = s.p[4 * net + 0];
        const float* __restrict__ B1 = s.p[4 * net + 1];
        const float* __restrict__ W2 = s.p[4 * net + 2];
        const float* __restrict__ B2 = s.p[4 * net + 3];
        __syncthreads();
        stage_matrix(W1, H, h, h, w1, Hp, LD1);
        stage_matrix(W2, h, H, H, w2, 32 * HT, LD2);
        for (int idx = threadIdx.x; idx < Hp; idx += 256) b1[idx] = idx < H ? B1[idx] : 0.f;
        for (int idx = threadIdx.x; idx < 32 * HT; idx += 256) b2[idx] = idx < h ? B2[idx] : 0.f;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) out[net][t][r] = b2[32 * t + acc_row(r, hh)];
        for (int c = 0; c < HC; ++c) {
          f32x16 pre;
#pragma unroll
          for (int r = 0; r < 16; ++r) pre[r] = b1[32 * c + acc_row(r, hh)];
#pragma unroll 4
          for (int ks = 0; ks < hk / 2; ++ks) {
            const int k = 2 * ks + hh;
            // for odd h the last k-step also reads row h (x1): its W1 column is staged as zero
            const float v = zt[k * 32 + r32];
            const float b = m != 0.f ? v : 0.f;
            pre = MmaF32::mma(w1[(32 * c + r32) * LD1 + k], b, pre);
          }
          act_tile<-1, false>(s.act, pre);
          // hidden unit acc_row(q, hh) of this chunk is k-step q of the second product: no data movement
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int k = 32 * c + acc_row(q, hh);
#pragma unroll
            for (int t = 0; t < HT; ++t) out[net][t] = MmaF32::mma(w2[(32 * t + r32) * LD2 + k], pre[q], out[net][t]);
          }
        }
      }
      // fold: forward x1 = (x1 e^s + t) m, inverse x1 = ((x1 - t) e^-s) m; x0 = x0 m; log-det += / -= sum s
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = 32 * t + acc_row(r, hh);
          if (f < h) {
            const float sv = out[0][t][r], tv = out[1][t][r];
            float x = m != 0.f ? zt[(h + f) * 32 + r32] : 0.f;
            if (!a.inverse) {
              x *= fast_exp(sv);
              x = (x + tv) * m;
              ld += sv;
            } else {
              x -= tv;
              x = (x * fast_exp(-sv)) * m;
              ld -= sv;
            }
            zt[(h + f) * 32 + r32] = x;
          }
        }
      for (int f = hh; f < h; f += 2) zt[f * 32 + r32] *= m;
    }
  }

  T* zout = reinterpret_cast<T*>(a.z_out) + tile0 * D;
  for (int idx = lane; idx < 32 * D; idx += 64) {
    const int tl = idx / D, f = idx - tl * D;
    if (tile0 + tl < a.N) {
      float v[1] = {zt[f * 32 + tl]};
      RowIO<T, 1>::store(zout + idx, v);
    }
  }
  float tv = ld * m;
  tv += __shfl_xor(tv, 32, 64);
  if (valid && hh == 0 && a.tok_val) a.tok_val[tok] = tv;
}

}  // namespace

// 1 when (D, largest coupling hidden width) fits the one-launch mixed stack
extern "C" int mivae_mixed_stack_supported(int64_t D, int64_t Hmax) {
  return D >= 2 && D <= kMaxD && D % 2 == 0 && Hmax <= kMaxH;
}

// One launch over a list of K flow steps, executed in the order given (the caller reverses the list and passes
// inverse = 1 for the sampling direction, and hands an inverted matrix to a linear step).
//   steps: HOST array, 12 int64 per step: kind (0 NICE-FF, 1 ActNorm, 2 linear), H, activation id, head size,
//          8 device pointers (fp32 parameters; see MixStep).  Read during the call, not kept.
//   z_in, z_out (Bn, S, D) latent dtype (may alias); mask (Bn / mask_div, S) int64 or null;
//   tok_val (Bn S) fp32: masked per-token sum of the NICE log-scales (negated for inverse = 1), or null.
extern "C" int mivae_mixed_stack(const void* z_in, void* z_out, const int64_t* mask, float* tok_val,
                                 const int64_t* steps, int64_t K, int64_t Bn, int64_t S, int64_t mask_div, int64_t D,
                                 int inverse, int dtype, void* stream) {
  MIVAE_REQUIRE(Bn >= 0 && S >= 1 && mask_div >= 1 && Bn * S < (1ll << 40), "mixed_stack: bad sizes");
  MIVAE_REQUIRE(K >= 1 && K <= kMaxSteps, "mixed_stack: 1..%d steps, got %lld", kMaxSteps, (long long)K);
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "mixed_stack: dtype must be f32 or bf16");
  MIVAE_REQUIRE(steps, "mixed_stack: null step table");
  if (D < 2 || D > kMaxD || (D & 1)) {
    mivae_set_error("mixed_stack: unsupported latent width D=%lld (even, 2..%d)", (long long)D, kMaxD);
    return MIVAE_EUNSUPPORTED;
  }
  const int64_t N = Bn * S;
  if (N == 0) return MIVAE_OK;
  MIVAE_REQUIRE(z_in && z_out, "mixed_stack: null latent pointer");
  MixArgs a;
  a.z_in = z_in; a.z_out = z_out; a.mask = mask; a.tok_val = tok_val;
  a.N = N; a.S = (int)S; a.mask_div = (int)mask_div; a.D = (int)D; a.K = (int)K; a.inverse = inverse ? 1 : 0;
  const int h = (int)D / 2, HT = (h + 31) / 32;
  int region = 0;
  for (int i = 0; i < K; ++i) {
    const int64_t* e = steps + 12 * i;
    MixStep& s = a.steps[i];
    s.kind = (int)e[0]; s.H = (int)e[1]; s.act = (int)e[2]; s.hs = (int)e[3];
    for (int j = 0; j < 8; ++j) s.p[j] = reinterpret_cast<const float*>(e[4 + j]);
    int need = 0, nptr = 0;
    if (s.kind == STEP_NICE) {
      if (s.H < 1 || s.H > kMaxH) {
        mivae_set_error("mixed_stack: step %d: coupling hidden width %d outside 1..%d", i, s.H, kMaxH);
        return MIVAE_EUNSUPPORTED;
      }
      MIVAE_REQUIRE(s.act >= ACT_LINEAR && s.act <= ACT_SIGMOID, "mixed_stack: step %d: bad activation id %d", i, s.act);
      need = nice_region(h, s.H, HT); nptr = 8;
    } else if (s.kind == STEP_ACTNORM) {
      nptr = 2;
    } else if (s.kind == STEP_LINEAR) {
      MIVAE_REQUIRE(s.hs >= 1 && D % s.hs == 0, "mixed_stack: step %d: head size %d does not divide D=%lld", i, s.hs,
                    (long long)D);
      need = linear_region((int)D, HT); nptr = 1;
    } else {
      MIVAE_REQUIRE(false, "mixed_stack: step %d: unknown kind %d", i, s.kind);
    }
    for (int j = 0; j < nptr; ++j) MIVAE_REQUIRE(s.p[j], "mixed_stack: step %d: null parameter pointer %d", i, j);
    if (need > region) region = need;
  }
  const size_t smem = (size_t)(4 * D * 32 + region) * sizeof(float);
  MIVAE_REQUIRE(smem <= 160 * 1024, "mixed_stack: %zu bytes of LDS needed", smem);
  const unsigned grid = (unsigned)((N + 127) / 128);
  hipStream_t st = (hipStream_t)stream;
#define MIX_LAUNCH(T, HTV)                                                                                      \
  do {                                                                                                          \
    if (hipFuncSetAttribute((const void*)k_mixed_stack<T, HTV>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                            160 * 1024) != hipSuccess) {                                                        \
      mivae_set_error("mixed_stack: cannot raise the dynamic LDS limit");                                       \
      return MIVAE_ELAUNCH;                                                                                     \
    }                                                                                                           \
    k_mixed_stack<T, HTV><<<grid, 256, smem, st>>>(a);                                                          \
  } while (0)
  if (dtype == MIVAE_F32) {
    if (HT == 1) MIX_LAUNCH(float, 1); else MIX_LAUNCH(float, 2);
  } else {
    if (HT == 1) MIX_LAUNCH(unsigned short, 1); else MIX_LAUNCH(unsigned short, 2);
  }
#undef MIX_LAUNCH
  MIVAE_CHECK_LAUNCH("mixed_stack");
  return MIVAE_OK;
}
