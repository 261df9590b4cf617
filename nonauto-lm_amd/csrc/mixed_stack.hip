// Mixed flow stack in ONE launch for narrow latents (D <= 128): any sequence of NICE feed-forward coupling steps,
// ActNorm steps and invertible-linear steps (single matrix or one matrix shared by all heads), in the density
// direction or the sampling direction (k_mixed_stack), and the TRAINING path of the density direction: the same launch
// also writes the input of every step (fp32 history), and ONE backward launch (k_mixed_bwd) walks the steps in
// reverse with the cotangent tile resident in LDS -- z-gradient chain of all three step kinds, the coupling nets'
// backward with the token-blocked side outputs that nice_stack.hip's split-K weight-gradient kernels consume, the
// ActNorm parameter sums as per-block partials (one fold launch for all ActNorm steps), and the output cotangent of
// every linear step for its d M product (mivae_rowgemm_wgrad).
//
// Reference semantics, per step: NICECoupling.forward / .backward (vae_lm/models/base/flows/couplings.py:141-183) with
// FeedForwardCoupling nets (:50-51), ActNorm.forward / .backward (flows/actnorm.py:30-54), InvertibleLinearFlow and
// InvertibleMultiHeadFlow (flows/invertible_flows.py:26-44, :58-83), iterated like FlowPrior.forward_pass /
// backward_pass (nonauto_lm/priors/flow_prior.py:37-70).  The parameter-only log-determinants (sum log_scale x tokens,
// slogdet x tokens x heads) are added by the caller; the kernel returns the data-dependent part (the NICE scales).
//
// Design: a wave64 owns 32 consecutive tokens for the whole step list; the tile lives transposed (feature x token,
// fp32, odd pitch) in the wave's LDS area, so z is read once and written once whatever the number of steps.  All products run
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
  int ord;             // ordinal among the steps of its kind (training path: section index of the side outputs)
  const float* p[8];   // NICE: W1s b1s W2s b2s W1t b1t W2t b2t | ActNorm: log_scale bias | linear: M (hs, hs)
};
struct MixArgs {
  const void* z_in;
  void* z_out;
  const int64_t* mask;
  float* tok_val;
  float* hist;         // optional (K, N, D) fp32: the input of every step, row-major (training path)
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
  const int nr = blockDim.x >> 5;                  // row groups of the block (8 for 256 threads)
  for (int rb = r0; rb < prow; rb += 8 * nr)
    for (int c = c0; c < pld; c += 32) {
      const int cc = c < cols ? c : cols - 1;
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + nr * u;
        v[u] = src[(r < rows ? r : rows - 1) * src_ld + cc];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + nr * u;
        if (r < prow) dst[r * pld + c] = (r < rows && c < cols) ? v[u] : 0.f;
      }
    }
}

// tile pitch (floats): odd, so that the row-major <-> transposed moves (z in / out, history) are free of bank conflicts
constexpr int TP = 33;

// Row-major (32 tokens x D, `nel` valid elements) <-> transposed LDS tile, 64 consecutive elements of the row-major image
// per wave instruction (coalesced).  The load keeps 8 global loads in flight before the first LDS write waits.
// (Measured and not kept: division-free index walks -- the integer divisions are not what these moves cost; un-batched
// loads are 30 % slower.)
template <typename T>
__device__ __forceinline__ void tile_load_t(const T* __restrict__ src, int nel, int D, float* __restrict__ tile, int lane) {
  for (int base = 0; base < 32 * D; base += 8 * 64) {         // D is even: 32 D is a multiple of 64
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + 64 * u + lane;
      float t1[1] = {0.f};
      if (nel > 0) RowIO<T, 1>::load(src + (idx < nel ? idx : nel - 1), t1);   // wave-uniform branch
      v[u] = idx < nel ? t1[0] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + 64 * u + lane;
      if (idx < 32 * D) {
        const int tl = idx / D, f = idx - tl * D;
        tile[f * TP + tl] = v[u];
      }
    }
  }
}
template <typename T>
__device__ __forceinline__ void tile_store_t(T* __restrict__ dst, int nel, int D, const float* __restrict__ tile, int lane) {
  for (int idx = lane; idx < nel; idx += 64) {
    const int tl = idx / D, f = idx - tl * D;
    float v[1] = {tile[f * TP + tl]};
    RowIO<T, 1>::store(dst + idx, v);
  }
}

template <typename T, int HT>
__global__ void __launch_bounds__(256) k_mixed_stack(MixArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int DT = 2 * HT;
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int D = a.D, h = D / 2;
  float* zt = sm + wave * (D * TP);   // this wave's tile: zt[f * TP + token]
  float* wr = sm + 4 * D * TP;        // step matrices, staged by the block
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
  const int64_t left = (a.N - tile0) * D;                        // elements of z from the tile's start to the end
  const int nel = left <= 0 ? 0 : (left < 32 * D ? (int)left : 32 * D);
  tile_load_t<T>(zin, nel, D, zt, lane);
  float ld = 0.f;

  for (int si = 0; si < a.K; ++si) {
    const MixStep& s = a.steps[si];
    if (a.hist)                                     // the step's input, for the backward launch
      tile_store_t<float>(a.hist + ((int64_t)si * a.N + tile0) * D, nel, D, zt, lane);
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
        const float v = zt[f * TP + r32];
        zt[f * TP + r32] = a.inverse ? ((v - b) * m) / (e + 1e-13f) : (v * e + b) * m;
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
        const float b = zt[k * TP + r32];               // D is even: k < D
#pragma unroll
        for (int t = 0; t < DT; ++t) acc[t] = MmaF32::mma(wr[(32 * t + r32) * LD + k], b, acc[t]);
      }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = 32 * t + acc_row(r, hh);
          if (f < D) zt[f * TP + r32] = acc[t][r];
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
            const float v = zt[k * TP + r32];
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
            float x = m != 0.f ? zt[(h + f) * TP + r32] : 0.f;
            if (!a.inverse) {
              x *= fast_exp(sv);
              x = (x + tv) * m;
              ld += sv;
            } else {
              x -= tv;
              x = (x * fast_exp(-sv)) * m;
              ld -= sv;
            }
            zt[(h + f) * TP + r32] = x;
          }
        }
      for (int f = hh; f < h; f += 2) zt[f * TP + r32] *= m;
    }
  }

  tile_store_t<T>(reinterpret_cast<T*>(a.z_out) + tile0 * D, nel, D, zt, lane);
  float tv = ld * m;
  tv += __shfl_xor(tv, 32, 64);
  if (valid && hh == 0 && a.tok_val) a.tok_val[tok] = tv;
}


// ---- training path: backward --------------------------------------------------------------------------------------

struct MixBwdArgs {
  const float* hist;     // (K, N, D) fp32: input of every step (k_mixed_stack with hist)
  const void* dz_out;    // (N, D) latent dtype: cotangent of z_K, or null (zero)
  const float* dval;     // cotangent of tok_val: one per sentence (Bn) or, dval_tok = 1, one per token (N); or null
  void* dz_in;           // (N, D) latent dtype
  const int64_t* mask;
  float* gcot;           // (nL, N, D) fp32: cotangent at the OUTPUT of every linear step (for d M), row-major
  float* an_part;        // (gridDim.x, nA, 2, D): per-block sums  [sum g m x e^ls | sum g m]  of every ActNorm step
  float *Gt, *Ht, *dSt, *x0t;   // fp32 token-blocked side outputs of the coupling steps (nice_common.hpp), by NICE ordinal
  int64_t N, Npad;
  int S, mask_div, D, K, dval_tok, nA, Hp, hp;
  MixStep steps[kMaxSteps];
};

// One backward launch over the step list (density direction), steps K-1 .. 0.  A wave owns 32 tokens: gt = cotangent
// tile, xt = the step's input tile (from the history), both transposed fp32 in LDS.
//   ActNorm (actnorm.py:30-39):  d z = g m e^ls;  d ls = sum g m z e^ls,  d b = sum g m  (per-block partials)
//   linear  (invertible_flows.py:29, :61-63):  d z = g M per head  (d M = g^T z is the caller's mivae_rowgemm_wgrad on
//     the stored cotangent)
//   NICE    (couplings.py:141-162):  s recomputed from x0;  d x1 = g1 m e^s,  d s = g1 m e^s x1 + d(log-det),
//     d t = g1 m,  then both nets backward (d hid = W2^T d out, d pre = d hid act', d x0 += W1^T d pre) with hid / d pre /
//     d s / d t / x0 emitted for the weight-gradient kernels.
template <typename T, int HT>
__global__ void __launch_bounds__(256) k_mixed_bwd(MixBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int DT = 2 * HT;
  constexpr int TB = 4;                            // fp32 side outputs: 4 tokens per 16-byte block
  const int W = blockDim.x >> 6;
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int D = a.D, h = D / 2;
  float* gt = sm + wave * (2 * D * TP);            // cotangent tile  gt[f * TP + token]
  float* xt = gt + D * TP;                         // input tile of the current step
  float* wr = sm + W * (2 * D * TP);               // step matrices / block reductions, shared by the block
  const int64_t tile0 = ((int64_t)blockIdx.x * W + wave) * 32;
  const int64_t tok = tile0 + r32;
  const bool valid = tok < a.N;
  const bool emit = tile0 < a.Npad;                // wave-uniform: Npad is a multiple of 64
  float m = 0.f, dl = 0.f;
  if (valid) {
    const int64_t bn = tok / a.S;
    m = 1.f;
    if (a.mask) m = (float)a.mask[(bn / a.mask_div) * a.S + (tok - bn * a.S)];
    if (a.dval) dl = a.dval[a.dval_tok ? tok : bn] * m;
  }
  const int64_t left = (a.N - tile0) * D;
  const int nel = left <= 0 ? 0 : (left < 32 * D ? (int)left : 32 * D);
  if (a.dz_out) tile_load_t<T>(reinterpret_cast<const T*>(a.dz_out) + tile0 * D, nel, D, gt, lane);
  else
    for (int f = hh; f < D; f += 2) gt[f * TP + r32] = 0.f;
  const unsigned tblk = (unsigned)(tok / TB), tin = (unsigned)(tok % TB);
  const unsigned loffH = tblk * (unsigned)(a.Hp * TB) + tin + 4 * hh * TB;
  const unsigned loffh = tblk * (unsigned)(a.hp * TB) + tin + 4 * hh * TB;

  for (int si = a.K - 1; si >= 0; --si) {
    const MixStep& s = a.steps[si];
    tile_load_t<float>(a.hist + ((int64_t)si * a.N + tile0) * D, nel, D, xt, lane);
    if (s.kind == STEP_ACTNORM) {
      const float* __restrict__ ls = s.p[0];
      // token lanes: xt <- g m x, gt <- g m
      for (int f = hh; f < D; f += 2) {
        const float gm = gt[f * TP + r32] * m;
        xt[f * TP + r32] *= gm;
        gt[f * TP + r32] = gm;
      }
      __syncthreads();                             // the previous step's matrices in wr have been consumed
      // feature lanes: lane l sums feature l (+ 64) over the 32 tokens (rotated start: two lanes per bank), scales gt
      for (int f = lane; f < D; f += 64) {
        const float e = expf(ls[f]);
        float a0 = 0.f, a1 = 0.f;
        for (int t = 0; t < 32; ++t) {
          const int tt = (t + lane) & 31;
          const float g = gt[f * TP + tt];
          a0 += xt[f * TP + tt];
          a1 += g;
          gt[f * TP + tt] = g * e;
        }
        wr[(wave * 2 + 0) * D + f] = a0 * e;
        wr[(wave * 2 + 1) * D + f] = a1;
      }
      __syncthreads();
      for (int idx = threadIdx.x; idx < 2 * D; idx += blockDim.x) {
        float sum = 0.f;
        for (int w = 0; w < W; ++w) sum += wr[w * 2 * D + idx];
        a.an_part[((int64_t)blockIdx.x * a.nA + s.ord) * 2 * D + idx] = sum;
      }
    } else if (s.kind == STEP_LINEAR) {
      // the cotangent at this step's output, row-major, for d M = g^T x
      tile_store_t<float>(a.gcot + ((int64_t)s.ord * a.N + tile0) * D, nel, D, gt, lane);
      // d x^T = Mfull^T . g^T: stage the transposed block-diagonal image  wr[c * LD + r] = Mfull[r][c]
      const int Dk = even_up(D), LD = Dk + 1, hs = s.hs;
      __syncthreads();
      {
        const float* __restrict__ Mh = s.p[0];
        const int c0 = threadIdx.x & 31, r0 = (int)(threadIdx.x >> 5);
        const int rstep = (int)(blockDim.x >> 5);
        for (int rb = r0; rb < LD; rb += rstep)            // rb = source row r (a column of the image)
          for (int c = c0; c < 64 * HT; c += 32) {         // c = source column (a row of the image)
            const int ri = rb % hs, ci = c - (rb - ri);
            const bool in = rb < D && c < D && ci >= 0 && ci < hs;
            const float v = Mh[in ? ri * hs + ci : 0];
            wr[c * LD + rb] = in ? v : 0.f;
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
        const float b = gt[k * TP + r32];               // D is even: k < D
#pragma unroll
        for (int t = 0; t < DT; ++t) acc[t] = MmaF32::mma(wr[(32 * t + r32) * LD + k], b, acc[t]);
      }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = 32 * t + acc_row(r, hh);
          if (f < D) gt[f * TP + r32] = acc[t][r];
        }
    } else {
      const int H = s.H, Hp = (H + 31) / 32 * 32, HC = Hp / 32;   // Hp == a.Hp (one hidden width per list)
      const int hk = even_up(h), LD1 = hk + 1, LD2 = Hp + 1;
      float* w1 = wr;                       // [Hp][LD1]
      float* w2 = w1 + Hp * LD1;            // [32 HT][LD2]
      float* b1 = w2 + 32 * HT * LD2;       // [Hp]
      float* b2 = b1 + Hp;                  // [32 HT]
      const int64_t ldn = a.Npad;
      float* dS = a.dSt + (int64_t)s.ord * 2 * a.hp * ldn;
      if (emit)                              // x0 (zeros at padding) for the dW1 products
        for (int f = hh; f < a.hp; f += 2)
          a.x0t[(int64_t)s.ord * a.hp * ldn + ((int64_t)tblk * a.hp + f) * TB + tin] = (f < h && m != 0.f) ? xt[f * TP + r32] : 0.f;
      f32x16 dx0[HT], dtv[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dx0[t][r] = 0.f;
#pragma unroll 1
      for (int net = 0; net < 2; ++net) {
        const float* __restrict__ W1 = s.p[4 * net + 0];
        const float* __restrict__ B1 = s.p[4 * net + 1];
        const float* __restrict__ W2 = s.p[4 * net + 2];
        const float* __restrict__ B2 = s.p[4 * net + 3];
        __syncthreads();
        stage_matrix(W1, H, h, h, w1, Hp, LD1);
        stage_matrix(W2, h, H, H, w2, 32 * HT, LD2);
        for (int idx = threadIdx.x; idx < Hp; idx += blockDim.x) b1[idx] = idx < H ? B1[idx] : 0.f;
        for (int idx = threadIdx.x; idx < 32 * HT; idx += blockDim.x) b2[idx] = idx < h ? B2[idx] : 0.f;
        __syncthreads();
        // first layer of one hidden chunk: pre = b1[c] + W1[c] . x0^T (masked tokens run on zeros, like the forward)
        auto layer1 = [&](int c) {
          f32x16 pre;
#pragma unroll
          for (int r = 0; r < 16; ++r) pre[r] = b1[32 * c + acc_row(r, hh)];
#pragma unroll 4
          for (int ks = 0; ks < hk / 2; ++ks) {
            const int k = 2 * ks + hh;
            const float v = xt[k * TP + r32];             // odd h: row h is read against a zero column of W1
            pre = MmaF32::mma(w1[(32 * c + r32) * LD1 + k], m != 0.f ? v : 0.f, pre);
          }
          return pre;
        };
        if (net == 0) {
          // s = W2 act(pre) + b2, then the affine tail in reverse
          f32x16 sv[HT];
#pragma unroll
          for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[t][r] = b2[32 * t + acc_row(r, hh)];
#pragma unroll 1
          for (int c = 0; c < HC; ++c) {
            f32x16 hid = layer1(c);
            act_tile<-1, false>(s.act, hid);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const int k = 32 * c + acc_row(q, hh);
#pragma unroll
              for (int t = 0; t < HT; ++t) sv[t] = MmaF32::mma(w2[(32 * t + r32) * LD2 + k], hid[q], sv[t]);
            }
          }
#pragma unroll
          for (int t = 0; t < HT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int f = 32 * t + acc_row(r, hh);
              float ds = 0.f, dt = 0.f;
              if (f < h) {
                const float g1 = gt[(h + f) * TP + r32] * m;
                const float ge = g1 * fast_exp(sv[t][r]);
                const float x1 = m != 0.f ? xt[(h + f) * TP + r32] : 0.f;
                ds = ge * x1 + dl;
                dt = g1;
                gt[(h + f) * TP + r32] = ge;
                xt[(h + f) * TP + r32] = ds;               // B operand of d hid = W2^T d s
              }
              sv[t][r] = ds;
              dtv[t][r] = dt;
            }
            if (emit) {
              store_tile_t<float>(dS, 32 * t, loffh, sv[t]);
              store_tile_t<float>(dS + (int64_t)a.hp * ldn, 32 * t, loffh, dtv[t]);
            }
          }
        } else {
#pragma unroll
          for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int f = 32 * t + acc_row(r, hh);
              if (f < h) xt[(h + f) * TP + r32] = dtv[t][r];
            }
        }
        // this net's backward, chunk by chunk (the s-net's first layer is computed a second time: keeping four
        // pre-activation tiles across the affine tail spilled registers)
        float* Gt = a.Gt + (int64_t)(s.ord * 2 + net) * a.Hp * ldn;
        float* Ht = a.Ht + (int64_t)(s.ord * 2 + net) * a.Hp * ldn;
#pragma unroll 1
        for (int c = 0; c < HC; ++c) {
          f32x16 dh;
#pragma unroll
          for (int r = 0; r < 16; ++r) dh[r] = 0.f;
#pragma unroll 4
          for (int ks = 0; ks < hk / 2; ++ks) {
            const int k = 2 * ks + hh;                      // out feature (contraction index)
            const float b = k < h ? xt[(h + k) * TP + r32] : 0.f;
            dh = MmaF32::mma(w2[k * LD2 + 32 * c + r32], b, dh);
          }
          f32x16 hid = layer1(c), dact;
          act_tile_fd<-1, false>(s.act, hid, dact);
#pragma unroll
          for (int r = 0; r < 16; ++r) dh[r] *= dact[r];
          if (emit) {
            store_tile_t<float>(Ht, 32 * c, loffH, hid);
            store_tile_t<float>(Gt, 32 * c, loffH, dh);
          }
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int k = 32 * c + acc_row(q, hh);
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              const int col = 32 * t + r32;
              dx0[t] = MmaF32::mma(w1[k * LD1 + (col < hk ? col : hk)], dh[q], dx0[t]);
            }
          }
        }
      }
      // x0 passes through (y0 = x0 m) and feeds both nets
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = 32 * t + acc_row(r, hh);
          if (f < h) gt[f * TP + r32] = gt[f * TP + r32] * m + dx0[t][r];
        }
    }
  }
  tile_store_t<T>(reinterpret_cast<T*>(a.dz_in) + tile0 * D, nel, D, gt, lane);
}

// block shape of the backward launch: as many 32-token waves (4, 2, 1) as the LDS holds next to the step matrices
static int bwd_waves(int D, int region) {
  for (int W = 4; W >= 1; W >>= 1)
    if ((size_t)(W * 2 * D * TP + region) * sizeof(float) <= 160 * 1024) return W;
  return 0;
}
static int bwd_region(int D, int Hmax, bool any_linear) {
  const int h = D / 2, HT = (h + 31) / 32;
  int region = 4 * 2 * D;                           // ActNorm block reduction
  if (Hmax > 0) region = max(region, nice_region(h, Hmax, HT));
  if (any_linear) region = max(region, 64 * HT * (even_up(D) + 1));
  return region;
}

}  // namespace

// 1 when (D, largest coupling hidden width) fits the one-launch mixed stack
extern "C" int mivae_mixed_stack_supported(int64_t D, int64_t Hmax) {
  return D >= 2 && D <= kMaxD && D % 2 == 0 && Hmax <= kMaxH;
}

// host step table -> MixStep[]; counts per kind, widest coupling hidden layer, LDS floats of the widest staging region
struct StepSummary { int nN, nA, nL, Hmax, Hmin, region; };
static int parse_steps(const char* who, const int64_t* steps, int64_t K, int64_t D, MixStep* out, StepSummary& sum) {
  MIVAE_REQUIRE(K >= 1 && K <= kMaxSteps, "%s: 1..%d steps, got %lld", who, kMaxSteps, (long long)K);
  MIVAE_REQUIRE(steps, "%s: null step table", who);
  if (D < 2 || D > kMaxD || (D & 1)) {
    mivae_set_error("%s: unsupported latent width D=%lld (even, 2..%d)", who, (long long)D, kMaxD);
    return MIVAE_EUNSUPPORTED;
  }
  const int h = (int)D / 2, HT = (h + 31) / 32;
  sum = {0, 0, 0, 0, 1 << 30, 0};
  for (int i = 0; i < K; ++i) {
    const int64_t* e = steps + 12 * i;
    MixStep& s = out[i];
    s.kind = (int)e[0]; s.H = (int)e[1]; s.act = (int)e[2]; s.hs = (int)e[3];
    for (int j = 0; j < 8; ++j) s.p[j] = reinterpret_cast<const float*>(e[4 + j]);
    int need = 0, nptr = 0;
    if (s.kind == STEP_NICE) {
      if (s.H < 1 || s.H > kMaxH) {
        mivae_set_error("%s: step %d: coupling hidden width %d outside 1..%d", who, i, s.H, kMaxH);
        return MIVAE_EUNSUPPORTED;
      }
      MIVAE_REQUIRE(s.act >= ACT_LINEAR && s.act <= ACT_SIGMOID, "%s: step %d: bad activation id %d", who, i, s.act);
      need = nice_region(h, s.H, HT); nptr = 8;
      s.ord = sum.nN++;
      if (s.H > sum.Hmax) sum.Hmax = s.H;
      if (s.H < sum.Hmin) sum.Hmin = s.H;
    } else if (s.kind == STEP_ACTNORM) {
      nptr = 2;
      s.ord = sum.nA++;
    } else if (s.kind == STEP_LINEAR) {
      MIVAE_REQUIRE(s.hs >= 1 && D % s.hs == 0, "%s: step %d: head size %d does not divide D=%lld", who, i, s.hs,
                    (long long)D);
      need = linear_region((int)D, HT); nptr = 1;
      s.ord = sum.nL++;
    } else {
      MIVAE_REQUIRE(false, "%s: step %d: unknown kind %d", who, i, s.kind);
    }
    for (int j = 0; j < nptr; ++j) MIVAE_REQUIRE(s.p[j], "%s: step %d: null parameter pointer %d", who, i, j);
    if (need > sum.region) sum.region = need;
  }
  return MIVAE_OK;
}

static int mixed_fwd_impl(const char* who, const void* z_in, void* z_out, const int64_t* mask, float* tok_val, float* hist,
                          const int64_t* steps, int64_t K, int64_t Bn, int64_t S, int64_t mask_div, int64_t D,
                          int inverse, int dtype, void* stream) {
  MIVAE_REQUIRE(Bn >= 0 && S >= 1 && mask_div >= 1 && Bn * S < (1ll << 40), "%s: bad sizes", who);
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "%s: dtype must be f32 or bf16", who);
  MixArgs a;
  StepSummary sum;
  int rc = parse_steps(who, steps, K, D, a.steps, sum);
  if (rc) return rc;
  const int64_t N = Bn * S;
  if (N == 0) return MIVAE_OK;
  MIVAE_REQUIRE(z_in && z_out, "%s: null latent pointer", who);
  a.z_in = z_in; a.z_out = z_out; a.mask = mask; a.tok_val = tok_val; a.hist = hist;
  a.N = N; a.S = (int)S; a.mask_div = (int)mask_div; a.D = (int)D; a.K = (int)K; a.inverse = inverse ? 1 : 0;
  const int h = (int)D / 2, HT = (h + 31) / 32;
  const size_t smem = (size_t)(4 * D * TP + sum.region) * sizeof(float);
  MIVAE_REQUIRE(smem <= 160 * 1024, "%s: %zu bytes of LDS needed", who, smem);
  const unsigned grid = (unsigned)((N + 127) / 128);
  hipStream_t st = (hipStream_t)stream;
#define MIX_LAUNCH(T, HTV)                                                                                      \
  do {                                                                                                          \
    if (hipFuncSetAttribute((const void*)k_mixed_stack<T, HTV>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                            160 * 1024) != hipSuccess) {                                                        \
      mivae_set_error("%s: cannot raise the dynamic LDS limit", who);                                           \
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
  MIVAE_CHECK_LAUNCH(who);
  return MIVAE_OK;
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
  return mixed_fwd_impl("mixed_stack", z_in, z_out, mask, tok_val, nullptr, steps, K, Bn, S, mask_div, D, inverse, dtype,
                        stream);
}

// The density direction for training: the same launch, which also writes the input of every step to
// hist (K, Bn S, D) fp32 for mivae_mixed_stack_bwd.
extern "C" int mivae_mixed_stack_train_fwd(const void* z_in, void* z_out, const int64_t* mask, float* tok_val, float* hist,
                                           const int64_t* steps, int64_t K, int64_t Bn, int64_t S, int64_t mask_div,
                                           int64_t D, int dtype, void* stream) {
  MIVAE_REQUIRE(hist, "mixed_stack_train_fwd: null history pointer");
  return mixed_fwd_impl("mixed_stack_train_fwd", z_in, z_out, mask, tok_val, hist, steps, K, Bn, S, mask_div, D, 0, dtype,
                        stream);
}

// Blocks of the backward launch (= rows of its ActNorm partial-sum output) for N tokens, or 0 when the list does not
// fit the training path: D even in 2..128, every coupling hidden width the same H <= 128 (Hmax == Hmin).
extern "C" int64_t mivae_mixed_stack_bwd_blocks(int64_t N, int64_t D, int64_t Hmax, int64_t Hmin, int any_linear) {
  if (N < 1 || D < 2 || D > kMaxD || (D & 1) || Hmax > kMaxH || Hmax != Hmin) return 0;
  const int W = bwd_waves((int)D, bwd_region((int)D, (int)Hmax, any_linear != 0));
  if (W == 0) return 0;
  return (N + 32 * W - 1) / (32 * W);
}

// Backward of mivae_mixed_stack_train_fwd in ONE launch (k_mixed_bwd).
//   hist (K, N, D) fp32 from the forward; dz_out (N, D) latent dtype or null; dval: cotangent of tok_val, one per
//   sentence (Bn) or, dval_per_token = 1, one per token; either may be null (zero).
//   dz_in (N, D) latent dtype.  gcot (nL, N, D) fp32: cotangent at every linear step's output (d M = g^T x through
//   mivae_rowgemm_wgrad on (hist[k], gcot[i])); an_part (blocks, nA, 2, D) fp32 with blocks =
//   mivae_mixed_stack_bwd_blocks: sum over the block axis (mivae_colsum) = [d log_scale | d bias] of every ActNorm step
//   (z path only: the log-det term is the caller's); Gt, Ht (nN, 2, Hp, Npad), dSt (nN, 2, hp, Npad), x0t (nN, hp, Npad)
//   fp32 token-blocked with Npad = mivae_nice_stack_npad(N): operands of mivae_nice_dw_steps.
extern "C" int mivae_mixed_stack_bwd(const float* hist, const void* dz_out, const float* dval, int dval_per_token,
                                     const int64_t* mask, const int64_t* steps, int64_t K, void* dz_in, float* gcot,
                                     float* an_part, float* Gt, float* Ht, float* dSt, float* x0t, int64_t Bn, int64_t S,
                                     int64_t mask_div, int64_t D, int dtype, void* stream) {
  MIVAE_REQUIRE(Bn >= 1 && S >= 1 && mask_div >= 1, "mixed_stack_bwd: bad sizes");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "mixed_stack_bwd: dtype must be f32 or bf16");
  MixBwdArgs a;
  StepSummary sum;
  int rc = parse_steps("mixed_stack_bwd", steps, K, D, a.steps, sum);
  if (rc) return rc;
  const int64_t N = Bn * S;
  if (sum.nN > 0 && sum.Hmax != sum.Hmin) {
    mivae_set_error("mixed_stack_bwd: coupling steps of different hidden widths (%d .. %d)", sum.Hmin, sum.Hmax);
    return MIVAE_EUNSUPPORTED;
  }
  MIVAE_REQUIRE(hist && dz_in, "mixed_stack_bwd: null pointer");
  MIVAE_REQUIRE(sum.nL == 0 || gcot, "mixed_stack_bwd: linear steps need gcot");
  MIVAE_REQUIRE(sum.nA == 0 || an_part, "mixed_stack_bwd: ActNorm steps need an_part");
  MIVAE_REQUIRE(sum.nN == 0 || (Gt && Ht && dSt && x0t), "mixed_stack_bwd: coupling steps need the side-output buffers");
  const int h = (int)D / 2, HT = (h + 31) / 32;
  const int region = bwd_region((int)D, sum.Hmax, sum.nL > 0);
  const int W = bwd_waves((int)D, region);
  MIVAE_REQUIRE(W > 0, "mixed_stack_bwd: the step matrices do not fit the LDS");
  a.hist = hist; a.dz_out = dz_out; a.dval = dval; a.dz_in = dz_in; a.mask = mask;
  a.gcot = gcot; a.an_part = an_part; a.Gt = Gt; a.Ht = Ht; a.dSt = dSt; a.x0t = x0t;
  a.N = N; a.Npad = (N + 63) / 64 * 64;
  a.S = (int)S; a.mask_div = (int)mask_div; a.D = (int)D; a.K = (int)K; a.dval_tok = dval_per_token ? 1 : 0;
  a.nA = sum.nA; a.Hp = (sum.Hmax + 31) / 32 * 32; a.hp = 32 * HT;
  MIVAE_REQUIRE(a.Npad * (int64_t)(a.Hp > a.hp ? a.Hp : a.hp) < (1ll << 31), "mixed_stack_bwd: too many tokens");
  const size_t smem = (size_t)(W * 2 * D * TP + region) * sizeof(float);
  const unsigned grid = (unsigned)((N + 32 * W - 1) / (32 * W));
  hipStream_t st = (hipStream_t)stream;
#define MIXB_LAUNCH(T, HTV)                                                                                     \
  do {                                                                                                          \
    if (hipFuncSetAttribute((const void*)k_mixed_bwd<T, HTV>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                            160 * 1024) != hipSuccess) {                                                        \
      mivae_set_error("mixed_stack_bwd: cannot raise the dynamic LDS limit");                                   \
      return MIVAE_ELAUNCH;                                                                                     \
    }                                                                                                           \
    k_mixed_bwd<T, HTV><<<grid, 64 * W, smem, st>>>(a);                                                         \
  } while (0)
  if (dtype == MIVAE_F32) {
    if (HT == 1) MIXB_LAUNCH(float, 1); else MIXB_LAUNCH(float, 2);
  } else {
    if (HT == 1) MIXB_LAUNCH(unsigned short, 1); else MIXB_LAUNCH(unsigned short, 2);
  }
#undef MIXB_LAUNCH
  MIVAE_CHECK_LAUNCH("mixed_stack_bwd");
  return MIVAE_OK;
}
