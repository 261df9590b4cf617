// Masked autoregressive flows: MAFlow / IAFlow over a MADE network
// (vae_lm/models/base/flows/maf.py:29-62, iaf.py:21-23, made.py:39-40,124-127).
//
// Density direction (MAFlow.forward, one MADE pass): the masked MLP is a dense contraction and stays a library
// GEMM; k_maf_affine_fwd/bwd are its HBM-bound tail  out = flip(z * exp(s) + t) * mask,  tok_ld = mask * sum_d s
// reading the (rows, 2D) MADE output in place (no split copies): 4 * rows * D * e bytes per launch.
//
// Sequential direction (MAFlow.backward = sampling; IAFlow's forward): the reference runs D full MADE passes and
// keeps one column of each (maf.py:54-58), ~10 D launches.  For a one-hidden-layer MADE, pass i differs from pass
// i-1 only by x_{i-1} entering the first layer, so k_maf_seq keeps the hidden pre-activations of a token in
// registers, updates them with one rank-1 step per feature and evaluates only output rows i and D+i:
//   pre = b1;  for i in 0..D-1:  h = act(pre);  s = b2[i] + h . W2[i];  t = b2[D+i] + h . W2[D+i];
//                                x_i = (z_i - t) * exp(-s);  ld += s;  pre += x_i * W1m[:, i]
// -- the same arithmetic as the reference's loop (features visited in natural order whatever the MADE ordering
// is, log-det accumulated with +s), one launch, z read once and x written once.  L lanes share a token (16 hidden
// units each), weights are staged once per block in LDS, z / x tiles go through LDS so that HBM accesses are
// full rows.  VALU-bound: D * Hp * (activation + 3 fma) per token.
#include "common.hpp"

using namespace mivae;

namespace {

constexpr int kBlock = 256;
constexpr int kHPL = 16;          // hidden units per lane in k_maf_seq

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_maf_affine_fwd(const T* __restrict__ z, const T* __restrict__ st,
                                                            const int64_t* __restrict__ mask, T* __restrict__ out,
                                                            float* __restrict__ tok_ld, int64_t rows, int D, int L,
                                                            int parity) {
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t row = tid / L;
  const int lig = (int)(tid % L);
  float acc = 0.f;
  if (row < rows) {
    const float m = mask ? (float)mask[row] : 1.f;
    for (int c = lig * VEC; c < D; c += L * VEC) {
      float x[VEC], vs[VEC], vt[VEC], y[VEC];
      RowIO<T, VEC>::load(z + row * D + c, x);
      RowIO<T, VEC>::load(st + row * 2 * D + c, vs);
      RowIO<T, VEC>::load(st + row * 2 * D + D + c, vt);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float v = (x[i] * expf(vs[i]) + vt[i]) * m;
        acc += vs[i];
        if (parity) y[VEC - 1 - i] = v; else y[i] = v;
      }
      RowIO<T, VEC>::store(out + row * D + (parity ? D - VEC - c : c), y);
    }
    acc *= m;
  }
  acc = group_sum(acc, L);
  if (row < rows && lig == 0) tok_ld[row] = acc;
}

// dz = g e^s (direct term; the MADE term comes back through d st), d s = g e^s z + dtok, d t = g,  g = dout * mask
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_maf_affine_bwd(const T* __restrict__ dout, const float* __restrict__ dtok,
                                                            const T* __restrict__ z, const T* __restrict__ st,
                                                            const int64_t* __restrict__ mask, T* __restrict__ dz,
                                                            T* __restrict__ dst, int64_t rows, int D, int L,
                                                            int parity) {
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t row = tid / L;
  const int lig = (int)(tid % L);
  if (row >= rows) return;
  const float m = mask ? (float)mask[row] : 1.f;
  const float gl = dtok ? dtok[row] * m : 0.f;
  for (int c = lig * VEC; c < D; c += L * VEC) {
    float gr[VEC], x[VEC], vs[VEC], oz[VEC], os[VEC], ot[VEC];
    if (dout) {
      RowIO<T, VEC>::load(dout + row * D + (parity ? D - VEC - c : c), gr);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) gr[i] = 0.f;
    }
    RowIO<T, VEC>::load(z + row * D + c, x);
    RowIO<T, VEC>::load(st + row * 2 * D + c, vs);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float g = (parity ? gr[VEC - 1 - i] : gr[i]) * m;
      const float e = expf(vs[i]);
      oz[i] = g * e;
      os[i] = g * e * x[i] + gl;
      ot[i] = g;
    }
    RowIO<T, VEC>::store(dz + row * D + c, oz);
    RowIO<T, VEC>::store(dst + row * 2 * D + c, os);
    RowIO<T, VEC>::store(dst + row * 2 * D + D + c, ot);
  }
}

// activation ids follow ops.ACT_IDS (+ 6 = elu, made.py's default)
template <int ACT>
__device__ __forceinline__ float act_eval(float x) {
  if (ACT == 0) return x;
  if (ACT == 1) return x * 0.5f * (1.0f + erff(x * kInvSqrt2));
  if (ACT == 2) return x / (1.0f + expf(-x));
  if (ACT == 3) return fmaxf(x, 0.f);
  if (ACT == 4) return tanhf(x);
  if (ACT == 5) return 1.0f / (1.0f + expf(-x));
  return x > 0.f ? x : expm1f(x);
}

struct SeqArgs {
  const void* z;
  const float *w1t, *b1, *w2, *b2;   // w1t [D][Hp] (masked, transposed), b1 [Hp], w2 [2D][Hp] (masked), b2 [2D]
  const int64_t* mask;
  void* out;
  float* tok_ld;
  int64_t rows;
  int D, Hp, L, parity;
};

template <typename T, int ACT>
__global__ __launch_bounds__(kBlock) void k_maf_seq(SeqArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D, Hp = a.Hp, L = a.L;
  const int tpb = kBlock / L;                 // tokens per block
  const int zs = D + 1;                       // padded tile stride: lanes of different tokens hit different banks
  float* w1 = smem;                           // [D][Hp]
  float* w2 = w1 + D * Hp;                    // [2D][Hp]
  float* zt = w2 + 2 * D * Hp;                // [tpb][D+1]: flip_if_parity(z) on load, x_i replaces z'_i in place
  const T* zg = static_cast<const T*>(a.z);
  T* og = static_cast<T*>(a.out);
  const int64_t row0 = (int64_t)blockIdx.x * tpb;

  for (int i = threadIdx.x * 4; i < D * Hp; i += kBlock * 4)
    *reinterpret_cast<f32x4*>(w1 + i) = *reinterpret_cast<const f32x4*>(a.w1t + i);
  for (int i = threadIdx.x * 4; i < 2 * D * Hp; i += kBlock * 4)
    *reinterpret_cast<f32x4*>(w2 + i) = *reinterpret_cast<const f32x4*>(a.w2 + i);
  for (int i = threadIdx.x; i < tpb * D; i += kBlock) {      // consecutive lanes read consecutive elements
    const int tk = i / D, c = i - tk * D;
    const int64_t r = row0 + tk;
    float v = 0.f;
    if (r < a.rows) { float t1[1]; RowIO<T, 1>::load(zg + r * D + c, t1); v = t1[0]; }
    zt[tk * zs + (a.parity ? D - 1 - c : c)] = v;
  }
  __syncthreads();

  const int tk = threadIdx.x / L, l = threadIdx.x % L;
  const int hoff = l * kHPL;
  float pre[kHPL];
#pragma unroll
  for (int j = 0; j < kHPL; ++j) pre[j] = a.b1[hoff + j];
  float ld = 0.f;
  for (int i = 0; i < D; ++i) {
    const float* ws = w2 + i * Hp + hoff;
    const float* wt = w2 + (D + i) * Hp + hoff;
    float sp = 0.f, tp = 0.f;
#pragma unroll
    for (int q = 0; q < kHPL / 4; ++q) {
      const f32x4 vs = *reinterpret_cast<const f32x4*>(ws + 4 * q);
      const f32x4 vt = *reinterpret_cast<const f32x4*>(wt + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float h = act_eval<ACT>(pre[4 * q + e]);
        sp = fmaf(h, vs[e], sp);
        tp = fmaf(h, vt[e], tp);
      }
    }
    sp = group_sum(sp, L);
    tp = group_sum(tp, L);
    const float s = sp + a.b2[i], t = tp + a.b2[D + i];
    const float zi = zt[tk * zs + i];           // every lane of the wave reads before lane 0 of the group writes
    const float xi = (zi - t) * expf(-s);
    ld += s;
    if (l == 0) zt[tk * zs + i] = xi;
    const float* wc = w1 + i * Hp + hoff;
#pragma unroll
    for (int q = 0; q < kHPL / 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(wc + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) pre[4 * q + e] = fmaf(xi, v[e], pre[4 * q + e]);
    }
  }
  {
    const int64_t r = row0 + tk;
    if (l == 0 && r < a.rows) a.tok_ld[r] = a.mask ? ld * (float)a.mask[r] : ld;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < tpb * D; i += kBlock) {
    const int t2 = i / D, c = i - t2 * D;
    const int64_t r = row0 + t2;
    if (r < a.rows) {
      const float m = a.mask ? (float)a.mask[r] : 1.f;
      float o[1] = {zt[t2 * zs + c] * m};
      RowIO<T, 1>::store(og + r * D + c, o);
    }
  }
}

static size_t seq_lds_bytes(int D, int Hp) {
  const int L = Hp / kHPL;
  const int tpb = kBlock / L;
  return ((size_t)3 * D * Hp + (size_t)tpb * (D + 1)) * sizeof(float);
}

static bool seq_supported(int64_t D, int64_t Hp) {
  if (D < 1 || D > 256 || Hp < kHPL || Hp > 256 || Hp % kHPL) return false;
  const int L = (int)(Hp / kHPL);
  if (L & (L - 1)) return false;
  return seq_lds_bytes((int)D, (int)Hp) <= 144 * 1024;
}

template <typename T>
int launch_seq(const SeqArgs& a, int act, size_t lds, unsigned grid, hipStream_t st) {
#define MAF_SEQ_CASE(A)                                                                               \
  case A: {                                                                                           \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_maf_seq<T, A>),               \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
    if (e != hipSuccess) { mivae_set_error("maf_sequential: LDS opt-in failed: %s", hipGetErrorString(e)); \
                           return MIVAE_ELAUNCH; }                                                    \
    k_maf_seq<T, A><<<grid, kBlock, lds, st>>>(a);                                                    \
    break;                                                                                            \
  }
  switch (act) {
    MAF_SEQ_CASE(0) MAF_SEQ_CASE(1) MAF_SEQ_CASE(2) MAF_SEQ_CASE(3) MAF_SEQ_CASE(4) MAF_SEQ_CASE(5) MAF_SEQ_CASE(6)
    default: mivae_set_error("maf_sequential: unknown activation %d", act); return MIVAE_EINVAL;
  }
#undef MAF_SEQ_CASE
  return MIVAE_OK;
}

// ---- backward of the sequential direction (IAFlow's training direction / gradients through MAFlow.backward) ---------
// Autograd of the recurrence of k_maf_seq for one token, walked in REVERSE without any saved intermediate:
//   pre_D = b1 + sum_i x_i W1m[:, i]                      (a plain product of the saved outputs x: no sequential chain)
//   for i = D-1 .. 0:   pre_i = pre_{i+1} - x_i W1m[:, i];  h = act(pre_i);  s_i = b2[i] + h . W2[i]
//                       dx_i = g_i + A . W1m[:, i]          (A = d loss / d pre_{i+1}: the contributions of all later steps)
//                       dz'_i = dx_i e^{-s_i};  dt_i = -dz'_i;  ds_i = -dx_i x_i + g_ld
//                       dW1m[:, i] += x_i A;  dW2[i] += ds_i h;  dW2[D+i] += dt_i h;  db2[i] += ds_i;  db2[D+i] += dt_i
//                       A += act'(pre_i) (ds_i W2[i] + dt_i W2[D+i])
//   db1 += A
// One wave per block walks tokens one at a time: lane l owns hidden units l, l + 64, ... (HQ per lane), so the per-unit
// state (pre, A) lives in registers, the two dot products per step are wave reductions, and the parameter gradients
// accumulate in LDS columns that only their own lane touches (no atomics, no conflicts).  Padding tokens are skipped
// (wave-uniform).  Every block leaves one image [dW1t | dW2 | db1 | db2]; the images are summed in block order by
// mivae_colsum (fixed order: run-to-run reproducible).  VALU-bound, ~the forward's arithmetic x 3.
template <int ACT>
__device__ __forceinline__ void act_fd(float x, float& f, float& d) {
  if (ACT == 0) { f = x; d = 1.f; }
  else if (ACT == 1) {
    const float cdf = 0.5f * (1.0f + erff(x * kInvSqrt2));
    f = x * cdf; d = cdf + x * 0.3989422804014327f * expf(-0.5f * x * x);
  } else if (ACT == 2) { const float sg = 1.0f / (1.0f + expf(-x)); f = x * sg; d = sg * (1.0f + x * (1.0f - sg)); }
  else if (ACT == 3) { f = fmaxf(x, 0.f); d = x > 0.f ? 1.f : 0.f; }
  else if (ACT == 4) { f = tanhf(x); d = 1.0f - f * f; }
  else if (ACT == 5) { f = 1.0f / (1.0f + expf(-x)); d = f * (1.0f - f); }
  else { const float em = expm1f(x); f = x > 0.f ? x : em; d = x > 0.f ? 1.f : em + 1.0f; }
}

struct SeqBwdArgs {
  const void* x;        // (rows, D): the forward's output
  const void* dout;     // (rows, D) or null
  const float* dtok;    // (rows) or null: cotangent of tok_ld
  const int64_t* mask;  // (rows) or null
  const float *w1t, *b1, *w2, *b2;
  void* dz;             // (rows, D)
  float* part;          // [gridDim.x][3 D Hp + Hp + 2 D]
  int64_t rows;
  int D, Hp, parity;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <typename T, int ACT, int HQ>
__global__ __launch_bounds__(64) void k_maf_seq_bwd(SeqBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D, Hp = a.Hp, lane = threadIdx.x;
  float* gW1 = smem;                  // [D][Hp]
  float* gW2 = gW1 + D * Hp;          // [2D][Hp]
  float* gb2 = gW2 + 2 * D * Hp;      // [2D]
  float* xr = gb2 + 2 * D;            // [D]   x of the current token
  float* gr = xr + D;                 // [D]   d loss / d out of the current token
  float* zr = gr + D;                 // [D]   d z of the current token (already in z's feature order)
  for (int i = lane; i < 3 * D * Hp + 2 * D; i += 64) smem[i] = 0.f;
  int jq[HQ];
  bool jon[HQ];
  float gb1[HQ], b1v[HQ];
#pragma unroll
  for (int q = 0; q < HQ; ++q) {
    jq[q] = q * 64 + lane; jon[q] = jq[q] < Hp;
    if (!jon[q]) jq[q] = 0;           // idle lanes (Hp < 64) read unit 0 and contribute nothing
    gb1[q] = 0.f; b1v[q] = a.b1[jq[q]];
  }
  const T* X = static_cast<const T*>(a.x);
  const T* G = static_cast<const T*>(a.dout);
  T* DZ = static_cast<T*>(a.dz);
  __syncthreads();
  for (int64_t r = blockIdx.x; r < a.rows; r += gridDim.x) {
    const bool live = !a.mask || a.mask[r] != 0;
    if (!live) {                      // out = x * 0, tok_ld = ld * 0: no gradient reaches this token
      for (int c = lane; c < D; c += 64) { float o[1] = {0.f}; RowIO<T, 1>::store(DZ + r * D + c, o); }
      continue;
    }
    for (int c = lane; c < D; c += 64) {
      float t1[1];
      RowIO<T, 1>::load(X + r * D + c, t1);
      xr[c] = t1[0];
      float g1[1] = {0.f};
      if (G) RowIO<T, 1>::load(G + r * D + c, g1);
      gr[c] = g1[0];
    }
    const float gld = a.dtok ? a.dtok[r] : 0.f;
    __syncthreads();
    float pre[HQ], A[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) { pre[q] = b1v[q]; A[q] = 0.f; }
    for (int i = 0; i < D; ++i) {
      const float xi = xr[i];
#pragma unroll
      for (int q = 0; q < HQ; ++q) pre[q] = fmaf(xi, a.w1t[i * Hp + jq[q]], pre[q]);
    }
    for (int i = D - 1; i >= 0; --i) {
      const float xi = xr[i];
      float w1v[HQ], ws[HQ], wt[HQ], h[HQ], dh[HQ];
      float sp = 0.f, ad = 0.f;
#pragma unroll
      for (int q = 0; q < HQ; ++q) {
        w1v[q] = a.w1t[i * Hp + jq[q]];
        ws[q] = a.w2[i * Hp + jq[q]];
        wt[q] = a.w2[(D + i) * Hp + jq[q]];
        pre[q] = fmaf(-xi, w1v[q], pre[q]);
        act_fd<ACT>(pre[q], h[q], dh[q]);
        if (!jon[q]) { h[q] = 0.f; dh[q] = 0.f; }
        sp = fmaf(h[q], ws[q], sp);
        ad = fmaf(A[q], w1v[q], ad);
      }
      sp = wave_sum(sp);
      ad = wave_sum(ad);
      const float s = sp + a.b2[i];
      const float dx = gr[i] + ad;
      const float dzi = dx * expf(-s);
      const float dt = -dzi, ds = fmaf(-dx, xi, gld);
#pragma unroll
      for (int q = 0; q < HQ; ++q) {
        if (jon[q]) {
          gW1[i * Hp + jq[q]] += xi * A[q];
          gW2[i * Hp + jq[q]] += ds * h[q];
          gW2[(D + i) * Hp + jq[q]] += dt * h[q];
        }
        A[q] = fmaf(dh[q], fmaf(ds, ws[q], dt * wt[q]), A[q]);
      }
      if (lane == 0) {
        gb2[i] += ds; gb2[D + i] += dt;
        zr[a.parity ? D - 1 - i : i] = dzi;
      }
    }
#pragma unroll
    for (int q = 0; q < HQ; ++q) gb1[q] += jon[q] ? A[q] : 0.f;
    __syncthreads();
    for (int c = lane; c < D; c += 64) { float o[1] = {zr[c]}; RowIO<T, 1>::store(DZ + r * D + c, o); }
    __syncthreads();
  }
  float* P = a.part + (int64_t)blockIdx.x * (3 * D * Hp + Hp + 2 * D);
  for (int i = lane; i < 3 * D * Hp; i += 64) P[i] = smem[i];
#pragma unroll
  for (int q = 0; q < HQ; ++q)
    if (jon[q]) P[3 * D * Hp + jq[q]] = gb1[q];
  for (int i = lane; i < 2 * D; i += 64) P[3 * D * Hp + Hp + i] = gb2[i];
}

static size_t seq_bwd_lds_bytes(int D, int Hp) { return ((size_t)3 * D * Hp + 5 * D) * sizeof(float); }
static bool seq_bwd_supported(int64_t D, int64_t Hp) {
  return seq_supported(D, Hp) && seq_bwd_lds_bytes((int)D, (int)Hp) <= 156 * 1024;
}
static int seq_bwd_blocks(int64_t rows) { return (int)(rows < 768 ? (rows < 1 ? 1 : rows) : 768); }

template <typename T, int ACT>
int launch_seq_bwd_hq(const SeqBwdArgs& a, size_t lds, unsigned grid, hipStream_t st) {
  const int hq = (a.Hp + 63) / 64;
#define MAF_BWD_HQ(Q)                                                                                     \
  {                                                                                                       \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_maf_seq_bwd<T, ACT, Q>),          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
    if (e != hipSuccess) { mivae_set_error("maf_sequential_bwd: LDS opt-in failed: %s", hipGetErrorString(e)); \
                           return MIVAE_ELAUNCH; }                                                        \
    k_maf_seq_bwd<T, ACT, Q><<<grid, 64, lds, st>>>(a);                                                   \
  }
  if (hq <= 1) MAF_BWD_HQ(1) else if (hq == 2) MAF_BWD_HQ(2) else MAF_BWD_HQ(4)
#undef MAF_BWD_HQ
  return MIVAE_OK;
}
template <typename T>
int launch_seq_bwd(const SeqBwdArgs& a, int act, size_t lds, unsigned grid, hipStream_t st) {
  switch (act) {
    case 0: return launch_seq_bwd_hq<T, 0>(a, lds, grid, st);
    case 1: return launch_seq_bwd_hq<T, 1>(a, lds, grid, st);
    case 2: return launch_seq_bwd_hq<T, 2>(a, lds, grid, st);
    case 3: return launch_seq_bwd_hq<T, 3>(a, lds, grid, st);
    case 4: return launch_seq_bwd_hq<T, 4>(a, lds, grid, st);
    case 5: return launch_seq_bwd_hq<T, 5>(a, lds, grid, st);
    case 6: return launch_seq_bwd_hq<T, 6>(a, lds, grid, st);
    default: mivae_set_error("maf_sequential_bwd: unknown activation %d", act); return MIVAE_EINVAL;
  }
}

struct AffCfg { int vec, L; unsigned grid; };
static AffCfg aff_cfg(int64_t rows, int64_t D, int dtype, bool aligned) {
  AffCfg c;
  const int vmax = dtype == MIVAE_BF16 ? 8 : 4;
  c.vec = (aligned && D % vmax == 0) ? vmax : 1;
  c.L = lanes_per_row(D, c.vec);
  const int rpb = kBlock / c.L;
  c.grid = (unsigned)((rows + rpb - 1) / rpb);
  return c;
}

}  // namespace

extern "C" int mivae_maf_affine_fwd(const void* z, const void* st, const int64_t* mask, void* out, float* tok_ld,
                                    int64_t rows, int64_t D, int parity, int dtype, void* stream) {
  MIVAE_REQUIRE(rows >= 0 && D >= 1 && D <= (1 << 20), "maf_affine_fwd: bad sizes");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "maf_affine_fwd: dtype must be f32 or bf16");
  if (rows == 0) return MIVAE_OK;
  MIVAE_REQUIRE(z && st && out && tok_ld, "maf_affine_fwd: null pointer");
  MIVAE_REQUIRE(z != out, "maf_affine_fwd: in-place not supported");
  const AffCfg c = aff_cfg(rows, D, dtype, aligned16(z) && aligned16(st) && aligned16(out));
  hipStream_t s = S_(stream);
  if (dtype == MIVAE_F32) {
    if (c.vec == 4) k_maf_affine_fwd<float, 4><<<c.grid, kBlock, 0, s>>>((const float*)z, (const float*)st, mask, (float*)out, tok_ld, rows, (int)D, c.L, parity);
    else k_maf_affine_fwd<float, 1><<<c.grid, kBlock, 0, s>>>((const float*)z, (const float*)st, mask, (float*)out, tok_ld, rows, (int)D, c.L, parity);
  } else {
    typedef unsigned short U;
    if (c.vec == 8) k_maf_affine_fwd<U, 8><<<c.grid, kBlock, 0, s>>>((const U*)z, (const U*)st, mask, (U*)out, tok_ld, rows, (int)D, c.L, parity);
    else k_maf_affine_fwd<U, 1><<<c.grid, kBlock, 0, s>>>((const U*)z, (const U*)st, mask, (U*)out, tok_ld, rows, (int)D, c.L, parity);
  }
  MIVAE_CHECK_LAUNCH("maf_affine_fwd");
  return MIVAE_OK;
}

extern "C" int mivae_maf_affine_bwd(const void* dout, const float* dtok, const void* z, const void* st,
                                    const int64_t* mask, void* dz, void* dst, int64_t rows, int64_t D, int parity,
                                    int dtype, void* stream) {
  MIVAE_REQUIRE(rows >= 0 && D >= 1 && D <= (1 << 20), "maf_affine_bwd: bad sizes");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "maf_affine_bwd: dtype must be f32 or bf16");
  if (rows == 0) return MIVAE_OK;
  MIVAE_REQUIRE(z && st && dz && dst, "maf_affine_bwd: null pointer");
  const AffCfg c = aff_cfg(rows, D, dtype, aligned16(z) && aligned16(st) && aligned16(dz) && aligned16(dst) &&
                                               (!dout || aligned16(dout)));
  hipStream_t s = S_(stream);
  if (dtype == MIVAE_F32) {
    if (c.vec == 4) k_maf_affine_bwd<float, 4><<<c.grid, kBlock, 0, s>>>((const float*)dout, dtok, (const float*)z, (const float*)st, mask, (float*)dz, (float*)dst, rows, (int)D, c.L, parity);
    else k_maf_affine_bwd<float, 1><<<c.grid, kBlock, 0, s>>>((const float*)dout, dtok, (const float*)z, (const float*)st, mask, (float*)dz, (float*)dst, rows, (int)D, c.L, parity);
  } else {
    typedef unsigned short U;
    if (c.vec == 8) k_maf_affine_bwd<U, 8><<<c.grid, kBlock, 0, s>>>((const U*)dout, dtok, (const U*)z, (const U*)st, mask, (U*)dz, (U*)dst, rows, (int)D, c.L, parity);
    else k_maf_affine_bwd<U, 1><<<c.grid, kBlock, 0, s>>>((const U*)dout, dtok, (const U*)z, (const U*)st, mask, (U*)dz, (U*)dst, rows, (int)D, c.L, parity);
  }
  MIVAE_CHECK_LAUNCH("maf_affine_bwd");
  return MIVAE_OK;
}

extern "C" int mivae_maf_sequential_supported(int64_t D, int64_t Hp) { return seq_supported(D, Hp) ? 1 : 0; }

extern "C" int mivae_maf_sequential(const void* z, const float* w1t, const float* b1, const float* w2, const float* b2,
                                    const int64_t* mask, void* out, float* tok_ld, int64_t rows, int64_t D, int64_t Hp,
                                    int act, int parity, int dtype, void* stream) {
  MIVAE_REQUIRE(rows >= 0, "maf_sequential: rows < 0");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "maf_sequential: dtype must be f32 or bf16");
  if (!seq_supported(D, Hp)) {
    mivae_set_error("maf_sequential: unsupported shape D=%lld Hp=%lld (Hp = 16 * 2^k <= 256, weights + tiles <= 144 KiB LDS)",
                    (long long)D, (long long)Hp);
    return MIVAE_EUNSUPPORTED;
  }
  if (rows == 0) return MIVAE_OK;
  MIVAE_REQUIRE(z && w1t && b1 && w2 && b2 && out && tok_ld, "maf_sequential: null pointer");
  MIVAE_REQUIRE(aligned16(w1t) && aligned16(w2), "maf_sequential: packed weights must be 16-byte aligned");
  SeqArgs a = {z, w1t, b1, w2, b2, mask, out, tok_ld, rows, (int)D, (int)Hp, (int)(Hp / kHPL), parity ? 1 : 0};
  const int tpb = kBlock / a.L;
  const unsigned grid = (unsigned)((rows + tpb - 1) / tpb);
  const size_t lds = seq_lds_bytes(a.D, a.Hp);
  const int rc = dtype == MIVAE_F32 ? launch_seq<float>(a, act, lds, grid, S_(stream))
                                    : launch_seq<unsigned short>(a, act, lds, grid, S_(stream));
  if (rc != MIVAE_OK) return rc;
  MIVAE_CHECK_LAUNCH("maf_sequential");
  return MIVAE_OK;
}

extern "C" int mivae_maf_sequential_bwd_supported(int64_t D, int64_t Hp) { return seq_bwd_supported(D, Hp) ? 1 : 0; }
// rows of the partial matrix mivae_maf_sequential_bwd writes; each row holds [dW1t (D, Hp) | dW2 (2D, Hp) | db1 (Hp) |
// db2 (2D)] floats in the packed layout of mivae_maf_sequential
extern "C" int64_t mivae_maf_sequential_bwd_blocks(int64_t rows) { return seq_bwd_blocks(rows); }

// Backward of mivae_maf_sequential: x = that call's `out` (rows, D), dout (rows, D) or NULL and dtok (rows) or NULL the
// cotangents of out and tok_ld; writes dz (rows, D) and the per-block parameter-gradient partials (sum the rows of
// `part` with mivae_colsum).  Autograd of maf.py:47-62 (MAFlow.backward, the training direction of IAFlow, iaf.py:21-23).
extern "C" int mivae_maf_sequential_bwd(const void* x, const void* dout, const float* dtok, const float* w1t, const float* b1,
                                        const float* w2, const float* b2, const int64_t* mask, void* dz, float* part,
                                        int64_t rows, int64_t D, int64_t Hp, int act, int parity, int dtype, void* stream) {
  MIVAE_REQUIRE(rows >= 0, "maf_sequential_bwd: rows < 0");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "maf_sequential_bwd: dtype must be f32 or bf16");
  if (!seq_bwd_supported(D, Hp)) {
    mivae_set_error("maf_sequential_bwd: unsupported shape D=%lld Hp=%lld (gradient image 3 D Hp floats must fit the LDS)",
                    (long long)D, (long long)Hp);
    return MIVAE_EUNSUPPORTED;
  }
  if (rows == 0) return MIVAE_OK;
  MIVAE_REQUIRE(x && w1t && b1 && w2 && b2 && dz && part, "maf_sequential_bwd: null pointer");
  SeqBwdArgs a = {x, dout, dtok, mask, w1t, b1, w2, b2, dz, part, rows, (int)D, (int)Hp, parity ? 1 : 0};
  const unsigned grid = (unsigned)seq_bwd_blocks(rows);
  const size_t lds = seq_bwd_lds_bytes(a.D, a.Hp);
  const int rc = dtype == MIVAE_F32 ? launch_seq_bwd<float>(a, act, lds, grid, S_(stream))
                                    : launch_seq_bwd<unsigned short>(a, act, lds, grid, S_(stream));
  if (rc != MIVAE_OK) return rc;
  MIVAE_CHECK_LAUNCH("maf_sequential_bwd");
  return MIVAE_OK;
}
