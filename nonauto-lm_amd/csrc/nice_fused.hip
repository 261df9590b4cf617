// Fused NICE (affine-coupling) flow stack for gfx950: K coupling steps with feed-forward s/t nets
// (one hidden layer) in ONE launch, forward / inverse / backward.
//
// Reference semantics: NICECoupling.forward/.backward (vae_lm/models/base/flows/couplings.py:141-183)
// with FeedForwardCoupling nets (couplings.py:50-51, nn/feedforward.py:93-98), iterated by
// FlowPrior.forward_pass / backward_pass (nonauto_lm/priors/flow_prior.py:37-70), optionally followed
// by the base log-prob (priors/prior.py:148-158).
//
// Design (MI355X-first):
//  * one wave64 owns 32 consecutive tokens for ALL K steps: z is read once and written once.
//  * the reference never swaps halves (couplings.py:185-188), so x0 = z[..., :D/2] is the B operand
//    of every first-layer GEMM of every step: it is loaded once and stays in registers.
//  * everything is computed transposed (features x tokens) on v_mfma_f32_32x32x16_bf16 (bf16 mode)
//    or v_mfma_f32_32x32x2_f32 (exact fp32 mode):   a^T = W1 . x0^T  ->  act  ->  s^T = W2 . hid^T.
//    The first product's 32x32 accumulator (token on the lane, hidden unit in the registers) is
//    converted in registers into the B operand of the second product: no LDS round trip, no
//    barriers, waves are independent.  Weights are pre-packed once per call (k_nice_pack) in exactly
//    the per-lane fragment order (incl. the permuted k order the accumulator hand-off needs), so every
//    weight fetch is a fully coalesced 1 KiB (bf16) wave load.
//  * masked log-det: per-lane partial over the registers, one __shfl_xor across the two half-waves,
//    fp32 throughout; per-token value out, per-sentence sum is mivae_segment_sum (fixed order).
//  * backward recomputes s,t from x0 and the saved per-step x1, produces dz in the same launch and
//    emits token-contiguous ([feature][token]) ds/dt, hidden and d(pre-activation) tensors so the
//    weight gradients are K-contiguous NT GEMMs over the token axis.
#include "common.hpp"

using namespace mivae;

namespace {

enum Act { ACT_LINEAR = 0, ACT_GELU = 1, ACT_SWISH = 2, ACT_RELU = 3, ACT_TANH = 4, ACT_SIGMOID = 5 };

// activation applied to a whole accumulator tile; `act` is wave-uniform (kernel argument)
__device__ __forceinline__ void act_tile(int act, f32x16& a) {
  switch (act) {
    case ACT_GELU:
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = a[r] * 0.5f * (1.0f + erff(a[r] * kInvSqrt2));
      break;
    case ACT_SWISH:
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = a[r] / (1.0f + expf(-a[r]));
      break;
    case ACT_RELU:
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = a[r] > 0.f ? a[r] : 0.f;
      break;
    case ACT_TANH:
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = tanhf(a[r]);
      break;
    case ACT_SIGMOID:
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 1.0f / (1.0f + expf(-a[r]));
      break;
    default:
      break;
  }
}
// value (in place) and derivative
__device__ __forceinline__ void act_tile_fd(int act, f32x16& a, f32x16& d) {
  switch (act) {
    case ACT_GELU:
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = a[r];
        const float cdf = 0.5f * (1.0f + erff(x * kInvSqrt2));
        a[r] = x * cdf;
        d[r] = cdf + x * kInvSqrt2Pi * expf(-0.5f * x * x);
      }
      break;
    case ACT_SWISH:
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = a[r];
        const float sg = 1.0f / (1.0f + expf(-x));
        a[r] = x * sg;
        d[r] = sg + x * sg * (1.0f - sg);
      }
      break;
    case ACT_RELU:
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = a[r];
        a[r] = x > 0.f ? x : 0.f;
        d[r] = x > 0.f ? 1.f : 0.f;
      }
      break;
    case ACT_TANH:
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float y = tanhf(a[r]);
        a[r] = y;
        d[r] = 1.0f - y * y;
      }
      break;
    case ACT_SIGMOID:
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float y = 1.0f / (1.0f + expf(-a[r]));
        a[r] = y;
        d[r] = y * (1.0f - y);
      }
      break;
    default:
#pragma unroll
      for (int r = 0; r < 16; ++r) d[r] = 1.f;
      break;
  }
}

// ---- MFMA policies ---------------------------------------------------------------------------------
// C/D layout of both (guide section 3): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
__host__ __device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

struct MmaBF16 {
  typedef bf16x8 AF;  // A fragment: row = lane & 31, k = 8 * (lane >> 5) + j
  typedef bf16x8 BF;  // B fragment: col = lane & 31, k = 8 * (lane >> 5) + j
  typedef unsigned short E;
  static constexpr int KS = 16;   // k per MFMA
  static constexpr int NS = 2;    // MFMAs per 32-deep k block
  static constexpr int EPF = 8;   // elements per lane per fragment
  static __device__ __forceinline__ f32x16 mma(AF a, BF b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  // accumulator tile X (rows in registers) -> B fragment of k-step s of a product summing over X's rows.
  // element j of lane half hh is row 16 s + 8 (j >> 2) + 4 hh + (j & 3) = acc_row(8 s + j, hh).
  template <int s>
  static __device__ __forceinline__ BF from_acc(const f32x16& x) {
    BF f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16_t)x[8 * s + j];
    return f;
  }
  static __host__ __device__ __forceinline__ int nat_k(int ks, int hh, int j) { return 16 * ks + 8 * hh + j; }
  static __host__ __device__ __forceinline__ int perm_k(int s, int hh, int j) { return acc_row(8 * s + j, hh); }
  static __device__ __forceinline__ E enc(float v) { return f32_to_bf16(v); }
};

struct MmaF32 {
  typedef float AF;  // A: row = lane & 31, k = lane >> 5
  typedef float BF;  // B: col = lane & 31, k = lane >> 5
  typedef float E;
  static constexpr int KS = 2;
  static constexpr int NS = 16;
  static constexpr int EPF = 1;
  static __device__ __forceinline__ f32x16 mma(AF a, BF b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  template <int s>
  static __device__ __forceinline__ BF from_acc(const f32x16& x) { return x[s]; }
  static __host__ __device__ __forceinline__ int nat_k(int ks, int hh, int) { return 2 * ks + hh; }
  static __host__ __device__ __forceinline__ int perm_k(int s, int hh, int) { return acc_row(s, hh); }
  static __device__ __forceinline__ E enc(float v) { return v; }
};

// ---- packed weight geometry ------------------------------------------------------------------------
struct Geom {
  int h, H, hp, Hp, HT, HC;
  int esize;                              // bytes per packed weight element
  int64_t w_elems;                        // hp * Hp
  int64_t off_b1, off_b2;                 // byte offsets inside a net block
  int64_t net_bytes, step_bytes;
};
static Geom make_geom(int h, int H, int esize) {
  Geom g;
  g.h = h; g.H = H;
  g.hp = (h + 31) / 32 * 32; g.Hp = (H + 31) / 32 * 32;
  g.HT = g.hp / 32; g.HC = g.Hp / 32;
  g.esize = esize;
  g.w_elems = (int64_t)g.hp * g.Hp;
  g.off_b1 = 4 * g.w_elems * esize;
  g.off_b2 = g.off_b1 + (int64_t)g.Hp * 4;
  g.net_bytes = g.off_b2 + (int64_t)g.hp * 4;   // multiple of 128
  g.step_bytes = 2 * g.net_bytes;
  return g;
}

// Pack kernel: ptrs[k][8] = {W1s, b1s, W2s, b2s, W1t, b1t, W2t, b2t} (fp32, torch.nn.Linear layout:
// W1 (H, h), W2 (h, H)).  grid = (ceil(per_net_items / 256), 2 nets, K).
template <class P>
__global__ void k_nice_pack(const int64_t* __restrict__ ptrs, char* __restrict__ packed, Geom g) {
  const int net = blockIdx.y, k = blockIdx.z;
  const float* W1 = reinterpret_cast<const float*>(ptrs[k * 8 + net * 4 + 0]);
  const float* b1 = reinterpret_cast<const float*>(ptrs[k * 8 + net * 4 + 1]);
  const float* W2 = reinterpret_cast<const float*>(ptrs[k * 8 + net * 4 + 2]);
  const float* b2 = reinterpret_cast<const float*>(ptrs[k * 8 + net * 4 + 3]);
  char* base = packed + (int64_t)k * g.step_bytes + (int64_t)net * g.net_bytes;
  typename P::E* W = reinterpret_cast<typename P::E*>(base);
  const int64_t total = 4 * g.w_elems + g.Hp + g.hp;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  if (idx < 4 * g.w_elems) {
    const int which = (int)(idx / g.w_elems);
    int64_t i = idx % g.w_elems;
    const int j = (int)(i % P::EPF); i /= P::EPF;
    const int lane = (int)(i % 64); i /= 64;
    const int r = lane & 31, hh = lane >> 5;
    float v = 0.f;
    if (which == 0) {  // W1n [c][ks][lane][j] : W1[32c + r][nat k]
      const int nks = g.hp / P::KS;
      const int ks = (int)(i % nks), c = (int)(i / nks);
      const int row = 32 * c + r, col = P::nat_k(ks, hh, j);
      if (row < g.H && col < g.h) v = W1[(int64_t)row * g.h + col];
    } else {
      const int s = (int)(i % P::NS); i /= P::NS;
      const int t = (int)(i % g.HT), c = (int)(i / g.HT);
      const int pk = P::perm_k(s, hh, j);
      if (which == 1) {         // W2p : W2[32t + r][32c + pk]
        const int row = 32 * t + r, col = 32 * c + pk;
        if (row < g.h && col < g.H) v = W2[(int64_t)row * g.H + col];
      } else if (which == 2) {  // W2Tp : A[row = hidden 32c + r][k = out 32t + pk] = W2[32t + pk][32c + r]
        const int row = 32 * t + pk, col = 32 * c + r;
        if (row < g.h && col < g.H) v = W2[(int64_t)row * g.H + col];
      } else {                  // W1Tp : A[row = in 32t + r][k = hidden 32c + pk] = W1[32c + pk][32t + r]
        const int row = 32 * c + pk, col = 32 * t + r;
        if (row < g.H && col < g.h) v = W1[(int64_t)row * g.h + col];
      }
    }
    W[idx] = P::enc(v);
  } else {
    int64_t i = idx - 4 * g.w_elems;
    float* bp = reinterpret_cast<float*>(base + g.off_b1);
    if (i < g.Hp) {  // b1a [c][hh][16]
      const int reg = (int)(i % 16), hh = (int)((i / 16) % 2), c = (int)(i / 32);
      const int row = 32 * c + acc_row(reg, hh);
      bp[i] = row < g.H ? b1[row] : 0.f;
    } else {         // b2a [t][hh][16]
      i -= g.Hp;
      const int reg = (int)(i % 16), hh = (int)((i / 16) % 2), t = (int)(i / 32);
      const int row = 32 * t + acc_row(reg, hh);
      bp[g.Hp + i] = row < g.h ? b2[row] : 0.f;
    }
  }
}

// ---- per-lane I/O in fragment / accumulator layouts ------------------------------------------------
// x0 as B fragments (natural k order): lane (tok = lane & 31, hh) holds x0[tok][nat_k(ks, hh, j)].
template <typename T, bool VEC, int NKS>
__device__ __forceinline__ void load_x0_f32(const T* __restrict__ zrow, bool valid, int h, int hh,
                                            float (&x0f)[NKS]) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int f = 2 * ks + hh;
    float v = 0.f;
    if (valid && f < h) {
      float t1[1];
      RowIO<T, 1>::load(zrow + f, t1);
      v = t1[0];
    }
    x0f[ks] = v;
  }
}
template <typename T, bool VEC, int NKS>
__device__ __forceinline__ void load_x0_bf16(const T* __restrict__ zrow, bool valid, int h, int hh,
                                             bf16x8 (&x0f)[NKS]) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int f = 16 * ks + 8 * hh;
    u16x8 raw = {0, 0, 0, 0, 0, 0, 0, 0};
    if (VEC) {
      if (valid && f < h) raw = *reinterpret_cast<const u16x8*>(zrow + f);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (valid && f + j < h) raw[j] = ((const unsigned short*)zrow)[f + j];
    }
    x0f[ks] = __builtin_bit_cast(bf16x8, raw);
  }
}

// accumulator layout tile set: regs[t][r] <-> row[32 t + acc_row(r, hh)], groups of 4 contiguous.
template <typename T, bool VEC, int HT>
__device__ __forceinline__ void load_acc(const T* __restrict__ row, bool valid, int h, int hh,
                                         f32x16 (&x)[HT]) {
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = 32 * t + 8 * g + 4 * hh;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (VEC) {
        if (valid && f < h) RowIO<T, 4>::load(row + f, v);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (valid && f + i < h) {
            float t1[1];
            RowIO<T, 1>::load(row + f + i, t1);
            v[i] = t1[0];
          }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) x[t][4 * g + i] = v[i];
    }
}
template <typename T, bool VEC, int HT>
__device__ __forceinline__ void store_acc(T* __restrict__ row, bool valid, int h, int hh,
                                          const f32x16 (&x)[HT]) {
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = 32 * t + 8 * g + 4 * hh;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = x[t][4 * g + i];
      if (VEC) {
        if (valid && f < h) RowIO<T, 4>::store(row + f, v);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (valid && f + i < h) {
            float t1[1] = {v[i]};
            RowIO<T, 1>::store(row + f + i, t1);
          }
      }
    }
}

// transposed ([feature row][token]) store of one accumulator tile: row = row0 + acc_row(r, hh)
template <typename T>
__device__ __forceinline__ void store_tile_t(T* __restrict__ base, int64_t ld, int row0, int hh,
                                             int64_t col, const f32x16& x) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float t1[1] = {x[r]};
    RowIO<T, 1>::store(base + (int64_t)(row0 + acc_row(r, hh)) * ld + col, t1);
  }
}

struct StackArgs {
  const void* z_in;     // (N, D)
  void* z_out;          // (N, D)
  void* x1_hist;        // (K, N, h) or null            [fwd: written, bwd: read]
  const int64_t* mask;  // (Bn / mask_div, S) or null
  const char* packed;   // K * step_bytes
  float* tok_val;       // (N) per-token masked log-det (+ base log-prob)
  int64_t N;
  int S, mask_div, D, K;
  int inverse, base, act;
  float mu, inv2var, cst;  // base dist: -(z-mu)^2 * inv2var - cst
  // backward only
  const void* dz_out;   // (N, D) or null
  const float* dlogp;   // (Bn) or null: d loss / d (per-sentence sum of tok_val)
  void* dz_in;          // (N, D)
  void* Gt;             // (K, 2, Hp, Npad) d pre-activation, token-contiguous
  void* Ht;             // (K, 2, Hp, Npad) hidden activations
  void* dSt;            // (K, 2, hp, Npad) ds / dt
  int64_t Npad;
  Geom g;
};

template <class P>
__device__ __forceinline__ const typename P::AF* frag_ptr(const char* net, int64_t elem_off, int lane) {
  return reinterpret_cast<const typename P::AF*>(net + elem_off * sizeof(typename P::E)) + lane;
}

template <int N>
struct IntC { static constexpr int value = N; };

// second-layer style product: acc[t] += A[c][t][s] . Bfrag(s) for all t, with B taken from the
// accumulator tile `x` (sum over x's rows).  A block for chunk c starts at elem offset `off`.
template <class P, int HT, int s>
struct AccumFromAcc {
  static __device__ __forceinline__ void run(const char* net, int64_t off, int lane, const f32x16& x,
                                             f32x16 (&acc)[HT]) {
    typename P::BF b = P::template from_acc<s>(x);
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      typename P::AF a = frag_ptr<P>(net, off + ((int64_t)t * P::NS + s) * 64 * P::EPF, 0)[lane];
      acc[t] = P::mma(a, b, acc[t]);
    }
    if constexpr (s + 1 < P::NS) AccumFromAcc<P, HT, s + 1>::run(net, off, lane, x, acc);
  }
};
// reduce over the tile set: out += sum_t A[c][t][s] . Bfrag_t(s)   (B from accumulator tiles y[t])
template <class P, int HT, int s>
struct ReduceFromAcc {
  static __device__ __forceinline__ void run(const char* net, int64_t off, int lane,
                                             const f32x16 (&y)[HT], f32x16& out) {
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      typename P::BF b = P::template from_acc<s>(y[t]);
      typename P::AF a = frag_ptr<P>(net, off + ((int64_t)t * P::NS + s) * 64 * P::EPF, 0)[lane];
      out = P::mma(a, b, out);
    }
    if constexpr (s + 1 < P::NS) ReduceFromAcc<P, HT, s + 1>::run(net, off, lane, y, out);
  }
};

// first-layer pre-activation of hidden chunk c: a = b1 + W1[c] . x0^T
template <class P, int NKS>
__device__ __forceinline__ f32x16 preact(const char* net, const Geom& g, int c, int lane, int hh,
                                         const typename P::BF (&x0f)[NKS]) {
  const float* b1 = reinterpret_cast<const float*>(net + g.off_b1) + (c * 2 + hh) * 16;
  f32x16 a;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = *reinterpret_cast<const f32x4*>(b1 + 4 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[4 * q + i] = v[i];
  }
  const typename P::AF* w = frag_ptr<P>(net, (int64_t)c * NKS * 64 * P::EPF, lane);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) a = P::mma(w[(int64_t)ks * 64], x0f[ks], a);
  return a;
}

template <int HT>
__device__ __forceinline__ void init_bias2(const char* net, const Geom& g, int hh, f32x16 (&acc)[HT]) {
  const float* b2 = reinterpret_cast<const float*>(net + g.off_b2);
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = *reinterpret_cast<const f32x4*>(b2 + (t * 2 + hh) * 16 + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t][4 * q + i] = v[i];
    }
}

// s (or t) for one net: acc = b2 + W2 . act(b1 + W1 . x0^T)
template <class P, int HT, int NKS>
__device__ __forceinline__ void net_forward(const char* net, const Geom& g, int act, int lane, int hh,
                                            const typename P::BF (&x0f)[NKS], f32x16 (&acc)[HT]) {
  init_bias2<HT>(net, g, hh, acc);
  for (int c = 0; c < g.HC; ++c) {
    f32x16 a = preact<P, NKS>(net, g, c, lane, hh, x0f);
    act_tile(act, a);
    AccumFromAcc<P, HT, 0>::run(net, g.w_elems + (int64_t)c * HT * 1024, lane, a, acc);
  }
}

template <class P, typename T, int HT, bool VEC>
__global__ void __launch_bounds__(256) k_stack_fwd(StackArgs a) {
  constexpr int hp = HT * 32;
  constexpr int NKS = hp / P::KS;
  const int lane = threadIdx.x & 63;
  const int r32 = lane & 31, hh = lane >> 5;
  const int64_t tok0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32;
  if (tok0 >= a.N) return;  // wave-uniform
  const int64_t tok = tok0 + r32;
  const bool valid = tok < a.N;
  const Geom& g = a.g;
  const int h = g.h;
  float m = 0.f;
  if (valid) {
    m = 1.f;
    if (a.mask) {
      const int64_t bn = tok / a.S;
      m = (float)a.mask[(bn / a.mask_div) * a.S + (tok - bn * a.S)];
    }
  }
  const T* zrow = reinterpret_cast<const T*>(a.z_in) + tok * a.D;
  typename P::BF x0f[NKS];
  if constexpr (P::KS == 2) load_x0_f32<T, VEC, NKS>(zrow, valid, h, hh, x0f);
  else load_x0_bf16<T, VEC, NKS>(zrow, valid, h, hh, x0f);
  f32x16 x1[HT];
  load_acc<T, VEC, HT>(zrow + h, valid, h, hh, x1);
  float ld = 0.f;

  for (int kk = 0; kk < a.K; ++kk) {
    const int k = a.inverse ? a.K - 1 - kk : kk;
    const char* snet = a.packed + (int64_t)k * g.step_bytes;
    const char* tnet = snet + g.net_bytes;
    if (a.x1_hist)
      store_acc<T, VEC, HT>(reinterpret_cast<T*>(a.x1_hist) + ((int64_t)k * a.N + tok) * h, valid, h, hh, x1);
    f32x16 sacc[HT], tacc[HT];
    net_forward<P, HT, NKS>(snet, g, a.act, lane, hh, x0f, sacc);
    net_forward<P, HT, NKS>(tnet, g, a.act, lane, hh, x0f, tacc);
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float sv = sacc[t][r], tv = tacc[t][r];
        if (!a.inverse) {
          x1[t][r] = (expf(sv) * x1[t][r] + tv) * m;
          ld += sv;
        } else {
          x1[t][r] = ((x1[t][r] - tv) * expf(-sv)) * m;
          ld -= sv;
        }
      }
    if (kk == 0 && m == 0.f) {  // later steps see x0 * mask (couplings.py:160: z *= mask)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if constexpr (P::KS == 2) x0f[ks] = 0.f;
        else x0f[ks] = __builtin_bit_cast(bf16x8, (u16x8){0, 0, 0, 0, 0, 0, 0, 0});
      }
    }
  }

  // outputs ------------------------------------------------------------------------------------------
  T* orow = reinterpret_cast<T*>(a.z_out) + tok * a.D;
  store_acc<T, VEC, HT>(orow + h, valid, h, hh, x1);
  float base = 0.f;
  // x0 half: out = x0 * m, in fragment layout
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if constexpr (P::KS == 2) {
      const int f = 2 * ks + hh;
      const float v = x0f[ks] * m;   // (x0f already zeroed when m == 0)
      if (valid && f < h) {
        float t1[1] = {v};
        RowIO<T, 1>::store(orow + f, t1);
        const float d = v - a.mu;
        base += -(d * d) * a.inv2var - a.cst;
      }
    } else {
      const int f = 16 * ks + 8 * hh;
      u16x8 raw = __builtin_bit_cast(u16x8, x0f[ks]);
      if (VEC) {
        if (valid && f < h) *reinterpret_cast<u16x8*>(orow + f) = raw;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (valid && f + j < h) {
          if (!VEC) ((unsigned short*)orow)[f + j] = raw[j];
          const float d = bf16_to_f32(raw[j]) - a.mu;
          base += -(d * d) * a.inv2var - a.cst;
        }
    }
  }
  if (a.base) {
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = 32 * t + acc_row(r, hh);
        if (f < h) {
          float v = x1[t][r];
          if (sizeof(T) == 2) v = bf16_to_f32(f32_to_bf16(v));  // what the stored z_K holds
          const float d = v - a.mu;
          base += -(d * d) * a.inv2var - a.cst;
        }
      }
  } else {
    base = 0.f;
  }
  float tv = (ld + base) * m;
  tv += __shfl_xor(tv, 32, 64);
  if (valid && hh == 0 && a.tok_val) a.tok_val[tok] = tv;
}

// ---- backward ----------------------------------------------------------------------------------------
// one net's share of a step: given ds (accumulator tiles, rows = out features) computes, per hidden
// chunk, hid / act' (recomputed), d hid = W2^T ds, d pre = d hid * act', emits hid and d pre
// token-contiguous, and accumulates dx0 += W1^T d pre.
template <class P, typename T, int HT, int NKS>
__device__ __forceinline__ void net_backward(const char* net, const Geom& g, int act, int lane, int hh,
                                             const typename P::BF (&x0f)[NKS], const f32x16 (&ds)[HT],
                                             f32x16 (&dx0)[HT], T* __restrict__ Gt, T* __restrict__ Ht,
                                             int64_t ld, int64_t col) {
  for (int c = 0; c < g.HC; ++c) {
    f32x16 hid = preact<P, NKS>(net, g, c, lane, hh, x0f);
    f32x16 dact;
    act_tile_fd(act, hid, dact);
    store_tile_t<T>(Ht, ld, 32 * c, hh, col, hid);
    f32x16 dh;
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[r] = 0.f;
    ReduceFromAcc<P, HT, 0>::run(net, 2 * g.w_elems + (int64_t)c * HT * 1024, lane, ds, dh);
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[r] *= dact[r];
    store_tile_t<T>(Gt, ld, 32 * c, hh, col, dh);
    AccumFromAcc<P, HT, 0>::run(net, 3 * g.w_elems + (int64_t)c * HT * 1024, lane, dh, dx0);
  }
}

template <class P, typename T, int HT, bool VEC>
__global__ void __launch_bounds__(256) k_stack_bwd(StackArgs a) {
  constexpr int hp = HT * 32;
  constexpr int NKS = hp / P::KS;
  const int lane = threadIdx.x & 63;
  const int r32 = lane & 31, hh = lane >> 5;
  const int64_t tok0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32;
  if (tok0 >= a.Npad) return;  // wave-uniform; blocks in [N, Npad) still zero-fill the emitted columns
  const int64_t tok = tok0 + r32;
  const bool valid = tok < a.N;
  const Geom& g = a.g;
  const int h = g.h;
  float m = 0.f, dl = 0.f;
  if (valid) {
    const int64_t bn = tok / a.S;
    m = 1.f;
    if (a.mask) m = (float)a.mask[(bn / a.mask_div) * a.S + (tok - bn * a.S)];
    if (a.dlogp) dl = a.dlogp[bn] * m;
  }
  const T* zrow = reinterpret_cast<const T*>(a.z_in) + tok * a.D;
  // x0 * m as B fragments (masked tokens contribute exactly zero gradient)
  typename P::BF x0f[NKS];
  if constexpr (P::KS == 2) load_x0_f32<T, VEC, NKS>(zrow, valid && m != 0.f, h, hh, x0f);
  else load_x0_bf16<T, VEC, NKS>(zrow, valid && m != 0.f, h, hh, x0f);

  // upstream gradients in accumulator layout
  f32x16 gx1[HT], dx0[HT];
  if (a.dz_out) {
    const T* grow = reinterpret_cast<const T*>(a.dz_out) + tok * a.D;
    load_acc<T, VEC, HT>(grow, valid, h, hh, dx0);
    load_acc<T, VEC, HT>(grow + h, valid, h, hh, gx1);
  } else {
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dx0[t][r] = 0.f; gx1[t][r] = 0.f; }
  }
  if (a.base && a.dlogp) {  // d/dz_K of the base log-prob: dl * -(z_K - mu) / std^2
    const T* orow = reinterpret_cast<const T*>(a.z_out) + tok * a.D;
    f32x16 zk0[HT], zk1[HT];
    load_acc<T, VEC, HT>(orow, valid, h, hh, zk0);
    load_acc<T, VEC, HT>(orow + h, valid, h, hh, zk1);
    const float c2 = 2.0f * a.inv2var;
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool in = 32 * t + acc_row(r, hh) < h;
        dx0[t][r] += in ? -dl * (zk0[t][r] - a.mu) * c2 : 0.f;
        gx1[t][r] += in ? -dl * (zk1[t][r] - a.mu) * c2 : 0.f;
      }
  }
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dx0[t][r] *= m;

  const int64_t ldn = a.Npad;
  for (int k = a.K - 1; k >= 0; --k) {
    const char* snet = a.packed + (int64_t)k * g.step_bytes;
    const char* tnet = snet + g.net_bytes;
    f32x16 ds[HT], dt[HT];
    {
      f32x16 x1[HT];
      load_acc<T, VEC, HT>(reinterpret_cast<const T*>(a.x1_hist) + ((int64_t)k * a.N + tok) * h, valid, h, hh, x1);
      net_forward<P, HT, NKS>(snet, g, a.act, lane, hh, x0f, ds);  // ds holds s for now
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool in = 32 * t + acc_row(r, hh) < h;
          const float g1 = gx1[t][r] * m;
          const float ge = g1 * expf(ds[t][r]);
          dt[t][r] = g1;
          ds[t][r] = in ? ge * x1[t][r] + dl : 0.f;
          gx1[t][r] = ge;
        }
    }
    T* dS = reinterpret_cast<T*>(a.dSt) + (int64_t)k * 2 * g.hp * ldn;
    T* Gt = reinterpret_cast<T*>(a.Gt) + (int64_t)k * 2 * g.Hp * ldn;
    T* Ht = reinterpret_cast<T*>(a.Ht) + (int64_t)k * 2 * g.Hp * ldn;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      store_tile_t<T>(dS, ldn, 32 * t, hh, tok, ds[t]);
      store_tile_t<T>(dS + (int64_t)g.hp * ldn, ldn, 32 * t, hh, tok, dt[t]);
    }
    net_backward<P, T, HT, NKS>(snet, g, a.act, lane, hh, x0f, ds, dx0, Gt, Ht, ldn, tok);
    net_backward<P, T, HT, NKS>(tnet, g, a.act, lane, hh, x0f, dt, dx0, Gt + (int64_t)g.Hp * ldn,
                                     Ht + (int64_t)g.Hp * ldn, ldn, tok);
  }
  T* drow = reinterpret_cast<T*>(a.dz_in) + tok * a.D;
  store_acc<T, VEC, HT>(drow, valid, h, hh, dx0);
  store_acc<T, VEC, HT>(drow + h, valid, h, hh, gx1);
}

// ---- host dispatch -----------------------------------------------------------------------------------
template <class P, typename T, int HT>
int launch_stack(bool bwd, bool vec, const StackArgs& a, unsigned grid, unsigned block, hipStream_t st) {
  if (bwd) {
    if (vec) k_stack_bwd<P, T, HT, true><<<grid, block, 0, st>>>(a);
    else k_stack_bwd<P, T, HT, false><<<grid, block, 0, st>>>(a);
  } else {
    if (vec) k_stack_fwd<P, T, HT, true><<<grid, block, 0, st>>>(a);
    else k_stack_fwd<P, T, HT, false><<<grid, block, 0, st>>>(a);
  }
  return 0;
}
template <class P, typename T>
int launch_ht(int HT, bool bwd, bool vec, const StackArgs& a, unsigned grid, unsigned block, hipStream_t st) {
  switch (HT) {
    case 1: return launch_stack<P, T, 1>(bwd, vec, a, grid, block, st);
    case 2: return launch_stack<P, T, 2>(bwd, vec, a, grid, block, st);
    case 3: return launch_stack<P, T, 3>(bwd, vec, a, grid, block, st);
    case 4: return launch_stack<P, T, 4>(bwd, vec, a, grid, block, st);
  }
  return -1;
}

static int run_stack(const char* name, bool bwd, StackArgs a, int64_t Bn, int h, int H, int dtype,
                     int act, hipStream_t st) {
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "%s: bad dtype %d", name, dtype);
  MIVAE_REQUIRE(act >= 0 && act <= ACT_SIGMOID, "%s: unsupported activation id %d", name, act);
  MIVAE_REQUIRE(a.K >= 1 && a.K <= 4096, "%s: K=%d out of range", name, a.K);
  MIVAE_REQUIRE(h >= 1 && H >= 1 && a.D == 2 * h, "%s: D must be 2*h (D=%d h=%d)", name, a.D, h);
  MIVAE_REQUIRE(a.S >= 1 && a.mask_div >= 1 && Bn >= 0, "%s: bad shape", name);
  a.act = act;
  a.g = make_geom(h, H, dtype == MIVAE_BF16 ? 2 : 4);
  if (a.g.HT > 4) {
    mivae_set_error("%s: fused path supports D <= 256 (got D=%d); use the composed path", name, a.D);
    return MIVAE_EUNSUPPORTED;
  }
  a.N = Bn * a.S;
  if (a.N == 0) return MIVAE_OK;
  MIVAE_REQUIRE(a.z_in && a.packed, "%s: null pointer", name);
  const int esz = dtype == MIVAE_BF16 ? 2 : 4;
  // vector I/O needs 16-byte aligned rows and halves: D*esz % 16 == 0, h*esz % 16 == 0 (bf16: h % 8)
  bool vec = aligned16(a.z_in) && (a.z_out == nullptr || aligned16(a.z_out)) &&
             ((int64_t)h * esz) % 16 == 0 && (a.x1_hist == nullptr || aligned16(a.x1_hist));
  if (bwd) vec = vec && aligned16(a.dz_in) && (a.dz_out == nullptr || aligned16(a.dz_out));
  const int64_t ntok = bwd ? a.Npad : a.N;
  const int64_t nwaves = (ntok + 31) / 32;
  unsigned wpb = nwaves >= 4096 ? 4 : (nwaves >= 1024 ? 2 : 1);
  unsigned grid = (unsigned)((nwaves + wpb - 1) / wpb);
  int rc;
  if (dtype == MIVAE_F32) rc = launch_ht<MmaF32, float>(a.g.HT, bwd, vec, a, grid, wpb * 64, st);
  else rc = launch_ht<MmaBF16, unsigned short>(a.g.HT, bwd, vec, a, grid, wpb * 64, st);
  MIVAE_REQUIRE(rc == 0, "%s: no kernel instance for HT=%d", name, a.g.HT);
  MIVAE_CHECK_LAUNCH(name);
  return MIVAE_OK;
}

}  // namespace

// =====================================================================================================
extern "C" int64_t mivae_nice_pack_bytes(int64_t h, int64_t H, int dtype) {
  if (h < 1 || H < 1) return 0;
  return make_geom((int)h, (int)H, dtype == MIVAE_BF16 ? 2 : 4).step_bytes;
}

extern "C" int mivae_nice_pack(const int64_t* ptr_table, void* packed, int64_t K, int64_t h, int64_t H,
                               int dtype, void* stream) {
  MIVAE_REQUIRE(ptr_table && packed, "nice_pack: null pointer");
  MIVAE_REQUIRE(K >= 1 && K <= 4096 && h >= 1 && H >= 1, "nice_pack: bad shape K=%ld h=%ld H=%ld", (long)K, (long)h, (long)H);
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "nice_pack: bad dtype %d", dtype);
  Geom g = make_geom((int)h, (int)H, dtype == MIVAE_BF16 ? 2 : 4);
  const int64_t total = 4 * g.w_elems + g.Hp + g.hp;
  dim3 grid((unsigned)((total + 255) / 256), 2, (unsigned)K);
  if (dtype == MIVAE_BF16) k_nice_pack<MmaBF16><<<grid, 256, 0, (hipStream_t)stream>>>(ptr_table, (char*)packed, g);
  else k_nice_pack<MmaF32><<<grid, 256, 0, (hipStream_t)stream>>>(ptr_table, (char*)packed, g);
  MIVAE_CHECK_LAUNCH("nice_pack");
  return MIVAE_OK;
}

extern "C" int mivae_nice_stack_fwd(const void* z_in, void* z_out, void* x1_hist, const int64_t* mask,
                                    const void* packed, float* tok_val, int64_t Bn, int64_t mask_div,
                                    int64_t S, int64_t D, int64_t H, int64_t K, int act, int inverse,
                                    int base, float mu, float std, int dtype, void* stream) {
  MIVAE_REQUIRE(z_out, "nice_stack_fwd: null z_out");
  MIVAE_REQUIRE(!base || std > 0.f, "nice_stack_fwd: std must be > 0");
  MIVAE_REQUIRE(!(inverse && x1_hist), "nice_stack_fwd: no activation history in the inverse direction");
  StackArgs a = {};
  a.z_in = z_in; a.z_out = z_out; a.x1_hist = x1_hist; a.mask = mask;
  a.packed = (const char*)packed; a.tok_val = tok_val;
  a.S = (int)S; a.mask_div = (int)mask_div; a.D = (int)D; a.K = (int)K;
  a.inverse = inverse; a.base = base;
  a.mu = mu;
  a.inv2var = base ? 1.0f / (2.0f * std * std) : 0.f;
  a.cst = base ? logf(std) + 0.5f * kLog2Pi : 0.f;
  return run_stack("nice_stack_fwd", false, a, Bn, (int)(D / 2), (int)H, dtype, act, (hipStream_t)stream);
}

extern "C" int64_t mivae_nice_stack_npad(int64_t N) { return (N + 31) / 32 * 32; }

extern "C" int mivae_nice_stack_bwd(const void* z_in, const void* z_out, const void* x1_hist,
                                    const int64_t* mask, const void* packed, const void* dz_out,
                                    const float* dlogp, void* dz_in, void* Gt, void* Ht, void* dSt,
                                    int64_t Bn, int64_t mask_div, int64_t S, int64_t D, int64_t H,
                                    int64_t K, int act, int base, float mu, float std, int dtype,
                                    void* stream) {
  MIVAE_REQUIRE(x1_hist && dz_in && Gt && Ht && dSt, "nice_stack_bwd: null pointer");
  MIVAE_REQUIRE(!base || (std > 0.f && z_out), "nice_stack_bwd: base log-prob needs z_out and std > 0");
  StackArgs a = {};
  a.z_in = z_in; a.z_out = const_cast<void*>(z_out); a.x1_hist = const_cast<void*>(x1_hist);
  a.mask = mask; a.packed = (const char*)packed;
  a.S = (int)S; a.mask_div = (int)mask_div; a.D = (int)D; a.K = (int)K;
  a.inverse = 0; a.base = base;
  a.mu = mu;
  a.inv2var = base ? 1.0f / (2.0f * std * std) : 0.f;
  a.cst = 0.f;
  a.dz_out = dz_out; a.dlogp = dlogp; a.dz_in = dz_in; a.Gt = Gt; a.Ht = Ht; a.dSt = dSt;
  a.Npad = mivae_nice_stack_npad(Bn * S);
  return run_stack("nice_stack_bwd", true, a, Bn, (int)(D / 2), (int)H, dtype, act, (hipStream_t)stream);
}
