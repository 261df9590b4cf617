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
