// Invertible linear flows on the matrix cores: InvertibleLinearFlow / InvertibleMultiHeadFlow
// of vae_lm/models/base/flows/invertible_flows.py:16-83.
//
// A latent (N, D) with `heads` blocks of hs = D / heads features per token IS a row-major (R = N heads, hs) matrix, so
// both flows, both directions and the input gradient are one product  Y = X M^T  with a small square M (hs <= 512):
//   forward  M = W (F.linear, :29, :61-63);  inverse  M = W^-1 (:37-40);  d X = d Y M  ->  the same kernel on M^T.
// No mask multiply (the reference does not mask here); the log-det (slogdet(W) x token count x heads) is scalar work
// and stays with the caller.
//
//   k_rowgemm<Mma>   one wave per 32 rows: A fragments straight from X (each row is read once per 32 output columns,
//                    L1/L2 resident), B fragments straight from M (hs^2 elements, L2 resident), MFMA 32x32 tiles
//                    (bf16: v_mfma_f32_32x32x16_bf16; fp32: v_mfma_f32_32x32x2_f32, bit-identical to an fmaf chain).
//                    HBM-bound: reads X, writes Y once.
//   k_rowgemm_wgrad  d M[o][i] = sum_r dY[r][o] X[r][i]: the contraction runs over ROWS, which both operands hold as
//                    their slow index; the f32 MFMA's k = 2 fragments (lane = column, k = row) read them coalesced
//                    without any transpose.  A wave owns a 32 x 32 tile of d M and a slab of rows; slabs are summed in
//                    fixed order by mivae_colsum (run-to-run reproducible, no atomics).
#include "nice_common.hpp"

namespace {
using namespace mivae;

struct RowGemmArgs {
  const void* x;     // (R, hs)
  const void* m;     // (hs, hs) row major, same dtype as x
  void* y;           // (R, hs)
  int64_t R;
  int hs;
};

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemIO<unsigned short> {
  static __device__ __forceinline__ float ld(const unsigned short* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(unsigned short* p, float v) { *p = f32_to_bf16(v); }
};

// A / B fragment of k-step ks for lane (idx = lane & 31, hh = lane >> 5): elements row[nat_k(ks, hh, j)], zero past hs
template <typename Mma, typename T>
__device__ __forceinline__ typename Mma::AF load_frag(const T* __restrict__ row, bool on, int hs, int ks, int hh) {
  if constexpr (Mma::EPF == 8) {
    u16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
    const int k0 = 16 * ks + 8 * hh;
    if (on && k0 < hs) {
      if (k0 + 8 <= hs && (hs & 7) == 0) {
        r = *reinterpret_cast<const u16x8*>(row + k0);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (k0 + j < hs) r[j] = row[k0 + j];
      }
    }
    return __builtin_bit_cast(bf16x8, r);
  } else {
    const int k = 2 * ks + hh;
    return (on && k < hs) ? row[k] : 0.f;
  }
}

template <typename Mma, typename T>
__global__ void __launch_bounds__(256) k_rowgemm(RowGemmArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int idx = lane & 31, hh = lane >> 5;
  const int hs = a.hs;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  if (row0 >= a.R) return;
  const T* X = static_cast<const T*>(a.x);
  const T* M = static_cast<const T*>(a.m);
  T* Y = static_cast<T*>(a.y);
  const bool xon = row0 + idx < a.R;
  const T* xrow = X + (row0 + idx) * hs;
  const int nks = (hs + Mma::KS - 1) / Mma::KS;
  for (int ot = 0; ot * 32 < hs; ++ot) {
    const int o = ot * 32 + idx;             // B fragment column = output feature = row of M
    const bool mon = o < hs;
    const T* mrow = M + (int64_t)o * hs;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ks = 0; ks < nks; ++ks)
      acc = Mma::mma(load_frag<Mma, T>(xrow, xon, hs, ks, hh), load_frag<Mma, T>(mrow, mon, hs, ks, hh), acc);
    // C layout: column = lane & 31 (output feature), row = acc_row(reg, hh): 32 consecutive features per row
    if (mon) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t rr = row0 + acc_row(r, hh);
        if (rr < a.R) ElemIO<T>::st(Y + rr * hs + o, acc[r]);
      }
    }
  }
}

struct WgradArgs {
  const void* x;     // (R, hs)
  const void* dy;    // (R, hs)
  float* part;       // (nslab, hs, hs): part[s][o][i]
  int64_t R;
  int hs, slab_rows;
};

template <typename T>
__global__ void __launch_bounds__(64) k_rowgemm_wgrad(WgradArgs a) {
  const int lane = threadIdx.x;
  const int idx = lane & 31, hh = lane >> 5;
  const int hs = a.hs;
  const int ot = blockIdx.y, it = blockIdx.z;
  const int o = ot * 32 + idx, i = it * 32 + idx;
  const T* X = static_cast<const T*>(a.x);
  const T* G = static_cast<const T*>(a.dy);
  const int64_t lo = (int64_t)blockIdx.x * a.slab_rows;
  int64_t hi = lo + a.slab_rows;
  if (hi > a.R) hi = a.R;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // k = 2 rows per MFMA: lane half hh takes row r + hh; A[m = o][k = row] = dY[row][o], B[k = row][n = i] = X[row][i]
  for (int64_t r = lo; r < hi; r += 2) {
    const int64_t rr = r + hh;
    const bool on = rr < hi;
    const float ga = (on && o < hs) ? ElemIO<T>::ld(G + rr * hs + o) : 0.f;
    const float xb = (on && i < hs) ? ElemIO<T>::ld(X + rr * hs + i) : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, xb, acc, 0, 0, 0);
  }
  float* P = a.part + (int64_t)blockIdx.x * hs * hs;
  if (i < hs) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int oo = ot * 32 + acc_row(r, hh);
      if (oo < hs) P[(int64_t)oo * hs + i] = acc[r];
    }
  }
}

static void wgrad_geom(int64_t R, int hs, int& nslab, int& slab_rows) {
  const int tiles = ((hs + 31) / 32) * ((hs + 31) / 32);
  int64_t want = (2048 + tiles - 1) / tiles;            // ~8 waves per CU in total
  const int64_t maxs = (R + 63) / 64;                   // at least 64 rows per slab
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  int64_t sr = (R + want - 1) / want;
  sr = (sr + 1) / 2 * 2;                                // whole MFMA k-steps
  slab_rows = (int)sr;
  nslab = (int)((R + sr - 1) / sr);
}

}  // namespace

// Y (R, hs) = X (R, hs) M^T, M (hs, hs) row major in the latent dtype: F.linear(x, M) of invertible_flows.py:29, :39,
// :61-63, :75-77 with the heads folded into the rows (R = tokens x heads).  1 <= hs <= 512.
extern "C" int mivae_rowgemm(const void* x, const void* m, void* y, int64_t R, int64_t hs, int dtype, void* stream) {
  MIVAE_REQUIRE(R >= 0 && R < (1ll << 40), "rowgemm: bad row count");
  if (hs < 1 || hs > 512) {
    mivae_set_error("rowgemm: unsupported width hs=%lld (1..512)", (long long)hs);
    return MIVAE_EUNSUPPORTED;
  }
  if (R == 0) return MIVAE_OK;
  MIVAE_REQUIRE(x && m && y, "rowgemm: null pointer");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "rowgemm: dtype must be f32 or bf16");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || (hs % 8) != 0 || (aligned16(x) && aligned16(m)),
                "rowgemm: bf16 operands must be 16-byte aligned");
  RowGemmArgs a = {x, m, y, R, (int)hs};
  const unsigned grid = (unsigned)((R + 127) / 128);
  if (dtype == MIVAE_F32) k_rowgemm<MmaF32, float><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  else k_rowgemm<MmaBF16, unsigned short><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  MIVAE_CHECK_LAUNCH("rowgemm");
  return MIVAE_OK;
}

// number of row slabs (= rows of the partial matrix) mivae_rowgemm_wgrad writes for (R, hs)
extern "C" int64_t mivae_rowgemm_wgrad_slabs(int64_t R, int64_t hs) {
  if (R < 1 || hs < 1 || hs > 512) return 0;
  int nslab, slab_rows;
  wgrad_geom(R, (int)hs, nslab, slab_rows);
  return nslab;
}

// part (slabs, hs, hs) fp32: part[s][o][i] = sum over the rows of slab s of dY[r][o] X[r][i]; the gradient of M in
// Y = X M^T is the sum over slabs (mivae_colsum on the (slabs, hs hs) view: fixed order).  fp32 accumulation on the
// f32 MFMA for both dtypes.
extern "C" int mivae_rowgemm_wgrad(const void* x, const void* dy, float* part, int64_t R, int64_t hs, int dtype,
                                   void* stream) {
  MIVAE_REQUIRE(R >= 1 && R < (1ll << 40), "rowgemm_wgrad: bad row count");
  if (hs < 1 || hs > 512) {
    mivae_set_error("rowgemm_wgrad: unsupported width hs=%lld (1..512)", (long long)hs);
    return MIVAE_EUNSUPPORTED;
  }
  MIVAE_REQUIRE(x && dy && part, "rowgemm_wgrad: null pointer");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "rowgemm_wgrad: dtype must be f32 or bf16");
  WgradArgs a = {x, dy, part, R, (int)hs, 0};
  int nslab;
  wgrad_geom(R, (int)hs, nslab, a.slab_rows);
  const unsigned t = (unsigned)((hs + 31) / 32);
  dim3 grid((unsigned)nslab, t, t);
  if (dtype == MIVAE_F32) k_rowgemm_wgrad<float><<<grid, 64, 0, (hipStream_t)stream>>>(a);
  else k_rowgemm_wgrad<unsigned short><<<grid, 64, 0, (hipStream_t)stream>>>(a);
  MIVAE_CHECK_LAUNCH("rowgemm_wgrad");
  return MIVAE_OK;
}
