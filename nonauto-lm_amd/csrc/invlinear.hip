// Invertible linear flows on the matrix cores: InvertibleLinearFlow / InvertibleMultiHeadFlow
// of vae_lm/models/base/flows/invertible_flows.py:16-83.
//
// A latent (N, D) with `heads` blocks of hs = D / heads features per token IS a row-major (R = N heads, hs) matrix, so
// both flows, both directions and the input gradient are one product  Y = X M^T  with a small square M (hs <= 512):
//   forward  M = W (F.linear, :29, :61-63);  inverse  M = W^-1 (:37-40);  d X = d Y M  ->  the same kernel on M^T.
// No mask multiply (the reference does not mask here); the log-det (slogdet(W) x token count x heads) is scalar work
// and stays with the caller.
//
//   k_rowgemm_bf16<NKS>   production path (bf16, hs a multiple of 8): a block of 4 waves owns 128 rows; every wave keeps
//                    its 32 rows of X in registers as MFMA B fragments (ONE read of X) and walks the output features
//                    in tiles of 32 whose rows of M the block streams through a double-buffered LDS ring (M is read
//                    from L2 once per 128 rows, not once per 32); v_mfma_f32_32x32x16_bf16 with the output features
//                    on the accumulator rows, so each lane stores 4 consecutive bf16 (8 bytes) per quad.
//   k_rowgemm_wgrad_bf16<WT>  d M[o][i] = sum_r dY[r][o] X[r][i] contracts over ROWS: a block stages a slab of rows of
//                    both operands row-major in LDS (16-byte copies) and reads the k = row fragments with the
//                    transposing LDS read (ds_read_b64_tr_b16); 4 waves in a 2 x 2 arrangement own a (64 WT)^2 tile
//                    of d M; row slabs are summed in fixed order by mivae_colsum (reproducible, no atomics).
//   k_rowgemm<Mma>, k_rowgemm_wgrad  fp32 latents (the checking mode: v_mfma_f32_32x32x2_f32 is bit-identical to an
//                    fmaf chain) and bf16 widths that are not a multiple of 8: one wave per 32 rows, fragments straight
//                    from global memory.
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

// ---- bf16 production kernels ------------------------------------------------------------------------------------
// hs <= 16 NKS; LDS ring: 2 x [32 output features][16 NKS + 8] bf16 (row stride padded by 16 bytes: the 32 rows a
// fragment read touches fall on different bank groups)
template <int NKS>
__global__ void __launch_bounds__(256) k_rowgemm_bf16(RowGemmArgs a) {
  constexpr int SM = 16 * NKS + 8;               // LDS row stride (elements)
  constexpr int CPR = 2 * NKS;                   // 16-byte chunks per staged row
  constexpr int STG = (32 * CPR) / 256;          // chunks per thread and tile (NKS >= 4)
  extern __shared__ __attribute__((aligned(16))) unsigned short ring[];   // [2][32][SM]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int hs = a.hs;
  const int ntiles = (hs + 31) >> 5;
  const int per = (ntiles + (int)gridDim.y - 1) / (int)gridDim.y;     // blockIdx.y takes a contiguous share of the output tiles
  const int t0 = blockIdx.y * per, t1 = min(t0 + per, ntiles);
  const unsigned short* X = static_cast<const unsigned short*>(a.x);
  const unsigned short* M = static_cast<const unsigned short*>(a.m);
  unsigned short* Y = static_cast<unsigned short*>(a.y);
  const int64_t tok = ((int64_t)blockIdx.x * 4 + wave) * 32 + r32;
  const bool valid = tok < a.R;
  // the wave's 32 rows as B fragments: xf[ks][j] = X[tok][16 ks + 8 hh + j], zero past hs
  bf16x8 xf[NKS];
  {
    const unsigned short* xr = X + (valid ? tok : 0) * hs + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (valid && 16 * ks + 8 * hh < hs) v = *reinterpret_cast<const u16x8*>(xr + 16 * ks);
      xf[ks] = __builtin_bit_cast(bf16x8, v);
    }
  }
  u16x8 stg[STG];
  auto fetch = [&](int t) {                       // rows 32 t .. 32 t + 31 of M -> registers (zero past hs either way)
#pragma unroll
    for (int j = 0; j < STG; ++j) {
      const int c = threadIdx.x + 256 * j, row = c / CPR, c8 = (c - row * CPR) * 8;
      u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (32 * t + row < hs && c8 < hs) v = *reinterpret_cast<const u16x8*>(M + (int64_t)(32 * t + row) * hs + c8);
      stg[j] = v;
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int j = 0; j < STG; ++j) {
      const int c = threadIdx.x + 256 * j, row = c / CPR, c8 = (c - row * CPR) * 8;
      *reinterpret_cast<u16x8*>(ring + (buf * 32 + row) * SM + c8) = stg[j];
    }
  };
  if (t0 >= t1) return;                           // block-uniform, before the first barrier
  fetch(t0);
  park(t0 & 1);
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    if (t + 1 < t1) fetch(t + 1);                 // global loads in flight under this tile's MFMAs
    const unsigned short* mrow = ring + ((t & 1) * 32 + r32) * SM + 8 * hh;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // all NKS k-steps, no bound check: the ring and xf are zero past hs, and a branch-free chain lets the LDS reads
    // be issued many MFMAs ahead (one wave per SIMD: nothing else hides their latency)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(mrow + 16 * ks), xf[ks], acc, 0, 0, 0);
    // accumulator rows = output features 32 t + 8 q + 4 hh + i, column = token
    if (valid) {
      unsigned short* yr = Y + tok * hs + 32 * t + 4 * hh;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (32 * t + 8 * q + 4 * hh < hs) {       // hs is a multiple of 8: a quad is inside or outside as a whole
          const u16x4 o = {f32_to_bf16(acc[4 * q]), f32_to_bf16(acc[4 * q + 1]), f32_to_bf16(acc[4 * q + 2]),
                           f32_to_bf16(acc[4 * q + 3])};
          *reinterpret_cast<u16x4*>(yr + 8 * q) = o;
        }
    }
    if (t + 1 < t1) park((t + 1) & 1);            // the other buffer: its last readers passed the barrier of tile t - 1
    __syncthreads();
  }
}

// d M partial of one row slab and one (64 WT)^2 output tile.  LDS: dY and X slabs [KR rows][64 WT + 32] bf16.
template <int WT>
__global__ void __launch_bounds__(256) k_rowgemm_wgrad_bf16(WgradArgs a) {
  constexpr int TW = 64 * WT;                    // block tile edge
  constexpr int TS = TW + 32;                    // LDS row stride (elements): 16 banks (mod 64) per row
  constexpr int KR = WT == 1 ? 128 : 64;         // rows staged per pass
  constexpr int CPR = TW / 8;                    // 16-byte chunks per staged row and operand
  __shared__ __attribute__((aligned(16))) unsigned short gt[KR * TS];
  __shared__ __attribute__((aligned(16))) unsigned short xt[KR * TS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int wo = wave >> 1, wi = wave & 1;       // the wave's (32 WT)^2 corner of the block tile
  const int hs = a.hs;
  const int o0 = blockIdx.y * TW, i0 = blockIdx.z * TW;
  const unsigned short* X = static_cast<const unsigned short*>(a.x);
  const unsigned short* G = static_cast<const unsigned short*>(a.dy);
  const int64_t lo = (int64_t)blockIdx.x * a.slab_rows;
  int64_t hi = lo + a.slab_rows;
  if (hi > a.R) hi = a.R;
  f32x16 acc[WT][WT];
#pragma unroll
  for (int m = 0; m < WT; ++m)
#pragma unroll
    for (int n = 0; n < WT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  constexpr int STG = KR * CPR / 256;            // 16-byte chunks per thread, pass and operand
  u16x8 sg[STG], sx[STG];
  auto fetch = [&](int64_t rb) {
#pragma unroll
    for (int j = 0; j < STG; ++j) {
      const int c = threadIdx.x + 256 * j, row = c / CPR, c8 = (c - row * CPR) * 8;
      u16x8 vg = {0, 0, 0, 0, 0, 0, 0, 0}, vx = {0, 0, 0, 0, 0, 0, 0, 0};
      if (rb + row < hi) {
        if (o0 + c8 < hs) vg = *reinterpret_cast<const u16x8*>(G + (rb + row) * hs + o0 + c8);
        if (i0 + c8 < hs) vx = *reinterpret_cast<const u16x8*>(X + (rb + row) * hs + i0 + c8);
      }
      sg[j] = vg;
      sx[j] = vx;
    }
  };
  if (lo < hi) fetch(lo);
  for (int64_t rb = lo; rb < hi; rb += KR) {
    if (rb != lo) __syncthreads();               // the previous pass has been consumed
#pragma unroll
    for (int j = 0; j < STG; ++j) {
      const int c = threadIdx.x + 256 * j, row = c / CPR, c8 = (c - row * CPR) * 8;
      *reinterpret_cast<u16x8*>(gt + row * TS + c8) = sg[j];
      *reinterpret_cast<u16x8*>(xt + row * TS + c8) = sx[j];
    }
    __syncthreads();
    if (rb + KR < hi) fetch(rb + KR);            // the next pass's global loads fly under this pass's MFMAs
#pragma unroll
    for (int s = 0; s < KR / 16; ++s) {
      bf16x8 af[WT], bfr[WT];
#pragma unroll
      for (int m = 0; m < WT; ++m) af[m] = lds_tr_frag(gt, TS, 32 * (WT * wo + m), 16 * s, lane);
#pragma unroll
      for (int n = 0; n < WT; ++n) bfr[n] = lds_tr_frag(xt, TS, 32 * (WT * wi + n), 16 * s, lane);
#pragma unroll
      for (int m = 0; m < WT; ++m)
#pragma unroll
        for (int n = 0; n < WT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr[n], acc[m][n], 0, 0, 0);
    }
  }
  float* P = a.part + (int64_t)blockIdx.x * hs * hs;
#pragma unroll
  for (int m = 0; m < WT; ++m)
#pragma unroll
    for (int n = 0; n < WT; ++n) {
      const int i = i0 + 32 * (WT * wi + n) + r32;
      if (i < hs) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + 32 * (WT * wo + m) + acc_row(r, hh);
          if (o < hs) P[(int64_t)o * hs + i] = acc[m][n][r];
        }
      }
    }
}

// fp32 operands (the checking mode, and the fp32 history / cotangents of the mixed-stack training path): the same slab
// scheme on v_mfma_f32_32x32x2_f32.  Rows are staged row-major with an odd pitch, so the k = row fragments
// (lane = column) are plain conflict-free LDS reads; 64 x 64 output tile per block, 32 x 32 per wave.
__global__ void __launch_bounds__(256) k_rowgemm_wgrad_f32(WgradArgs a) {
  constexpr int KR = 64, TS = 65, CPR = 16;      // rows per pass, LDS pitch (floats), 16-byte chunks per staged row
  __shared__ float gt[KR * TS];
  __shared__ float xt[KR * TS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int wo = wave >> 1, wi = wave & 1;
  const int hs = a.hs;
  const int o0 = blockIdx.y * 64, i0 = blockIdx.z * 64;
  const float* X = static_cast<const float*>(a.x);
  const float* G = static_cast<const float*>(a.dy);
  const int64_t lo = (int64_t)blockIdx.x * a.slab_rows;
  int64_t hi = lo + a.slab_rows;
  if (hi > a.R) hi = a.R;
  const bool vec = (hs & 3) == 0;                // 16-byte global loads need hs % 4 == 0 (the bases are 16-byte aligned)
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int STG = KR * CPR / 256;
  f32x4 sg[STG], sx[STG];
  auto fetch = [&](int64_t rb) {
#pragma unroll
    for (int j = 0; j < STG; ++j) {
      const int c = threadIdx.x + 256 * j, row = c / CPR, c4 = (c - row * CPR) * 4;
      f32x4 vg = {0.f, 0.f, 0.f, 0.f}, vx = {0.f, 0.f, 0.f, 0.f};
      if (rb + row < hi) {
        const float* gp = G + (rb + row) * hs + o0 + c4;
        const float* xp = X + (rb + row) * hs + i0 + c4;
        if (vec) {
          if (o0 + c4 < hs) vg = *reinterpret_cast<const f32x4*>(gp);
          if (i0 + c4 < hs) vx = *reinterpret_cast<const f32x4*>(xp);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (o0 + c4 + e < hs) vg[e] = gp[e];
            if (i0 + c4 + e < hs) vx[e] = xp[e];
          }
        }
      }
      sg[j] = vg;
      sx[j] = vx;
    }
  };
  if (lo < hi) fetch(lo);
  for (int64_t rb = lo; rb < hi; rb += KR) {
    if (rb != lo) __syncthreads();
#pragma unroll
    for (int j = 0; j < STG; ++j) {
      const int c = threadIdx.x + 256 * j, row = c / CPR, c4 = (c - row * CPR) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gt[row * TS + c4 + e] = sg[j][e];
        xt[row * TS + c4 + e] = sx[j][e];
      }
    }
    __syncthreads();
    if (rb + KR < hi) fetch(rb + KR);
#pragma unroll 8
    for (int ks = 0; ks < KR / 2; ++ks) {
      const int k = 2 * ks + hh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[k * TS + 32 * wo + r32], xt[k * TS + 32 * wi + r32], acc, 0, 0, 0);
    }
  }
  float* P = a.part + (int64_t)blockIdx.x * hs * hs;
  const int i = i0 + 32 * wi + r32;
  if (i < hs) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + 32 * wo + acc_row(r, hh);
      if (o < hs) P[(int64_t)o * hs + i] = acc[r];
    }
  }
}

// row slabs of the weight gradient (both dtypes): ~256-512 blocks of (64 WT)^2 output tiles, >= 256 rows per slab, whole
// staging passes per slab
static void wgrad_geom(int64_t R, int hs, int& nslab, int& slab_rows) {
  const int wt = hs > 256 ? 2 : 1;
  const int tb = (hs + 64 * wt - 1) / (64 * wt);
  int64_t want = ((wt == 2 ? 256 : 512) + tb * tb - 1) / (tb * tb);
  const int64_t maxs = (R + 255) / 256;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  int64_t sr = (R + want - 1) / want;
  sr = (sr + 127) / 128 * 128;
  slab_rows = (int)sr;
  nslab = (int)((R + sr - 1) / sr);
}


// ---- log |det M| and M^-T of the flows' small square matrices -------------------------------------------------------
// torch.slogdet(self._weight)[1] of invertible_flows.py:30-33 / :64-66 and its gradient d log|det M| / d M = M^-T, for up
// to kSlogMats matrices in ONE launch (one block per matrix; n <= 128).  The library route is an LU factorisation per
// matrix forward (getf2 panel kernel 70-130 us + ~15 pivot / index launches) and two triangular solves backward: at the
// sizes of a flow list that is more GPU time and ten times more launches than the flows themselves.  Here: Gauss-Jordan
// with partial (row) pivoting on the augmented image [M | I] in LDS, fp32; pivot rows are neither swapped nor scaled
// in place (a used-row flag and the pivot values are kept instead), so a step is: column scan + arg-max (first wave) ->
// barrier -> rank-1 update of every other row -> barrier.
constexpr int kSlogMats = 24;
constexpr int kSlogMaxN = 128;
struct SlogArgs {
  const float* m[kSlogMats];
  float* invT[kSlogMats];       // may hold nulls (log-determinant only)
  int n[kSlogMats];
  float* logdet;                // [count]
};

__global__ void __launch_bounds__(1024) k_slogdet_inv(SlogArgs a) {
  extern __shared__ __attribute__((aligned(16))) float A[];     // [n][2 n + 1]
  __shared__ float fac[kSlogMaxN], pivs[kSlogMaxN];
  __shared__ int used[kSlogMaxN], perm[kSlogMaxN];
  __shared__ int s_p;
  const int n = a.n[blockIdx.x], LD = 2 * n + 1;
  const float* __restrict__ M = a.m[blockIdx.x];
  for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
    const int i = idx / n, j = idx - i * n;
    A[i * LD + j] = M[idx];
    A[i * LD + n + j] = i == j ? 1.f : 0.f;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) used[i] = 0;
  float logdet = 0.f;                                            // thread 0's copy is the result
  bool singular = false;
  const int nw = blockDim.x >> 6;                                // waves: 4 (n <= 64) or 16
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    if (threadIdx.x < 64) {                                      // first wave: column k of the unused rows, arg-max |.|
      const int lane = threadIdx.x;
      float best = -1.f;
      int bi = 0x7fffffff;
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const int t = lane + 64 * rep;
        if (t < n) {
          const float v = A[t * LD + k];
          fac[t] = v;
          const float av = used[t] ? -1.f : (fabsf(v) >= 0.f ? fabsf(v) : 3.0e38f);   // a NaN is taken (and propagates)
          if (av > best) { best = av; bi = t; }                  // strict: the lower row wins ties
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) {
        s_p = bi;
        used[bi] = 1;
        perm[k] = bi;
        pivs[k] = fac[bi];
        logdet += logf(best);
        singular = singular || best == 0.f;
      }
    }
    __syncthreads();
    const int p = s_p;
    const float inv = 1.0f / pivs[k];
    // every other row i: A[i][j] -= (A[i][k] / piv) A[p][j] over the n columns that can be non-zero and are still read:
    // the left block's j > k and, in the right block, the identity columns of the rows used so far (B = the product of
    // the row operations: a row that was never a pivot row still holds only its own 1).  Lanes own columns, waves rows.
    {
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nleft = n - k - 1;
      for (int c0 = 0; c0 < n; c0 += 64) {
        const int slot = c0 + lane;
        if (slot < n) {
          const int j = slot < nleft ? k + 1 + slot : n + perm[slot - nleft];
          const float pr = A[p * LD + j] * inv;
          for (int i0 = w; i0 < n; i0 += 8 * nw) {               // eight rows per trip: the LDS reads are in flight together
            float v[8], f[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int i = i0 + u * nw;
              if (i < n) { v[u] = A[i * LD + j]; f[u] = fac[i]; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int i = i0 + u * nw;
              if (i < n && i != p) A[i * LD + j] = v[u] - f[u] * pr;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) a.logdet[blockIdx.x] = singular ? -INFINITY : logdet;   // torch.slogdet: -inf
  float* __restrict__ out = a.invT[blockIdx.x];
  if (out) {
    // [M | I] -> [P | B] with P[perm[c]][c] = piv_c: M^-1[c][:] = B[perm[c]][:] / piv_c, so M^-T[r][c] = B[perm[c]][r] / piv_c
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
      const int r = idx / n, c = idx - r * n;
      out[idx] = A[perm[c] * LD + n + r] / pivs[c];
    }
  }
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
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIVAE_F32) {
    k_rowgemm<MmaF32, float><<<grid, 256, 0, st>>>(a);
  } else if ((hs & 7) == 0 && aligned16(y)) {
    const int nks = hs <= 64 ? 4 : hs <= 128 ? 8 : hs <= 256 ? 16 : 32;
    const size_t lds = (size_t)2 * 32 * (16 * nks + 8) * sizeof(unsigned short);
    // two resident blocks per CU hide each other's barriers: split the output tiles over blockIdx.y while the row blocks
    // alone would leave CUs with a single block (X is then read once per split, the second time from L2 / MALL)
    const int ntiles = (int)((hs + 31) / 32);
    int splits = (int)((512 + grid - 1) / grid);
    if (splits > ntiles) splits = ntiles;
    if (splits > 4) splits = 4;
    const dim3 g2(grid, (unsigned)splits);
    switch (nks) {
      case 4: k_rowgemm_bf16<4><<<g2, 256, lds, st>>>(a); break;
      case 8: k_rowgemm_bf16<8><<<g2, 256, lds, st>>>(a); break;
      case 16: k_rowgemm_bf16<16><<<g2, 256, lds, st>>>(a); break;
      default: {
        static const hipError_t once = hipFuncSetAttribute((const void*)k_rowgemm_bf16<32>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)once;
        k_rowgemm_bf16<32><<<g2, 256, lds, st>>>(a);
      }
    }
  } else {
    k_rowgemm<MmaBF16, unsigned short><<<grid, 256, 0, st>>>(a);
  }
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
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIVAE_BF16 && (hs & 7) == 0 && aligned16(x) && aligned16(dy)) {
    const int wt = hs > 256 ? 2 : 1;
    const unsigned tb = (unsigned)((hs + 64 * wt - 1) / (64 * wt));
    dim3 grid((unsigned)nslab, tb, tb);
    if (wt == 2) k_rowgemm_wgrad_bf16<2><<<grid, 256, 0, st>>>(a);
    else k_rowgemm_wgrad_bf16<1><<<grid, 256, 0, st>>>(a);
  } else if (dtype == MIVAE_F32 && aligned16(x) && aligned16(dy)) {
    const unsigned tb = (unsigned)((hs + 63) / 64);
    dim3 grid((unsigned)nslab, tb, tb);
    k_rowgemm_wgrad_f32<<<grid, 256, 0, st>>>(a);
  } else {
    const unsigned t = (unsigned)((hs + 31) / 32);
    dim3 grid((unsigned)nslab, t, t);
    if (dtype == MIVAE_F32) k_rowgemm_wgrad<float><<<grid, 64, 0, st>>>(a);
    else k_rowgemm_wgrad<unsigned short><<<grid, 64, 0, st>>>(a);
  }
  MIVAE_CHECK_LAUNCH("rowgemm_wgrad");
  return MIVAE_OK;
}

// logdet[i] = log |det M_i| and, where inv_t[i] is not NULL, inv_t[i] = M_i^-T (the gradient of logdet[i] with respect to
// M_i), for `count` <= 24 contiguous fp32 square matrices of n_i <= 128 rows, one launch.  mats / inv_t / n are HOST
// arrays (device pointers / sizes), read during the call.  A singular matrix gives -inf like torch.slogdet.
extern "C" int mivae_slogdet_inv(const void* const* mats, const int64_t* n, int64_t count, float* logdet,
                                 void* const* inv_t, void* stream) {
  MIVAE_REQUIRE(count >= 0 && count <= kSlogMats, "slogdet_inv: 0..%d matrices, got %lld", kSlogMats, (long long)count);
  if (count == 0) return MIVAE_OK;
  MIVAE_REQUIRE(mats && n && logdet, "slogdet_inv: null pointer");
  SlogArgs a = {};
  int nmax = 0;
  for (int i = 0; i < count; ++i) {
    if (n[i] < 1 || n[i] > kSlogMaxN) {
      mivae_set_error("slogdet_inv: matrix %d has %lld rows (1..%d)", i, (long long)n[i], kSlogMaxN);
      return MIVAE_EUNSUPPORTED;
    }
    MIVAE_REQUIRE(mats[i], "slogdet_inv: null matrix %d", i);
    a.m[i] = static_cast<const float*>(mats[i]);
    a.invT[i] = inv_t ? static_cast<float*>(inv_t[i]) : nullptr;
    a.n[i] = (int)n[i];
    if (a.n[i] > nmax) nmax = a.n[i];
  }
  a.logdet = logdet;
  const size_t lds = (size_t)nmax * (2 * nmax + 1) * sizeof(float);
  if (lds > 48 * 1024) {
    static const hipError_t once = hipFuncSetAttribute((const void*)k_slogdet_inv, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)((size_t)kSlogMaxN * (2 * kSlogMaxN + 1) * sizeof(float)));
    (void)once;
  }
  k_slogdet_inv<<<(unsigned)count, nmax <= 64 ? 256 : 1024, lds, (hipStream_t)stream>>>(a);
  MIVAE_CHECK_LAUNCH("slogdet_inv");
  return MIVAE_OK;
}
