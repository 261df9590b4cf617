// Token-parallel small Linear layers on the matrix cores: y = x W^T + b, dx = dy W, (dW, db) = (dy^T x, colsum dy)
// for many tokens (N >= 4096) and few features (32 .. 256, multiples of 32): the posterior heads
// (posterior.py:19-27: mu = Linear(E -> D), sigma = ...(Linear(E -> D))) and the q/k/v/out/FFN
// projections of the attention coupling / transformer bracket (nn/transformer.py:41-48, 129-130).
// hipBLASLt dispatches these (16384 x 64 x 64) products to a 256-workgroup 64x64 tiling at ~9 us per call; they
// are 4 MB of HBM traffic.  Here a wave owns 32 tokens, keeps their features as MFMA B fragments in registers and
// walks the output features in chunks of 32 (v_mfma_f32_32x32x16_bf16, fp32 accumulation, bf16 operands exactly
// like the library path): one read of x, one write of y.  The weight gradient contracts over tokens: a block stages
// 128 tokens of both operands row-major in LDS and reads the k = token fragments with the transposing LDS read
// (ds_read_b64_tr_b16); every block dumps its partial (dW | db) and mivae_colsum(_sink_multi) folds the partials in
// fixed order (no atomics).
#include "common.hpp"

using namespace mivae;

namespace {

typedef __bf16 lin_bf16x8 __attribute__((ext_vector_type(8)));

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
__device__ __forceinline__ int lin_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

constexpr int kMaxGroup = 4;

// A group of G Linear layers that share their input (q / k / v projections of self-attention, the mu / sigma heads
// of the posterior): forward = G outputs from one read of x in one launch; dgrad = dx = sum_g dy_g W_g as ONE
// contraction over K = G * Fo (no per-layer dx tensors, no accumulation kernels).
struct LinArgs {
  const unsigned short* x[kMaxGroup];     // fwd: x[0] = activations (N, Fi); dgrad: dy_g (N, Fo)
  const unsigned short* w[kMaxGroup];     // (Fo, Fi) bf16 row-major (torch.nn.Linear layout)
  const unsigned short* bias[kMaxGroup];  // (Fo) bf16 or null (fwd only)
  unsigned short* y[kMaxGroup];           // fwd: y_g (N, Fo); dgrad: y[0] = dx (N, Fi)
  const unsigned short* relu_src;         // dgrad only, optional (N, Fi): dx is zeroed where relu_src <= 0 (relu')
  const void* res;                        // fwd only, optional (N, Fo) residual: out = res + bf16(x W^T + b), in res's dtype
  void* out;                              //   (N, Fo) fp32 or bf16, written instead of y[0]
  int res_f32;
  int64_t N;
  int Fo, Fi, G;
};

// fwd  (TRANS = false): K = Fi, M = Fo, A fragments straight from W (rows = outputs, k = inputs)
// dgrad (TRANS = true): K = Fo, M = Fi, A[i][k] = W[k][i]: W is transposed into LDS once per block
template <int NK, bool TRANS>
__global__ __launch_bounds__(256) void k_lin(LinArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short wt[];   // TRANS: [M][K + 8]
  constexpr int K = 16 * NK;                          // fwd: Fi; dgrad: G * Fo (the whole group is one contraction)
  const int M = TRANS ? a.Fi : a.Fo;
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  if (TRANS) {
    // W_g (Fo rows, Fi = M cols) -> wt[m][g * Fo + k]; each thread moves 8 consecutive m of one k per step
    const int groups = K * (M >> 3);
    for (int g = threadIdx.x; g < groups; g += 256) {
      const int k = g / (M >> 3), m8 = (g % (M >> 3)) * 8;
      const int gi = k / a.Fo, kl = k - gi * a.Fo;
      const u16x8 v = *reinterpret_cast<const u16x8*>(a.w[gi] + (int64_t)kl * M + m8);
#pragma unroll
      for (int e = 0; e < 8; ++e) wt[(m8 + e) * (K + 8) + k] = v[e];
    }
    __syncthreads();
  }
  // one (32-token tile, 32-output chunk) unit per wave: 2-8 x more waves than tiles, the chip needs them
  const int cpg = M >> 5;                             // chunks per layer
  const int nch = TRANS ? cpg : cpg * a.G;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t tile = unit / nch;
  const int cg = (int)(unit - tile * nch);
  const int gi = TRANS ? 0 : cg / cpg;                // wave-uniform: which layer of the group (fwd)
  const int c = cg - gi * cpg;
  const int64_t tok = tile * 32 + r32;
  if (tile * 32 >= a.N) return;                       // wave-uniform, after the only barrier
  const bool valid = tok < a.N;
  lin_bf16x8 xf[NK];
  if (TRANS) {
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int k0 = 16 * ks, gk = k0 / a.Fo;
      xf[ks] = *reinterpret_cast<const lin_bf16x8*>(a.x[gk] + (valid ? tok : 0) * a.Fo + (k0 - gk * a.Fo) + 8 * hh);
    }
  } else {
    const unsigned short* xp = a.x[0] + (valid ? tok : 0) * K + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) xf[ks] = *reinterpret_cast<const lin_bf16x8*>(xp + 16 * ks);
  }
  {
    lin_bf16x8 wf[NK];
    if (TRANS) {
      const unsigned short* p = wt + (32 * c + r32) * (K + 8) + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) wf[ks] = *reinterpret_cast<const lin_bf16x8*>(p + 16 * ks);
    } else {
      const unsigned short* p = a.w[gi] + (int64_t)(32 * c + r32) * K + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) wf[ks] = *reinterpret_cast<const lin_bf16x8*>(p + 16 * ks);
    }
    const unsigned short* bias = TRANS ? nullptr : a.bias[gi];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias ? bf16_to_f32(bias[32 * c + lin_row(r, hh)]) : 0.f;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf[ks], acc, 0, 0, 0);
    if (valid) {
      unsigned short* yp = a.y[gi] + tok * M + 32 * c + 4 * hh;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u16x4 o = {f32_to_bf16(acc[4 * q]), f32_to_bf16(acc[4 * q + 1]), f32_to_bf16(acc[4 * q + 2]), f32_to_bf16(acc[4 * q + 3])};
        if (TRANS && a.relu_src) {          // the layer's input was a relu output: pass the gradient where it was > 0
          const u16x4 r = *reinterpret_cast<const u16x4*>(a.relu_src + tok * M + 32 * c + 4 * hh + 8 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (r[i] == 0 || (r[i] & 0x8000)) ? (unsigned short)0 : o[i];
        }
        if (!TRANS && a.res) {              // residual epilogue: the sum torch forms from the rounded bf16 product
          const int64_t at = tok * M + 32 * c + 4 * hh + 8 * q;
          if (a.res_f32) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(static_cast<const float*>(a.res) + at);
            const f32x4 v = {r[0] + bf16_to_f32(o[0]), r[1] + bf16_to_f32(o[1]), r[2] + bf16_to_f32(o[2]), r[3] + bf16_to_f32(o[3])};
            *reinterpret_cast<f32x4*>(static_cast<float*>(a.out) + at) = v;
          } else {
            const u16x4 r = *reinterpret_cast<const u16x4*>(static_cast<const unsigned short*>(a.res) + at);
            const u16x4 v = {f32_to_bf16(bf16_to_f32(r[0]) + bf16_to_f32(o[0])), f32_to_bf16(bf16_to_f32(r[1]) + bf16_to_f32(o[1])),
                             f32_to_bf16(bf16_to_f32(r[2]) + bf16_to_f32(o[2])), f32_to_bf16(bf16_to_f32(r[3]) + bf16_to_f32(o[3]))};
            *reinterpret_cast<u16x4*>(static_cast<unsigned short*>(a.out) + at) = v;
          }
          continue;
        }
        *reinterpret_cast<u16x4*>(yp + 8 * q) = o;
      }
    }
  }
}

struct WgArgs {
  const unsigned short* x;    // (N, Fi) bf16
  const unsigned short* dy;   // (N, Fo) bf16
  float* partial;             // (P, Fo * Fi + Fo): per wave [dW row-major | db]
  int64_t N;
  int Fo, Fi, tiles_per_wave;
};

// ---- weight gradient: no LDS transpose writes, no cross-wave reduction -----------------------------------------
// A block stages 128 tokens of dy and x ROW-major in LDS (coalesced 16-byte copies), and the MFMA fragments, which
// need 8 consecutive TOKENS of one feature, come out of gfx950's transposing LDS read (ds_read_b64_tr_b16: per
// 16-lane group a 4-row x 16-column block delivered column-major).  The 4 waves split the OUTPUT tiles of
// dW (Fo x Fi) instead of the tokens, so every wave owns its accumulators outright and writes its part of the
// block's partial row directly.  Row strides are padded to 16 (mod 64) banks: the transposed reads are conflict free.

// fragment f[j] = tile[tok0 + 8 hh + j][fbase + r32]   (A or B operand of v_mfma_f32_32x32x16_bf16, k = tokens)
__device__ __forceinline__ lin_bf16x8 lin_tr_frag(const unsigned short* tile, int stride, int fbase, int tok0, int lane) {
  return lds_tr_frag(tile, stride, fbase, tok0, lane);
}

constexpr int kSlab = 128;    // tokens staged per block and pass

template <int MO, int NI>
__device__ __forceinline__ void wgrad2_body(const WgArgs& a, unsigned short* lds, const unsigned bx) {
  constexpr int Fo = 32 * MO, Fi = 32 * NI;
  constexpr int SG = Fo + (MO == 1 ? 0 : 32), SX = Fi + (NI == 1 ? 0 : 32);   // row strides (elements)
  constexpr int T = MO * NI, TPW = (T + 3) / 4;
  unsigned short* gt = lds;                       // [kSlab][SG] dy rows
  unsigned short* xt = lds + kSlab * SG;          // [kSlab][SX] x rows
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int wave = threadIdx.x >> 6;
  f32x16 acc[TPW], accb[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accb[i][r] = 0.f; }
  const u16x8 ones_raw = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const lin_bf16x8 ones = __builtin_bit_cast(lin_bf16x8, ones_raw);
  const int slabs = a.tiles_per_wave;             // 128-token slabs per block (= 4 waves x tiles_per_wave tiles)
  for (int sl = 0; sl < slabs; ++sl) {
    const int64_t tb = ((int64_t)bx * slabs + sl) * kSlab;
    if (tb >= a.N) break;                         // block-uniform
    if (sl) __syncthreads();                      // the previous slab has been consumed
    for (int idx = threadIdx.x; idx < kSlab * (Fo >> 3); idx += 256) {
      const int tk = idx / (Fo >> 3), f8 = (idx % (Fo >> 3)) * 8;
      u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (tb + tk < a.N) v = *reinterpret_cast<const u16x8*>(a.dy + (tb + tk) * Fo + f8);
      *reinterpret_cast<u16x8*>(gt + tk * SG + f8) = v;
    }
    for (int idx = threadIdx.x; idx < kSlab * (Fi >> 3); idx += 256) {
      const int tk = idx / (Fi >> 3), f8 = (idx % (Fi >> 3)) * 8;
      u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (tb + tk < a.N) v = *reinterpret_cast<const u16x8*>(a.x + (tb + tk) * Fi + f8);
      *reinterpret_cast<u16x8*>(xt + tk * SX + f8) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int t = wave * TPW + i;
      if (t >= T) continue;                       // wave-uniform: EXEC stays all ones around the transposed reads
      const int m = t / NI, n = t - m * NI;
#pragma unroll
      for (int s = 0; s < kSlab / 16; ++s) {
        const lin_bf16x8 af = lin_tr_frag(gt, SG, 32 * m, 16 * s, lane);
        const lin_bf16x8 bf = lin_tr_frag(xt, SX, 32 * n, 16 * s, lane);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[i], 0, 0, 0);
        if (n == 0) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ones, accb[i], 0, 0, 0);
      }
    }
  }
  // partial[block][row * Fi + col] (row = output feature, col = input feature), db at [Fo * Fi + row]
  float* out = a.partial + (int64_t)bx * ((int64_t)Fo * Fi + Fo);
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = wave * TPW + i;
    if (t >= T) continue;
    const int m = t / NI, n = t - m * NI;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(32 * m + lin_row(r, hh)) * Fi + 32 * n + r32] = acc[i][r];
    if (n == 0 && r32 == 0) {                     // every column of accb holds the row sums
#pragma unroll
      for (int r = 0; r < 16; ++r) out[Fo * Fi + 32 * m + lin_row(r, hh)] = accb[i][r];
    }
  }
}

template <int MO, int NI>
__global__ __launch_bounds__(256) void k_lin_wgrad2(WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  wgrad2_body<MO, NI>(a, lds, blockIdx.x);
}

// Two independent weight-gradient problems over the same tokens in one launch (blockIdx.y picks the problem): the two
// Linear layers of a feed-forward sublayer, dW2 = dy^T hmid and dW1 = dh^T xn, otherwise two ~7 us launches each way.
template <int MO0, int NI0, int MO1, int NI1>
__global__ __launch_bounds__(256) void k_lin_wgrad2_pair(WgArgs a0, WgArgs a1) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  if (blockIdx.y == 0) wgrad2_body<MO0, NI0>(a0, lds, blockIdx.x);
  else wgrad2_body<MO1, NI1>(a1, lds, blockIdx.x);
}

// The same for a GROUP of G layers that share x (q / k / v, mu / sigma heads): x is staged once, the G dy tiles next to
// it, and the G * MO * NI output tiles are split over the 4 waves; layer g's partial row goes to its own buffer.
struct WgGroupArgs {
  const unsigned short* x;
  const unsigned short* dy[kMaxGroup];
  float* partial[kMaxGroup];
  int64_t N;
  int tiles_per_wave;
};

template <int MO, int NI, int G>
__device__ __forceinline__ void wgrad2g_body(const WgGroupArgs& a, unsigned short* lds, const unsigned bx) {
  constexpr int Fo = 32 * MO, Fi = 32 * NI;
  constexpr int SG = Fo + (MO == 1 ? 0 : 32), SX = Fi + (NI == 1 ? 0 : 32);
  constexpr int T = G * MO * NI, TPW = (T + 3) / 4;
  unsigned short* xt = lds;                          // [kSlab][SX]
  unsigned short* gt = lds + kSlab * SX;             // G x [kSlab][SG]
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int wave = threadIdx.x >> 6;
  f32x16 acc[TPW], accb[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accb[i][r] = 0.f; }
  const u16x8 ones_raw = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  const lin_bf16x8 ones = __builtin_bit_cast(lin_bf16x8, ones_raw);
  const int slabs = a.tiles_per_wave;
  for (int sl = 0; sl < slabs; ++sl) {
    const int64_t tb = ((int64_t)bx * slabs + sl) * kSlab;
    if (tb >= a.N) break;
    if (sl) __syncthreads();
    for (int idx = threadIdx.x; idx < kSlab * (Fi >> 3); idx += 256) {
      const int tk = idx / (Fi >> 3), f8 = (idx % (Fi >> 3)) * 8;
      u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (tb + tk < a.N) v = *reinterpret_cast<const u16x8*>(a.x + (tb + tk) * Fi + f8);
      *reinterpret_cast<u16x8*>(xt + tk * SX + f8) = v;
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
      for (int idx = threadIdx.x; idx < kSlab * (Fo >> 3); idx += 256) {
        const int tk = idx / (Fo >> 3), f8 = (idx % (Fo >> 3)) * 8;
        u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (tb + tk < a.N) v = *reinterpret_cast<const u16x8*>(a.dy[g] + (tb + tk) * Fo + f8);
        *reinterpret_cast<u16x8*>(gt + (g * kSlab + tk) * SG + f8) = v;
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int t = wave * TPW + i;
      if (t >= T) continue;                          // wave-uniform
      const int g = t / (MO * NI), m = (t % (MO * NI)) / NI, n = t % NI;
#pragma unroll
      for (int s = 0; s < kSlab / 16; ++s) {
        const lin_bf16x8 af = lin_tr_frag(gt + g * kSlab * SG, SG, 32 * m, 16 * s, lane);
        const lin_bf16x8 bf = lin_tr_frag(xt, SX, 32 * n, 16 * s, lane);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[i], 0, 0, 0);
        if (n == 0) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ones, accb[i], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = wave * TPW + i;
    if (t >= T) continue;
    const int g = t / (MO * NI), m = (t % (MO * NI)) / NI, n = t % NI;
    float* out = a.partial[g] + (int64_t)bx * ((int64_t)Fo * Fi + Fo);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(32 * m + lin_row(r, hh)) * Fi + 32 * n + r32] = acc[i][r];
    if (n == 0 && r32 == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) out[Fo * Fi + 32 * m + lin_row(r, hh)] = accb[i][r];
    }
  }
}

template <int MO, int NI, int G>
__global__ __launch_bounds__(256) void k_lin_wgrad2g(WgGroupArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  wgrad2g_body<MO, NI, G>(a, lds, blockIdx.x);
}

// A single layer and a group of G layers (different inputs, same tokens, all F x F) in one launch: the output projection
// of self-attention and its q / k / v projections (blockIdx.y = 0 / 1).
template <int MO, int G>
__global__ __launch_bounds__(256) void k_lin_wgrad2_sg(WgArgs a0, WgGroupArgs a1) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  if (blockIdx.y == 0) wgrad2_body<MO, MO>(a0, lds, blockIdx.x);
  else wgrad2g_body<MO, MO, G>(a1, lds, blockIdx.x);
}

// Many weight-gradient problems over the same N tokens in one launch (blockIdx.y = problem): the training step queues
// every small-Linear weight gradient of the backward pass and issues them together right before the partials are folded
// (they feed nothing else), instead of one ~7 us latency-bound launch per layer / group.  Descriptors travel by value in
// the kernel arguments (a captured graph bakes them in; no device table to keep alive).
constexpr int kMaxMulti = 16;
struct MultiWgArgs {
  struct Prob {
    const unsigned short* x;
    const unsigned short* dy[kMaxGroup];
    float* partial[kMaxGroup];
    int Fo, Fi, G;
  } p[kMaxMulti];
  int64_t N;
  int tiles_per_wave;
};

__global__ __launch_bounds__(256) void k_lin_wgrad_multi_single(MultiWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const MultiWgArgs::Prob& p = a.p[blockIdx.y];
  const WgArgs w = {p.x, p.dy[0], p.partial[0], a.N, p.Fo, p.Fi, a.tiles_per_wave};
  const int code = (p.Fo / 32) * 8 + p.Fi / 32;          // block-uniform
  switch (code) {
    case 1 * 8 + 1: wgrad2_body<1, 1>(w, lds, blockIdx.x); break;
    case 1 * 8 + 2: wgrad2_body<1, 2>(w, lds, blockIdx.x); break;
    case 2 * 8 + 1: wgrad2_body<2, 1>(w, lds, blockIdx.x); break;
    case 2 * 8 + 2: wgrad2_body<2, 2>(w, lds, blockIdx.x); break;
    case 2 * 8 + 4: wgrad2_body<2, 4>(w, lds, blockIdx.x); break;
    case 4 * 8 + 2: wgrad2_body<4, 2>(w, lds, blockIdx.x); break;
    default: break;
  }
}

__global__ __launch_bounds__(256) void k_lin_wgrad_multi_group(MultiWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const MultiWgArgs::Prob& p = a.p[blockIdx.y];
  WgGroupArgs w = {};
  w.x = p.x; w.N = a.N; w.tiles_per_wave = a.tiles_per_wave;
#pragma unroll
  for (int g = 0; g < kMaxGroup; ++g) { w.dy[g] = p.dy[g]; w.partial[g] = p.partial[g]; }
  const int code = (p.Fo / 32) * 8 + p.G;                 // square layers: Fo == Fi
  switch (code) {
    case 1 * 8 + 2: wgrad2g_body<1, 1, 2>(w, lds, blockIdx.x); break;
    case 1 * 8 + 3: wgrad2g_body<1, 1, 3>(w, lds, blockIdx.x); break;
    case 2 * 8 + 2: wgrad2g_body<2, 2, 2>(w, lds, blockIdx.x); break;
    case 2 * 8 + 3: wgrad2g_body<2, 2, 3>(w, lds, blockIdx.x); break;
    default: break;
  }
}

// ---- pre-norm feed-forward sublayer, forward, in one launch ----------------------------------------------------
// y = x + W2 relu(W1 LN(x) + b1) + b2   (nn/transformer.py:147-162 SublayerConnection around :120-144
// PositionwiseFeedForward; dropout inactive).  Everything is per token: a wave owns 32 tokens, normalises its rows in
// registers (two half-rows per token on lanes l and l + 32), runs W1 on the matrix cores, applies bias + relu on the
// accumulator tiles and feeds them straight back as the B operand of the W2 product (the accumulator layout IS a
// B-fragment layout for a permuted contraction index; W2's fragments are read with the same permutation), then adds
// the residual.  The two weight matrices are staged once per block in LDS.  The LayerNorm output, its statistics and the relu output are written once for
// the backward pass.  Intermediate roundings match the layer-by-layer path (bf16 after LN, after W1 and after W2).
struct FfnArgs {
  const void* x;                    // (N, D) fp32 or bf16 residual stream
  const float *gamma, *beta;        // (D)
  const unsigned short *w1, *b1;    // (H, D), (H) bf16
  const unsigned short *w2, *b2;    // (D, H), (D) bf16
  void* y;                          // (N, D), dtype of x
  unsigned short* xn;               // (N, D) bf16: LN(x)
  unsigned short* hmid;             // (N, H) bf16: relu(W1 xn + b1)
  float *mean, *rstd;               // (N)
  int64_t N;
  float eps;
};

template <typename TI, int ND, int NH>   // D = 16 ND, H = 32 NH
__global__ __launch_bounds__(256) void k_ffn_fwd(FfnArgs a) {
  constexpr int D = 16 * ND, H = 32 * NH, MD = D / 32;
  // both weight matrices staged once per block (the 4 waves share them); row strides padded by 16 bytes so that the
  // 32 rows a fragment read touches fall on different banks
  constexpr int S1 = D + 8, S2 = H + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned short wlds[];
  unsigned short* w1s = wlds;                 // [H][S1]
  unsigned short* w2s = wlds + H * S1;        // [D][S2]
  for (int i = threadIdx.x; i < H * (D / 8); i += 256) {
    const int r = i / (D / 8), c8 = (i % (D / 8)) * 8;
    *reinterpret_cast<u16x8*>(w1s + r * S1 + c8) = *reinterpret_cast<const u16x8*>(a.w1 + (int64_t)r * D + c8);
  }
  for (int i = threadIdx.x; i < D * (H / 8); i += 256) {
    const int r = i / (H / 8), c8 = (i % (H / 8)) * 8;
    *reinterpret_cast<u16x8*>(w2s + r * S2 + c8) = *reinterpret_cast<const u16x8*>(a.w2 + (int64_t)r * H + c8);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int64_t tok0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
  if (tok0 >= a.N) return;                                  // wave-uniform, after the only barrier
  const int64_t tok = tok0 + r32;
  const bool valid = tok < a.N;
  const TI* xrow = static_cast<const TI*>(a.x) + (valid ? tok : 0) * D;
  // half-row in B-fragment order: k-step ks holds features 16 ks + 8 hh + j
  float xv[ND][8];
  float s = 0.f;
#pragma unroll
  for (int ks = 0; ks < ND; ++ks) {
    if (sizeof(TI) == 4) {
      float t0[4], t1[4];
      RowIO<float, 4>::load(reinterpret_cast<const float*>(xrow) + 16 * ks + 8 * hh, t0);
      RowIO<float, 4>::load(reinterpret_cast<const float*>(xrow) + 16 * ks + 8 * hh + 4, t1);
#pragma unroll
      for (int j = 0; j < 4; ++j) { xv[ks][j] = t0[j]; xv[ks][4 + j] = t1[j]; }
    } else {
      RowIO<unsigned short, 8>::load(reinterpret_cast<const unsigned short*>(xrow) + 16 * ks + 8 * hh, xv[ks]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += xv[ks][j];
  }
  s += __shfl_xor(s, 32, 64);
  const float mean = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int ks = 0; ks < ND; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = xv[ks][j] - mean; q += d * d; }
  q += __shfl_xor(q, 32, 64);
  const float rstd = rsqrtf(q / (float)D + a.eps);
  if (valid && hh == 0) { a.mean[tok] = mean; a.rstd[tok] = rstd; }
  lin_bf16x8 xf[ND];
#pragma unroll
  for (int ks = 0; ks < ND; ++ks) {
    u16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = 16 * ks + 8 * hh + j;
      o[j] = f32_to_bf16(fmaf((xv[ks][j] - mean) * rstd, a.gamma[f], a.beta[f]));
    }
    xf[ks] = __builtin_bit_cast(lin_bf16x8, o);
    if (valid) *reinterpret_cast<u16x8*>(a.xn + tok * D + 16 * ks + 8 * hh) = o;
  }
  // stage 1: hidden^T (H x 32 tokens) = W1 xn^T + b1, relu, bf16
  u16x4 hb[NH][4];
#pragma unroll
  for (int t = 0; t < NH; ++t) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bf16_to_f32(a.b1[32 * t + lin_row(r, hh)]);
#pragma unroll
    for (int ks = 0; ks < ND; ++ks) {
      const lin_bf16x8 wf = *reinterpret_cast<const lin_bf16x8*>(w1s + (32 * t + r32) * S1 + 16 * ks + 8 * hh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[ks], acc, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u16x4 o = {f32_to_bf16(fmaxf(acc[4 * g], 0.f)), f32_to_bf16(fmaxf(acc[4 * g + 1], 0.f)),
                       f32_to_bf16(fmaxf(acc[4 * g + 2], 0.f)), f32_to_bf16(fmaxf(acc[4 * g + 3], 0.f))};
      hb[t][g] = o;
      if (valid) *reinterpret_cast<u16x4*>(a.hmid + tok * H + 32 * t + 8 * g + 4 * hh) = o;
    }
  }
  // stage 2: out^T (D x 32 tokens) = W2 hidden^T + b2; contraction slot j of k-step s2 of hidden tile t is hidden
  // unit 32 t + 16 s2 + 8 (j >> 2) + 4 hh + (j & 3): what the accumulator registers of stage 1 hold
#pragma unroll
  for (int m = 0; m < MD; ++m) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bf16_to_f32(a.b2[32 * m + lin_row(r, hh)]);
#pragma unroll
    for (int t = 0; t < NH; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const unsigned short* wp = w2s + (32 * m + r32) * S2 + 32 * t + 16 * s2 + 4 * hh;
        const u16x4 w0 = *reinterpret_cast<const u16x4*>(wp), w1v = *reinterpret_cast<const u16x4*>(wp + 8);
        const u16x8 araw = {w0[0], w0[1], w0[2], w0[3], w1v[0], w1v[1], w1v[2], w1v[3]};
        const u16x8 braw = {hb[t][2 * s2][0], hb[t][2 * s2][1], hb[t][2 * s2][2], hb[t][2 * s2][3],
                            hb[t][2 * s2 + 1][0], hb[t][2 * s2 + 1][1], hb[t][2 * s2 + 1][2], hb[t][2 * s2 + 1][3]};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(lin_bf16x8, araw),
                                                      __builtin_bit_cast(lin_bf16x8, braw), acc, 0, 0, 0);
      }
    if (valid) {
      TI* yrow = static_cast<TI*>(a.y) + tok * D;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f = 32 * m + 8 * g + 4 * hh;
        float xr[4], o[4];
        RowIO<TI, 4>::load(xrow + f, xr);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = xr[i] + bf16_to_f32(f32_to_bf16(acc[4 * g + i]));   // W2 output is bf16
        RowIO<TI, 4>::store(yrow + f, o);
      }
    }
  }
}

bool lin_dim_ok(int64_t f) { return f == 32 || f == 64 || f == 128 || f == 256; }

// K = contraction length (fwd: Fi; dgrad: G * Fo), M = output features per layer, units = 32-wide output chunks
template <bool TRANS>
int lin_launch(const LinArgs& a, int K, int M, hipStream_t st) {
  const int chunks = (M / 32) * (TRANS ? 1 : a.G);
  const unsigned grid = (unsigned)((((a.N + 31) / 32) * chunks + 3) / 4);
  const size_t lds = TRANS ? (size_t)M * (K + 8) * 2 : 0;
  auto big = [&](const void* fn) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  };
  switch (K) {
    case 32: big((const void*)k_lin<2, TRANS>); k_lin<2, TRANS><<<grid, 256, lds, st>>>(a); return 0;
    case 64: big((const void*)k_lin<4, TRANS>); k_lin<4, TRANS><<<grid, 256, lds, st>>>(a); return 0;
    case 96: big((const void*)k_lin<6, TRANS>); k_lin<6, TRANS><<<grid, 256, lds, st>>>(a); return 0;
    case 128: big((const void*)k_lin<8, TRANS>); k_lin<8, TRANS><<<grid, 256, lds, st>>>(a); return 0;
    case 192: big((const void*)k_lin<12, TRANS>); k_lin<12, TRANS><<<grid, 256, lds, st>>>(a); return 0;
    case 256: big((const void*)k_lin<16, TRANS>); k_lin<16, TRANS><<<grid, 256, lds, st>>>(a); return 0;
  }
  return -1;
}

// waves (= partial rows) for the weight gradient: ~1024 waves, each a whole number of 32-token tiles
void wgrad_geom(int64_t N, int& tiles_per_wave, int64_t& P) {
  const int64_t ntiles = (N + 31) / 32;
  int64_t tpw = (ntiles + 1023) / 1024;
  if (tpw < 1) tpw = 1;
  tiles_per_wave = (int)tpw;
  P = ((ntiles + tpw - 1) / tpw + 3) / 4;         // blocks of 4 waves = partial rows
}

}  // namespace

// y (N, Fo) = x (N, Fi) W^T + bias; all bf16 (bias may be NULL), fp32 accumulation.  Fi, Fo in {32, 64, 128, 256}.
extern "C" int mivae_linear_fwd(const void* x, const void* w, const void* bias, void* y, int64_t N, int64_t Fi,
                                int64_t Fo, void* stream) {
  MIVAE_REQUIRE(N >= 0 && N < (1ll << 31), "linear_fwd: bad N");
  MIVAE_REQUIRE(lin_dim_ok(Fi) && lin_dim_ok(Fo), "linear_fwd: features must be 32, 64, 128 or 256 (Fi=%lld Fo=%lld)",
                (long long)Fi, (long long)Fo);
  if (N == 0) return MIVAE_OK;
  MIVAE_REQUIRE(x && w && y, "linear_fwd: null pointer");
  MIVAE_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y), "linear_fwd: operands must be 16-byte aligned");
  LinArgs a = {};
  a.x[0] = (const unsigned short*)x; a.w[0] = (const unsigned short*)w; a.bias[0] = (const unsigned short*)bias;
  a.y[0] = (unsigned short*)y; a.N = N; a.Fo = (int)Fo; a.Fi = (int)Fi; a.G = 1;
  MIVAE_REQUIRE(lin_launch<false>(a, (int)Fi, (int)Fo, S_(stream)) == 0, "linear_fwd: no kernel instance");
  MIVAE_CHECK_LAUNCH("linear_fwd");
  return MIVAE_OK;
}

// out (N, Fo) = res + bf16(x W^T + b): the projection that closes a pre-norm residual sublayer, sum in res's dtype.
extern "C" int mivae_linear_fwd_residual(const void* x, const void* w, const void* bias, const void* res, void* out,
                                         int64_t N, int64_t Fi, int64_t Fo, int res_dtype, void* stream) {
  MIVAE_REQUIRE(N >= 0 && N < (1ll << 31), "linear_fwd_residual: bad N");
  MIVAE_REQUIRE(lin_dim_ok(Fi) && lin_dim_ok(Fo), "linear_fwd_residual: features must be 32, 64, 128 or 256 (Fi=%lld Fo=%lld)",
                (long long)Fi, (long long)Fo);
  MIVAE_REQUIRE(res_dtype == MIVAE_F32 || res_dtype == MIVAE_BF16, "linear_fwd_residual: bad residual dtype %d", res_dtype);
  if (N == 0) return MIVAE_OK;
  MIVAE_REQUIRE(x && w && res && out, "linear_fwd_residual: null pointer");
  MIVAE_REQUIRE(aligned16(x) && aligned16(w) && aligned16(res) && aligned16(out),
                "linear_fwd_residual: operands must be 16-byte aligned");
  LinArgs a = {};
  a.x[0] = (const unsigned short*)x; a.w[0] = (const unsigned short*)w; a.bias[0] = (const unsigned short*)bias;
  a.y[0] = (unsigned short*)out; a.res = res; a.out = out; a.res_f32 = res_dtype == MIVAE_F32;
  a.N = N; a.Fo = (int)Fo; a.Fi = (int)Fi; a.G = 1;
  MIVAE_REQUIRE(lin_launch<false>(a, (int)Fi, (int)Fo, S_(stream)) == 0, "linear_fwd_residual: no kernel instance");
  MIVAE_CHECK_LAUNCH("linear_fwd_residual");
  return MIVAE_OK;
}

// dx (N, Fi) = dy (N, Fo) W, W (Fo, Fi) row-major.
extern "C" int mivae_linear_dgrad(const void* dy, const void* w, void* dx, int64_t N, int64_t Fi, int64_t Fo,
                                  void* stream) {
  MIVAE_REQUIRE(N >= 0 && N < (1ll << 31), "linear_dgrad: bad N");
  MIVAE_REQUIRE(lin_dim_ok(Fi) && lin_dim_ok(Fo), "linear_dgrad: features must be 32, 64, 128 or 256 (Fi=%lld Fo=%lld)",
                (long long)Fi, (long long)Fo);
  if (N == 0) return MIVAE_OK;
  MIVAE_REQUIRE(dy && w && dx, "linear_dgrad: null pointer");
  MIVAE_REQUIRE(aligned16(dy) && aligned16(w) && aligned16(dx), "linear_dgrad: operands must be 16-byte aligned");
  LinArgs a = {};
  a.x[0] = (const unsigned short*)dy; a.w[0] = (const unsigned short*)w; a.y[0] = (unsigned short*)dx;
  a.N = N; a.Fo = (int)Fo; a.Fi = (int)Fi; a.G = 1;
  MIVAE_REQUIRE(lin_launch<true>(a, (int)Fo, (int)Fi, S_(stream)) == 0, "linear_dgrad: no kernel instance");
  MIVAE_CHECK_LAUNCH("linear_dgrad");
  return MIVAE_OK;
}

// G (2..4) Linear layers of the same shape applied to the SAME input: y_g (N, Fo) = x W_g^T + bias_g, one launch.
// w / bias / y: host arrays of G device pointers (bias entries or the bias array itself may be NULL).
extern "C" int mivae_linear_fwd_group(const void* x, const void* const* w, const void* const* bias, void* const* y,
                                      int64_t G, int64_t N, int64_t Fi, int64_t Fo, void* stream) {
  MIVAE_REQUIRE(G >= 1 && G <= kMaxGroup, "linear_fwd_group: G must be 1..%d", kMaxGroup);
  MIVAE_REQUIRE(N >= 0 && N < (1ll << 31), "linear_fwd_group: bad N");
  MIVAE_REQUIRE(lin_dim_ok(Fi) && lin_dim_ok(Fo), "linear_fwd_group: features must be 32, 64, 128 or 256 (Fi=%lld Fo=%lld)",
                (long long)Fi, (long long)Fo);
  if (N == 0) return MIVAE_OK;
  MIVAE_REQUIRE(x && w && y && aligned16(x), "linear_fwd_group: null or misaligned pointer");
  LinArgs a = {};
  a.x[0] = (const unsigned short*)x; a.N = N; a.Fo = (int)Fo; a.Fi = (int)Fi; a.G = (int)G;
  for (int g = 0; g < G; ++g) {
    MIVAE_REQUIRE(w[g] && y[g] && aligned16(w[g]) && aligned16(y[g]), "linear_fwd_group: null or misaligned pointer");
    a.w[g] = (const unsigned short*)w[g]; a.y[g] = (unsigned short*)y[g];
    a.bias[g] = bias ? (const unsigned short*)bias[g] : nullptr;
  }
  MIVAE_REQUIRE(lin_launch<false>(a, (int)Fi, (int)Fo, S_(stream)) == 0, "linear_fwd_group: no kernel instance");
  MIVAE_CHECK_LAUNCH("linear_fwd_group");
  return MIVAE_OK;
}

// dx (N, Fi) = sum_g dy_g (N, Fo) W_g: the input gradient of the group as one contraction over G * Fo (<= 256).
extern "C" int mivae_linear_dgrad_group(const void* const* dy, const void* const* w, void* dx, int64_t G, int64_t N,
                                        int64_t Fi, int64_t Fo, void* stream) {
  MIVAE_REQUIRE(G >= 1 && G <= kMaxGroup, "linear_dgrad_group: G must be 1..%d", kMaxGroup);
  MIVAE_REQUIRE(N >= 0 && N < (1ll << 31), "linear_dgrad_group: bad N");
  MIVAE_REQUIRE(lin_dim_ok(Fi) && lin_dim_ok(Fo) && G * Fo <= 256,
                "linear_dgrad_group: features must be 32, 64, 128 or 256 and G * Fo <= 256 (Fi=%lld Fo=%lld G=%lld)",
                (long long)Fi, (long long)Fo, (long long)G);
  if (N == 0) return MIVAE_OK;
  MIVAE_REQUIRE(dy && w && dx && aligned16(dx), "linear_dgrad_group: null or misaligned pointer");
  LinArgs a = {};
  a.y[0] = (unsigned short*)dx; a.N = N; a.Fo = (int)Fo; a.Fi = (int)Fi; a.G = (int)G;
  for (int g = 0; g < G; ++g) {
    MIVAE_REQUIRE(dy[g] && w[g] && aligned16(dy[g]) && aligned16(w[g]), "linear_dgrad_group: null or misaligned pointer");
    a.x[g] = (const unsigned short*)dy[g]; a.w[g] = (const unsigned short*)w[g];
  }
  MIVAE_REQUIRE(lin_launch<true>(a, (int)(G * Fo), (int)Fi, S_(stream)) == 0, "linear_dgrad_group: no kernel instance");
  MIVAE_CHECK_LAUNCH("linear_dgrad_group");
  return MIVAE_OK;
}

// mivae_linear_dgrad fused with the derivative of a relu in front of the layer: dx = (dy W) * (relu_out > 0), with
// relu_out (N, Fi) bf16 the layer's input (e.g. PositionwiseFeedForward: w_2(relu(w_1 x)), nn/transformer.py:141-144).
extern "C" int mivae_linear_dgrad_relu(const void* dy, const void* w, const void* relu_out, void* dx, int64_t N,
                                       int64_t Fi, int64_t Fo, void* stream) {
  MIVAE_REQUIRE(N >= 0 && N < (1ll << 31), "linear_dgrad_relu: bad N");
  MIVAE_REQUIRE(lin_dim_ok(Fi) && lin_dim_ok(Fo), "linear_dgrad_relu: features must be 32, 64, 128 or 256 (Fi=%lld Fo=%lld)",
                (long long)Fi, (long long)Fo);
  if (N == 0) return MIVAE_OK;
  MIVAE_REQUIRE(dy && w && dx && relu_out, "linear_dgrad_relu: null pointer");
  MIVAE_REQUIRE(aligned16(dy) && aligned16(w) && aligned16(dx) && aligned16(relu_out),
                "linear_dgrad_relu: operands must be 16-byte aligned");
  LinArgs a = {};
  a.x[0] = (const unsigned short*)dy; a.w[0] = (const unsigned short*)w; a.y[0] = (unsigned short*)dx;
  a.relu_src = (const unsigned short*)relu_out;
  a.N = N; a.Fo = (int)Fo; a.Fi = (int)Fi; a.G = 1;
  MIVAE_REQUIRE(lin_launch<true>(a, (int)Fo, (int)Fi, S_(stream)) == 0, "linear_dgrad_relu: no kernel instance");
  MIVAE_CHECK_LAUNCH("linear_dgrad_relu");
  return MIVAE_OK;
}

// rows of the partial matrix mivae_linear_wgrad writes; each row is Fo * Fi + Fo floats
extern "C" int64_t mivae_linear_wgrad_rows(int64_t N) {
  if (N <= 0) return 0;
  int tpw; int64_t P;
  wgrad_geom(N, tpw, P);
  return P;
}

// partial (rows, Fo * Fi + Fo) fp32: per-block [dy^T x | colsum dy] over its token tiles; the column sums of this
// matrix (mivae_colsum) are [dW (Fo, Fi) row-major | db (Fo)].  Fi, Fo in {32, 64, 128}.
extern "C" int mivae_linear_wgrad(const void* x, const void* dy, float* partial, int64_t N, int64_t Fi, int64_t Fo,
                                  void* stream) {
  MIVAE_REQUIRE(N >= 1 && N < (1ll << 31), "linear_wgrad: bad N");
  MIVAE_REQUIRE((Fi == 32 || Fi == 64 || Fi == 128) && (Fo == 32 || Fo == 64 || Fo == 128),
                "linear_wgrad: features must be 32, 64 or 128 (Fi=%lld Fo=%lld)", (long long)Fi, (long long)Fo);
  MIVAE_REQUIRE(x && dy && partial, "linear_wgrad: null pointer");
  MIVAE_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(partial), "linear_wgrad: operands must be 16-byte aligned");
  WgArgs a = {(const unsigned short*)x, (const unsigned short*)dy, partial, N, (int)Fo, (int)Fi, 0};
  int64_t P;
  wgrad_geom(N, a.tiles_per_wave, P);
  const unsigned grid = (unsigned)P;
  const size_t lds = (size_t)kSlab * ((Fo + (Fo == 32 ? 0 : 32)) + (Fi + (Fi == 32 ? 0 : 32))) * 2;
  hipStream_t st = S_(stream);
#define WG(MO, NI)                                                                                         \
  do {                                                                                                     \
    if (lds > 64 * 1024)                                                                                   \
      (void)hipFuncSetAttribute((const void*)k_lin_wgrad2<MO, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    k_lin_wgrad2<MO, NI><<<grid, 256, lds, st>>>(a);                                                       \
  } while (0)
  const int mo = (int)Fo / 32, ni = (int)Fi / 32;
  if (mo == 1 && ni == 1) WG(1, 1); else if (mo == 1 && ni == 2) WG(1, 2); else if (mo == 1 && ni == 4) WG(1, 4);
  else if (mo == 2 && ni == 1) WG(2, 1); else if (mo == 2 && ni == 2) WG(2, 2); else if (mo == 2 && ni == 4) WG(2, 4);
  else if (mo == 4 && ni == 1) WG(4, 1); else if (mo == 4 && ni == 2) WG(4, 2); else WG(4, 4);
#undef WG
  MIVAE_CHECK_LAUNCH("linear_wgrad");
  return MIVAE_OK;
}

// mivae_linear_wgrad for the two layers of a feed-forward sublayer in one launch: (x0, dy0) with (Fi0, Fo0) = (H, D) and
// (x1, dy1) with (Fi1, Fo1) = (D, H) over the same N tokens; instantiated for D = 64, H = 128 (MIVAE_EUNSUPPORTED
// otherwise: call mivae_linear_wgrad twice).  partial0 / partial1 as mivae_linear_wgrad writes them.
extern "C" int mivae_linear_wgrad_pair(const void* x0, const void* dy0, float* partial0, int64_t Fi0, int64_t Fo0,
                                       const void* x1, const void* dy1, float* partial1, int64_t Fi1, int64_t Fo1,
                                       int64_t N, void* stream) {
  MIVAE_REQUIRE(N >= 1 && N < (1ll << 31), "linear_wgrad_pair: bad N");
  if (!(Fi0 == 128 && Fo0 == 64 && Fi1 == 64 && Fo1 == 128)) {
    mivae_set_error("linear_wgrad_pair: unsupported shapes (%lld, %lld) / (%lld, %lld)", (long long)Fi0, (long long)Fo0,
                    (long long)Fi1, (long long)Fo1);
    return MIVAE_EUNSUPPORTED;
  }
  MIVAE_REQUIRE(x0 && dy0 && partial0 && x1 && dy1 && partial1, "linear_wgrad_pair: null pointer");
  MIVAE_REQUIRE(aligned16(x0) && aligned16(dy0) && aligned16(partial0) && aligned16(x1) && aligned16(dy1) &&
                    aligned16(partial1), "linear_wgrad_pair: operands must be 16-byte aligned");
  WgArgs a0 = {(const unsigned short*)x0, (const unsigned short*)dy0, partial0, N, (int)Fo0, (int)Fi0, 0};
  WgArgs a1 = {(const unsigned short*)x1, (const unsigned short*)dy1, partial1, N, (int)Fo1, (int)Fi1, 0};
  int64_t P;
  wgrad_geom(N, a0.tiles_per_wave, P);
  a1.tiles_per_wave = a0.tiles_per_wave;
  const size_t lds = (size_t)kSlab * ((64 + 32) + (128 + 32)) * 2;      // both problems stage a 64- and a 128-wide tile
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)k_lin_wgrad2_pair<2, 4, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_lin_wgrad2_pair<2, 4, 4, 2><<<dim3((unsigned)P, 2), 256, lds, S_(stream)>>>(a0, a1);
  MIVAE_CHECK_LAUNCH("linear_wgrad_pair");
  return MIVAE_OK;
}

// nprob (<= 16) weight-gradient problems over the same N tokens in one launch.  desc: nprob rows of 12 int64 on the HOST:
// {Fo, Fi, G, x, dy[0..4), partial[0..4)}.  group == 0: every problem is a single layer (G = 1) with (Fo, Fi) in
// {(32,32), (32,64), (64,32), (64,64), (64,128), (128,64)}; group != 0: every problem is a group of G in {2, 3} square
// layers (Fo = Fi in {32, 64}) sharing x.  Results as mivae_linear_wgrad / mivae_linear_wgrad_group write them.
// Returns MIVAE_EUNSUPPORTED (-3) if any problem is outside those shapes (nothing is launched then).
extern "C" int mivae_linear_wgrad_multi(const int64_t* desc, int64_t nprob, int64_t N, int group, void* stream) {
  MIVAE_REQUIRE(N >= 1 && N < (1ll << 31), "linear_wgrad_multi: bad N");
  MIVAE_REQUIRE(desc && nprob >= 1 && nprob <= kMaxMulti, "linear_wgrad_multi: 1 .. %d problems", kMaxMulti);
  MultiWgArgs a = {};
  a.N = N;
  int64_t P;
  wgrad_geom(N, a.tiles_per_wave, P);
  size_t lds = 0;
  for (int i = 0; i < nprob; ++i) {
    const int64_t* d = desc + 12 * i;
    const int64_t Fo = d[0], Fi = d[1], G = d[2];
    bool ok;
    if (group) ok = Fo == Fi && (Fo == 32 || Fo == 64) && (G == 2 || G == 3);
    else ok = G == 1 && ((Fo == 32 && (Fi == 32 || Fi == 64)) || (Fo == 64 && (Fi == 32 || Fi == 64 || Fi == 128)) ||
                         (Fo == 128 && Fi == 64));
    if (!ok) {
      mivae_set_error("linear_wgrad_multi: unsupported problem %d (Fo=%lld Fi=%lld G=%lld)", i, (long long)Fo, (long long)Fi,
                      (long long)G);
      return MIVAE_EUNSUPPORTED;
    }
    MultiWgArgs::Prob& p = a.p[i];
    p.Fo = (int)Fo; p.Fi = (int)Fi; p.G = (int)G;
    p.x = reinterpret_cast<const unsigned short*>(d[3]);
    MIVAE_REQUIRE(p.x && aligned16(p.x), "linear_wgrad_multi: null or misaligned x");
    for (int g = 0; g < G; ++g) {
      p.dy[g] = reinterpret_cast<const unsigned short*>(d[4 + g]);
      p.partial[g] = reinterpret_cast<float*>(d[8 + g]);
      MIVAE_REQUIRE(p.dy[g] && p.partial[g] && aligned16(p.dy[g]) && aligned16(p.partial[g]),
                    "linear_wgrad_multi: null or misaligned pointer");
    }
    const size_t so = (size_t)Fo + (Fo == 32 ? 0 : 32), si = (size_t)Fi + (Fi == 32 ? 0 : 32);
    const size_t need = group ? (size_t)kSlab * si * 2 * (size_t)(G + 1) : (size_t)kSlab * (so + si) * 2;
    if (need > lds) lds = need;
  }
  const dim3 grid((unsigned)P, (unsigned)nprob);
  hipStream_t st = S_(stream);
  if (group) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_lin_wgrad_multi_group, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_lin_wgrad_multi_group<<<grid, 256, lds, st>>>(a);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_lin_wgrad_multi_single, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_lin_wgrad_multi_single<<<grid, 256, lds, st>>>(a);
  }
  MIVAE_CHECK_LAUNCH("linear_wgrad_multi");
  return MIVAE_OK;
}

// mivae_linear_wgrad(x0, dy0) and mivae_linear_wgrad_group(x, dy[0..G)) in ONE launch, all layers F x F over the same N
// tokens: the output projection of MultiHeadAttention and its q / k / v projections.  Instantiated for F in {32, 64},
// G = 3; MIVAE_EUNSUPPORTED (-3) otherwise.
extern "C" int mivae_linear_wgrad_single_group(const void* x0, const void* dy0, float* partial0, const void* x,
                                               const void* const* dy, float* const* partial, int64_t G, int64_t N,
                                               int64_t F, void* stream) {
  MIVAE_REQUIRE(N >= 1 && N < (1ll << 31), "linear_wgrad_single_group: bad N");
  if (!(G == 3 && (F == 32 || F == 64))) {
    mivae_set_error("linear_wgrad_single_group: unsupported G=%lld F=%lld", (long long)G, (long long)F);
    return MIVAE_EUNSUPPORTED;
  }
  MIVAE_REQUIRE(x0 && dy0 && partial0 && x && dy && partial && aligned16(x0) && aligned16(dy0) && aligned16(partial0) &&
                    aligned16(x), "linear_wgrad_single_group: null or misaligned pointer");
  WgArgs a0 = {(const unsigned short*)x0, (const unsigned short*)dy0, partial0, N, (int)F, (int)F, 0};
  WgGroupArgs a1 = {};
  a1.x = (const unsigned short*)x; a1.N = N;
  for (int g = 0; g < G; ++g) {
    MIVAE_REQUIRE(dy[g] && partial[g] && aligned16(dy[g]) && aligned16(partial[g]),
                  "linear_wgrad_single_group: null or misaligned pointer");
    a1.dy[g] = (const unsigned short*)dy[g]; a1.partial[g] = partial[g];
  }
  int64_t P;
  wgrad_geom(N, a0.tiles_per_wave, P);
  a1.tiles_per_wave = a0.tiles_per_wave;
  const int pad = F == 32 ? 0 : 32;
  const size_t lds = (size_t)kSlab * (F + pad) * 2 * (size_t)(G + 1);       // the group's tiles; the single layer needs 2
  const dim3 grid((unsigned)P, 2);
  hipStream_t st = S_(stream);
  if (F == 64) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_lin_wgrad2_sg<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_lin_wgrad2_sg<2, 3><<<grid, 256, lds, st>>>(a0, a1);
  } else {
    k_lin_wgrad2_sg<1, 3><<<grid, 256, lds, st>>>(a0, a1);
  }
  MIVAE_CHECK_LAUNCH("linear_wgrad_single_group");
  return MIVAE_OK;
}

// mivae_linear_wgrad for G (2..4) same-shape layers that share x, in one launch: partial[g] (rows, Fo * Fi + Fo) as
// above for layer g.  dy / partial: host arrays of G device pointers.  Instantiated for Fi = Fo in {32, 64};
// MIVAE_EUNSUPPORTED (-3) otherwise (call mivae_linear_wgrad per layer).
extern "C" int mivae_linear_wgrad_group(const void* x, const void* const* dy, float* const* partial, int64_t G, int64_t N,
                                        int64_t Fi, int64_t Fo, void* stream) {
  MIVAE_REQUIRE(N >= 1 && N < (1ll << 31), "linear_wgrad_group: bad N");
  if (!(G >= 2 && G <= kMaxGroup && Fi == Fo && (Fi == 32 || Fi == 64))) {
    mivae_set_error("linear_wgrad_group: unsupported G=%lld Fi=%lld Fo=%lld", (long long)G, (long long)Fi, (long long)Fo);
    return MIVAE_EUNSUPPORTED;
  }
  MIVAE_REQUIRE(x && dy && partial && aligned16(x), "linear_wgrad_group: null or misaligned pointer");
  WgGroupArgs a = {};
  a.x = (const unsigned short*)x; a.N = N;
  for (int g = 0; g < G; ++g) {
    MIVAE_REQUIRE(dy[g] && partial[g] && aligned16(dy[g]) && aligned16(partial[g]),
                  "linear_wgrad_group: null or misaligned pointer");
    a.dy[g] = (const unsigned short*)dy[g]; a.partial[g] = partial[g];
  }
  int64_t P;
  wgrad_geom(N, a.tiles_per_wave, P);
  const unsigned grid = (unsigned)P;
  const int pad = Fi == 32 ? 0 : 32;
  const size_t lds = (size_t)kSlab * (Fi + pad) * 2 * (size_t)(G + 1);
  hipStream_t st = S_(stream);
#define WGG(MO, GG)                                                                                          \
  do {                                                                                                       \
    if (lds > 64 * 1024)                                                                                     \
      (void)hipFuncSetAttribute((const void*)k_lin_wgrad2g<MO, MO, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    k_lin_wgrad2g<MO, MO, GG><<<grid, 256, lds, st>>>(a);                                                    \
  } while (0)
  if (Fi == 64) { if (G == 2) WGG(2, 2); else if (G == 3) WGG(2, 3); else WGG(2, 4); }
  else { if (G == 2) WGG(1, 2); else if (G == 3) WGG(1, 3); else WGG(1, 4); }
#undef WGG
  MIVAE_CHECK_LAUNCH("linear_wgrad_group");
  return MIVAE_OK;
}

// y = x + W2 relu(W1 LN(x) + b1) + b2 for (N, D) rows in one launch (pre-norm feed-forward sublayer forward);
// x / y fp32 (x_dtype 0) or bf16 (1); gamma / beta fp32; w1 (H, D), b1 (H), w2 (D, H), b2 (D) bf16.  Also written for
// the backward pass: xn = LN(x) (N, D) bf16, hmid = relu(W1 xn + b1) (N, H) bf16, mean / rstd (N) fp32.
// Supported: D in {64, 128}, H in {128, 256}; MIVAE_EUNSUPPORTED otherwise.
extern "C" int mivae_ffn_fwd(const void* x, const float* gamma, const float* beta, const void* w1, const void* b1,
                             const void* w2, const void* b2, void* y, void* xn, void* hmid, float* mean, float* rstd,
                             int64_t N, int64_t D, int64_t H, float eps, int x_dtype, void* stream) {
  MIVAE_REQUIRE(N >= 0 && N < (1ll << 31), "ffn_fwd: bad N");
  MIVAE_REQUIRE(x_dtype == MIVAE_F32 || x_dtype == MIVAE_BF16, "ffn_fwd: x dtype must be f32 or bf16");
  if (!((D == 64 || D == 128) && (H == 128 || H == 256))) {
    mivae_set_error("ffn_fwd: unsupported shape D=%lld H=%lld (D in {64, 128}, H in {128, 256})", (long long)D, (long long)H);
    return MIVAE_EUNSUPPORTED;
  }
  if (N == 0) return MIVAE_OK;
  MIVAE_REQUIRE(x && gamma && beta && w1 && b1 && w2 && b2 && y && xn && hmid && mean && rstd, "ffn_fwd: null pointer");
  MIVAE_REQUIRE(aligned16(x) && aligned16(w1) && aligned16(w2) && aligned16(y) && aligned16(xn) && aligned16(hmid),
                "ffn_fwd: operands must be 16-byte aligned");
  FfnArgs a = {x, gamma, beta, (const unsigned short*)w1, (const unsigned short*)b1, (const unsigned short*)w2,
               (const unsigned short*)b2, y, (unsigned short*)xn, (unsigned short*)hmid, mean, rstd, N, eps};
  const unsigned grid = (unsigned)(((N + 31) / 32 + 3) / 4);
  hipStream_t st = S_(stream);
  const size_t lds = ((size_t)H * (D + 8) + (size_t)D * (H + 8)) * 2;
#define FFN(TI, ND, NH)                                                                                           \
  do {                                                                                                            \
    if (lds > 64 * 1024)                                                                                          \
      (void)hipFuncSetAttribute((const void*)k_ffn_fwd<TI, ND, NH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    k_ffn_fwd<TI, ND, NH><<<grid, 256, lds, st>>>(a);                                                             \
  } while (0)
  if (x_dtype == MIVAE_F32) {
    if (D == 64 && H == 128) FFN(float, 4, 4); else if (D == 64) FFN(float, 4, 8);
    else if (H == 128) FFN(float, 8, 4); else FFN(float, 8, 8);
  } else {
    if (D == 64 && H == 128) FFN(unsigned short, 4, 4); else if (D == 64) FFN(unsigned short, 4, 8);
    else if (H == 128) FFN(unsigned short, 8, 4); else FFN(unsigned short, 8, 8);
  }
#undef FFN
  MIVAE_CHECK_LAUNCH("ffn_fwd");
  return MIVAE_OK;
}
