// Fused NICE (affine-coupling) flow stack for gfx950: K coupling steps with feed-forward s/t nets
// (one hidden layer) in ONE launch -- forward / inverse / backward -- plus the split-K weight-gradient
// kernel that consumes the backward's token-contiguous side outputs.
//
// Reference semantics: NICECoupling.forward/.backward (vae_lm/models/base/flows/couplings.py:141-183)
// with FeedForwardCoupling nets (couplings.py:50-51, nn/feedforward.py:93-98), iterated by
// FlowPrior.forward_pass / backward_pass (nonauto_lm/priors/flow_prior.py:37-70), optionally followed
// by the base log-prob (priors/prior.py:148-158).
//
// Design (MI355X-first):
//  * a workgroup is W waves (1..8); each wave64 owns 32 consecutive tokens for ALL K steps: z is read
//    once and written once.  The reference never swaps halves (couplings.py:185-188), so
//    x0 = z[..., :D/2] is the B operand of every first-layer product of every step: it is loaded once
//    and stays in registers; only x1 evolves.
//  * everything is computed transposed (features x tokens) on v_mfma_f32_32x32x16_bf16 (bf16 mode) or
//    v_mfma_f32_32x32x2_f32 (exact fp32 mode):  a^T = W1 . x0^T -> act -> s^T = W2 . hid^T.  The first
//    product's 32x32 accumulator (token on the lane, hidden unit in the registers) is converted in
//    registers into the B operand of the second product: activations never touch LDS.
//  * weights are the streamed operand.  They are pre-packed once per call (k_nice_pack) into "task"
//    regions -- one per (step, net, 32-wide hidden chunk) -- in exact MFMA-fragment order, and streamed
//    global -> LDS with global_load_lds (16 B/lane, lane-linear 1 KiB pieces) into a double buffer:
//    the region of task i+1 is in flight while the W waves of the block read task i's fragments with
//    conflict-free ds_read_b128.  One barrier per task; biases come in through the scalar cache so
//    that no ordinary vector load sits between the LDS-DMA issue and its wait.
//  * masked log-det: per-lane partial over the registers, one __shfl_xor across the two half-waves,
//    fp32 throughout; per-token value out, per-sentence sum is mivae_segment_sum (fixed order).
//  * backward recomputes s from x0, walks the affine chain in reverse with the saved per-step x1,
//    produces dz in the same launch and emits token-contiguous ([feature][token]) ds/dt, hidden, d(pre-
//    activation) and x0 tensors; k_dw_partial/k_dw_reduce turn those into dW1/db1/dW2/db2 with
//    K-contiguous NT MFMA products, split over token slabs and reduced in fixed order (no atomics).
#include "nice_common.hpp"
#include <stdlib.h>

using namespace mivae;

namespace {


// Pack kernel: ptrs[k][8] = {W1s, b1s, W2s, b2s, W1t, b1t, W2t, b2t} (fp32, torch.nn.Linear layout:
// W1 (H, h), W2 (h, H)).  grid = (ceil(slots per step / 256), 1, K consumption steps).
template <class P>
__global__ void k_nice_pack(const int64_t* __restrict__ ptrs, char* __restrict__ packed, Geom g, int inverse) {
  const int kk = blockIdx.z;
  const int htns = g.HT * g.ns;
  const int64_t slots_f = g.nfe_f + 256, slots_b = g.nfe_b + 256;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * g.HC * slots_f + 3 * g.HC * slots_b) return;
  int k, net, c, content;   // content 0 = forward fragments, 1 = backward fragments
  int64_t slot, nfe;
  char* base;
  if (idx < 2 * g.HC * slots_f) {
    const int tq = (int)(idx / slots_f);
    slot = idx % slots_f;
    c = tq / 2;
    net = tq % 2;
    k = inverse ? g.K - 1 - kk : kk;
    content = 0; nfe = g.nfe_f;
    base = packed + ((int64_t)(kk * g.HC + c) * 2 + net) * g.tbf;
  } else {
    idx -= 2 * g.HC * slots_f;
    const int tq = (int)(idx / slots_b);
    slot = idx % slots_b;
    const int ph = tq / g.HC;
    c = tq % g.HC;
    k = g.K - 1 - kk;
    net = ph == 2 ? 1 : 0;
    content = ph == 0 ? 0 : 1; nfe = g.nfe_b;
    base = packed + g.off_b + ((int64_t)(kk * 3 + ph) * g.HC + c) * g.tbb;
  }
  const float* W1 = reinterpret_cast<const float*>(ptrs[k * 8 + net * 4 + 0]);
  const float* b1 = reinterpret_cast<const float*>(ptrs[k * 8 + net * 4 + 1]);
  const float* W2 = reinterpret_cast<const float*>(ptrs[k * 8 + net * 4 + 2]);
  const float* b2 = reinterpret_cast<const float*>(ptrs[k * 8 + net * 4 + 3]);
  if (slot >= nfe) {   // tail
    const int t = (int)(slot - nfe);
    float v = 0.f;
    if (t < 32) { if (32 * c + t < g.H) v = b1[32 * c + t]; }
    else if (t - 32 < g.h) v = b2[t - 32];
    reinterpret_cast<float*>(base + nfe * g.esize)[t] = v;
    return;
  }
  int frag = (int)(slot / g.fe);
  const int e = (int)(slot % g.fe);
  const int lane = e / P::EPF, j = e % P::EPF;
  const int r = lane & 31, hh = lane >> 5;
  float v = 0.f;
  if (frag < g.nks) {  // W1n: W1[32c + r][nat k]
    const int row = 32 * c + r, col = P::nat_k(frag, hh, j);
    if (row < g.H && col < g.h) v = W1[(int64_t)row * g.h + col];
  } else {
    frag -= g.nks;
    const int which = frag / htns;  // forward: 0 = W2p (1 = padding); backward: 0 = W2Tp, 1 = W1Tp
    frag %= htns;
    const int t = frag / g.ns, s = frag % g.ns;
    const int pk = P::perm_k(s, hh, j);
    if (content == 0) {       // W2p : W2[32t + r][32c + pk]
      const int row = 32 * t + r, col = 32 * c + pk;
      if (which == 0 && row < g.h && col < g.H) v = W2[(int64_t)row * g.H + col];
    } else if (which == 0) {  // W2Tp : A[row = hidden 32c + r][k = out 32t + pk] = W2[32t + pk][32c + r]
      const int row = 32 * t + pk, col = 32 * c + r;
      if (row < g.h && col < g.H) v = W2[(int64_t)row * g.H + col];
    } else {                  // W1Tp : A[row = in 32t + r][k = hidden 32c + pk] = W1[32c + pk][32t + r]
      const int row = 32 * c + pk, col = 32 * t + r;
      if (row < g.H && col < g.h) v = W1[(int64_t)row * g.h + col];
    }
  }
  reinterpret_cast<typename P::E*>(base)[slot] = P::enc(v);
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_cvoid_t;

// Asynchronous global -> LDS copy of `bytes` (multiple of 1 KiB), split over the W waves.
__device__ __forceinline__ void stage_region(const char* src, char* dst, int bytes, int wave, int W, int lane) {
  const int pieces = bytes >> 10;
  for (int p = wave; p < pieces; p += W)
    __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(src + ((int64_t)p << 10) + lane * 16),
                                     (lds_void_t*)(dst + (p << 10)), 16, 0, 0);
}

// Double-buffered sequential stream of task regions: block b (G tasks) lives in LDS buffer b & 1; the
// next block is in flight (LDS-DMA) while the current one is consumed.  One barrier per block.
struct Stream {
  const char* src;   // global stream base
  char* lds;         // two buffers of G * tb bytes
  int tb, G, ntasks, blk, g;
  const char* cur;
  int wave, W, lane;

  __device__ __forceinline__ void init(const char* s, char* l, int tb_, int G_, int n, int wave_, int W_,
                                       int lane_) {
    src = s; lds = l; tb = tb_; G = G_; ntasks = n; blk = 0; g = G_; cur = l;
    wave = wave_; W = W_; lane = lane_;
    stage_region(src, lds, (n < G_ ? n : G_) * tb_, wave, W, lane);
  }
  // LDS address of the next task region (call once per task, by every wave of the block)
  __device__ __forceinline__ const char* next() {
    if (g == G) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // block `blk` has landed for every wave; the other buffer is free again
      const int first = (blk + 1) * G;
      if (first < ntasks) {
        const int cnt = ntasks - first < G ? ntasks - first : G;
        stage_region(src + (int64_t)first * tb, lds + ((blk + 1) & 1) * G * tb, cnt * tb, wave, W, lane);
      }
      cur = lds + (blk & 1) * G * tb;
      ++blk;
      g = 0;
    }
    const char* p = cur + g * tb;
    ++g;
    return p;
  }
};

// LLVM SchedGroupMask bits for __builtin_amdgcn_sched_group_barrier
#define SG_MFMA 0x008
#define SG_DS_READ 0x100

// Fragment sets are read from LDS into registers in one burst (the compiler otherwise sinks every
// ds_read next to its MFMA and serialises LDS latency with the matrix pipe).
template <class P, int N>
__device__ __forceinline__ void load_frags(const typename P::AF* w, typename P::AF (&f)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) f[i] = w[i * 64];
}

// acc[t] += A[t][s] . Bfrag_s(x) for all t; A fragment (t, s) is f[t * NS + s]
template <class P, int HT, int s>
struct AccumFromAcc {
  static __device__ __forceinline__ void run(const typename P::AF (&f)[HT * P::NS], const f32x16& x,
                                             f32x16 (&acc)[HT]) {
    typename P::BF b = P::template from_acc<s>(x);
#pragma unroll
    for (int t = 0; t < HT; ++t) acc[t] = P::mma(f[t * P::NS + s], b, acc[t]);
    if constexpr (s + 1 < P::NS) AccumFromAcc<P, HT, s + 1>::run(f, x, acc);
  }
};
// out += sum_t A[t][s] . Bfrag_s(y[t])
template <class P, int HT, int s>
struct ReduceFromAcc {
  static __device__ __forceinline__ void run(const typename P::AF (&f)[HT * P::NS], const f32x16 (&y)[HT],
                                             f32x16& out) {
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      typename P::BF b = P::template from_acc<s>(y[t]);
      out = P::mma(f[t * P::NS + s], b, out);
    }
    if constexpr (s + 1 < P::NS) ReduceFromAcc<P, HT, s + 1>::run(f, y, out);
  }
};

// one forward task: acc += W2[:, c] . act(b1[c] + W1[c] . x0^T)
template <class P, int ACT, int HT, int NKS>
__device__ __forceinline__ void fwd_chunk(const char* buf, const float* tail, int act, int lane, int hh,
                                          const typename P::BF (&x0f)[NKS], f32x16 (&acc)[HT]) {
  const typename P::AF* w1 = reinterpret_cast<const typename P::AF*>(buf) + lane;
  typename P::AF f1[NKS], f2[HT * P::NS];
  f32x16 a = bias_tile(tail, hh);
  load_frags<P, NKS>(w1, f1);
  load_frags<P, HT * P::NS>(w1 + NKS * 64, f2);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) a = P::mma(f1[ks], x0f[ks], a);
  // issue order: every LDS read first (bias, W1 frags, W2 frags -- LDS returns in order, so the first
  // MFMA only waits for its own operands), then the first-layer MFMA chain
  __builtin_amdgcn_sched_group_barrier(SG_DS_READ, NKS + HT * P::NS + 4, 0);
  __builtin_amdgcn_sched_group_barrier(SG_MFMA, NKS, 0);
  act_tile<ACT, P::KS == 16>(act, a);
  AccumFromAcc<P, HT, 0>::run(f2, a, acc);
}

// fold one net's output (acc + b2) into the parked x1 tiles; returns the lane's sum of s
//   MODE 0: x1 *= exp(s)        MODE 1: x1 = (x1 + t) * m        (forward,  couplings.py:151-160)
//   MODE 2: x1 -= t             MODE 3: x1 = x1 * exp(-s) * m    (inverse,  couplings.py:172-181)
template <int MODE, int HT>
__device__ __forceinline__ float fold_tiles(float* stash, const float* b2, int hh, float m, const f32x16 (&acc)[HT]) {
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    const f32x16 bt = bias_tile(b2 + 32 * t, hh);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 x = *reinterpret_cast<const f32x4*>(stash + t * 1024 + q * 256);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[t][4 * q + e] + bt[4 * q + e];
        if constexpr (MODE == 0) { x[e] *= fast_exp(v); sum += v; }
        else if constexpr (MODE == 1) x[e] = (x[e] + v) * m;
        else if constexpr (MODE == 2) x[e] -= v;
        else { x[e] = (x[e] * fast_exp(-v)) * m; sum += v; }
      }
      *reinterpret_cast<f32x4*>(stash + t * 1024 + q * 256) = x;
    }
  }
  return sum;
}

template <typename T, bool VEC, int NKS, class P>
__device__ __forceinline__ void load_x0(const T* zrow, bool ok, int h, int hh, typename P::BF (&x0f)[NKS]) {
  if constexpr (P::KS == 2) load_x0_f32<T, VEC, NKS>(zrow, ok, h, hh, x0f);
  else load_x0_bf16<T, VEC, NKS>(zrow, ok, h, hh, x0f);
}

// Forward / inverse kernel.  A block is Wt token tiles x 2 roles: for every 32-token tile, wave role 0
// runs the s-net and wave role 1 the t-net of each step concurrently (they only depend on x0); x1 lives
// in a per-tile LDS area that the two roles update in turn (exp(s) scaling, then + t; reversed for
// the inverse), so the tile offers two independent instruction streams to its SIMD.
template <class P, typename T, int HT, bool VEC, int LB, int ACT>
__global__ void __launch_bounds__(LB) k_stack_fwd(StackArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int hp = HT * 32;
  constexpr int NKS = hp / P::KS;
  const int lane = threadIdx.x & 63;
  const int r32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = blockDim.x >> 6;
  const int tile = wave >> 1, role = wave & 1;
  const Geom& g = a.g;
  const int h = g.h;
  const int64_t tok = ((int64_t)blockIdx.x * (W >> 1) + tile) * 32 + r32;
  const bool valid = tok < a.N;
  float m = 0.f;
  if (valid) {
    m = 1.f;
    if (a.mask) {
      const int64_t bn = tok / a.S;
      m = (float)a.mask[(bn / a.mask_div) * a.S + (tok - bn * a.S)];
    }
  }
  // a tile whose tokens are all padding (or out of range) produces zeros: skip its arithmetic
  const bool live = __any(m != 0.f) != 0;
  const T* zrow = reinterpret_cast<const T*>(a.z_in) + tok * a.D;
  // masked tokens are computed on zeros: every output of such a token is multiplied by the mask
  // anyway (couplings.py:159-161), and zeros keep exp() finite at padding positions.
  typename P::BF x0f[NKS];
  load_x0<T, VEC, NKS, P>(zrow, valid && m != 0.f, h, hh, x0f);
  float ld = 0.f;

  const int HC = g.HC;
  const int tbf = (int)g.tbf;
  Stream st;
  st.init(a.packed, smem, 2 * tbf, a.G, a.K * HC, wave, W, lane);
  // x1 (the half the steps transform) is parked in a per-tile LDS area shared by the two roles
  float* stash = reinterpret_cast<float*>(smem + 2 * a.G * 2 * tbf) + tile * HT * 1024 + lane * 4;
  if (role == 0) {
    f32x16 x1[HT];
    load_acc<T, VEC, HT>(zrow + h, valid && m != 0.f, h, hh, x1);
    stash_put<HT>(stash, x1);
  }
  __syncthreads();
  if (!live) {
    // dead tile: keep the block's barrier / staging cadence, no arithmetic
    for (int kk = 0; kk < a.K; ++kk) {
      for (int c = 0; c < HC; ++c) (void)st.next();
      __syncthreads();
      __syncthreads();
    }
  } else {
    f32x16 acc[HT];
    for (int kk = 0; kk < a.K; ++kk) {
      const int k = a.inverse ? a.K - 1 - kk : kk;
      if (a.x1_hist && role == 0) {
        f32x16 x1[HT];
        stash_get<HT>(stash, x1);
        store_acc<T, VEC, HT>(reinterpret_cast<T*>(a.x1_hist) + ((int64_t)k * a.N + tok) * h, valid, h, hh, x1);
      }
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      const char* buf = nullptr;
#pragma unroll 1
      for (int c = 0; c < HC; ++c) {
        buf = st.next() + role * tbf;
        fwd_chunk<P, ACT, HT, NKS>(buf, reinterpret_cast<const float*>(buf + tbf - kTailBytes), a.act, lane, hh, x0f, acc);
      }
      // fold the nets' outputs (+ b2) into x1: forward x1 = (exp(s) x1 + t) m, inverse x1 = (x1 - t) exp(-s) m
      const float* b2 = reinterpret_cast<const float*>(buf + tbf - kTailBytes) + 32;
      if (!a.inverse) {
        if (role == 0) ld += fold_tiles<0, HT>(stash, b2, hh, m, acc);
        __syncthreads();
        if (role == 1) fold_tiles<1, HT>(stash, b2, hh, m, acc);
        __syncthreads();
      } else {
        if (role == 1) fold_tiles<2, HT>(stash, b2, hh, m, acc);
        __syncthreads();
        if (role == 0) ld -= fold_tiles<3, HT>(stash, b2, hh, m, acc);
        __syncthreads();
      }
    }
  }
  if (role != 0) return;   // role 0 owns the tile's outputs (no barriers below)
  f32x16 x1[HT];
  stash_get<HT>(stash, x1);

  // outputs ------------------------------------------------------------------------------------------
  T* orow = reinterpret_cast<T*>(a.z_out) + tok * a.D;
  store_acc<T, VEC, HT>(orow + h, valid, h, hh, x1);
  float base = 0.f;
  // x0 half: out = x0 * m (x0f is already masked), in fragment layout
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if constexpr (P::KS == 2) {
      const int f = 2 * ks + hh;
      const float v = x0f[ks];
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
// one backward task of one net: given d(out) (accumulator tiles, rows = out features) recompute
// hid / act', d hid = W2^T d(out), d pre = d hid * act', emit hid and d pre token-contiguous, and
// accumulate dx0 += W1^T d pre.
template <class P, int ACT, typename T, int HT, int NKS>
__device__ __forceinline__ void bwd_chunk(const char* buf, const float* tail, int act, int lane, int hh,
                                          const typename P::BF (&x0f)[NKS], const f32x16 (&dout)[HT],
                                          f32x16 (&dx0)[HT], T* __restrict__ Gt, T* __restrict__ Ht,
                                          int row0, unsigned loff) {
  const typename P::AF* w1 = reinterpret_cast<const typename P::AF*>(buf) + lane;
  typename P::AF f1[NKS], f2[HT * P::NS], f3[HT * P::NS];
  load_frags<P, NKS>(w1, f1);
  load_frags<P, HT * P::NS>(w1 + NKS * 64, f2);
  load_frags<P, HT * P::NS>(w1 + (NKS + HT * P::NS) * 64, f3);
  f32x16 hid = bias_tile(tail, hh);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) hid = P::mma(f1[ks], x0f[ks], hid);
  __builtin_amdgcn_sched_group_barrier(SG_DS_READ, NKS + 2 * HT * P::NS + 4, 0);
  __builtin_amdgcn_sched_group_barrier(SG_MFMA, NKS, 0);
  f32x16 dh;
#pragma unroll
  for (int r = 0; r < 16; ++r) dh[r] = 0.f;
  ReduceFromAcc<P, HT, 0>::run(f2, dout, dh);   // independent of the activation: overlaps its VALU work
  f32x16 dact;
  act_tile_fd<ACT, P::KS == 16>(act, hid, dact);
  store_tile_t<T>(Ht, row0, loff, hid);
#pragma unroll
  for (int r = 0; r < 16; ++r) dh[r] *= dact[r];
  store_tile_t<T>(Gt, row0, loff, dh);
  AccumFromAcc<P, HT, 0>::run(f3, dh, dx0);
}

template <class P, typename T, int HT, bool VEC, int LB, int ACT>
__global__ void __launch_bounds__(LB) k_stack_bwd(StackArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int hp = HT * 32;
  constexpr int NKS = hp / P::KS;
  const int lane = threadIdx.x & 63;
  const int r32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = blockDim.x >> 6;
  const Geom& g = a.g;
  const int h = g.h;
  const int64_t tok0 = ((int64_t)blockIdx.x * W + wave) * 32;
  const int64_t tok = tok0 + r32;
  const bool active = tok0 < a.Npad;   // wave-uniform; tiles in [N, Npad) still zero-fill emitted columns
  const bool valid = tok < a.N;
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
  load_x0<T, VEC, NKS, P>(zrow, valid && m != 0.f, h, hh, x0f);
  const int64_t ldn = a.Npad;
  // per-lane parts of the token-blocked side-output addresses (sections with Hp / hp rows)
  constexpr int TB = 16 / sizeof(T);
  const unsigned tblk = (unsigned)(tok / TB), tin = (unsigned)(tok % TB);
  const unsigned loffH = tblk * (unsigned)(g.Hp * TB) + tin + 4 * hh * TB;
  const unsigned loffh = tblk * (unsigned)(g.hp * TB) + tin + 4 * hh * TB;
  if (active) {  // x0^T for the dW1 products
    T* xt = reinterpret_cast<T*>(a.x0t) + (tblk * (unsigned)(g.hp * TB) + tin);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if constexpr (P::KS == 2) {
        float t1[1] = {x0f[ks]};
        RowIO<T, 1>::store(xt + (2 * ks + hh) * TB, t1);
      } else {
        u16x8 raw = __builtin_bit_cast(u16x8, x0f[ks]);
#pragma unroll
        for (int j = 0; j < 8; ++j) ((unsigned short*)xt)[(16 * ks + 8 * hh + j) * TB] = raw[j];
      }
    }
  }

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
    f32x16 zk[HT];
    const float c2 = 2.0f * a.inv2var;
    load_acc<T, VEC, HT>(orow, valid, h, hh, zk);
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dx0[t][r] += (32 * t + acc_row(r, hh) < h) ? -dl * (zk[t][r] - a.mu) * c2 : 0.f;
    load_acc<T, VEC, HT>(orow + h, valid, h, hh, zk);
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        gx1[t][r] += (32 * t + acc_row(r, hh) < h) ? -dl * (zk[t][r] - a.mu) * c2 : 0.f;
  }
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dx0[t][r] *= m;

  const int HC = g.HC;
  const int tbb = (int)g.tbb;
  // LDS: [2 stream buffers of G * tbb][per-wave dt stash: HT tiles x 4 KiB]
  Stream st;
  st.init(a.packed + g.off_b, smem, tbb, a.G, a.K * 3 * HC, wave, W, lane);
  float* stash = reinterpret_cast<float*>(smem + 2 * a.G * tbb) + wave * HT * 1024 + lane * 4;
  if (!active) {
    // tile past the padded token range: keep the block's barrier / staging cadence only
    for (int i = a.K * 3 * HC; i > 0; --i) (void)st.next();
  } else {
    f32x16 dcur[HT];   // phase 0: s accumulators, then ds; phase 2: dt
    for (int k = a.K - 1; k >= 0; --k) {
      // ---- phase 0: recompute s = s_net(x0)
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dcur[t][r] = 0.f;
      const char* buf = nullptr;
#pragma unroll 1
      for (int c = 0; c < HC; ++c) {
        buf = st.next();
        fwd_chunk<P, ACT, HT, NKS>(buf, reinterpret_cast<const float*>(buf + tbb - kTailBytes), a.act, lane, hh, x0f, dcur);
      }
      T* dS = reinterpret_cast<T*>(a.dSt) + (int64_t)k * 2 * g.hp * ldn;
      {
        // ---- affine chain, reversed: x1 is the step's saved input half
        const float* b2 = reinterpret_cast<const float*>(buf + tbb - kTailBytes) + 32;
        f32x16 x1[HT];
        // masked tokens read as zeros: the forward does not write the history of all-padding tiles
        load_acc<T, VEC, HT>(reinterpret_cast<const T*>(a.x1_hist) + ((int64_t)k * a.N + tok) * h, valid && m != 0.f, h, hh, x1);
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          const f32x16 bt = bias_tile(b2 + 32 * t, hh);
          f32x16 dt;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool in = 32 * t + acc_row(r, hh) < h;
            const float g1 = gx1[t][r] * m;
            const float ge = g1 * fast_exp(dcur[t][r] + bt[r]);
            dt[r] = g1;
            dcur[t][r] = in ? ge * x1[t][r] + dl : 0.f;
            gx1[t][r] = ge;
          }
          store_tile_t<T>(dS, 32 * t, loffh, dcur[t]);
          store_tile_t<T>(dS + (int64_t)g.hp * ldn, 32 * t, loffh, dt);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {dt[4 * q], dt[4 * q + 1], dt[4 * q + 2], dt[4 * q + 3]};
            *reinterpret_cast<f32x4*>(stash + t * 1024 + q * 256) = v;
          }
        }
      }
      // ---- phases 1, 2: s-net then t-net backward
#pragma unroll 1
      for (int net = 0; net < 2; ++net) {
        if (net == 1) stash_get<HT>(stash, dcur);
        T* Gt = reinterpret_cast<T*>(a.Gt) + (int64_t)(k * 2 + net) * g.Hp * ldn;
        T* Ht = reinterpret_cast<T*>(a.Ht) + (int64_t)(k * 2 + net) * g.Hp * ldn;
#pragma unroll 1
        for (int c = 0; c < HC; ++c) {
          buf = st.next();
          bwd_chunk<P, ACT, T, HT, NKS>(buf, reinterpret_cast<const float*>(buf + tbb - kTailBytes), a.act, lane, hh,
                                        x0f, dcur, dx0, Gt, Ht, 32 * c, loffH);
        }
      }
    }
  }
  T* drow = reinterpret_cast<T*>(a.dz_in) + tok * a.D;
  store_acc<T, VEC, HT>(drow, valid, h, hh, dx0);
  store_acc<T, VEC, HT>(drow + h, valid, h, hh, gx1);
}

// ---- weight gradients --------------------------------------------------------------------------------
// dW1[k,n] = Gt[k,n] . x0t^T, db1 = rowsum(Gt); dW2[k,n]^T = Ht[k,n] . dSt[k,n]^T, db2 = rowsum(dSt).
// All operands are [feature][token] (token-contiguous), so both MFMA fragments are plain 16-byte
// per-lane loads along the contraction axis.  One wave per (k, net, type, 32-row tile of A, token
// slab); partial tiles are dumped in register order and k_dw_reduce sums the slabs in fixed order.
template <class P, typename T, int HT>
__global__ void __launch_bounds__(256) k_dw_partial(DwArgs a) {
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  int64_t item = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (item >= a.nitems) return;  // wave-uniform
  int64_t id = item;
  const int slab = (int)(id % a.nslab); id /= a.nslab;
  const int mt = (int)(id % a.HC); id /= a.HC;
  const int type = (int)(id & 1);
  const int kn = (int)(id >> 1);
  const int64_t ldn = a.Npad;
  // token-blocked operands: element (row f, token tau) at ((tau / TB) * F + f) * TB + tau % TB
  constexpr int TB = 16 / sizeof(T);
  const int64_t sA = (int64_t)a.Hp * TB, sB = (int64_t)a.hp * TB;   // elements per token block
  const T* A = reinterpret_cast<const T*>(type == 0 ? a.Gt : a.Ht) + (int64_t)kn * a.Hp * ldn + (32 * mt + r32) * TB;
  const T* B = (type == 0 ? reinterpret_cast<const T*>(a.x0t) + (a.x0_per_step ? (int64_t)(kn >> 1) * a.hp * ldn : 0)
                          : reinterpret_cast<const T*>(a.dSt) + (int64_t)kn * a.hp * ldn) + r32 * TB;
  const int64_t t0 = (int64_t)slab * a.slab_len;
  int64_t t1 = t0 + a.slab_len;
  if (t1 > ldn) t1 = ldn;
  f32x16 acc[HT], accb[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; accb[t][r] = 0.f; }
  const bool want_b2 = type == 1 && mt == 0;
  if constexpr (P::KS == 16) {
    const u16x8 ones_raw = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_raw);
    // Batches of UB k-steps, the next batch's fragments requested before the current batch's MFMAs: the loop was a chain
    // of (global load -> wait -> MFMA) round trips, ~0.5 us per 16 tokens at C2 (slabs are multiples of 64 tokens).
    constexpr int UB = HT <= 2 ? 4 : 2;
    bf16x8 afc[UB], bfc[UB][HT];
    auto loadb = [&](int64_t k0, bf16x8 (&af)[UB], bf16x8 (&bf)[UB][HT]) {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int64_t blk = ((k0 + 16 * u) >> 3) + hh;
        af[u] = *reinterpret_cast<const bf16x8*>(A + blk * sA);
#pragma unroll
        for (int t = 0; t < HT; ++t) bf[u][t] = *reinterpret_cast<const bf16x8*>(B + blk * sB + 32 * t * TB);
      }
    };
    if (t0 < t1) loadb(t0, afc, bfc);
    for (int64_t k0 = t0; k0 < t1; k0 += 16 * UB) {
      bf16x8 afn[UB], bfn[UB][HT];
      const bool more = k0 + 16 * UB < t1;           // wave-uniform
      if (more) loadb(k0 + 16 * UB, afn, bfn);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
#pragma unroll
        for (int t = 0; t < HT; ++t) acc[t] = P::mma(afc[u], bfc[u][t], acc[t]);
        if (type == 0) accb[0] = P::mma(afc[u], ones, accb[0]);
        if (want_b2) {
#pragma unroll
          for (int t = 0; t < HT; ++t) accb[t] = P::mma(ones, bfc[u][t], accb[t]);
        }
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          afc[u] = afn[u];
#pragma unroll
          for (int t = 0; t < HT; ++t) bfc[u][t] = bfn[u][t];
        }
      }
    }
  } else {
    for (int64_t k0 = t0; k0 < t1; k0 += 8) {
      const int64_t blk = (k0 >> 2) + hh;
      const f32x4 af = *reinterpret_cast<const f32x4*>(A + blk * sA);
      f32x4 bf[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) bf[t] = *reinterpret_cast<const f32x4*>(B + blk * sB + 32 * t * TB);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < HT; ++t) acc[t] = P::mma(af[j], bf[t][j], acc[t]);
        if (type == 0) accb[0] = P::mma(af[j], 1.0f, accb[0]);
        if (want_b2) {
#pragma unroll
          for (int t = 0; t < HT; ++t) accb[t] = P::mma(1.0f, bf[t][j], accb[t]);
        }
      }
    }
  }
  float* out = a.part + ((item * HT) * 64 + lane) * 16;
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
      *reinterpret_cast<f32x4*>(out + (int64_t)t * 1024 + 4 * q) = v;
    }
  if (type == 0 && r32 == 0) {   // every column of accb[0] holds the row sums of the A tile
#pragma unroll
    for (int r = 0; r < 16; ++r) a.partb1[item * 32 + acc_row(r, hh)] = accb[0][r];
  }
  if (want_b2 && hh == 0) {      // every row of accb[t] holds the row sums of B tile t (column = B row)
#pragma unroll
    for (int t = 0; t < HT; ++t) a.partb2[((int64_t)kn * a.nslab + slab) * a.hp + 32 * t + r32] = accb[t][0];
  }
}

__global__ void __launch_bounds__(256) k_dw_reduce(DwArgs a) {
  // A block of 4 waves owns 64 consecutive dumped accumulator elements (dump order: for a fixed slab, the
  // 64 lanes of a wave read 64 consecutive floats); wave g sums slabs g, g+4, ... and the four partial sums
  // are combined through LDS in wave order (fixed order => bit-reproducible).  The slabs of one (step, net,
  // type, row tile) are nslab consecutive dumps.  The permuted store is 4 B per lane but the outputs are
  // only the weight shapes.
  __shared__ float red[4][64];
  const int64_t n1 = (int64_t)a.K * 2 * 2 * a.HC * a.HT * 1024;   // dumped elements per slab
  const int64_t nblk1 = n1 / 64;
  const int64_t nb1 = (int64_t)a.K * 2 * a.H, nb2 = (int64_t)a.K * 2 * a.h;
  if ((int64_t)blockIdx.x < nblk1) {
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    int64_t idx = (int64_t)blockIdx.x * 64 + lane;
    const int e = (int)(idx & 1023); int64_t q = idx >> 10;
    const int t = (int)(q % a.HT); q /= a.HT;
    const int mt = (int)(q % a.HC); q /= a.HC;
    const int type = (int)(q & 1); const int kn = (int)(q >> 1);
    const int64_t item0 = (((int64_t)kn * 2 + type) * a.HC + mt) * a.nslab;
    const float* p = a.part + (item0 * a.HT + t) * 1024 + e;
    const int64_t stride = (int64_t)a.HT * 1024;
    float s = 0.f;
    int sl = grp;
    for (; sl + 28 < a.nslab; sl += 32) {      // eight loads in flight, summed in slab order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(sl + 4 * u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sl < a.nslab; sl += 4) s += p[sl * stride];
    red[grp][lane] = s;
    __syncthreads();
    if (grp != 0) return;
    s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    const int r = e & 15, cl = e >> 4, cc = cl & 31, hh = cl >> 5;
    const int row = 32 * mt + (r & 3) + 4 * hh + 8 * (r >> 2), col = 32 * t + cc;
    if (row >= a.H || col >= a.h) return;
    if (a.sink) {          // straight into the parameters' gradient buffers (+=): no per-parameter add kernels
      if (type == 0) reinterpret_cast<float*>(a.sink[kn * 4 + 0])[row * a.h + col] += s;
      else reinterpret_cast<float*>(a.sink[kn * 4 + 2])[col * a.H + row] += s;
      return;
    }
    if (type == 0) a.dW1[((int64_t)kn * a.H + row) * a.h + col] = s;
    else a.dW2[((int64_t)kn * a.h + col) * a.H + row] = s;
    return;
  }
  int64_t idx = ((int64_t)blockIdx.x - nblk1) * blockDim.x + threadIdx.x;
  if (idx < nb1) {
    const int row = (int)(idx % a.H); const int kn = (int)(idx / a.H);
    const int64_t item0 = (((int64_t)kn * 2 + 0) * a.HC + row / 32) * a.nslab;
    const float* p = a.partb1 + item0 * 32 + row % 32;
    float s = 0.f;
    int sl = 0;
    for (; sl + 8 <= a.nslab; sl += 8) {         // eight loads in flight, summed in slab order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(sl + u) * 32];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sl < a.nslab; ++sl) s += p[sl * 32];
    if (a.sink) reinterpret_cast<float*>(a.sink[kn * 4 + 1])[row] += s;
    else a.db1[idx] = s;
  } else if (idx < nb1 + nb2) {
    idx -= nb1;
    const int out = (int)(idx % a.h); const int kn = (int)(idx / a.h);
    const float* p = a.partb2 + (int64_t)kn * a.nslab * a.hp + out;
    float s = 0.f;
    int sl = 0;
    for (; sl + 8 <= a.nslab; sl += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(sl + u) * a.hp];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sl < a.nslab; ++sl) s += p[(int64_t)sl * a.hp];
    if (a.sink) reinterpret_cast<float*>(a.sink[kn * 4 + 3])[out] += s;
    else a.db2[idx] = s;
  }
}

}  // namespace

// fixed-order fold of dumped weight-gradient partials (shared with nice_small.hip)
int mivae::nice_dw_reduce_launch(const DwArgs& a, hipStream_t st) {
  const int64_t nblk = (int64_t)a.K * 2 * 2 * a.HC * a.HT * 16 + ((int64_t)a.K * 2 * (a.H + a.h) + 255) / 256;
  k_dw_reduce<<<(unsigned)nblk, 256, 0, st>>>(a);
  MIVAE_CHECK_LAUNCH("nice_dw_reduce");
  return MIVAE_OK;
}

namespace {

// ---- host dispatch -----------------------------------------------------------------------------------
template <class P, typename T, int HT, int ACT>
int launch_stack(bool bwd, bool vec, const StackArgs& a, unsigned grid, unsigned block, size_t lds, hipStream_t st) {
  // dynamic LDS above 64 KiB has to be opted into per kernel
  auto big = [&](const void* fn) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  };
  if (bwd) {
    constexpr int LB = HT <= 1 ? 512 : 256;
    if (vec) { big((const void*)k_stack_bwd<P, T, HT, true, LB, ACT>); k_stack_bwd<P, T, HT, true, LB, ACT><<<grid, block, lds, st>>>(a); }
    else { big((const void*)k_stack_bwd<P, T, HT, false, LB, ACT>); k_stack_bwd<P, T, HT, false, LB, ACT><<<grid, block, lds, st>>>(a); }
  } else {
    constexpr int LB = 512;
    if (vec) { big((const void*)k_stack_fwd<P, T, HT, true, LB, ACT>); k_stack_fwd<P, T, HT, true, LB, ACT><<<grid, block, lds, st>>>(a); }
    else { big((const void*)k_stack_fwd<P, T, HT, false, LB, ACT>); k_stack_fwd<P, T, HT, false, LB, ACT><<<grid, block, lds, st>>>(a); }
  }
  return 0;
}
template <class P, typename T, int ACT>
int launch_ht(int HT, bool bwd, bool vec, const StackArgs& a, unsigned grid, unsigned block, size_t lds, hipStream_t st) {
  switch (HT) {
    case 1: return launch_stack<P, T, 1, ACT>(bwd, vec, a, grid, block, lds, st);
    case 2: return launch_stack<P, T, 2, ACT>(bwd, vec, a, grid, block, lds, st);
    case 3: return launch_stack<P, T, 3, ACT>(bwd, vec, a, grid, block, lds, st);
    case 4: return launch_stack<P, T, 4, ACT>(bwd, vec, a, grid, block, lds, st);
  }
  return -1;
}

constexpr size_t kLdsLimit = 160 * 1024;
constexpr int64_t kStageBufBytes = 40 * 1024;   // per LDS stream buffer (two of them)

static int run_stack(const char* name, bool bwd, StackArgs a, int64_t Bn, int h, int H, int dtype,
                     int act, hipStream_t st) {
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "%s: bad dtype %d", name, dtype);
  MIVAE_REQUIRE(act >= 0 && act <= ACT_SIGMOID, "%s: unsupported activation id %d", name, act);
  MIVAE_REQUIRE(a.K >= 1 && a.K <= 4096, "%s: K=%d out of range", name, a.K);
  MIVAE_REQUIRE(h >= 1 && H >= 1 && a.D == 2 * h, "%s: D must be 2*h (D=%d h=%d)", name, a.D, h);
  MIVAE_REQUIRE(a.S >= 1 && a.mask_div >= 1 && Bn >= 0, "%s: bad shape", name);
  a.act = act;
  a.g = make_geom(h, H, a.K, dtype == MIVAE_BF16);
  if (a.g.HT > 4) {
    mivae_set_error("%s: fused path supports D <= 256 (got D=%d); use the composed path", name, a.D);
    return MIVAE_EUNSUPPORTED;
  }
  a.N = Bn * a.S;
  if (a.N == 0) return MIVAE_OK;
  MIVAE_REQUIRE(a.z_in && a.packed, "%s: null pointer", name);
  MIVAE_REQUIRE(aligned16(a.packed), "%s: packed weights must be 16-byte aligned", name);
  const int esz = dtype == MIVAE_BF16 ? 2 : 4;
  // vector I/O needs 16-byte aligned rows and halves: D*esz % 16 == 0, h*esz % 16 == 0 (bf16: h % 8)
  bool vec = aligned16(a.z_in) && (a.z_out == nullptr || aligned16(a.z_out)) &&
             ((int64_t)h * esz) % 16 == 0 && (a.x1_hist == nullptr || aligned16(a.x1_hist));
  if (bwd) vec = vec && aligned16(a.dz_in) && (a.dz_out == nullptr || aligned16(a.dz_out));
  if (bwd && nice_small_bwd_takes(a, dtype == MIVAE_BF16, vec)) return nice_small_bwd_launch(a, st);
  const int64_t ntok = bwd ? a.Npad : a.N;
  const int64_t ntiles = (ntok + 31) / 32;
  const int64_t tb = bwd ? a.g.tbb : 2 * a.g.tbf;   // forward consumes (s, t) task pairs
  const int64_t ntasks = (int64_t)a.K * (bwd ? 3 : 1) * a.g.HC;
  // tasks per staged block: as many as fit one stream buffer (fewer barriers, deeper prefetch)
  int64_t G = kStageBufBytes / tb;
  if (G < 1) G = 1;
  if (G > ntasks) G = ntasks;
  a.G = (int)G;
  // waves per block: keep >= 256 blocks (one per CU) before growing the block; W waves share each
  // staged weight block, so larger W cuts L2 -> LDS traffic per token.
  int W = 1;   // token tiles per block
  const int wmax = bwd ? (a.g.HT > 1 ? 4 : 8) : 4;   // must match the kernels' launch bounds
  while (W < wmax && ntiles / (2 * W) >= 256) W *= 2;
  auto lds_need = [&](int w) -> size_t {
    return 2 * (size_t)(G * tb) + (size_t)w * a.g.HT * 4096;   // stream buffers + per-tile stash
  };
  while (W > 1 && lds_need(W) > kLdsLimit) W /= 2;
  MIVAE_REQUIRE(lds_need(W) <= kLdsLimit, "%s: task region too large for LDS (h=%d)", name, h);
  const unsigned grid = (unsigned)((ntiles + W - 1) / W);
  const size_t lds = lds_need(W);
  const unsigned block = (unsigned)(bwd ? W * 64 : W * 128);   // forward: two role waves per tile
  int rc;
  if (act == ACT_GELU) {
    if (dtype == MIVAE_F32) rc = launch_ht<MmaF32, float, ACT_GELU>(a.g.HT, bwd, vec, a, grid, block, lds, st);
    else rc = launch_ht<MmaBF16, unsigned short, ACT_GELU>(a.g.HT, bwd, vec, a, grid, block, lds, st);
  } else {
    if (dtype == MIVAE_F32) rc = launch_ht<MmaF32, float, -1>(a.g.HT, bwd, vec, a, grid, block, lds, st);
    else rc = launch_ht<MmaBF16, unsigned short, -1>(a.g.HT, bwd, vec, a, grid, block, lds, st);
  }
  MIVAE_REQUIRE(rc == 0, "%s: no kernel instance for HT=%d", name, a.g.HT);
  MIVAE_CHECK_LAUNCH(name);
  return MIVAE_OK;
}

// slab geometry of the weight-gradient kernel, shared by the workspace query and the launch
static void dw_geom(int64_t Npad, const Geom& g, int& nslab, int& slab_len, int64_t& nitems) {
  const int64_t items0 = (int64_t)g.K * 2 * 2 * g.HC;
  // work items (waves) to aim for: small stacks are bound by the fold of the slab partials (C2: 1024 items -> 27 us
  // pair, 2048 -> 33 us, 4096 -> 48 us), wide ones by the MFMA work of the partial kernel (keep 2048)
  static const int64_t forced = getenv("MIVAE_DW_ITEMS") ? atoll(getenv("MIVAE_DW_ITEMS")) : 0;
  const int64_t target = forced > 0 ? forced : (items0 <= 128 ? 1024 : 2048);
  int64_t want = (target + items0 - 1) / items0;
  const int64_t maxs = (Npad + 63) / 64;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  slab_len = (int)(((Npad + want - 1) / want + 63) / 64 * 64);
  nslab = (int)((Npad + slab_len - 1) / slab_len);
  nitems = items0 * nslab;
}

template <class P, typename T>
int launch_dw(int HT, const DwArgs& a, hipStream_t st) {
  const unsigned grid = (unsigned)((a.nitems + 3) / 4);
  switch (HT) {
    case 1: k_dw_partial<P, T, 1><<<grid, 256, 0, st>>>(a); return 0;
    case 2: k_dw_partial<P, T, 2><<<grid, 256, 0, st>>>(a); return 0;
    case 3: k_dw_partial<P, T, 3><<<grid, 256, 0, st>>>(a); return 0;
    case 4: k_dw_partial<P, T, 4><<<grid, 256, 0, st>>>(a); return 0;
  }
  return -1;
}

}  // namespace

// =====================================================================================================
extern "C" int64_t mivae_nice_pack_bytes(int64_t h, int64_t H, int dtype) {
  if (h < 1 || H < 1) return 0;
  return make_geom((int)h, (int)H, 1, dtype == MIVAE_BF16).total;   // bytes per step; total = K * this
}

extern "C" int mivae_nice_pack(const int64_t* ptr_table, void* packed, int64_t K, int64_t h, int64_t H,
                               int dtype, int inverse, void* stream) {
  MIVAE_REQUIRE(ptr_table && packed, "nice_pack: null pointer");
  MIVAE_REQUIRE(K >= 1 && K <= 4096 && h >= 1 && H >= 1, "nice_pack: bad shape K=%ld h=%ld H=%ld", (long)K, (long)h, (long)H);
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "nice_pack: bad dtype %d", dtype);
  Geom g = make_geom((int)h, (int)H, (int)K, dtype == MIVAE_BF16);
  const int64_t total = 2 * g.HC * (g.nfe_f + 256) + 3 * g.HC * (g.nfe_b + 256);
  dim3 grid((unsigned)((total + 255) / 256), 1, (unsigned)K);
  if (dtype == MIVAE_BF16) k_nice_pack<MmaBF16><<<grid, 256, 0, (hipStream_t)stream>>>(ptr_table, (char*)packed, g, inverse);
  else k_nice_pack<MmaF32><<<grid, 256, 0, (hipStream_t)stream>>>(ptr_table, (char*)packed, g, inverse);
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

extern "C" int64_t mivae_nice_stack_npad(int64_t N) { return (N + 63) / 64 * 64; }

extern "C" int mivae_nice_stack_bwd(const void* z_in, const void* z_out, const void* x1_hist,
                                    const int64_t* mask, const void* packed, const void* dz_out,
                                    const float* dlogp, void* dz_in, void* Gt, void* Ht, void* dSt,
                                    void* x0t, int64_t Bn, int64_t mask_div, int64_t S, int64_t D,
                                    int64_t H, int64_t K, int act, int base, float mu, float std,
                                    int dtype, void* stream) {
  MIVAE_REQUIRE(x1_hist && dz_in && Gt && Ht && dSt && x0t, "nice_stack_bwd: null pointer");
  MIVAE_REQUIRE(!base || (std > 0.f && z_out), "nice_stack_bwd: base log-prob needs z_out and std > 0");
  StackArgs a = {};
  a.z_in = z_in; a.z_out = const_cast<void*>(z_out); a.x1_hist = const_cast<void*>(x1_hist);
  a.mask = mask; a.packed = (const char*)packed;
  a.S = (int)S; a.mask_div = (int)mask_div; a.D = (int)D; a.K = (int)K;
  a.inverse = 0; a.base = base;
  a.mu = mu;
  a.inv2var = base ? 1.0f / (2.0f * std * std) : 0.f;
  a.cst = 0.f;
  a.dz_out = dz_out; a.dlogp = dlogp; a.dz_in = dz_in; a.Gt = Gt; a.Ht = Ht; a.dSt = dSt; a.x0t = x0t;
  a.Npad = mivae_nice_stack_npad(Bn * S);
  return run_stack("nice_stack_bwd", true, a, Bn, (int)(D / 2), (int)H, dtype, act, (hipStream_t)stream);
}

extern "C" int64_t mivae_nice_dw_workspace_floats(int64_t N, int64_t h, int64_t H, int64_t K) {
  if (N < 1 || h < 1 || H < 1 || K < 1) return 0;
  Geom g = make_geom((int)h, (int)H, (int)K, true);
  int nslab, slab_len;
  int64_t nitems;
  dw_geom(mivae_nice_stack_npad(N), g, nslab, slab_len, nitems);
  return nitems * ((int64_t)g.HT * 1024 + 32) + (int64_t)K * 2 * nslab * g.hp;
}

static int nice_dw_impl(const void* Gt, const void* Ht, const void* dSt, const void* x0t,
                        float* workspace, float* dW1, float* db1, float* dW2, float* db2, const int64_t* sink,
                        int64_t N, int64_t h, int64_t H, int64_t K, int dtype, void* stream, int x0_per_step = 0) {
  MIVAE_REQUIRE(Gt && Ht && dSt && x0t && workspace && (sink || (dW1 && db1 && dW2 && db2)), "nice_dw: null pointer");
  MIVAE_REQUIRE(N >= 1 && h >= 1 && H >= 1 && K >= 1 && K <= 4096, "nice_dw: bad shape");
  MIVAE_REQUIRE(dtype == MIVAE_F32 || dtype == MIVAE_BF16, "nice_dw: bad dtype %d", dtype);
  Geom g = make_geom((int)h, (int)H, (int)K, dtype == MIVAE_BF16);
  MIVAE_REQUIRE(g.HT <= 4, "nice_dw: supports D <= 256");
  MIVAE_REQUIRE(aligned16(Gt) && aligned16(Ht) && aligned16(dSt) && aligned16(x0t) && aligned16(workspace),
                "nice_dw: operands must be 16-byte aligned");
  DwArgs a = {};
  a.Gt = Gt; a.Ht = Ht; a.dSt = dSt; a.x0t = x0t;
  a.Npad = mivae_nice_stack_npad(N);
  a.K = (int)K; a.h = (int)h; a.H = (int)H; a.hp = g.hp; a.Hp = g.Hp; a.HT = g.HT; a.HC = g.HC;
  dw_geom(a.Npad, g, a.nslab, a.slab_len, a.nitems);
  a.part = workspace;
  a.partb1 = workspace + a.nitems * (int64_t)g.HT * 1024;
  a.partb2 = a.partb1 + a.nitems * 32;
  a.dW1 = dW1; a.db1 = db1; a.dW2 = dW2; a.db2 = db2; a.sink = sink;
  a.x0_per_step = x0_per_step;
  int rc = dtype == MIVAE_BF16 ? launch_dw<MmaBF16, unsigned short>(g.HT, a, (hipStream_t)stream)
                               : launch_dw<MmaF32, float>(g.HT, a, (hipStream_t)stream);
  MIVAE_REQUIRE(rc == 0, "nice_dw: no kernel instance");
  MIVAE_CHECK_LAUNCH("nice_dw_partial");
  return nice_dw_reduce_launch(a, (hipStream_t)stream);
}

extern "C" int mivae_nice_dw(const void* Gt, const void* Ht, const void* dSt, const void* x0t,
                             float* workspace, float* dW1, float* db1, float* dW2, float* db2,
                             int64_t N, int64_t h, int64_t H, int64_t K, int dtype, void* stream) {
  return nice_dw_impl(Gt, Ht, dSt, x0t, workspace, dW1, db1, dW2, db2, nullptr, N, h, H, K, dtype, stream);
}

// Same products, ADDED into the parameters' own fp32 gradient buffers: grad_ptrs is a device int64[K][8] table
// of addresses in mivae_nice_pack's order {W1s,b1s,W2s,b2s,W1t,b1t,W2t,b2t} (each element written by one thread).
extern "C" int mivae_nice_dw_sink(const void* Gt, const void* Ht, const void* dSt, const void* x0t,
                                  float* workspace, const int64_t* grad_ptrs, int64_t N, int64_t h, int64_t H,
                                  int64_t K, int dtype, void* stream) {
  MIVAE_REQUIRE(grad_ptrs, "nice_dw_sink: null pointer table");
  return nice_dw_impl(Gt, Ht, dSt, x0t, workspace, nullptr, nullptr, nullptr, nullptr, grad_ptrs, N, h, H, K, dtype, stream);
}

// mivae_nice_dw for K coupling steps that are NOT consecutive (other flows sit between them and change x0): x0t holds
// one (hp, Npad) section per step.  Used by the mixed-stack training path (csrc/mixed_stack.hip); same outputs.
extern "C" int mivae_nice_dw_steps(const void* Gt, const void* Ht, const void* dSt, const void* x0t,
                                   float* workspace, float* dW1, float* db1, float* dW2, float* db2,
                                   int64_t N, int64_t h, int64_t H, int64_t K, int dtype, void* stream) {
  return nice_dw_impl(Gt, Ht, dSt, x0t, workspace, dW1, db1, dW2, db2, nullptr, N, h, H, K, dtype, stream, 1);
}
