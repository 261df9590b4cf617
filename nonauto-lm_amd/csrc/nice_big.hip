// MFMA-bound NICE (affine-coupling) flow stack for wide latents (96 < D/2 <= 256, bf16 latents): the same
// arithmetic as nice_stack.hip (NICECoupling.forward/.backward, couplings.py:141-183, with FeedForwardCoupling
// nets couplings.py:50-51, iterated by FlowPrior.forward_pass / backward_pass, flow_prior.py:37-70, plus the base
// log-prob prior.py:148-158), re-tiled for the regime where the coupling MLPs dominate (C3: D = 256, C5: D = 512).
//
// Design (MI355X-first):
//  * one wave per SIMD (256 threads, up to 512 registers per lane): a wave owns 32 tokens for ALL K steps and keeps
//    x0 (B fragments), x1 and the current net's accumulator tiles in registers -- there is no cross-wave data
//    exchange at all; the only barrier is the weight ring's, once per 32-wide hidden chunk.
//  * the s-net and the t-net of a step run one after the other on the same wave (x1 *= exp(s); x1 += t); the three
//    independent instruction streams of the software pipeline -- second-layer MFMAs of chunk i, first-layer MFMAs of
//    chunk i + 2 and the activation (VALU) of chunk i + 1 -- sit in one basic block so that the matrix pipe, the
//    LDS fragment reads and the VALU overlap inside the wave.
//  * weights are packed (k_big_pack) into "items" of 2 HT KiB in exactly the order the pipeline consumes them and
//    streamed global -> LDS by LDS-DMA (global_load_lds, 16 B / lane) through a ring of R slots with COUNTED
//    vmcnt waits: several items stay in flight across the barrier, nothing is waited for that is not needed yet.
//    Biases travel in their own 4 KiB blocks (one per net) through four side slots.
//  * backward = three launches without the K x 2 x H x N side outputs of the small-D design:
//      k_big_chain<CHAIN>  recomputes s_k (s-nets only), walks the affine chain in reverse and writes d s_k, d t_k;
//      k_big_nets          for every (step, net): d hidden = W2^T d out, d pre = d hidden * act', d x0 += W1^T d pre;
//      k_big_dw            weight-stationary: a block owns (step, net, 4 hidden chunks), streams token tiles through
//                          LDS, RECOMPUTES hidden / d pre (two small products) and accumulates dW1, dW2, db1, db2
//                          in registers; token slabs are summed in fixed order by k_big_dw_reduce (no atomics).
#include "nice_common.hpp"
#include <stdlib.h>

using namespace mivae;

namespace {

typedef MmaBF16 P;
typedef unsigned short T;   // bf16 storage

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_cvoid_t;

enum Stream { ST_FWD = 0, ST_INV = 1, ST_CHAIN = 2, ST_NETS = 3, ST_BIAS = 4 };
enum Kind { KIND_W1 = 0, KIND_W2 = 1, KIND_W2T = 2, KIND_W1T = 3, KIND_ZERO = 4 };

constexpr int kBiasBlockFloats = 1024;   // [b1: Hp][b2: hp] zero padded to 4 KiB
constexpr int kBiasSlots = 4;

struct BigGeom {
  int h, H, hp, Hp, HT, HC, K;
  int item;          // bytes per item: 2 HT KiB
};
static BigGeom big_geom(int h, int H, int K) {
  BigGeom g;
  g.h = h; g.H = H; g.K = K;
  g.hp = (h + 31) / 32 * 32; g.Hp = (H + 31) / 32 * 32;
  g.HT = g.hp / 32; g.HC = g.Hp / 32;
  g.item = 2 * g.HT * 1024;
  return g;
}
// number of tasks / items of a stream
static __host__ __device__ int64_t stream_tasks(const BigGeom& g, int st) { return (int64_t)g.K * (st == ST_CHAIN ? 1 : 2) * g.HC; }
static __host__ __device__ int64_t stream_items(const BigGeom& g, int st) {
  const int64_t Tn = stream_tasks(g, st);
  return st == ST_NETS ? 3 * Tn + 4 : 2 * Tn + 2;
}

// ---- small ALU helpers ---------------------------------------------------------------------------------
// GELU (nn/activation.py:9-16) for bf16 operands: gelu(x) = max(x, 0) - |x| Phi(-|x|), Phi(-u) = 2^(u q(u) - 1),
// q = cubic least-squares fit on [0, 5]; max abs error 1.2e-5 in fp32 arithmetic (the result is rounded to bf16
// for the second MFMA, 4e-3 relative).  8 VALU instructions per element, one of them v_exp_f32.
__device__ __forceinline__ float phi_neg_abs(float x) {
  const float a = fminf(fabsf(x), 5.0f);
  float q = 0.004080009517455981f;
  q = fmaf(q, a, -0.045314069231662556f);
  q = fmaf(q, a, -0.4655827256271148f);
  q = fmaf(q, a, -1.149269309564982f);
  return __builtin_amdgcn_exp2f(fmaf(q, a, -1.0f));
}
__device__ __forceinline__ float gelu_fast(float x) { return fmaf(-fabsf(x), phi_neg_abs(x), fmaxf(x, 0.f)); }
// value and derivative: d/dx = Phi(x) + x phi(x)
__device__ __forceinline__ void gelu_fast_fd(float x, float& y, float& d) {
  const float e = phi_neg_abs(x);
  y = fmaf(-fabsf(x), e, fmaxf(x, 0.f));
  const float cdf = x > 0.f ? 1.0f - e : e;
  d = fmaf(x * kInvSqrt2Pi, __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f), cdf);
}

template <int ACT>
__device__ __forceinline__ void big_act(int act, f32x16& a) {
  if constexpr (ACT == ACT_GELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = gelu_fast(a[r]);
  } else {
    act_tile<-1, true>(act, a);
  }
}
template <int ACT>
__device__ __forceinline__ void big_act_fd(int act, f32x16& a, f32x16& d) {
  if constexpr (ACT == ACT_GELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float y, dd;
      gelu_fast_fd(a[r], y, dd);
      a[r] = y; d[r] = dd;
    }
  } else {
    act_tile_fd<-1, true>(act, a, d);
  }
}

// 32 consecutive floats from LDS -> accumulator layout (broadcast ds_read_b128)
__device__ __forceinline__ f32x16 lds_bias_tile(const float* B, int hh) {
  f32x16 a;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(B + 8 * q + 4 * hh);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[4 * q + i] = v[i];
  }
  return a;
}

// does any token of this block's 128-token range carry mask != 0?  Evaluated redundantly by every wave (two mask
// loads per lane): no LDS, no barrier.
__device__ __forceinline__ bool block_live(const int64_t* mask, int64_t N, int S, int mask_div, int lane) {
  bool any = false;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int64_t tk = (int64_t)blockIdx.x * 128 + 64 * u + lane;
    if (tk < N) {
      if (!mask) any = true;
      else {
        const int64_t bn = tk / S;
        any = any || mask[(bn / mask_div) * S + (tk - bn * S)] != 0;
      }
    }
  }
  return __any(any) != 0;
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- weight ring --------------------------------------------------------------------------------------
// Items of ITEM bytes are consumed strictly in stream order: the prologue consumes P0 items, body i the next IPB.
// Slot of item q = q % R.  enter(dead) is called by every wave once per body: it waits until the items the body
// reads have landed for this wave's own DMA pieces, joins the barrier (=> landed for every wave, and every wave
// has finished reading the items below `dead`) and refills the freed slots.
template <int ITEM, int R, int P0, int IPB>
struct Ring {
  static constexpr int PI = ITEM / 4096;   // DMA pieces (1 KiB) per wave and item
  const char* src;
  char* lds;
  int total, issued, slot;   // items
  int wave, lane;

  __device__ __forceinline__ void issue_one() {
    const char* s = src + (int64_t)issued * ITEM + wave * 1024 + lane * 16;
    char* d = lds + slot * ITEM + wave * 1024;
#pragma unroll
    for (int p = 0; p < PI; ++p)
      __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(s + p * 4096), (lds_void_t*)(d + p * 4096), 16, 0, 0);
    ++issued;
    slot = slot + 1 == R ? 0 : slot + 1;
  }
  __device__ __forceinline__ void init(const char* s, char* l, int total_, int wave_, int lane_) {
    src = s; lds = l; total = total_; issued = 0; slot = 0; wave = wave_; lane = lane_;
    while (issued < R && issued < total) issue_one();
  }
  // prologue: items [0, P0) readable afterwards
  __device__ __forceinline__ void enter_prologue() {
    if (issued < total) wait_vm<(R - P0) * PI>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  // body i: items [P0 + IPB i, P0 + IPB (i + 1)) readable afterwards; slots below P0 + IPB i are refilled
  __device__ __forceinline__ void enter(int i) {
    // issued so far: R + (i > 0 ? P0 + IPB (i - 1) : 0); needed: everything below P0 + IPB (i + 1)
    if (issued >= total) wait_vm<0>();
    else if (i == 0) wait_vm<(R - P0 - IPB) * PI>();
    else wait_vm<(R - 2 * IPB) * PI>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the slots about to be refilled are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int lim = P0 + IPB * i + R;
    while (issued < lim && issued < total) issue_one();
  }
  __device__ __forceinline__ const char* item(int q) const { return lds + (q % R) * ITEM; }
};

// ---- pack ----------------------------------------------------------------------------------------------
// ptrs[k][8] = {W1s, b1s, W2s, b2s, W1t, b1t, W2t, b2t} (fp32, torch.nn.Linear layout: W1 (H, h), W2 (h, H)).
// One thread per bf16 element of the stream (or per float of the bias array).
__device__ __forceinline__ void task_of(const BigGeom& g, int st, int64_t i, int& k, int& net, int& c) {
  c = (int)(i % g.HC);
  if (st == ST_CHAIN) { k = g.K - 1 - (int)(i / g.HC); net = 0; return; }
  const int kk = (int)(i / (2 * g.HC)), nn = (int)((i / g.HC) & 1);
  if (st == ST_FWD) { k = kk; net = nn; }
  else if (st == ST_INV) { k = g.K - 1 - kk; net = 1 - nn; }
  else { k = g.K - 1 - kk; net = nn; }
}
__global__ void __launch_bounds__(256) k_big_pack(const int64_t* __restrict__ ptrs, char* __restrict__ out, BigGeom g, int st) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (st == ST_BIAS) {
    if (idx >= (int64_t)g.K * 2 * kBiasBlockFloats) return;
    const int kn = (int)(idx / kBiasBlockFloats), e = (int)(idx % kBiasBlockFloats);
    const float* b1 = reinterpret_cast<const float*>(ptrs[(kn >> 1) * 8 + (kn & 1) * 4 + 1]);
    const float* b2 = reinterpret_cast<const float*>(ptrs[(kn >> 1) * 8 + (kn & 1) * 4 + 3]);
    float v = 0.f;
    if (e < g.Hp) { if (e < g.H) v = b1[e]; }
    else if (e - g.Hp < g.h) v = b2[e - g.Hp];
    reinterpret_cast<float*>(out)[idx] = v;
    return;
  }
  const int per_item = g.item / 2;
  const int64_t nitems = stream_items(g, st), Tn = stream_tasks(g, st);
  if (idx >= nitems * per_item) return;
  const int64_t q = idx / per_item;
  const int e = (int)(idx % per_item);
  int64_t task;
  int kind;
  if (st == ST_NETS) {
    if (q < 4) { task = q >> 1; kind = (q & 1) ? KIND_W2T : KIND_W1; }
    else {
      const int64_t i = (q - 4) / 3; const int r = (int)((q - 4) % 3);
      if (r == 0) { task = i; kind = KIND_W1T; }
      else { task = i + 2; kind = r == 1 ? KIND_W1 : KIND_W2T; }
    }
  } else {
    if (q < 2) { task = q; kind = KIND_W1; }
    else {
      const int64_t i = (q - 2) >> 1;
      if (((q - 2) & 1) == 0) { task = i; kind = KIND_W2; } else { task = i + 2; kind = KIND_W1; }
    }
  }
  float v = 0.f;
  if (task < Tn) {
    int k, net, c;
    task_of(g, st, task, k, net, c);
    const float* W1 = reinterpret_cast<const float*>(ptrs[k * 8 + net * 4 + 0]);
    const float* W2 = reinterpret_cast<const float*>(ptrs[k * 8 + net * 4 + 2]);
    const int frag = e >> 9, lane = (e >> 3) & 63, j = e & 7;
    const int r = lane & 31, hh = lane >> 5;
    if (kind == KIND_W1) {          // frag = ks: A[row = hidden 32c + r][k = in nat]
      const int row = 32 * c + r, col = P::nat_k(frag, hh, j);
      if (row < g.H && col < g.h) v = W1[(int64_t)row * g.h + col];
    } else if (kind == KIND_W2) {   // frag = (t, s): A[row = out 32t + r][k = hidden 32c + perm]
      const int t = frag >> 1, s = frag & 1;
      const int row = 32 * t + r, col = 32 * c + P::perm_k(s, hh, j);
      if (row < g.h && col < g.H) v = W2[(int64_t)row * g.H + col];
    } else if (kind == KIND_W2T) {  // frag = ks: A[row = hidden 32c + r][k = out nat] = W2[out][hidden]
      const int row = P::nat_k(frag, hh, j), col = 32 * c + r;
      if (row < g.h && col < g.H) v = W2[(int64_t)row * g.H + col];
    } else {                        // KIND_W1T, frag = (t, s): A[row = in 32t + r][k = hidden 32c + perm] = W1[hidden][in]
      const int t = frag >> 1, s = frag & 1;
      const int row = 32 * c + P::perm_k(s, hh, j), col = 32 * t + r;
      if (row < g.H && col < g.h) v = W1[(int64_t)row * g.h + col];
    }
  }
  reinterpret_cast<unsigned short*>(out)[idx] = f32_to_bf16(v);
}

// ---- forward / inverse / backward-chain kernel --------------------------------------------------------------
struct BigArgs {
  const T* z_in;        // (N, D)
  T* z_out;             // (N, D)            [FWD / INV: written; CHAIN: read (base log-prob gradient)]
  T* x1_hist;           // (K, N, h) or null [FWD: written; CHAIN: read]
  const int64_t* mask;  // (Bn / mask_div, S) or null
  const char* stream;
  const float* bias;    // (K, 2, 1024)
  float* tok_val;       // (N)
  int64_t N;
  int S, mask_div, D, K, h, H, HC;
  int act, base;
  float mu, inv2var, cst;
  // CHAIN
  const T* dz_out;      // (N, D) or null
  const float* dlogp;   // (N) per-token cotangent of tok_val, or null
  T* dz_in;             // (N, D): the x1 half is written by CHAIN, the x0 half by k_big_nets
  T* dSdT;              // (K, 2, N, h): d s_k, d t_k
};

enum Mode { M_FWD = 0, M_INV = 1, M_CHAIN = 2 };

template <int HT> struct RingCfg { static constexpr int R = HT <= 4 ? 12 : 9; };

// first layer of one chunk: b1 + W1[c] . x0^T  (A fragments from the item, B = x0 fragments in registers)
template <int HT>
__device__ __forceinline__ f32x16 big_l1(const char* item, const float* b1c, int lane, int hh, const bf16x8 (&x0f)[2 * HT]) {
  const bf16x8* w = reinterpret_cast<const bf16x8*>(item) + lane;
  f32x16 a = lds_bias_tile(b1c, hh);
#pragma unroll
  for (int ks = 0; ks < 2 * HT; ++ks) a = P::mma(w[ks * 64], x0f[ks], a);
  return a;
}
// second layer of one chunk: acc[t] += W2[t, c] . hid^T
template <int HT>
__device__ __forceinline__ void big_l2(const char* item, int lane, bf16x8 b0, bf16x8 b1, f32x16 (&acc)[HT]) {
  const bf16x8* w = reinterpret_cast<const bf16x8*>(item) + lane;
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    acc[t] = P::mma(w[(2 * t) * 64], b0, acc[t]);
    acc[t] = P::mma(w[(2 * t + 1) * 64], b1, acc[t]);
  }
}

template <int HT, int MODE, int ACT>
__global__ void __launch_bounds__(256) k_big_chain(BigArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKS = 2 * HT;
  constexpr int ITEM = NKS * 1024;
  constexpr int R = RingCfg<HT>::R;
  constexpr int NETS = MODE == M_CHAIN ? 1 : 2;
  const int lane = threadIdx.x & 63;
  const int r32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = a.h, HC = a.HC;
  const int64_t tok = ((int64_t)blockIdx.x * 4 + wave) * 32 + r32;
  const bool valid = tok < a.N;
  float m = 0.f, dl = 0.f;
  if (valid) {
    m = 1.f;
    if (a.mask) {
      const int64_t bn = tok / a.S;
      m = (float)a.mask[(bn / a.mask_div) * a.S + (tok - bn * a.S)];
    }
    if (MODE == M_CHAIN && a.dlogp) dl = a.dlogp[tok] * m;
  }
  const bool live = __any(m != 0.f) != 0;
  if (!block_live(a.mask, a.N, a.S, a.mask_div, lane)) {
    // the whole block is padding: outputs are zeros; the history / d s, d t rows of masked tokens are never read
    if (valid) {
      const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
      if (MODE == M_CHAIN) {
        T* drow = a.dz_in + tok * a.D + h;
        for (int f = 8 * hh; f < h; f += 16) *reinterpret_cast<u16x8*>(drow + f) = zero;
      } else {
        T* orow = a.z_out + tok * a.D;
        for (int f = 8 * hh; f < a.D; f += 16) *reinterpret_cast<u16x8*>(orow + f) = zero;
        if (hh == 0 && a.tok_val) a.tok_val[tok] = 0.f;
      }
    }
    return;
  }
  const bool on = valid && m != 0.f;   // masked tokens are computed on zeros (their outputs are multiplied by the mask)
  const T* zrow = a.z_in + tok * a.D;
  bf16x8 x0f[NKS];
  load_x0_bf16<T, true, NKS>(zrow, on, h, hh, x0f);
  f32x16 x1[HT];   // FWD / INV: the transformed half; CHAIN: d loss / d x1
  if constexpr (MODE == M_CHAIN) {
    if (a.dz_out) load_acc<T, true, HT>(a.dz_out + tok * a.D + h, valid, h, hh, x1);
    else {
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) x1[t][r] = 0.f;
    }
    if (a.base && a.dlogp) {   // d/dz_K of the base log-prob: dl * -(z_K - mu) / std^2
      f32x16 zk[HT];
      const float c2 = 2.0f * a.inv2var;
      load_acc<T, true, HT>(a.z_out + tok * a.D + h, valid, h, hh, zk);
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          x1[t][r] += (32 * t + acc_row(r, hh) < h) ? -dl * (zk[t][r] - a.mu) * c2 : 0.f;
    }
  } else {
    load_acc<T, true, HT>(zrow + h, on, h, hh, x1);
  }
  float ld = 0.f;

  Ring<ITEM, R, 2, 2> ring;
  float* bias_lds = reinterpret_cast<float*>(smem + R * ITEM);
  const int NJ = a.K * NETS;            // nets in stream order
  const int Tn = NJ * HC;               // tasks
  // bias block of net-index j in stream order
  auto bias_src = [&](int j) -> const char* {
    int k, net;
    if (MODE == M_CHAIN) { k = a.K - 1 - j; net = 0; }
    else if (MODE == M_FWD) { k = j >> 1; net = j & 1; }
    else { k = a.K - 1 - (j >> 1); net = 1 - (j & 1); }
    return reinterpret_cast<const char*>(a.bias + (int64_t)(k * 2 + net) * kBiasBlockFloats);
  };
  auto issue_bias = [&](int j) {
    __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(bias_src(j) + wave * 1024 + lane * 16),
                                     (lds_void_t*)((char*)bias_lds + (j & 3) * 4096 + wave * 1024), 16, 0, 0);
  };
  for (int j = 0; j < kBiasSlots && j < NJ; ++j) issue_bias(j);   // oldest DMA: complete before any ring item
  ring.init(a.stream, smem, 2 * Tn + 2, wave, lane);

  f32x16 acc[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- prologue: L1(0) -> act -> hid(0); L1(1) -> a1
  ring.enter_prologue();
  bf16x8 hb0, hb1;
  f32x16 a1;
  {
    f32x16 a0 = big_l1<HT>(ring.item(0), bias_lds, lane, hh, x0f);
    big_act<ACT>(a.act, a0);
    hb0 = P::from_acc<0>(a0); hb1 = P::from_acc<1>(a0);
    // task 1: chunk 1 of net 0 (HC >= 2)
    a1 = big_l1<HT>(ring.item(1), bias_lds + 32, lane, hh, x0f);
  }
  int j = 0, c = 0;   // net-index and chunk of task i
  for (int i = 0; i < Tn; ++i) {
    ring.enter(i);
    if (i > 0 && c == 0 && j - 1 + kBiasSlots < NJ) issue_bias(j - 1 + kBiasSlots);   // slot of the net folded in body i-1
    const char* it2 = ring.item(2 + 2 * i);
    const char* it1 = ring.item(3 + 2 * i);
    // task i + 2
    int c2 = c + 2, j2 = j;
    if (c2 >= HC) { c2 -= HC; ++j2; }
    if (live) {
      f32x16 a2 = a1;
      if (i + 2 < Tn) a2 = big_l1<HT>(it1, bias_lds + (j2 & 3) * kBiasBlockFloats + 32 * c2, lane, hh, x0f);
      big_l2<HT>(it2, lane, hb0, hb1, acc);
      if (i + 1 < Tn) {
        big_act<ACT>(a.act, a1);
        hb0 = P::from_acc<0>(a1); hb1 = P::from_acc<1>(a1);
      }
      a1 = a2;
    }
    if (c == HC - 1) {
      // ---- fold the finished net into the chain
      if (live) {
        const float* b2 = bias_lds + (j & 3) * kBiasBlockFloats + HC * 32;
        const int nn = MODE == M_CHAIN ? 0 : (j & 1);
        const int k = MODE == M_FWD ? (j >> 1) : a.K - 1 - (MODE == M_CHAIN ? j : (j >> 1));
        if constexpr (MODE == M_FWD) {
          if (nn == 0) {
            if (a.x1_hist) store_acc<T, true, HT>(a.x1_hist + ((int64_t)k * a.N + tok) * h, valid, h, hh, x1);
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              const f32x16 bt = lds_bias_tile(b2 + 32 * t, hh);
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float s = acc[t][r] + bt[r];
                x1[t][r] *= fast_exp(s);
                ld += s;
                acc[t][r] = 0.f;
              }
            }
          } else {
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              const f32x16 bt = lds_bias_tile(b2 + 32 * t, hh);
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                x1[t][r] = (x1[t][r] + (acc[t][r] + bt[r])) * m;
                acc[t][r] = 0.f;
              }
            }
          }
        } else if constexpr (MODE == M_INV) {
          if (nn == 0) {   // t first: x1 -= t
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              const f32x16 bt = lds_bias_tile(b2 + 32 * t, hh);
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                x1[t][r] -= acc[t][r] + bt[r];
                acc[t][r] = 0.f;
              }
            }
          } else {         // then x1 = x1 exp(-s) m
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              const f32x16 bt = lds_bias_tile(b2 + 32 * t, hh);
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float s = acc[t][r] + bt[r];
                x1[t][r] = (x1[t][r] * fast_exp(-s)) * m;
                ld -= s;
                acc[t][r] = 0.f;
              }
            }
          }
        } else {           // CHAIN: x1 holds d loss / d x1_k; emit d s_k, d t_k and step back
          T* dS = a.dSdT + ((int64_t)(k * 2) * a.N + tok) * h;
          T* dT = dS + a.N * (int64_t)h;
          const T* hrow = a.x1_hist + ((int64_t)k * a.N + tok) * h;
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            const f32x16 bt = lds_bias_tile(b2 + 32 * t, hh);
            f32x16 xp[1], ds[1], dt[1];
            // masked tokens read as zeros (the forward does not write the history of all-padding blocks)
            load_acc<T, true, 1>(hrow + 32 * t, on && 32 * t < h, h - 32 * t, hh, xp);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const bool in = 32 * t + acc_row(r, hh) < h;
              const float g1 = x1[t][r] * m;
              const float ge = g1 * fast_exp(acc[t][r] + bt[r]);
              dt[0][r] = g1;
              ds[0][r] = in ? ge * xp[0][r] + dl : 0.f;
              x1[t][r] = ge;
              acc[t][r] = 0.f;
            }
            store_acc<T, true, 1>(dS + 32 * t, valid && 32 * t < h, h - 32 * t, hh, ds);
            store_acc<T, true, 1>(dT + 32 * t, valid && 32 * t < h, h - 32 * t, hh, dt);
          }
        }
      }
      c = 0; ++j;
    } else {
      ++c;
    }
  }

  // ---- outputs ---------------------------------------------------------------------------------------------
  if constexpr (MODE == M_CHAIN) {
    if (!live) {   // an all-padding tile of a live block: its gradient is zero
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) x1[t][r] = 0.f;
    }
    store_acc<T, true, HT>(a.dz_in + tok * a.D + h, valid, h, hh, x1);
    return;
  } else {
    T* orow = a.z_out + tok * a.D;
    if (!live) {
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) x1[t][r] = 0.f;
    }
    store_acc<T, true, HT>(orow + h, valid, h, hh, x1);
    float base = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {   // x0 half: out = x0 * m (x0f is already masked), fragment layout
      const int f = 16 * ks + 8 * hh;
      const u16x8 raw = __builtin_bit_cast(u16x8, x0f[ks]);
      if (valid && f < h) *reinterpret_cast<u16x8*>(orow + f) = raw;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        if (valid && f + jj < h) {
          const float d = bf16_to_f32(raw[jj]) - a.mu;
          base += -(d * d) * a.inv2var - a.cst;
        }
    }
    if (a.base) {
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = 32 * t + acc_row(r, hh);
          if (f < h) {
            const float v = bf16_to_f32(f32_to_bf16(x1[t][r]));   // what the stored z_K holds
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
}

// ---- backward through the nets: d x0 ----------------------------------------------------------------------------
struct NetsArgs {
  const T* z_in;        // (N, D)
  const T* z_out;       // (N, D) (base log-prob gradient) or null
  const int64_t* mask;
  const char* stream;   // ST_NETS
  const float* bias;
  const T* dz_out;      // (N, D) or null
  const float* dlogp;   // (N) or null
  const T* dSdT;        // (K, 2, N, h)
  T* dz_in;             // (N, D): x0 half written here
  int64_t N;
  int S, mask_div, D, K, h, H, HC;
  int act, base;
  float mu, inv2var;
};

template <int HT, int ACT>
__global__ void __launch_bounds__(256) k_big_nets(NetsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKS = 2 * HT;
  constexpr int ITEM = NKS * 1024;
  constexpr int R = RingCfg<HT>::R;
  const int lane = threadIdx.x & 63;
  const int r32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = a.h, HC = a.HC;
  const int64_t tok = ((int64_t)blockIdx.x * 4 + wave) * 32 + r32;
  const bool valid = tok < a.N;
  float m = 0.f, dl = 0.f;
  if (valid) {
    m = 1.f;
    if (a.mask) {
      const int64_t bn = tok / a.S;
      m = (float)a.mask[(bn / a.mask_div) * a.S + (tok - bn * a.S)];
    }
    if (a.dlogp) dl = a.dlogp[tok] * m;
  }
  const bool live = __any(m != 0.f) != 0;
  if (!block_live(a.mask, a.N, a.S, a.mask_div, lane)) {
    if (valid) {
      const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
      T* drow = a.dz_in + tok * a.D;
      for (int f = 8 * hh; f < h; f += 16) *reinterpret_cast<u16x8*>(drow + f) = zero;
    }
    return;
  }
  const bool on = valid && m != 0.f;
  bf16x8 x0f[NKS];
  load_x0_bf16<T, true, NKS>(a.z_in + tok * a.D, on, h, hh, x0f);
  f32x16 dx0[HT];
  if (a.dz_out) load_acc<T, true, HT>(a.dz_out + tok * a.D, valid, h, hh, dx0);
  else {
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dx0[t][r] = 0.f;
  }
  if (a.base && a.dlogp && a.z_out) {
    f32x16 zk[HT];
    const float c2 = 2.0f * a.inv2var;
    load_acc<T, true, HT>(a.z_out + tok * a.D, valid, h, hh, zk);
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dx0[t][r] += (32 * t + acc_row(r, hh) < h) ? -dl * (zk[t][r] - a.mu) * c2 : 0.f;
  }
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dx0[t][r] *= m;

  Ring<ITEM, R, 4, 3> ring;
  float* bias_lds = reinterpret_cast<float*>(smem + R * ITEM);
  const int NJ = a.K * 2;
  const int Tn = NJ * HC;
  auto kn_of = [&](int j) { return (a.K - 1 - (j >> 1)) * 2 + (j & 1); };
  auto issue_bias = [&](int j) {
    const char* src = reinterpret_cast<const char*>(a.bias + (int64_t)kn_of(j) * kBiasBlockFloats);
    __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(src + wave * 1024 + lane * 16),
                                     (lds_void_t*)((char*)bias_lds + (j & 3) * 4096 + wave * 1024), 16, 0, 0);
  };
  // d out of net-index j as B fragments (natural k order over the out features)
  bf16x8 dob[NKS];
  auto load_dout = [&](int j) {
    load_x0_bf16<T, true, NKS>(a.dSdT + ((int64_t)kn_of(j) * a.N + tok) * h, on, h, hh, dob);
  };
  load_dout(0);
  for (int j = 0; j < kBiasSlots && j < NJ; ++j) issue_bias(j);
  ring.init(a.stream, smem, 3 * Tn + 4, wave, lane);

  // stage A of a task: pre = b1 + W1 x0, dh = W2^T d out
  auto stage_a = [&](const char* itW1, const char* itW2T, const float* b1c, f32x16& pre, f32x16& dh) {
    const bf16x8* w1 = reinterpret_cast<const bf16x8*>(itW1) + lane;
    const bf16x8* w2 = reinterpret_cast<const bf16x8*>(itW2T) + lane;
    pre = lds_bias_tile(b1c, hh);
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      pre = P::mma(w1[ks * 64], x0f[ks], pre);
      dh = P::mma(w2[ks * 64], dob[ks], dh);
    }
  };
  ring.enter_prologue();
  f32x16 pre1, dh1;
  bf16x8 gb0, gb1;   // d pre of task i as B fragments
  {
    f32x16 pre0, dh0, dact;
    stage_a(ring.item(0), ring.item(1), bias_lds, pre0, dh0);
    big_act_fd<ACT>(a.act, pre0, dact);
#pragma unroll
    for (int r = 0; r < 16; ++r) dh0[r] *= dact[r];
    gb0 = P::from_acc<0>(dh0); gb1 = P::from_acc<1>(dh0);
    stage_a(ring.item(2), ring.item(3), bias_lds + 32, pre1, dh1);
  }
  int j = 0, c = 0;
  for (int i = 0; i < Tn; ++i) {
    ring.enter(i);
    if (i > 0 && c == 0 && j - 1 + kBiasSlots < NJ) issue_bias(j - 1 + kBiasSlots);
    const char* itT = ring.item(4 + 3 * i);
    const char* itA = ring.item(5 + 3 * i);
    const char* itB = ring.item(6 + 3 * i);
    int c2 = c + 2, j2 = j;
    if (c2 >= HC) { c2 -= HC; ++j2; }
    if (live) {
      // the first task of the next net enters stage A here: switch the d out fragments
      if (c2 == 0 && j2 < NJ) load_dout(j2);
      f32x16 pre2 = pre1, dh2 = dh1;
      if (i + 2 < Tn) stage_a(itA, itB, bias_lds + (j2 & 3) * kBiasBlockFloats + 32 * c2, pre2, dh2);
      {   // stage C of task i: dx0[t] += W1T[t] . d pre
        const bf16x8* w = reinterpret_cast<const bf16x8*>(itT) + lane;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          dx0[t] = P::mma(w[(2 * t) * 64], gb0, dx0[t]);
          dx0[t] = P::mma(w[(2 * t + 1) * 64], gb1, dx0[t]);
        }
      }
      if (i + 1 < Tn) {   // stage B of task i + 1
        f32x16 dact;
        big_act_fd<ACT>(a.act, pre1, dact);
#pragma unroll
        for (int r = 0; r < 16; ++r) dh1[r] *= dact[r];
        gb0 = P::from_acc<0>(dh1); gb1 = P::from_acc<1>(dh1);
      }
      pre1 = pre2; dh1 = dh2;
    }
    if (c == HC - 1) { c = 0; ++j; } else ++c;
  }
  store_acc<T, true, HT>(a.dz_in + tok * a.D, valid, h, hh, dx0);
}

// ---- weight gradients (weight-stationary, recompute) -----------------------------------------------------------
struct DwBigArgs {
  const T* z_in;        // (N, D): x0 = z_in[:, :h]
  const T* dSdT;        // (K, 2, N, h)
  const int64_t* mask;
  const char* stream;   // ST_NETS (W1 and W2T fragments are read straight from it)
  const float* bias;
  float* part;          // [kn][c][slab][2][HT][64][16]
  float* partb1;        // [kn][c][slab][32]
  float* partb2;        // [kn][slab][hp]
  int64_t N;
  int S, mask_div, D, K, h, H, HC, act;
  int nslab, slab_tiles;
};

// LDS tile image: row-major [32 tokens][hp] bf16 with the 16-byte chunks of a row XOR-swizzled so that both the
// row reads (ds_read_b128, k = features) and the transposing reads (ds_read_b64_tr_b16, k = tokens) are conflict free.
__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 3) | ((row & 3) << 2); }

typedef __bf16 bf16x4v __attribute__((__vector_size__(4 * sizeof(__bf16))));
#define BIG_LDS __attribute__((address_space(3)))

// A fragment with k = tokens in the order from_acc() gives the other operand:
// f[j] = tile[16 s + 8 (j >> 2) + 4 hh + (j & 3)][32 t + r32]
template <int HPB>   // row bytes
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int t, int s, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int hh = g >> 1;
  const int row0 = 16 * s + 4 * hh + q;          // rows row0 and row0 + 8
  const int chunk = 4 * t + 2 * (g & 1) + (p >> 1);
  const int sub = 8 * (p & 1);
  const char* a0 = tile + row0 * HPB + ((chunk ^ swz(row0)) << 4) + sub;
  const char* a1 = tile + (row0 + 8) * HPB + ((chunk ^ swz(row0 + 8)) << 4) + sub;
  const bf16x4v v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((BIG_LDS bf16x4v*)(a0));
  const bf16x4v v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((BIG_LDS bf16x4v*)(a1));
  const auto v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int HT, int ACT>
__global__ void __launch_bounds__(256) k_big_dw(DwBigArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKS = 2 * HT;
  constexpr int ITEM = NKS * 1024;
  constexpr int HPB = HT * 64;              // bytes per tile row
  constexpr int TILE = 32 * HPB;            // bytes per operand tile
  constexpr int CPT = TILE / 16 / 256;      // 16-byte chunks per thread and operand tile (HT/2)
  const int lane = threadIdx.x & 63;
  const int r32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slab = blockIdx.x, kn = blockIdx.y, c = blockIdx.z * 4 + wave;
  const int h = a.h, HC = a.HC;
  const bool wave_on = c < HC;
  // my task in the ST_NETS stream: kk = K-1-k, task = (kk * 2 + net) * HC + c
  const int k = kn >> 1, net = kn & 1;
  bf16x8 w1f[NKS], w2f[NKS];
  float b1v = 0.f;
  if (wave_on) {
    const int64_t task = ((int64_t)(a.K - 1 - k) * 2 + net) * HC + c;
    const int64_t q1 = task < 2 ? 2 * task : 4 + 3 * (task - 2) + 1;
    const bf16x8* p1 = reinterpret_cast<const bf16x8*>(a.stream + q1 * ITEM) + lane;
    const bf16x8* p2 = reinterpret_cast<const bf16x8*>(a.stream + (q1 + 1) * ITEM) + lane;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) { w1f[ks] = p1[ks * 64]; w2f[ks] = p2[ks * 64]; }
    b1v = a.bias[(int64_t)kn * kBiasBlockFloats + 32 * c + r32];
  }
  f32x16 dW1[HT], dW2[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dW1[t][r] = 0.f; dW2[t][r] = 0.f; }
  float db1 = 0.f;
  float db2[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t) db2[t] = 0.f;
  const bool want_b2 = blockIdx.z == 0 && wave == 0;

  // staging: thread -> (row, chunk) of each operand tile
  const T* dO = a.dSdT + (int64_t)kn * a.N * h;
  const int64_t tile0 = (int64_t)slab * a.slab_tiles;
  int64_t ntiles = (a.N + 31) / 32 - tile0;
  if (ntiles > a.slab_tiles) ntiles = a.slab_tiles;
  u16x8 sx[CPT], sg[CPT];
  auto gload = [&](int64_t tile) {
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int id = threadIdx.x + 256 * u;
      const int row = id / (HPB / 16), ch = id % (HPB / 16);
      const int64_t tk = tile * 32 + row;
      u16x8 vx = {0, 0, 0, 0, 0, 0, 0, 0}, vg = {0, 0, 0, 0, 0, 0, 0, 0};
      if (tk < a.N && ch * 8 < h) {
        bool keep = true;
        if (a.mask) {
          const int64_t bn = tk / a.S;
          keep = a.mask[(bn / a.mask_div) * a.S + (tk - bn * a.S)] != 0;
        }
        if (keep) {
          vx = *reinterpret_cast<const u16x8*>(a.z_in + tk * a.D + ch * 8);
          vg = *reinterpret_cast<const u16x8*>(dO + tk * h + ch * 8);
        }
      }
      sx[u] = vx; sg[u] = vg;
    }
  };
  auto lwrite = [&](int buf) {
    char* bx = smem + buf * 2 * TILE;
    char* bg = bx + TILE;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int id = threadIdx.x + 256 * u;
      const int row = id / (HPB / 16), ch = id % (HPB / 16);
      const int off = row * HPB + ((ch ^ swz(row)) << 4);
      *reinterpret_cast<u16x8*>(bx + off) = sx[u];
      *reinterpret_cast<u16x8*>(bg + off) = sg[u];
    }
  };
  if (ntiles > 0) {
    gload(tile0);
    lwrite(0);
  }
  __syncthreads();
  for (int64_t n = 0; n < ntiles; ++n) {
    const int buf = (int)(n & 1);
    if (n + 1 < ntiles) gload(tile0 + n + 1);
    if (wave_on) {
      const char* bx = smem + buf * 2 * TILE;
      const char* bg = bx + TILE;
      // preT (tokens x hidden) = x0 . W1^T ; dhT = d out . W2
      f32x16 pre, dh;
#pragma unroll
      for (int r = 0; r < 16; ++r) { pre[r] = b1v; dh[r] = 0.f; }
      const int rowoff = r32 * HPB, sw = swz(r32);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int ch = 2 * ks + hh;
        const bf16x8 ax = *reinterpret_cast<const bf16x8*>(bx + rowoff + ((ch ^ sw) << 4));
        const bf16x8 ag = *reinterpret_cast<const bf16x8*>(bg + rowoff + ((ch ^ sw) << 4));
        pre = P::mma(ax, w1f[ks], pre);
        dh = P::mma(ag, w2f[ks], dh);
      }
      f32x16 dact;
      big_act_fd<ACT>(a.act, pre, dact);   // pre := hidden activations
#pragma unroll
      for (int r = 0; r < 16; ++r) { dh[r] *= dact[r]; db1 += dh[r]; }
      const bf16x8 hb0 = P::from_acc<0>(pre), hb1 = P::from_acc<1>(pre);
      const bf16x8 gb0 = P::from_acc<0>(dh), gb1 = P::from_acc<1>(dh);
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const bf16x8 x0a = tr_frag<HPB>(bx, t, 0, lane), x0b = tr_frag<HPB>(bx, t, 1, lane);
        const bf16x8 g0a = tr_frag<HPB>(bg, t, 0, lane), g0b = tr_frag<HPB>(bg, t, 1, lane);
        dW1[t] = P::mma(x0a, gb0, dW1[t]);
        dW1[t] = P::mma(x0b, gb1, dW1[t]);
        dW2[t] = P::mma(g0a, hb0, dW2[t]);
        dW2[t] = P::mma(g0b, hb1, dW2[t]);
        if (want_b2) {
          float s = 0.f;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) s += (float)g0a[jj] + (float)g0b[jj];
          db2[t] += s;
        }
      }
    }
    if (n + 1 < ntiles) lwrite(buf ^ 1);
    __syncthreads();
  }
  if (!wave_on) return;
  // dump: accumulator element (reg r, lane) of dW1[t] = d W1[hidden 32c + lane&31][in 32t + acc_row(r, hh)]
  const int64_t item = ((int64_t)kn * HC + c) * a.nslab + slab;
  float* out = a.part + (item * 2 * HT) * 1024 + lane * 16;
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v1 = {dW1[t][4 * q], dW1[t][4 * q + 1], dW1[t][4 * q + 2], dW1[t][4 * q + 3]};
      f32x4 v2 = {dW2[t][4 * q], dW2[t][4 * q + 1], dW2[t][4 * q + 2], dW2[t][4 * q + 3]};
      *reinterpret_cast<f32x4*>(out + (int64_t)t * 1024 + 4 * q) = v1;
      *reinterpret_cast<f32x4*>(out + (int64_t)(HT + t) * 1024 + 4 * q) = v2;
    }
  db1 += __shfl_xor(db1, 32, 64);
  if (hh == 0) a.partb1[item * 32 + r32] = db1;
  if (want_b2) {
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      const float s = db2[t] + __shfl_xor(db2[t], 32, 64);
      if (hh == 0) a.partb2[((int64_t)kn * a.nslab + slab) * (32 * HT) + 32 * t + r32] = s;
    }
  }
}

struct DwRedArgs {
  const float *part, *partb1, *partb2;
  float *dW1, *db1, *dW2, *db2;   // (K, 2, H, h), (K, 2, H), (K, 2, h, H), (K, 2, h)
  const int64_t* sink;            // optional device table [K*2][4] of fp32 gradient addresses {W1, b1, W2, b2}: +=
  int K, h, H, HT, HC, nslab;
};
// One thread per dumped accumulator element: sums the slabs in slab order (fixed order => bit-reproducible).
__global__ void __launch_bounds__(256) k_big_dw_reduce(DwRedArgs a) {
  const int64_t n1 = (int64_t)a.K * 2 * a.HC * 2 * a.HT * 1024;
  const int64_t nb1 = (int64_t)a.K * 2 * a.HC * 32, nb2 = (int64_t)a.K * 2 * a.HT * 32;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n1) {
    const int e = (int)(idx & 1023);
    int64_t q = idx >> 10;
    const int tt = (int)(q % (2 * a.HT)); q /= 2 * a.HT;
    const int c = (int)(q % a.HC);
    const int kn = (int)(q / a.HC);
    const float* p = a.part + ((((int64_t)kn * a.HC + c) * a.nslab) * 2 * a.HT + tt) * 1024 + e;
    const int64_t stride = (int64_t)2 * a.HT * 1024;
    float s = 0.f;
    for (int sl = 0; sl < a.nslab; ++sl) s += p[sl * stride];
    const int r = e & 15, ln = e >> 4;
    const int hid = 32 * c + (ln & 31);
    const int t = tt % a.HT;
    const int feat = 32 * t + acc_row(r, ln >> 5);
    if (hid >= a.H || feat >= a.h) return;
    if (tt < a.HT) {   // dW1[hid][feat]
      if (a.sink) reinterpret_cast<float*>(a.sink[kn * 4 + 0])[(int64_t)hid * a.h + feat] += s;
      else a.dW1[((int64_t)kn * a.H + hid) * a.h + feat] = s;
    } else {           // dW2[feat][hid]
      if (a.sink) reinterpret_cast<float*>(a.sink[kn * 4 + 2])[(int64_t)feat * a.H + hid] += s;
      else a.dW2[((int64_t)kn * a.h + feat) * a.H + hid] = s;
    }
    return;
  }
  idx -= n1;
  if (idx < nb1) {
    const int col = (int)(idx & 31);
    int64_t q = idx >> 5;
    const int c = (int)(q % a.HC);
    const int kn = (int)(q / a.HC);
    const int hid = 32 * c + col;
    if (hid >= a.H) return;
    const float* p = a.partb1 + (((int64_t)kn * a.HC + c) * a.nslab) * 32 + col;
    float s = 0.f;
    for (int sl = 0; sl < a.nslab; ++sl) s += p[sl * 32];
    if (a.sink) reinterpret_cast<float*>(a.sink[kn * 4 + 1])[hid] += s;
    else a.db1[(int64_t)kn * a.H + hid] = s;
    return;
  }
  idx -= nb1;
  if (idx < nb2) {
    const int hp = 32 * a.HT;
    const int f = (int)(idx % hp);
    const int kn = (int)(idx / hp);
    if (f >= a.h) return;
    float s = 0.f;
    for (int sl = 0; sl < a.nslab; ++sl) s += a.partb2[((int64_t)kn * a.nslab + sl) * hp + f];
    if (a.sink) reinterpret_cast<float*>(a.sink[kn * 4 + 3])[f] += s;
    else a.db2[(int64_t)kn * a.h + f] = s;
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static bool big_shape_ok(int64_t h, int64_t H) {
  const int64_t hp = (h + 31) / 32 * 32, Hp = (H + 31) / 32 * 32;
  return h >= 97 && h <= 256 && h % 8 == 0 && H >= 33 && Hp + hp <= kBiasBlockFloats;
}
static int big_ht(int64_t h) { return h <= 128 ? 4 : 8; }
// geometry with hp rounded up to the instantiated tile counts (HT = 4 or 8)
static BigGeom big_geom_inst(int h, int H, int K) {
  BigGeom g = big_geom(h, H, K);
  g.HT = big_ht(h);
  g.hp = 32 * g.HT;
  g.item = 2 * g.HT * 1024;
  return g;
}

template <int HT, int ACT>
static void launch_chain(int mode, const BigArgs& a, unsigned grid, size_t lds, hipStream_t st) {
  auto opt = [&](const void* fn) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); };
  if (mode == M_FWD) { opt((const void*)k_big_chain<HT, M_FWD, ACT>); k_big_chain<HT, M_FWD, ACT><<<grid, 256, lds, st>>>(a); }
  else if (mode == M_INV) { opt((const void*)k_big_chain<HT, M_INV, ACT>); k_big_chain<HT, M_INV, ACT><<<grid, 256, lds, st>>>(a); }
  else { opt((const void*)k_big_chain<HT, M_CHAIN, ACT>); k_big_chain<HT, M_CHAIN, ACT><<<grid, 256, lds, st>>>(a); }
}
template <int HT, int ACT>
static void launch_nets(const NetsArgs& a, unsigned grid, size_t lds, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)k_big_nets<HT, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_big_nets<HT, ACT><<<grid, 256, lds, st>>>(a);
}
template <int HT, int ACT>
static void launch_dw(const DwBigArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)k_big_dw<HT, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_big_dw<HT, ACT><<<grid, 256, lds, st>>>(a);
}
#define BIG_DISPATCH(HTv, actv, CALL)                                           \
  do {                                                                          \
    if ((HTv) == 4) { if ((actv) == ACT_GELU) { CALL(4, ACT_GELU); } else { CALL(4, -1); } } \
    else { if ((actv) == ACT_GELU) { CALL(8, ACT_GELU); } else { CALL(8, -1); } }            \
  } while (0)

static size_t ring_lds(int HT) { return (size_t)(HT <= 4 ? RingCfg<4>::R : RingCfg<8>::R) * 2 * HT * 1024 + kBiasSlots * 4096; }

static void dw_big_geom(int64_t N, const BigGeom& g, int& nslab, int& slab_tiles) {
  const int64_t ntiles = (N + 31) / 32;
  const int64_t blocks0 = (int64_t)g.K * 2 * ((g.HC + 3) / 4);
  int64_t want = (768 + blocks0 - 1) / blocks0;          // ~3 blocks per CU in total
  const int64_t maxs = (ntiles + 15) / 16;               // at least 16 token tiles per slab
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  slab_tiles = (int)((ntiles + want - 1) / want);
  nslab = (int)((ntiles + slab_tiles - 1) / slab_tiles);
}

}  // namespace

// =====================================================================================================================
extern "C" int mivae_nice_big_supported(int64_t h, int64_t H, int dtype) {
  return dtype == MIVAE_BF16 && big_shape_ok(h, H) ? 1 : 0;
}

extern "C" int64_t mivae_nice_big_pack_bytes(int64_t K, int64_t h, int64_t H, int stream_kind) {
  if (K < 1 || !big_shape_ok(h, H) || stream_kind < 0 || stream_kind > ST_BIAS) return 0;
  const BigGeom g = big_geom_inst((int)h, (int)H, (int)K);
  if (stream_kind == ST_BIAS) return K * 2 * kBiasBlockFloats * 4;
  return stream_items(g, stream_kind) * g.item;
}

extern "C" int mivae_nice_big_pack(const int64_t* ptr_table, void* out, int64_t K, int64_t h, int64_t H,
                                   int stream_kind, void* stream) {
  MIVAE_REQUIRE(ptr_table && out, "nice_big_pack: null pointer");
  MIVAE_REQUIRE(K >= 1 && K <= 4096 && big_shape_ok(h, H), "nice_big_pack: unsupported shape K=%ld h=%ld H=%ld", (long)K, (long)h, (long)H);
  MIVAE_REQUIRE(stream_kind >= 0 && stream_kind <= ST_BIAS, "nice_big_pack: bad stream kind %d", stream_kind);
  MIVAE_REQUIRE(aligned16(out), "nice_big_pack: output must be 16-byte aligned");
  const BigGeom g = big_geom_inst((int)h, (int)H, (int)K);
  const int64_t n = stream_kind == ST_BIAS ? K * 2 * kBiasBlockFloats : stream_items(g, stream_kind) * (g.item / 2);
  k_big_pack<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(ptr_table, (char*)out, g, stream_kind);
  MIVAE_CHECK_LAUNCH("nice_big_pack");
  return MIVAE_OK;
}

static int big_common_checks(const char* name, int64_t Bn, int64_t S, int64_t D, int64_t H, int64_t K, int act,
                             int64_t mask_div) {
  MIVAE_REQUIRE(K >= 1 && K <= 4096 && D % 2 == 0 && big_shape_ok(D / 2, H), "%s: unsupported shape D=%ld H=%ld K=%ld", name,
                (long)D, (long)H, (long)K);
  MIVAE_REQUIRE(act >= 0 && act <= ACT_SIGMOID, "%s: unsupported activation id %d", name, act);
  MIVAE_REQUIRE(S >= 1 && mask_div >= 1 && Bn >= 0 && Bn * S < (1ll << 31), "%s: bad shape", name);
  return MIVAE_OK;
}

// z_out, tok_val as mivae_nice_stack_fwd; packed_stream = ST_FWD (inverse = 0) or ST_INV (inverse = 1) image.
extern "C" int mivae_nice_big_fwd(const void* z_in, void* z_out, void* x1_hist, const int64_t* mask,
                                  const void* packed_stream, const void* packed_bias, float* tok_val, int64_t Bn,
                                  int64_t mask_div, int64_t S, int64_t D, int64_t H, int64_t K, int act, int inverse,
                                  int base, float mu, float std, void* stream) {
  int rc = big_common_checks("nice_big_fwd", Bn, S, D, H, K, act, mask_div);
  if (rc) return rc;
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(z_in && z_out && packed_stream && packed_bias, "nice_big_fwd: null pointer");
  MIVAE_REQUIRE(!base || std > 0.f, "nice_big_fwd: std must be > 0");
  MIVAE_REQUIRE(!(inverse && x1_hist), "nice_big_fwd: no activation history in the inverse direction");
  MIVAE_REQUIRE(aligned16(z_in) && aligned16(z_out) && aligned16(packed_stream) && aligned16(packed_bias) &&
                (!x1_hist || aligned16(x1_hist)), "nice_big_fwd: operands must be 16-byte aligned");
  const BigGeom g = big_geom_inst((int)(D / 2), (int)H, (int)K);
  BigArgs a = {};
  a.z_in = (const T*)z_in; a.z_out = (T*)z_out; a.x1_hist = (T*)x1_hist; a.mask = mask;
  a.stream = (const char*)packed_stream; a.bias = (const float*)packed_bias; a.tok_val = tok_val;
  a.N = Bn * S; a.S = (int)S; a.mask_div = (int)mask_div; a.D = (int)D; a.K = (int)K; a.h = g.h; a.H = g.H; a.HC = g.HC;
  a.act = act; a.base = base; a.mu = mu;
  a.inv2var = base ? 1.0f / (2.0f * std * std) : 0.f;
  a.cst = base ? logf(std) + 0.5f * kLog2Pi : 0.f;
  const unsigned grid = (unsigned)((a.N + 127) / 128);
  const size_t lds = ring_lds(g.HT);
  const int mode = inverse ? M_INV : M_FWD;
#define CALL(HTc, ACTc) launch_chain<HTc, ACTc>(mode, a, grid, lds, (hipStream_t)stream)
  BIG_DISPATCH(g.HT, act, CALL);
#undef CALL
  MIVAE_CHECK_LAUNCH("nice_big_fwd");
  return MIVAE_OK;
}

extern "C" int64_t mivae_nice_big_dw_workspace_floats(int64_t N, int64_t h, int64_t H, int64_t K) {
  if (N < 1 || K < 1 || !big_shape_ok(h, H)) return 0;
  const BigGeom g = big_geom_inst((int)h, (int)H, (int)K);
  int nslab, slab_tiles;
  dw_big_geom(N, g, nslab, slab_tiles);
  return (int64_t)K * 2 * g.HC * nslab * (2 * g.HT * 1024 + 32) + (int64_t)K * 2 * nslab * g.hp;
}

// Backward of mivae_nice_big_fwd (forward direction): dz_in (N, D) and the parameter gradients.
//   packed_chain / packed_nets / packed_bias: ST_CHAIN / ST_NETS / ST_BIAS images of the same weights;
//   dSdT: scratch (K, 2, N, h) in the latent dtype; workspace: mivae_nice_big_dw_workspace_floats floats;
//   gradients go to dW1 (K,2,H,h), db1 (K,2,H), dW2 (K,2,h,H), db2 (K,2,h), or, when grad_ptrs (device int64[K][8] in
//   mivae_nice_pack's order) is given, are ADDED into those buffers.
extern "C" int mivae_nice_big_bwd(const void* z_in, const void* z_out, const void* x1_hist, const int64_t* mask,
                                  const void* packed_chain, const void* packed_nets, const void* packed_bias,
                                  const void* dz_out, const float* dlogp, void* dz_in, void* dSdT, float* workspace,
                                  float* dW1, float* db1, float* dW2, float* db2, const int64_t* grad_ptrs,
                                  int64_t Bn, int64_t mask_div, int64_t S, int64_t D, int64_t H, int64_t K, int act,
                                  int base, float mu, float std, void* stream) {
  int rc = big_common_checks("nice_big_bwd", Bn, S, D, H, K, act, mask_div);
  if (rc) return rc;
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(z_in && x1_hist && packed_chain && packed_nets && packed_bias && dz_in && dSdT && workspace,
                "nice_big_bwd: null pointer");
  MIVAE_REQUIRE(grad_ptrs || (dW1 && db1 && dW2 && db2), "nice_big_bwd: no gradient destination");
  MIVAE_REQUIRE(!base || (std > 0.f && z_out), "nice_big_bwd: base log-prob needs z_out and std > 0");
  MIVAE_REQUIRE(aligned16(z_in) && aligned16(x1_hist) && aligned16(dz_in) && aligned16(dSdT) && aligned16(workspace) &&
                (!z_out || aligned16(z_out)) && (!dz_out || aligned16(dz_out)) && aligned16(packed_chain) &&
                aligned16(packed_nets) && aligned16(packed_bias), "nice_big_bwd: operands must be 16-byte aligned");
  const BigGeom g = big_geom_inst((int)(D / 2), (int)H, (int)K);
  hipStream_t st = (hipStream_t)stream;
  const int64_t N = Bn * S;
  const unsigned grid = (unsigned)((N + 127) / 128);
  const size_t lds = ring_lds(g.HT);
  {
    BigArgs a = {};
    a.z_in = (const T*)z_in; a.z_out = (T*)const_cast<void*>(z_out); a.x1_hist = (T*)const_cast<void*>(x1_hist);
    a.mask = mask; a.stream = (const char*)packed_chain; a.bias = (const float*)packed_bias;
    a.N = N; a.S = (int)S; a.mask_div = (int)mask_div; a.D = (int)D; a.K = (int)K; a.h = g.h; a.H = g.H; a.HC = g.HC;
    a.act = act; a.base = base; a.mu = mu; a.inv2var = base ? 1.0f / (2.0f * std * std) : 0.f;
    a.dz_out = (const T*)dz_out; a.dlogp = dlogp; a.dz_in = (T*)dz_in; a.dSdT = (T*)dSdT;
#define CALL(HTc, ACTc) launch_chain<HTc, ACTc>(M_CHAIN, a, grid, lds, st)
    BIG_DISPATCH(g.HT, act, CALL);
#undef CALL
    MIVAE_CHECK_LAUNCH("nice_big_bwd_chain");
  }
  {
    NetsArgs a = {};
    a.z_in = (const T*)z_in; a.z_out = (const T*)z_out; a.mask = mask; a.stream = (const char*)packed_nets;
    a.bias = (const float*)packed_bias; a.dz_out = (const T*)dz_out; a.dlogp = dlogp; a.dSdT = (const T*)dSdT;
    a.dz_in = (T*)dz_in; a.N = N; a.S = (int)S; a.mask_div = (int)mask_div; a.D = (int)D; a.K = (int)K;
    a.h = g.h; a.H = g.H; a.HC = g.HC; a.act = act; a.base = base; a.mu = mu;
    a.inv2var = base ? 1.0f / (2.0f * std * std) : 0.f;
#define CALL(HTc, ACTc) launch_nets<HTc, ACTc>(a, grid, lds, st)
    BIG_DISPATCH(g.HT, act, CALL);
#undef CALL
    MIVAE_CHECK_LAUNCH("nice_big_bwd_nets");
  }
  {
    DwBigArgs a = {};
    a.z_in = (const T*)z_in; a.dSdT = (const T*)dSdT; a.mask = mask; a.stream = (const char*)packed_nets;
    a.bias = (const float*)packed_bias; a.N = N; a.S = (int)S; a.mask_div = (int)mask_div; a.D = (int)D; a.K = (int)K;
    a.h = g.h; a.H = g.H; a.HC = g.HC; a.act = act;
    dw_big_geom(N, g, a.nslab, a.slab_tiles);
    const int64_t nitems = (int64_t)K * 2 * g.HC * a.nslab;
    a.part = workspace;
    a.partb1 = workspace + nitems * 2 * g.HT * 1024;
    a.partb2 = a.partb1 + nitems * 32;
    dim3 dgrid((unsigned)a.nslab, (unsigned)(K * 2), (unsigned)((g.HC + 3) / 4));
    const size_t dlds = (size_t)4 * 32 * g.HT * 64;
#define CALL(HTc, ACTc) launch_dw<HTc, ACTc>(a, dgrid, dlds, st)
    BIG_DISPATCH(g.HT, act, CALL);
#undef CALL
    MIVAE_CHECK_LAUNCH("nice_big_dw");
    DwRedArgs r = {};
    r.part = a.part; r.partb1 = a.partb1; r.partb2 = a.partb2;
    r.dW1 = dW1; r.db1 = db1; r.dW2 = dW2; r.db2 = db2; r.sink = nullptr;
    r.K = (int)K; r.h = g.h; r.H = g.H; r.HT = g.HT; r.HC = g.HC; r.nslab = a.nslab;
    // the sink table is [K][8] = {W1s,b1s,W2s,b2s,W1t,b1t,W2t,b2t}: exactly [kn][4] with kn = 2k + net
    r.sink = grad_ptrs;
    const int64_t nthreads = (int64_t)K * 2 * g.HC * 2 * g.HT * 1024 + (int64_t)K * 2 * g.HC * 32 + (int64_t)K * 2 * g.HT * 32;
    k_big_dw_reduce<<<(unsigned)((nthreads + 255) / 256), 256, 0, st>>>(r);
    MIVAE_CHECK_LAUNCH("nice_big_dw_reduce");
  }
  return MIVAE_OK;
}
