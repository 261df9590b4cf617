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
#include <type_traits>
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

#ifndef BIG_DW_NBUF
#define BIG_DW_NBUF 6   // tile buffers of the weight-gradient kernel at h <= 128 (3 in use + prefetch depth)
#endif
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
// for the second MFMA, 4e-3 relative).  act_pair / act_pair_d / act_pair_fd below evaluate it stage by stage over the
// eight accumulator registers of a 32-wide hidden chunk (independent chains interleave), the polynomials on whole
// vectors (v_pk_fma_f32); v_med3_f32 instead of fmin / fmax avoids the canonicalising v_max the IEEE forms cost.
// max(x, 0) as ONE instruction: v_med3_f32 with a finite upper bound (fmaxf and med3 with an infinite bound lower to
// v_max(x, x) + v_max).  NOT inline asm: hipcc pads no MFMA -> VALU hazard wait states around an asm statement, and an
// asm v_max reading a fresh MFMA result returned the accumulator without the last product (found the hard way).
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 3.0e38f); }
constexpr float kGq3 = 0.004080009517455981f, kGq2 = -0.045314069231662556f, kGq1 = -0.4655827256271148f,
                kGq0 = -1.149269309564982f;
// Derivative with the SAME exponential: d/du Phi(-u) of the fit is Phi(-u) ln2 (q + u q'), so with
// r(u) = ln2 (4 q3 u^3 + 3 q2 u^2 + 2 q1 u + q0) and G = Phi(-u) (1 + u r(u)):  gelu'(x) = x > 0 ? 1 - G : G.
// It is the exact derivative of the function the forward evaluates; against the true gelu' its error is 6.2e-5 (the
// two-exponential form Phi + x phi(x) it replaces: 4.8e-5; bf16 rounding of the product that consumes it: 2e-3).
constexpr float kLn2f = 0.6931471805599453f;
constexpr float kGr3 = 4.f * kGq3 * kLn2f, kGr2 = 3.f * kGq2 * kLn2f, kGr1 = 2.f * kGq1 * kLn2f, kGr0 = kGq0 * kLn2f;
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
// Items of ITEM bytes are consumed strictly in stream order: the prologue consumes P0 items, then every group of G
// bodies the next GI = G * IPB.  Slot of item q = q % R with R = 2 GI: while a group is being consumed the next
// one is in flight (LDS-DMA, 16 B / lane).  One barrier per GROUP: enter_group(gi) waits for this wave's own DMA
// pieces (everything issued one group ago), joins the barrier (=> the group has landed for every wave, and every
// wave has finished reading the previous group) and issues the group after next.
template <int ITEM, int GI, int P0>
struct Ring {
  static constexpr int R = 2 * GI > P0 + GI ? 2 * GI : P0 + GI;
  static constexpr int PI = ITEM / 8192;   // DMA pieces (1 KiB) per wave and item (8 waves)
  const char* src;
  char* lds;
  int total, issued, slot;   // items
  int wave, lane;

  __device__ __forceinline__ void issue_one() {
    const char* s = src + (int64_t)issued * ITEM + wave * 1024 + lane * 16;
    char* d = lds + slot * ITEM + wave * 1024;
#pragma unroll
    for (int p = 0; p < PI; ++p)
      __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(s + p * 8192), (lds_void_t*)(d + p * 8192), 16, 0, 0);
    ++issued;
    slot = slot + 1 == R ? 0 : slot + 1;
  }
  __device__ __forceinline__ void issue_upto(int lim) {
    while (issued < lim && issued < total) issue_one();
  }
  __device__ __forceinline__ void init(const char* s, char* l, int total_, int wave_, int lane_) {
    static_assert(P0 + GI <= R, "prologue + first group must fit the ring");
    src = s; lds = l; total = total_; issued = 0; slot = 0; wave = wave_; lane = lane_;
    issue_upto(P0 + GI);
  }
  __device__ __forceinline__ void sync() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  // prologue: items [0, P0) (and group 0) readable afterwards
  __device__ __forceinline__ void enter_prologue() { sync(); }
  // group gi: items [P0 + GI gi, P0 + GI (gi + 1)) readable afterwards; the group after it is issued
  __device__ __forceinline__ void enter_group(int gi) {
    sync();
    issue_upto(P0 + GI * (gi + 2));
  }
  __device__ __forceinline__ const char* item(int q) const { return lds + (q % R) * ITEM; }
};

// ---- pack ----------------------------------------------------------------------------------------------
// ptrs[k][8] = {W1s, b1s, W2s, b2s, W1t, b1t, W2t, b2t} (fp32, torch.nn.Linear layout: W1 (H, h), W2 (h, H)).
// One thread per eight bf16 elements of the stream (or per float of the bias array).
__device__ __forceinline__ void task_of(const BigGeom& g, int st, int64_t i, int& k, int& net, int& c) {
  c = (int)(i % g.HC);
  if (st == ST_CHAIN) { k = g.K - 1 - (int)(i / g.HC); net = 0; return; }
  const int kk = (int)(i / (2 * g.HC)), nn = (int)((i / g.HC) & 1);
  if (st == ST_FWD) { k = kk; net = nn; }
  else if (st == ST_INV) { k = g.K - 1 - kk; net = 1 - nn; }
  else { k = g.K - 1 - kk; net = nn; }
}
__device__ __forceinline__ void pack_elem(const int64_t* __restrict__ ptrs, char* __restrict__ out, const BigGeom& g, int st,
                                          int64_t idx) {
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
  // one thread per 16-byte group of the stream: the eight k-slots j of one lane of one fragment (eight loads in flight,
  // one 16-byte store; was one element per thread: C5 137 us, C3 21 us per launch)
  const int per_item = g.item / 2;
  const int64_t nitems = stream_items(g, st), Tn = stream_tasks(g, st);
  if (idx >= nitems * (per_item / 8)) return;
  const int64_t q = idx / (per_item / 8);
  const int e = (int)(idx % (per_item / 8)) * 8;
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
  u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
  if (task < Tn) {
    int k, net, c;
    task_of(g, st, task, k, net, c);
    const float* W1 = reinterpret_cast<const float*>(ptrs[k * 8 + net * 4 + 0]);
    const float* W2 = reinterpret_cast<const float*>(ptrs[k * 8 + net * 4 + 2]);
    // fragments of v_mfma_f32_16x16x32_bf16: lane (r = lane & 15, g = lane >> 4) holds A[row r][k-slot (g, j)]
    const int frag = e >> 9, lane = (e >> 3) & 63;
    const int r = lane & 15, gq = lane >> 4;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int perm = j < 4 ? 4 * gq + j : 16 + 4 * gq + (j - 4);   // k order of an accumulator pair used as B operand
      v[j] = 0.f;
      if (kind == KIND_W1) {          // frag = u * HT + ks: A[row = hidden 32c + 16u + r][k = in 32ks + 8g + j]
        const int u = frag / g.HT, ks = frag % g.HT;
        const int row = 32 * c + 16 * u + r, col = 32 * ks + 8 * gq + j;
        if (row < g.H && col < g.h) v[j] = W1[(int64_t)row * g.h + col];
      } else if (kind == KIND_W2) {   // frag = v: A[row = out 16v + r][k = hidden 32c + perm]
        const int row = 16 * frag + r, col = 32 * c + perm;
        if (row < g.h && col < g.H) v[j] = W2[(int64_t)row * g.H + col];
      } else if (kind == KIND_W2T) {  // frag = u * HT + ks: A[row = hidden 32c + 16u + r][k = out 32ks + 8g + j] = W2[out][hidden]
        const int u = frag / g.HT, ks = frag % g.HT;
        const int row = 32 * ks + 8 * gq + j, col = 32 * c + 16 * u + r;
        if (row < g.h && col < g.H) v[j] = W2[(int64_t)row * g.H + col];
      } else {                        // KIND_W1T, frag = v: A[row = in 16v + r][k = hidden 32c + perm] = W1[hidden][in]
        const int row = 32 * c + perm, col = 16 * frag + r;
        if (row < g.H && col < g.h) v[j] = W1[(int64_t)row * g.h + col];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f32_to_bf16(v[j]);
  }
  *reinterpret_cast<u16x8*>(reinterpret_cast<unsigned short*>(out) + idx * 8) = o;
}
__global__ void __launch_bounds__(256) k_big_pack(const int64_t* __restrict__ ptrs, char* __restrict__ out, BigGeom g, int st) {
  pack_elem(ptrs, out, g, st, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}
// several streams of the same weights in ONE launch: blocks [first[s], first[s + 1]) pack stream s
struct PackMulti {
  char* out[ST_BIAS + 1];
  unsigned first[ST_BIAS + 2];
};
__global__ void __launch_bounds__(256) k_big_pack_multi(const int64_t* __restrict__ ptrs, PackMulti m, BigGeom g) {
  int st = 0;
#pragma unroll
  for (int s2 = 1; s2 <= ST_BIAS; ++s2)
    if (blockIdx.x >= m.first[s2]) st = s2;
  pack_elem(ptrs, m.out[st], g, st, (int64_t)(blockIdx.x - m.first[st]) * blockDim.x + threadIdx.x);
}

// ---- forward / inverse / backward-chain kernel --------------------------------------------------------------
// Tiling: v_mfma_f32_16x16x32_bf16, a wave owns 16 tokens (columns), 8 waves = 128 tokens per block, two waves per
// SIMD (<= 256 registers): the VALU-heavy activation of one wave runs at the full VALU rate beside the other
// wave's MFMAs.  Lane (tk = lane & 15, g = lane >> 4); accumulator tile v (16 features x 16 tokens) holds
// features 16 v + 4 g + {0..3} of token tk in its 4 registers.
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
  int zero;             // always 0, but opaque to the compiler: see k_big_chain (`live`, ring.init)
  float mu, inv2var, cst;
  // CHAIN
  const T* dz_out;      // (N, D) or null
  const float* dlogp;   // (Bn) cotangent of the per-sentence sum of tok_val (Bn = N / S; S = 1: per token), or null
  T* dz_in;             // (N, D): the x1 half is written by CHAIN, the x0 half by k_big_nets
  T* dSdT;              // (K, 2, N, h): d s_k, d t_k
};

enum Mode { M_FWD = 0, M_INV = 1, M_CHAIN = 2 };

// bodies per barrier group: R = max(2 GI, P0 + GI) items of 2 HT KiB + the 16 KiB of bias slots must fit the 160 KiB of LDS
#ifndef BIG_G2
#define BIG_G2 4
#endif
template <int HT> struct RingCfg { static constexpr int G2 = HT <= 4 ? BIG_G2 : 2, G3 = HT <= 4 ? 3 : 1; };

__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// two stacked accumulator tiles (32 rows) -> the B fragment of a K = 32 product over those rows; element j of
// lane group g is row (j < 4 ? 4 g + j : 16 + 4 g + j - 4): the order k_big_pack gives the other operand.
__device__ __forceinline__ bf16x8 pair_to_b(const f32x4& lo, const f32x4& hi) {
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  const f32x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_convertvector(v, bf16x8);
}
// token-row I/O in the two layouts.  B fragments (natural k): lane holds features 32 ks + 8 g + {0..7}.
template <int HT>
__device__ __forceinline__ void load_bfrags(const T* __restrict__ row, bool on, int h, int g, bf16x8 (&f)[HT]) {
#pragma unroll
  for (int ks = 0; ks < HT; ++ks) {
    const int ft = 32 * ks + 8 * g;
    u16x8 raw = {0, 0, 0, 0, 0, 0, 0, 0};
    if (on && ft < h) raw = *reinterpret_cast<const u16x8*>(row + ft);
    f[ks] = __builtin_bit_cast(bf16x8, raw);
  }
}
// accumulator layout: tile v holds features 16 v + 4 g + {0..3}
template <int NT>
__device__ __forceinline__ void load_tiles(const T* __restrict__ row, bool on, int h, int g, f32x4 (&x)[NT]) {
#pragma unroll
  for (int v = 0; v < NT; ++v) {
    const int ft = 16 * v + 4 * g;
    float t4[4] = {0.f, 0.f, 0.f, 0.f};
    if (on && ft < h) RowIO<T, 4>::load(row + ft, t4);
    x[v] = f32x4{t4[0], t4[1], t4[2], t4[3]};
  }
}
template <int NT>
__device__ __forceinline__ void store_tiles(T* __restrict__ row, bool on, int h, int g, const f32x4 (&x)[NT]) {
#pragma unroll
  for (int v = 0; v < NT; ++v) {
    const int ft = 16 * v + 4 * g;
    const float t4[4] = {x[v][0], x[v][1], x[v][2], x[v][3]};
    if (on && ft < h) RowIO<T, 4>::store(row + ft, t4);
  }
}
// activation of the two tiles of a 32-wide hidden chunk (handled as one 8-register vector)
typedef float f32x8v __attribute__((ext_vector_type(8)));
template <int ACT>
__device__ __forceinline__ void act_pair(int act, f32x4& lo, f32x4& hi) {
  if constexpr (ACT == ACT_GELU) {
    f32x8v x = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, a, q;
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = __builtin_amdgcn_fmed3f(fabsf(x[r]), 0.f, 5.0f);
    q = kGq3 * a + kGq2;          // whole-vector cubic: v_pk_fma_f32 (see act_pair_d)
    q = q * a + kGq1;
    q = q * a + kGq0;
    q = q * a - 1.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) q[r] = __builtin_amdgcn_exp2f(q[r]);
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = fmaf(-fabsf(x[r]), q[r], relu1(x[r]));
    lo = f32x4{x[0], x[1], x[2], x[3]}; hi = f32x4{x[4], x[5], x[6], x[7]};
  } else {
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) { t[r] = lo[r]; t[4 + r] = hi[r]; t[8 + r] = 0.f; t[12 + r] = 0.f; }
    act_tile<-1, true>(act, t);
#pragma unroll
    for (int r = 0; r < 4; ++r) { lo[r] = t[r]; hi[r] = t[4 + r]; }
  }
}
// derivative of the activation at the pre-activations (lo, hi) -> (dlo, dhi)
template <int ACT>
__device__ __forceinline__ void act_pair_d(int act, const f32x4& lo, const f32x4& hi, f32x4& dlo, f32x4& dhi) {
  if constexpr (ACT == ACT_GELU) {
    f32x8v x = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, a, q, e, d;
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = __builtin_amdgcn_fmed3f(fabsf(x[r]), 0.f, 5.0f);
    // the two cubics on whole vectors: v_pk_fma_f32, two elements per instruction (this phase runs between the MFMA
    // phases, where the packed form is a gain; the file is built without SLP vectorisation for the MFMA-paced loops)
    q = kGq3 * a + kGq2;
    e = kGr3 * a + kGr2;
    q = q * a + kGq1;
    e = e * a + kGr1;
    q = q * a + kGq0;
    e = e * a + kGr0;
    q = q * a - 1.0f;
    e = e * a + 1.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) q[r] = __builtin_amdgcn_exp2f(q[r]);                          // Phi(-|x|)
    e = 0.5f - q * e;                                                                          // 1/2 - G (see kGr3 above)
    // gelu'(x) = x > 0 ? 1 - G : G = 1/2 + copysign(1/2 - G, x): one v_bfi_b32 instead of compare, subtract and select
#pragma unroll
    for (int r = 0; r < 8; ++r) e[r] = __builtin_copysignf(e[r], x[r]);   // (1/2 - G >= 0: G <= 1/2 on u >= 0)
    d = e + 0.5f;
    dlo = f32x4{d[0], d[1], d[2], d[3]}; dhi = f32x4{d[4], d[5], d[6], d[7]};
  } else {
    f32x16 t, d;
#pragma unroll
    for (int r = 0; r < 4; ++r) { t[r] = lo[r]; t[4 + r] = hi[r]; t[8 + r] = 0.f; t[12 + r] = 0.f; }
    act_tile_fd<-1, true>(act, t, d);
#pragma unroll
    for (int r = 0; r < 4; ++r) { dlo[r] = d[r]; dhi[r] = d[4 + r]; }
  }
}
// value and derivative of the activation on the two token sub-tiles of a (32 tokens x 16 hidden) block
template <int ACT>
__device__ __forceinline__ void act_pair_fd(int act, f32x4& lo, f32x4& hi, f32x4& dlo, f32x4& dhi) {
  if constexpr (ACT == ACT_GELU) {
    f32x8v x = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}, a, q, e, d;
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = __builtin_amdgcn_fmed3f(fabsf(x[r]), 0.f, 5.0f);
    // the two cubics on whole vectors: v_pk_fma_f32, two elements per instruction (this phase runs between the MFMA
    // phases, where the packed form is a gain; the file is built without SLP vectorisation for the MFMA-paced loops)
    q = kGq3 * a + kGq2;
    e = kGr3 * a + kGr2;
    q = q * a + kGq1;
    e = e * a + kGr1;
    q = q * a + kGq0;
    e = e * a + kGr0;
    q = q * a - 1.0f;
    e = e * a + 1.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) q[r] = __builtin_amdgcn_exp2f(q[r]);                          // Phi(-|x|)
    e = 0.5f - q * e;                                                                          // 1/2 - G (see kGr3 above)
    // gelu'(x) = x > 0 ? 1 - G : G = 1/2 + copysign(1/2 - G, x): one v_bfi_b32 instead of compare, subtract and select
#pragma unroll
    for (int r = 0; r < 8; ++r) e[r] = __builtin_copysignf(e[r], x[r]);   // (1/2 - G >= 0: G <= 1/2 on u >= 0)
    d = e + 0.5f;
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = fmaf(-fabsf(x[r]), q[r], relu1(x[r]));
    lo = f32x4{x[0], x[1], x[2], x[3]}; hi = f32x4{x[4], x[5], x[6], x[7]};
    dlo = f32x4{d[0], d[1], d[2], d[3]}; dhi = f32x4{d[4], d[5], d[6], d[7]};
  } else {
    f32x16 t, d;
#pragma unroll
    for (int r = 0; r < 4; ++r) { t[r] = lo[r]; t[4 + r] = hi[r]; t[8 + r] = 0.f; t[12 + r] = 0.f; }
    act_tile_fd<-1, true>(act, t, d);
#pragma unroll
    for (int r = 0; r < 4; ++r) { lo[r] = t[r]; hi[r] = t[4 + r]; dlo[r] = d[r]; dhi[r] = d[4 + r]; }
  }
}

// four consecutive floats from LDS (biases of features 16 v + 4 g + {0..3})
__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// d s_k, d t_k of a padding token are zeros: the weight-gradient kernel reads every row below N unmasked
__device__ __forceinline__ void zero_dsdt_rows(T* dSdT, int64_t N, int K, int h, int64_t tok, int g) {
  const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int kn = 0; kn < 2 * K; ++kn) {
    T* row = dSdT + ((int64_t)kn * N + tok) * h;
    for (int f = 8 * g; f < h; f += 32) *reinterpret_cast<u16x8*>(row + f) = zero;
  }
}

template <int HT, int MODE, int ACT>
__global__ void __launch_bounds__(512) k_big_chain(BigArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NF = 2 * HT;           // fragments per item; also accumulator tiles of the h features
  constexpr int ITEM = NF * 1024;
  constexpr int G = RingCfg<HT>::G2;
  typedef Ring<ITEM, 2 * G, 2> RingT;
  constexpr int R = RingT::R;
  constexpr int NETS = MODE == M_CHAIN ? 1 : 2;
  const int lane = threadIdx.x & 63;
  const int tk = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = a.h, HC = a.HC;
  const int64_t tok = ((int64_t)blockIdx.x * 8 + wave) * 16 + tk;
  const bool valid = tok < a.N;
  float m = 0.f, dl = 0.f;
  if (valid) {
    m = 1.f;
    if (a.mask) {
      const int64_t bn = tok / a.S;
      m = (float)a.mask[(bn / a.mask_div) * a.S + (tok - bn * a.S)];
    }
    if (MODE == M_CHAIN && a.dlogp) dl = a.dlogp[tok / a.S] * m;   // one cotangent per sentence (S = 1: per token)
  }
  // (`a.zero` is 0.  With `live` and the ring length provably loop-invariant hipcc unswitches the task loop into two
  //  copies and the h = 256 instantiations, already at 256 registers, spill 136-292 bytes per lane inside it: C5 forward
  //  1.32 -> 1.67 ms.  An opaque term keeps the single loop.)
  const bool live = (__any(m != 0.f) != 0) && !(a.zero & 2);
  if (!block_live(a.mask, a.N, a.S, a.mask_div, lane)) {
    // the whole block is padding: outputs are zeros; the history / d s, d t rows of masked tokens are never read
    if (valid) {
      const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
      if (MODE == M_CHAIN) {
        T* drow = a.dz_in + tok * a.D + h;
        for (int f = 8 * g; f < h; f += 32) *reinterpret_cast<u16x8*>(drow + f) = zero;
        zero_dsdt_rows(a.dSdT, a.N, a.K, h, tok, g);
      } else {
        T* orow = a.z_out + tok * a.D;
        for (int f = 8 * g; f < a.D; f += 32) *reinterpret_cast<u16x8*>(orow + f) = zero;
        if (g == 0 && a.tok_val) a.tok_val[tok] = 0.f;
      }
    }
    return;
  }
  const bool on = valid && m != 0.f;   // masked tokens are computed on zeros (their outputs are multiplied by the mask)
  const T* zrow = a.z_in + tok * a.D;
  bf16x8 x0f[HT];
  load_bfrags<HT>(zrow, on, h, g, x0f);
  f32x4 x1[NF];   // FWD / INV: the transformed half; CHAIN: d loss / d x1
  if constexpr (MODE == M_CHAIN) {
    load_tiles<NF>(a.dz_out ? a.dz_out + tok * a.D + h : zrow, valid && a.dz_out != nullptr, h, g, x1);
    if (a.base && a.dlogp) {   // d/dz_K of the base log-prob: dl * -(z_K - mu) / std^2
      f32x4 zk[NF];
      const float c2 = 2.0f * a.inv2var;
      load_tiles<NF>(a.z_out + tok * a.D + h, valid, h, g, zk);
#pragma unroll
      for (int v = 0; v < NF; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) x1[v][r] += (16 * v + 4 * g + r < h) ? -dl * (zk[v][r] - a.mu) * c2 : 0.f;
    }
  } else {
    load_tiles<NF>(zrow + h, on, h, g, x1);
  }
  f32x4 ld4 = {0.f, 0.f, 0.f, 0.f};   // log-det partial sums, one per accumulator register

  RingT ring;
  float* bias_lds = reinterpret_cast<float*>(smem + R * ITEM);
  const int NJ = a.K * NETS;            // nets in stream order
  const int Tn = NJ * HC;               // tasks
  auto bias_src = [&](int j) -> const char* {   // bias block of net-index j in stream order
    int k, net;
    if (MODE == M_CHAIN) { k = a.K - 1 - j; net = 0; }
    else if (MODE == M_FWD) { k = j >> 1; net = j & 1; }
    else { k = a.K - 1 - (j >> 1); net = 1 - (j & 1); }
    return reinterpret_cast<const char*>(a.bias + (int64_t)(k * 2 + net) * kBiasBlockFloats);
  };
  auto issue_bias = [&](int j) {   // 4 KiB = 4 pieces; waves 4-7 repeat pieces 0-3 (uniform DMA count per wave)
    __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(bias_src(j) + (wave & 3) * 1024 + lane * 16),
                                     (lds_void_t*)((char*)bias_lds + (j & 3) * 4096 + (wave & 3) * 1024), 16, 0, 0);
  };
  int nbias = 0;
  for (; nbias < kBiasSlots && nbias < NJ; ++nbias) issue_bias(nbias);
  ring.init(a.stream, smem, (a.zero & 1) ? 0 : 2 * Tn + 2, wave, lane);

  f32x4 acc[NF];
#pragma unroll
  for (int v = 0; v < NF; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};

  // first layer of one chunk: tiles (lo, hi) = b1 + W1[c] . x0^T
  auto l1 = [&](const char* item, const float* b1c, f32x4& lo, f32x4& hi) {
    const bf16x8* w = reinterpret_cast<const bf16x8*>(item) + lane;
    lo = lds4(b1c + 4 * g); hi = lds4(b1c + 16 + 4 * g);
#pragma unroll
    for (int ks = 0; ks < HT; ++ks) {
      lo = mma16(w[ks * 64], x0f[ks], lo);
      hi = mma16(w[(HT + ks) * 64], x0f[ks], hi);
    }
  };
  // ---- prologue: L1(0) -> act -> hid(0); L1(1) -> (a1l, a1h)
  ring.enter_prologue();
  bf16x8 hb;
  f32x4 a1l, a1h;
  {
    f32x4 l, hi2;
    l1(ring.item(0), bias_lds, l, hi2);
    act_pair<ACT>(a.act, l, hi2);
    hb = pair_to_b(l, hi2);
    l1(ring.item(1), bias_lds + 32, a1l, a1h);   // task 1: chunk 1 of net 0 (HC >= 2)
  }
  // CHAIN: fold of net j (step k = K-1-j) -- x1 holds d loss / d x1_k; emit d s_k, d t_k and step back.  Tile by tile
  // (no NF-sized fp32 temporaries: at h = 256 they would not fit next to acc and x1); masked tokens read the history as
  // zeros (the forward does not write the history of all-padding blocks).  At h <= 128 the history rows were requested a
  // group earlier (xpre); at h = 256 there is no register room for that and they are loaded here.
  constexpr bool kPrefetchHist = MODE == M_CHAIN && HT <= 4;
  u16x4 xpre[kPrefetchHist ? NF : 1];
  int xpre_net = -1;
  auto load_xpre = [&](int j) {
    if constexpr (kPrefetchHist) {
      const T* xh = a.x1_hist + ((int64_t)(a.K - 1 - j) * a.N + tok) * h;
#pragma unroll
      for (int v = 0; v < NF; ++v) {
        const int ft = 16 * v + 4 * g;
        const u16x4 z4 = {0, 0, 0, 0};
        xpre[v] = z4;
        if (on && ft < h) xpre[v] = *reinterpret_cast<const u16x4*>(xh + ft);
      }
    }
  };
  auto fold_chain = [&](int j) {
    if constexpr (MODE == M_CHAIN) {
      const float* b2 = bias_lds + (j & 3) * kBiasBlockFloats + HC * 32 + 4 * g;
      const int k = a.K - 1 - j;
      T* dS = a.dSdT + ((int64_t)(k * 2) * a.N + tok) * h;
      T* dT = dS + a.N * (int64_t)h;
      const T* xh = a.x1_hist + ((int64_t)k * a.N + tok) * h;
      const bool have = kPrefetchHist && xpre_net == j;       // wave-uniform
#pragma unroll
      for (int v = 0; v < NF; ++v) {
        const int ft = 16 * v + 4 * g;
        float xp[4] = {0.f, 0.f, 0.f, 0.f};
        if (have) {
          if constexpr (kPrefetchHist) {
#pragma unroll
            for (int r = 0; r < 4; ++r) xp[r] = bf16_to_f32(xpre[v][r]);
          }
        } else if (on && ft < h) {
          RowIO<T, 4>::load(xh + ft, xp);
        }
        if constexpr (HT <= 4) {
          // whole-vector arithmetic (v_pk_mul / v_pk_fma: two elements per instruction)
          const f32x4 g1 = x1[v] * m;
          const f32x4 t4 = (acc[v] + lds4(b2 + 16 * v)) * 1.44269504088896341f;
          f32x4 e4;
  #pragma unroll
          for (int r = 0; r < 4; ++r) e4[r] = __builtin_amdgcn_exp2f(t4[r]);
          const f32x4 ge = g1 * e4;
          const f32x4 ds4 = ge * f32x4{xp[0], xp[1], xp[2], xp[3]} + dl;
          x1[v] = ge;
          acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (valid && ft < h) {
            const float ds[4] = {ds4[0], ds4[1], ds4[2], ds4[3]}, dt[4] = {g1[0], g1[1], g1[2], g1[3]};
            RowIO<T, 4>::store(dS + ft, ds);
            RowIO<T, 4>::store(dT + ft, dt);
          }
        } else {   // h = 256: the vector temporaries cost registers the kernel does not have (C5: 1.09 -> 1.24 ms)
          const f32x4 bt = lds4(b2 + 16 * v);
          float ds[4], dt[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float g1 = x1[v][r] * m;
            const float ge = g1 * fast_exp(acc[v][r] + bt[r]);
            dt[r] = g1;
            ds[r] = ge * xp[r] + dl;
            x1[v][r] = ge;
            acc[v][r] = 0.f;
          }
          if (valid && ft < h) {
            RowIO<T, 4>::store(dS + ft, ds);
            RowIO<T, 4>::store(dT + ft, dt);
          }
        }
      }
    }
  };
  // The loop nest is written twice -- with and without the arithmetic -- so that the live path carries its register
  // tiles through straight-line loops without conditional merges.
  if (!live) {
    if (MODE == M_CHAIN && valid) zero_dsdt_rows(a.dSdT, a.N, a.K, h, tok, g);
    int i = 0;
    for (int j = 0; j < NJ; ++j)
      for (int c = 0; c < HC; ++c, ++i) {
        if (i % G == 0) {
          ring.enter_group(i / G);
          while (nbias < j + kBiasSlots - (MODE == M_CHAIN ? 1 : 0) && nbias < NJ) issue_bias(nbias++);
        }
      }
  } else {
    int i = 0;
#pragma unroll 1
    for (int j = 0; j < NJ; ++j) {
#pragma unroll 1
      for (int c = 0; c < HC; ++c, ++i) {
        if (i % G == 0) {
          ring.enter_group(i / G);
          // nets below j are folded by every wave: their bias slots are free (CHAIN folds net j-1 just below)
          while (nbias < j + kBiasSlots - (MODE == M_CHAIN ? 1 : 0) && nbias < NJ) issue_bias(nbias++);
          if constexpr (MODE == M_CHAIN && HT <= 4) {
            // x1 history of this net's step: requested one group ahead of the fold that consumes it
            if (c > 0 && xpre_net != j) { load_xpre(j); xpre_net = j; }
          }
        }
        // A group boundary waits for every outstanding vector memory operation of the wave (vmcnt(0)), so stores are
        // issued right AFTER one, not before: CHAIN folds net j-1 (d s / d t stores) here and they have the whole group
        // to complete.  (The same move for the forward's x1-history store costs registers in the task loop and measured
        // slower: C3 237 -> 249 us, C5 1.36 -> 1.61 ms; it stays in the fold.)
        if constexpr (MODE == M_CHAIN) {
          if (c == 0 && j > 0) fold_chain(j - 1);
        }
        const char* it2 = ring.item(2 + 2 * i);
        const char* it1 = ring.item(3 + 2 * i);
        int c2 = c + 2, j2 = j;       // task i + 2
        if (c2 >= HC) { c2 -= HC; ++j2; }
        // three independent streams in one basic block (tasks past the end run on the stream's zero padding):
        //   L2(i): acc[v] += W2 frags . hid(i);  L1(i+2): (a2l, a2h) = b1 + W1 frags . x0;  act(i+1) -> hid(i+1)
        const bf16x8* w2 = reinterpret_cast<const bf16x8*>(it2) + lane;
        f32x4 a2l, a2h;
        if constexpr (HT <= 4) {
          // Phases pinned in program order: all fragment reads of the body first, the activation (VALU) while they
          // are in flight, then the MFMAs back to back.  Left to itself the scheduler keeps the reads one MFMA ahead,
          // which exposes the LDS latency at every other MFMA; the co-resident wave fills the VALU / MFMA phases.
          const bf16x8* w1 = reinterpret_cast<const bf16x8*>(it1) + lane;
          const float* b1c = bias_lds + (j2 & 3) * kBiasBlockFloats + 32 * c2;
          bf16x8 f2[NF], f1[2 * HT];
#pragma unroll
          for (int v = 0; v < NF; ++v) f2[v] = w2[v * 64];
#pragma unroll
          for (int u = 0; u < 2 * HT; ++u) f1[u] = w1[u * 64];
          a2l = lds4(b1c + 4 * g); a2h = lds4(b1c + 16 + 4 * g);
          __builtin_amdgcn_sched_barrier(0);
          act_pair<ACT>(a.act, a1l, a1h);
          const bf16x8 hbn = pair_to_b(a1l, a1h);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int v = 0; v < NF; ++v) acc[v] = mma16(f2[v], hb, acc[v]);
#pragma unroll
          for (int ks = 0; ks < HT; ++ks) {
            a2l = mma16(f1[ks], x0f[ks], a2l);
            a2h = mma16(f1[HT + ks], x0f[ks], a2h);
          }
          hb = hbn;
        } else {
#pragma unroll
          for (int v = 0; v < NF; ++v) acc[v] = mma16(w2[v * 64], hb, acc[v]);
          l1(it1, bias_lds + (j2 & 3) * kBiasBlockFloats + 32 * c2, a2l, a2h);
          act_pair<ACT>(a.act, a1l, a1h);
          hb = pair_to_b(a1l, a1h);
        }
        a1l = a2l; a1h = a2h;
      }
      // ---- fold the finished net into the chain (CHAIN: deferred to the start of the next net, see above)
      if constexpr (MODE != M_CHAIN) {
        const float* b2 = bias_lds + (j & 3) * kBiasBlockFloats + HC * 32 + 4 * g;
        const int nn = MODE == M_CHAIN ? 0 : (j & 1);
        const int k = MODE == M_FWD ? (j >> 1) : a.K - 1 - (MODE == M_CHAIN ? j : (j >> 1));
        if constexpr (MODE == M_FWD) {
          if (nn == 0) {
            if (a.x1_hist) store_tiles<NF>(a.x1_hist + ((int64_t)k * a.N + tok) * h, valid, h, g, x1);
#pragma unroll
            for (int v = 0; v < NF; ++v) {     // whole-vector arithmetic: v_pk_add / v_pk_mul, two elements per instruction
              const f32x4 s4 = acc[v] + lds4(b2 + 16 * v);
              const f32x4 t4 = s4 * 1.44269504088896341f;
              f32x4 e4;
#pragma unroll
              for (int r = 0; r < 4; ++r) e4[r] = __builtin_amdgcn_exp2f(t4[r]);
              x1[v] *= e4;
              if constexpr (HT <= 4) ld4 += s4; else ld4[0] += (s4[0] + s4[1]) + (s4[2] + s4[3]);   // (h = 256: no registers for four sums)
              acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          } else {
#pragma unroll
            for (int v = 0; v < NF; ++v) {
              x1[v] = (x1[v] + (acc[v] + lds4(b2 + 16 * v))) * m;
              acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
        } else if constexpr (MODE == M_INV) {
          if (nn == 0) {   // t first: x1 -= t
#pragma unroll
            for (int v = 0; v < NF; ++v) {
              x1[v] -= acc[v] + lds4(b2 + 16 * v);
              acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          } else {         // then x1 = x1 exp(-s) m
#pragma unroll
            for (int v = 0; v < NF; ++v) {
              const f32x4 s4 = acc[v] + lds4(b2 + 16 * v);
              const f32x4 t4 = s4 * -1.44269504088896341f;
              f32x4 e4;
#pragma unroll
              for (int r = 0; r < 4; ++r) e4[r] = __builtin_amdgcn_exp2f(t4[r]);
              x1[v] = (x1[v] * e4) * m;
              if constexpr (HT <= 4) ld4 -= s4; else ld4[0] -= (s4[0] + s4[1]) + (s4[2] + s4[3]);
              acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
        }
      }
    }
    if constexpr (MODE == M_CHAIN) fold_chain(NJ - 1);
  }

  // ---- outputs ---------------------------------------------------------------------------------------------
  if (!live) {   // an all-padding tile of a live block: zeros
#pragma unroll
    for (int v = 0; v < NF; ++v) x1[v] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (MODE == M_CHAIN) {
    store_tiles<NF>(a.dz_in + tok * a.D + h, valid, h, g, x1);
    return;
  } else {
    T* orow = a.z_out + tok * a.D;
    store_tiles<NF>(orow + h, valid, h, g, x1);
    float base = 0.f;
#pragma unroll
    for (int ks = 0; ks < HT; ++ks) {   // x0 half: out = x0 * m (x0f is already masked), fragment layout
      const int f = 32 * ks + 8 * g;
      const u16x8 raw = __builtin_bit_cast(u16x8, x0f[ks]);
      if (valid && f < h) {
        *reinterpret_cast<u16x8*>(orow + f) = raw;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const float d = bf16_to_f32(raw[jj]) - a.mu;
          base += -(d * d) * a.inv2var - a.cst;
        }
      }
    }
    if (a.base) {
#pragma unroll
      for (int v = 0; v < NF; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * v + 4 * g + r < h) {
            const float vv = bf16_to_f32(f32_to_bf16(x1[v][r]));   // what the stored z_K holds
            const float d = vv - a.mu;
            base += -(d * d) * a.inv2var - a.cst;
          }
    } else {
      base = 0.f;
    }
    const float ld = (ld4[0] + ld4[1]) + (ld4[2] + ld4[3]);
    float tv = (ld + base) * m;
    tv += __shfl_xor(tv, 16, 64);
    tv += __shfl_xor(tv, 32, 64);
    if (valid && g == 0 && a.tok_val) a.tok_val[tok] = tv;
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
  const float* dlogp;   // (Bn) or null
  const T* dSdT;        // (K, 2, N, h)
  T* dz_in;             // (N, D): x0 half written here
  T* hbuf;              // SAVE: (K, 2, HC, N, 32) hidden activations: chunk-major, 64 contiguous bytes per (chunk, token),
                        //   the 32 units of a chunk in accumulator-pair order (position 8 g + j = unit 4 g + j, j < 4, or
                        //   16 + 4 g + j - 4): one 16-byte store per lane and body, 1 KiB contiguous per wave instruction
  T* gbuf;              // SAVE: same layout, d pre-activation
  int64_t N;
  int S, mask_div, D, K, h, H, HC;
  int act, base;
  float mu, inv2var;
};

// hidden rows of a token no wave computes (dead tile): zeros, so that the weight-gradient products stay finite
__device__ __forceinline__ void zero_hidden_rows(T* hbuf, T* gbuf, int64_t N, int K, int HC, int64_t tok, int g) {
  const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = 0; q < 2 * K * HC; ++q) {            // (kn, chunk) sections of (N, 32): the lane's 16-byte piece g
    const int64_t at = ((int64_t)q * N + tok) * 32 + 8 * g;
    *reinterpret_cast<u16x8*>(hbuf + at) = zero;
    *reinterpret_cast<u16x8*>(gbuf + at) = zero;
  }
}

// SAVE: also write the hidden activations and d pre of every (step, net, token) for k_big_dw2 (the weight-gradient
// kernel then contracts them directly instead of recomputing both from x0 and d out).
template <int HT, int ACT, bool SAVE>
__global__ void __launch_bounds__(512) k_big_nets(NetsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NF = 2 * HT;
  constexpr int ITEM = NF * 1024;
  constexpr int G = RingCfg<HT>::G3;
  typedef Ring<ITEM, 3 * G, 4> RingT;
  constexpr int R = RingT::R;
  const int lane = threadIdx.x & 63;
  const int tk = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = a.h, HC = a.HC;
  const int64_t tok = ((int64_t)blockIdx.x * 8 + wave) * 16 + tk;
  const bool valid = tok < a.N;
  float m = 0.f, dl = 0.f;
  if (valid) {
    m = 1.f;
    if (a.mask) {
      const int64_t bn = tok / a.S;
      m = (float)a.mask[(bn / a.mask_div) * a.S + (tok - bn * a.S)];
    }
    if (a.dlogp) dl = a.dlogp[tok / a.S] * m;
  }
  const bool live = __any(m != 0.f) != 0;
  if (!block_live(a.mask, a.N, a.S, a.mask_div, lane)) {
    if (valid) {
      const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
      T* drow = a.dz_in + tok * a.D;
      for (int f = 8 * g; f < h; f += 32) *reinterpret_cast<u16x8*>(drow + f) = zero;
      if constexpr (SAVE) zero_hidden_rows(a.hbuf, a.gbuf, a.N, a.K, HC, tok, g);
    }
    return;
  }
  if constexpr (SAVE) {
    if (!live && valid) zero_hidden_rows(a.hbuf, a.gbuf, a.N, a.K, HC, tok, g);
  }
  const bool on = valid && m != 0.f;
  bf16x8 x0f[HT];
  load_bfrags<HT>(a.z_in + tok * a.D, on, h, g, x0f);
  f32x4 dx0[NF];
  load_tiles<NF>(a.dz_out ? a.dz_out + tok * a.D : a.z_in, valid && a.dz_out != nullptr, h, g, dx0);
  if (a.base && a.dlogp && a.z_out) {
    f32x4 zk[NF];
    const float c2 = 2.0f * a.inv2var;
    load_tiles<NF>(a.z_out + tok * a.D, valid, h, g, zk);
#pragma unroll
    for (int v = 0; v < NF; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) dx0[v][r] += (16 * v + 4 * g + r < h) ? -dl * (zk[v][r] - a.mu) * c2 : 0.f;
  }
#pragma unroll
  for (int v = 0; v < NF; ++v)
#pragma unroll
    for (int r = 0; r < 4; ++r) dx0[v][r] *= m;

  RingT ring;
  float* bias_lds = reinterpret_cast<float*>(smem + R * ITEM);
  const int NJ = a.K * 2;
  const int Tn = NJ * HC;
  auto kn_of = [&](int j) { return (a.K - 1 - (j >> 1)) * 2 + (j & 1); };
  auto issue_bias = [&](int j) {
    const char* src = reinterpret_cast<const char*>(a.bias + (int64_t)kn_of(j) * kBiasBlockFloats);
    __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(src + (wave & 3) * 1024 + lane * 16),
                                     (lds_void_t*)((char*)bias_lds + (j & 3) * 4096 + (wave & 3) * 1024), 16, 0, 0);
  };
  // d out of net-index j as B fragments (natural k order over the out features)
  bf16x8 dob[HT];
  auto load_dout = [&](int j) { load_bfrags<HT>(a.dSdT + ((int64_t)kn_of(j) * a.N + tok) * h, on, h, g, dob); };
  load_dout(0);
  int nbias = 0;
  for (; nbias < kBiasSlots && nbias < NJ; ++nbias) issue_bias(nbias);
  ring.init(a.stream, smem, 3 * Tn + 4, wave, lane);

  // stage A of a task: pre = b1 + W1 x0, dh = W2^T d out   (two 16-row tiles each)
  auto stage_a = [&](const char* itW1, const char* itW2T, const float* b1c, f32x4 (&pre)[2], f32x4 (&dh)[2]) {
    const bf16x8* w1 = reinterpret_cast<const bf16x8*>(itW1) + lane;
    const bf16x8* w2 = reinterpret_cast<const bf16x8*>(itW2T) + lane;
    pre[0] = lds4(b1c + 4 * g); pre[1] = lds4(b1c + 16 + 4 * g);
    dh[0] = f32x4{0.f, 0.f, 0.f, 0.f}; dh[1] = dh[0];
#pragma unroll
    for (int ks = 0; ks < HT; ++ks) {
      pre[0] = mma16(w1[ks * 64], x0f[ks], pre[0]);
      pre[1] = mma16(w1[(HT + ks) * 64], x0f[ks], pre[1]);
      dh[0] = mma16(w2[ks * 64], dob[ks], dh[0]);
      dh[1] = mma16(w2[(HT + ks) * 64], dob[ks], dh[1]);
    }
  };
  // stage B: d pre = dh * act'(pre) -> B fragment
  //          (SAVE: task (j1, c1)'s hidden activations and d pre go to hbuf / gbuf, one 16-byte store each)
  auto stage_b = [&](const f32x4 (&pre)[2], const f32x4 (&dh)[2], int j1, int c1) -> bf16x8 {
    f32x4 dlo, dhi;
    if constexpr (SAVE) {
      f32x4 hlo = pre[0], hhi = pre[1];
      act_pair_fd<ACT>(a.act, hlo, hhi, dlo, dhi);
      dlo *= dh[0]; dhi *= dh[1];
      if (valid && j1 < NJ) {
        const int64_t at = (((int64_t)kn_of(j1) * HC + c1) * a.N + tok) * 32 + 8 * g;
        *reinterpret_cast<bf16x8*>(a.hbuf + at) = pair_to_b(hlo, hhi);
        *reinterpret_cast<bf16x8*>(a.gbuf + at) = pair_to_b(dlo, dhi);
      }
    } else {
      act_pair_d<ACT>(a.act, pre[0], pre[1], dlo, dhi);
      dlo *= dh[0]; dhi *= dh[1];
    }
    return pair_to_b(dlo, dhi);
  };
  ring.enter_prologue();
  f32x4 pre1[2], dh1[2];
  bf16x8 gb;   // d pre of task i as B fragment
  {
    f32x4 pre0[2], dh0[2];
    stage_a(ring.item(0), ring.item(1), bias_lds, pre0, dh0);
    gb = stage_b(pre0, dh0, 0, 0);
    stage_a(ring.item(2), ring.item(3), bias_lds + 32, pre1, dh1);
  }
  if (!live) {
    int i = 0;
    for (int j = 0; j < NJ; ++j)
      for (int c = 0; c < HC; ++c, ++i) {
        if (i % G == 0) {
          ring.enter_group(i / G);
          while (nbias < j + kBiasSlots && nbias < NJ) issue_bias(nbias++);
        }
      }
  } else {
    int i = 0;
#pragma unroll 1
    for (int j = 0; j < NJ; ++j) {
#pragma unroll 1
      for (int c = 0; c < HC; ++c, ++i) {
        if (i % G == 0) {
          ring.enter_group(i / G);
          while (nbias < j + kBiasSlots && nbias < NJ) issue_bias(nbias++);
        }
        const char* itT = ring.item(4 + 3 * i);
        const char* itA = ring.item(5 + 3 * i);
        const char* itB = ring.item(6 + 3 * i);
        int c2 = c + 2, j2 = j;
        if (c2 >= HC) { c2 -= HC; ++j2; }
        int c1 = c + 1, j1 = j;
        if (c1 >= HC) { c1 -= HC; ++j1; }
        // the first task of the next net enters stage A here: switch the d out fragments
        if (c2 == 0 && j2 < NJ) load_dout(j2);
        // C(i): dx0 += W1T frags . d pre(i);  A(i+2);  B(i+1) -> d pre(i+1)   (tasks past the end: zero padding)
        const bf16x8* wt = reinterpret_cast<const bf16x8*>(itT) + lane;
        f32x4 pre2[2], dh2[2];
        if constexpr (HT <= 4) {
          // Phases pinned in program order: fragment reads a whole phase ahead of the MFMAs that consume them (left to
          // itself the scheduler waits for every read right before its MFMA: 24 exposed LDS latencies per task).
          //   reads C(i) + A(i+2) first layer | B(i+1) (VALU) | MFMAs C | reads A(i+2) second layer | MFMAs A
          const bf16x8* w1 = reinterpret_cast<const bf16x8*>(itA) + lane;
          const bf16x8* w2 = reinterpret_cast<const bf16x8*>(itB) + lane;
          const float* b1c = bias_lds + (j2 & 3) * kBiasBlockFloats + 32 * c2;
          bf16x8 ft[NF], f1[2 * HT], f2[2 * HT];
#pragma unroll
          for (int v = 0; v < NF; ++v) ft[v] = wt[v * 64];
#pragma unroll
          for (int u = 0; u < 2 * HT; ++u) f1[u] = w1[u * 64];
          pre2[0] = lds4(b1c + 4 * g); pre2[1] = lds4(b1c + 16 + 4 * g);
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 gbn = stage_b(pre1, dh1, j1, c1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int v = 0; v < NF; ++v) dx0[v] = mma16(ft[v], gb, dx0[v]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 2 * HT; ++u) f2[u] = w2[u * 64];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < HT; ++ks) {
            pre2[0] = mma16(f1[ks], x0f[ks], pre2[0]);
            pre2[1] = mma16(f1[HT + ks], x0f[ks], pre2[1]);
          }
          dh2[0] = f32x4{0.f, 0.f, 0.f, 0.f}; dh2[1] = dh2[0];
#pragma unroll
          for (int ks = 0; ks < HT; ++ks) {
            dh2[0] = mma16(f2[ks], dob[ks], dh2[0]);
            dh2[1] = mma16(f2[HT + ks], dob[ks], dh2[1]);
          }
          gb = gbn;
        } else {
          // B(i+1) first: with SAVE its stores are issued at the start of the body and have the whole task to complete
          // before the next group boundary waits for every outstanding vector memory operation
          const bf16x8 gbn = stage_b(pre1, dh1, j1, c1);
#pragma unroll
          for (int v = 0; v < NF; ++v) dx0[v] = mma16(wt[v * 64], gb, dx0[v]);
          stage_a(itA, itB, bias_lds + (j2 & 3) * kBiasBlockFloats + 32 * c2, pre2, dh2);
          gb = gbn;
        }
        pre1[0] = pre2[0]; pre1[1] = pre2[1]; dh1[0] = dh2[0]; dh1[1] = dh2[1];
      }
    }
  }
  store_tiles<NF>(a.dz_in + tok * a.D, valid, h, g, dx0);
}

// ---- weight gradients (weight-stationary, recompute) -----------------------------------------------------------
// v_mfma_f32_16x16x32_bf16 again: a wave owns 16 hidden units of one (step, net) -- its rows of W1 and columns of W2
// stay in registers as B fragments -- and a block of 8 waves (128 hidden units) walks a slab of 32-token tiles staged
// in LDS.  For every tile it RECOMPUTES pre = x0 W1^T + b1 and dh = d out W2 (tokens x hidden), applies the
// activation and its derivative, and accumulates dW1^T = x0^T d pre and dW2 = d out^T hid with the tokens as the
// contraction (transposing LDS reads give the [feature][token] fragments).  ~250 registers at h = 128 (64 of them the two
// double-buffered fragment sets): two waves per SIMD, one block per CU.
struct DwBigArgs {
  const T* z_in;        // (N, D): x0 = z_in[:, :h]
  const T* dSdT;        // (K, 2, N, h)
  const int64_t* mask;
  const int64_t* ptrs;  // device table [K][8] of the fp32 parameters (mivae_nice_pack's order)
  const float* bias;
  float* part;          // [kn][hw][slab][2][2 HT][64][4]     hw = 16-wide hidden group
  float* partb1;        // [kn][hw][slab][16]
  float* partb2;        // [kn][slab][hp]
  int64_t N;
  int S, mask_div, D, K, h, H, HW, act;   // HW = Hp / 16
  int nslab, slab_tiles;
};

// LDS tile image: row-major [32 tokens][hp] bf16, the 16-byte chunks of a row XOR-swizzled by the row so that the
// transposing reads (ds_read_b64_tr_b16: 4 token rows x 16 features per 16-lane group) are conflict free.
// Both read patterns are conflict free with it: a transposing read's 32-lane group covers rows 4 g + q (two g) x chunk
// pairs, told apart by bits 3:1; a ds_read_b128's 16-lane group is lanes {0-3, 12-15} of one 16-lane quarter plus lanes
// {4-11} of its neighbour (chunk index differing in bit 0), i.e. rows r with chunk bit 0 flipped for 4 <= r <= 11:
// (r1, r0, r2, r2 ^ r3) is still a bijection of the 16 rows onto the 16 chunks.
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | (((row >> 2) & 1) << 1); }

typedef __bf16 bf16x4v __attribute__((__vector_size__(4 * sizeof(__bf16))));
#define BIG_LDS __attribute__((address_space(3)))

// byte offsets (within a tile) of the two transposing reads of A fragment v: A[feature 16 v + (lane & 15)][k-slot]
// with k = tokens in pair_to_b()'s order: slot (g, j) = token (j < 4 ? 4 g + j : 16 + 4 g + j - 4)
template <int HPB>
__device__ __forceinline__ void tr_offsets(int v, int lane, int& o0, int& o1) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int row0 = 4 * g + q, row1 = 16 + 4 * g + q;
  const int chunk = 2 * v + (p >> 1), sub = 8 * (p & 1);
  o0 = row0 * HPB + ((chunk ^ swz(row0)) << 4) + sub;
  o1 = row1 * HPB + ((chunk ^ swz(row1)) << 4) + sub;
}
__device__ __forceinline__ bf16x8 tr_read(const char* tile, int o0, int o1) {
  const bf16x4v v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((BIG_LDS bf16x4v*)(tile + o0));
  const bf16x4v v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((BIG_LDS bf16x4v*)(tile + o1));
  const auto v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
// 16 bytes per lane global -> LDS (wave-uniform destination base, lanes land 16 bytes apart), issued behind the
// compiler's back: after a __builtin_amdgcn_global_load_lds it cannot prove disjoint from a later LDS read, hipcc's
// wait-count pass puts s_waitcnt vmcnt(0) in front of that read, which would expose the whole global latency in
// every loop body here.  The kernel orders the DMA itself (finish_tile's wait + the block barrier).
__device__ __forceinline__ void dma16(const void* src, void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src) : "memory", "m0");
}

// PART: 0 = both weight gradients in one pass (h <= 128); 1 = d W1, d b1; 2 = d W2, d b2 (h > 128: the accumulators of
// one product fill the registers two waves per SIMD leave, so the products take a pass each).
template <int HT, int ACT, int PART>
__global__ void __launch_bounds__(512) k_big_dw(DwBigArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NF = 2 * HT;                // 16-feature tiles of the h features
  constexpr int HPB = HT * 64;              // bytes per tile row
  constexpr int CPR = HPB / 16;             // 16-byte chunks per tile row
  constexpr int TILE = 32 * HPB;            // bytes per operand tile
  constexpr int NTH = 512;
  constexpr int CPT = TILE / 16 / NTH;      // 16-byte chunks per thread and operand tile
  constexpr bool DO1 = PART != 2, DO2 = PART != 1;
  static_assert(CPT >= 1, "tile too small for the block");
  const int lane = threadIdx.x & 63;
  const int r16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware block order: consecutive block ids go round-robin over the 8 XCDs (each with its own L2), and the Z
  // blocks that walk the SAME token tiles of the same (step, net) for different hidden units should share one L2:
  // within an XCD (ids id % 8 equal) consecutive blocks are the Z siblings of one (slab, kn) group.
  const int Z = (a.HW + 7) / 8;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int zb = jx % Z, grp = (jx / Z) * 8 + xcd;
  if (grp >= a.nslab * 2 * a.K) return;
  const int slab = grp % a.nslab, kn = grp / a.nslab, hw = zb * 8 + wave;
  const int h = a.h;
  const bool wave_on = hw < a.HW;
  const int k = kn >> 1, net = kn & 1;
  const int hid = 16 * hw + r16;            // this lane's hidden unit (column of the products)
  bf16x8 w1f[HT], w2f[DO1 ? HT : 1];
  float b1v = 0.f;
  if (wave_on) {
    // stationary operands as B fragments (col = hidden unit, k natural): W1[hid][32ks + 8g + j], W2[32ks + 8g + j][hid],
    // rounded to bf16 exactly like the packed streams of the other kernels
    const float* W1 = reinterpret_cast<const float*>(a.ptrs[k * 8 + net * 4 + 0]);
    const float* W2 = reinterpret_cast<const float*>(a.ptrs[k * 8 + net * 4 + 2]);
#pragma unroll
    for (int ks = 0; ks < HT; ++ks) {
      u16x8 r1 = {0, 0, 0, 0, 0, 0, 0, 0}, r2 = r1;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int f = 32 * ks + 8 * g + jj;
        if (hid < a.H && f < h) {
          r1[jj] = f32_to_bf16(W1[(int64_t)hid * h + f]);
          if (DO1) r2[jj] = f32_to_bf16(W2[(int64_t)f * a.H + hid]);
        }
      }
      w1f[ks] = __builtin_bit_cast(bf16x8, r1);
      if (DO1) w2f[ks] = __builtin_bit_cast(bf16x8, r2);
    }
    b1v = a.bias[(int64_t)kn * kBiasBlockFloats + hid];   // zero beyond H (padded block)
  }
  f32x4 dW1[DO1 ? NF : 1], dW2[DO2 ? NF : 1];
#pragma unroll
  for (int v = 0; v < (DO1 ? NF : 1); ++v) dW1[v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int v = 0; v < (DO2 ? NF : 1); ++v) dW2[v] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db1 = 0.f;
  float db2[DO2 ? NF : 1];
#pragma unroll
  for (int v = 0; v < (DO2 ? NF : 1); ++v) db2[v] = 0.f;
  const bool want_b2 = DO2 && zb == 0 && wave == 0;

  // ---- staging: LDS-DMA, 16 bytes per lane, NBUF tile buffers: tiles n-1, n, n+1 are in use by body n and PD = NBUF - 3
  // tiles are in flight (tile n+2 lands by the end of body n).  Chunk id = threadIdx.x + NTH u of a tile lands at LDS byte 16 id (lane linear); the XOR
  // swizzle is applied on the SOURCE side: position (row, pc) receives the row's chunk pc ^ swz(row).  Every row
  // below N is read as it is (the chain kernel writes zero d s / d t rows for padding tokens); rows beyond N or beyond
  // the slab are fetched from a clamped valid address and their d out chunks overwritten with zeros once the DMA has
  // landed (x0 stays: finite values times d out = 0).  No compiler-visible global load remains in the loop, so the
  // only vmcnt waits are the counted ones below.
  constexpr int NBUF = HT <= 4 ? BIG_DW_NBUF : 4;
  constexpr int PD = NBUF - 3;
  const int64_t tile0 = (int64_t)slab * a.slab_tiles;
  int ntiles = (int)((a.N + 31) / 32 - tile0);
  if (ntiles > a.slab_tiles) ntiles = a.slab_tiles;
  // lane offsets of the chunks against a per-tile scalar base (32-bit: a tile spans < 2^31 bytes)
  int srow[CPT], scol[CPT], vox[CPT], vog[CPT];
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const int id = threadIdx.x + NTH * u;
    const int row = id / CPR;
    int sc = (id % CPR) ^ swz(row);
    if (sc * 8 >= h) sc = h / 8 - 1;      // padded feature chunks: any finite in-row data (their weights are zero)
    srow[u] = row; scol[u] = sc * 16;
    vox[u] = row * a.D * 2 + sc * 16;
    vog[u] = row * h * 2 + sc * 16;
  }
  const char* gx = reinterpret_cast<const char*>(a.z_in);
  const char* gg = reinterpret_cast<const char*>(a.dSdT + (int64_t)kn * a.N * h);
  auto rows_of = [&](int n) -> int {      // rows of tile n that count (0 for tiles beyond the slab)
    if (n >= ntiles) return 0;
    const int64_t left = a.N - (tile0 + n) * 32;
    return left < 32 ? (int)left : 32;
  };
  auto issue_tile = [&](int n, int buf) {   // tile tile0 + n -> buffer buf
    const int nn = n < ntiles ? n : ntiles - 1;
    const int64_t t0 = (tile0 + nn) * 32;
    const int64_t left = a.N - t0;
    const int lastrow = left < 32 ? (int)left - 1 : 31;
    char* bx = smem + buf * 2 * TILE + wave * 1024;
    const char* tx = gx + t0 * a.D * 2;                       // wave-uniform bases
    const char* tg = gg + t0 * h * 2;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const bool inrow = srow[u] <= lastrow;
      const unsigned ox = inrow ? vox[u] : lastrow * a.D * 2 + scol[u];
      const unsigned og = inrow ? vog[u] : lastrow * h * 2 + scol[u];
      dma16(tx + ox, bx + u * NTH * 16);
      dma16(tg + og, bx + TILE + u * NTH * 16);
    }
  };
  auto finish_tile = [&](int n, int buf, auto pending) {  // tile n has landed once at most `pending` DMAs are in flight
    __builtin_amdgcn_sched_barrier(0);   // the body's arithmetic stays above the wait
    wait_vm<decltype(pending)::value>();
    const int nrows = rows_of(n);
    if (nrows < 32) {                    // zero the d out chunks of rows that do not count (wave-uniform branch)
      char* bg = smem + buf * 2 * TILE + TILE + threadIdx.x * 16;
      const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < CPT; ++u)
        if (srow[u] >= nrows) *reinterpret_cast<u16x8*>(bg + u * NTH * 16) = zero;
    }
  };
  // per-lane fragment offsets inside a tile
  // (the swizzle touches only the low four chunk bits: chunks 16 and up repeat the pattern 256 bytes further on)
  int ro[2][4];        // row reads: A[token 16 tt + r16][features 32 ks + 8 g ..]
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int row = 16 * tt + r16;
      ro[tt][ks] = row * HPB + (((4 * ks + g) ^ swz(row)) << 4);
    }
  int to0[8], to1[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) tr_offsets<HPB>(v, lane, to0[v], to1[v]);

  // Software pipeline over the token tiles (4 LDS buffers): body n runs A(n+1) (pre, dh of tile n+1: row reads),
  // C(n-1) (the weight-gradient products of tile n-1: transposing reads) and B(n) (activation, VALU) while tile n+2
  // travels global -> LDS.  Tiles outside the slab are zeros, so bodies 0 .. ntiles need no peeling.
  {   // tile -1 (buffer NBUF - 1): zeros
    const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 2 * CPT; ++u)
      *reinterpret_cast<u16x8*>(smem + (NBUF - 1) * 2 * TILE + (threadIdx.x + NTH * u) * 16) = zero;
  }
  // tile t lives in buffer t % NBUF; prologue: tiles 0 .. PD issued, tiles 0 and 1 landed
#pragma unroll
  for (int t = 0; t < PD + 1; ++t) issue_tile(t, t);
  finish_tile(0, 0, std::integral_constant<int, PD * 2 * CPT>{});
  finish_tile(1, 1, std::integral_constant<int, (PD - 1) * 2 * CPT>{});
  __syncthreads();
  auto stage_a = [&](const char* bx, f32x4 (&pre)[2], f32x4 (&dh)[2]) {
    const char* bg = bx + TILE;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) { pre[tt] = f32x4{b1v, b1v, b1v, b1v}; dh[tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < HT; ++ks)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const bf16x8 ax = *reinterpret_cast<const bf16x8*>(bx + ro[tt][ks & 3] + (ks >> 2) * 256);
        pre[tt] = mma16(ax, w1f[ks], pre[tt]);
        if (DO1) {
          const bf16x8 ag = *reinterpret_cast<const bf16x8*>(bg + ro[tt][ks & 3] + (ks >> 2) * 256);
          dh[tt] = mma16(ag, w2f[ks], dh[tt]);
        }
      }
  };
  f32x4 pre_c[2], dh_c[2];
  if (wave_on) stage_a(smem, pre_c, dh_c);
  const u16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 hb = __builtin_bit_cast(bf16x8, z8), gb = hb;   // fragments of tile n-1
  int bm = NBUF - 1, bp = 1, bl = 2, bi = (1 + PD) % NBUF;
  auto run = [&](auto with_b2) {   // (variants live in their own copies of the loop: no branch per tile)
#pragma unroll 1
  for (int n = 0; n <= ntiles; ++n) {
    // buffers: bm = tile n-1, bp = tile n+1, bl = tile n+2 (to land by the end of the body), bi = tile n+1+PD (issued
    // now, into the buffer tile n-2 has left)
    issue_tile(n + 1 + PD, bi);
    if (wave_on) {
      // Phases pinned in program order (sched_barrier): LDS reads are issued a whole phase ahead of the MFMAs that
      // consume them -- left alone the scheduler keeps them one MFMA ahead and every other MFMA waits out the LDS
      // latency.  reads of step 0 | reads 1 | [activation of tile n (VALU) before the first A chunk] MFMAs 0 | reads 2 | ...
      const char* bxa = smem + bp * 2 * TILE;
      const char* bga = bxa + TILE;
      const char* bxc = smem + bm * 2 * TILE;
      const char* bgc = bxc + TILE;
      // steps: NKA chunks of A(n+1) (two k-steps each), then NCV chunks of C(n-1) (CV feature tiles each); the reads of
      // step st+1 are issued before the MFMAs of step st, into the other half of fb
      constexpr int KA = HT <= 4 ? 2 : 1;                       // k-steps per A chunk
      constexpr int CV = (DO1 && DO2) ? 4 : (HT <= 4 ? 8 : 4);  // feature tiles per C chunk
      constexpr int NCV = NF / CV, NKA = HT / KA, NST = NKA + NCV;
      constexpr int QA = 2 * KA, FBN = HT <= 4 ? 8 : 4;         // fragments per chunk: x | g halves of fb[sl]
      static_assert((DO1 ? 2 : 1) * QA <= FBN && ((DO1 && DO2) ? 2 : 1) * CV <= FBN, "fragment buffer too small");
      bf16x8 fb[2][FBN];
      auto reads = [&](int st, int sl) {
        if (st < NKA) {
#pragma unroll
          for (int q = 0; q < QA; ++q) {
            const int ks = KA * st + (q >> 1), off = ro[q & 1][ks & 3] + (ks >> 2) * 256;
            fb[sl][q] = *reinterpret_cast<const bf16x8*>(bxa + off);
            if (DO1) fb[sl][QA + q] = *reinterpret_cast<const bf16x8*>(bga + off);
          }
        } else {
#pragma unroll
          for (int i = 0; i < CV; ++i) {
            const int v = (st - NKA) * CV + i;
            if (DO1) fb[sl][i] = tr_read(bxc + (v >> 3) * 256, to0[v & 7], to1[v & 7]);
            if (DO2) fb[sl][(DO1 ? CV : 0) + i] = tr_read(bgc + (v >> 3) * 256, to0[v & 7], to1[v & 7]);
          }
        }
      };
      auto mfmas = [&](int st, int sl) {
        if (st < NKA) {
#pragma unroll
          for (int q = 0; q < QA; ++q) {
            const int ks = KA * st + (q >> 1);
            pre_c[q & 1] = mma16(fb[sl][q], w1f[ks], pre_c[q & 1]);
            if (DO1) dh_c[q & 1] = mma16(fb[sl][QA + q], w2f[ks], dh_c[q & 1]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < CV; ++i) {
            const int v = (st - NKA) * CV + i;
            if (DO1) dW1[v] = mma16(fb[sl][i], gb, dW1[v]);
            if (DO2) {
              const bf16x8 gf = fb[sl][(DO1 ? CV : 0) + i];
              dW2[v] = mma16(gf, hb, dW2[v]);
              if constexpr (decltype(with_b2)::value) {   // db2 = sum over tokens of d out (one wave per (slab, step, net))
                float sm = 0.f;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) sm += (float)gf[jj];
                db2[v] += sm;
              }
            }
          }
        }
      };
      auto step_of = [](int i) { return i; };
      bf16x8 hbn = hb, gbn = gb;
      reads(step_of(0), 0);
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        // (the reads of step 1 go out after the activation: both fragment buffers live across it cost registers)
        if (i + 1 < NST && i != 0) reads(step_of(i + 1), (i + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (step_of(i) == 0) {   // the activation of tile n, right before the first A chunk overwrites pre_c / dh_c
          if (DO1) {
            f32x4 dlo, dhi;
            act_pair_fd<ACT>(a.act, pre_c[0], pre_c[1], dlo, dhi);   // pre_c := hidden activations of tile n
#pragma unroll
            for (int r = 0; r < 4; ++r) { dlo[r] *= dh_c[0][r]; dhi[r] *= dh_c[1][r]; db1 += dlo[r] + dhi[r]; }
            gbn = pair_to_b(dlo, dhi);
          } else {
            act_pair<ACT>(a.act, pre_c[0], pre_c[1]);
          }
          if (DO2) hbn = pair_to_b(pre_c[0], pre_c[1]);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) { pre_c[tt] = f32x4{b1v, b1v, b1v, b1v}; dh_c[tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
          __builtin_amdgcn_sched_barrier(0);
          if (i + 1 < NST) { reads(step_of(i + 1), (i + 1) & 1); __builtin_amdgcn_sched_barrier(0); }
        }
        mfmas(step_of(i), i & 1);
      }
      hb = hbn; gb = gbn;
    }
    finish_tile(n + 2, bl, std::integral_constant<int, (PD - 1) * 2 * CPT>{});
    __syncthreads();
    bm = bm + 1 == NBUF ? 0 : bm + 1; bp = bp + 1 == NBUF ? 0 : bp + 1;
    bl = bl + 1 == NBUF ? 0 : bl + 1; bi = bi + 1 == NBUF ? 0 : bi + 1;
  }
  };
  if (want_b2) run(std::true_type{}); else run(std::false_type{});
  if (!wave_on) return;
  // dump: register r of lane of dW1[v] = d W1[hidden 16 hw + (lane & 15)][in 16 v + 4 (lane >> 4) + r]
  const int64_t item = ((int64_t)kn * a.HW + hw) * a.nslab + slab;
  float* out = a.part + item * (2 * NF * 256) + lane * 4;
#pragma unroll
  for (int v = 0; v < NF; ++v) {
    if (DO1) *reinterpret_cast<f32x4*>(out + v * 256) = dW1[v];
    if (DO2) *reinterpret_cast<f32x4*>(out + (NF + v) * 256) = dW2[v];
  }
  if (DO1) {
    db1 += __shfl_xor(db1, 16, 64);
    db1 += __shfl_xor(db1, 32, 64);
    if (g == 0) a.partb1[item * 16 + r16] = db1;
  }
  if (want_b2) {
#pragma unroll
    for (int v = 0; v < NF; ++v) {
      float s = db2[v] + __shfl_xor(db2[v], 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (g == 0) a.partb2[((int64_t)kn * a.nslab + slab) * (16 * NF) + 16 * v + r16] = s;
    }
  }
}

// ---- weight gradients from the saved hidden activations -----------------------------------------------------------
// k_big_nets<SAVE> leaves hid = act(pre) and d pre of every (step, net, token) in HBM; the weight gradients are then two
// plain contractions over the tokens, dW1^T = x0^T d pre and dW2 = d out^T hid, with no recomputation and no activation
// arithmetic (the recompute kernel above spends half its MFMAs and all of its VALU time on them).  Same decomposition as
// k_big_dw: a block = 8 waves x 16 hidden units of one (step, net), walking a slab of 32-token tiles; per tile the x0 and
// d out tiles ([32][hp]) and the block's 128-unit slices of hid and d pre ([32][128]) are LDS-DMA staged (source-side
// XOR swizzle) and every fragment is a transposing read: A = x0^T / d out^T feature tiles, B = hid^T / d pre^T of the wave's
// 2 x 16 units (a wave owns two unit groups and half of the feature tiles: every A fragment feeds two MFMAs).  Accumulators and dump layout are k_big_dw's, so k_big_dw_reduce serves both.
struct Dw2Args {
  const T* z_in;        // (N, D)
  const T* dSdT;        // (K, 2, N, h)
  const T* hbuf;        // (K, 2, HC, N, 32): chunk-major, units of a chunk in accumulator-pair order (NetsArgs)
  const T* gbuf;        // same layout
  float* part;
  float* partb1;
  float* partb2;
  int64_t N;
  int D, K, h, HW, Hp;
  int nslab, slab_tiles;
};

template <int HT>
__global__ void __launch_bounds__(512) k_big_dw2(Dw2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NF = 2 * HT;
  constexpr int HPB = HT * 64;
  constexpr int CPR = HPB / 16;
  constexpr int TILE = 32 * HPB;
  constexpr int HTILE = 32 * 256;           // [32 tokens][128 hidden] bf16
  constexpr int NTH = 512;
  constexpr int CPT = TILE / 16 / NTH;
  constexpr int STAGE = 2 * TILE + 2 * HTILE;
  constexpr int NBUF = HT <= 4 ? 4 : 3;
  constexpr int PER = 2 * CPT + 2;          // DMAs per thread and tile
  static_assert(CPT >= 1 && HTILE / 16 == NTH, "tile / block mismatch");
  const int lane = threadIdx.x & 63;
  const int r16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Z = (a.HW + 7) / 8;             // same XCD-aware block order as k_big_dw
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int zb = jx % Z, grp = (jx / Z) * 8 + xcd;
  if (grp >= a.nslab * 2 * a.K) return;
  // a wave owns TWO 16-unit groups (2 hp, 2 hp + 1 of the block's eight) and HALF of the feature tiles (fh): every
  // A fragment it reads feeds two MFMAs, so the block reads each x0 / d out tile four times instead of eight
  const int hp = wave & 3, fh = wave >> 2;
  const int slab = grp % a.nslab, kn = grp / a.nslab, hw0 = zb * 8 + 2 * hp;
  const int h = a.h;
  const bool wave_on = hw0 < a.HW;                  // HW is even (Hp is a multiple of 32): the pair is on or off
  const bool want_b1 = fh == 0;
  const bool want_b2 = zb == 0 && hp == 0;
  constexpr int NV = NF / 2;                        // feature tiles of this wave: v = fh NV + i
  f32x4 dW1[NV][2], dW2[NV][2];
  float db1[2] = {0.f, 0.f}, db2[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    dW1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; dW1[i][1] = dW1[i][0]; dW2[i][0] = dW1[i][0]; dW2[i][1] = dW1[i][0];
    db2[i] = 0.f;
  }

  const int64_t tile0 = (int64_t)slab * a.slab_tiles;
  int ntiles = (int)((a.N + 31) / 32 - tile0);
  if (ntiles > a.slab_tiles) ntiles = a.slab_tiles;
  int srow[CPT], scol[CPT], vox[CPT], vog[CPT];
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const int id = threadIdx.x + NTH * u;
    const int row = id / CPR;
    int sc = (id % CPR) ^ swz(row);
    if (sc * 8 >= h) sc = h / 8 - 1;
    srow[u] = row; scol[u] = sc * 16;
    vox[u] = row * a.D * 2 + sc * 16;
    vog[u] = row * h * 2 + sc * 16;
  }
  // hidden tiles: one chunk per thread; chunks past Hp (units no wave of this block owns) re-read a valid chunk
  const int hrow = threadIdx.x >> 4;
  int hsc = (threadIdx.x & 15) ^ swz(hrow);
  if (128 * zb + 8 * hsc >= a.Hp) hsc = 0;
  const int hchunk = (128 * zb + 8 * hsc) >> 5;                 // 32-unit chunk of this thread's 8 units
  const int hcol = ((128 * zb + 8 * hsc) & 31) * 2;             // byte offset inside the chunk's 64-byte token row
  const char* gx = reinterpret_cast<const char*>(a.z_in);
  const char* gg = reinterpret_cast<const char*>(a.dSdT + (int64_t)kn * a.N * h);
  const char* gh = reinterpret_cast<const char*>(a.hbuf + (int64_t)kn * a.N * a.Hp);
  const char* gd = reinterpret_cast<const char*>(a.gbuf + (int64_t)kn * a.N * a.Hp);
  auto issue_tile = [&](int n, int buf) {
    const int nn = n < ntiles ? n : ntiles - 1;
    const int64_t t0 = (tile0 + nn) * 32;
    const int64_t left = a.N - t0;
    const int lastrow = left < 32 ? (int)left - 1 : 31;
    char* bx = smem + buf * STAGE + wave * 1024;
    const char* tx = gx + t0 * a.D * 2;
    const char* tg = gg + t0 * h * 2;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const bool inrow = srow[u] <= lastrow;
      const unsigned ox = inrow ? vox[u] : lastrow * a.D * 2 + scol[u];
      const unsigned og = inrow ? vog[u] : lastrow * h * 2 + scol[u];
      dma16(tx + ox, bx + u * NTH * 16);
      dma16(tg + og, bx + TILE + u * NTH * 16);
    }
    const int64_t oh = ((int64_t)hchunk * a.N + t0 + (hrow <= lastrow ? hrow : lastrow)) * 64 + hcol;
    dma16(gh + oh, bx + 2 * TILE);
    dma16(gd + oh, bx + 2 * TILE + HTILE);
  };
  auto rows_of = [&](int n) -> int {
    if (n >= ntiles) return 0;
    const int64_t left = a.N - (tile0 + n) * 32;
    return left < 32 ? (int)left : 32;
  };
  // rows that do not count: zero d out and d pre (x0 and hid stay: finite values times zero)
  auto fix_tile = [&](int n, int buf) {
    const int nrows = rows_of(n);
    if (nrows < 32) {
      const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
      char* bg = smem + buf * STAGE + TILE + threadIdx.x * 16;
#pragma unroll
      for (int u = 0; u < CPT; ++u)
        if (srow[u] >= nrows) *reinterpret_cast<u16x8*>(bg + u * NTH * 16) = zero;
      if (hrow >= nrows) *reinterpret_cast<u16x8*>(smem + buf * STAGE + 2 * TILE + HTILE + threadIdx.x * 16) = zero;
    }
  };
  int to0[8], to1[8], th0[2], th1[2];
#pragma unroll
  for (int v = 0; v < 8; ++v) tr_offsets<HPB>(v, lane, to0[v], to1[v]);
  tr_offsets<256>(2 * hp, lane, th0[0], th1[0]);
  tr_offsets<256>(2 * hp + 1, lane, th0[1], th1[1]);

  // tile t lives in stage t % NBUF; NBUF - 1 tiles are issued ahead.  Per tile: wait for its DMAs, one block barrier (the
  // tile is visible to every wave AND the stage tile n-1 used is free), refill that stage, compute.
#pragma unroll
  for (int t = 0; t < NBUF - 1; ++t) issue_tile(t, t);
  int bc = 0, bi = NBUF - 1;
#pragma unroll 1
  for (int n = 0; n < ntiles; ++n) {
    __builtin_amdgcn_sched_barrier(0);
    wait_vm<(NBUF - 2) * PER>();
    fix_tile(n, bc);
    __syncthreads();
    issue_tile(n + NBUF - 1, bi);
    if (wave_on) {
      const char* bx = smem + bc * STAGE;
      const char* bg = bx + TILE;
      bf16x8 hb[2], gb[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        hb[u] = tr_read(bx + 2 * TILE, th0[u], th1[u]);
        gb[u] = tr_read(bx + 2 * TILE + HTILE, th0[u], th1[u]);
      }
      if (want_b1) {                             // wave-uniform
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float sm = 0.f;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) sm += (float)gb[u][jj];
          db1[u] += sm;
        }
      }
      // feature tiles in chunks of CV: the reads of chunk c+1 are issued before the MFMAs of chunk c
      constexpr int CV = 4, NCH = NV / CV;
      static_assert(NV % CV == 0, "feature tiles per wave must be a multiple of the chunk");
      bf16x8 fb[2][2 * CV];
      auto reads = [&](int c, int sl) {
#pragma unroll
        for (int i = 0; i < CV; ++i) {
          const int v = fh * NV + c * CV + i;
          fb[sl][i] = tr_read(bx + (v >> 3) * 256, to0[v & 7], to1[v & 7]);
          fb[sl][CV + i] = tr_read(bg + (v >> 3) * 256, to0[v & 7], to1[v & 7]);
        }
      };
      reads(0, 0);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NCH) reads(c + 1, (c + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CV; ++i) {
          const int q = c * CV + i;
          const bf16x8 xf = fb[c & 1][i], gf = fb[c & 1][CV + i];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            dW1[q][u] = mma16(xf, gb[u], dW1[q][u]);
            dW2[q][u] = mma16(gf, hb[u], dW2[q][u]);
          }
          if (want_b2) {                       // wave-uniform
            float sm = 0.f;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) sm += (float)gf[jj];
            db2[q] += sm;
          }
        }
      }
    }
    bc = bc + 1 == NBUF ? 0 : bc + 1; bi = bi + 1 == NBUF ? 0 : bi + 1;
  }
  wait_vm<0>();                                // the refills issued past the slab's end land before the block exits
  if (!wave_on) return;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int64_t item = ((int64_t)kn * a.HW + hw0 + u) * a.nslab + slab;
    float* out = a.part + item * (2 * NF * 256) + lane * 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = fh * NV + i;
      *reinterpret_cast<f32x4*>(out + v * 256) = dW1[i][u];
      *reinterpret_cast<f32x4*>(out + (NF + v) * 256) = dW2[i][u];
    }
    if (want_b1) {
      float s1 = db1[u] + __shfl_xor(db1[u], 16, 64);
      s1 += __shfl_xor(s1, 32, 64);
      if (g == 0) a.partb1[item * 16 + r16] = s1;
    }
  }
  if (want_b2) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = fh * NV + i;
      float s2 = db2[i] + __shfl_xor(db2[i], 16, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (g == 0) a.partb2[((int64_t)kn * a.nslab + slab) * (16 * NF) + 16 * v + r16] = s2;
    }
  }
}

struct DwRedArgs {
  const float *part, *partb1, *partb2;
  float *dW1, *db1, *dW2, *db2;   // (K, 2, H, h), (K, 2, H), (K, 2, h, H), (K, 2, h)
  const int64_t* sink;            // optional device table [K*2][4] of fp32 gradient addresses {W1, b1, W2, b2}: +=
  int K, h, H, HT, HW, nslab;
  int perm;                       // 1: the 16-unit groups are in accumulator-pair order inside their 32-unit chunk (k_big_dw2)
};
// hidden unit of column `col` of 16-unit group `hw`
__device__ __forceinline__ int dw_unit(int hw, int col, int perm) {
  if (!perm) return 16 * hw + col;
  const int p = 16 * (hw & 1) + col, gq = p >> 3, j = p & 7;
  return 32 * (hw >> 1) + (j < 4 ? 4 * gq + j : 16 + 4 * gq + (j - 4));
}
// One thread per dumped accumulator quad: sums the slabs in slab order (fixed order => bit-reproducible).
__global__ void __launch_bounds__(256) k_big_dw_reduce(DwRedArgs a) {
  const int NF = 2 * a.HT;
  const int64_t per_item = (int64_t)2 * NF * 256;
  const int64_t n1 = (int64_t)a.K * 2 * a.HW * per_item;
  const int64_t nb1 = (int64_t)a.K * 2 * a.HW * 16, nb2 = (int64_t)a.K * 2 * NF * 16;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n1 / 4) {   // one thread per accumulator quad (the four registers r of a lane: consecutive features)
    const int e = (int)(idx % (per_item / 4)) * 4;
    const int64_t q = idx / (per_item / 4);
    const int hw = (int)(q % a.HW);
    const int kn = (int)(q / a.HW);
    const float* p = a.part + (((int64_t)kn * a.HW + hw) * a.nslab) * per_item + e;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int sl = 0;
    for (; sl + 4 <= a.nslab; sl += 4) {           // four 16-byte loads in flight, summed in slab order
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (sl + u) * per_item);
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u];
    }
    for (; sl < a.nslab; ++sl) s += *reinterpret_cast<const f32x4*>(p + sl * per_item);
    const int tt = e >> 8, ln = (e >> 2) & 63;
    const int hid = dw_unit(hw, ln & 15, a.perm);
    const int v = tt % NF;
    const int feat = 16 * v + 4 * (ln >> 4);
    if (hid >= a.H) return;
    if (tt < NF) {   // dW1[hid][feat .. feat + 3]
      float* d = a.sink ? reinterpret_cast<float*>(a.sink[kn * 4 + 0]) + (int64_t)hid * a.h + feat
                        : a.dW1 + ((int64_t)kn * a.H + hid) * a.h + feat;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (feat + r < a.h) d[r] = a.sink ? d[r] + s[r] : s[r];
    } else {         // dW2[feat + r][hid]
      float* d = a.sink ? reinterpret_cast<float*>(a.sink[kn * 4 + 2]) + (int64_t)feat * a.H + hid
                        : a.dW2 + ((int64_t)kn * a.h + feat) * a.H + hid;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (feat + r < a.h) d[(int64_t)r * a.H] = a.sink ? d[(int64_t)r * a.H] + s[r] : s[r];
    }
    return;
  }
  idx -= n1 / 4;
  if (idx < nb1) {
    const int col = (int)(idx & 15);
    const int64_t q = idx >> 4;
    const int hw = (int)(q % a.HW);
    const int kn = (int)(q / a.HW);
    const int hid = dw_unit(hw, col, a.perm);
    if (hid >= a.H) return;
    const float* p = a.partb1 + (((int64_t)kn * a.HW + hw) * a.nslab) * 16 + col;
    float s = 0.f;
    for (int sl = 0; sl < a.nslab; ++sl) s += p[sl * 16];
    if (a.sink) reinterpret_cast<float*>(a.sink[kn * 4 + 1])[hid] += s;
    else a.db1[(int64_t)kn * a.H + hid] = s;
    return;
  }
  idx -= nb1;
  if (idx < nb2) {
    const int hp = 16 * NF;
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
  return h >= 97 && h <= 256 && h % 8 == 0 && H >= 65 && Hp + hp <= kBiasBlockFloats;
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
  if (mode == M_FWD) { opt((const void*)k_big_chain<HT, M_FWD, ACT>); k_big_chain<HT, M_FWD, ACT><<<grid, 512, lds, st>>>(a); }
  else if (mode == M_INV) { opt((const void*)k_big_chain<HT, M_INV, ACT>); k_big_chain<HT, M_INV, ACT><<<grid, 512, lds, st>>>(a); }
  else { opt((const void*)k_big_chain<HT, M_CHAIN, ACT>); k_big_chain<HT, M_CHAIN, ACT><<<grid, 512, lds, st>>>(a); }
}
template <int HT, int ACT>
static void launch_nets(const NetsArgs& a, unsigned grid, size_t lds, hipStream_t st) {
  if (a.hbuf) {
    (void)hipFuncSetAttribute((const void*)k_big_nets<HT, ACT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_big_nets<HT, ACT, true><<<grid, 512, lds, st>>>(a);
  } else {
    (void)hipFuncSetAttribute((const void*)k_big_nets<HT, ACT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_big_nets<HT, ACT, false><<<grid, 512, lds, st>>>(a);
  }
}
template <int HT, int ACT>
static void launch_dw(const DwBigArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  if constexpr (HT <= 4) {
    (void)hipFuncSetAttribute((const void*)k_big_dw<HT, ACT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_big_dw<HT, ACT, 0><<<grid, 512, lds, st>>>(a);
  } else {
    (void)hipFuncSetAttribute((const void*)k_big_dw<HT, ACT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_big_dw<HT, ACT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_big_dw<HT, ACT, 1><<<grid, 512, lds, st>>>(a);
    k_big_dw<HT, ACT, 2><<<grid, 512, lds, st>>>(a);
  }
}
template <int HT>
static void launch_dw2(const Dw2Args& a, dim3 grid, hipStream_t st) {
  const size_t lds = (size_t)(HT <= 4 ? 4 : 3) * (2 * 32 * HT * 64 + 2 * 32 * 256);
  (void)hipFuncSetAttribute((const void*)k_big_dw2<HT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_big_dw2<HT><<<grid, 512, lds, st>>>(a);
}
#define BIG_DISPATCH(HTv, actv, CALL)                                           \
  do {                                                                          \
    if ((HTv) == 4) { if ((actv) == ACT_GELU) { CALL(4, ACT_GELU); } else { CALL(4, -1); } } \
    else { if ((actv) == ACT_GELU) { CALL(8, ACT_GELU); } else { CALL(8, -1); } }            \
  } while (0)

static size_t ring_lds(int HT, int ipb) {
  const int g = ipb == 2 ? (HT <= 4 ? RingCfg<4>::G2 : RingCfg<8>::G2) : (HT <= 4 ? RingCfg<4>::G3 : RingCfg<8>::G3);
  const int gi = g * ipb, p0 = ipb == 2 ? 2 : 4;
  const int r = 2 * gi > p0 + gi ? 2 * gi : p0 + gi;
  return (size_t)r * 2 * HT * 1024 + kBiasSlots * 4096;
}

static void dw_big_geom(int64_t N, const BigGeom& g, int& nslab, int& slab_tiles) {
  const int64_t ntiles = (N + 31) / 32;
  const int nw = 8;                                      // hidden 16-groups (waves) per block
  const int64_t blocks0 = (int64_t)g.K * 2 * ((g.Hp / 16 + nw - 1) / nw);
  // one block per CU in total (MIVAE_BIG_DW_BLOCKS overrides the 256): every further slab costs a prologue (the wave's
  // weight fragments from the fp32 parameters) and 16 KB of partials per wave to dump and fold -- C3 backward with 768 /
  // 512 / 256 blocks: 1037 / 1023 / 995 us
  static const int target = [] { const char* e = getenv("MIVAE_BIG_DW_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 256; }();
  int64_t want = (target + blocks0 - 1) / blocks0;
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
  const int64_t n = stream_kind == ST_BIAS ? K * 2 * kBiasBlockFloats : stream_items(g, stream_kind) * (g.item / 16);   // threads
  k_big_pack<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(ptr_table, (char*)out, g, stream_kind);
  MIVAE_CHECK_LAUNCH("nice_big_pack");
  return MIVAE_OK;
}

// The same packing for several streams at once: outs[kind] (kind as above; NULL = skip), each
// mivae_nice_big_pack_bytes(K, h, H, kind) bytes, 16-byte aligned.  One launch instead of one per stream.
extern "C" int mivae_nice_big_pack_multi(const int64_t* ptr_table, void* const* outs, int64_t K, int64_t h, int64_t H,
                                         void* stream) {
  MIVAE_REQUIRE(ptr_table && outs, "nice_big_pack_multi: null pointer");
  MIVAE_REQUIRE(K >= 1 && K <= 4096 && big_shape_ok(h, H), "nice_big_pack_multi: unsupported shape K=%ld h=%ld H=%ld", (long)K, (long)h, (long)H);
  const BigGeom g = big_geom_inst((int)h, (int)H, (int)K);
  PackMulti m = {};
  unsigned total = 0;
  for (int st = 0; st <= ST_BIAS; ++st) {
    m.first[st] = total;
    m.out[st] = (char*)outs[st];
    if (!outs[st]) continue;
    MIVAE_REQUIRE(aligned16(outs[st]), "nice_big_pack_multi: outputs must be 16-byte aligned");
    const int64_t n = st == ST_BIAS ? K * 2 * kBiasBlockFloats : stream_items(g, st) * (g.item / 16);   // threads
    total += (unsigned)((n + 255) / 256);
  }
  m.first[ST_BIAS + 1] = total;
  if (total == 0) return MIVAE_OK;
  k_big_pack_multi<<<total, 256, 0, (hipStream_t)stream>>>(ptr_table, m, g);
  MIVAE_CHECK_LAUNCH("nice_big_pack_multi");
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
  const size_t lds = ring_lds(g.HT, 2);
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
  return (int64_t)K * 2 * (g.Hp / 16) * nslab * (4 * g.HT * 256 + 16) + (int64_t)K * 2 * nslab * g.hp;
}

// elements (latent dtype) of the optional hidden-activation scratch of mivae_nice_big_bwd
extern "C" int64_t mivae_nice_big_hidden_elems(int64_t N, int64_t h, int64_t H, int64_t K) {
  if (N < 1 || K < 1 || !big_shape_ok(h, H)) return 0;
  const BigGeom g = big_geom_inst((int)h, (int)H, (int)K);
  return 2 * K * 2 * N * g.Hp;
}

// Backward of mivae_nice_big_fwd (forward direction): dz_in (N, D) and the parameter gradients.
//   dlogp (Bn): cotangent of every sentence's sum of tok_val (as mivae_nice_stack_bwd); per-token cotangents: call with
//   Bn = N, S = 1 and the mask flattened;
//   packed_chain / packed_nets / packed_bias: ST_CHAIN / ST_NETS / ST_BIAS images of the same weights;
//   dSdT: scratch (K, 2, N, h) in the latent dtype; workspace: mivae_nice_big_dw_workspace_floats floats;
//   hidden: scratch of mivae_nice_big_hidden_elems elements (latent dtype, 16-byte aligned) or NULL.  With it the nets
//   kernel saves the hidden activations and d pre and the weight gradients contract them directly (k_big_dw2); without
//   it they are recomputed from x0 and d s / d t (k_big_dw: no extra memory, ~2x the weight-gradient time);
//   gradients go to dW1 (K,2,H,h), db1 (K,2,H), dW2 (K,2,h,H), db2 (K,2,h), or, when grad_ptrs (device int64[K][8] in
//   mivae_nice_pack's order) is given, are ADDED into those buffers.
extern "C" int mivae_nice_big_bwd(const void* z_in, const void* z_out, const void* x1_hist, const int64_t* mask,
                                  const int64_t* ptr_table, const void* packed_chain, const void* packed_nets,
                                  const void* packed_bias, const void* dz_out, const float* dlogp, void* dz_in,
                                  void* dSdT, void* hidden, float* workspace,
                                  float* dW1, float* db1, float* dW2, float* db2, const int64_t* grad_ptrs,
                                  int64_t Bn, int64_t mask_div, int64_t S, int64_t D, int64_t H, int64_t K, int act,
                                  int base, float mu, float std, void* stream) {
  int rc = big_common_checks("nice_big_bwd", Bn, S, D, H, K, act, mask_div);
  if (rc) return rc;
  if (Bn == 0) return MIVAE_OK;
  MIVAE_REQUIRE(z_in && x1_hist && ptr_table && packed_chain && packed_nets && packed_bias && dz_in && dSdT && workspace,
                "nice_big_bwd: null pointer");
  MIVAE_REQUIRE(grad_ptrs || (dW1 && db1 && dW2 && db2), "nice_big_bwd: no gradient destination");
  MIVAE_REQUIRE(!base || (std > 0.f && z_out), "nice_big_bwd: base log-prob needs z_out and std > 0");
  MIVAE_REQUIRE(aligned16(z_in) && aligned16(x1_hist) && aligned16(dz_in) && aligned16(dSdT) && aligned16(workspace) &&
                (!z_out || aligned16(z_out)) && (!dz_out || aligned16(dz_out)) && (!hidden || aligned16(hidden)) &&
                aligned16(packed_chain) &&
                aligned16(packed_nets) && aligned16(packed_bias), "nice_big_bwd: operands must be 16-byte aligned");
  const BigGeom g = big_geom_inst((int)(D / 2), (int)H, (int)K);
  hipStream_t st = (hipStream_t)stream;
  const int64_t N = Bn * S;
  const unsigned grid = (unsigned)((N + 127) / 128);
  const size_t lds = ring_lds(g.HT, 2), lds_nets = ring_lds(g.HT, 3);
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
    if (hidden) { a.hbuf = (T*)hidden; a.gbuf = (T*)hidden + (int64_t)K * 2 * N * g.Hp; }
#define CALL(HTc, ACTc) launch_nets<HTc, ACTc>(a, grid, lds_nets, st)
    BIG_DISPATCH(g.HT, act, CALL);
#undef CALL
    MIVAE_CHECK_LAUNCH("nice_big_bwd_nets");
  }
  {
    DwBigArgs a = {};
    a.z_in = (const T*)z_in; a.dSdT = (const T*)dSdT; a.mask = mask; a.ptrs = ptr_table;
    a.bias = (const float*)packed_bias; a.N = N; a.S = (int)S; a.mask_div = (int)mask_div; a.D = (int)D; a.K = (int)K;
    a.h = g.h; a.H = g.H; a.act = act;
    dw_big_geom(N, g, a.nslab, a.slab_tiles);
    a.HW = g.Hp / 16;
    const int64_t nitems = (int64_t)K * 2 * a.HW * a.nslab;
    a.part = workspace;
    a.partb1 = workspace + nitems * 4 * g.HT * 256;
    a.partb2 = a.partb1 + nitems * 16;
    const int nw = 8;
    const int64_t ngrp = (int64_t)a.nslab * K * 2, zsib = (a.HW + nw - 1) / nw;
    dim3 dgrid((unsigned)(((ngrp + 7) / 8) * 8 * zsib));   // 1-D: see the block order in k_big_dw
    if (hidden) {
      Dw2Args b = {};
      b.z_in = a.z_in; b.dSdT = a.dSdT; b.hbuf = (const T*)hidden; b.gbuf = (const T*)hidden + (int64_t)K * 2 * N * g.Hp;
      b.part = a.part; b.partb1 = a.partb1; b.partb2 = a.partb2; b.N = N; b.D = (int)D; b.K = (int)K; b.h = g.h;
      b.HW = a.HW; b.Hp = g.Hp; b.nslab = a.nslab; b.slab_tiles = a.slab_tiles;
      if (g.HT == 4) launch_dw2<4>(b, dgrid, st); else launch_dw2<8>(b, dgrid, st);
    } else {
      const size_t dlds = (size_t)(g.HT <= 4 ? BIG_DW_NBUF : 4) * 2 * 32 * g.HT * 64;   // NBUF tile buffers x 2 operands
#define CALL(HTc, ACTc) launch_dw<HTc, ACTc>(a, dgrid, dlds, st)
      BIG_DISPATCH(g.HT, act, CALL);
#undef CALL
    }
    MIVAE_CHECK_LAUNCH("nice_big_dw");
    DwRedArgs r = {};
    r.part = a.part; r.partb1 = a.partb1; r.partb2 = a.partb2;
    r.dW1 = dW1; r.db1 = db1; r.dW2 = dW2; r.db2 = db2;
    r.K = (int)K; r.h = g.h; r.H = g.H; r.HT = g.HT; r.HW = a.HW; r.nslab = a.nslab;
    // the sink table is [K][8] = {W1s,b1s,W2s,b2s,W1t,b1t,W2t,b2t}: exactly [kn][4] with kn = 2k + net
    r.sink = grad_ptrs;
    r.perm = hidden ? 1 : 0;
    const int64_t nthreads = (int64_t)K * 2 * a.HW * g.HT * 256 + (int64_t)K * 2 * a.HW * 16 + (int64_t)K * 2 * 2 * g.HT * 16;
    k_big_dw_reduce<<<(unsigned)((nthreads + 255) / 256), 256, 0, st>>>(r);
    MIVAE_CHECK_LAUNCH("nice_big_dw_reduce");
  }
  return MIVAE_OK;
}
