// Backward of small fused NICE stacks with the coupling nets of all steps running side by side (gfx950).
// Reference semantics: autograd of NICECoupling.forward (vae_lm/models/base/flows/couplings.py:141-162) with
// FeedForwardCoupling nets (couplings.py:50-51, nn/feedforward.py:93-98) iterated by FlowPrior.forward_pass
// (nonauto_lm/priors/flow_prior.py:37-52) + base log-prob (priors/prior.py:148-158) -- the same outputs as
// mivae_nice_stack_bwd + mivae_nice_dw (nice_stack.hip), re-split for the shapes where that pair is latency-bound.
#include "nice_common.hpp"
#include <stdlib.h>

using namespace mivae;

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_cvoid_t;

// ---- [token][feature] bf16 images in LDS: written from accumulator tiles, read back through transposing reads -----
constexpr int kImgBytes = 2048;   // 32 tokens x 32 features bf16, 64-byte rows

__device__ __forceinline__ int img_off(int row, int slot) { return row * 64 + 8 * (slot ^ ((row >> 1) & 7)); }

// Per-lane byte offsets inside an image, computed once per kernel: po[g] = the 8-byte slot of accumulator registers
// 4g .. 4g+3 (features 8g + 4hh ..) in the lane's token row; fo[0 / 1] = the two transposing reads of k-step 0
// (k-step 1 is 1 KiB further: 16 rows, same swizzle).
struct ImgLane {
  int po[4], fo[2];
  __device__ __forceinline__ void init(int lane) {
    const int r32 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) po[g] = img_off(r32, 2 * g + hh);
    const int g4 = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = 8 * (g4 >> 1) + q, slot = 4 * (g4 & 1) + p;
    fo[0] = img_off(row, slot);
    fo[1] = img_off(row + 4, slot);
  }
};
// accumulator tile (feature rows in the registers, token on the lane) -> [token][feature] image
__device__ __forceinline__ void img_put(char* img, const ImgLane& L, const f32x16& x) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    u16x4 v = {f32_to_bf16(x[4 * g]), f32_to_bf16(x[4 * g + 1]), f32_to_bf16(x[4 * g + 2]), f32_to_bf16(x[4 * g + 3])};
    *reinterpret_cast<u16x4*>(img + L.po[g]) = v;
  }
}
// MFMA fragment of k-step s with the TOKENS as the contraction: f[j] = img[token 16 s + 8 hh + j][feature r32]
// (A operand: row = feature; B operand: column = feature).  EXEC must be all ones.
__device__ __forceinline__ bf16x8 img_frag(const char* img, const ImgLane& L, int s) {
  typedef __attribute__((address_space(3))) bf16x4_tr* lds_ptr;
  const bf16x4_tr v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(img + s * 1024 + L.fo[0]));
  const bf16x4_tr v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(img + s * 1024 + L.fo[1]));
  return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Backward of mivae_nice_stack_bwd for the shapes where nice_stack.hip's kernel is latency-bound (bf16, h <= 32,
// H <= 64, K <= 4: BASELINE C2 / C4).  There a wave owns 32 tokens and walks 3 K HC staged weight tasks one after
// another (24 barrier-separated tasks at C2, two waves per CU).  Here the work is re-split: the K x 2 coupling nets only
// depend on x0, so a block of 2K waves gives every (step, net) its own wave for a 32-token tile.
//   phase 1  every wave recomputes ITS net's hidden layer (value + derivative) from x0 with its own weight fragments
//            (straight from the packed stream in L2: no LDS staging, no per-task barrier); the s-net waves finish s_k
//            and publish exp(s_k) in LDS;                                                            -- barrier A
//   chain    every wave walks the affine chain down to its own step on the published exponentials (<= K x 16 fmuls
//            per lane, redundantly) and forms its d s_k / d t_k in registers;
//   phase 2  d hid = W2^T d out, d pre = d hid act', d x0 partial = W1^T d pre (accumulator -> B operand);
//   emit     hid, d pre, d out (and x0) are parked as [token][feature] bf16 images in the wave's LDS (8-byte slots
//            XOR-swizzled: packed writes and transposing reads both conflict free) and leave through
//            ds_read_b64_tr_b16 as 16-byte pieces of k_dw_partial's token-blocked layout: a lane holds 8 consecutive
//            tokens of one feature, 32 lanes cover 512 contiguous bytes (the old kernel stored 2 bytes per lane);
//                                                                                                    -- barrier B
//   reduce   the 2K partial d x0 tiles are summed in wave order and stored with the upstream x0 cotangent.
// Block barrier that only waits for the wave's LDS traffic.  __syncthreads() also drains vmcnt: here that would wait out
// the weight fragments requested ahead of barrier A and the side-output stores issued ahead of barrier B (1-2 us each).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int ACT, int HC>
__global__ void __launch_bounds__(512) k_stack_bwd_par(StackArgs a, int ntiles, int zero) {
  typedef MmaBF16 P;
  typedef unsigned short T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Geom& g = a.g;
  const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NW = blockDim.x >> 6;                       // = 2 K
  // waves 0 .. K-1 run the s-nets, K .. 2K-1 the t-nets: with waves dealt round-robin to the four SIMDs every SIMD gets one
  // of each (the s-net waves also finish s_k and publish exp(s_k): paired on a SIMD they were the last at barrier A)
  const int k = wave % a.K, net = wave / a.K;
  const int h = g.h;
  // LDS: [K] exp(s_k) tiles | [NW] d x0 partial tiles | x0 image | per wave: HC hid + HC (act' then d pre) + d out
  // images, biases (b1 | b2)
  float* sbuf = reinterpret_cast<float*>(smem);
  float* dxbuf = sbuf + a.K * 1024;
  char* x0img = reinterpret_cast<char*>(dxbuf + NW * 1024);
  constexpr int kWaveBytes = (2 * HC + 1) * kImgBytes + 512;   // biases: HC x [b1 chunk (32 floats) | b2 (32 floats)]
  char* const wbase = x0img + kImgBytes + wave * kWaveBytes;
  constexpr int oHid = 0, oDpre = HC * kImgBytes, oDout = 2 * HC * kImgBytes, oBias = (2 * HC + 1) * kImgBytes;
  ImgLane L;
  L.init(lane);
  // this wave's weights: fragments of the packed backward stream (k_nice_pack), tasks (phase, c): 0 = [W1n s][W2p s],
  // 1 / 2 = s / t net [W1n][W2Tp][W1Tp]; 16 KB per wave, read once (below) and kept in registers
  const char* pk = a.packed + g.off_b + (int64_t)(a.K - 1 - k) * 3 * HC * g.tbb;
  const bf16x8* f0 = reinterpret_cast<const bf16x8*>(pk) + lane;
  const int tstride = (int)(g.tbb / 16);
  // biases: the 256-byte tail of the net's task c ([b1 chunk c | b2]) goes straight to the wave's LDS by LDS-DMA (one
  // instruction per chunk, no register, no wait here: the first tile's loads are issued behind it and one vmcnt(0)
  // ahead of phase 1 covers both)
#pragma unroll
  for (int c = 0; c < HC; ++c) {
    const char* tail = pk + (int64_t)((1 + net) * HC + c) * g.tbb + g.nfe_b * 2;
    __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(tail + lane * 4), (lds_void_t*)(wbase + oBias + c * 256), 4, 0, 0);
  }
  const float c2 = 2.0f * a.inv2var;
  const T* zin = reinterpret_cast<const T*>(a.z_in);
  const T* zout = reinterpret_cast<const T*>(a.z_out);
  const T* gout = reinterpret_cast<const T*>(a.dz_out);
  T* dzin = reinterpret_cast<T*>(a.dz_in);
  // side outputs (token-blocked, 8 tokens per 16 bytes): this wave's sections
  const int64_t ldn = a.Npad;
  T* const Gt = reinterpret_cast<T*>(a.Gt) + (int64_t)(2 * k + net) * g.Hp * ldn;
  T* const Ht = reinterpret_cast<T*>(a.Ht) + (int64_t)(2 * k + net) * g.Hp * ldn;
  T* const dSt = reinterpret_cast<T*>(a.dSt) + (int64_t)(2 * k + net) * g.hp * ldn;
  T* const x0t = reinterpret_cast<T*>(a.x0t);

  // this wave's 16 weight fragments stay in registers for all of the block's tiles.  (Fetched per tile they were the
  // largest single cost of a tile: 8 waves x 16 KB through the CU's 64 B/clk vector-memory path = 2 k cycles of issue,
  // during which the last wave has not even started -- stamped with s_memtime: loads issued 5.5 k / 9.2 k cycles after
  // tile start for waves 0 / 7 of a 22 k-cycle tile.)  They were live through most of a tile anyway: same register peak.
  bf16x8 w1n[HC][2], w2p[HC][2], w2t[HC][2], w1t[HC][2];
#pragma unroll
  for (int c = 0; c < HC; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      w1n[c][i] = f0[((1 + net) * HC + c) * tstride + i * 64];
      w2p[c][i] = f0[c * tstride + (2 + i) * 64];       // s-net waves only use it
      w2t[c][i] = f0[((1 + net) * HC + c) * tstride + (2 + i) * 64];
      w1t[c][i] = f0[((1 + net) * HC + c) * tstride + (4 + i) * 64];
    }

  // What phase 1 needs of a tile -- x0, the mask word, the sentence's log-det cotangent -- is requested one tile AHEAD
  // (a tile's HBM round trip is ~3.6 k of its ~21 k cycles: stamped); everything else of a tile is requested at its top
  // and lands under phase 1.  `head` issues those three loads for the tile of lane-token `t` (clamped into range).
  u16x8 x0raw[2];
  int mraw;
  float dlraw;
  auto head = [&](int64_t t, u16x8 (&x0r)[2], int& mr, float& dr) __attribute__((always_inline)) {
    const int64_t tcl = t < a.N ? t : a.N - 1;
    const int t32 = (int)tcl, bn = t32 / a.S;              // N < 2^31 (host check): 32-bit divisions
    // the int64 0 / 1 mask is read as its low word
    mr = a.mask ? reinterpret_cast<const int*>(a.mask)[2 * ((int64_t)(bn / a.mask_div) * a.S + (t32 - bn * a.S))] : 1;
    dr = a.dlogp ? a.dlogp[bn] : 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      x0r[ks] = *reinterpret_cast<const u16x8*>(zin + tcl * a.D + (16 * ks + 8 * hh < h ? 16 * ks + 8 * hh : 0));
  };
  head((int64_t)blockIdx.x * 32 + r32, x0raw, mraw, dlraw);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(mraw), "+v"(dlraw) : : "memory");   // also the LDS-DMA of the biases, which the
                                                                                // compiler does not count
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // `vz` is zero, but not to the compiler: what is addressed through it is computed inside the loop instead of being
    // hoisted out of it (hoisted, the bias tiles and image addresses cost > 100 registers)
    const int vz = tile * zero;
    char* const wb = wbase + vz;
    const float* const bl = reinterpret_cast<const float*>(wb + oBias);
    char* const x0i = x0img + vz;
    const int64_t tok = (int64_t)tile * 32 + r32;
    const bool valid = tok < a.N;
    const bool withbase = a.base && a.dlogp;
    // ---- a tile of padding only (every wave of the block sees the same 32 mask words: block-uniform, no barrier inside):
    // zero gradient rows and zero side-output columns (k_dw_partial reads all Npad columns), nothing else read or computed
    if (!__any(valid && mraw != 0)) {
      head((int64_t)(tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile) * 32 + r32, x0raw, mraw, dlraw);
      const u16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
      const u16x4 z4 = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int64_t tb = (int64_t)tile * 4 + 2 * s + hh;
#pragma unroll
        for (int c = 0; c < HC; ++c) {
          *reinterpret_cast<u16x8*>(Ht + (tb * g.Hp + 32 * c + r32) * 8) = z8;
          *reinterpret_cast<u16x8*>(Gt + (tb * g.Hp + 32 * c + r32) * 8) = z8;
        }
        *reinterpret_cast<u16x8*>(dSt + (tb * g.hp + r32) * 8) = z8;
        if (wave == 0) *reinterpret_cast<u16x8*>(x0t + (tb * g.hp + r32) * 8) = z8;
      }
      for (int q = wave; q < 8; q += NW) {                 // d z_in row of the lane's token: eight 4-feature groups per half pair
        const int f = 8 * (q & 3) + 4 * hh;
        if (valid && f < h) *reinterpret_cast<u16x4*>(dzin + tok * a.D + (q >> 2) * h + f) = z4;
      }
      continue;
    }
    // ---- the tile's other global loads, branch-free and back to back: addresses are clamped into range (token N - 1,
    // feature 0) and what must not count is zeroed by a select afterwards.  (A load inside `if (valid && f < h)` makes the
    // compiler wait for it inside the branch: twelve serialised round trips per tile.)
    const int64_t tc = valid ? tok : a.N - 1;
    int fcl[4];                                            // the lane's four 4-feature groups, clamped
#pragma unroll
    for (int q = 0; q < 4; ++q) fcl[q] = 8 * q + 4 * hh < h ? 8 * q + 4 * hh : 0;
    u16x4 g1raw[4], zkraw[4], x1raw[4];
    const u16x4 zero4 = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      g1raw[q] = gout ? *reinterpret_cast<const u16x4*>(gout + tc * a.D + h + fcl[q]) : zero4;
      zkraw[q] = withbase ? *reinterpret_cast<const u16x4*>(zout + tc * a.D + h + fcl[q]) : zero4;
      x1raw[q] = net == 0 ? *reinterpret_cast<const u16x4*>(reinterpret_cast<const T*>(a.x1_hist) + ((int64_t)k * a.N + tc) * h + fcl[q])
                          : zero4;
    }
    // upstream cotangent (+ base term input) of the x0 slices this wave reduces after barrier B
    u16x4 g0raw[2], z0raw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = wave + i * NW < 4 ? wave + i * NW : 0;
      g0raw[i] = gout ? *reinterpret_cast<const u16x4*>(gout + tc * a.D + fcl[q]) : zero4;
      z0raw[i] = withbase ? *reinterpret_cast<const u16x4*>(zout + tc * a.D + fcl[q]) : zero4;
    }
    // ---- selects: padding tokens, tokens past N and padded features contribute exact zeros
    const float m = valid ? (float)mraw : 0.f;
    const float dl = valid ? dlraw * m : 0.f;
    const bool on = m != 0.f;
    bf16x8 x0f[2];
    {
      const u16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        x0f[ks] = __builtin_bit_cast(bf16x8, (on && 16 * ks + 8 * hh < h) ? x0raw[ks] : zero8);
    }
    // the next tile's head, behind this tile's loads in the queue (the last tile re-reads its own: harmless)
    head((int64_t)(tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile) * 32 + r32, x0raw, mraw, dlraw);
    if (wave == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const u16x8 raw = __builtin_bit_cast(u16x8, x0f[ks]);
        const u16x4 lo = {raw[0], raw[1], raw[2], raw[3]}, hi = {raw[4], raw[5], raw[6], raw[7]};
        *reinterpret_cast<u16x4*>(x0i + img_off(r32, 4 * ks + 2 * hh)) = lo;
        *reinterpret_cast<u16x4*>(x0i + img_off(r32, 4 * ks + 2 * hh + 1)) = hi;
      }
    }
    // ---- phase 1: hidden layer of this wave's net (value -> image, derivative -> lane-linear bf16), s-net: exp(s_k)
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    {
#pragma unroll
      for (int c = 0; c < HC; ++c) {
        f32x16 hid = bias_tile(bl + 64 * c, hh), dact;
        hid = P::mma(w1n[c][0], x0f[0], hid);
        hid = P::mma(w1n[c][1], x0f[1], hid);
        act_tile_fd<ACT, true>(a.act, hid, dact);
        img_put(wb + oHid + c * kImgBytes, L, hid);
        {   // act' parked as bf16 in the lane's own 32 bytes (d pre is rounded to bf16 for its products anyway), in the
            // area the d pre image of the chunk takes over in phase 2
          u16x8 d0, d1;
#pragma unroll
          for (int j = 0; j < 8; ++j) { d0[j] = f32_to_bf16(dact[j]); d1[j] = f32_to_bf16(dact[8 + j]); }
          *reinterpret_cast<u16x8*>(wb + oDpre + c * kImgBytes + lane * 16) = d0;
          *reinterpret_cast<u16x8*>(wb + oDpre + c * kImgBytes + 1024 + lane * 16) = d1;
        }
        if (net == 0) {
          sacc = P::mma(w2p[c][0], P::from_acc<0>(hid), sacc);
          sacc = P::mma(w2p[c][1], P::from_acc<1>(hid), sacc);
        }
      }
    }
    if (net == 0) {
      const f32x16 bt = bias_tile(bl + 32, hh);
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = fast_exp(sacc[r] + bt[r]);
      const f32x16 e1[1] = {sacc};
      stash_put<1>(sbuf + k * 1024 + lane * 4 + vz, e1);
    }
    lds_barrier();     // A: every exp(s_j) of this tile is published, the x0 image is complete
    // ---- affine chain down to this wave's step: d t_k = g m prod_{j > k} e_j, d s_k = d t_k e_k x1_k + d logdet
    f32x16 dout, x1[1];      // the cotangent / history words requested at the top of the tile have landed under phase 1
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool in = 8 * q + 4 * hh < h;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float gv = (valid && in) ? bf16_to_f32(g1raw[q][i]) : 0.f;
        // d / d z_K of the base log-prob: dl * -(z_K - mu) / std^2
        if (withbase) gv += in ? -dl * (bf16_to_f32(zkraw[q][i]) - a.mu) * c2 : 0.f;
        dout[4 * q + i] = gv * m;
        x1[0][4 * q + i] = (on && in) ? bf16_to_f32(x1raw[q][i]) : 0.f;
      }
    }
    for (int j = a.K - 1; j > k; --j) {
      f32x16 ej[1];
      stash_get<1>(sbuf + j * 1024 + lane * 4 + vz, ej);
#pragma unroll
      for (int r = 0; r < 16; ++r) dout[r] *= ej[0][r];
    }
    if (net == 0) {
      f32x16 ge[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ge[0][r] = dout[r] * sacc[r];
        dout[r] = (acc_row(r, hh) < h) ? ge[0][r] * x1[0][r] + dl : 0.f;
      }
      if (k == 0) store_acc<T, true, 1>(dzin + tok * a.D + h, valid, h, hh, ge);   // d z_in, x1 half
    }
    // ---- phase 2: this net's backward
    img_put(wb + oDout, L, dout);
    f32x16 dx0p;
#pragma unroll
    for (int r = 0; r < 16; ++r) dx0p[r] = 0.f;
    {
      const bf16x8 bo0 = P::from_acc<0>(dout), bo1 = P::from_acc<1>(dout);
#pragma unroll
      for (int c = 0; c < HC; ++c) {
        f32x16 dh;
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = 0.f;
        dh = P::mma(w2t[c][0], bo0, dh);
        dh = P::mma(w2t[c][1], bo1, dh);
        {
          const u16x8 d0 = *reinterpret_cast<const u16x8*>(wb + oDpre + c * kImgBytes + lane * 16);
          const u16x8 d1 = *reinterpret_cast<const u16x8*>(wb + oDpre + c * kImgBytes + 1024 + lane * 16);
#pragma unroll
          for (int j = 0; j < 8; ++j) { dh[j] *= bf16_to_f32(d0[j]); dh[8 + j] *= bf16_to_f32(d1[j]); }
        }
        // every lane has read its act' (LDS accesses of a wave complete in order): the area becomes the d pre image
        img_put(wb + oDpre + c * kImgBytes, L, dh);
        dx0p = P::mma(w1t[c][0], P::from_acc<0>(dh), dx0p);
        dx0p = P::mma(w1t[c][1], P::from_acc<1>(dh), dx0p);
      }
    }
    {
      const f32x16 d1[1] = {dx0p};
      stash_put<1>(dxbuf + wave * 1024 + lane * 4 + vz, d1);
    }
    // ---- emit the operands of the weight-gradient products: element (feature f, token t) of a section with F rows
    // at ((t / 8) F + f) 8 + t % 8; a fragment = 8 consecutive tokens (16 s + 8 hh ..) of feature r32 = one 16-byte piece
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int64_t tb = (int64_t)tile * 4 + 2 * s + hh;      // 8-token block of this lane's piece
#pragma unroll
      for (int c = 0; c < HC; ++c) {
        *reinterpret_cast<bf16x8*>(Ht + (tb * g.Hp + 32 * c + r32) * 8) = img_frag(wb + oHid + c * kImgBytes, L, s);
        *reinterpret_cast<bf16x8*>(Gt + (tb * g.Hp + 32 * c + r32) * 8) = img_frag(wb + oDpre + c * kImgBytes, L, s);
      }
      *reinterpret_cast<bf16x8*>(dSt + (tb * g.hp + r32) * 8) = img_frag(wb + oDout, L, s);
      if (wave == 0) *reinterpret_cast<bf16x8*>(x0t + (tb * g.hp + r32) * 8) = img_frag(x0i, L, s);
    }
    lds_barrier();     // B: all partial d x0 tiles are in LDS; every wave is done with sbuf and the x0 image
    // ---- d z_in, x0 half: upstream cotangent (+ base term) + the 2K partial tiles in wave order
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      const int q = wave + qi * NW;
      if (q >= 4) break;                       // wave-uniform
      const int f = 8 * q + 4 * hh;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = (valid && f < h) ? bf16_to_f32(g0raw[qi][i]) : 0.f;
        if (withbase && f < h) v[i] += -dl * (bf16_to_f32(z0raw[qi][i]) - a.mu) * c2;     // dl is zero past N
      }
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
      for (int w = NW - 1; w >= 0; --w) sum += *reinterpret_cast<const f32x4*>(dxbuf + w * 1024 + q * 256 + lane * 4 + vz);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = v[i] * m + sum[i];
      if (valid && f < h) RowIO<T, 4>::store(dzin + tok * a.D + f, v);
    }
  }
}

}  // namespace

// ---- host side: called by mivae_nice_stack_bwd (nice_stack.hip) --------------------------------------
bool mivae::nice_small_bwd_takes(const StackArgs& a, bool bf16, bool vec) {
  const char* e = getenv("MIVAE_NICE_BWD_PAR");     // read per call: tests and A/B timings switch it inside one process
  const bool off = e && atoi(e) == 0;
  const Geom& g = a.g;
  return !off && bf16 && vec && g.HT == 1 && g.HC <= 2 && g.h % 8 == 0 && a.K <= 4 && a.x1_hist && a.Gt && a.Ht && a.dSt &&
         a.x0t && aligned16(a.Gt) && aligned16(a.Ht) && aligned16(a.dSt) && aligned16(a.x0t);
}

int mivae::nice_small_bwd_launch(const StackArgs& a, hipStream_t st) {
  const Geom& g = a.g;
  MIVAE_REQUIRE(a.Npad < (int64_t)1 << 31, "nice_stack_bwd: too many tokens for the small-stack kernel");
  const int ntiles = (int)(a.Npad / 32);                    // Npad is a multiple of 64: tiles past N emit zero columns
  const int nblk = ntiles < 256 ? ntiles : 256, NW = 2 * a.K;
  const size_t lds = (size_t)(a.K + NW) * 4096 + kImgBytes + (size_t)NW * ((2 * g.HC + 1) * kImgBytes + 512);
#define MIVAE_BWD_PAR_LAUNCH(A_, HC_)                                                                              \
  do {                                                                                                             \
    if (lds > 64 * 1024)                                                                                           \
      (void)hipFuncSetAttribute((const void*)k_stack_bwd_par<A_, HC_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                         \
    k_stack_bwd_par<A_, HC_><<<nblk, NW * 64, lds, st>>>(a, ntiles, 0);                                            \
  } while (0)
  if (a.act == ACT_GELU) {
    if (g.HC == 1) MIVAE_BWD_PAR_LAUNCH(ACT_GELU, 1); else MIVAE_BWD_PAR_LAUNCH(ACT_GELU, 2);
  } else {
    if (g.HC == 1) MIVAE_BWD_PAR_LAUNCH(-1, 1); else MIVAE_BWD_PAR_LAUNCH(-1, 2);
  }
#undef MIVAE_BWD_PAR_LAUNCH
  MIVAE_CHECK_LAUNCH("nice_stack_bwd (parallel nets)");
  return MIVAE_OK;
}
