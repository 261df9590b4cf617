// Masked multi-head self-attention of the transformer bracket (nn/transformer.py:7-117: scores = q k^T / sqrt(d),
// masked_fill(-1e13) on padded keys and, for the autoregressive family, on keys after the query, softmax, probs v)
// for the short sequences this model family trains on (S <= 2048, head sizes 16 / 32 / 64 / 128), forward and
// backward, bf16 operands, fp32 softmax.  The library attention the bracket used before spends 58 / 161 us
// (forward / backward) on the 67 / 134 MB of a BASELINE config-3 layer; these kernels read q, k, v in the
// (B, S, heads * d) layout the projections write (no head transposes) and never materialise the scores.
//
// All three kernels work on 32 x 32 score tiles with v_mfma_f32_32x32x16_bf16 and pick, per product, the operand
// order whose accumulator layout is directly the next product's B operand, so no score tile ever goes through LDS:
//   forward / dQ:  a wave owns 32 QUERIES.  S^T = K Q^T puts the query on the lane and the keys in the lane's
//                  registers: the softmax statistics are per lane, and P^T (dS^T) is the B operand of
//                  O^T = V^T P^T (dQ^T = K^T dS^T), whose A operand comes out of the transposing LDS read.
//   dK / dV:       a wave owns 32 KEYS.  S = Q K^T puts the key on the lane and the queries in registers: P (dS) is the
//                  B operand of dV^T = dO^T P (dK^T = Q^T dS).
// The staged 64-row operand tiles are read both row-wise (16-byte fragments) and through ds_read_b64_tr_b16; the row
// pitch (4 or 12 mod 16 granules) plus an XOR of the 16-byte granule index with bits 2-3 of the row makes both
// patterns bank-conflict free.  The backward recomputes the score tile in both kernels (7 products instead of 5)
// instead of accumulating dQ with atomics: every output is written once, in bf16, in fixed order.
//
// Mask semantics are the reference's: a masked score is -1e13 (not -inf), so a query row without any valid key
// attends uniformly to all S keys (forward), passes that uniform weight to dV and nothing to dQ / dK (backward,
// masked_fill's gradient).  Such rows are flagged by lse = +inf.
#include "common.hpp"

using namespace mivae;

namespace {

#define ATT_LDS __attribute__((address_space(3)))
typedef __bf16 att_bf16x4 __attribute__((__vector_size__(4 * sizeof(__bf16))));

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
__device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

constexpr int kChunk = 64;        // rows of the streamed operands staged per pass
constexpr int kMaxTiles = 64;     // S <= 32 * kMaxTiles
constexpr float kFill2 = -1.4426950408889634e13f;   // the reference's -1e13 fill in base-2 exponent units
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kSlack = 8.f;       // forward: exponent slack before the accumulators are rescaled

struct AttArgs {
  const unsigned short *q, *k, *v;    // element (b, s, head, d) at base + b * sb + s * sr + head * HD + d
  int64_t q_sb, k_sb, v_sb;
  int q_sr, k_sr, v_sr;
  const int64_t* mask;                // (B, S), != 0 keeps the key; may be null
  unsigned short* out;                // (B, S, heads * HD) contiguous
  float* lse;                         // (B, heads, S): log2-domain log-sum-exp of the scaled scores, +inf = no valid key
  const unsigned short* dout;         // (B, S, heads * HD) contiguous
  float* delta;                       // (B, heads, S): sum_d dout * out
  unsigned short *dq, *dk, *dv;       // element (b, s, head, d) at base + (b * S + s) * g_sr + head * HD + d
  int g_sr;                           //   (g_sr = heads * HD: contiguous; 3 * heads * HD: column blocks of one dq|dk|dv row)
  int S, heads, causal;
  float scale;
};

template <int HD>
struct Geo {
  static constexpr int HDP = HD < 32 ? 32 : HD;              // staged columns
  static constexpr int PITCH = HDP == 32 ? 32 : HDP + 32;    // elements: 4 / 12 / 20 granules = 4 or 12 (mod 16)
  static constexpr int NS = HD / 16;                         // 16-wide contraction slices over d
  static constexpr int NDB = HDP / 32;                       // 32-wide blocks of d
};

__device__ __forceinline__ int swz(int r, int c) { return (c ^ ((r >> 2) & 3)) << 3; }

// One operand chunk (kChunk rows of one head) on its way global -> registers -> swizzled LDS tile.  The loads of chunk
// i + 1 are issued before the products of chunk i and land in the tile after them, so their latency is covered by
// the products (the block is NW waves; every index below is a compile-time function of the thread id).
template <int HD, int NW>
struct Stage {
  static constexpr int GR = HD / 8;                                   // real 16-byte granules per row
  static constexpr int N = (kChunk * GR + 64 * NW - 1) / (64 * NW);   // granules per thread
  u16x8 r[N];
  __device__ __forceinline__ void load(const unsigned short* src, int sr, int row0, int S) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int idx = threadIdx.x + i * 64 * NW, row = idx / GR, c = idx % GR;
      const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      r[i] = z;
      if (idx < kChunk * GR && row0 + row < S) r[i] = *reinterpret_cast<const u16x8*>(src + (int64_t)(row0 + row) * sr + 8 * c);
    }
  }
  __device__ __forceinline__ void store(unsigned short* tile) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int idx = threadIdx.x + i * 64 * NW, row = idx / GR, c = idx % GR;
      if (idx < kChunk * GR) *reinterpret_cast<u16x8*>(tile + row * Geo<HD>::PITCH + swz(row, c)) = r[i];
    }
  }
};

// head size 16: the staged tiles are 32 columns wide, the upper 16 stay zero (written once, never staged)
template <int HD, int NW>
__device__ __forceinline__ void zero_pad_columns(unsigned short* tiles, int ntiles) {
  if (HD >= 32) return;
  const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int idx = threadIdx.x; idx < ntiles * kChunk * 2; idx += 64 * NW) {
    const int row = idx >> 1, c = 2 + (idx & 1);
    *reinterpret_cast<u16x8*>(tiles + row * Geo<HD>::PITCH + swz(row & (kChunk - 1), c)) = z;
  }
}

// A operand, rows of the tile: f[j] = tile[r][8 c + j]
template <int HD>
__device__ __forceinline__ bf16x8 row_frag(const unsigned short* tile, int r, int c) {
  return *reinterpret_cast<const bf16x8*>(tile + r * Geo<HD>::PITCH + swz(r, c));
}

// A operand, columns of the tile (k = rows): f[j] = tile[row0 + 4 hh + (j & 3) + 8 (j >> 2)][32 db + r32], the row
// order in which a 32x32 accumulator's registers 8 s .. 8 s + 7 (row0 = 16 s) hold their rows
template <int HD>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned short* tile, int row0, int db, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int r0 = row0 + 4 * (g >> 1) + q, r1 = r0 + 8;
  const int c = 4 * db + 2 * (g & 1) + (p >> 1), e = 4 * (p & 1);
  const unsigned short* a0 = tile + r0 * Geo<HD>::PITCH + swz(r0, c) + e;
  const unsigned short* a1 = tile + r1 * Geo<HD>::PITCH + swz(r1, c) + e;
  const att_bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ATT_LDS att_bf16x4*)(a0));
  const att_bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ATT_LDS att_bf16x4*)(a1));
  return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// B operand from global memory: row `row` of one head, f[j] = x[row][16 s + 8 hh + j] (zero for rows >= S)
template <int HD>
__device__ __forceinline__ void load_rows(bf16x8 (&f)[Geo<HD>::NS], const unsigned short* base, int sr, int row, int S,
                                          int hh) {
  const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < Geo<HD>::NS; ++s) {
    u16x8 v = z;
    if (row < S) v = *reinterpret_cast<const u16x8*>(base + (int64_t)row * sr + 16 * s + 8 * hh);
    f[s] = __builtin_bit_cast(bf16x8, v);
  }
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int s) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (bf16_t)x[8 * s + j];
  return f;
}

// additive score bias of the keys in base-2 exponent units: 0 = keep, the reference's fill for a masked key, -inf for
// the tile padding beyond S (kbias has 32 * ceil(S / 32) entries)
template <int NW>
__device__ __forceinline__ void stage_keybias(float* kbias, const int64_t* mask, int64_t b, int S) {
  const int n = (S + 31) & ~31;
  for (int key = threadIdx.x; key < n; key += 64 * NW)
    kbias[key] = key >= S ? -INFINITY : (mask == nullptr || mask[b * S + key] != 0) ? 0.f : kFill2;
}

// rows 4 a .. 4 a + 3 of accumulator block `x` (d = 32 db + 8 a + 4 hh ..) of the wave's row `row` -> bf16 (B, S, heads * HD)
template <int HD>
__device__ __forceinline__ void store_rows(unsigned short* dst, const f32x16 (&x)[Geo<HD>::NDB], float mul, int hh) {
#pragma unroll
  for (int db = 0; db < Geo<HD>::NDB; ++db)
#pragma unroll
    for (int a4 = 0; a4 < 4; ++a4) {
      const int d = 32 * db + 8 * a4 + 4 * hh;
      if (d < HD) {
        const u16x4 o = {f32_to_bf16(x[db][4 * a4] * mul), f32_to_bf16(x[db][4 * a4 + 1] * mul),
                         f32_to_bf16(x[db][4 * a4 + 2] * mul), f32_to_bf16(x[db][4 * a4 + 3] * mul)};
        *reinterpret_cast<u16x4*>(dst + d) = o;
      }
    }
}

#define ATT_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0)

// Issue order of a run of N products whose A operands each take DS LDS reads: the compiler's own order is
// read, wait, multiply per product, which exposes the whole LDS latency N times (a wave alone on its SIMD in the
// hd-128 dK / dV kernel has nothing else to run).  Keep the reads two products ahead of the matrix core instead.
template <int N, int DS>
__device__ __forceinline__ void products_pipelined() {
  if (N > 2) {
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * DS, 0);
#pragma unroll
    for (int i = 0; i < N - 2; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, DS, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
  }
}

// ---- forward: a wave owns 32 queries and walks the keys ---------------------------------------------------------
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW, 2) void k_att_fwd(AttArgs a) {
  using G = Geo<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned short* kt = lds;
  unsigned short* vt = lds + kChunk * G::PITCH;
  float* kbias = reinterpret_cast<float*>(lds + 2 * kChunk * G::PITCH);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r32 = lane & 31, hh = lane >> 5;
  const int64_t b = blockIdx.z;
  const int head = blockIdx.y, S = a.S, W = a.heads * HD;
  const int qrow = (blockIdx.x * NW + wave) * 32 + r32;
  const unsigned short* kbase = a.k + b * a.k_sb + head * HD;
  const unsigned short* vbase = a.v + b * a.v_sb + head * HD;
  Stage<HD, NW> sk, sv;                 // every global load of the prologue is in flight before the first wait
  sk.load(kbase, a.k_sr, 0, S);
  sv.load(vbase, a.v_sr, 0, S);
  bf16x8 qf[G::NS];
  load_rows<HD>(qf, a.q + b * a.q_sb + head * HD, a.q_sr, qrow, S, hh);
  stage_keybias<NW>(kbias, a.mask, b, S);
  zero_pad_columns<HD, NW>(lds, 2);
  f32x16 o[G::NDB];
#pragma unroll
  for (int db = 0; db < G::NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const float c2 = a.scale * kLog2e;
  for (int kc = 0; kc < S; kc += kChunk) {
    __syncthreads();                    // the previous chunk has been consumed (first pass: the key biases are written)
    sk.store(kt);
    sv.store(vt);
    __syncthreads();
    if (kc + kChunk < S) {              // block-uniform: the next chunk travels while this one is multiplied
      sk.load(kbase, a.k_sr, kc + kChunk, S);
      sv.load(vbase, a.v_sr, kc + kChunk, S);
    }
#pragma unroll
    for (int t = 0; t < kChunk / 32; ++t) {
      const int key0 = kc + 32 * t;
      if (key0 >= S) break;             // block-uniform
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < G::NS; ++s) sc = ATT_MFMA(row_frag<HD>(kt, 32 * t + r32, 2 * s + hh), qf[s], sc);
      products_pipelined<G::NS, 1>();
      __builtin_amdgcn_sched_barrier(0);
      float tmax = -INFINITY;
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) {
        const f32x4 kb = *reinterpret_cast<const f32x4*>(kbias + key0 + 8 * a4 + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float sv = fmaf(sc[4 * a4 + i], c2, kb[i]);       // fill + score rounds to the fill exactly
          if (a.causal) sv = key0 + 8 * a4 + 4 * hh + i <= qrow ? sv : fminf(kb[i], kFill2);
          sc[4 * a4 + i] = sv;
          tmax = fmaxf(tmax, sv);
        }
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      // lazy rescaling: the reference exponent m only follows the running maximum when that has grown by more than
      // 2^kSlack (p stays below 2^kSlack, harmless in fp32 / bf16); most tiles then skip the rescaling of o and l
      if (__builtin_amdgcn_ballot_w64(tmax > m + kSlack) != 0) {       // wave-uniform
        const float mn = tmax > m + kSlack ? tmax : m;
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        l *= alpha;
        m = mn;
#pragma unroll
        for (int db = 0; db < G::NDB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      }
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(sc[r] - m);
        sc[r] = p;
        ps += p;
      }
      l += ps;
      const bf16x8 p0 = pack8(sc, 0), p1 = pack8(sc, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int db = 0; db < G::NDB; ++db) {
        o[db] = ATT_MFMA(tr_frag<HD>(vt, 32 * t, db, lane), p0, o[db]);
        o[db] = ATT_MFMA(tr_frag<HD>(vt, 32 * t + 16, db, lane), p1, o[db]);
      }
      products_pipelined<2 * G::NDB, 2>();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  l += __shfl_xor(l, 32, 64);           // the two half-waves hold disjoint keys of the same query
  const bool anykeep = m > 0.5f * kFill2;           // the reference exponent of a query without valid keys is the fill
  if (qrow < S) {
    store_rows<HD>(a.out + (b * S + qrow) * W + head * HD, o, 1.f / l, hh);
    if (a.lse && hh == 0) a.lse[(b * a.heads + head) * S + qrow] = anykeep ? m + __builtin_amdgcn_logf(l) : INFINITY;
  }
}

// ---- backward, dQ (and delta = rowsum(dO * O) for the dK / dV kernel): a wave owns 32 queries ----------------------
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW, 2) void k_att_bwd_dq(AttArgs a) {
  using G = Geo<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned short* kt = lds;
  unsigned short* vt = lds + kChunk * G::PITCH;
  float* kbias = reinterpret_cast<float*>(lds + 2 * kChunk * G::PITCH);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r32 = lane & 31, hh = lane >> 5;
  const int64_t b = blockIdx.z;
  const int head = blockIdx.y, S = a.S, W = a.heads * HD;
  const int qrow = (blockIdx.x * NW + wave) * 32 + r32;
  const unsigned short* kbase = a.k + b * a.k_sb + head * HD;
  const unsigned short* vbase = a.v + b * a.v_sb + head * HD;
  Stage<HD, NW> sk, sv;
  sk.load(kbase, a.k_sr, 0, S);
  sv.load(vbase, a.v_sr, 0, S);
  stage_keybias<NW>(kbias, a.mask, b, S);
  zero_pad_columns<HD, NW>(lds, 2);
  bf16x8 qf[G::NS], dof[G::NS];
  load_rows<HD>(qf, a.q + b * a.q_sb + head * HD, a.q_sr, qrow, S, hh);
  load_rows<HD>(dof, a.dout + b * S * W + head * HD, W, qrow, S, hh);
  float delta = 0.f;
  {
    bf16x8 of[G::NS];
    load_rows<HD>(of, a.out + b * S * W + head * HD, W, qrow, S, hh);
#pragma unroll
    for (int s = 0; s < G::NS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += (float)dof[s][j] * (float)of[s][j];
    delta += __shfl_xor(delta, 32, 64);
  }
  float lse2 = 0.f;
  if (qrow < S) {
    lse2 = a.lse[(b * a.heads + head) * S + qrow];
    if (hh == 0) a.delta[(b * a.heads + head) * S + qrow] = delta;
  }
  f32x16 dq[G::NDB];
#pragma unroll
  for (int db = 0; db < G::NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
  const float c2 = a.scale * kLog2e;
  for (int kc = 0; kc < S; kc += kChunk) {
    __syncthreads();
    sk.store(kt);
    sv.store(vt);
    __syncthreads();
    if (kc + kChunk < S) {              // block-uniform: the next chunk travels while this one is multiplied
      sk.load(kbase, a.k_sr, kc + kChunk, S);
      sv.load(vbase, a.v_sr, kc + kChunk, S);
    }
#pragma unroll
    for (int t = 0; t < kChunk / 32; ++t) {
      const int key0 = kc + 32 * t;
      if (key0 >= S) break;
      f32x16 sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < G::NS; ++s) {
        sc = ATT_MFMA(row_frag<HD>(kt, 32 * t + r32, 2 * s + hh), qf[s], sc);
        dp = ATT_MFMA(row_frag<HD>(vt, 32 * t + r32, 2 * s + hh), dof[s], dp);
      }
      products_pipelined<2 * G::NS, 1>();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) {
        const f32x4 kb = *reinterpret_cast<const f32x4*>(kbias + key0 + 8 * a4 + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * a4 + i;
          // a masked key has exponent fill - lse2 (p = 0).  A query without valid keys has lse2 = +inf: p = 0, and
          // masked_fill passes no gradient to its scores anyway.  The 1 / sqrt(d) of dS is applied at the store.
          float x = fmaf(sc[r], c2, kb[i] - lse2);
          if (a.causal) x = key0 + 8 * a4 + 4 * hh + i <= qrow ? x : -INFINITY;
          sc[r] = __builtin_amdgcn_exp2f(x) * (dp[r] - delta);
        }
      }
      const bf16x8 d0 = pack8(sc, 0), d1 = pack8(sc, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int db = 0; db < G::NDB; ++db) {
        dq[db] = ATT_MFMA(tr_frag<HD>(kt, 32 * t, db, lane), d0, dq[db]);
        dq[db] = ATT_MFMA(tr_frag<HD>(kt, 32 * t + 16, db, lane), d1, dq[db]);
      }
      products_pipelined<2 * G::NDB, 2>();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (qrow < S) store_rows<HD>(a.dq + (b * S + qrow) * a.g_sr + head * HD, dq, a.scale, hh);
}

// ---- backward, dK and dV: a wave owns 32 keys and walks the queries ------------------------------------------------
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW, HD == 128 ? 1 : 2) void k_att_bwd_dkv(AttArgs a) {
  using G = Geo<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned short* qt = lds;
  unsigned short* dt = lds + kChunk * G::PITCH;
  float* lses = reinterpret_cast<float*>(lds + 2 * kChunk * G::PITCH);   // [kChunk] lse2, [kChunk] delta
  float* dels = lses + kChunk;
  float* fuls = dels + kChunk;                         // 1 / S for a query without valid keys, else 0
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r32 = lane & 31, hh = lane >> 5;
  const int64_t b = blockIdx.z;
  const int head = blockIdx.y, S = a.S, W = a.heads * HD;
  const int key = (blockIdx.x * NW + wave) * 32 + r32;
  const bool kok = key < S;
  const bool keepk = kok && (a.mask == nullptr || a.mask[b * S + key] != 0);
  const float kbias = keepk ? 0.f : (kok ? kFill2 : -INFINITY);       // the lane's key: exponent bias
  bf16x8 kf[G::NS], vf[G::NS];
  load_rows<HD>(kf, a.k + b * a.k_sb + head * HD, a.k_sr, key, S, hh);
  load_rows<HD>(vf, a.v + b * a.v_sb + head * HD, a.v_sr, key, S, hh);
  const unsigned short* qbase = a.q + b * a.q_sb + head * HD;
  const unsigned short* dbase = a.dout + b * S * W + head * HD;
  const float* lse_g = a.lse + (b * a.heads + head) * S;
  const float* del_g = a.delta + (b * a.heads + head) * S;
  f32x16 dk[G::NDB], dv[G::NDB];
#pragma unroll
  for (int db = 0; db < G::NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
  const float c2 = a.scale * kLog2e;
  const float inv_s = 1.f / (float)S;
  Stage<HD, NW> sq, sd;
  float pl = 0.f, pd = 0.f;             // lse2 / delta of chunk row threadIdx.x (the first kChunk threads)
  auto load_chunk = [&](int qc) {
    sq.load(qbase, a.q_sr, qc, S);
    sd.load(dbase, W, qc, S);
    const bool ok = threadIdx.x < kChunk && qc + (int)threadIdx.x < S;
    pl = ok ? lse_g[qc + threadIdx.x] : 0.f;     // rows beyond S: q = dO = 0, p stays finite, every product is 0
    pd = ok ? del_g[qc + threadIdx.x] : 0.f;
  };
  load_chunk(0);
  zero_pad_columns<HD, NW>(lds, 2);
  for (int qc = 0; qc < S; qc += kChunk) {
    __syncthreads();
    sq.store(qt);
    sd.store(dt);
    if (threadIdx.x < kChunk) {
      lses[threadIdx.x] = pl;
      dels[threadIdx.x] = pd;
      fuls[threadIdx.x] = pl == INFINITY ? inv_s : 0.f;
    }
    __syncthreads();
    if (qc + kChunk < S) load_chunk(qc + kChunk);
#pragma unroll
    for (int t = 0; t < kChunk / 32; ++t) {
      const int q0 = qc + 32 * t;
      if (q0 >= S) break;
      f32x16 sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < G::NS; ++s) {
        sc = ATT_MFMA(row_frag<HD>(qt, 32 * t + r32, 2 * s + hh), kf[s], sc);
        dp = ATT_MFMA(row_frag<HD>(dt, 32 * t + r32, 2 * s + hh), vf[s], dp);
      }
      products_pipelined<2 * G::NS, 1>();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) {
        const f32x4 L = *reinterpret_cast<const f32x4*>(lses + 32 * t + 8 * a4 + 4 * hh);
        const f32x4 D = *reinterpret_cast<const f32x4*>(dels + 32 * t + 8 * a4 + 4 * hh);
        const f32x4 F = *reinterpret_cast<const f32x4*>(fuls + 32 * t + 8 * a4 + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * a4 + i;
          float x = fmaf(sc[r], c2, kbias - L[i]);         // masked key: fill - lse2 -> p = 0
          if (a.causal) x = key <= q0 + 8 * a4 + 4 * hh + i ? x : -INFINITY;
          const float p = __builtin_amdgcn_exp2f(x);
          dp[r] = p * (dp[r] - D[i]);                      // dS without its 1 / sqrt(d): applied at the store
          sc[r] = p + F[i];                                // query without a valid key: p = 0 + uniform 1 / S
        }
      }
      const bf16x8 p0 = pack8(sc, 0), p1 = pack8(sc, 1), d0 = pack8(dp, 0), d1 = pack8(dp, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int db = 0; db < G::NDB; ++db) {
        dv[db] = ATT_MFMA(tr_frag<HD>(dt, 32 * t, db, lane), p0, dv[db]);
        dv[db] = ATT_MFMA(tr_frag<HD>(dt, 32 * t + 16, db, lane), p1, dv[db]);
        dk[db] = ATT_MFMA(tr_frag<HD>(qt, 32 * t, db, lane), d0, dk[db]);
        dk[db] = ATT_MFMA(tr_frag<HD>(qt, 32 * t + 16, db, lane), d1, dk[db]);
      }
      products_pipelined<4 * G::NDB, 2>();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (kok) {
    store_rows<HD>(a.dk + (b * S + key) * a.g_sr + head * HD, dk, a.scale, hh);
    store_rows<HD>(a.dv + (b * S + key) * a.g_sr + head * HD, dv, 1.f, hh);
  }
}

// ---- backward for S <= 64 (head sizes up to 64): one launch, one staging --------------------------------------------
// Q, K, V and dO of a (batch, head) fit one LDS tile each; the block is 2 NW waves: waves 0 .. NW - 1 play the dQ
// kernel's part for queries 32 w .. 32 w + 31 while waves NW .. 2 NW - 1 play the dK / dV kernel's part for the same
// keys, side by side after the single barrier (delta = rowsum(dO * O) is formed per row in the prologue and goes through
// LDS, not HBM).  At BASELINE config 2 (S = 64, d = 16) the two-kernel backward is launch- and latency-bound (8 + 9 us
// for 2 MB).
template <int HD, int NW>
__global__ __launch_bounds__(128 * NW, 2) void k_att_bwd_small(AttArgs a) {
  using G = Geo<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int T = kChunk * G::PITCH;
  unsigned short *qt = lds, *kt = lds + T, *vt = lds + 2 * T, *dt = lds + 3 * T;
  float* kb = reinterpret_cast<float*>(lds + 4 * T);   // [kChunk] key bias, lse2, delta, uniform weight
  float* lses = kb + kChunk;
  float* dels = lses + kChunk;
  float* fuls = dels + kChunk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r32 = lane & 31, hh = lane >> 5;
  const int64_t b = blockIdx.z;
  const int head = blockIdx.y, S = a.S, W = a.heads * HD;
  const bool dq_role = wave < NW;       // wave-uniform
  const int row = (dq_role ? wave : wave - NW) * 32 + r32;      // the lane's query (dQ part) or key (dK / dV part)
  Stage<HD, 2 * NW> sq, sk, sv, sd;
  sq.load(a.q + b * a.q_sb + head * HD, a.q_sr, 0, S);
  sk.load(a.k + b * a.k_sb + head * HD, a.k_sr, 0, S);
  sv.load(a.v + b * a.v_sb + head * HD, a.v_sr, 0, S);
  sd.load(a.dout + b * S * W + head * HD, W, 0, S);
  if (threadIdx.x < kChunk) {           // 128 * NW >= kChunk threads: the per-row scalars
    const int r = threadIdx.x;
    const bool ok = r < S;
    float delta = 0.f;
    if (ok) {
      const unsigned short* orow = a.out + (b * S + r) * W + head * HD;
      const unsigned short* drow = a.dout + (b * S + r) * W + head * HD;
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) {
        const u16x8 ov = *reinterpret_cast<const u16x8*>(orow + 8 * c), dv8 = *reinterpret_cast<const u16x8*>(drow + 8 * c);
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += bf16_to_f32(ov[j]) * bf16_to_f32(dv8[j]);
      }
    }
    const float l = ok ? a.lse[(b * a.heads + head) * S + r] : 0.f;
    lses[r] = l;
    dels[r] = delta;                    // rows beyond S: 0
    fuls[r] = l == INFINITY ? 1.f / (float)S : 0.f;
    kb[r] = !ok ? -INFINITY : (a.mask == nullptr || a.mask[b * S + r] != 0) ? 0.f : kFill2;
  }
  zero_pad_columns<HD, 2 * NW>(lds, 4);
  sq.store(qt);
  sk.store(kt);
  sv.store(vt);
  sd.store(dt);
  __syncthreads();
  const float c2 = a.scale * kLog2e;
  if (dq_role) {                        // ---- dQ of queries 32 w .. (k_att_bwd_dq's tile body)
    bf16x8 qf[G::NS], dof[G::NS];
#pragma unroll
    for (int s = 0; s < G::NS; ++s) {
      qf[s] = row_frag<HD>(qt, row, 2 * s + hh);
      dof[s] = row_frag<HD>(dt, row, 2 * s + hh);
    }
    const float delta = dels[row], lse2 = lses[row];
    f32x16 dq[G::NDB];
#pragma unroll
    for (int db = 0; db < G::NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
#pragma unroll
    for (int t = 0; t < NW; ++t) {
      f32x16 sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < G::NS; ++s) {
        sc = ATT_MFMA(row_frag<HD>(kt, 32 * t + r32, 2 * s + hh), qf[s], sc);
        dp = ATT_MFMA(row_frag<HD>(vt, 32 * t + r32, 2 * s + hh), dof[s], dp);
      }
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) {
        const f32x4 kbv = *reinterpret_cast<const f32x4*>(kb + 32 * t + 8 * a4 + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * a4 + i;
          float x = fmaf(sc[r], c2, kbv[i] - lse2);
          if (a.causal) x = 32 * t + 8 * a4 + 4 * hh + i <= row ? x : -INFINITY;
          sc[r] = __builtin_amdgcn_exp2f(x) * (dp[r] - delta);
        }
      }
      const bf16x8 d0 = pack8(sc, 0), d1 = pack8(sc, 1);
#pragma unroll
      for (int db = 0; db < G::NDB; ++db) {
        dq[db] = ATT_MFMA(tr_frag<HD>(kt, 32 * t, db, lane), d0, dq[db]);
        dq[db] = ATT_MFMA(tr_frag<HD>(kt, 32 * t + 16, db, lane), d1, dq[db]);
      }
    }
    if (row < S) store_rows<HD>(a.dq + (b * S + row) * a.g_sr + head * HD, dq, a.scale, hh);
  } else {                              // ---- dK, dV of keys 32 w .. (k_att_bwd_dkv's tile body)
    bf16x8 kf[G::NS], vf[G::NS];
#pragma unroll
    for (int s = 0; s < G::NS; ++s) {
      kf[s] = row_frag<HD>(kt, row, 2 * s + hh);
      vf[s] = row_frag<HD>(vt, row, 2 * s + hh);
    }
    const float kbias = kb[row];
    f32x16 dk[G::NDB], dv[G::NDB];
#pragma unroll
    for (int db = 0; db < G::NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
#pragma unroll
    for (int t = 0; t < NW; ++t) {
      f32x16 sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < G::NS; ++s) {
        sc = ATT_MFMA(row_frag<HD>(qt, 32 * t + r32, 2 * s + hh), kf[s], sc);
        dp = ATT_MFMA(row_frag<HD>(dt, 32 * t + r32, 2 * s + hh), vf[s], dp);
      }
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) {
        const f32x4 L = *reinterpret_cast<const f32x4*>(lses + 32 * t + 8 * a4 + 4 * hh);
        const f32x4 D = *reinterpret_cast<const f32x4*>(dels + 32 * t + 8 * a4 + 4 * hh);
        const f32x4 F = *reinterpret_cast<const f32x4*>(fuls + 32 * t + 8 * a4 + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * a4 + i;
          float x = fmaf(sc[r], c2, kbias - L[i]);
          if (a.causal) x = row <= 32 * t + 8 * a4 + 4 * hh + i ? x : -INFINITY;
          const float p = __builtin_amdgcn_exp2f(x);
          dp[r] = p * (dp[r] - D[i]);
          sc[r] = p + F[i];
        }
      }
      const bf16x8 p0 = pack8(sc, 0), p1 = pack8(sc, 1), d0 = pack8(dp, 0), d1 = pack8(dp, 1);
#pragma unroll
      for (int db = 0; db < G::NDB; ++db) {
        dv[db] = ATT_MFMA(tr_frag<HD>(dt, 32 * t, db, lane), p0, dv[db]);
        dv[db] = ATT_MFMA(tr_frag<HD>(dt, 32 * t + 16, db, lane), p1, dv[db]);
        dk[db] = ATT_MFMA(tr_frag<HD>(qt, 32 * t, db, lane), d0, dk[db]);
        dk[db] = ATT_MFMA(tr_frag<HD>(qt, 32 * t + 16, db, lane), d1, dk[db]);
      }
    }
    if (row < S) {
      store_rows<HD>(a.dk + (b * S + row) * a.g_sr + head * HD, dk, a.scale, hh);
      store_rows<HD>(a.dv + (b * S + row) * a.g_sr + head * HD, dv, 1.f, hh);
    }
  }
}

template <int HD>
size_t lds_bytes(int S) {     // two operand tiles + the key biases (forward, dQ) or the three per-query rows (dK / dV)
  const size_t tail = (size_t)((S + 31) & ~31) * 4;
  return (size_t)2 * kChunk * Geo<HD>::PITCH * 2 + (tail > 3 * kChunk * 4 ? tail : 3 * kChunk * 4);
}

template <int HD, int NW>
void launch_nw(const AttArgs& a, int64_t B, int which, hipStream_t st) {
  const int tiles = (a.S + 31) / 32;
  const dim3 grid((tiles + NW - 1) / NW, a.heads, (unsigned)B), block(64 * NW);
  const size_t lds = lds_bytes<HD>(a.S);
  if (which == 0) hipLaunchKernelGGL((k_att_fwd<HD, NW>), grid, block, lds, st, a);
  else if (which == 1) hipLaunchKernelGGL((k_att_bwd_dq<HD, NW>), grid, block, lds, st, a);
  else hipLaunchKernelGGL((k_att_bwd_dkv<HD, NW>), grid, block, lds, st, a);
}

template <int HD, int NW>
void launch_small(const AttArgs& a, int64_t B, hipStream_t st) {
  const size_t lds = (size_t)4 * kChunk * Geo<HD>::PITCH * 2 + 4 * kChunk * 4;
  hipLaunchKernelGGL((k_att_bwd_small<HD, NW>), dim3(1, a.heads, (unsigned)B), dim3(128 * NW), lds, st, a);
}

// S <= 64 and head size <= 64: the one-launch backward
bool dispatch_small(const AttArgs& a, int64_t B, int hd, hipStream_t st) {
  if (a.S > kChunk || hd > 64) return false;
  const bool one = a.S <= 32;
  switch (hd) {
    case 16: one ? launch_small<16, 1>(a, B, st) : launch_small<16, 2>(a, B, st); break;
    case 32: one ? launch_small<32, 1>(a, B, st) : launch_small<32, 2>(a, B, st); break;
    case 64: one ? launch_small<64, 1>(a, B, st) : launch_small<64, 2>(a, B, st); break;
    default: return false;
  }
  return true;
}

template <int HD>
void launch(const AttArgs& a, int64_t B, int which, hipStream_t st) {
  const int tiles = (a.S + 31) / 32;       // a block is 1, 2 or 4 waves of 32 rows
  if (tiles == 1) launch_nw<HD, 1>(a, B, which, st);
  else if (tiles == 2) launch_nw<HD, 2>(a, B, which, st);
  else launch_nw<HD, 4>(a, B, which, st);
}

int dispatch(const AttArgs& a, int64_t B, int hd, int which, hipStream_t st) {
  switch (hd) {
    case 16: launch<16>(a, B, which, st); break;
    case 32: launch<32>(a, B, which, st); break;
    case 64: launch<64>(a, B, which, st); break;
    case 128: launch<128>(a, B, which, st); break;
    default: return MIVAE_EUNSUPPORTED;
  }
  return MIVAE_OK;
}

int fill_args(AttArgs& a, const char* name, const void* q, const void* k, const void* v, const int64_t* mask, int64_t B,
              int64_t S, int64_t heads, int64_t hd, const int64_t (&strides)[6], int causal, float scale) {
  MIVAE_REQUIRE(hd == 16 || hd == 32 || hd == 64 || hd == 128, "%s: head size %lld not in {16, 32, 64, 128}", name,
                (long long)hd);
  MIVAE_REQUIRE(S >= 1 && S <= 32 * kMaxTiles, "%s: S = %lld outside 1 .. %d", name, (long long)S, 32 * kMaxTiles);
  MIVAE_REQUIRE(B >= 1 && B <= 65535 && heads >= 1 && heads <= 65535, "%s: B = %lld / heads = %lld outside 1 .. 65535",
                name, (long long)B, (long long)heads);
  MIVAE_REQUIRE(q && k && v, "%s: null operand", name);
  for (int i = 0; i < 6; ++i)
    MIVAE_REQUIRE(strides[i] >= 0 && strides[i] % 8 == 0 && strides[i] < (1ll << 40), "%s: stride %d = %lld not a multiple of 8",
                  name, i, (long long)strides[i]);
  MIVAE_REQUIRE(strides[1] >= heads * hd && strides[3] >= heads * hd && strides[5] >= heads * hd &&
                    strides[1] < (1ll << 30) && strides[3] < (1ll << 30) && strides[5] < (1ll << 30),
                "%s: row stride smaller than heads * head size", name);
  MIVAE_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v), "%s: q / k / v must be 16-byte aligned", name);
  a.q = static_cast<const unsigned short*>(q);
  a.k = static_cast<const unsigned short*>(k);
  a.v = static_cast<const unsigned short*>(v);
  a.q_sb = strides[0]; a.q_sr = (int)strides[1];
  a.k_sb = strides[2]; a.k_sr = (int)strides[3];
  a.v_sb = strides[4]; a.v_sr = (int)strides[5];
  a.mask = mask;
  a.S = (int)S; a.heads = (int)heads; a.causal = causal ? 1 : 0;
  a.scale = scale;
  return MIVAE_OK;
}

}  // namespace

extern "C" {

int mivae_attention_supported(int64_t head_size, int64_t S) {
  return (head_size == 16 || head_size == 32 || head_size == 64 || head_size == 128) && S >= 1 && S <= 32 * kMaxTiles;
}

int mivae_attention_fwd(const void* q, const void* k, const void* v, const int64_t* mask, void* out, float* lse,
                        int64_t B, int64_t S, int64_t heads, int64_t head_size, int64_t q_batch_stride,
                        int64_t q_row_stride, int64_t k_batch_stride, int64_t k_row_stride, int64_t v_batch_stride,
                        int64_t v_row_stride, int causal, float scale, void* stream) {
  AttArgs a = {};
  const int64_t strides[6] = {q_batch_stride, q_row_stride, k_batch_stride, k_row_stride, v_batch_stride, v_row_stride};
  const int rc = fill_args(a, "mivae_attention_fwd", q, k, v, mask, B, S, heads, head_size, strides, causal, scale);
  if (rc != MIVAE_OK) return rc;
  MIVAE_REQUIRE(out && aligned16(out), "mivae_attention_fwd: out must be a 16-byte aligned buffer");
  a.out = static_cast<unsigned short*>(out);
  a.lse = lse;
  dispatch(a, B, (int)head_size, 0, S_(stream));
  MIVAE_CHECK_LAUNCH("mivae_attention_fwd");
  return MIVAE_OK;
}

int mivae_attention_bwd(const void* q, const void* k, const void* v, const int64_t* mask, const void* out,
                        const void* dout, const float* lse, float* delta, void* dq, void* dk, void* dv, int64_t B,
                        int64_t S, int64_t heads, int64_t head_size, int64_t q_batch_stride, int64_t q_row_stride,
                        int64_t k_batch_stride, int64_t k_row_stride, int64_t v_batch_stride, int64_t v_row_stride,
                        int64_t grad_row_stride, int causal, float scale, void* stream) {
  AttArgs a = {};
  const int64_t strides[6] = {q_batch_stride, q_row_stride, k_batch_stride, k_row_stride, v_batch_stride, v_row_stride};
  const int rc = fill_args(a, "mivae_attention_bwd", q, k, v, mask, B, S, heads, head_size, strides, causal, scale);
  if (rc != MIVAE_OK) return rc;
  MIVAE_REQUIRE(out && dout && lse && delta && dq && dk && dv, "mivae_attention_bwd: null operand");
  MIVAE_REQUIRE(aligned16(out) && aligned16(dout) && aligned16(dq) && aligned16(dk) && aligned16(dv),
                "mivae_attention_bwd: out / dout / dq / dk / dv must be 16-byte aligned");
  a.out = static_cast<unsigned short*>(const_cast<void*>(out));
  a.dout = static_cast<const unsigned short*>(dout);
  a.lse = const_cast<float*>(lse);
  a.delta = delta;
  a.dq = static_cast<unsigned short*>(dq);
  a.dk = static_cast<unsigned short*>(dk);
  a.dv = static_cast<unsigned short*>(dv);
  MIVAE_REQUIRE(grad_row_stride >= heads * head_size && grad_row_stride % 8 == 0 && grad_row_stride < (1ll << 30),
                "mivae_attention_bwd: gradient row stride %lld must be a multiple of 8 and >= heads * head size",
                (long long)grad_row_stride);
  a.g_sr = (int)grad_row_stride;
  if (dispatch_small(a, B, (int)head_size, S_(stream))) {
    MIVAE_CHECK_LAUNCH("mivae_attention_bwd (one launch)");
    return MIVAE_OK;
  }
  dispatch(a, B, (int)head_size, 1, S_(stream));
  MIVAE_CHECK_LAUNCH("mivae_attention_bwd (dq)");
  dispatch(a, B, (int)head_size, 2, S_(stream));
  MIVAE_CHECK_LAUNCH("mivae_attention_bwd (dk, dv)");
  return MIVAE_OK;
}

}  // extern "C"
