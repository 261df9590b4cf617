/* libmivae -- C ABI of the MI355X (gfx950) latent hot path of Nemexur/nonauto-lm.
 *
 * Boundary contract:
 *  - plain pointers and sizes only; no torch types.  All tensor pointers are DEVICE pointers to
 *    contiguous row-major buffers that the caller owns (borrowed for the duration of the launch);
 *    outputs are pre-allocated by the caller.  The library never allocates or frees tensor memory
 *    and keeps no mutable global state besides a thread-local error string.
 *  - every entry point enqueues its kernels on `stream` (a hipStream_t passed as void*) and returns
 *    immediately: 0 = MIVAE_OK, <0 = error (mivae_last_error() has the text).  No device sync.
 *  - `dtype`: 0 = float32, 1 = bfloat16 latents.  Log-dets, log-probs, KL and ELBO are always fp32.
 *  - masks are int64 0/1 `(B, S)` exactly as vae_lm/nn/utils.py:12-14 produces them, or NULL.
 *    `mask_div` = posterior samples n when a `(B, S)` mask addresses `(B*n, S)` rows (row b*n+j uses
 *    mask row b, the einops `repeat "b s -> (b n) s"` of posterior.py:125-127), else 1.
 *  - a 2-D `(B, D)` latent (autoregressive model) is passed as S = 1, mask = NULL.
 *
 * Citations are file:line under the reference tree (vae_lm/...).
 */
#ifndef MIVAE_H
#define MIVAE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIVAE_OK 0
#define MIVAE_EINVAL (-1)
#define MIVAE_ELAUNCH (-2)
#define MIVAE_EUNSUPPORTED (-3)
#define MIVAE_F32 0
#define MIVAE_BF16 1

/* activation ids of the fused coupling MLP (subset of nn/activation.py:59-77) */
#define MIVAE_ACT_LINEAR 0
#define MIVAE_ACT_GELU 1 /* erf GELU, nn/activation.py:9-16 */
#define MIVAE_ACT_SWISH 2 /* nn/activation.py:21-23 */
#define MIVAE_ACT_RELU 3
#define MIVAE_ACT_TANH 4
#define MIVAE_ACT_SIGMOID 5

int mivae_version(void);
/* thread-local text of the last error on the calling thread (replaces the reference's Python
 * exceptions, training/trainers/trainer.py:110-113) */
const char* mivae_last_error(void);

/* ---- integer helpers (bit exact) ------------------------------------------------------------ */
/* mask[i] = tokens[i] != 0.  Replaces get_tokens_mask, nn/utils.py:12-14. */
int mivae_tokens_mask(const int64_t* tokens, int64_t* mask, int64_t n, void* stream);
/* The same, plus the mask as fp32 in mask_f32 (n): the per-token weights of LabelSmoothingNLL
 * (models/base/losses.py:128-164) without a cast launch. */
int mivae_tokens_mask_f32(const int64_t* tokens, int64_t* mask, float* mask_f32, int64_t n, void* stream);
/* sigma head of the posterior in one launch: sigma = Hardtanh(Softplus(pre), lo, hi) elementwise over n values of the
 * latent dtype (models/nonauto_lm/posteriors/posterior.py:23-27, auto posterior.py:23-27: Softplus(beta 1, threshold
 * 20) + Hardtanh(1e-4, 5)); backward: d pre = d sigma * sigmoid(pre) where lo < softplus(pre) < hi, else 0. */
int mivae_sigma_head_fwd(const void* pre, void* sigma, int64_t n, float lo, float hi, int dtype, void* stream);
int mivae_sigma_head_bwd(const void* pre, const void* dsigma, void* dpre, int64_t n, float lo, float hi, int dtype,
                         void* stream);
/* Entry of the transformer encoder / decoder in one launch: y = (x + pe[s]) * mask[b, s] for x (B, S, D) of the latent
 * dtype, pe the fp32 sinusoid table (>= S rows of D), mask (B, S) int64 -- PositionalEncoding.forward
 * (vae_lm/nn/transformer.py:199-202) followed by the mask multiply of the encoder / decoder.  backward = 1: y = x * mask
 * (the input gradient; pe may be NULL). */
int mivae_posenc_mask(const void* x, const float* pe, const int64_t* mask, void* y, int64_t B, int64_t S, int64_t D,
                      int backward, int dtype, void* stream);
/* Additive attention bias from the token mask, built once per encoder / decoder call instead of per layer: 0 where
 * the key may be attended, -inf at padded keys (and at k > q when causal).  out (B, 1, 1, S), or (B, 1, S, S) when
 * causal, f32 or bf16.  Replaces the mask.ne(0) / tril / masked_fill chain of MultiHeadAttention.forward,
 * vae_lm/nn/transformer.py:75-89 (bracket of the path, row f). */
int mivae_attn_bias(const int64_t* mask, void* out, int64_t B, int64_t S, int causal, int dtype, void* stream);
/* Masked multi-head self-attention, forward and backward, on bf16 operands with fp32 softmax: replaces
 * MultiHeadAttention.forward's score / masked_fill / softmax / matmul chain, nn/transformer.py:62-117 (scores =
 * q k^T / sqrt(d), masked_fill(-1e13) where mask == 0 or, with causal != 0, key > query, softmax over keys, probs v).
 * q, k, v: element (b, s, head, d) at base + b * batch_stride + s * row_stride + head * head_size + d (bf16; the
 * (B, S, heads * head_size) outputs of the projections, no head transposes); strides in elements, multiples of 8.
 * mask (B, S) int64 or NULL; out / dout (B, S, heads * head_size) bf16 contiguous; dq / dk / dv rows of heads *
 * head_size bf16 every grad_row_stride elements (heads * head_size: contiguous; 3 x that with dk = dq + heads *
 * head_size, dv = dk + heads * head_size: one packed dq|dk|dv row per token, the K-concatenated operand of a fused
 * q|k|v projection's input gradient); lse and delta
 * (B, heads, S) fp32 (lse may be NULL in forward-only use; it holds log2-domain log-sum-exp, +inf for a query
 * without any valid key, which attends uniformly to all S keys like the reference's finite fill).
 * head_size in {16, 32, 64, 128}, 1 <= S <= 2048 (mivae_attention_supported). */
int mivae_attention_supported(int64_t head_size, int64_t S);
int mivae_attention_fwd(const void* q, const void* k, const void* v, const int64_t* mask, void* out, float* lse,
                        int64_t B, int64_t S, int64_t heads, int64_t head_size, int64_t q_batch_stride,
                        int64_t q_row_stride, int64_t k_batch_stride, int64_t k_row_stride, int64_t v_batch_stride,
                        int64_t v_row_stride, int causal, float scale, void* stream);
int mivae_attention_bwd(const void* q, const void* k, const void* v, const int64_t* mask, const void* out,
                        const void* dout, const float* lse, float* delta, void* dq, void* dk, void* dv, int64_t B,
                        int64_t S, int64_t heads, int64_t head_size, int64_t q_batch_stride, int64_t q_row_stride,
                        int64_t k_batch_stride, int64_t k_row_stride, int64_t v_batch_stride, int64_t v_row_stride,
                        int64_t grad_row_stride, int causal, float scale, void* stream);
/* last_idx[b] = sum_s mask[b,s] - 1; ctx[b,:] = out[b, last_idx[b], :] (index -1 wraps like torch).
 * Replaces Encoder.forward's gather, models/base/encoders/encoder.py:35-37.  last_idx may be NULL. */
int mivae_gather_last(const void* out, const int64_t* mask, void* ctx, int64_t* last_idx, int64_t B,
                      int64_t S, int64_t E, int dtype, void* stream);
/* count[bn] = sum_s mask[bn / mask_div, s]  (the integer token count of actnorm.py:35-38 and
 * invertible_flows.py:30-31). */
int mivae_mask_count(const int64_t* mask, int64_t* count, int64_t Bn, int64_t mask_div, int64_t S,
                     void* stream);
/* out[b] = sum_s tok[b, s] (+ add[b]); one wave per segment, fixed order (the reduction
 * `einsum("b...->b")` of posterior.py:170-172 / prior.py:156-158 over tokens). */
int mivae_segment_sum(const float* tok, const float* add, float* out, int64_t nseg, int64_t S,
                      void* stream);

/* ---- posterior: reparameterised sample + log q ---------------------------------------------- */
/* z[(b n), s, :] = (mu[b,s,:] + eps[(b n),s,:] * sigma[b,s,:]) * mask[b,s]  and
 * tok_logq[(b n), s] = mask * sum_d -1/2((z-mu)^2/sigma^2 + 2 log sigma + log 2pi).
 * Replaces Posterior.sample (nonauto posterior.py:115-133, auto :107-119) and
 * Posterior.log_probability (nonauto :157-172, auto :141-156).  eps NULL = zeros (random=False). */
int mivae_reparam_logq_fwd(const void* mu, const void* sigma, const void* eps, const int64_t* mask,
                           void* z, float* tok_logq, int64_t B, int64_t n, int64_t S, int64_t D,
                           int dtype, void* stream);
/* backward of the pair above w.r.t. mu and sigma (autograd of posterior.py:115-172); dz (B n,S,D)
 * and dlogq (B n) may each be NULL. */
int mivae_reparam_logq_bwd(const void* dz, const float* dlogq, const void* sigma, const void* eps,
                           const int64_t* mask, void* dmu, void* dsigma, int64_t B, int64_t n,
                           int64_t S, int64_t D, int dtype, void* stream);

/* The same pair with eps drawn INSIDE the kernels (Posterior.sample draws it with base_dist.sample, nonauto
 * posterior.py:115-118, auto :107-110): Philox4x32-10 keyed by the seed, counter = (element / 4, draw offset), two
 * Box-Muller pairs per counter -- eps is a pure function of (seed, offset, element) and never touches HBM.
 * rng_state: device int64 {seed, offset}; the forward copies it to rng_used (device int64[2], to be kept for the
 * backward), draws with it and advances rng_state's offset by one, all stream ordered, so a captured HIP graph draws
 * fresh noise at every replay.  The backward regenerates eps from rng_used.  mivae_philox_normal writes the first
 * `count` fp32 values of the draw recorded in rng_used (what the kernels used; for tests). */
int mivae_reparam_logq_fwd_rng(const void* mu, const void* sigma, int64_t* rng_state, int64_t* rng_used,
                               const int64_t* mask, void* z, float* tok_logq, int64_t B, int64_t n, int64_t S,
                               int64_t D, int dtype, void* stream);
int mivae_reparam_logq_bwd_rng(const void* dz, const float* dlogq, const void* sigma, const int64_t* rng_used,
                               const int64_t* mask, void* dmu, void* dsigma, int64_t B, int64_t n, int64_t S,
                               int64_t D, int dtype, void* stream);
int mivae_philox_normal(const int64_t* rng_used, float* out, int64_t count, void* stream);

/* a3 tail + a4 + a5 in ONE launch each way (posterior.py:23-27 sigma head Hardtanh(Softplus(.), lo, hi); :86-133 sample;
 * :135-172 log q).  sigma_or_pre (B, S, D) is sigma itself (sigma_is_pre = 0) or the sigma head's pre-activation (1; sigma
 * is then also written to sigma_out when that is not NULL).  Noise: eps ((B n), S, D) injected; or drawn in-kernel from
 * rng_state = device int64[2] {seed, offset}: the launch copies the state to rng_used (int64[2], for the backward) and
 * does not advance it -- follow up with mivae_rng_advance(rng_state), or hand rng_state to mivae_latent_tail_fwd of the
 * same forward (the training step: no launch spent on the generator); both NULL: eps = 0.  tok_logq ((B n) S) fp32 =
 * per-token log q (sum per sentence: mivae_segment_sum / mivae_latent_tail_fwd).
 * Backward: d mu, and d sigma or d pre, (B, S, D); dlogq per sentence (B n) or, dlogq_per_token = 1, per token. */
int mivae_rng_advance(int64_t* rng_state, void* stream);
int mivae_posterior_fwd(const void* mu, const void* sigma_or_pre, int sigma_is_pre, float lo, float hi, const void* eps,
                        const int64_t* rng_state, int64_t* rng_used, const int64_t* mask, void* z, void* sigma_out,
                        float* tok_logq, int64_t B, int64_t n, int64_t S, int64_t D, int dtype, void* stream);
int mivae_posterior_bwd(const void* dz, const float* dlogq, int dlogq_per_token, const void* sigma_or_pre, int sigma_is_pre,
                        float lo, float hi, const void* eps, const int64_t* rng_used, const int64_t* mask, void* dmu,
                        void* dsigma, int64_t B, int64_t n, int64_t S, int64_t D, int dtype, void* stream);

/* ---- base distribution ----------------------------------------------------------------------- */
/* tok[bn, s] = mask * sum_d Normal(mu, std).log_prob(z).  Replaces DefaultPrior.masked_log_prob,
 * nonauto priors/prior.py:148-158; auto priors/prior.py:93. */
int mivae_gauss_logprob_fwd(const void* z, const int64_t* mask, float* tok, int64_t Bn,
                            int64_t mask_div, int64_t S, int64_t D, float mu, float std, int dtype,
                            void* stream);
int mivae_gauss_logprob_bwd(const void* z, const int64_t* mask, const float* dlogp, void* dz,
                            int64_t Bn, int64_t mask_div, int64_t S, int64_t D, float mu, float std,
                            int dtype, void* stream);

/* ---- NICE affine tail given s, t (any coupling net) ------------------------------------------- */
/* out = cat(x0, exp(s) x1 + t) * mask, tok_ld = mask * sum_d s   (inverse: (x1 - t) exp(-s), -s).
 * Replaces NICECoupling.forward couplings.py:151-162 and .backward :172-183 after the s/t nets. */
int mivae_nice_affine_fwd(const void* z, const void* s, const void* t, const int64_t* mask, void* out,
                          float* tok_ld, int64_t Bn, int64_t mask_div, int64_t S, int64_t D, int dtype,
                          int inverse, void* stream);
int mivae_nice_affine_bwd(const void* dout, const float* dld, const void* z, const void* s,
                          const void* t, const int64_t* mask, void* dz, void* ds, void* dt, int64_t Bn,
                          int64_t mask_div, int64_t S, int64_t D, int dtype, int inverse, void* stream);

/* ---- fused NICE feed-forward stack (the dominant kernel) --------------------------------------- */
/* Bytes of packed weights for ONE step (two nets): the forward and backward task streams (weights in
 * MFMA fragment order + biases, in the order the kernels consume them); a K-step image is K times
 * this.  `inverse` != 0 packs the forward stream for the sampling direction (steps K-1..0, t then s). */
int64_t mivae_nice_pack_bytes(int64_t h, int64_t H, int dtype);
/* ptr_table: device int64[K][8] of fp32 parameter addresses {W1s,b1s,W2s,b2s,W1t,b1t,W2t,b2t} in
 * torch.nn.Linear layout (W1 (H,h), W2 (h,H)) -- the tensors behind the state-dict keys
 * `_flows.{k}._{s,t}_net._feedforward.{_linear_layers.0,_output}.{weight,bias}` (couplings.py:26-48). */
int mivae_nice_pack(const int64_t* ptr_table, void* packed, int64_t K, int64_t h, int64_t H, int dtype,
                    int inverse, void* stream);
/* K NICE steps with one-hidden-layer feed-forward nets in ONE launch.  Replaces the loop of
 * FlowPrior.forward_pass (nonauto flow_prior.py:45-52, auto :43-50) / .backward_pass (:65-69) over
 * NICECoupling(FeedForwardCoupling) steps, incl. FeedForward.forward (nn/feedforward.py:93-98).
 * tok_val[token] = mask * (sum_k sum_d s_k  [+ base log-prob of z_K when base != 0]).
 * x1_hist (K, N, h): per-step input of the transformed half, saved for the backward (or NULL).
 * inverse != 0 runs steps K-1..0 in the sampling direction (x1_hist must be NULL). */
int mivae_nice_stack_fwd(const void* z_in, void* z_out, void* x1_hist, const int64_t* mask,
                         const void* packed, float* tok_val, int64_t Bn, int64_t mask_div, int64_t S,
                         int64_t D, int64_t H, int64_t K, int act, int inverse, int base, float mu,
                         float std, int dtype, void* stream);
/* token count padded to the slab granule (leading dimension of Gt / Ht / dSt / x0t) */
int64_t mivae_nice_stack_npad(int64_t N);
/* Backward of the forward direction.  dz_out (N,D) / dlogp (Bn) may be NULL.  Emits dz_in and the
 * operands of the weight-gradient products: Gt, Ht (K*2*Hp*Npad elements) = d pre-activation and hidden
 * activations, dSt (K*2*hp*Npad) = ds / dt, x0t (hp*Npad) = (x0 * mask)^T; Hp, hp = H, h rounded up to
 * 32, Npad = mivae_nice_stack_npad(N).  Layout of each (F, Npad) matrix is TOKEN-BLOCKED:
 * element (feature f, token t) at ((t / TB) * F + f) * TB + t % TB with TB = 16 bytes of tokens (8 bf16,
 * 4 fp32), so both kernels move whole 16-byte vectors along the token axis.  The four buffers are
 * opaque hand-offs to mivae_nice_dw.  (Autograd of couplings.py:141-162 + nn/feedforward.py:93-98
 * w.r.t. z.) */
int mivae_nice_stack_bwd(const void* z_in, const void* z_out, const void* x1_hist, const int64_t* mask,
                         const void* packed, const void* dz_out, const float* dlogp, void* dz_in,
                         void* Gt, void* Ht, void* dSt, void* x0t, int64_t Bn, int64_t mask_div,
                         int64_t S, int64_t D, int64_t H, int64_t K, int act, int base, float mu,
                         float std, int dtype, void* stream);
/* floats of workspace mivae_nice_dw needs for N tokens */
int64_t mivae_nice_dw_workspace_floats(int64_t N, int64_t h, int64_t H, int64_t K);
/* Weight gradients of the K x 2 feed-forward nets from the backward's side outputs, in nn.Linear
 * layout: dW1 (K,2,H,h), db1 (K,2,H), dW2 (K,2,h,H), db2 (K,2,h), fp32.  Split over token slabs and
 * reduced in fixed order (deterministic).  (Autograd of nn/feedforward.py:93-98 w.r.t. the
 * `_linear_layers.0.{weight,bias}` / `_output.{weight,bias}` parameters of both nets.) */
int mivae_nice_dw(const void* Gt, const void* Ht, const void* dSt, const void* x0t, float* workspace,
                  float* dW1, float* db1, float* dW2, float* db2, int64_t N, int64_t h, int64_t H,
                  int64_t K, int dtype, void* stream);

/* As mivae_nice_dw, but the gradients are ADDED into the parameters' own fp32 gradient buffers: grad_ptrs is a
 * device int64[K][8] table of addresses in mivae_nice_pack's order (32 per-parameter add kernels less per step). */
int mivae_nice_dw_sink(const void* Gt, const void* Ht, const void* dSt, const void* x0t, float* workspace,
                       const int64_t* grad_ptrs, int64_t N, int64_t h, int64_t H, int64_t K, int dtype,
                       void* stream);

/* As mivae_nice_dw for K coupling steps that are not consecutive in their flow list (ActNorm / linear steps between
 * them change x0, flow_prior.py:37-52): x0t holds one (hp, Npad) section per step.  Weight gradients of the coupling
 * nets of a mixed list (couplings.py:141-162 under autograd) after mivae_mixed_stack_bwd. */
int mivae_nice_dw_steps(const void* Gt, const void* Ht, const void* dSt, const void* x0t, float* workspace,
                        float* dW1, float* db1, float* dW2, float* db2, int64_t N, int64_t h, int64_t H,
                        int64_t K, int dtype, void* stream);

/* ---- ActNorm ------------------------------------------------------------------------------------- */
/* out = (z exp(log_scale) + bias) * mask  (inverse: ((z - bias) * mask) / (exp(log_scale) + 1e-13));
 * logdet[i] = +-sum(log_scale) * (count_weighted ? token count of row i : 1), i < n_logdet.
 * Replaces ActNorm.forward actnorm.py:31-39 and .backward :45-54 (after init). */
int mivae_actnorm_fwd(const void* z, const float* log_scale, const float* bias, const int64_t* mask,
                      void* out, float* logdet, int64_t Bn, int64_t mask_div, int64_t S, int64_t D,
                      int64_t n_logdet, int count_weighted, int dtype, int inverse, void* stream);
/* floats of workspace needed by mivae_actnorm_bwd / mivae_actnorm_stats */
int64_t mivae_col_workspace_floats(int64_t rows, int64_t D, int dtype);
/* dz and dparams = [dlog_scale (D) | dbias (D)] of the forward direction (autograd of :31-33). */
int mivae_actnorm_bwd(const void* dout, const void* z, const float* log_scale, const int64_t* mask,
                      void* dz, float* dparams, float* workspace, int64_t Bn, int64_t mask_div,
                      int64_t S, int64_t D, int dtype, void* stream);
/* column mean and unbiased std over all rows, padding included: the data-dependent init statistics
 * of actnorm.py:22-28. */
int mivae_actnorm_stats(const void* z, float* mean, float* std, float* workspace, int64_t rows,
                        int64_t D, int dtype, void* stream);

/* out[c] = sum_rows x[row, c], fp32, fixed order: the bias gradient of a Linear over `rows` tokens (autograd of
 * the posterior heads' torch.nn.Linear, posterior.py:19-27).  workspace: mivae_col_workspace_floats(rows, D). */
int mivae_colsum(const void* x, float* out, float* workspace, int64_t rows, int64_t D, int dtype,
                 void* stream);

/* Allocation-free variant of mivae_colsum.  No counterpart in the reference: it exists for callers that cannot
 * allocate device memory (code under HIP-graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same
 * result as mivae_colsum, bit for bit; the split-row partial sums live in a 4 MiB file-scope device array that every
 * GPU the library runs on holds until the process ends (resolved once per device with hipGetSymbolAddress).  Limit:
 * mivae_col_workspace_floats(rows, D, dtype) <= mivae_colsum_static_max_floats() (= 2^20), else MIVAE_EUNSUPPORTED (-3)
 * with a message.  Calls no hipMalloc / hipFree.  Calls on one GPU share the array: calls that overlap in time on the
 * same GPU (different streams) race.  The array used is the one of the STREAM's device, which must be the current
 * device (EINVAL otherwise).  mivae_colsum_static_init() resolves the current device's array eagerly: call it once per
 * device before the first use under HIP-graph capture (a first use inside a capture is rejected with EINVAL, since
 * module load / hipGetSymbolAddress are not capture-safe); eager callers may skip it. */
int64_t mivae_colsum_static_max_floats(void);
int mivae_colsum_static_init(void);
int mivae_colsum_static(const void* x, float* out, int64_t rows, int64_t D, int dtype, void* stream);

/* First stage of mivae_colsum only: per-block partial column sums of x (rows, D) into workspace (count, D) fp32, count
 * returned (> 0; negative = error).  Fold them with mivae_colsum_sink_multi, e.g. once per gradient bucket for all bias
 * gradients of a step (the wide Linear layers of the bracket, nn/transformer.py:41-48, 129-130). */
int64_t mivae_colsum_partial(const void* x, float* workspace, int64_t rows, int64_t D, int dtype, void* stream);
/* As mivae_colsum, but ADDS columns [0, split) into out0 and [split, D) into out1: a finished [dW | db] row goes
 * straight into the parameters' gradient buffers (the accumulation autograd's AccumulateGrad would do with one
 * extra kernel per parameter). */
int mivae_colsum_sink(const void* x, float* out0, float* out1, int64_t split, float* workspace, int64_t rows,
                      int64_t D, int dtype, void* stream);
/* Weight gradient of torch.nn.Embedding (the Embedder in front of VAELmModel.encode, models/base/model.py:213-222):
 * dW (V, E) fp32 (+)= sum over tokens with ids[t] == v of dy[t] (dy (N, E) fp32, ids (N) int64), rows of padding_idx
 * (< 0: none) skipped.  Deterministic (token order, no atomics, no sort).  accumulate != 0: add into dW (gradient sink),
 * else dW is fully written.  E <= 512, else MIVAE_EUNSUPPORTED (-3). */
int mivae_embedding_bwd(const float* dy, const int64_t* ids, float* dW, int64_t N, int64_t V, int64_t E,
                        int64_t padding_idx, int accumulate, void* stream);

/* Fold of the vocabulary projection's weight gradient (nonauto_lm/model.py:105-112, the Linear behind the decoder):
 * part (G, V, E + A) fp32 = G token-slab partials of dlogits^T [h | 1 ... 1] (A ones columns appended to h);
 * dW (V, E) += sum_g part[g][:, :E], db (V) += sum_g part[g][:, E], fixed order, straight into the gradient buffers. */
int mivae_fold_aug_sink(const float* part, int64_t G, int64_t V, int64_t E, int64_t A, float* dW, float* db, void* stream);
/* mivae_colsum_sink for MANY short matrices in one launch: jobs = njobs x {x, out0, out1, rows (<= 512), D, split,
 * stride} as 7 x int64 per job in device memory (stride = floats between the rows of x, >= D: x may be a column block
 * of a wider partial matrix); bit-identical to njobs calls of mivae_colsum_sink (same summation order).
 * Folds the [dW | db] partials of every small Linear of a training step at the end of backward
 * (trainer.py:284, loss.backward()).  max_D = the largest D in the table. */
int mivae_colsum_sink_multi(const void* jobs, int64_t njobs, int64_t max_D, void* stream);

/* ---- InfoVAE MMD: pairwise squared distances (a19 / f4) ------------------------------------------------- */
/* dens (Bn, Bn) fp32: dens[i][j] = sum_{s,d} mask[i][s] * Normal(mu[j / n], sigma[j / n]).log_prob(z[i]) -- every latent
 * code against every sentence's posterior, the pairwise term of Posterior.calc_mutual_info (nonauto posterior.py:174-222,
 * auto posterior.py:158-200 with S = 1, mask NULL) without its (Bn, Bn, S, D) intermediate.  z (Bn, S, D), mu / sigma
 * (Bn / n, S, D) in `dtype`; mask (Bn, S) int64 or NULL.  The caller takes logsumexp over j. */
int mivae_pair_gauss_logdens(const void* z, const void* mu, const void* sigma, const int64_t* mask, float* out, int64_t Bn,
                             int64_t n, int64_t S, int64_t D, int dtype, void* stream);
/* out (B1, B2) fp32, out[i][j] = sum_f (a[i][f] - b[j][f])^2 for a (B1, F), b (B2, F): the pairwise term of
 * InfoVAEKLLoss._compute_rbf / _compute_inv_mult_quad (nn/kl_loss.py:124-149; the reference builds the (B, B, F)
 * difference tensor) before the kernel function, in the reference's own arithmetic (differences, not a Gram identity),
 * tiled through LDS, fixed summation order. */
int mivae_pair_sqdist(const void* a, const void* b, float* out, int64_t B1, int64_t B2, int64_t F, int dtype,
                      void* stream);
/* gradients of sum_ij G[i][j] out[i][j] w.r.t. a and b: da (B1, F), db (B2, F) fp32, each may be NULL; accumulate != 0
 * adds into them.  Autograd of the expression above. */
int mivae_pair_sqdist_bwd(const void* a, const void* b, const float* G, float* da, float* db, int64_t B1, int64_t B2,
                          int64_t F, int dtype, int accumulate, void* stream);

/* ---- invertible linear flows (a14) ------------------------------------------------------------------- */
/* Y (R, hs) = X (R, hs) M^T on the matrix cores, M (hs, hs) row major in the latent dtype, 1 <= hs <= 512.
 * Replaces F.linear(z, W) of InvertibleLinearFlow.forward / .backward (models/base/flows/invertible_flows.py:29,
 * :39; M = W or W^-1) and of InvertibleMultiHeadFlow (:61-63, :75-77: the rearrange to heads is the identity on the
 * memory image, R = tokens x heads, hs = D / heads).  The input gradient d X = d Y M is the same call on M^T.
 * fp32: v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain); bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulation. */
int mivae_rowgemm(const void* x, const void* m, void* y, int64_t R, int64_t hs, int dtype, void* stream);
/* d M of Y = X M^T as per-slab partial sums: part (mivae_rowgemm_wgrad_slabs(R, hs), hs, hs) fp32,
 * part[s][o][i] = sum_{r in slab s} dY[r][o] X[r][i]; sum the slabs with mivae_colsum (fixed order).  This is the
 * weight gradient autograd derives for F.linear in invertible_flows.py:29 / :61-63. */
int64_t mivae_rowgemm_wgrad_slabs(int64_t R, int64_t hs);
int mivae_rowgemm_wgrad(const void* x, const void* dy, float* part, int64_t R, int64_t hs, int dtype, void* stream);

/* log |det M_i| of `count` <= 24 small square fp32 matrices (n_i <= 128) and, where inv_t[i] != NULL, M_i^-T = its
 * gradient, in one launch: torch.slogdet(self._weight)[1] of invertible_flows.py:30-33 / :64-66 and what autograd
 * derives from it.  mats / n / inv_t are HOST arrays (device pointers, sizes), read during the call; logdet (count) is
 * a device array.  Gauss-Jordan with partial pivoting in LDS, fp32; a singular matrix gives -inf. */
int mivae_slogdet_inv(const void* const* mats, const int64_t* n, int64_t count, float* logdet, void* const* inv_t,
                      void* stream);

/* Mixed flow stack, one launch, no gradients (evaluation / sampling): K steps executed in the order given, each a
 * NICE feed-forward coupling (couplings.py:141-183 + :50-51), an ActNorm (actnorm.py:30-54) or an invertible linear
 * map with one matrix per head (invertible_flows.py:26-44, :58-83), the loop of FlowPrior.forward_pass /
 * backward_pass (flow_prior.py:37-70) for lists that are not coupling-only.  D even, <= 128; coupling hidden <= 128.
 *   steps: HOST array of 12 int64 per step {kind 0 NICE / 1 ActNorm / 2 linear, H, activation id, head size, 8 device
 *   pointers to fp32 parameters (NICE: W1s b1s W2s b2s W1t b1t W2t b2t; ActNorm: log_scale bias; linear: M (hs, hs))},
 *   read during the call.  For the sampling direction pass the list reversed, inverse = 1 and the inverted matrices.
 *   tok_val (Bn S) fp32 or NULL: masked per-token sum of the coupling log-scales (negated when inverse); the
 *   parameter-only log-determinants (sum log_scale, slogdet M) x token count are the caller's to add. */
int mivae_mixed_stack_supported(int64_t D, int64_t Hmax);
int mivae_mixed_stack(const void* z_in, void* z_out, const int64_t* mask, float* tok_val, const int64_t* steps,
                      int64_t K, int64_t Bn, int64_t S, int64_t mask_div, int64_t D, int inverse, int dtype,
                      void* stream);

/* Training path of the same lists (density direction): what autograd runs through the per-flow modules of
 * FlowPrior.forward_pass (flow_prior.py:37-52; actnorm.py:30-39, invertible_flows.py:26-33 / :58-66,
 * couplings.py:141-162), in one launch each way.
 *   mivae_mixed_stack_train_fwd = mivae_mixed_stack(inverse = 0) that also writes the input of every step to
 *     hist (K, Bn S, D) fp32.
 *   mivae_mixed_stack_bwd: steps K-1 .. 0 with the cotangent tile resident in LDS.  dz_out (Bn S, D) latent dtype or
 *     NULL; dval = cotangent of tok_val, one per sentence (Bn) or, dval_per_token = 1, one per token, or NULL.
 *     Outputs: dz_in; gcot (nL, Bn S, D) fp32 = cotangent at every linear step's output, row-major (d M = gcot^T hist[k]
 *     through mivae_rowgemm_wgrad); an_part (blocks, nA, 2, D) fp32, blocks = mivae_mixed_stack_bwd_blocks: its sum over
 *     the block axis (mivae_colsum) is [d log_scale | d bias] of every ActNorm step (z path; the log-det terms of the
 *     parameters are the caller's, like the forward's); Gt, Ht (nN, 2, Hp, Npad), dSt (nN, 2, hp, Npad), x0t (nN, hp,
 *     Npad) fp32, Npad = mivae_nice_stack_npad(Bn S): operands of mivae_nice_dw_steps (dtype MIVAE_F32).
 *     nN / nA / nL = number of coupling / ActNorm / linear steps, ordinals in list order.  All coupling steps must share
 *     one hidden width (MIVAE_EUNSUPPORTED otherwise; mivae_mixed_stack_bwd_blocks returns 0 for such lists). */
int mivae_mixed_stack_train_fwd(const void* z_in, void* z_out, const int64_t* mask, float* tok_val, float* hist,
                                const int64_t* steps, int64_t K, int64_t Bn, int64_t S, int64_t mask_div, int64_t D,
                                int dtype, void* stream);
int64_t mivae_mixed_stack_bwd_blocks(int64_t N, int64_t D, int64_t Hmax, int64_t Hmin, int any_linear);
int mivae_mixed_stack_bwd(const float* hist, const void* dz_out, const float* dval, int dval_per_token,
                          const int64_t* mask, const int64_t* steps, int64_t K, void* dz_in, float* gcot, float* an_part,
                          float* Gt, float* Ht, float* dSt, float* x0t, int64_t Bn, int64_t S, int64_t mask_div,
                          int64_t D, int dtype, void* stream);

/* ---- KL / ELBO tail -------------------------------------------------------------------------------- */
/* kl = clamp(logq - logp, min = free_bits) (clamp only if use_fb).  Replaces DefaultKLLoss.forward
 * nn/kl_loss.py:38-49 and _free_bits_kl :14-21. */
int mivae_kl_fwd(const float* logq, const float* logp, float* kl, int64_t n, int use_fb,
                 float free_bits, void* stream);
int mivae_kl_bwd(const float* logq, const float* logp, const float* dkl, float* dq, float* dp,
                 int64_t n, int use_fb, float free_bits, void* stream);
/* out3 = {batch-loss, batch-kl, batch-nll}: loss = wr*recon + wk*kl per (b, j); IWAE softmax-weighted
 * sum over the n samples (weights not detached); means.  Replaces VAELmModel.forward's tail,
 * models/base/model.py:171-193.  weights_dev (may be NULL): device floats {recon weight, kl weight} read
 * INSTEAD of the by-value weights -- the scheduler weights of nn/weight_scheduler.py:33-60 change every step, and a
 * launch captured in a HIP graph would otherwise replay with the weights of the capture step. */
int mivae_elbo_fwd(const float* recon, const float* kl, float* out3, int64_t B, int64_t n,
                   float recon_weight, float kl_weight, int iwae, const float* weights_dev, void* stream);
int mivae_elbo_bwd(const float* recon, const float* kl, const float* gout3, float* drecon, float* dkl,
                   int64_t B, int64_t n, float recon_weight, float kl_weight, int iwae,
                   const float* weights_dev, void* stream);
/* The per-sentence reductions of log q / log p (posterior.py:157-172, priors/prior.py:148-158: sums of the per-token
 * values), DefaultKLLoss (nn/kl_loss.py:38-49) and the ELBO tail (models/base/model.py:171-193) in ONE launch each way:
 * tok_logq, tok_logp ((B n), S) fp32, recon (B n) -> logq, logp, kl (B n) and out3 = {batch-loss, batch-kl, batch-nll};
 * the backward turns the cotangents of the three scalars (one device float each, NULL = zero) into d recon, d log q,
 * d log p per sentence.  rng_state (or NULL): a generator state whose
 * offset the forward advances by one (see mivae_posterior_fwd).  One block: B n <= 16384, (B n) S <= 2^21
 * (mivae_latent_tail_supported); larger shapes use mivae_segment_sum + mivae_kl_* + mivae_elbo_*. */
int mivae_latent_tail_supported(int64_t B, int64_t n, int64_t S);
int mivae_latent_tail_fwd(const float* tok_logq, const float* tok_logp, const float* recon, float* logq, float* logp,
                          float* kl, float* out3, int64_t B, int64_t n, int64_t S, int use_free_bits, float free_bits,
                          float recon_weight, float kl_weight, int iwae, const float* weights_dev, int64_t* rng_state,
                          void* stream);
int mivae_latent_tail_bwd(const float* recon, const float* logq, const float* logp, const float* g_loss, const float* g_kl,
                          const float* g_nll, float* drecon, float* dlogq, float* dlogp, int64_t B, int64_t n,
                          int use_free_bits, float free_bits, float recon_weight, float kl_weight, int iwae,
                          const float* weights_dev, void* stream);

/* ---- reconstruction term: label-smoothing NLL over the vocabulary ------------------------------ */
/* per_seq[b] = sum_s w[b,s] * nll[b,s] / clamp(sum_s w[b,s], 1e-13) with
 * nll = -sum_v q_v log_softmax(logits[b,s,:])_v, q = (1 - smoothing) onehot(target) + smoothing / V
 * (smoothing = 0: plain gather).  Replaces LabelSmoothingNLL.forward with size_average=False,
 * models/base/losses.py:128-169 (the caller passes smoothing = 0 in eval mode, losses.py:144).
 * logits (Bn,S,V) in `dtype`; target (Bn,S) i64; weights (Bn,S) f32.  Also written for the backward:
 * lse (Bn*S) log-sum-exp per token, tokscale (Bn*S) = w / clamp(sum w); tok_nll (Bn*S) is scratch.
 * Rows with weight 0 are not read.  One read of the logits. */
int mivae_nll_fwd(const void* logits, const int64_t* target, const float* weights, float* per_seq,
                  float* lse, float* tokscale, float* tok_nll, int64_t Bn, int64_t S, int64_t V,
                  float smoothing, int dtype, void* stream);
/* dlogits[b,s,v] = dper_seq[b] * tokscale[b,s] * (softmax(logits[b,s,:])_v - q_v): autograd of the
 * chain above (losses.py:141-164).  One read of the logits, one write of dlogits (same dtype). */
int mivae_nll_bwd(const void* logits, const int64_t* target, const float* lse, const float* tokscale,
                  const float* dper_seq, void* dlogits, int64_t Bn, int64_t S, int64_t V,
                  float smoothing, int dtype, void* stream);

/* ---- fused vocabulary projection + NLL (the logits never reach HBM) --------------------------------- */
/* Same outputs as mivae_nll_fwd for logits = h W^T + bias, computed tile by tile on the matrix cores:
 * `self._vocab_projection(...)` of NonAutoModel.decode (nonauto_lm/model.py:105-112) followed by
 * LabelSmoothingNLL.forward (models/base/losses.py:128-169).  h (Bn*S, E) bf16; `W` is the packed weight + bias
 * image from mivae_lce_pack; E in {32, 64, 128, 256}; V a multiple of 64.  workspace: mivae_lce_workspace_floats(Bn*S, V) floats. */
int64_t mivae_lce_workspace_floats(int64_t N, int64_t V);
/* W (V, E) bf16 row-major + bias (V) fp32 or NULL -> Wp (mivae_lce_pack_elems(V, E) = V * (E + 16) bf16): the
 * MFMA-fragment-ordered image the two kernels below read as `W` (every wave load is one contiguous 1 KiB); the
 * bias rides along as an extra k-step (bf16 hi + lo halves against a [1, 1, 0, ...] column of h). */
int64_t mivae_lce_pack_elems(int64_t V, int64_t E);
int mivae_lce_pack(const void* W, const float* bias, void* Wp, int64_t V, int64_t E, void* stream);
int mivae_lce_fwd(const void* h, const void* W, const int64_t* target,
                  const float* weights, float* per_seq, float* lse, float* tokscale, float* workspace,
                  int64_t Bn, int64_t S, int64_t E, int64_t V, float smoothing, void* stream);
/* dlogits (Bn*S, V) bf16 = dper_seq[b] * tokscale * (softmax(h W^T + bias) - q), recomputed from h and W (autograd
 * of the chain above w.r.t. the logits); d h = dlogits W, d W = dlogits^T h, d bias = colsum(dlogits) are the
 * caller's dense products. */
int mivae_lce_bwd(const void* h, const void* W, const int64_t* target, const float* lse,
                  const float* tokscale, const float* dper_seq, void* dlogits, int64_t Bn, int64_t S,
                  int64_t E, int64_t V, float smoothing, void* stream);
/* As mivae_lce_bwd, and d h (Bn*S, E) bf16 = dlogits W (autograd of the vocabulary projection w.r.t. its input,
 * nonauto_lm/model.py:105) is formed in the same pass on the matrix cores from the bf16 dlogits tiles while they are
 * in registers: the (N, V) x (V, E) GEMM and its read of dlogits disappear.  Wtp (V * E bf16): the W^T fragment
 * image written by mivae_lce_pack2 (mivae_lce_pack + Wtp, one launch; Wtp may be NULL there).  workspace:
 * mivae_lce_dh_workspace_floats(Bn*S, E, V) floats (per-vocabulary-slice fp32 partials, summed in slice order). */
int mivae_lce_pack2(const void* W, const float* bias, void* Wp, void* Wtp, int64_t V, int64_t E, void* stream);
int64_t mivae_lce_dh_workspace_floats(int64_t N, int64_t E, int64_t V);
int mivae_lce_bwd_dh(const void* h, const void* W, const void* Wtp, const int64_t* target, const float* lse,
                     const float* tokscale, const float* dper_seq, void* dlogits, float* workspace, void* dh,
                     int64_t Bn, int64_t S, int64_t E, int64_t V, float smoothing, void* stream);
/* d W = dlogits^T h and d bias = dlogits^T 1 of the same chain (autograd of the vocabulary projection w.r.t. its
 * parameters, nonauto_lm/model.py:105 under trainer.py:284) WITHOUT a stored dlogits: a block owns 64 vocabulary rows
 * and a share of the tokens, recomputes the logits tiles from h and W, forms the bf16 dlogits of mivae_lce_bwd in
 * registers and contracts them with h on the matrix cores.  part (slabs, V, E + 8) fp32 with slabs =
 * mivae_lce_dw_slabs(Bn*S, E, V) (0: shape not supported; E in {32, 64}): [d W | d bias | 7 unused columns] per token
 * split, summed over the slab axis by mivae_fold_aug_sink (A = 8) or mivae_colsum.  With it mivae_lce_bwd_dh may be
 * called with dlogits = NULL (E <= 64): the (N, V) gradient is then neither written nor read. */
int64_t mivae_lce_dw_slabs(int64_t N, int64_t E, int64_t V);
int mivae_lce_dw(const void* h, const void* W, const int64_t* target, const float* lse, const float* tokscale,
                 const float* dper_seq, float* part, int64_t Bn, int64_t S, int64_t E, int64_t V, float smoothing,
                 void* stream);

/* ---- token-parallel small Linear layers ---------------------------------------------------------------- */
/* torch.nn.Linear over many tokens and few features, bf16 operands, fp32 accumulation on the matrix cores: the
 * posterior heads (posterior.py:19-27, row a3) and the q/k/v/out/FFN projections of the attention coupling /
 * transformer layers (nn/transformer.py:41-48, 129-130).  x (N, Fi), w (Fo, Fi) row-major (nn.Linear layout),
 * bias (Fo) or NULL, y (N, Fo); Fi, Fo in {32, 64, 128, 256}. */
int mivae_linear_fwd(const void* x, const void* w, const void* bias, void* y, int64_t N, int64_t Fi, int64_t Fo,
                     void* stream);
/* out (N, Fo) = res + bf16(x w^T + bias), the sum formed in res's dtype (fp32 or bf16) exactly as torch forms
 * `res + linear(x)` under bf16 autocast: the output projection of MultiHeadAttention inside SublayerConnection
 * (nn/transformer.py:116-117 + 160-162) with the residual add as the product's epilogue. */
int mivae_linear_fwd_residual(const void* x, const void* w, const void* bias, const void* res, void* out, int64_t N,
                              int64_t Fi, int64_t Fo, int res_dtype, void* stream);
/* dx (N, Fi) = dy (N, Fo) w  (autograd of the above w.r.t. x). */
int mivae_linear_dgrad(const void* dy, const void* w, void* dx, int64_t N, int64_t Fi, int64_t Fo, void* stream);
/* partial (mivae_linear_wgrad_rows(N), Fo*Fi + Fo) fp32: per-wave [dy^T x | colsum dy]; its column sums
 * (mivae_colsum) are [dW (Fo, Fi) | db (Fo)] (autograd w.r.t. weight and bias).  Fi, Fo in {32, 64, 128}. */
/* G (1..4) Linear layers of one shape applied to the SAME input -- the query / key / value projections of
 * self-attention (nn/transformer.py:41-48, 60-62) and the mu / sigma heads of the posterior (posterior.py:19-27,
 * row a3): y_g = x W_g^T + bias_g for all g in one launch (one read of x), and dx = sum_g dy_g W_g as ONE contraction
 * over G * Fo <= 256 (no per-layer dx, no accumulation kernels).  w / bias / y / dy: HOST arrays of G device pointers
 * (bias or its entries may be NULL).  Weight gradients: mivae_linear_wgrad per layer. */
int mivae_linear_fwd_group(const void* x, const void* const* w, const void* const* bias, void* const* y, int64_t G,
                           int64_t N, int64_t Fi, int64_t Fo, void* stream);
int mivae_linear_dgrad_group(const void* const* dy, const void* const* w, void* dx, int64_t G, int64_t N, int64_t Fi,
                             int64_t Fo, void* stream);
/* Pre-norm feed-forward sublayer forward in ONE launch: y = x + W2 relu(W1 LN(x) + b1) + b2 over (N, D) token rows --
 * SublayerConnection (nn/transformer.py:147-162) around PositionwiseFeedForward (:120-144) with inactive dropout, in
 * the transformer bracket and the attention coupling (row a11).  x / y fp32 (x_dtype 0) or bf16 (1); gamma / beta fp32;
 * w1 (H, D), b1 (H), w2 (D, H), b2 (D) bf16.  Written for the backward pass: xn = LN(x) (N, D) bf16, hmid = relu(.)
 * (N, H) bf16, mean / rstd (N) fp32; the backward is mivae_linear_dgrad / _wgrad + mivae_layernorm_bwd*.
 * D in {64, 128}, H in {128, 256}, else MIVAE_EUNSUPPORTED (-3). */
int mivae_ffn_fwd(const void* x, const float* gamma, const float* beta, const void* w1, const void* b1, const void* w2,
                  const void* b2, void* y, void* xn, void* hmid, float* mean, float* rstd, int64_t N, int64_t D,
                  int64_t H, float eps, int x_dtype, void* stream);
/* Up to 16 weight-gradient problems over the same N tokens in one launch: every small-Linear weight gradient a training
 * step's backward pass queued (nn/transformer.py:41-48, 129-130; posterior.py:19-27), issued together right before the
 * partials are folded.  desc: nprob rows of 12 int64 in HOST memory, {Fo, Fi, G, x, dy[0..4), partial[0..4)} (device
 * addresses); group == 0: single layers (G = 1), (Fo, Fi) in {(32,32), (32,64), (64,32), (64,64), (64,128), (128,64)};
 * group != 0: groups of G in {2, 3} square layers (32 or 64 wide) sharing x.  Outputs as mivae_linear_wgrad /
 * mivae_linear_wgrad_group write them; MIVAE_EUNSUPPORTED (-3) and no launch if a problem is outside those shapes. */
int mivae_linear_wgrad_multi(const int64_t* desc, int64_t nprob, int64_t N, int group, void* stream);
/* mivae_linear_wgrad(x0, dy0) -> partial0 and mivae_linear_wgrad_group(x, dy[0..G)) -> partial[0..G) in one launch, every
 * layer F x F over the same N tokens: the output projection of MultiHeadAttention and its q / k / v projections
 * (nn/transformer.py:41-48).  dy / partial: host arrays of G device pointers.  F in {32, 64}, G = 3; else -3. */
int mivae_linear_wgrad_single_group(const void* x0, const void* dy0, float* partial0, const void* x, const void* const* dy,
                                    float* const* partial, int64_t G, int64_t N, int64_t F, void* stream);
/* mivae_linear_wgrad for the two Linear layers of PositionwiseFeedForward (nn/transformer.py:120-133) in one launch:
 * problem 0 = (x0 (N, Fi0), dy0 (N, Fo0)) -> partial0, problem 1 likewise, same N.  Instantiated for (Fi0, Fo0) =
 * (128, 64), (Fi1, Fo1) = (64, 128); MIVAE_EUNSUPPORTED (-3) otherwise. */
int mivae_linear_wgrad_pair(const void* x0, const void* dy0, float* partial0, int64_t Fi0, int64_t Fo0, const void* x1,
                            const void* dy1, float* partial1, int64_t Fi1, int64_t Fo1, int64_t N, void* stream);
/* mivae_linear_wgrad for G (2..4) same-shape layers sharing x in one launch (x staged once): partial[g] as above for
 * layer g; dy / partial are host arrays of G device pointers.  Fi = Fo in {32, 64}, else MIVAE_EUNSUPPORTED (-3). */
int mivae_linear_wgrad_group(const void* x, const void* const* dy, float* const* partial, int64_t G, int64_t N,
                             int64_t Fi, int64_t Fo, void* stream);
/* mivae_linear_dgrad fused with the derivative of a relu in front of the layer: dx = (dy W) * (relu_out > 0);
 * relu_out (N, Fi) bf16 is the layer's input (PositionwiseFeedForward, nn/transformer.py:141-144). */
int mivae_linear_dgrad_relu(const void* dy, const void* w, const void* relu_out, void* dx, int64_t N, int64_t Fi,
                            int64_t Fo, void* stream);
int64_t mivae_linear_wgrad_rows(int64_t N);
int mivae_linear_wgrad(const void* x, const void* dy, float* partial, int64_t N, int64_t Fi, int64_t Fo,
                       void* stream);

/* ---- LayerNorm (pre-norm sublayers of the attention coupling / transformer bracket) ------------------ */
/* y = (x - mean) * rstd * gamma + beta over the last axis (torch.nn.LayerNorm inside SublayerConnection,
 * nn/transformer.py:147-162; used by AttentionCoupling, couplings.py:54-81).  x (rows, D) in in_dtype, y in
 * out_dtype (bf16 output from the fp32 residual stream saves the cast before the next projection); mean, rstd
 * (rows) fp32 are kept for the backward.  D <= 1024 (multiple of 4, aligned) or <= 256 otherwise. */
int mivae_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                        float* rstd, int64_t rows, int64_t D, float eps, int in_dtype, int out_dtype,
                        void* stream);
int64_t mivae_layernorm_workspace_floats(int64_t rows, int64_t D);
/* dx (in_dtype) and dgamma_dbeta = [dgamma (D) | dbeta (D)] from dy (out_dtype); fixed-order reductions. */
int mivae_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                        const float* rstd, void* dx, float* dgamma_dbeta, float* workspace, int64_t rows,
                        int64_t D, int in_dtype, int out_dtype, void* stream);

/* As mivae_layernorm_bwd with d gamma ADDED into dgamma (D) and d beta ADDED into dbeta (D). */
int mivae_layernorm_bwd_sink(const void* dy, const void* x, const float* gamma, const float* mean,
                             const float* rstd, void* dx, float* dgamma, float* dbeta, float* workspace,
                             int64_t rows, int64_t D, int in_dtype, int out_dtype, void* stream);
/* dx as mivae_layernorm_bwd; the parameter gradients stay as per-block partials: workspace is a (count, 2 D) fp32
 * matrix whose column sums are [d gamma | d beta]; count (> 0) is RETURNED (negative = error status).  Fold with
 * mivae_colsum_sink(_multi) (split = D), e.g. once per step for all LayerNorm layers. */
int64_t mivae_layernorm_bwd_partial(const void* dy, const void* x, const float* gamma, const float* mean,
                                    const float* rstd, void* dx, float* workspace, int64_t rows, int64_t D,
                                    int in_dtype, int out_dtype, void* stream);
/* mivae_layernorm_bwd_partial with the residual branch of a pre-norm sublayer folded in: dx = add + LN input gradient
 * (`add` (rows, D) in x's dtype) -- the x-gradient of y = x + f(LayerNorm(x)), SublayerConnection.forward
 * (vae_lm/nn/transformer.py:160-162), in one store instead of a separate add launch. */
int64_t mivae_layernorm_bwd_partial_add(const void* dy, const void* x, const float* gamma, const float* mean,
                                        const float* rstd, const void* add, void* dx, float* workspace, int64_t rows,
                                        int64_t D, int in_dtype, int out_dtype, void* stream);
/* mivae_layernorm_bwd_partial_add that also writes dx rounded to bf16 into dx16 (rows, D): the gradient of the fp32
 * residual stream in the dtype the previous sublayer's projection backward works in (SublayerConnection,
 * nn/transformer.py:147-162), so that backward launches no cast. */
int64_t mivae_layernorm_bwd_partial_add_twin(const void* dy, const void* x, const float* gamma, const float* mean,
                                             const float* rstd, const void* add, void* dx, void* dx16, float* workspace,
                                             int64_t rows, int64_t D, int in_dtype, int out_dtype, void* stream);

/* ---- step-harness tail on flat buffers ---------------------------------------------------------- */
/* Gradient clipping + Adam + zero_grad of DefaultTrainer._train_batch (training/trainers/trainer.py:286-289
 * clip_grad_norm_ / clip_grad_value_, :359-367 optimizer.step() / zero_grad(), optimizer = torch.optim.Adam
 * via nn/optimizer.py:11-33) over the flat parameter / gradient buckets of the DP reducer.
 * mivae_sumsq_partial writes mivae_sumsq_blocks(n) fixed-order partial sums of g^2; the partial arrays of all
 * buckets are laid out back to back.  mivae_adam_clip_step folds ALL npartial partials into the global norm
 * (same order in every block), scales by min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: off), clamps to
 * +-max_value (<= 0: off), applies Adam (torch formula, L2 weight_decay) with step count *step + 1, writes
 * p, m, v (and, when p_bf16 is not NULL, a bf16 copy of the new p: the shadow the bf16 forward reads, so no
 * per-layer weight casts are needed) and zeroes g.  *step is advanced separately with mivae_scalar_add(step, 1) after all buckets.
 * n must be a multiple of 4 floats, buffers 16-byte aligned.  norm_out may be NULL.  grad_scale (> 0) multiplies
 * every gradient before anything else (and so the norm): 1 / world_size after a SUM all-reduce of the buckets, the
 * averaging DistributedDataParallel does (trainer.py:51-60), without a pass of its own. */
int64_t mivae_sumsq_blocks(int64_t n);
int mivae_sumsq_partial(const float* g, float* partial, int64_t n, void* stream);
int mivae_adam_clip_step(float* p, void* p_bf16, float* g, float* m, float* v, const float* partial, int64_t npartial,
                         const float* step, float* norm_out, int64_t n, double lr, double beta1,
                         double beta2, float eps, float weight_decay, float max_norm, float max_value,
                         float grad_scale, void* stream);
int mivae_scalar_add(float* x, float delta, void* stream);
/* *dst[i] += (src[i] ? *src[i] : inc[i]) for n <= 4 device scalars in one launch (host arrays of device pointers): the
 * running metric sums of models/base/model.py:200-210 and the optimizer's step counter inside a replayed graph. */
int mivae_accumulate_scalars(float* const* dst, const float* const* src, const float* inc, int64_t n, void* stream);
/* Measurement aid, no counterpart in the reference: dst[0:bytes] = src[0:bytes] as a plain streaming copy (16 bytes per
 * lane).  bench.py times it on 256 MiB to report the HBM bandwidth this box reaches beside the datasheet peak
 * (both denominators).  bytes % 16 == 0, 16-byte aligned buffers.  mode bit 0: non-temporal loads / stores;
 * mode / 16 = workgroups per CU (0: 16). */
int mivae_hbm_copy(const void* src, void* dst, int64_t bytes, int mode, void* stream);

/* ---- masked autoregressive flows ----------------------------------------------- */
/* Tail of MAFlow.forward (vae_lm/models/base/flows/maf.py:35-45) given st = MADE(z) as one (rows, 2D) tensor
 * [s | t]:  out = flip_if_parity(z * exp(s) + t) * mask,  tok_ld[row] = mask * sum_d s.  mask: one int64 0/1 per
 * row or NULL.  The backward takes the cotangents of out (may be NULL) and of tok_ld (may be NULL) and writes the
 * direct term dz = g e^s and d st = [g e^s z + dtok | g]; the MADE term of dz comes back through d st. */
int mivae_maf_affine_fwd(const void* z, const void* st, const int64_t* mask, void* out, float* tok_ld,
                         int64_t rows, int64_t D, int parity, int dtype, void* stream);
int mivae_maf_affine_bwd(const void* dout, const float* dtok, const void* z, const void* st, const int64_t* mask,
                         void* dz, void* dst, int64_t rows, int64_t D, int parity, int dtype, void* stream);

/* MAFlow.backward (maf.py:47-62; IAFlow.forward through iaf.py:21-23) for a MADE with ONE hidden layer, in one
 * launch: D sequential steps in natural feature order, x_i = (z'_i - t_i(x)) exp(-s_i(x)), z' = flip_if_parity(z),
 * tok_ld = mask * sum_i s_i (the reference accumulates +s here), out = x * mask.
 * Packed weights (fp32, masks already multiplied in, made.py:39-40):  w1t [D][Hp] = (W1 * mask1)^T,
 * b1 [Hp], w2 [2D][Hp] = W2 * mask2, b2 [2D]; hidden units padded with zeros to Hp = 16 * 2^k <= 256.
 * act: 0 linear, 1 gelu (erf), 2 swish, 3 relu, 4 tanh, 5 sigmoid, 6 elu.  Returns MIVAE_EUNSUPPORTED (-3) when
 * mivae_maf_sequential_supported(D, Hp) is 0 (weights + token tiles must fit 144 KiB of LDS). */
int mivae_maf_sequential_supported(int64_t D, int64_t Hp);
int mivae_maf_sequential(const void* z, const float* w1t, const float* b1, const float* w2, const float* b2,
                         const int64_t* mask, void* out, float* tok_ld, int64_t rows, int64_t D, int64_t Hp, int act,
                         int parity, int dtype, void* stream);
/* Backward of mivae_maf_sequential -- autograd of the loop of MAFlow.backward (maf.py:47-62), which is the TRAINING
 * direction of IAFlow (iaf.py:21-23): x = that call's `out`; dout (rows, D) / dtok (rows) = cotangents of out / tok_ld
 * (each may be NULL).  Writes dz (rows, D) and mivae_maf_sequential_bwd_blocks(rows) rows of
 * [dW1t (D, Hp) | dW2 (2D, Hp) | db1 (Hp) | db2 (2D)] fp32 partials in the packed (masked) layout above; sum the rows
 * with mivae_colsum (fixed order).  Supported when the gradient image (3 D Hp floats) fits the LDS. */
int mivae_maf_sequential_bwd_supported(int64_t D, int64_t Hp);
int64_t mivae_maf_sequential_bwd_blocks(int64_t rows);
int mivae_maf_sequential_bwd(const void* x, const void* dout, const float* dtok, const float* w1t, const float* b1,
                             const float* w2, const float* b2, const int64_t* mask, void* dz, float* part, int64_t rows,
                             int64_t D, int64_t Hp, int act, int parity, int dtype, void* stream);

/* ---- NICE stack, wide latents (96 < D/2 <= 256, bf16): MFMA-bound re-tiling of the same path ------------------
 * Same arithmetic and outputs as mivae_nice_stack_fwd / _bwd + mivae_nice_dw (NICECoupling.forward / .backward,
 * couplings.py:141-183; FlowPrior.forward_pass / backward_pass, flow_prior.py:37-70; base log-prob prior.py:148-158)
 * for D/2 in (96, 256], D/2 % 8 == 0, H >= 65, bf16 latents (BASELINE configs C3: D = 256 and C5: D = 512).
 * The weights are packed once per call into per-kernel streams (stream_kind: 0 forward, 1 inverse, 2 backward chain,
 * 3 backward nets, 4 biases) from the same device pointer table as mivae_nice_pack. */
int mivae_nice_big_supported(int64_t h, int64_t H, int dtype);
int64_t mivae_nice_big_pack_bytes(int64_t K, int64_t h, int64_t H, int stream_kind);
int mivae_nice_big_pack(const int64_t* ptr_table, void* out, int64_t K, int64_t h, int64_t H, int stream_kind,
                        void* stream);
/* several streams in one launch: outs[kind] = destination of that stream or NULL (5 entries, host array) */
int mivae_nice_big_pack_multi(const int64_t* ptr_table, void* const* outs, int64_t K, int64_t h, int64_t H, void* stream);
/* z_out, x1_hist (K, N, h) or NULL, tok_val (N) exactly as mivae_nice_stack_fwd; packed_stream = kind 0 (inverse = 0)
 * or kind 1 (inverse = 1), packed_bias = kind 4. */
int mivae_nice_big_fwd(const void* z_in, void* z_out, void* x1_hist, const int64_t* mask, const void* packed_stream,
                       const void* packed_bias, float* tok_val, int64_t Bn, int64_t mask_div, int64_t S, int64_t D,
                       int64_t H, int64_t K, int act, int inverse, int base, float mu, float std, void* stream);
int64_t mivae_nice_big_dw_workspace_floats(int64_t N, int64_t h, int64_t H, int64_t K);
/* autograd of mivae_nice_big_fwd (forward direction) in three launches + a fixed-order slab fold: dz_in (N, D) and the
 * gradients of all 4K Linear layers -- written to dW1 (K,2,H,h), db1 (K,2,H), dW2 (K,2,h,H), db2 (K,2,h), or ADDED
 * into the buffers of grad_ptrs (device int64[K][8], mivae_nice_pack's order) when that is given.  dlogp (Bn) = cotangent
 * of every sentence's sum of tok_val, as mivae_nice_stack_bwd (per-token cotangents: Bn = N, S = 1, mask flattened); ptr_table = the device pointer table given to mivae_nice_big_pack (the weight-gradient kernel reads
 * its stationary operands from the fp32 parameters); dSdT = scratch (K, 2, N, h) in the latent dtype; workspace:
 * mivae_nice_big_dw_workspace_floats.  hidden = scratch of mivae_nice_big_hidden_elems(N, h, H, K) elements of the
 * latent dtype, or NULL: with it the nets launch saves the hidden activations and d pre and the weight gradients are two
 * plain contractions over the tokens; without it they are recomputed (no extra memory, ~2x the weight-gradient time). */
int64_t mivae_nice_big_hidden_elems(int64_t N, int64_t h, int64_t H, int64_t K);
int mivae_nice_big_bwd(const void* z_in, const void* z_out, const void* x1_hist, const int64_t* mask,
                       const int64_t* ptr_table, const void* packed_chain, const void* packed_nets,
                       const void* packed_bias, const void* dz_out,
                       const float* dlogp, void* dz_in, void* dSdT, void* hidden, float* workspace, float* dW1, float* db1,
                       float* dW2, float* db2, const int64_t* grad_ptrs, int64_t Bn, int64_t mask_div, int64_t S,
                       int64_t D, int64_t H, int64_t K, int act, int base, float mu, float std, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIVAE_H */
