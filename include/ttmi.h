/* libttmi - MI355X-native Transformer-Transducer hot path, C ABI.
 *
 * Every entry point: plain pointers and sizes, no torch types; all pointers are
 * DEVICE pointers unless noted; `stream` is a hipStream_t (0 = default stream);
 * nothing is allocated or freed inside (ctx / workspace arenas are caller-provided,
 * sizes via the *_floats / *_bytes queries, 256-byte aligned); launches are
 * asynchronous on `stream`, no device-wide sync.
 * Return value: 0 ok, <0 invalid argument (message in ttmi_last_error()), >0 a
 * hipError_t.  No C++ exception crosses the boundary.
 *
 * The reference (zzpDapeng/Transformer-Transducer) is pure Python and has no FFI
 * layer; each function cites the reference lines whose arithmetic it replaces
 * (paths relative to the reference root).  The Python classes that bind these are
 * in transformer-transducer_amd/{tt,warprnnt_pytorch}; see INTEGRATION.md.
 *
 * Conventions
 *   prec        0 = exact-f32 MFMA (v_mfma_f32_32x32x2_f32), all activations f32:
 *                   the parity path (loss/grads within 1e-4 rel of the reference).
 *               1 = bf16 MFMA, f32 accumulate; activations that feed GEMMs are bf16
 *                   in HBM (q/k/v, attention output, FFN inner, joint hidden, logits),
 *                   residual stream / LayerNorm / softmax / lattice stay f32(/f64).
 *   dropout     p_drop / p_layer = drop probabilities (0 in eval).  Masks are counter-based: element i of site s is kept
 *               iff hash(seed ^ salt_s, i) >= p * 2^32 and scaled by 1/(1-p); backward regenerates them from the same
 *               seed.  Sites (reference modules): attention `drop` (tt/transformer.py:173) salt 0xA1, FFN CoreNet.2 /
 *               CoreNet.4 (:47,49) salts 0xB2 / 0xC3, RelLearnableDecoderLayer.dropout (:196) salt 0xD4 (p_layer,
 *               fused into the FFN's output LayerNorm).  ttmi_dropout_apply exposes the same masks.
 *   g_*         parameter-gradient buffers are ACCUMULATED into (+=): zero them once
 *               per optimiser step (optimizer.zero_grad()).
 *   batch-major activations are [B, L, d] row-major (the reference's [L, B, d] layout is
 *               only a transpose away and no op mixes batch elements).
 */
#ifndef TTMI_H
#define TTMI_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

int ttmi_version(void);
const char* ttmi_last_error(void);   /* thread-local, valid until the next failing call */

/* ---- relative-position self-attention sub-layer ------------------------------------------------------
 * RelLearnableMultiHeadAttn.forward incl. _rel_shift, mask, softmax, o_net, residual + LayerNorm
 * (tt/transformer.py:82-89,106-177).  x,y f32 [B,L,d]; qkv_w [3*H*Dh, d]; o_w [d, H*Dh]; r_emb [K,H,Dh];
 * r_w_bias [H,Dh]; r_bias [K,H].  mask_kind 0 none | 1 causal (tt/utils.py:233-239) | 2 band: masked iff
 * j > i+right or j < i-left (tt/utils.py:242-251) | 3 uint8 tensor, element (b,i,j) at
 * mask[b*mask_sb + i*mask_si + j], nonzero = masked (any [L,L,1] / [L,L,B] / (klen,bsz) mask of
 * tt/transformer.py:154-159 after a permute) | 4 per-row key intervals: `mask` points at int32 pairs, (lo, hi) of query i of batch b at
 * ((const int*)mask)[b*mask_sb + 2*i], key j masked iff j < lo or j > hi (what chunk / band masks are; mask_sb = 0 shares one table);
 * with kind 4, mask_left / mask_right may carry bounds on the intervals' reach from the diagonal (lo_i >= i - left, hi_i <= i + right,
 * -1 / -1 or 0 / 0 = not known).  The fused attention kernels skip key tiles that are masked for a whole query block (kinds 1, 2, 4),
 * and for narrow bands (kind 2, or kind 4 with bounds) the backward also skips the query tiles a key block never meets. */
size_t ttmi_attn_ctx_floats(int B, int L, int d, int H, int Dh, int prec);
size_t ttmi_attn_ws_floats(int B, int L, int d, int H, int Dh, int prec);
int ttmi_attn_fwd(const float* x, const float* qkv_w, const float* o_w, const float* ln_g, const float* ln_b,
                  const float* r_emb, const float* r_w_bias, const float* r_bias, int B, int L, int d, int H, int Dh, int K,
                  int mask_kind, int mask_left, int mask_right, const unsigned char* mask, long mask_sb, long mask_si,
                  int prec, float p_drop, unsigned seed, float* ctx, float* ws, float* y, void* stream);
int ttmi_attn_bwd(const float* dy, const float* x, const float* qkv_w, const float* o_w, const float* ln_g,
                  const float* r_emb, const float* r_w_bias, const float* r_bias, int B, int L, int d, int H, int Dh, int K, int mask_kind,
                  int mask_left, int mask_right, const unsigned char* mask, long mask_sb, long mask_si, int prec,
                  float p_drop, unsigned seed, const float* ctx, float* ws, float* dx, float* g_qkv_w, float* g_o_w,
                  float* g_ln_g, float* g_ln_b, float* g_r_emb, float* g_r_w_bias, float* g_r_bias, void* stream);

/* ---- position-wise feed-forward sub-layer -----------------------------------------------------------------
 * PositionwiseFF.forward: z = LN(y + W2 relu(W1 LN(y) + b1) + b2), ONE LayerNorm used twice
 * (tt/transformer.py:36-58).  rows = B*L; w1 [Di,d]; w2 [d,Di]. */
size_t ttmi_ffn_ctx_floats(long rows, int d, int Di, int prec);
size_t ttmi_ffn_ws_floats(long rows, int d, int Di, int prec);
int ttmi_ffn_fwd(const float* y, const float* w1, const float* b1, const float* w2, const float* b2, const float* ln_g,
                 const float* ln_b, long rows, int d, int Di, int prec, float p_drop, float p_layer, unsigned seed, float* ctx,
                 float* ws, float* z, void* stream);
int ttmi_ffn_bwd(const float* dz, const float* y, const float* w1, const float* w2, const float* ln_g, long rows, int d, int Di,
                 int prec, float p_drop, float p_layer, unsigned seed, const float* ctx, float* ws, float* dy, float* g_w1,
                 float* g_b1, float* g_w2, float* g_b2, float* g_ln_g, float* g_ln_b, void* stream);

/* ---- label-encoder embedding: nn.Embedding(V, d, padding_idx=0) (tt/decoder.py:26,39) ------------------------ */
int ttmi_embed_fwd(const long* tokens, const float* W, long n, int d, int V, float* out, void* stream);
int ttmi_embed_bwd(const long* tokens, const float* dout, long n, int d, int V, int padding_idx, float* gW, void* stream);

/* ---- joint network: JointNet.forward (tt/model.py:20-39) in split-weight form ---------------------------------
 * logits[b,t,u,:] = wp tanh(wf[:, :de] enc[b,t] + wf[:, de:] dec[b,u] + bf) + bp.  enc [B,T,de], dec [B,U1,dd],
 * wf [J, de+dd], wp [V,J].  ttmi_joint_logits_dtype(prec,J): 0 -> logits/dlogits are f32, 1 -> bf16.  Rows
 * (b,t,u) of logits/dlogits have pitch ldv/ldg >= V elements; in the bf16 case the pitch must be a multiple of 8,
 * the base 16-byte aligned, and dlogits must be ZERO in columns [V, ldg) (ttmi_rnnt_loss_bwd guarantees it). */
int ttmi_joint_logits_dtype(int prec, int J);
size_t ttmi_joint_ctx_floats(int B, int T, int U1, int J);
size_t ttmi_joint_ws_floats(int B, int T, int U1, int J, int V);
/* the same for a given precision code: prec 2 (TTMI_PRECISION=bf16x3: the f32 data flow of prec 0 with its large dense products on the bf16
 * MFMA in three terms, hi . hi + lo . hi + hi . lo) needs room for the split operands behind the ordinary workspace */
size_t ttmi_joint_ws_floats_prec(int B, int T, int U1, int J, int V, int prec);
/* ... and ctx: prec 2 keeps the hidden rows as two bf16 blocks [hi | lo] instead of one f32, columns padded to a multiple of 64: callers that
 * pass prec 2 to ttmi_joint_fwd / ttmi_joint_bwd size ctx with this one (the same size as ttmi_joint_ctx_floats when J % 64 == 0) */
size_t ttmi_joint_ctx_floats_prec(int B, int T, int U1, int J, int prec);
int ttmi_joint_fwd(const float* enc, const float* dec, const float* wf, const float* bf, const float* wp, const float* bp,
                   int B, int T, int U1, int de, int dd, int J, int V, int prec, float* ctx, float* ws, void* logits, long ldv,
                   void* stream);
int ttmi_joint_bwd(const void* dlogits, long ldg, const float* enc, const float* dec, const float* wf, const float* wp, int B,
                   int T, int U1, int de, int dd, int J, int V, int prec, const float* ctx, float* ws, float* denc, float* ddec,
                   float* g_wf, float* g_bf, float* g_wp, float* g_bp, void* stream);

/* ---- RNN-T loss: replaces warprnnt_pytorch.RNNTLoss (train.py:13,53,231) --------------------------------------
 * logits [B,T,U1,V] (dtype 0 = f32, 1 = bf16), row pitch ldv; labels i32 [B,U1-1]; act_lens,label_lens i32 [B];
 * costs f32 [B] = -log P(y|x).  bwd: grad = scale * grad_out[b*grad_out_stride] * d costs[b]/d logits in the logits'
 * dtype with row pitch ldg (columns [V,ldg) zeroed); grad may alias logits when ldg == ldv. */
size_t ttmi_rnnt_workspace_bytes(int B, int T, int U1);
int ttmi_rnnt_loss_fwd(const void* logits, int dtype, long ldv, const int* labels, const int* act_lens, const int* label_lens,
                       int B, int T, int U1, int V, int blank, void* workspace, float* costs, void* stream);
int ttmi_rnnt_loss_bwd(const void* logits, int dtype, long ldv, const int* labels, const int* act_lens, const int* label_lens,
                       int B, int T, int U1, int V, int blank, const void* workspace, const float* grad_out,
                       int grad_out_stride, float scale, void* grad, long ldg, void* stream);
/* bf16x3 mode (TTMI_PRECISION=bf16x3, prec 2; no reference counterpart - the reference's loss is warprnnt_pytorch's, tt/model.py:5): the gradient of f32 logits
 * written OVER them as the two bf16 planes the three-term joint backward multiplies - row r becomes [hi(0 .. ldv) | lo(0 .. ldv)], hi = bf16(g), lo = bf16(g - hi),
 * 2 ldv bf16 in the bytes of ldv f32, columns [V, ldv) zero in both planes - so that ttmi_joint_bwd_split need not read d logits again to split them.  Ask the two
 * _ok functions first (rows 16-byte aligned, pitch == roundup(V, 64), the product large enough for the three-term kernels); otherwise use ttmi_rnnt_loss_bwd +
 * ttmi_joint_bwd.  Workspace / grad_out / scale as in ttmi_rnnt_loss_bwd; the other arguments of ttmi_joint_bwd_split as in ttmi_joint_bwd. */
int ttmi_rnnt_loss_bwd_split_ok(long ldv, const void* logits);
int ttmi_rnnt_loss_bwd_split(void* logits, long ldv, const int* labels, const int* act_lens, const int* label_lens, int B, int T, int U1, int V,
                             int blank, const void* workspace, const float* grad_out, int grad_out_stride, float scale, void* stream);
int ttmi_joint_bwd_split_ok(int B, int T, int U1, int J, int V, int prec, long ldg);
int ttmi_joint_bwd_split(const void* dlogits_split, long ldg, const float* enc, const float* dec, const float* wf, const float* wp, int B,
                         int T, int U1, int de, int dd, int J, int V, int prec, const float* ctx, float* ws, float* denc, float* ddec,
                         float* g_wf, float* g_bf, float* g_wp, float* g_bp, void* stream);

/* ---- fused joint + loss fast path ("exp-domain" forms; training-sized bf16 problems, ask ttmi_joint_exp_supported first) --------
 * Replaces the pair JointNet.forward + RNNTLoss (train.py:47-53) when the caller wants the loss only.  The projection GEMM's epilogue
 * stores P[row, v] = bf16(exp(logit - *shift)) (pitch ldv, a multiple of 64, zeros in columns [V, ldv)) and per-row partial sums of the
 * unrounded values (rowsum f32 [nparts, rows], nparts = ttmi_joint_exp_nparts(V)); the loss forward then reads the sums and two entries
 * per row instead of walking the lattice's rows; the loss backward writes per-row factors srow (f32) / srow16 (bf16) and patches the
 * blank / label entries of P so that d logits[r, :] = srow[r] * P[r, :]; ttmi_joint_bwd_exp consumes that product without forming it
 * (row factor in the dgrad epilogue and on the wgrad's activation operand; ctx is overwritten).
 * shift / shift_cur: device scalar (nullable = 0) subtracted before exp; shift_next (device scalar, nullable):
 * max(itself, max_rows(log-sum-exp) - 40), the value to pass as shift on the next step.
 * emis (nullable, f32 [rows, 4], 16-byte aligned): ttmi_joint_fwd_exp also leaves the logits of `blank` and of each row's next label
 * (labels int32 [B, U1-1]) in f32, twice: formed from the bf16 operands the projection multiplies (entries 0, 1: what P and its row
 * sums contain) and from the unrounded hidden row with the f32 weight (entries 2, 3).  ttmi_rnnt_loss_fwd_exp exchanges the two columns'
 * terms of the row sum for the accurate ones and takes the emission log-probs from entries 2, 3: the bf16 rounding of the projection
 * weight is the same in every row, so the blank column's error does not average out along an alignment.
 * flag (nullable, device int): bit 0 is set when a lattice row's sum underflowed or overflowed (the shift no longer fits the logits);
 * the costs and gradients of that step are then NaN (never finite-and-wrong): drop the step, run one step in the plain form and take a new
 * shift from its workspace with ttmi_rnnt_shift_seed (no host synchronisation anywhere in the protocol). */
int ttmi_joint_exp_supported(int B, int T, int U1, int J, int V, int prec, long ldv);
int ttmi_joint_exp_fwd_supported(int B, int T, int U1, int J, int V, int prec, long ldv);     /* forward + loss only (no gradients wanted) */
int ttmi_joint_exp_nparts(int V);
/* lattice rows of a chunk padded to the exp-domain wgrad's 64-row reduction tile: the row count P / srow / srow16 / emis / ctx must have
 * room for (the pad rows get zero row factors and add exact zeros; tt/model.py:20-39 has no counterpart - the reference materialises
 * [B, T, U+1, V] logits of exactly B*T*U1 rows) */
long ttmi_joint_exp_padded_rows(int B, int T, int U1);
int ttmi_joint_fwd_exp(const float* enc, const float* dec, const float* wf, const float* bf, const float* wp, const float* bp,
                       int B, int T, int U1, int de, int dd, int J, int V, int prec, float* ctx, float* ws, void* P, long ldv,
                       float* rowsum, int nparts, const float* shift, const int* labels, int blank, float* emis, void* stream);
int ttmi_rnnt_loss_fwd_exp(const void* P, long ldv, const float* rowsum, int nparts, const int* labels, const int* act_lens,
                           const int* label_lens, int B, int T, int U1, int V, int blank, void* workspace, float* costs,
                           const float* shift_cur, float* shift_next, const float* emis, int* flag, void* stream);
int ttmi_rnnt_shift_seed(const void* workspace /* of a plain ttmi_rnnt_loss_fwd */, const int* act_lens, const int* label_lens, int B, int T,
                         int U1, float* shift_next, void* stream);
int ttmi_rnnt_loss_bwd_exp(void* P, long ldv, const int* labels, const int* act_lens, const int* label_lens, int B, int T, int U1,
                           int V, int blank, const void* workspace, const float* grad_out, int grad_out_stride, float scale,
                           float* srow, void* srow16, void* stream);
int ttmi_joint_bwd_exp(const void* P, long ldg, const float* srow, const void* srow16, const float* enc, const float* dec,
                       const float* wf, const float* wp, int B, int T, int U1, int de, int dd, int J, int V, int prec, float* ctx,
                       float* ws, float* denc, float* ddec, float* g_wf, float* g_bf, float* g_wp, float* g_bp, void* stream);

/* ---- grouped weight gradients (data-parallel hot path): the four weight-gradient GEMMs of an encoder layer are a quarter
 * of the chip each step; deferred and launched four layers at a time they are 256 tiles, one per CU over the whole reduction - no split along
 * K, no atomics, bit-identical from run to run and across ranks.  ttmi_attn_bwd_defer / ttmi_ffn_bwd_defer are ttmi_attn_bwd / ttmi_ffn_bwd
 * minus those GEMMs: the bf16 operands that would have lived in the scratch workspace go to `keep` (ttmi_*_keep_bytes; 256-byte aligned,
 * caller-owned until the group has run) and `out` receives the problems (2 per call) for ttmi_wgrad_group.  bf16 pipeline only (returns an
 * error otherwise: callers test ttmi_wgrad_defer_supported first). */
typedef struct ttmi_wgrad_desc {
    const void* A;      /* bf16 [K, M], row pitch lda */
    const void* B;      /* bf16 [K, N], row pitch ldb */
    float* C;           /* f32 [M, N], row pitch ldc: C += A^T B */
    float* colsum;      /* nullable, f32 [M]: += column sums of A */
    int M, N, K;
    long lda, ldb, ldc;
} ttmi_wgrad_desc;
int ttmi_wgrad_group(const ttmi_wgrad_desc* descs /* host array */, int n, void* stream);
int ttmi_wgrad_defer_supported(long rows, int d, int H, int Dh, int Di, int prec);
size_t ttmi_attn_bwd_keep_bytes(int B, int L, int d, int H, int Dh);
size_t ttmi_ffn_bwd_keep_bytes(long rows, int d, int Di);
int ttmi_attn_bwd_defer(const float* dy, const float* x, const float* qkv_w, const float* o_w, const float* ln_g,
                        const float* r_emb, const float* r_w_bias, const float* r_bias, int B, int L, int d, int H, int Dh, int K, int mask_kind,
                        int mask_left, int mask_right, const unsigned char* mask, long mask_sb, long mask_si, int prec,
                        float p_drop, unsigned seed, const float* ctx, float* ws, float* dx, float* g_qkv_w, float* g_o_w,
                        float* g_ln_g, float* g_ln_b, float* g_r_emb, float* g_r_w_bias, float* g_r_bias, void* keep,
                        ttmi_wgrad_desc* out /* 2 entries */, void* stream);
int ttmi_ffn_bwd_defer(const float* dz, const float* y, const float* w1, const float* w2, const float* ln_g, long rows, int d, int Di,
                       int prec, float p_drop, float p_layer, unsigned seed, const float* ctx, float* ws, float* dy, float* g_w1,
                       float* g_b1, float* g_w2, float* g_b2, float* g_ln_g, float* g_ln_b, void* keep,
                       ttmi_wgrad_desc* out /* 2 entries */, void* stream);

/* ---- one encoder layer per call: RelLearnableDecoderLayer.forward (tt/transformer.py:188-197) = ttmi_attn_fwd + ttmi_ffn_fwd, and
 * their backward, behind ONE entry each.  Where ttmi_layer_fused() is 1 (bf16 pipeline, d % 4 == 0, d <= 512) the calls share three passes
 * the sub-layer boundary forces apart: the FFN's pre-norm comes out of the attention sub-layer's post-norm pass, the two LayerNorm backward
 * passes that meet at y are one kernel (dy is never stored), and the bf16 copy of the layer's input is taken from its producer (x16_in: bf16
 * [B*L, d] or NULL; z16_out: bf16 [B*L, d] or NULL - hand a layer's z16_out to the next layer as x16_in, to forward AND backward).
 * Otherwise they run the sub-layer calls back to back.  ctx_attn / ctx_ffn: ttmi_attn_ctx_floats / ttmi_ffn_ctx_floats, saved for backward;
 * ws: ttmi_layer_ws_floats; y ([B, L, d], the attention sub-layer's output) is an output of forward and an input of backward.
 * ttmi_layer_bwd with keep_attn / keep_ffn / out (ttmi_*_bwd_keep_bytes; 4 descriptors: CoreNet.3, CoreNet.0, o_net, qkv_net) defers the
 * weight-gradient GEMMs as ttmi_*_bwd_defer do; all three NULL runs them here. */
int ttmi_layer_fused(int d, int H, int Dh, int Di, int prec);
size_t ttmi_layer_ws_floats(int B, int L, int d, int H, int Dh, int Di, int prec);
int ttmi_layer_fwd(const float* x, const void* x16_in, const float* qkv_w, const float* o_w, const float* ln_g, const float* ln_b, const float* r_emb,
                   const float* r_w_bias, const float* r_bias, const float* w1, const float* b1, const float* w2, const float* b2,
                   const float* ff_ln_g, const float* ff_ln_b, int B, int L, int d, int H, int Dh, int K, int Di, int mask_kind, int mask_left,
                   int mask_right, const unsigned char* mask, long mask_sb, long mask_si, int prec, float p_drop_attn, unsigned seed_attn,
                   float p_drop_ffn, float p_layer, unsigned seed_ffn, float* ctx_attn, float* ctx_ffn, float* ws, float* y, float* z,
                   void* z16_out, void* stream);
int ttmi_layer_bwd(const float* dz, const float* x, const void* x16_in, const float* y, const float* qkv_w, const float* o_w, const float* ln_g,
                   const float* r_emb, const float* r_w_bias, const float* r_bias, const float* w1, const float* w2, const float* ff_ln_g, int B,
                   int L, int d, int H, int Dh, int K, int Di, int mask_kind, int mask_left, int mask_right, const unsigned char* mask,
                   long mask_sb, long mask_si, int prec, float p_drop_attn, unsigned seed_attn, float p_drop_ffn, float p_layer,
                   unsigned seed_ffn, const float* ctx_attn, const float* ctx_ffn, float* ws, float* dx, float* g_qkv_w, float* g_o_w,
                   float* g_ln_g, float* g_ln_b, float* g_r_emb, float* g_r_w_bias, float* g_r_bias, float* g_w1, float* g_b1, float* g_w2,
                   float* g_b2, float* g_ff_ln_g, float* g_ff_ln_b, void* keep_attn, void* keep_ffn, ttmi_wgrad_desc* out /* 4 entries */,
                   void* stream);

/* ---- greedy decoding support (Transducer.decode, tt/model.py:70-90): logits rows = consecutive frames against one label
 * state; *out (device u64) = (first row whose argmax != blank) << 32 | symbol, or n << 32 if all rows are blank. */
int ttmi_greedy_scan(const void* logits, int dtype, long ld, int n, int V, int blank, unsigned long long* out, void* stream);
/* Batched greedy decoding (tt/model.py:92-108 recognize -> decode for every utterance), all utterances of a batch in lockstep over SYMBOL steps:
 * after step s every utterance still decoding holds exactly s + 1 tokens, so one label-encoder call of length s + 1 serves the batch exactly.
 * ttmi_greedy_scan_batch: logits [B, n, V] (row pitch ld) = the joint of frames t[b] .. t[b] + n - 1 of every utterance against its own label state;
 * key[b] (device u64, n << 32 before the call) = min over the utterance's frames that exist (t[b] + r < T_len[b]) and whose argmax is not `blank`
 * of (r << 32 | symbol); utterances with need[b] == 0 are left alone.  ttmi_greedy_advance consumes key: symbol found -> hist[b][n_hist] =
 * symbol, t[b] += r + 1, count[b] += 1, need[b] = 0; none -> t[b] += n, and need[b] = 0, done[b] = 1 once t[b] >= T_len[b]; key is reset;
 * flags[0] = utterances that still need a symbol in this step, flags[1] = utterances not finished (one 8-byte read per scan for the host). */
int ttmi_greedy_scan_batch(const void* logits, int dtype, long ld, int B, int n, int V, int blank, const int* t, const int* T_len,
                           const int* need, unsigned long long* key, void* stream);
int ttmi_greedy_advance(unsigned long long* key, int B, int n, int n_hist, long* hist, long ld_hist, int* t, const int* T_len, int* need,
                        int* done, int* count, int* flags, void* stream);

/* ---- feature front-end on the GPU: replaces the data loader's per-utterance numpy code.
 * ttmi_logmel: get_feature / get_feature2 (tt/utils.py:182-207: librosa.feature.melspectrogram(y, sr, n_fft=512, hop_length=160, n_mels), then
 * log) for a batch.  wave i16 [B, pitch >= nmax] zero padded, n_samples i32 [B] (device) -> out f32 [B, Fmax, n_mels], Fmax = 1 + nmax / hop,
 * frames beyond 1 + n_b / hop zero.  The STFT is a GEMM with `dft` [2*(n_fft/2+1), n_fft] (rows 2k / 2k+1 = w[n] cos(2 pi k n / n_fft) /
 * -w[n] sin(...), analysis window w folded in), the filterbank a GEMM with mel_w [n_mels, n_fft/2+1]; log_mode 0 = np.ma.log(..).filled(0),
 * 1 = log10 with zeros -> float64 eps, 2 = none.  ws: ttmi_logmel_ws_floats floats, 256-byte aligned.
 * ttmi_stack_subsample: concat_frame + subsampling + Dataset.pad (tt/utils.py:120-151, tt/dataset.py:40-57) in one pass: feat f32 [B, Tin, F],
 * n_frames i32 [B] (device, nullable) -> out f32 [B, Tout, F*(1+left+right)], rows >= ceil(n_b / subsample) zero; out_lens nullable.
 * ttmi_spec_mask: frequency_mask_augment + time_mask_augment (tt/utils.py:297-329, train.py:41-44) in place on x f32 [B, T, F]; the (start,
 * width) pairs are HOST arrays (<= 32 per axis), drawn by the caller with the reference's RNG protocol. */
size_t ttmi_logmel_ws_floats(int B, int nmax, int n_fft, int hop);
int ttmi_logmel(const short* wave, long pitch, const int* n_samples, int B, int nmax, int n_fft, int hop, int n_mels, const float* dft,
                const float* mel_w, int log_mode, float* ws, float* out, void* stream);
int ttmi_stack_subsample(const float* feat, const int* n_frames, int B, int Tin, int F, int left, int right, int subsample, int Tout,
                         float* out, int* out_lens, void* stream);
int ttmi_spec_mask(float* x, int B, int T, int F, const int* time_spans, int n_time, const int* freq_spans, int n_freq, void* stream);

/* ---- training-step tail on flat f32 buffers: clip_grad_norm_ + optimizer.step (train.py:62-65, tt/optim.py:57-73)
 * normsq: device scalar holding sum(g^2) over ALL gradients (ttmi_sumsq accumulates into it); NULL = no clipping and no drop.
 * The effective gradient is g * grad_scale (1/world_size after a SUM all-reduce) clipped to max_norm (max_norm <= 0: not clipped).
 * A step whose *normsq is inf / NaN is DROPPED (parameters and state untouched) whatever max_norm is: the exp-domain loss form fails with
 * NaN costs and gradients by construction, and such a step must not reach the weights.
 * hyper (device float[3], nullable): the values that change between steps, read by the kernels at RUN time so that a step captured into a
 * HIP graph follows them - hyper[0] = learning rate (replaces `lr`: tt/optim.py:30-33 decay_lr, train.py:257), hyper[1] = optimiser steps
 * TAKEN, hyper[2] = steps DROPPED; every *_step call with hyper != NULL first advances hyper[1] by one on the device - or, when *normsq is
 * not finite, hyper[2]: a dropped step consumes no bias-correction step (Adam's corrections 1 - beta^hyper[1] replace the host's `step`). */
int ttmi_sumsq(const float* x, long n, float* out, void* stream);
int ttmi_sgd_step(float* p, const float* g, float* mom, long n, float lr, float momentum, float weight_decay, int nesterov,
                  float max_norm, const float* normsq, float grad_scale, float* hyper, void* stream);
int ttmi_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float max_norm, const float* normsq, float grad_scale, float* hyper, void* stream);
/* torch.optim.Adadelta(lr, rho, eps, weight_decay), the third type of tt/optim.py:74-81 */
int ttmi_adadelta_step(float* p, const float* g, float* square_avg, float* acc_delta, long n, float lr, float rho, float eps,
                       float weight_decay, float max_norm, const float* normsq, float grad_scale, float* hyper, void* stream);

/* Mark a stream that is itself forked from the caller's main stream (the label encoder's side stream, tt/model.py _encode): calls on it never
 * fork onto the library's own side streams inside a stream capture (no reference counterpart; ttmi_set_option(18, 1) lets the capturing stream
 * itself fork as in eager mode). */
int ttmi_stream_set_nofork(void* stream, int on);

/* Device word (nullable) mixed into every dropout seed when a kernel starts.  Seeds are drawn on the host per sub-layer call; in a step that
 * is captured as a HIP graph they are baked into the kernel arguments, so the caller bumps this word on the device before each replay
 * (ttmi.train.GraphedStep) and every step still draws fresh masks.  Forward and backward of one step must see the same value. */
int ttmi_set_dropout_salt(const unsigned* salt);

/* ---- bring-up / measurement helpers ------------------------------------------------------------------------------
 * generic MFMA GEMM (every layout / dtype / epilogue; flags = GemmFlags of csrc/gemm.h) and the two throughput
 * kernels; HIP-event probes recorded on the launch stream at five points (0 = joint vocabulary projection, 1 = ttmi_rnnt_loss_fwd's
 * kernels, 2 = ttmi_rnnt_loss_bwd's kernel, 3 = the fused attention backward kernel of a layer with L >= 256, 4 = a grouped weight-gradient launch of
 * ttmi_wgrad_group that fills at least half the chip): ttmi_probe_arm(i), 0 <= i < 64, makes the NEXT launch at every point record into its event
 * pair i; ttmi_probe_point_read_ms(point, i) waits for that pair and returns its duration in ms (< 0: never fired);
 * ttmi_probe_read_ms(i) = point 0. */
int ttmi_gemm(const void* A, const void* B, void* C, const float* bias, const float* aux, int a_dtype, int b_dtype,
              int c_dtype, int M, int N, int K, long lda, long ldb, long ldc, int nz1, int nz2, long sA1, long sA2,
              long sB1, long sB2, long sC1, long sC2, float alpha, float beta, int flags, int splitk, void* stream);
int ttmi_gemm_nt_bf16(const void* A, const void* B, void* C, int c_dtype, const float* bias, int M, int N, int K, long lda,
                      long ldb, long ldc, void* stream);
/* bring-up entry of the two-term weight form (option 13): C = epi(A.(B + B_lo)^T), B_lo [N, K] bf16 with B's pitch, in ONE launch - the persistent
 * kernels walk A's K-tiles a second time against B_lo, the 128x128 kernel takes (A, B_lo) as its second operand pair */
int ttmi_gemm_nt_bf16_two_term(const void* A, const void* B, const void* B_lo, void* C, int c_dtype, const float* bias, int relu, int M, int N, int K,
                               long lda, long ldb, long ldc, void* stream);
/* ... with the second walk limited to A's first k_lo columns (0 = all K): C = act(A.B^T + A[:, :k_lo].B_lo[:, :k_lo]^T + bias) - the layout of the
 * bf16x3 products, A = [hi | lo], B = [hi | hi], B_lo = lo (k_lo = K / 2) */
int ttmi_gemm_nt_bf16_two_term_klo(const void* A, const void* B, const void* B_lo, void* C, int c_dtype, const float* bias, int relu, int M, int N, int K,
                                   int k_lo, long lda, long ldb, long ldc, void* stream);
/* bring-up entry of the "exp store" epilogue of the persistent 256x256 kernel: C = bf16(exp(A.B^T + bias - shift)) with zeros in columns [N, ldc),
 * rowsum [nparts, M] = per-row partial sums (nparts >= 4 * ceil(N / 256); the entries of a row add up to its sum of exponentials) */
int ttmi_gemm_nt_bf16_exp(const void* A, const void* B, void* C, const float* bias, float* rowsum, int nparts, const float* shift /* device, nullable */, int M, int N, int K,
                          long lda, long ldb, long ldc, void* stream);
/* bring-up entry of the "row factor" epilogue of the persistent 256x256 kernel (the joint's dgrad in the exp-domain loss form, the adjoint of
 * tt/model.py:34-38 with the loss gradient kept factored): C = bf16((A.B^T) * (1 - mask^2) * rowscale[m]) and, in place, mask <- rowscale[m] * mask
 * (mask bf16 [M, ldc], the layout of C; rowscale f32 [M]) */
int ttmi_gemm_nt_bf16_rowscale(const void* A, const void* B, void* C, void* mask, const float* rowscale, int M, int N, int K, long lda, long ldb, long ldc,
                               void* stream);
int ttmi_gemm_tn_bf16(const void* A, const void* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, int accumulate,
                      float* colsum_a /* nullable: colsum_a[m] += sum_k A[k][m] */, void* stream);
/* bring-up entry of the weighted column sums (persistent 256x256 TN kernel only: K >= 32768, M >= 1024, N % 256 == 0, N >= 1024):
 * C += A^T B and colsum_a[m] += sum_k colsum_w[k] A[k][m] (colsum_w bf16 [K], 16-byte aligned) */
int ttmi_gemm_tn_bf16_wsum(const void* A, const void* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, float* colsum_a,
                           const void* colsum_w, void* stream);
/* bf16 shadows of GEMM weights.  Every forward call otherwise converts its f32 master weights to bf16 (plain + transposed copies: 118
 * launches per step at C2 for weights that change once per optimiser step).  A training loop registers, per weight w [R, C], a plain bf16
 * copy w16 [R, C] and a transposed one wT16 [C, ldT] (ldT >= R, columns [R, ldT) zero) and rebuilds ALL of them with one launch after each
 * update (ttmi_weight_shadow_refresh: `table` = device array of n rows of 8 longs (w, R, C, wT16, ldT, w16, first 32x32 tile, tiles along C),
 * sorted by first tile; total_tiles = their sum).  The sub-layer calls look a shadow up by the f32 pointer and use it when its geometry
 * matches, else convert as before.  The caller owns the memory and the freshness (ttmi.train.FlatModel). */
int ttmi_weight_shadow_register(const float* w, int R, int C, const void* w16, const void* wT16, long ldT);
int ttmi_weight_shadow_clear(const float* w /* NULL = all */);
int ttmi_weight_shadow_refresh(const long* table, int n, long total_tiles, void* stream);
/* round 6: the second term of a shadowed weight's bf16 split, w16lo = bf16(w - float(w16)) (the operand of ttmi_set_option(13, ...)), kept with the shadows: one
 * buffer of twice the plain copies' size, w16lo at w16 + lo_delta elements; ttmi_weight_shadow_refresh_lo rebuilds all three copies in its one launch.  Without
 * it every forward call that takes the second term forms it itself (one small launch per GEMM). */
int ttmi_weight_shadow_register_lo(const float* w, const void* w16lo);
int ttmi_weight_shadow_refresh_lo(const long* table, int n, long total_tiles, long lo_delta, void* stream);

/* data-parallel runs (train.py:55-56,214-219 replaced by one process per GPU + RCCL): the gradient all-reduce kernels run beside the
 * backward pass; the encoder-sized persistent GEMMs launched on `stream` (and on the library's fork streams serving it) leave n CUs to
 * them.  Per-stream state read at launch time; 0 = whole chip (default). */
int ttmi_stream_reserve_cus(void* stream, int n);
/* process-wide A/B switches for MEASUREMENTS ONLY (not thread-safe against concurrent launches, no product path depends on them): key 0: 1 = no fused attention kernels; 1: throughput-GEMM generation (csrc/gemm_fast.hip; 5 = nothing persistent: every eligible NT problem on 64x64 tiles; 14 / 15 = streaming output stores off / on; 16 + n = the 64x64-tile bf16 NT kernel from n of its tiles on, 16 = never, default 48);
 * 2: flash-kernel timing bits; 3: 0 = no side-stream wgrad fork; 4: split-K workgroup target; 5: 1 = position-term slab by batched GEMM;
 * 6: process-wide default of ttmi_stream_reserve_cus for streams that never set one;
 * 7: exact-f32 products with at most n rows use the skinny 32x32 split-reduction kernel (default 128, 0 = never: greedy decode A/B);
 * 8: 0 = the fused attention kernels read the position term from a [B,H,L,L+1] bf16 slab (round-1 design) instead of forming it themselves;
 * 9: 1 = one-wave lattice kernel (round 2) instead of the workgroup-per-utterance one; 10: batch slices of the attention backward; 11: 1 = dq / dE /
 * dc by the round-2 GEMM launches instead of attn_dqde_kernel; 12: workgroups of the grid-stride LayerNorm backward kernels; 13 (default 2 since round 6): 1 = the four
 * forward GEMMs of an encoder layer (qkv_net, o_net, CoreNet.0, CoreNet.3) take the second term of their weight's bf16 split as a second K range, one launch each; 2 = o_net and
 * CoreNet.3 only, in stacks of at least 4096 rows (the audio encoder; the label states' value has its own pass); + 4 = all four in stacks of fewer than 4096 rows (the label encoder); 0 = off (2: +0.4 ms per C2 step - with the label encoder's value pass
 * of tt.model the timed mode's batch-mean loss stays within 8.1e-5 of the fp32 mode over 56 training states, profiles/r06_loss_error_batch_mean_fixed.log); 14: 0 = the tiled attention
 * forward kernel instead of the one-workgroup-per-head one; 15: 1 = the round-3 attention backward kernel instead of flash_bwd_rel2_kernel;
 * 16: 1 = the position-table gradients go through dE / dc and a relpos_scatter launch (round 3) instead of straight out of attn_dqde_kernel;
 * 17: exact-f32 NT products: 0 = the kernels of csrc/gemm.hip only (round 1), 1 = default rule (persistent 256x128 kernel with f32 operands from 512 of its tiles on,
 * 64x64 tiles from 72 of those on), 2 / 3 = the 64x64-tile / the persistent kernel wherever it can run; 18: 1 = fork inside a stream capture;
 * 19: 0 = the joint's input layer (bf16 mode) without the second bf16 term of the label-encoder states (round 6: that rounding is one pattern in all T lattice rows of a label
 * position and was the joint's whole share of the batch-mean loss error; A/B);
 * 20: 0 = grouped weight gradients never cut an XCD's surplus tiles into K-pieces (default 1: under a CU reservation - ttmi_stream_reserve_cus - 256 tiles on 224 ... 248
 * workgroups end with one atomically added piece per workgroup instead of a second round; without a reservation the launches stay free of atomics either way);
 * 21: 0 = the joint's sums over frames (dPD) by f32 atomics as in rounds 1 - 5 (default 1: partial rows + an ordered second pass - the same bits in every run, same time);
 * 22: 0 = bf16x3 weight gradients as three accumulating launches (round 5) instead of one launch over the plane pairs + a fold */
int ttmi_set_option(int key, int value);
int ttmi_dropout_apply(const float* in, long n, float p, unsigned seed, float* out, void* stream);
int ttmi_probe_arm(int slot);
float ttmi_probe_read_ms(int slot);
float ttmi_probe_point_read_ms(int point, int slot);

#ifdef __cplusplus
}
#endif
#endif
