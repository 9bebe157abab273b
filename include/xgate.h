/*
 * xgate.h -- C ABI of libxgate_hip.so: the MI355X (gfx950) implementation of the
 * gated-fusion caption decoder hot path of vsislab/Controllable_XGating.
 *
 * The reference has no FFI / operator layer: its hot path sits behind the
 * Python class SAModel (reference caption_src/SAModel.py:13-219).  This header is the
 * boundary that sits UNDER that class surface: every entry point below replaces the stock
 * PyTorch kernels the reference launches from the cited lines.  A reference maintainer
 * binds it with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - All tensor pointers are DEVICE pointers (HIP), fp32 unless noted, row-major,
 *     contiguous in the documented shape.  Token / index tensors are int64.
 *   - The caller owns all memory; scratch and saved activations live in one caller-provided
 *     workspace sized by xg_workspace_bytes().  The only thing the library allocates is the
 *     optional side-stream handle of xg_aux_create (two HIP streams + events, no memory).
 *     A workspace must be ZERO-FILLED once after allocation (hipMemset), before its first use: it
 *     also holds the inter-workgroup synchronisation words of the step kernels, which every call
 *     leaves at zero again.
 *   - No library-global mutable state: arithmetic mode, packed weights, side streams and
 *     data-parallel events all travel in XgRun.
 *     The workspace written by a *_fwd call must be handed unchanged to the matching *_bwd.
 *   - `stream` is a hipStream_t passed as void*.  Every entry point only ENQUEUES work on
 *     that stream and returns; there is no host synchronisation inside the library.
 *   - Return value: 0 on success, a negative XG_E* code otherwise (xg_strerror()).
 *   - Re-entrant across devices / streams; not across concurrent calls sharing a workspace.
 *
 * Dimension names: B batch, K frames, R rnn_size, A att_size, E input_encoding_size,
 * V vocab, C categories, H classifier hidden (128), F1/F2 rgb/opfl feature sizes,
 * T decoder steps of this call (seq.size(1) for XE, seq_length+1 for rollouts).
 */
#ifndef XGATE_H
#define XGATE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XG_VERSION 206   /* 200: XgRun gained packed/aux/event fields, xg_set_grad_event removed, xg_vproj takes XgRun;
                            201: + xg_clip_adam_zero; 202: XgRun.prof_event0/1, xg_adam_tick, xg_clip_adam_dev;
                            203: + xg_rollout_pair_compact; 204: + xg_abi_check;
                            205: + xg_rollout_pair_videos, xg_workspace_bytes_mode; 206: + xg_pack_weights_part */

enum {
    XG_OK = 0,
    XG_EINVAL = -1,     /* bad dimension / null pointer / misaligned pointer */
    XG_EARCH = -2,      /* device is not gfx950 */
    XG_EHIP = -3,       /* a HIP call or kernel launch failed (hipGetLastError) */
    XG_EWORKSPACE = -4  /* workspace too small */
};

typedef struct XgDims {
    int32_t B, K, R, A, E, V, C, H, F1, F2, T;
} XgDims;

/*
 * Parameter block: one device pointer per state_dict entry of the reference model, in the
 * order of SURVEY.md Appendix B (reference caption_src/SAModel.py:32-49,
 * caption_src/sub_modules.py:94-110,661-669,741-745).  The same struct type carries the
 * gradients (each pointer then addresses the .grad buffer of that parameter).  Field order
 * is part of the ABI; xg_param_name(i) returns the state_dict key of field i.
 */
typedef struct XgParams {
    float *emb_rgb_w, *emb_rgb_b, *bn_rgb_g, *bn_rgb_b;       /* two_spatial_encoder.visual_emb_rgb.{0,1} */
    float *emb_opfl_w, *emb_opfl_b, *bn_opfl_g, *bn_opfl_b;   /* two_spatial_encoder.visual_emb_opfl.{0,1} */
    float *lstm_rgb_wih, *lstm_rgb_whh, *lstm_rgb_bih, *lstm_rgb_bhh;     /* lstmcell_rgb  (gate order i,f,g,o) */
    float *lstm_opfl_wih, *lstm_opfl_whh, *lstm_opfl_bih, *lstm_opfl_bhh; /* lstmcell_opfl */
    float *gate_rgb_w, *gate_rgb_b, *gate_opfl_w, *gate_opfl_b;           /* gate_{rgb,opfl}.gate.0 */
    float *fusion_w, *fusion_b;                                           /* fusion.late_fusion.0 */
    float *ih1_w, *ih1_b, *ic1_w, *ic1_b, *ih2_w, *ih2_b, *ic2_w, *ic2_b; /* img_embed_{h_1,c_1,h_2,c_2} */
    float *dgate_w, *dgate_b;                                             /* lstmcore.gate.gate.0 */
    float *l1_i2h_w, *l1_i2h_b, *l1_a2h_w, *l1_a2h_b, *l1_h2h_w, *l1_h2h_b; /* lstmcore.lstm_1 (order i,f,o,g) */
    float *l2_i2h_w, *l2_i2h_b, *l2_a2h_w, *l2_a2h_b, *l2_h2h_w, *l2_h2h_b; /* lstmcore.lstm_2 */
    float *v2a_w, *v2a_b, *h2a_w, *h2a_b, *a2w_w, *a2w_b;                 /* lstmcore.{v2a,h2a,a2w} */
    float *embed_w;                                                       /* embed.weight */
    float *logit_w, *logit_b;                                             /* logit */
    float *cls0_w, *cls0_b, *cls3_w, *cls3_b;                             /* classifer.{0,3} */
} XgParams;

/* BatchNorm running statistics (buffers of visual_emb_{rgb,opfl}.1); updated in train mode. */
typedef struct XgBnState {
    float *rgb_mean, *rgb_var, *opfl_mean, *opfl_var;
} XgBnState;

/* One batch of inputs (reference caption_src/starttrain.py:114-121). */
typedef struct XgBatch {
    const float *feats_rgb;   /* (B,K,F1) */
    const float *feats_opfl;  /* (B,K,F2) */
    const float *feat_mask;   /* (B,K) 0/1 */
    const float *pos_feats;   /* (B,R) */
    const int64_t *seq;       /* (B,T) teacher-forced tokens, col 0 = BOS = 0; NULL for rollouts */
    const float *seq_mask;    /* (B,T) 0/1; NULL for rollouts */
} XgBatch;

/* Run-time options. */
typedef struct XgRun {
    int32_t train;        /* 1: BatchNorm batch statistics + dropout active (model.train()) */
    float drop_p;         /* drop_prob_lm; 0 disables dropout */
    uint32_t seed;        /* dropout stream for this call (integer hash, see csrc/xg_common.h) */
    int32_t save;         /* 1: keep activations in the workspace for a following *_bwd */
    float bn_momentum;    /* 0.1 */
    float bn_eps;         /* 1e-5 */
    int32_t gemm_mode;    /* arithmetic of the products: 0 = fp32 MFMA, exact fp32 (default); 3 = split-bf16 (three bf16 planes,
                             6 MFMAs per 16-deep block, fp32-class accuracy) for the large AND the per-step products;
                             1 = bf16 operands, fp32 accumulate, for the large AND the per-step products
                             (BASELINE.json configs[4], tolerance 1e-2).  Accumulation, the cell arithmetic, the
                             attention and every reduction are fp32 in all modes. */
    int32_t packed_dtype; /* element type of `packed`: 0 = fp32 tiles (gemm_mode 0 / 3), 1 = bf16 tiles (gemm_mode 1),
                             2 = three pre-split bf16 planes (gemm_mode 3 only: same results as 0 there, fewer instructions) */
    const void *packed;   /* optional: the recurrent weights in MFMA-fragment order (xg_pack_weights), valid for the
                             CURRENT parameter values; NULL (or a dtype that does not fit gemm_mode) = stream the plain
                             weights through LDS.  Same results (bf16: the same rounding, done once instead of per pass). */
    void *aux;            /* optional: handle from xg_aux_create -- side streams on which the entry points overlap work that
                             nothing downstream waits for; everything is joined back onto `stream` before the call
                             returns.  NULL = one stream.  One handle per caller stream in use at a time. */
    void *grad_event;     /* optional hipEvent_t (data parallel): backward entry points record it at the moment
                             every gradient except two_spatial_encoder.* is final, so that a caller who keeps the gradients
                             in xg_param_name order can start the RCCL all-reduce of that suffix under the CG encoder's
                             backward.  NULL = not recorded. */
    void *grad_event_head;/* optional hipEvent_t: recorded earlier still, when logit.weight / logit.bias gradients (a third of
                             all gradient bytes at V = 20000) are final AND logit.* is no longer read by this backward --
                             before the reverse-time decoder loop starts.  From grad_event on, no parameter outside
                             two_spatial_encoder.* is read either: a caller may start its optimizer update of those groups
                             behind the events (train.ClipAdam(overlap=True)). */
    void *prof_event0, *prof_event1; /* optional hipEvent_t pair (created WITH timing): the forward entry points that run the
                             decoder's time loop (xg_forward_xe, xg_xe_loss_fwd, xg_rollout, xg_rollout_pair) record them on
                             `stream` right before the first and right after the last decoder step, so a caller can read the
                             in-situ duration of the T steps (bench.py: roofline.in_situ_us_per_step).  NULL = nothing recorded. */
} XgRun;

enum { XG_ROLLOUT_GREEDY = 0, XG_ROLLOUT_SAMPLE = 1, XG_ROLLOUT_REPLAY = 2 };

/* ---- library / ABI introspection ------------------------------------------------- */
int xg_version(void);
/* A binding checks ITS struct layouts against the library's before the first real call: pass XG_VERSION as compiled into the
 * binding and sizeof of its XgDims, XgParams, XgBnState, XgBatch, XgRun.  XG_OK if all six agree with the library, XG_EINVAL
 * otherwise (a stale stub -- e.g. an XgRun without the newest trailing fields -- would make the library read past the
 * caller's struct).  Needs no GPU. */
int xg_abi_check(int version, size_t sz_dims, size_t sz_params, size_t sz_bn, size_t sz_batch, size_t sz_run);
const char *xg_strerror(int code);
int xg_param_count(void);
const char *xg_param_name(int index);                 /* state_dict key of XgParams field `index` */
int xg_param_numel(const XgDims *d, int index, int64_t *numel);
size_t xg_workspace_bytes(const XgDims *d);           /* covers every entry point below for dims d, in every gemm_mode */
/* the same for ONE arithmetic mode: only gemm_mode 1 (bf16) uses the bf16 mirror region, a third of xg_workspace_bytes; a
 * workspace of this size is accepted by every entry point (gemm_mode 1 on a mirror-less workspace converts operands on the fly) */
size_t xg_workspace_bytes_mode(const XgDims *d, int gemm_mode);
/* zero-fills a freshly allocated workspace (the one-time initialisation the conventions above ask for; a hipMemsetAsync) */
int xg_workspace_init(void *stream, void *ws, size_t ws_bytes);

/* ---- building block: fp32 GEMM on MFMA -------------------------------------------
 * C[M,N] = op(A) * op(B) (+ bias[n]) (+ C if accumulate), optional ReLU.
 *   transA = 0: A is (M,K) row-major lda;  transA = 1: A is (K,M) row-major lda.
 *   transB = 0: B is (K,N) row-major ldb;  transB = 1: B is (N,K) row-major ldb  (nn.Linear weight).
 * Replaces the stock GEMMs behind nn.Linear (e.g. caption_src/SAModel.py:109). */
int xg_gemm(void *stream, int transA, int transB, int M, int N, int K,
            const float *A, int lda, const float *B, int ldb, float *C, int ldc,
            const float *bias, int relu, int accumulate);
/* same product with the arithmetic of XgRun.gemm_mode (0, 1 or 3) */
int xg_gemm_mode(void *stream, int mode, int transA, int transB, int M, int N, int K,
                 const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                 const float *bias, int relu, int accumulate);

/* bf16 arithmetic (gemm_mode 1) with operands that ALREADY ARE bf16 in memory: A16 / B16 are optional bf16 copies of A / B (same
 * shape, layout and leading dimension; 16-byte aligned; NULL = convert the fp32 operand on the fly).  Where a copy exists the
 * kernel moves half the operand bytes.  xg_cvt_bf16 makes such a copy (round to nearest even).  BASELINE.json configs[4]. */
int xg_gemm_bf16_operands(void *stream, int transA, int transB, int M, int N, int K, const float *A, const void *A16, int lda,
                          const float *B, const void *B16, int ldb, float *C, int ldc, const float *bias, int relu, int accumulate);
int xg_cvt_bf16(void *stream, const float *src, void *dst /* bf16 */, int64_t n);

/* ---- CG encoder: EncoderLstm_two_fc.forward (caption_src/sub_modules.py:118-159) ---- */
int xg_encoder_fwd(void *stream, const XgDims *d, const XgParams *p, const XgBnState *bn,
                   const XgBatch *x, const XgRun *run, void *ws, size_t ws_bytes,
                   float *V /* out (B,K,R) */);
/* dV: (B,K,R) gradient wrt the encoder output; parameter gradients are ACCUMULATED into g. */
int xg_encoder_bwd(void *stream, const XgDims *d, const XgParams *p, const XgParams *g,
                   const XgBatch *x, const XgRun *run, void *ws, size_t ws_bytes,
                   const float *dV);

/* ---- decoder init: SAModel.init_hidden (caption_src/SAModel.py:58-65) ----
 * state: (4,B,R) = h1,c1,h2,c2.  The masked mean is detached, as in the reference. */
int xg_init_hidden(void *stream, const XgDims *d, const XgParams *p, const float *V,
                   const float *feat_mask, void *ws, size_t ws_bytes, float *state);

/* ---- hoisted attention projection: lstmcore.v2a(V) (caption_src/sub_modules.py:677) ---- */
int xg_vproj(void *stream, const XgDims *d, const XgParams *p, const float *V, float *vproj /* (B,K,A) */,
             const XgRun *run /* gemm_mode only; NULL = fp32 */);

/* ---- one decoder step: LSTMCore_two_layer_gate.forward + logit/log_softmax
 *      (caption_src/sub_modules.py:671-687, caption_src/SAModel.py:117-127 get_logprobs_state).
 * tokens (B) int64; xt_mask (B) 0/1 or NULL (= ones); state (4,B,R) updated in place;
 * logp (B,V) out or NULL; alpha (B,K) out or NULL.  `step` selects the dropout stream. */
int xg_step_fwd(void *stream, const XgDims *d, const XgParams *p, const int64_t *tokens,
                const float *xt_mask, const float *V, const float *vproj, const float *pos_feats,
                const XgRun *run, int step, void *ws, size_t ws_bytes,
                float *state, float *logp, float *alpha);

/* Backward of one step: xg_step_fwd must have run with run->save = 1 on the same workspace (it then keeps the old state,
 * the activated gates, p, alpha, af and the gate values there).  state_new: the (4,B,R) state AFTER the step;
 * dstate_new: gradient wrt it (the step's output is h2' = state_new[2]: add its gradient there); dstate (out, overwritten):
 * gradient wrt the state BEFORE the step.  dV (B,K,R), dvproj (B,K,A), dpos (B,R): optional, ACCUMULATED.  Parameter
 * gradients (lstmcore.*, embed.weight) are accumulated into g.  The logit head is not part of it (xg_gemm / xg_nll_*). */
int xg_step_bwd(void *stream, const XgDims *d, const XgParams *p, const XgParams *g, const int64_t *tokens,
                const float *xt_mask, const float *V, const float *vproj, const float *pos_feats,
                const XgRun *run, int step, void *ws, size_t ws_bytes, const float *state_new,
                const float *dstate_new, float *dstate, float *dV, float *dvproj, float *dpos);

/* ---- teacher-forced forward: SAModel.forward (caption_src/SAModel.py:67-115), ss_prob = 0 ----
 * Runs encoder, init_hidden, T decoder steps and both heads.
 * logp (B,T,V), cat_logp (B,T,C) out. */
int xg_forward_xe(void *stream, const XgDims *d, const XgParams *p, const XgBnState *bn,
                  const XgBatch *x, const XgRun *run, void *ws, size_t ws_bytes,
                  float *logp, float *cat_logp);
/* Backward of xg_forward_xe given dlogp (B,T,V) and dcat_logp (B,T,C) (either may be NULL = 0).
 * Parameter gradients are ACCUMULATED into g (caller zeroes them: optimizer.zero_grad(),
 * caption_src/starttrain.py:123). */
int xg_backward_xe(void *stream, const XgDims *d, const XgParams *p, const XgParams *g,
                   const XgBatch *x, const XgRun *run, void *ws, size_t ws_bytes,
                   const float *dlogp, const float *dcat_logp);

/* ---- teacher forcing with scheduled sampling: SAModel.forward, ss_prob > 0 (caption_src/SAModel.py:89-99) ----
 * At steps t >= 1 a row whose u_sel[t,b] < ss_prob is fed a token drawn from exp(log-probs of step t-1) by inverse
 * CDF with u_tok[t,b] (the reference draws both from torch's global RNG; the caller supplies the uniforms, (T,B) each).
 * The vocabulary head therefore runs inside the time loop.  Same outputs as xg_forward_xe. */
int xg_forward_ss(void *stream, const XgDims *d, const XgParams *p, const XgBnState *bn,
                  const XgBatch *x, const XgRun *run, float ss_prob, const float *u_sel, const float *u_tok,
                  void *ws, size_t ws_bytes, float *logp, float *cat_logp);
int xg_backward_ss(void *stream, const XgDims *d, const XgParams *p, const XgParams *g,
                   const XgBatch *x, const XgRun *run, void *ws, size_t ws_bytes,
                   const float *dlogp, const float *dcat_logp);

/* ---- fused XE loss path (same maths as forward_xe + LanguageModelCriterion +
 *      ClassiferCriterion, caption_src/SAModel.py:225-253, caption_src/starttrain.py:126-129)
 *      without materialising d(logp): loss = L_xe + weight_class * L_cls.
 * losses: device float[3] = {loss, L_xe, L_cls}.  cap_classes (B,T) int64, class_mask (B,T) or NULL. */
int xg_xe_loss_fwd(void *stream, const XgDims *d, const XgParams *p, const XgBnState *bn,
                   const XgBatch *x, const int64_t *cap_classes, const float *class_mask,
                   float weight_class, const XgRun *run, void *ws, size_t ws_bytes, float *losses);
/* dloss_dev: DEVICE scalar d(objective)/d(loss) (NULL = 1), so loss.backward() needs no host sync. */
int xg_xe_loss_bwd(void *stream, const XgDims *d, const XgParams *p, const XgParams *g,
                   const XgBatch *x, const int64_t *cap_classes, const float *class_mask,
                   float weight_class, const float *dloss_dev, const XgRun *run, void *ws, size_t ws_bytes);

/* ---- side streams (see XgRun.aux).  The handle owns two HIP streams and a few events on the CURRENT device; create it
 *      once per caller stream, destroy it when done.  The library keeps no global state. */
int xg_aux_create(void **aux);
int xg_aux_destroy(void *aux);

/* ---- rollouts: SAModel.sample (caption_src/SAModel.py:163-219), beam_size = 1 ----
 * mode GREEDY: argmax (ties -> lowest index, :186); SAMPLE: inverse-CDF draw from
 * exp(logp/temperature) with caller-supplied uniforms (T,B) in [0,1) (:190-194);
 * REPLAY: take `forced` (B,T-1) tokens.  T = seq_length+1 core steps are always run
 * (no per-step host sync, cf. :206); seq / seq_logp are (B,T-1), rows past the reference's
 * early exit are zero tokens.  n_steps (device int32) receives the reference's n. */
int xg_rollout(void *stream, const XgDims *d, const XgParams *p, const XgBnState *bn,
               const XgBatch *x, const XgRun *run, int mode, const float *uniforms,
               const int64_t *forced, float temperature, void *ws, size_t ws_bytes,
               int64_t *seq, float *seq_logp, int32_t *n_steps);
/* The two rollouts of one SCST iteration -- the sampled one (caption_src/starttrain.py:131) and the greedy
 * baseline (caption_src/myutils.py:45-48) -- as ONE batch of d2->B = n_sample + n_greedy rows: rows
 * [0, n_sample) sample with uniforms (T, n_sample), the remaining rows decode greedily; x2 holds the
 * features of every row (the caller repeats the videos).  seq / seq_logp are (d2->B, T-1); n_steps[0..1]
 * receive the reference's n of the sampled and of the greedy part.  Rows are independent and BatchNorm
 * statistics of a repeated batch equal those of the batch, so each part equals its own xg_rollout. */
int xg_rollout_pair(void *stream, const XgDims *d2, const XgParams *p, const XgBnState *bn,
                    const XgBatch *x2, const XgRun *run, int n_sample, const float *uniforms,
                    float temperature, void *ws2, size_t ws2_bytes, int64_t *seq, float *seq_logp,
                    int32_t *n_steps);
/* Copies what xg_rollout_bwd reads, for the first d1->B rows of a rollout that ran with d2 (d2->B >= d1->B,
 * all other extents equal) from workspace ws2 into workspace ws1, so that xg_rollout_bwd(d1, ws1) is the
 * backward of the sampled part alone. */
int xg_rollout_compact(void *stream, const XgDims *d2, const void *ws2, size_t ws2_bytes,
                       const XgDims *d1, void *ws1, size_t ws1_bytes);
/* xg_rollout_pair followed by xg_rollout_compact(d1 = the n_sample sampled rows) as ONE call: same outputs, same contents of
 * ws1 -- but the sampled rows' raw logits (two thirds of the bytes a compaction copies) are written into ws1 by the rollout
 * itself whenever its token choice runs over tile statistics (<= 128 rows, gemm_mode 0); d1->B must equal n_sample. */
int xg_rollout_pair_compact(void *stream, const XgDims *d2, const XgParams *p, const XgBnState *bn,
                            const XgBatch *x2, const XgRun *run, int n_sample, const float *uniforms,
                            float temperature, void *ws2, size_t ws2_bytes, const XgDims *d1, void *ws1,
                            size_t ws1_bytes, int64_t *seq, float *seq_logp, int32_t *n_steps);
/* The same SCST pair over the SAME videos without repeating them: x1 holds the m = d1->B videos once (d2->B = 2 m; rows
 * [0, m) of the outputs sample, rows [m, 2m) decode greedily).  The CG encoder, v2a(V) and the initial state are computed once,
 * in ws1 (the workspace of the un-repeated batch), and row-repeated on the device for the 2m-row decoder loop; in train mode
 * the BatchNorm running statistics receive the reference's TWO updates with the m-row batch statistics (one per sample()
 * call: caption_src/starttrain.py:131, caption_src/myutils.py:45).  compact != 0: ws1 is left as xg_rollout_pair_compact
 * leaves it (xg_rollout_bwd(d1, ws1) is the backward of the sampled rollout); compact = 0: ws1 only hosts the encoder. */
int xg_rollout_pair_videos(void *stream, const XgDims *d2, const XgParams *p, const XgBnState *bn, const XgBatch *x1,
                           const XgRun *run, const float *uniforms, float temperature, void *ws2, size_t ws2_bytes,
                           const XgDims *d1, void *ws1, size_t ws1_bytes, int compact, int64_t *seq, float *seq_logp,
                           int32_t *n_steps);
/* Backward of a rollout run with run->save = 1, given d(seq_logp) (B,T-1)
 * (RewardCriterion, caption_src/SAModel.py:259-267; caption_src/starttrain.py:131-134). */
int xg_rollout_bwd(void *stream, const XgDims *d, const XgParams *p, const XgParams *g,
                   const XgBatch *x, const XgRun *run, void *ws, size_t ws_bytes,
                   const float *dseq_logp);

/* ---- criteria (caption_src/SAModel.py:221-267) ---- */
/* out: device float[2] = {sum(-logp[target]*mask), sum(mask)}; loss = out[0]/out[1].
 * roll = 1 rolls the target left by one (LanguageModelCriterion :228); mask2 optional
 * second mask (ClassiferCriterion class_mask :250). */
int xg_nll_fwd(void *stream, const float *logp, const int64_t *target, const float *mask,
               const float *mask2, int B, int T, int V, int roll, float *out);
/* dlogp (B,T,V) is OVERWRITTEN with the dense gradient scale * scale_dev[0] * d(loss)/d(logp); scale_dev is an
 * optional DEVICE scalar (the incoming d(objective)/d(loss) of loss.backward(): no host sync, no extra pass). */
int xg_nll_bwd(void *stream, const int64_t *target, const float *mask, const float *mask2,
               int B, int T, int V, int roll, const float *sums, float scale, const float *scale_dev,
               float *dlogp);

/* RewardCriterion (caption_src/SAModel.py:255-267; caption_src/starttrain.py:133): out = {sum(-slp * reward * mask),
 * sum(mask)} with mask[:, 0] = 1, mask[:, t] = seq[:, t-1] > 0; loss = out[0] / out[1].  slp / seq are (m, L) with row
 * pitches ld_*; reward element (b, t) at reward[b * rs_b + t * rs_t] (rs_t = 0: one reward per video).  n_dev: optional
 * DEVICE int32 = the rollout's early-exit width n (xg_rollout's n_steps): columns t >= n are ignored, so full-width
 * rollout outputs can be fed without a host round trip.  xg_reward_bwd writes d(loss)/d(slp) * scale_dev[0] into dslp. */
int xg_reward_fwd(void *stream, const float *slp, int ld_slp, const int64_t *seq, int ld_seq, const float *reward,
                  int rs_b, int rs_t, const int32_t *n_dev, int m, int L, float *out);
int xg_reward_bwd(void *stream, const int64_t *seq, int ld_seq, const float *reward, int rs_b, int rs_t,
                  const int32_t *n_dev, int m, int L, const float *sums, const float *scale_dev, float *dslp, int ld_d);

/* ---- packed recurrent weights (no reference counterpart: a layout shadow of lstmcore.* / lstmcell_*.weight_hh,
 *      caption_src/sub_modules.py:661-669,104-105) ----
 * The per-timestep products stream each weight matrix once per step; a second copy of those matrices in the order the
 * matrix cores consume them (32 x 32 tiles, xg_pack.hip) lets the step kernels load the B operand straight into
 * registers.  The caller owns the buffer (xg_packed_bytes, 16-byte aligned), refreshes it with xg_pack_weights after
 * every parameter update (with_backward = 0 skips the data-gradient tiles: inference) and passes it in XgRun.packed
 * with XgRun.packed_dtype.  dtype 0: fp32 tiles; dtype 1: bf16 tiles for gemm_mode 1 (weights rounded once, half the bytes);
 * dtype 2: three bf16 planes per weight for gemm_mode 3 (the exact split x = p0 + p1 + p2 done once per update instead of in
 * registers on every pass of every step; 6 bytes per weight; bit-identical results to dtype 0 under gemm_mode 3). */
size_t xg_packed_bytes(const XgDims *d, int dtype);
int xg_pack_weights(void *stream, const XgDims *d, const XgParams *p, void *packed, size_t packed_bytes,
                    int dtype, int with_backward);
/* The same refresh in two parts: part 1 = every matrix but the CG encoder's (two_spatial_encoder.*), part 2 = those
 * (part 0 = xg_pack_weights).  An optimizer that updates the parameter groups as their gradients become final -- the decoder's
 * before the CG encoder's backward, train.ClipAdam(overlap=True) -- refreshes part 1 right behind that update, under the
 * encoder's backward, and part 2 at the head of the next iteration (6 us instead of 42 in front of the first product). */
int xg_pack_weights_part(void *stream, const XgDims *d, const XgParams *p, void *packed, size_t packed_bytes,
                         int dtype, int with_backward, int part);

/* ---- update: clip_gradient + Adam (caption_src/myutils.py:79-85, caption_src/starttrain.py:76,137) ----
 * Elementwise clamp of g to +-clip (clip <= 0 disables), then torch.optim.Adam semantics
 * (L2 weight decay folded into the gradient, bias correction with `step` >= 1). */
int xg_clip_adam(void *stream, int64_t n, float *param, float *grad, float *exp_avg,
                 float *exp_avg_sq, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, float clip);
/* The same update, but the gradient is left at ZERO instead of clamped: the next iteration's optimizer.zero_grad()
 * (caption_src/starttrain.py:123) folded into the pass that reads the gradient last -- the clamp already writes it, so
 * the 144 MB memset at the head of every iteration costs nothing here (train.ClipAdam(fused_zero=True)). */
int xg_clip_adam_zero(void *stream, int64_t n, float *param, float *grad, float *exp_avg, float *exp_avg_sq, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int step, float clip);

/* The same update with its step-dependent scalars in DEVICE memory, so that the launch arguments never change and the whole
 * training iteration can be captured in a HIP graph and replayed (train.GraphedXEStep): hyper = device float[4] =
 * {lr, 1 - beta1^t, sqrt(1 - beta2^t), t (int bits)}.  xg_adam_tick advances t by one and refreshes the two corrections (call
 * it once per optimizer step, before the xg_clip_adam_dev launches of that step); the caller writes lr (and zero-fills the
 * block once).  zero_grad != 0: the gradient is left at zero (xg_clip_adam_zero). */
int xg_adam_tick(void *stream, float *hyper, float beta1, float beta2);
int xg_clip_adam_dev(void *stream, int64_t n, float *param, float *grad, float *exp_avg, float *exp_avg_sq, const float *hyper,
                     float beta1, float beta2, float eps, float weight_decay, float clip, int zero_grad);

#ifdef __cplusplus
}
#endif
#endif /* XGATE_H */
