// Internal launcher prototypes (host side).  Every launcher enqueues on `st` and returns XG_OK / XG_EHIP.
#pragma once
#include "xg_common.h"
#include "xg_select.h"
#include <cstddef>

// ---- xg_gemm.hip
// `mode` = arithmetic of the product (XgRun.gemm_mode): 0 exact fp32 MFMA, 3 split-bf16, 1 bf16 operands; modes 1 / 3 apply
// to LARGE products only (M >= 256, N >= 64, K >= 64), everything else is fp32.  An explicit argument: no per-thread state.
// Tile rows per group of the GEMM kernels' tile order: the ~64 tiles an XCD works on at a time span gm tile rows x 64/gm
// tile columns, so its 4 MB L2 holds gm + 64/gm operand panels of 128 x k_depth floats instead of 1 + 64 (measured: the
// bf16 vocabulary product re-read W from HBM at 4.3 TB/s in row-major order).  Deep reductions keep the row-major order:
// their panels do not fit anyway and consecutive tiles should then share the bigger operand's panel.
static inline int xgk_group_rows(int k_depth) {
    const int g = 4096 / (k_depth > 0 ? k_depth : 1);
    return g < 1 ? 1 : (g > 8 ? 8 : g);
}
// `mode | XGK_GEMM_BG`: the product is launched on a side stream beside a latency-bound chain of small launches; the
// persistent kernel then takes half of every CU instead of all of it (xg_gemm.hip: launch_pk)
enum { XGK_GEMM_BG = 0x100,
       // `mode | XGK_GEMM_ALONE`: nothing latency-bound runs beside this product (the encoder's weight gradients at the tail of an
       // iteration): the weight-gradient layout takes the operands-from-memory kernel (gemm_td_kernel), 18 % faster alone on the
       // mid-size shapes and kept away from them elsewhere because it slows a launch chain on another stream (xg_gemm.hip: launch_td)
       XGK_GEMM_ALONE = 0x200 };
int xgk_gemm(hipStream_t st, int mode, bool transA, bool transB, int M, int N, int K, const float* A, int lda,
             const float* B, int ldb, float* C, int ldc, const float* bias, bool relu, bool accumulate);
// the same product for the weight-gradient layout (transA) with the column sums of A = dY^T as a side output: cs[m] += sum_k A(k, m)
// for up to three accumulators (cs2 / cs3 may be null) -- the bias gradient(s) of the same dY (xg_gemm.hip)
int xgk_gemm_cs(hipStream_t st, int mode, bool transA, bool transB, int M, int N, int K, const float* A, int lda,
                const float* B, int ldb, float* C, int ldc, const float* bias, bool relu, bool accumulate, float* cs1, float* cs2,
                float* cs3);
// ... and with optional bf16 copies of the operands (same shape / layout / leading dimension; null = none): used by the bf16
// arithmetic (mode 1) of large products, ignored otherwise
int xgk_gemm_x(hipStream_t st, int mode, bool transA, bool transB, int M, int N, int K, const float* A, const unsigned short* A16,
               int lda, const float* B, const unsigned short* B16, int ldb, float* C, int ldc, const float* bias, bool relu,
               bool accumulate, float* cs1 = nullptr, float* cs2 = nullptr, float* cs3 = nullptr);
// xg_gemm_bf16.hip: split-bf16 / bf16 arithmetic for large products (planes = 3 or 1)
// cs1..cs3 (optional, weight-gradient layout transA only): column sums of A = the bias gradient(s) of the same dY, a side output
int xgk_gemm_bf16(hipStream_t st, int planes, bool transA, bool transB, int M, int N, int K, const float* A, int lda,
                  const float* B, int ldb, float* C, int ldc, const float* bias, bool relu, bool accumulate, float* cs1 = nullptr,
                  float* cs2 = nullptr, float* cs3 = nullptr);
int xgk_gemm_bf16x(hipStream_t st, int planes, bool transA, bool transB, int M, int N, int K, const float* A, const unsigned short* A16,
                   int lda, const float* B, const unsigned short* B16, int ldb, float* C, int ldc, const float* bias, bool relu,
                   bool accumulate, float* cs1 = nullptr, float* cs2 = nullptr, float* cs3 = nullptr);
// xg_gemm_g16.hip: both operands bf16 in memory, tiles by LDS-DMA (round 5); xgk_gemm_g16_ok says whether a product qualifies
bool xgk_gemm_g16_ok(bool transA, bool transB, int M, int N, int K, const unsigned short* A16, int lda, const unsigned short* B16, int ldb);
int xgk_gemm_g16(hipStream_t st, bool transA, bool transB, int M, int N, int K, const unsigned short* A16, int lda,
                 const unsigned short* B16, int ldb, float* C, int ldc, const float* bias, bool relu, bool accumulate, int splitk,
                 float* cs1, float* cs2, float* cs3);
int xgk_cvt_bf16(hipStream_t st, const float* src, unsigned short* dst, size_t n);      // fp32 -> bf16 (RNE), both 16-byte aligned
// Y[M,N] (+)= X[M,K] W[N,K]^T + bias   (nn.Linear forward)
static inline int xgk_linear(hipStream_t st, int mode, int M, int N, int K, const float* X, int ldx, const float* W,
                             const float* bias, float* Y, int ldy, bool relu = false, bool acc = false) {
    return xgk_gemm(st, mode, false, true, M, N, K, X, ldx, W, K, Y, ldy, bias, relu, acc);
}

// ---- xg_pointwise.hip
enum { XG_ORDER_IFOG = 0 /* decoder two_inputs_lstmcell */, XG_ORDER_IFGO = 1 /* nn.LSTMCell */ };
enum { XG_MASK_HOLD = 0 /* decoder: keep previous state */, XG_MASK_ZERO = 1 /* encoder: zero state */ };

struct LstmFwdArgs {
    const float* s;      int lds_;    // (B,4R) pre-activation from the recurrent GEMM (bias included)
    const float* add;    int ldadd;   // optional second pre-activation term (hoisted input side), may be null
    const float* c_prev; int ldcp;
    const float* h_prev; int ldhp;    // only read in HOLD mode
    const float* mask;   int ldm;     // (B) element b at mask[b*ldm]; null = ones
    float* gates;        int ldg;     // (B,4R) activated gates, same column order as s; may be null
    float* c_out;        int ldco;
    float* h_out;        int ldho;    // post-dropout hidden (stored state AND output)
    int B, R, order, mask_mode;
    XgDrop drop;
};
int xgk_lstm_fwd(hipStream_t st, const LstmFwdArgs& a);

struct LstmBwdArgs {
    const float* gates;  int ldg;
    const float* c_prev; int ldcp;
    const float* c_out;  int ldco;    // post-mask new cell state
    const float* mask;   int ldm;
    const float* dh_out; int lddh;    // grad wrt post-dropout hidden
    const float* dh_add; int lddha;   // optional second term added to dh_out (e.g. the heads' gradient); may be null
    const float* dc_out; int lddc;    // grad wrt new cell state (from the next step); may be null (= 0)
    float* ds;           int ldds;    // (B,4R) out
    float* dc_prev;      int lddcp;   // out (overwritten)
    float* dh_prev;      int lddhp;   // HOLD mode: overwritten with (1-m)*dh' (caller accumulates GEMM terms after)
    int B, R, order, mask_mode;
    XgDrop drop;
};
int xgk_lstm_bwd(hipStream_t st, const LstmBwdArgs& a);
int xgk_lstm_bwd2(hipStream_t st, const LstmBwdArgs& a, const LstmBwdArgs& b);   // two independent cells, one launch

// y = g*t + t with g = dropout(pre) (pre already ReLU'd by the GEMM epilogue).  g is written back in place of pre.
// Row r of the (rows,R) operands: dropout step = drop.step + (r / s_div) % s_mod, dropout element index =
// ((r / b_div) % b_mod) * R + j; target row = r % t_mod when t_mod > 0 (broadcast over steps).  t / y may be null
// (plain in-place dropout of a ReLU'd activation).
int xgk_gate_fwd(hipStream_t st, float* pre_g, int ldp, const float* t, int ldt, int t_mod, float* y, int ldy,
                 int rows, int R, XgDrop drop, int s_div, int s_mod, int b_div, int b_mod);
// dpre = dy*t*keep*(g>0) ; dt (+)= dy*(g+1)
int xgk_gate_bwd(hipStream_t st, const float* dy, int lddy, const float* g, int ldg, const float* t, int ldt, int t_mod,
                 float* dpre, int lddp, float* dt, int lddt, bool dt_accumulate, int rows, int R, XgDrop drop);

// two gates of the same shape in ONE launch (the encoder's cross gates: forward, and backward with plain stores)
int xgk_gate_fwd2(hipStream_t st, float* pre_g0, float* pre_g1, int ldp, const float* t0, const float* t1, int ldt, int t_mod, float* y0,
                  float* y1, int ldy, int rows, int R, XgDrop drop0, XgDrop drop1, int s_div, int s_mod, int b_div, int b_mod);
int xgk_gate_bwd2(hipStream_t st, const float* dy0, const float* dy1, int lddy, const float* g0, const float* g1, int ldg, const float* t0,
                  const float* t1, int ldt, float* dpre0, float* dpre1, int lddp, float* dt0, float* dt1, int lddt, int rows, int R,
                  XgDrop drop0, XgDrop drop1);

// y = dropout(x) (x already ReLU'd) in place ; backward: dx = dy*keep*(y>0) in place
int xgk_relu_drop_fwd(hipStream_t st, float* x, int64_t n, XgDrop drop);
int xgk_relu_drop_bwd(hipStream_t st, float* dy, const float* y, int64_t n, XgDrop drop, const float* src = nullptr /* dy = f(src): out of place */);

// column reductions over rows of X (N,Cn) ld: out[c] += sum_r X[r][c]  (atomic accumulate; caller zeroes)
int xgk_colsum(hipStream_t st, const float* X, int ld, int N, int Cn, float* out);
// the same sums added into up to three accumulators (out2 / out3 may be null)
int xgk_colsum3(hipStream_t st, const float* X, int ld, int N, int Cn, float* out, float* out2, float* out3);
// out1[c] += sum_r X*Y ; (used for BN dgamma and a2w weight grad)
int xgk_colsum_prod(hipStream_t st, const float* X, int ldx, const float* Y, int ldy, int N, int Cn, float* out);

// BatchNorm1d over rows of Z (N,R): statistics, apply (+ReLU, dropout, row mask), backward
// train-mode BatchNorm forward in one launch: batch statistics (+ running statistics when rmean / rvar are given) and, when X is
// given, X = dropout(relu(bn(Z))) * rowmask.  Returns 1 when the shape is not the fused kernel's kind (R % 16, N <= 5120).
int xgk_bn_train_fwd(hipStream_t st, const float* Z, int N, int R, float* mean, float* var, float* rmean, float* rvar, float momentum,
                     const float* gamma, const float* beta, const float* rowmask, float* X, float eps, XgDrop drop);
int xgk_bn_stats(hipStream_t st, const float* Z, int N, int R, float* mean, float* var /*biased*/, float* rmean = nullptr,
                 float* rvar = nullptr, float momentum = 0.f /* running statistics updated in the same launch when given */);
int xgk_bn_running(hipStream_t st, const float* mean, const float* var, float* rmean, float* rvar, int N, int R,
                   float momentum);
// X[r][c] = dropout(relu((Z-mean)*rsqrt(var+eps)*gamma+beta)) * rowmask[r]
int xgk_bn_apply(hipStream_t st, const float* Z, const float* mean, const float* var, const float* gamma,
                 const float* beta, const float* rowmask, float* X, int N, int R, float eps, XgDrop drop);
// dY[r][c] = dX*rowmask*keep*(X>0)  (in place on dX) and sums: sum_dy[c], sum_dyxhat[c] (atomic; caller zeroes)
int xgk_bn_bwd_reduce(hipStream_t st, float* dX, const float* X, const float* Z, const float* mean, const float* var,
                      const float* rowmask, int N, int R, float eps, XgDrop drop, float* sum_dy, float* sum_dyxhat);
// dZ = gamma*invstd*(dY - sum_dy/N - xhat*sum_dyxhat/N) (train) or gamma*invstd*dY (eval), in place on dY
int xgk_bn_bwd_apply(hipStream_t st, float* dY, const float* Z, const float* mean, const float* var,
                     const float* gamma, const float* sum_dy, const float* sum_dyxhat, int N, int R, float eps,
                     bool train, float* g_beta = nullptr, float* g_gamma = nullptr /* += sum_dy, += sum_dyxhat in the same launch */);

// row i uses token tok[(i % inner) * s_inner + (i / inner) * s_outer]
int xgk_step_prep(hipStream_t st, const float* table, int E, const int64_t* tok, int V, float* xt, int B,
                  const float* state, float* state_copy, int64_t nstate);
int xgk_embed_gather(hipStream_t st, const float* table, int E, const int64_t* tok, int inner, int64_t s_inner,
                     int64_t s_outer, int n, int V, float* out, int ldo);
// dtable[tok(i)] += dX[i]   (atomic)
int xgk_embed_scatter_add(hipStream_t st, float* dtable, int E, const int64_t* tok, int inner, int64_t s_inner,
                          int64_t s_outer, int n, int V, const float* dX, int ldx);
// vbar[b][r] = sum_k V[b][k][r] / sum_k mask[b][k]
int xgk_masked_mean(hipStream_t st, const float* V, const float* mask, float* out, int B, int K, int R);
int xgk_axpy(hipStream_t st, float* y, const float* x, float a, int64_t n);          // y += a*x
int xgk_fill(hipStream_t st, float* y, float v, int64_t n);
// strided 2-D copy / add: dst[r*ldd + c] (=|+=) src[r*lds + c]
int xgk_copy2d(hipStream_t st, float* dst, int ldd, const float* src, int lds, int rows, int cols, bool add);

// ---- xg_step.hip : multi-job skinny split-K MFMA GEMM with optional LSTM-cell epilogue
enum { SK_EPI_STORE = 0, SK_EPI_LSTM = 1, SK_EPI_GATE = 2,
       // LSTM cell BACKWARD of the previous time step in the epilogue of the product that completes its dh:
       //   dh = product (+ C when accumulate) (+ add), then the cell's pointwise backward (xg_pointwise.hip:lstm_bwd_body)
       //   -> ds (M,4R), dc_prev, and in HOLD mode the (1-m) dh pass-through into dh_hold.  N = R.
       SK_EPI_LSTMB = 3,
       // (fast kernel only) zero-fill job: C[0 .. M*N) = 0, one tile per 4096 floats
       SK_EPI_ZERO = 4,
       // (fast kernel only) copy job: C[0 .. M*N) = seg[0].A[0 .. M*N), one tile per 4096 floats, 16-byte aligned
       SK_EPI_COPY = 6,

       // (fast kernel only) temporal attention of one step, TWO workgroups per video (sub_modules.py:678-680):
       //   workgroup (b, part) scores its half of the K frames, e_k = w . tanh(p_b + q_bk), and adds its share of the
       //   UNNORMALISED softmax -- ex_k = exp(e_k - e_0), s = sum ex_k, c = sum ex_k V_bk -- into attn_s[b] / attn_c[b][:]
       //   (zeroed by a ZERO job of the previous launch; two commutative adds per element, so the sum is deterministic).
       //   The cell-2 job of the NEXT launch normalises while it stages af (SkSeg.row_scale) -- no merge launch.
       SK_EPI_ATTN = 5 };
constexpr int SK_MAX_JOBS = 5;
constexpr int XGK_SKPART_TILES = 256;            // (tile, publishing part) slabs of 1024 8-byte granules in one job's split scratch
constexpr size_t SKPART_INTS = (size_t)XGK_SKPART_TILES * 1024 * 2;        // int32 words of one job's split scratch
// Field ORDER matters to the fast kernel (xg_step.hip: skf_kernel): it reads a job's 64-byte head and the first 48 bytes of each
// segment with ONE round of wide scalar loads at kernel entry (s_load_dwordx16 / x8 / x4) instead of field by field behind
// branches -- a chain of ~15 dependent scalar-cache round trips in front of the first operand request before round 5.
struct SkSeg {
    // ---- hot (48 B: what it takes to request the segment's first operands)
    const float* A;                    // A (M,K) row-major lda (rows gathered through `gather` when set)
    // packed form of B (xg_pack.hip: 32 x 32 tiles in MFMA-fragment order, nck tiles per 32-column slice) or null: when
    // every segment of a launch has one, the launch takes the fast kernel (B operand global -> VGPR, no LDS)
    const float* Bp;                   // (bf16 tiles when the launch runs with gemm_mode 1: xgk_skinny's argument)
    // optional row gather on A: row m of the operand is A + clamp(gather[m], 0, gather_max) * lda
    // (embedding lookup folded into the product: caption_src/SAModel.py:105,198)
    const int64_t* gather;
    int lda, K, nck;
    int a_bytes;                       // filled by xgk_skinny: extent of the A operand in bytes, ((rows - 1) lda + K) * 4 -- the fast kernel's buffer descriptor clamps there
    int gather_max;
    int sflags;                        // filled by xgk_skinny: SKS_SCALED | SKS_WRITEBACK | SKS_EX
    // ---- cold
    const float* B; int ldb, b_ncontig;   // B (N,K) row-major ldb, or (K,N) when b_ncontig (LDS-staged kernel, fallbacks)
    // optional per-row scale of A: row m is multiplied by 1 / row_scale[m] while it is staged (the attention context
    // arrives unnormalised: af = c / s).  scaled_out (lda_out): the scaled rows are written back by the tn == 0 tiles
    // (the normalised context is a saved tensor); ex / ex_ld / ex_K: those tiles also normalise the (M, ex_K) unnormalised
    // attention weights in place.
    const float* row_scale; float* scaled_out; float* ex; int ld_out, ex_ld, ex_K;
    int ex_magic, ex_per, pad_;        // ceil(2^32 / ex_K) and the weights per n-tile of the m-tile's 32 x ex_K block: filled by xgk_skinny
};
enum { SKS_SCALED = 1, SKS_WRITEBACK = 2, SKS_EX = 4 };
enum { SKH_CELL_TILES = 1 /* weight rows in the cell tiling (LSTM epilogue or cell_cols) */, SKH_LOW_PRIO = 2, SKH_HAS_SCALED = 4,
       SKH_SELECT = 8 /* the token choice runs in front of the gathered segment (SkJob.select) */,
       SKH_GATHER_SHIFT = 4 /* bits 4-5: 1 + index of the first gathered segment (0 = none) */,
       SKH_POW2_NTM = 64 /* the m-tile count is a power of two: tile decode by shifts */, SKH_LGNTM_SHIFT = 8 /* bits 8-11 */,
       SKH_LGKS_SHIFT = 12 /* bits 12-15: log2 of the cross-workgroup split */ };
struct alignas(64) SkJob {
    // ---- head (64 B).  ntm / ntn / ntiles / hflags / ksplit / nck_all are filled in by xgk_skinny.
    int epi, M, N, R;
    int nseg, ksplit, ntm, ntn;
    int ntiles, hflags, ldc, nck_all;     // nck_all: 32-deep chunks of all segments together (the fast kernel splits THAT among its waves)
    float* C;                          // STORE epilogue output (M,N) ldc
    int* tickets;                      // cross-workgroup split-K of an LSTMB job: XGK_SKPART_TILES x 8 KB of scratch, ZERO between launches (see ksplit_ok)
    SkSeg seg[3];
    // ---- epilogue INPUTS, contiguous (requested before the K loop so that their latency hides under it)
    const float* bias[3];
    // LSTM epilogue (N must be 4R; weight rows / bias / add columns in gate-major order like the reference)
    const float* add; const float* c_prev; const float* h_prev; const float* mask;
    int ldadd, ldcp, ldhp, ldm;
    int accumulate, relu, order, mask_mode;
    // GATE epilogue (sub_modules.py:42-47): g = dropout(relu(.)) -> C ; y = g*t + t
    const float* gate_t; int ldt, ldy; float* gate_y;
    // ---- epilogue OUTPUTS
    float* gates; float* c_out; float* h_out;
    int ldg, ldco, ldho;
    // STORE epilogue over a (M,4R) gate-major pre-activation with the CELL tiling of the weight rows (tile tn = hidden units
    // 8 tn .. 8 tn + 7 of all four gates, like the LSTM epilogue): a partial cell product another launch finishes
    int cell_cols;
    XgDrop drop;
    // LSTMB epilogue (reads gates / c_prev / c_out / mask / add from the LSTM fields above)
    const float* dc_in; float* ds; float* dc_prev; float* dh_hold; int lddci, ldds, lddcp, lddhh;
    // ATTN job: p (M,A) at attn_p, q = v2a(V) (M,K,A) at attn_q, V (M,K,R) at attn_v, w_a (A) at attn_w; outputs
    // attn_ex (M,K) unnormalised weights, attn_s (M) and attn_c (M,R) accumulated with atomics
    const float *attn_p, *attn_q, *attn_v, *attn_w; float *attn_ex, *attn_s, *attn_c; int attn_K, attn_A;
    // Cross-workgroup split-K (fast kernel; STORE and LSTMB jobs whose result ACCUMULATES into C): ksplit_ok = 1 lets
    // xgk_skinny spread a tile's reduction over several workgroups when the launch would leave CUs idle (the backward
    // chains' dh = ds W products are 64-128 tiles, K = 1536-2048 deep).  STORE: each part adds its partial tile into C with fp32
    // atomics.  LSTMB: parts 0 .. ksplit - 2 publish their tiles as tagged 8-byte granules in the scratch at `tickets` (1024
    // granules per tile and part, left at zero again) and the last part -- dispatched last -- sums them in part order and runs the
    // pointwise backward (xg_step.hip: sk_epilogue_split).  ksplit is filled in by xgk_skinny.
    int ksplit_ok;
    int ksplit_cap;                    // > 0: upper bound of the cross-workgroup split of THIS launch (a side chain that must not crowd the main one)
    int low_prio;                      // 1: the job's waves drop back to default wave priority (off-critical-path side chains)
    int select;                        // 1: the job's workgroups first CHOOSE the tokens of their 32 rows (SkArgs.sel) and gather by them
};
static_assert(offsetof(SkJob, seg) == 64 && sizeof(SkSeg) == 112, "skf_kernel reads the head and the segments' hot parts by offset");
static_assert(sizeof(SkJob) % 64 == 0 && sizeof(SkJob) * SK_MAX_JOBS + 64 <= 4096, "SkJob array stride / kernel-argument budget");
// `sel` (round 6): the token choice of a rollout step as the prologue of the launch's SELECT job (SkJob.select: the POS-gate tiles,
// whose row gather then takes the tokens their own workgroup has just chosen): xg_select.h.  Read only by that job's workgroups.
struct SkArgs { int njobs; int tile0[SK_MAX_JOBS]; int pad_[10]; SkJob job[SK_MAX_JOBS]; RollSelectArgs sel; };
static_assert(offsetof(SkArgs, job) == 64, "descriptor lines");
static_assert(sizeof(SkArgs) <= 4096, "kernel-argument budget");
// gemm_mode 1 (plain bf16) rounds the staged chunks to bf16 (LDS-staged kernel); 0 / 3: exact fp32
// `gemm_mode | XGK_SK_PLANES` (mode 3 only): the packed tiles (SkSeg::Bp) hold three pre-split bf16 planes (xg_pack.hip, dtype 2)
enum { XGK_SK_PLANES = 0x200 };
int xgk_skinny(hipStream_t st, SkArgs& a, int gemm_mode);

// ---- xg_dstep.hip : one decoder step (attention + POS gate + the two cells, sub_modules.py:671-687) as ONE dataflow launch
struct DStepArgs {
    int B, R, A, E, K, V1;                       // V1 = vocabulary size - 1 (clamp of the token gather)
    const float *h1, *c1, *h2, *c2;              // state before the step, (B,R) each
    float *h1o, *c1o, *h2o, *c2o;                // state after the step (may alias the input state: copy_back)
    float *h1w, *h2w;                            // where the cell epilogues write h1' / h2': h1o / h2o, or scratch rows when in place
    int copy_back;                               // 1: h1w / h2w are scratch; the last cell-2 tile of each m-tile copies them into h1o / h2o
    const float* xt; const int64_t* tok; const float* embed;     // xt (B,E) rows, or tokens + embed.weight (gathered inside the products)
    const float* pos; float* gp; float* posg;    // POS gate: raw feature, saved gate values, gated feature (written unless pre1)
    const float* pre1;                           // teacher forcing: hoisted token side of cell 1 (B,4R), biases included; else null
    const float* mask; int ldm;                  // xt_mask: element b at mask[b * ldm]; null = ones
    float *P, *alpha, *af, *g1, *g2;             // p (B,A), attention weights (B,K) [may be null], context (B,R), gates (B,4R) [may be null]
    const float *V, *vproj, *a2w;
    const float *pk_h2a1, *pk_h2a2, *pk_dgate, *pk_l1_i2h, *pk_l1_a2h, *pk_l1_h2h, *pk_l2_i2h, *pk_l2_a2h, *pk_l2_h2h;   // packed tiles
    const float *h2a_b, *dgate_b, *l1_i2h_b, *l1_a2h_b, *l1_h2h_b, *l2_i2h_b, *l2_a2h_b, *l2_h2h_b;
    int* ctr;                                    // sync words (xgk_dstep_sync_bytes), zero on entry, left at zero
    XgDrop drop_gate, drop_l1, drop_l2;
    int t0_p, t0_att, t0_c1, t0_c2, total;       // first block of each job after the gate tiles (filled in by xgk_dstep)
};
constexpr size_t XGK_DSTEP_SYNC_BYTES = 1024;   // reserved in every workspace; the kernel itself exists in the -DXG_DIAG build only
int xgk_dstep_err_word();
bool xgk_dstep_ok(const XgDims& d);              // shapes the dataflow kernel takes
int xgk_dstep(hipStream_t st, DStepArgs& a, int gemm_mode);

// ---- xg_pack.hip : weights re-tiled into MFMA-fragment order (caller-owned shadow, XgRun.packed)
enum { PK_H2A1 = 0, PK_H2A2, PK_DGATE, PK_L1_I2H, PK_L1_A2H, PK_L1_H2H, PK_L2_I2H, PK_L2_A2H, PK_L2_H2H, PK_ENC_RGB, PK_ENC_OPFL,
       PKB_L2_A2H, PKB_L2_H2H, PKB_H2A2, PKB_L1_H2H, PKB_ENC_RGB, PKB_ENC_OPFL, PKB_L2_I2H, PKB_H2A1, PK_COUNT };
// plain bf16 copies (row-major, the parameter's own shape) of the weights of the LARGE products, part of the dtype-1 shadow:
// the bf16 GEMMs read them instead of converting the fp32 weights on every pass (BASELINE.json configs[4])
enum { W16_LOGIT = 0, W16_EMB_RGB, W16_EMB_OPFL, W16_WIH_RGB, W16_WIH_OPFL, W16_GATE_RGB, W16_GATE_OPFL, W16_FUSION, W16_V2A,
       W16_DGATE, W16_L1_I2H, W16_L1_A2H, W16_COUNT };
struct PackedView { const float* m[PK_COUNT]; int nck[PK_COUNT]; int dtype;       // dtype 0: fp32 tiles (4 KB), 1: bf16 tiles (2 KB), 2: three bf16 planes (6 KB)
                    const unsigned short* w16[W16_COUNT]; };                      // (null for dtype 0)
size_t xgk_packed_floats(const XgDims& d);
size_t xgk_packed_total_bytes(const XgDims& d, int dtype);
bool xgk_packed_view(const XgDims& d, const void* packed, int dtype, PackedView* v);   // false: no / unusable shadow

// ---- xg_attn.hip
// per-sample additive attention: e_k = w . tanh(p + q_k), alpha = softmax_k(e), af = sum_k alpha_k V_k
int xgk_attn_fwd(hipStream_t st, const float* p, const float* vproj, const float* V, const float* w, float* alpha,
                 float* af, int B, int K, int R, int A, bool half_cu = false);   // half_cu: <= 64 VGPRs, fits beside a background GEMM workgroup
// de[b][k], dp[b][a] from daf[b][r]
int xgk_attn_bwd(hipStream_t st, const float* daf, int lddaf, const float* p, const float* vproj, const float* V,
                 const float* w, const float* alpha, float* de, float* dp, int B, int K, int R, int A);
// after the time loop: dvproj[b][k][a] = w_a sum_t de_t (1-th^2) ; dw[a] += sum de_t th ; dV[b][k][r] (+)= sum_t alpha_t daf_t
int xgk_attn_bwd_post(hipStream_t st, const float* P /*(T,B,A)*/, const float* vproj, const float* w,
                      const float* DE /*(T,B,K)*/, float* dvproj, float* dw, int T, int B, int K, int A);
// dV = sum_t alpha_t dAF_t (plain store) and dq / dw of the hoisted projection as ONE launch (T <= 32; otherwise the two passes)
int xgk_attn_post_dV(hipStream_t st, const float* P, const float* vproj, const float* w, const float* DE, float* dvproj, float* dw,
                     const float* ALPHA, const float* DAF, int lddaf, int64_t daf_tstride, float* dV, int T, int B, int K, int A, int R);
int xgk_attn_dV(hipStream_t st, const float* ALPHA /*(T,B,K)*/, const float* DAF /*(T,B,ldaf)*/, int lddaf,
                int64_t daf_tstride, float* dV, int T, int B, int K, int R, bool accumulate);

// ---- xg_heads.hip
// out[orow(i)] = log_softmax(in[i]) ; orow = (i % inner) * outer + i / inner when permute (time-major -> batch-major)
int xgk_log_softmax(hipStream_t st, const float* in, int ldin, float* out, int ldout, int rows, int V, int inner,
                    int outer, bool permute);
// dlogits[i] = dlogp[drow] - exp(logp[lrow]) * sum_v dlogp[drow]; permute 0: drow = lrow = i; 1: both = orow(i);
// 2: drow = orow(i), lrow = i (logp kept time-major in the workspace; may alias dlogits)
int xgk_log_softmax_bwd(hipStream_t st, const float* dlogp, const float* logp, int ldp, float* dlogits, int ldd,
                        int rows, int V, int inner, int outer, int permute);
int xgk_nll_fwd(hipStream_t st, const float* logp, const int64_t* target, const float* mask, const float* mask2,
                int B, int T, int V, int roll, float* out2);
int xgk_nll_bwd(hipStream_t st, const int64_t* target, const float* mask, const float* mask2, int B, int T, int V,
                int roll, const float* sums, float scale, const float* scale_dev, float* dlogp);
// fused: rows are time-major logits (T*B,V): per-row lse, loss accumulation, and dlogits in place
int xgk_xent_fwd(hipStream_t st, const float* logits, int ld, const int64_t* seq, const float* mask,
                 const float* mask2, int B, int T, int V, int roll, float* lse, float* sums2, int row0 = 0,
                 int nrows = -1, bool zero_sums = true);      // rows [row0, row0 + nrows) of the T*B (nrows < 0: to the end)
// d16: optional bf16 copy of the gradient (same pitch in elements), written in the same pass
int xgk_xent_bwd(hipStream_t st, float* logits_inout, int ld, const int64_t* seq, const float* mask,
                 const float* mask2, int B, int T, int V, int roll, const float* lse, const float* sums2,
                 const float* scale_dev, float scale, int row0 = 0, int nrows = -1, unsigned short* d16 = nullptr);
// rollout token choice from a (B,V) log-prob matrix
int xgk_choose(hipStream_t st, const float* logp, int B, int V, int mode, const float* uniforms,
               const int64_t* forced, int64_t forced_stride, float temperature, int64_t* tok, float* tok_logp);
// fused rollout step: choice from raw logits + bookkeeping + embedding gather (xg_heads.hip)
int xgk_rollout_step(hipStream_t st, int B, const float* logits, const float* uniforms, const int64_t* forced,
                     int64_t fstride, const float* unf_prev, const float* table, int64_t* tok, float* tok_logp, float* unf,
                     float* lse, int64_t* seq, float* seq_logp, int32_t* maxf, float* xt, float temperature, int V, int E,
                     int t, int T, int mode, int split);
int xgk_rollout_finalize(hipStream_t st, const int32_t* maxf, int32_t* n_steps, int Tm1, int nparts);
// rollout steps of <= 128 rows: vocabulary product with per-tile row statistics + token choice over them (xg_heads.hip)
bool xgk_vocab_select_ok(int B, int R, int V, const float* H, int ldh, const float* W);
int xgk_vocab_tile_width(int V);                 // columns per tile statistic of xgk_vocab_part for this vocabulary
int xgk_vocab_part(hipStream_t st, int B, int R, int V, const float* H, int ldh, const float* W, const float* bias, float* logits,
                   int wr_rows, float* part, float temperature);
RollSelectArgs xgk_roll_select_args(const float* logits, const float* part, const float* uniforms, const int64_t* forced, int64_t fstride,
                                    const float* unf_prev, const float* table, int64_t* tok, float* tok_logp, float* unf, float* lse,
                                    int64_t* seq, float* seq_logp, int32_t* maxf, float* xt, float temperature, int V, int E, int t, int T,
                                    int mode, int split);
int xgk_roll_select(hipStream_t st, int B, const float* logits, const float* part, const float* uniforms, const int64_t* forced,
                    int64_t fstride, const float* unf_prev, const float* table, int64_t* tok, float* tok_logp, float* unf,
                    float* lse, int64_t* seq, float* seq_logp, int32_t* maxf, float* xt, float temperature, int V, int E,
                    int t, int T, int mode, int split);
// all steps in one launch: rows (steps, B) of logits / lse / tok, dslp (B, dstride)
int xgk_rollout_dlogits_lse(hipStream_t st, float* logits, const float* lse, const int64_t* tok, const float* dslp,
                            int64_t dstride, int B, int V, int steps);
// gather of row-block prefixes (xg_rollout_compact): entry e copies nblocks blocks of dst_block floats, source pitch src_pitch
constexpr int XG_COMPACT_MAX = 48;
struct CompactEntry { float* dst; const float* src; int64_t dst_block, src_pitch, total; };
struct CompactArgs { int n; int start[XG_COMPACT_MAX + 1]; CompactEntry e[XG_COMPACT_MAX]; };
int xgk_compact(hipStream_t st, CompactArgs& a);
// scheduled sampling (SAModel.py:89-99): tok[b] = u_sel[b] < ss_prob ? sampled[b] : seq[b*T + t]
int xgk_ss_select(hipStream_t st, const int64_t* seq, int T, int t, int B, const float* u_sel, float ss_prob,
                  const int64_t* sampled, int64_t* tok);
// rollout bookkeeping (SAModel.py:200-215)
int xgk_rollout_book(hipStream_t st, int t, int B, int Tm1, int replay, const int64_t* tok, const float* tok_logp,
                     float* unfinished, int64_t* seq, float* seq_logp, int32_t* n_steps, int32_t* alive);
// dlogits (B,V) row b = d * (onehot(tok) - softmax) for the sampled token (rollout backward)
int xgk_rollout_dlogits(hipStream_t st, const float* logp, const int64_t* tok, const float* dslp, int64_t dstride,
                        float* dlogits, int B, int V);

// ---- xg_optim.hip
int xgk_clip_adam(hipStream_t st, int64_t n, float* p, float* g, float* m, float* v, float lr, float b1, float b2,
                  float eps, float wd, int step, float clip, bool zero_grad = false);
