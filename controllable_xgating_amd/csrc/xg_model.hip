// Host-side orchestration of the caption decoder hot path on one HIP stream (gfx950).
// Every function only enqueues kernels; no host synchronisation, no allocation.
//
// Reference maths: caption_src/sub_modules.py:118-159 (CG encoder), caption_src/SAModel.py:58-65
// (init_hidden), caption_src/sub_modules.py:671-687 + :750-770 (decoder step), caption_src/SAModel.py:67-115
// (teacher-forced forward), :163-219 (rollouts), :221-267 (criteria).
//
// Layouts in the workspace (all fp32):
//   encoder tensors are (B*K, .) rows in (b,k) order, exactly the reference's .view(-1, F);
//   decoder tensors are time-major (T, B, .) so each step's rows are contiguous and the stacked
//   (T*B, .) matrices feed the batched head / weight-gradient GEMMs directly.
#include "xg_common.h"
#include <cstring>
#include <ctime>
#include "xg_kernels.h"

#include <new>
#include <cstdlib>
#include <functional>

namespace {

// ---- auxiliary streams: weight-gradient / token-side GEMMs that nothing downstream waits for run there, under the
// latency-bound recurrent kernels of the main chain.  The two streams and their events belong to an EXPLICIT handle the
// caller creates (xg_aux_create), passes in XgRun.aux and destroys (xg_aux_destroy): no library-global state.  One handle
// serves one caller stream at a time (two callers on two streams -- the SCST sampled / greedy rollouts -- use two
// handles).  Everything is joined back onto the caller's stream before an entry point returns, so the stream semantics of
// the C ABI are unchanged.  XgRun.aux == NULL (or XG_NO_OVERLAP=1): everything runs on the caller's stream.
constexpr int XG_NEV = 64, XG_NRING = 56;     // events: a ring for fork / join pairs + slots for long-lived marks
struct XgAux { uint32_t magic; int device; hipStream_t s = nullptr, s2 = nullptr; hipEvent_t ev[XG_NEV]; };
constexpr uint32_t XG_AUX_MAGIC = 0x58474158u;
XgAux* aux_of(const XgRun* run) {
    static const bool disabled = xg_diag_env("XG_NO_OVERLAP") != nullptr;
    if (disabled || !run || !run->aux) return nullptr;
    XgAux* a = static_cast<XgAux*>(run->aux);
    int dev = -1;
    if (a->magic != XG_AUX_MAGIC || hipGetDevice(&dev) != hipSuccess || dev != a->device) return nullptr;
    return a;
}

struct Streams {
    hipStream_t main, aux, aux2;              // aux2: a second side chain (the decoder backward's cell-1 recurrence)
    XgAux* a;
    int next = 0, next_mark = 0;
    bool forked = false, forked2 = false;
    int dh_split_step = 0, dh_mark = -1;      // heads_bwd -> decoder_bwd_core hand-off (see heads_bwd)
    hipEvent_t grad_event = nullptr;          // XgRun.grad_event: recorded when every gradient but the encoder's is final
    hipEvent_t grad_event_head = nullptr;     // XgRun.grad_event_head: recorded when the logit.* gradients are final
    Streams(hipStream_t m, const XgRun* run) : main(m), aux(m), aux2(m), a(aux_of(run)) {
        if (a) { aux = a->s; aux2 = a->s2; }
        if (run) { grad_event = static_cast<hipEvent_t>(run->grad_event); grad_event_head = static_cast<hipEvent_t>(run->grad_event_head); }
    }
    bool overlap() const { return a != nullptr; }
    // aux may start work that depends on everything enqueued on main so far
    int fork() {
        if (!a) return XG_OK;
        hipEvent_t e = a->ev[next++ % XG_NRING];
        if (hipEventRecord(e, main) != hipSuccess || hipStreamWaitEvent(aux, e, 0) != hipSuccess) return XG_EHIP;
        forked = true;
        return XG_OK;
    }
    // aux2 may start work that depends on everything enqueued on main so far
    int fork2() {
        if (!a) return XG_OK;
        hipEvent_t e = a->ev[next++ % XG_NRING];
        if (hipEventRecord(e, main) != hipSuccess || hipStreamWaitEvent(aux2, e, 0) != hipSuccess) return XG_EHIP;
        forked2 = true;
        return XG_OK;
    }
    // main waits for everything enqueued on aux2 so far
    int join2() {
        if (!a || !forked2) return XG_OK;
        hipEvent_t e = a->ev[next++ % XG_NRING];
        if (hipEventRecord(e, aux2) != hipSuccess || hipStreamWaitEvent(main, e, 0) != hipSuccess) return XG_EHIP;
        return XG_OK;
    }
    // aux waits for everything enqueued on aux2 so far (then a join() of aux covers both)
    int chain2_into_aux() {
        if (!a || !forked2) return XG_OK;
        hipEvent_t e = a->ev[next++ % XG_NRING];
        if (hipEventRecord(e, aux2) != hipSuccess || hipStreamWaitEvent(aux, e, 0) != hipSuccess) return XG_EHIP;
        forked = true;
        return XG_OK;
    }
    // record a point on aux that main can wait for later (returns an event slot, or -1 when not overlapping)
    // (marks live across many steps of a loop: they come from their own slots, not from the fork / join ring)
    int mark() {
        if (!a) return -1;
        const int i = XG_NRING + (next_mark++ % (XG_NEV - XG_NRING));
        if (hipEventRecord(a->ev[i], aux) != hipSuccess) return -2;
        return i;
    }
    int wait_mark(int i) {
        if (!a || i < 0) return XG_OK;
        return hipStreamWaitEvent(main, a->ev[i], 0) == hipSuccess ? XG_OK : XG_EHIP;
    }
    int wait_mark2(int i) {                   // ... and the second side chain
        if (!a || i < 0) return XG_OK;
        return hipStreamWaitEvent(aux2, a->ev[i], 0) == hipSuccess ? XG_OK : XG_EHIP;
    }
    // main waits for everything enqueued on aux so far
    int join() {
        if (!a || !forked) return XG_OK;
        hipEvent_t e = a->ev[next++ % XG_NRING];
        if (hipEventRecord(e, aux) != hipSuccess || hipStreamWaitEvent(main, e, 0) != hipSuccess) return XG_EHIP;
        return XG_OK;
    }
};

struct Ws {
    // ---- encoder
    float *Z[2], *X[2], *PRE[2], *Hs[2], *Cs[2], *G[2], *GG[2], *Hprev[2];
    float *bn_mean[2], *bn_var[2], *bn_s1[2], *bn_s2[2];
    float *zeroBR, *S, *S2, *Y, *Venc;
    float *dVw, *dY, *dHs[2], *dGG[2], *dS[2], *dX[2], *dHrec[2], *dCrec[2][2];
    // ---- decoder
    float *vbar, *vproj, *Xe, *GP, *POSG, *PRE1, *H1, *C1, *H2, *C2, *G1, *G2, *P, *ALPHA, *AF;
    float *LOGITS, *HC, *CL, *LSE, *LSEC, *sums;   // sums: 8 floats
    float *DH2OUT, *DHC, *DCL, *DS1, *DS2, *DP, *DE, *DAF, *dst[2][4], *DVPROJ, *DV, *DPOSG, *DGP, *DXe, *DH1X;
    float *state_tmp;
    float *AFU, *ATS;                  // unnormalised attention context (B,R) and softmax denominators (B), contiguous
    // ---- rollout
    int64_t* TOK; float *TOKLP, *UNF; int32_t* alive; float* VPART;
    int32_t* tickets;                  // split-K hand-off scratch of the LSTMB jobs (xg_step.hip): SK_MAX_JOBS blocks of XGK_SKPART_TILES x 8 KB, zero between launches
    int32_t* dsync;                    // sync words of the dataflow step kernel (xg_dstep.hip): zero between launches
    // everything a backward pass needs ZERO on entry is one contiguous block (dst[0][*], DAF, the encoder's carried
    // gradients, the BatchNorm sums, the tickets): one memset on a side stream instead of a dozen on the critical path
    char* zblock; size_t zbytes; bool zeroed;
    size_t bytes, core_bytes;
    // ---- packed recurrent weights (XgRun.packed; not part of the workspace)
    PackedView pk; bool packed;
    int gm;                            // XgRun.gemm_mode of this call (0 / 1 / 3), handed to every product explicitly
    // bf16 mirror region (gemm_mode 1): the fp32 element at workspace offset o has its bf16 copy at sh16 + o.  A mirror is valid
    // only where the code below converted it (cvt16) or a producer wrote it (xent_bwd); the large products of mode 1 read
    // operands through it where it is valid (m16) -- half the operand bytes from L2, no convert in the GEMM.
    const float* base_f; const float* end_f; unsigned short* sh16;
};

struct Carver {
    char* base; size_t off;
    template <typename T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

Ws carve(const XgDims& d, void* base) {
    Ws w{};
    Carver c{static_cast<char*>(base), 0};
    const size_t B = d.B, K = d.K, R = d.R, A = d.A, E = d.E, V = d.V, C = d.C, H = d.H, T = d.T;
    const size_t N = B * K, TB = T * B;
    for (int m = 0; m < 2; ++m) {
        w.Z[m] = c.take<float>(N * R); w.X[m] = c.take<float>(N * R); w.PRE[m] = c.take<float>(N * 4 * R);
        w.Hs[m] = c.take<float>(N * R); w.Cs[m] = c.take<float>(N * R); w.G[m] = c.take<float>(N * 4 * R);
        w.GG[m] = c.take<float>(N * R); w.Hprev[m] = c.take<float>(N * R);
        w.bn_mean[m] = c.take<float>(R); w.bn_var[m] = c.take<float>(R);
        w.dHs[m] = c.take<float>(N * R); w.dGG[m] = c.take<float>(N * R); w.dS[m] = c.take<float>(N * 4 * R);
        w.dX[m] = c.take<float>(N * R);
        w.dCrec[m][1] = c.take<float>(B * R);
    }
    w.zeroBR = c.take<float>(B * R); w.S = c.take<float>(B * 4 * R); w.S2 = c.take<float>(B * 4 * R);
    w.Y = c.take<float>(N * 2 * R); w.Venc = c.take<float>(N * R);
    w.dVw = c.take<float>(N * R); w.dY = c.take<float>(N * 2 * R);
    w.vbar = c.take<float>(B * R); w.vproj = c.take<float>(N * A);
    w.Xe = c.take<float>(TB * E); w.GP = c.take<float>(TB * R); w.POSG = c.take<float>(TB * R);
    w.PRE1 = c.take<float>(TB * 4 * R);
    w.H1 = c.take<float>((T + 1) * B * R); w.C1 = c.take<float>((T + 1) * B * R);
    w.H2 = c.take<float>((T + 1) * B * R); w.C2 = c.take<float>((T + 1) * B * R);
    w.G1 = c.take<float>(TB * 4 * R); w.G2 = c.take<float>(TB * 4 * R);
    w.P = c.take<float>(TB * A); w.ALPHA = c.take<float>(TB * K); w.AF = c.take<float>(TB * R);
    w.LOGITS = c.take<float>(TB * V); w.HC = c.take<float>(TB * H); w.CL = c.take<float>(TB * C);
    w.LSE = c.take<float>(TB); w.LSEC = c.take<float>(TB); w.sums = c.take<float>(8);
    w.DH2OUT = c.take<float>(TB * R); w.DHC = c.take<float>(TB * H); w.DCL = c.take<float>(TB * C);
    w.DS1 = c.take<float>(TB * 4 * R); w.DS2 = c.take<float>(TB * 4 * R); w.DP = c.take<float>(TB * A);
    w.DE = c.take<float>(TB * K);
    for (int j = 0; j < 4; ++j) w.dst[1][j] = c.take<float>(B * R);
    w.DVPROJ = c.take<float>(N * A); w.DV = c.take<float>(N * R); w.DPOSG = c.take<float>(TB * R); w.DH1X = c.take<float>(TB * R);
    w.DGP = c.take<float>(TB * R); w.DXe = c.take<float>(TB * E);
    w.state_tmp = c.take<float>(4 * B * R);
    w.AFU = c.take<float>(B * R + ((B + 3) & ~(size_t)3)); w.ATS = w.AFU + B * R;
    w.TOK = c.take<int64_t>(TB); w.TOKLP = c.take<float>(TB); w.UNF = c.take<float>(TB);
    w.alive = c.take<int32_t>(4);
    c.off = (c.off + 15) & ~(size_t)15;
    w.VPART = c.take<float>(B * ((V + 31) / 32 + 16) * 4);    // per-tile row statistics of a rollout step's vocabulary product (row pitch: xg_select.h rs_pitch)
    w.dsync = c.take<int32_t>(XGK_DSTEP_SYNC_BYTES / sizeof(int32_t));
    {   // the zero block
        c.off = (c.off + 255) & ~(size_t)255;
        const size_t z0 = c.off;
        w.zblock = c.base ? c.base + z0 : nullptr;
        w.DAF = c.take<float>(TB * R);
        for (int j = 0; j < 4; ++j) w.dst[0][j] = c.take<float>(B * R);
        for (int m = 0; m < 2; ++m) {
            w.dHrec[m] = c.take<float>(B * R); w.dCrec[m][0] = c.take<float>(B * R);
            w.bn_s1[m] = c.take<float>(R); w.bn_s2[m] = c.take<float>(R);
        }
        w.tickets = c.take<int32_t>((size_t)SK_MAX_JOBS * SKPART_INTS);
        w.zbytes = c.off - z0;
    }
    w.bytes = (c.off + 255) & ~(size_t)255;
    w.base_f = reinterpret_cast<const float*>(c.base);
    w.end_f = c.base ? reinterpret_cast<const float*>(c.base + w.bytes) : nullptr;
    w.sh16 = c.base ? reinterpret_cast<unsigned short*>(c.base + w.bytes) : nullptr;
    w.core_bytes = w.bytes;                      // everything but the bf16 mirror region (only gemm_mode 1 reads mirrors)
    w.bytes += (w.bytes / 2 + 255) & ~(size_t)255;
    return w;
}

bool dims_ok(const XgDims* d) {
    return d && d->B > 0 && d->K > 0 && d->R > 0 && d->A > 0 && d->E > 0 && d->V > 1 && d->C > 0 && d->H > 0 &&
           d->F1 > 0 && d->F2 > 0 && d->T > 0;
}

#define ZERO(ptr, nfloats) \
    do { if (hipMemsetAsync((ptr), 0, sizeof(float) * (size_t)(nfloats), st) != hipSuccess) return XG_EHIP; } while (0)

// NN data-gradient GEMM: dX[M,N] (+)= dY[M,Kc] * W[Kc,N]   (W row-major, ldw)
inline int gemm_nn(hipStream_t st, int mode, int M, int N, int Kc, const float* dY, int lddy, const float* W, int ldw,
                   float* dX, int lddx, bool acc) {
    return xgk_gemm(st, mode, false, false, M, N, Kc, dY, lddy, W, ldw, dX, lddx, nullptr, false, acc);
}
// TN weight-gradient GEMM: dW[N,K] += dY[M,N]^T * X[M,K]
inline int gemm_tn(hipStream_t st, int mode, int Mrows, int N, int K, const float* dY, int lddy, const float* X, int ldx,
                   float* dW, int lddw) {
    return xgk_gemm(st, mode, true, false, N, K, Mrows, dY, lddy, X, ldx, dW, lddw, nullptr, false, true);
}

// ... and the bias gradient(s) of the same dY as a side output of the product: b1[n] (b2, b3) += sum_m dY[m, n]
inline int gemm_tn_cs(hipStream_t st, int mode, int Mrows, int N, int K, const float* dY, int lddy, const float* X, int ldx,
                      float* dW, int lddw, float* b1, float* b2 = nullptr, float* b3 = nullptr) {
    return xgk_gemm_cs(st, mode, true, false, N, K, Mrows, dY, lddy, X, ldx, dW, lddw, nullptr, false, true, b1, b2, b3);
}

// ---- bf16 operand mirrors / weight copies of the plain-bf16 mode (all null / no-ops in the other modes)
inline const unsigned short* m16(const Ws& w, const float* p) {
    return (w.gm == 1 && w.sh16 && p >= w.base_f && p < w.end_f) ? w.sh16 + (p - w.base_f) : nullptr;
}
inline int cvt16(hipStream_t st, const Ws& w, const float* p, size_t n) {          // refresh the mirror of p[0 .. n)
    if (w.gm != 1 || n == 0) return XG_OK;
    if (!w.sh16) return XG_OK;                   // workspace without a mirror region: the consumers convert on the fly
    const unsigned short* dst = m16(w, p);
    if (!dst) return XG_EINVAL;
    return xgk_cvt_bf16(st, p, const_cast<unsigned short*>(dst), n);
}
inline const unsigned short* w16(const Ws& w, int which) {
    return (w.gm == 1 && w.packed && w.pk.dtype == 1) ? w.pk.w16[which] : nullptr;
}
// Y = X W^T + b ; dX = dY W ; dW += dY^T X (+ bias gradients) with optional bf16 copies of the operands
inline int lin16(hipStream_t st, int mode, int M, int N, int K, const float* X, const unsigned short* X16, int ldx, const float* W,
                 const unsigned short* W16, const float* bias, float* Y, int ldy, bool relu = false, bool acc = false) {
    return xgk_gemm_x(st, mode, false, true, M, N, K, X, X16, ldx, W, W16, K, Y, ldy, bias, relu, acc);
}
inline int nn16(hipStream_t st, int mode, int M, int N, int Kc, const float* dY, const unsigned short* dY16, int lddy, const float* W,
                const unsigned short* W16, int ldw, float* dX, int lddx, bool acc) {
    return xgk_gemm_x(st, mode, false, false, M, N, Kc, dY, dY16, lddy, W, W16, ldw, dX, lddx, nullptr, false, acc);
}
inline int tn16(hipStream_t st, int mode, int Mrows, int N, int K, const float* dY, const unsigned short* dY16, int lddy, const float* X,
                const unsigned short* X16, int ldx, float* dW, int lddw, float* b1 = nullptr, float* b2 = nullptr, float* b3 = nullptr) {
    return xgk_gemm_x(st, mode, true, false, N, K, Mrows, dY, dY16, lddy, X, X16, ldx, dW, lddw, nullptr, false, true, b1, b2, b3);
}

// ---- skinny-job builders
inline SkSeg seg_nt(const float* A, int lda, const float* W, int ldw, int K) {
    SkSeg s{}; s.A = A; s.B = W; s.lda = lda; s.ldb = ldw; s.K = K; s.b_ncontig = 0; return s;
}
inline SkSeg seg_nn(const float* dY, int lddy, const float* W, int ldw, int Kc) {
    SkSeg s{}; s.A = dY; s.B = W; s.lda = lddy; s.ldb = ldw; s.K = Kc; s.b_ncontig = 1; return s;
}
// the same segments with the packed form of the weight attached when the caller supplied one (XgRun.packed)
inline SkSeg seg_nt(const Ws& w, int which, const float* A, int lda, const float* W, int ldw, int K) {
    SkSeg s = seg_nt(A, lda, W, ldw, K);
    if (w.packed) { s.Bp = w.pk.m[which]; s.nck = w.pk.nck[which]; }
    return s;
}
inline SkSeg seg_nn(const Ws& w, int which, const float* dY, int lddy, const float* W, int ldw, int Kc) {
    SkSeg s = seg_nn(dY, lddy, W, ldw, Kc);
    if (w.packed) { s.Bp = w.pk.m[which]; s.nck = w.pk.nck[which]; }
    return s;
}
// let xgk_skinny split the reduction of job j across workgroups (its result accumulates into C: see SkJob.ksplit_ok)
inline void allow_split(SkArgs& sk, int j, const Ws& w) { sk.job[j].ksplit_ok = 1; sk.job[j].tickets = w.tickets + j * SKPART_INTS; }
// (the last arriver of every tile leaves its counter at zero; the memsets only make a launch independent of whatever an
// aborted run left behind.  njobs: how many 1024-counter blocks the caller's launches on THIS stream use -- the decoder
// backward's cell-1 chain runs on a side stream with block 2 while the encoder backward uses blocks 0-1.)
inline int zero_tickets(hipStream_t st, const Ws& w, int njobs) {
    return hipMemsetAsync(w.tickets, 0, sizeof(int32_t) * njobs * SKPART_INTS, st) == hipSuccess ? XG_OK : XG_EHIP;
}
// The dataflow step kernel's sync words are zero between launches (its last workgroup leaves them so); whole-sequence entry
// points clear them once more per sequence, so that a sequence never inherits what an aborted launch left behind.  The
// single-step entry point (xg_step_fwd) relies on the invariant alone: the workspace contract asks for a zero-filled
// workspace at first use (include/xgate.h).
inline int zero_dsync(hipStream_t st, const Ws& w) {
#ifdef XG_DIAG
    return hipMemsetAsync(w.dsync, 0, XGK_DSTEP_SYNC_BYTES, st) == hipSuccess ? XG_OK : XG_EHIP;
#else
    (void)st; (void)w;               // the dataflow kernel is not part of the product library
    return XG_OK;
#endif
}
// (the tile element type must fit the arithmetic: bf16 tiles for gemm_mode 1; fp32 tiles otherwise; pre-split planes -- dtype 2 --
// for gemm_mode 3 only)
inline void attach_packed(Ws& w, const XgDims& d, const XgRun* run) {
    static const bool disabled = xg_diag_env("XG_NO_PACKED") != nullptr;
    w.packed = !disabled && run && run->packed && (run->gemm_mode == 1) == (run->packed_dtype == 1) &&
               (run->packed_dtype != 2 || run->gemm_mode == 3) &&
               xgk_packed_view(d, run->packed, run->packed_dtype, &w.pk);
}
// what the per-step launcher is told: the arithmetic mode, and whether the attached tiles are pre-split planes
inline int sk_mode(const Ws& w) { return w.gm | ((w.gm == 3 && w.packed && w.pk.dtype == 2) ? XGK_SK_PLANES : 0); }
inline SkJob job_store(int M, int N, float* C, int ldc, bool acc, bool relu = false) {
    SkJob j{};
    j.M = M; j.N = N; j.C = C; j.ldc = ldc; j.accumulate = acc ? 1 : 0; j.relu = relu ? 1 : 0; j.epi = SK_EPI_STORE;
    return j;
}
// product job whose epilogue runs the pointwise LSTM backward `a` (a.dh_out is ignored: dh = product (+ *acc_from) + a.dh_add)
inline SkJob job_lstm_bwd(const LstmBwdArgs& a, const float* acc_from, int ldacc) {
    SkJob j{};
    j.M = a.B; j.N = a.R; j.R = a.R; j.epi = SK_EPI_LSTMB; j.order = a.order; j.mask_mode = a.mask_mode;
    j.C = const_cast<float*>(acc_from); j.ldc = ldacc; j.accumulate = acc_from ? 1 : 0;
    j.add = a.dh_add; j.ldadd = a.lddha;
    j.gates = const_cast<float*>(a.gates); j.ldg = a.ldg; j.c_prev = a.c_prev; j.ldcp = a.ldcp;
    j.c_out = const_cast<float*>(a.c_out); j.ldco = a.ldco; j.mask = a.mask; j.ldm = a.ldm;
    j.dc_in = a.dc_out; j.lddci = a.lddc; j.ds = a.ds; j.ldds = a.ldds; j.dc_prev = a.dc_prev; j.lddcp = a.lddcp;
    j.dh_hold = a.dh_prev; j.lddhh = a.lddhp; j.drop = a.drop;
    return j;
}
inline SkJob job_lstm(const LstmFwdArgs& a) {
    SkJob j{};
    j.M = a.B; j.N = 4 * a.R; j.R = a.R; j.epi = SK_EPI_LSTM; j.order = a.order; j.mask_mode = a.mask_mode;
    j.add = a.add; j.ldadd = a.ldadd; j.c_prev = a.c_prev; j.ldcp = a.ldcp; j.h_prev = a.h_prev; j.ldhp = a.ldhp;
    j.mask = a.mask; j.ldm = a.ldm; j.gates = a.gates; j.ldg = a.ldg; j.c_out = a.c_out; j.ldco = a.ldco;
    j.h_out = a.h_out; j.ldho = a.ldho; j.drop = a.drop;
    return j;
}

// ================================================================== encoder
int encoder_fwd(hipStream_t st, const XgDims& d, const XgParams& p, const XgBnState* bn, const XgBatch& x,
                const XgRun& run, Ws& w, Streams* ss = nullptr) {
    const int B = d.B, K = d.K, R = d.R, N = B * K;
    const float* feats[2] = {x.feats_rgb, x.feats_opfl};
    const int F[2] = {d.F1, d.F2};
    const float* emb_w[2] = {p.emb_rgb_w, p.emb_opfl_w};
    const float* emb_b[2] = {p.emb_rgb_b, p.emb_opfl_b};
    const float* bn_g[2] = {p.bn_rgb_g, p.bn_opfl_g};
    const float* bn_b[2] = {p.bn_rgb_b, p.bn_opfl_b};
    float* rmean[2] = {bn ? bn->rgb_mean : nullptr, bn ? bn->opfl_mean : nullptr};
    float* rvar[2] = {bn ? bn->rgb_var : nullptr, bn ? bn->opfl_var : nullptr};
    const float* wih[2] = {p.lstm_rgb_wih, p.lstm_opfl_wih};
    const float* whh[2] = {p.lstm_rgb_whh, p.lstm_opfl_whh};
    const float* bih[2] = {p.lstm_rgb_bih, p.lstm_opfl_bih};
    const float* bhh[2] = {p.lstm_rgb_bhh, p.lstm_opfl_bhh};
    // the two modalities' embed -> BatchNorm -> W_ih pipelines are independent until the recurrence: the optical-flow one
    // runs on the second auxiliary stream
    const bool side = ss && ss->overlap();
    if (side) XG_TRY(ss->fork2());
    hipStream_t st_main = st;
    for (int m = 0; m < 2; ++m) {
        hipStream_t st = (m == 1 && side) ? ss->aux2 : st_main;
        XG_TRY(lin16(st, w.gm, N, R, F[m], feats[m], nullptr, F[m], emb_w[m], w16(w, m == 0 ? W16_EMB_RGB : W16_EMB_OPFL), emb_b[m],
                     w.Z[m], R));                                                                           // sub_modules.py:121,126
        const XgDrop emb_drop = xg_make_drop(&run, m == 0 ? XG_SITE_EMB_RGB : XG_SITE_EMB_OPFL, 0);
        bool applied = false;
        if (run.train) {
            const bool upd = rmean[m] && rvar[m];                  // (running statistics: updated by the same launch)
            // statistics + running statistics + the layer's output in ONE launch when the shape allows
            const int rc = xgk_bn_train_fwd(st, w.Z[m], N, R, w.bn_mean[m], w.bn_var[m], upd ? rmean[m] : nullptr, upd ? rvar[m] : nullptr,
                                            run.bn_momentum, bn_g[m], bn_b[m], x.feat_mask, w.X[m], run.bn_eps, emb_drop);
            if (rc == XG_OK) applied = true;
            else if (rc != 1) return rc;
            else XG_TRY(xgk_bn_stats(st, w.Z[m], N, R, w.bn_mean[m], w.bn_var[m], upd ? rmean[m] : nullptr, upd ? rvar[m] : nullptr,
                                     run.bn_momentum));
        } else {
            if (!rmean[m] || !rvar[m]) return XG_EINVAL;
            if (hipMemcpyAsync(w.bn_mean[m], rmean[m], sizeof(float) * R, hipMemcpyDeviceToDevice, st) != hipSuccess) return XG_EHIP;
            if (hipMemcpyAsync(w.bn_var[m], rvar[m], sizeof(float) * R, hipMemcpyDeviceToDevice, st) != hipSuccess) return XG_EHIP;
        }
        if (!applied)
            XG_TRY(xgk_bn_apply(st, w.Z[m], w.bn_mean[m], w.bn_var[m], bn_g[m], bn_b[m], x.feat_mask, w.X[m], N, R, run.bn_eps, emb_drop));
        XG_TRY(cvt16(st, w, w.X[m], (size_t)N * R));
        XG_TRY(lin16(st, w.gm, N, 4 * R, R, w.X[m], m16(w, w.X[m]), R, wih[m], w16(w, m == 0 ? W16_WIH_RGB : W16_WIH_OPFL), bih[m],
                     w.PRE[m], 4 * R));                                                                     // hoisted over all K frames
    }
    if (hipMemsetAsync(w.zeroBR, 0, sizeof(float) * (size_t)B * R, (side ? ss->aux2 : st_main)) != hipSuccess) return XG_EHIP;   // (off the main chain)
    if (side) XG_TRY(ss->join2());
    XgRun nodrop = run; nodrop.drop_p = 0.f;
    for (int i = 0; i < K; ++i) {                                                                  // sub_modules.py:132-148
        SkArgs sk{};
        sk.njobs = 2;                                           // both modalities' cells in ONE launch
        for (int m = 0; m < 2; ++m) {
            const float* hp = i == 0 ? w.zeroBR : w.Hs[m] + (size_t)(i - 1) * R;
            const float* cp = i == 0 ? w.zeroBR : w.Cs[m] + (size_t)(i - 1) * R;
            const int ldp = i == 0 ? R : K * R;
            LstmFwdArgs a{};
            a.add = w.PRE[m] + (size_t)i * 4 * R; a.ldadd = K * 4 * R;
            a.c_prev = cp; a.ldcp = ldp; a.h_prev = hp; a.ldhp = ldp;
            a.mask = x.feat_mask + i; a.ldm = K;
            a.gates = w.G[m] + (size_t)i * 4 * R; a.ldg = K * 4 * R;
            a.c_out = w.Cs[m] + (size_t)i * R; a.ldco = K * R;
            a.h_out = w.Hs[m] + (size_t)i * R; a.ldho = K * R;
            a.B = B; a.R = R; a.order = XG_ORDER_IFGO; a.mask_mode = XG_MASK_ZERO;
            a.drop = xg_make_drop(&nodrop, 0, 0);
            if (R % 8 == 0) {
                sk.job[m] = job_lstm(a);
                sk.job[m].nseg = 1; sk.job[m].seg[0] = seg_nt(w, m == 0 ? PK_ENC_RGB : PK_ENC_OPFL, hp, ldp, whh[m], R, R);
                sk.job[m].bias[0] = bhh[m];
            } else {
                float* S = m == 0 ? w.S : w.S2;
                XG_TRY(xgk_linear(st, w.gm, B, 4 * R, R, hp, ldp, whh[m], bhh[m], S, 4 * R));
                a.s = S; a.lds_ = 4 * R;
                XG_TRY(xgk_lstm_fwd(st, a));
            }
        }
        if (R % 8 == 0) XG_TRY(xgk_skinny(st, sk, sk_mode(w)));
    }
    // cross gates, all frames at once (gated values are not fed back): sub_modules.py:151-152
    XG_TRY(cvt16(st, w, w.Hs[0], (size_t)N * R));
    XG_TRY(cvt16(st, w, w.Hs[1], (size_t)N * R));
    XG_TRY(lin16(st, w.gm, N, R, R, w.Hs[1], m16(w, w.Hs[1]), R, p.gate_rgb_w, w16(w, W16_GATE_RGB), p.gate_rgb_b, w.GG[0], R, true));
    XG_TRY(lin16(st, w.gm, N, R, R, w.Hs[0], m16(w, w.Hs[0]), R, p.gate_opfl_w, w16(w, W16_GATE_OPFL), p.gate_opfl_b, w.GG[1], R, true));
    XG_TRY(xgk_gate_fwd2(st, w.GG[0], w.GG[1], R, w.Hs[0], w.Hs[1], R, 0, w.Y, w.Y + R, 2 * R, N, R,
                         xg_make_drop(&run, XG_SITE_GATE_RGB, 0), xg_make_drop(&run, XG_SITE_GATE_OPFL, 0), /*step=row%K*/ 1, K,
                         /*b=row/K*/ K, 1 << 30));                                        // (both gates: one launch)
    XG_TRY(cvt16(st, w, w.Y, (size_t)N * 2 * R));
    XG_TRY(lin16(st, w.gm, N, R, 2 * R, w.Y, m16(w, w.Y), 2 * R, p.fusion_w, w16(w, W16_FUSION), p.fusion_b, w.Venc, R, true));   // :69-70
    XG_TRY(xgk_relu_drop_fwd(st, w.Venc, (int64_t)N * R, xg_make_drop(&run, XG_SITE_FUSION, 0)));
    XG_TRY(cvt16(st, w, w.Venc, (size_t)N * R));
    return XG_OK;
}

int encoder_bwd(Streams& ss, const XgDims& d, const XgParams& p, const XgParams& g, const XgBatch& x,
                const XgRun& run, Ws& w, const float* dV_in) {
    hipStream_t st = ss.main, sx = ss.aux;     // sx: parameter-gradient work nothing downstream waits for
    const int B = d.B, K = d.K, R = d.R, N = B * K;
    const float* feats[2] = {x.feats_rgb, x.feats_opfl};
    const int F[2] = {d.F1, d.F2};
    const float* wih[2] = {p.lstm_rgb_wih, p.lstm_opfl_wih};
    const float* whh[2] = {p.lstm_rgb_whh, p.lstm_opfl_whh};
    float* g_wih[2] = {g.lstm_rgb_wih, g.lstm_opfl_wih};
    float* g_whh[2] = {g.lstm_rgb_whh, g.lstm_opfl_whh};
    float* g_bih[2] = {g.lstm_rgb_bih, g.lstm_opfl_bih};
    float* g_bhh[2] = {g.lstm_rgb_bhh, g.lstm_opfl_bhh};
    const float* gate_w[2] = {p.gate_rgb_w, p.gate_opfl_w};
    float* g_gate_w[2] = {g.gate_rgb_w, g.gate_opfl_w};
    float* g_gate_b[2] = {g.gate_rgb_b, g.gate_opfl_b};
    const float* bn_g[2] = {p.bn_rgb_g, p.bn_opfl_g};
    float* g_bn_g[2] = {g.bn_rgb_g, g.bn_opfl_g};
    float* g_bn_b[2] = {g.bn_rgb_b, g.bn_opfl_b};
    float* g_emb_w[2] = {g.emb_rgb_w, g.emb_opfl_w};
    float* g_emb_b[2] = {g.emb_rgb_b, g.emb_opfl_b};

    XG_TRY(xgk_relu_drop_bwd(st, w.dVw, w.Venc, (int64_t)N * R, xg_make_drop(&run, XG_SITE_FUSION, 0), dV_in));   // dVw = f(dV_in)
    XG_TRY(cvt16(st, w, w.dVw, (size_t)N * R));
    XG_TRY(ss.fork());
    XG_TRY(tn16(sx, w.gm, N, R, 2 * R, w.dVw, m16(w, w.dVw), R, w.Y, m16(w, w.Y), 2 * R, g.fusion_w, 2 * R, g.fusion_b));
    XG_TRY(nn16(st, w.gm, N, 2 * R, R, w.dVw, m16(w, w.dVw), R, p.fusion_w, w16(w, W16_FUSION), 2 * R, w.dY, 2 * R, false));
    // y = g*h + h : dpre -> dGG, dh -> dHs (overwrite); both gates in one launch
    XG_TRY(xgk_gate_bwd2(st, w.dY, w.dY + R, 2 * R, w.GG[0], w.GG[1], R, w.Hs[0], w.Hs[1], R, w.dGG[0], w.dGG[1], R, w.dHs[0], w.dHs[1], R,
                         N, R, xg_make_drop(&run, XG_SITE_GATE_RGB, 0), xg_make_drop(&run, XG_SITE_GATE_OPFL, 0)));
    for (int m = 0; m < 2; ++m) XG_TRY(cvt16(st, w, w.dGG[m], (size_t)N * R));
    XG_TRY(ss.fork());
    for (int m = 0; m < 2; ++m) {   // gate m takes source = hidden of the OTHER modality
        const int o = 1 - m;
        XG_TRY(tn16(sx, w.gm, N, R, R, w.dGG[m], m16(w, w.dGG[m]), R, w.Hs[o], m16(w, w.Hs[o]), R, g_gate_w[m], R, g_gate_b[m]));
        XG_TRY(nn16(st, w.gm, N, R, R, w.dGG[m], m16(w, w.dGG[m]), R, gate_w[m], w16(w, m == 0 ? W16_GATE_RGB : W16_GATE_OPFL), R,
                    w.dHs[o], R, true));
    }
    XgRun nodrop = run; nodrop.drop_p = 0.f;
    int curc = 0;
    if (!w.zeroed) {
        for (int m = 0; m < 2; ++m) { ZERO(w.dHrec[m], (size_t)B * R); ZERO(w.dCrec[m][0], (size_t)B * R); }
        XG_TRY(zero_tickets(st, w, 2));
    }
    // cell backward of frame i for modality m, reading / writing the carried dc of parity c.  Stand-alone it takes
    // dh = dHs[i] + dHrec; fused into the product dS[i+1] Whh (epilogue) it takes dh = product + dHs[i].
    auto enc_cell_bwd = [&](int m, int i, int c, bool fused) {
        LstmBwdArgs a{};
        a.gates = w.G[m] + (size_t)i * 4 * R; a.ldg = K * 4 * R;
        a.c_prev = i == 0 ? w.zeroBR : w.Cs[m] + (size_t)(i - 1) * R; a.ldcp = i == 0 ? R : K * R;
        a.c_out = w.Cs[m] + (size_t)i * R; a.ldco = K * R;
        a.mask = x.feat_mask + i; a.ldm = K;
        if (fused) { a.dh_out = nullptr; a.lddh = 0; a.dh_add = nullptr; a.lddha = 0; }      // dHs[i] is the job's accumulate operand
        else { a.dh_out = w.dHs[m] + (size_t)i * R; a.lddh = K * R; a.dh_add = w.dHrec[m]; a.lddha = R; }
        a.dc_out = w.dCrec[m][c]; a.lddc = R;
        a.ds = w.dS[m] + (size_t)i * 4 * R; a.ldds = K * 4 * R;
        a.dc_prev = w.dCrec[m][c ^ 1]; a.lddcp = R;
        a.dh_prev = nullptr; a.lddhp = 0;
        a.B = B; a.R = R; a.order = XG_ORDER_IFGO; a.mask_mode = XG_MASK_ZERO;
        a.drop = xg_make_drop(&nodrop, 0, 0);
        return a;
    };
    const bool fuse = R % 4 == 0;
    for (int i = K - 1; i >= 0; --i) {                          // both modalities per launch
        if (!fuse || i == K - 1) XG_TRY(xgk_lstm_bwd2(st, enc_cell_bwd(0, i, curc, false), enc_cell_bwd(1, i, curc, false)));
        curc ^= 1;
        if (i > 0) {
            SkArgs sk{};
            sk.njobs = 2;
            for (int m = 0; m < 2; ++m) {
                // dh of frame i-1 = dS[i] Whh (+ dHs[i-1]); fused: frame i-1's cell backward in the epilogue
                // (the product accumulates into dHs[i-1], which nothing reads afterwards: split-K across workgroups allowed)
                if (fuse) {
                    sk.job[m] = job_lstm_bwd(enc_cell_bwd(m, i - 1, curc, true), w.dHs[m] + (size_t)(i - 1) * R, K * R); allow_split(sk, m, w);
                }
                else sk.job[m] = job_store(B, R, w.dHrec[m], R, false);
                sk.job[m].nseg = 1;
                sk.job[m].seg[0] = seg_nn(w, m == 0 ? PKB_ENC_RGB : PKB_ENC_OPFL, w.dS[m] + (size_t)i * 4 * R, K * 4 * R, whh[m], R, 4 * R);
            }
            XG_TRY(xgk_skinny(st, sk, sk_mode(w)));
        }
    }
    for (int m = 0; m < 2; ++m) XG_TRY(cvt16(st, w, w.dS[m], (size_t)N * 4 * R));      // read by three products each
    const int gm_tail = w.gm | XGK_GEMM_ALONE;
    XG_TRY(ss.fork2());                          // before modality 0's work is enqueued on the main stream
    for (int m = 0; m < 2; ++m) {
        // Hprev[b,k] = H[b,k-1], zero at k = 0 : one clean TN GEMM for dW_hh
        if (m == 0) XG_TRY(ss.fork());           // dS of both modalities is final after the loop
        if (hipMemsetAsync(w.Hprev[m], 0, sizeof(float) * (size_t)N * R, sx) != hipSuccess) return XG_EHIP;
        if (K > 1)   // one strided 2-D copy covers all videos: rows = B, cols = (K-1)*R
            XG_TRY(xgk_copy2d(sx, w.Hprev[m] + R, K * R, w.Hs[m], K * R, B, (K - 1) * R, false));
        XG_TRY(cvt16(sx, w, w.Hprev[m], (size_t)N * R));
        // (the recurrence is over: no launch chain runs beside these weight gradients -- XGK_GEMM_ALONE, xg_kernels.h)
        XG_TRY(tn16(sx, gm_tail, N, 4 * R, R, w.dS[m], m16(w, w.dS[m]), 4 * R, w.Hprev[m], m16(w, w.Hprev[m]), R, g_whh[m], R, g_bih[m], g_bhh[m]));
        XG_TRY(tn16(sx, gm_tail, N, 4 * R, R, w.dS[m], m16(w, w.dS[m]), 4 * R, w.X[m], m16(w, w.X[m]), R, g_wih[m], R));
        // the optical-flow modality's input-side backward runs beside the rgb one (second auxiliary stream, forked above)
        hipStream_t st_outer = st;
        hipStream_t st = (m == 1 && ss.overlap()) ? ss.aux2 : st_outer;
        XG_TRY(nn16(st, w.gm, N, R, 4 * R, w.dS[m], m16(w, w.dS[m]), 4 * R, wih[m], w16(w, m == 0 ? W16_WIH_RGB : W16_WIH_OPFL), R,
                    w.dX[m], R, false));
        // BatchNorm + ReLU + dropout + mask backward (sub_modules.py:121-123)
        if (!w.zeroed) { ZERO(w.bn_s1[m], R); ZERO(w.bn_s2[m], R); }
        XG_TRY(xgk_bn_bwd_reduce(st, w.dX[m], w.X[m], w.Z[m], w.bn_mean[m], w.bn_var[m], x.feat_mask, N, R, run.bn_eps,
                                 xg_make_drop(&run, m == 0 ? XG_SITE_EMB_RGB : XG_SITE_EMB_OPFL, 0), w.bn_s1[m], w.bn_s2[m]));
        XG_TRY(xgk_bn_bwd_apply(st, w.dX[m], w.Z[m], w.bn_mean[m], w.bn_var[m], bn_g[m], w.bn_s1[m], w.bn_s2[m], N, R,
                                run.bn_eps, run.train != 0, g_bn_b[m], g_bn_g[m]));     // (+ the two parameter gradients)
        XG_TRY(gemm_tn_cs(st, gm_tail, N, R, F[m], w.dX[m], R, feats[m], F[m], g_emb_w[m], F[m], g_emb_b[m]));
    }
    return ss.chain2_into_aux();              // the caller's join() of aux then covers the second side chain too
}

// ================================================================== decoder pieces
int init_hidden(hipStream_t st, const XgDims& d, const XgParams& p, const float* V, const float* feat_mask, Ws& w,
                float* h1, float* c1, float* h2, float* c2) {
    const int B = d.B, R = d.R;
    XG_TRY(xgk_masked_mean(st, V, feat_mask, w.vbar, B, d.K, R));
    float* out[4] = {h1, c1, h2, c2};
    const float* wt[4] = {p.ih1_w, p.ic1_w, p.ih2_w, p.ic2_w};
    const float* bs[4] = {p.ih1_b, p.ic1_b, p.ih2_b, p.ic2_b};
    if (R % 4 == 0) {                          // the four (B,R)x(R,R) products as four jobs of one skinny launch
        SkArgs sk{};
        sk.njobs = 4;
        for (int j = 0; j < 4; ++j) {
            sk.job[j] = job_store(B, R, out[j], R, false);
            sk.job[j].nseg = 1; sk.job[j].seg[0] = seg_nt(w.vbar, R, wt[j], R, R); sk.job[j].bias[0] = bs[j];
        }
        return xgk_skinny(st, sk, sk_mode(w));
    }
    for (int j = 0; j < 4; ++j) XG_TRY(xgk_linear(st, w.gm, B, R, R, w.vbar, R, wt[j], bs[j], out[j], R));
    return XG_OK;
}

// initial decoder state (main stream) beside the hoisted v2a(V) projection (second auxiliary stream), sub_modules.py:677
int init_and_vproj(Streams& ss, const XgDims& d, const XgParams& p, const float* feat_mask, Ws& w) {
    const int N = d.B * d.K;
    XG_TRY(ss.fork2());
    XG_TRY(lin16(ss.aux2, w.gm, N, d.A, d.R, w.Venc, m16(w, w.Venc), d.R, p.v2a_w, w16(w, W16_V2A), p.v2a_b, w.vproj, d.A));
    XG_TRY(init_hidden(ss.main, d, p, w.Venc, feat_mask, w, w.H1, w.C1, w.H2, w.C2));
    return ss.join2();
}

struct StepIO {
    const float* xt;      // (B,E) embedding rows of the step's tokens, or null: gathered from `tok` inside the products
    const int64_t* tok;   // (B) tokens (row gather on embed.weight) when xt == null
    const float* pos;     // (B,R) raw POS feature: the gate runs inside the step when pre1 == null
    float* gp;            // (B,R) gate values g (saved for backward), written when pre1 == null
    float* posg;          // (B,R) gated POS feature (input when pre1 != null, output otherwise)
    const float* pre1;    // (B,4R) hoisted xt/pos' contribution of cell 1, or null (computed here)
    const float* mask; int ldm;
    const float *h1, *c1, *h2, *c2;   // previous state (B,R) contiguous
    float *h1o, *c1o, *h2o, *c2o;     // new state (may alias the previous state on the packed path: see core_step)
    float *P, *alpha, *af, *g1, *g2;  // saved per-step tensors (alpha / gates may be scratch)
    int t;
    bool half_attn;       // stand-alone attention in its half-CU form (a background product runs beside the steps)
    const RollSelectArgs* sel;   // rollout steps t >= 1 (packed form, xt == null): the step's first launch CHOOSES the tokens (xg_select.h) and writes tok / mask / xt rows itself
};

// the packed-weight form of the step needs 16-byte rows everywhere (R % 8 covers R; E and A are checked here)
inline bool step_packed(const Ws& w, const XgDims& d) { return w.packed && d.R % 8 == 0 && d.E % 4 == 0 && d.A % 4 == 0; }

// attention + the two cells for one step (sub_modules.py:677-684).
//
// Packed form, 3 launches, every launch a multi-job skinny launch (xg_step.hip):
//   L1  what the attention and cell 1 wait for:  p = h2a([h1;h2]) | teacher forcing: cell 1 (h1 W_h2h1 + hoisted token side)
//                                                                 | rollout: POS gate, S1' = xt W_i2h1 + h1 W_h2h1 + b
//   L2  attention, two workgroups per video (needs p) | rollout: cell 1 = pos' W_a2h1 + S1' | S2' = h2 W_h2h2 + b
//   L3  cell 2 = h1' W_i2h2 + (c / s) W_a2h2 + S2'   (normalises the attention context while it stages it)
// No launch reads the old state as a matrix operand after L1, and a cell epilogue reads its own (b, j) element of the old
// h / c before it writes the new one: the state may be updated IN PLACE (xg_step_fwd) with no copy.
// LDS-staged form (no packed weights): [p || cell 1 / gate] -> (cell 1) -> attention -> cell 2 as in round 1.
int core_step(hipStream_t st, const XgDims& d, const XgParams& p, const XgRun& run, Ws& w, const float* V,
              const float* vproj, const StepIO& s) {
    const int B = d.B, R = d.R, A = d.A, E = d.E;
    LstmFwdArgs a{};
    a.add = s.pre1; a.ldadd = 4 * R;
    a.c_prev = s.c1; a.ldcp = R; a.h_prev = s.h1; a.ldhp = R; a.mask = s.mask; a.ldm = s.ldm;
    a.gates = s.g1; a.ldg = 4 * R; a.c_out = s.c1o; a.ldco = R; a.h_out = s.h1o; a.ldho = R;
    a.B = B; a.R = R; a.order = XG_ORDER_IFOG; a.mask_mode = XG_MASK_HOLD;
    a.drop = xg_make_drop(&run, XG_SITE_L1, s.t);
    LstmFwdArgs c{};
    c.add = nullptr; c.ldadd = 0;
    c.c_prev = s.c2; c.ldcp = R; c.h_prev = s.h2; c.ldhp = R; c.mask = s.mask; c.ldm = s.ldm;
    c.gates = s.g2; c.ldg = 4 * R; c.c_out = s.c2o; c.ldco = R; c.h_out = s.h2o; c.ldho = R;
    c.B = B; c.R = R; c.order = XG_ORDER_IFOG; c.mask_mode = XG_MASK_HOLD;
    c.drop = xg_make_drop(&run, XG_SITE_L2, s.t);
    // The step as ONE dataflow launch (xg_dstep.hip) exists for measurement only (-DXG_DIAG build, XG_DSTEP=1): at 128 rows it
    // takes 77 us against 46 us for the three launches below (DESIGN.md 4.3 has the in-kernel timeline); it is ahead only
    // below ~16 rows (30 vs 34 us at 8 rows).
#ifdef XG_DIAG
    static const bool use_dstep = xg_diag_env("XG_DSTEP") != nullptr;
    if (use_dstep && !s.sel && w.pk.dtype != 2 && step_packed(w, d) && xgk_dstep_ok(d) && ((uintptr_t)V % 16 == 0) && ((uintptr_t)vproj % 16 == 0) &&
        ((uintptr_t)p.a2w_w % 16 == 0)) {
        DStepArgs a2{};
        a2.B = B; a2.R = R; a2.A = A; a2.E = E; a2.K = d.K; a2.V1 = d.V - 1;
        a2.h1 = s.h1; a2.c1 = s.c1; a2.h2 = s.h2; a2.c2 = s.c2;
        a2.h1o = s.h1o; a2.c1o = s.c1o; a2.h2o = s.h2o; a2.c2o = s.c2o;
        a2.copy_back = (s.h1o == s.h1 || s.h2o == s.h2) ? 1 : 0;             // in-place state: new h rows via scratch
        a2.h1w = a2.copy_back ? w.state_tmp : s.h1o;
        a2.h2w = a2.copy_back ? w.state_tmp + (size_t)B * R : s.h2o;
        a2.xt = s.xt; a2.tok = s.tok; a2.embed = p.embed_w;
        a2.pos = s.pos; a2.gp = s.gp; a2.posg = s.posg; a2.pre1 = s.pre1;
        a2.mask = s.mask; a2.ldm = s.ldm;
        a2.P = s.P; a2.alpha = s.alpha; a2.af = s.af; a2.g1 = s.g1; a2.g2 = s.g2;
        a2.V = V; a2.vproj = vproj; a2.a2w = p.a2w_w;
        a2.pk_h2a1 = w.pk.m[PK_H2A1]; a2.pk_h2a2 = w.pk.m[PK_H2A2]; a2.pk_dgate = w.pk.m[PK_DGATE];
        a2.pk_l1_i2h = w.pk.m[PK_L1_I2H]; a2.pk_l1_a2h = w.pk.m[PK_L1_A2H]; a2.pk_l1_h2h = w.pk.m[PK_L1_H2H];
        a2.pk_l2_i2h = w.pk.m[PK_L2_I2H]; a2.pk_l2_a2h = w.pk.m[PK_L2_A2H]; a2.pk_l2_h2h = w.pk.m[PK_L2_H2H];
        a2.h2a_b = p.h2a_b; a2.dgate_b = p.dgate_b; a2.l1_i2h_b = p.l1_i2h_b; a2.l1_a2h_b = p.l1_a2h_b; a2.l1_h2h_b = p.l1_h2h_b;
        a2.l2_i2h_b = p.l2_i2h_b; a2.l2_a2h_b = p.l2_a2h_b; a2.l2_h2h_b = p.l2_h2h_b;
        a2.ctr = w.dsync;
        a2.drop_gate = xg_make_drop(&run, XG_SITE_DGATE, s.t); a2.drop_l1 = a.drop; a2.drop_l2 = c.drop;
        return xgk_dstep(st, a2, w.gm);
    }
#endif
    if (step_packed(w, d)) {
        // xt as a matrix operand: the materialised rows, or embed.weight gathered by token
        auto xt_seg = [&](int which, const float* W) {
            SkSeg g = seg_nt(w, which, s.xt ? s.xt : p.embed_w, E, W, E, E);
            if (!s.xt) { g.gather = s.tok; g.gather_max = d.V - 1; }
            return g;
        };
        // the attention rides in the second launch as two workgroups per video (SK_EPI_ATTN) when its shapes allow
        // (rollout form only: there cell 1 and S2' share the launch with it.  Under teacher forcing the three candidates --
        // B: S2' in launch 1 + stand-alone attention, D: cell 2 keeps its h2 segment (K = 3R) + stand-alone attention,
        // E: the fused attention like the rollout form -- measure 6.77 / 6.73 / 6.74 ms per iteration (the iteration is
        // throughput-bound across three streams, not bound by this chain); D is the simplest and the default.)
        // (at hidden 1024 / 40 frames E wins instead: 8.51 vs 8.63 ms -- the stand-alone attention is then 24 us per step)
        const char xe_form = R >= 1024 ? 'E' : 'D';
        const bool fused_attn = (!s.pre1 || xe_form == 'E') && A <= 2048 && d.K <= 128 && ((uintptr_t)V % 8 == 0) &&
                                ((uintptr_t)vproj % 16 == 0) && ((uintptr_t)p.a2w_w % 16 == 0);
        // S2' = h2 W_h2h2 + b rides in launch 1 (teacher forcing form B; the rollout form beyond 64 rows, where the three
        // launches are then 512 / 512 / 256 workgroups, one round each, instead of 256 / 768 / 256: 49.4 -> 46.5 us per step
        // at 128 rows) or in launch 2 (rollout form up to 64 rows: 37.1 us against 39.2 us in launch 1)
        const bool s2_first = !s.pre1 && B > 64;
        const bool s2_in_cell2 = s.pre1 != nullptr && xe_form == 'D';       // ... or stays a segment of cell 2 (measured for the rollout form at 128 rows too: 48.3 us)
        SkArgs k1{}, k2{}, k3{};
        int n1 = 0, n2 = 0, n3 = 0;
        // the state may be updated IN PLACE (xg_step_fwd): cell 1 then runs in launch 2 beside products that still read the
        // old h1, so its new h1 goes to a scratch row block and rides back into the state with launch 3
        const bool h1_alias = s.h1o == s.h1;
        float* h1_new = h1_alias ? w.state_tmp : s.h1o;
        // ---- launch 1: what the attention and cell 1 wait for (jobs that gather come first: the index load is one more
        // dependent round trip)
        // Job order is dispatch order, and with two workgroups per CU resident at once a CU ends up with tile c of the first 256
        // and tile c of the next 256: the 256 S2' tiles (K = R) go FIRST so that every heavy p tile (K = 2R) is paired with a light
        // one instead of with another p tile (round 4: 45.3 -> 44.5 us per step at 128 rows)
        auto s2_job_early = [&](SkJob& j) {
            j = job_store(B, 4 * R, w.S2, 4 * R, false);
            j.cell_cols = 1; j.R = R; j.nseg = 1;
            j.seg[0] = seg_nt(w, PK_L2_H2H, s.h2, R, p.l2_h2h_w, R, R); j.bias[0] = p.l2_h2h_b;
        };
        if (s2_first) s2_job_early(k1.job[n1++]);
        if (!s.pre1) {  // POS gate: pos' = dropout(relu(W_g xt + b)) * pos + pos                          :682
            SkJob& j = k1.job[n1++];
            j = job_store(B, R, s.gp, R, false);
            j.epi = SK_EPI_GATE; j.nseg = 1;
            j.seg[0] = xt_seg(PK_DGATE, p.dgate_w); j.bias[0] = p.dgate_b;
            j.gate_t = s.pos; j.ldt = R; j.gate_y = s.posg; j.ldy = R;
            j.drop = xg_make_drop(&run, XG_SITE_DGATE, s.t);
            if (s.sel) {
                if (s.xt) return XG_EINVAL;
                j.select = 1; k1.sel = *s.sel;
            }
        }
        {   // p = h2a([h1 ; h2])                                                                         :677
            SkJob& j = k1.job[n1++];
            j = job_store(B, A, s.P, A, false);
            j.nseg = 2;
            j.seg[0] = seg_nt(w, PK_H2A1, s.h1, R, p.h2a_w, 2 * R, R);
            j.seg[1] = seg_nt(w, PK_H2A2, s.h2, R, p.h2a_w + R, 2 * R, R);
            j.bias[0] = p.h2a_b;
        }
        if (s.pre1) {   // teacher forcing: cell 1 is only its recurrent product (token side hoisted)        :683
            SkJob& j = k1.job[n1++];
            j = job_lstm(a);
            j.nseg = 1;
            j.seg[0] = seg_nt(w, PK_L1_H2H, s.h1, R, p.l1_h2h_w, R, R); j.bias[0] = p.l1_h2h_b;
        }
        auto s2_job = [&](SkJob& j) {                     // S2' = h2 W_h2h2 + b (gate-major, cell tiling) -> w.S2
            j = job_store(B, 4 * R, w.S2, 4 * R, false);
            j.cell_cols = 1; j.R = R;
            j.nseg = 1;
            j.seg[0] = seg_nt(w, PK_L2_H2H, s.h2, R, p.l2_h2h_w, R, R); j.bias[0] = p.l2_h2h_b;
        };
        if (fused_attn) {   // the attention's accumulators start from zero
            SkJob& j = k1.job[n1++];
            j = SkJob{};
            j.epi = SK_EPI_ZERO; j.M = 1; j.N = B * R + ((B + 3) & ~3); j.C = w.AFU;
        }
        k1.njobs = n1;
        XG_TRY(xgk_skinny(st, k1, sk_mode(w)));
        // ---- launch 2: attention || rollout: cell 1 = h1 W_h2h + pos' W_a2h + xt W_i2h || S2'
        auto attn_job = [&](SkJob& j) {
            j = SkJob{};
            j.epi = SK_EPI_ATTN; j.M = B; j.R = R; j.attn_K = d.K; j.attn_A = A;
            j.attn_p = s.P; j.attn_q = vproj; j.attn_v = V; j.attn_w = p.a2w_w;
            j.attn_ex = s.alpha; j.attn_s = w.ATS; j.attn_c = w.AFU;
        };
        if (fused_attn) attn_job(k2.job[n2++]);
        if (!s.pre1) {
            SkJob& j = k2.job[n2++];
            a.h_out = h1_new;
            j = job_lstm(a);
            j.nseg = 3;
            j.seg[0] = seg_nt(w, PK_L1_H2H, s.h1, R, p.l1_h2h_w, R, R); j.bias[0] = p.l1_h2h_b;
            j.seg[1] = seg_nt(w, PK_L1_A2H, s.posg, R, p.l1_a2h_w, R, R); j.bias[1] = p.l1_a2h_b;
            j.seg[2] = xt_seg(PK_L1_I2H, p.l1_i2h_w); j.bias[2] = p.l1_i2h_b;
        }
        if (!s2_first && !s2_in_cell2) s2_job(k2.job[n2++]);
        k2.njobs = n2;
        if (n2 > 0) XG_TRY(xgk_skinny(st, k2, sk_mode(w)));
        if (!fused_attn) XG_TRY(xgk_attn_fwd(st, s.P, vproj, V, p.a2w_w, s.alpha, s.af, B, d.K, R, A, s.half_attn));
        // ---- launch 3: cell 2 = h1' W_i2h + af W_a2h + S2'                                                :684
        if (!s2_in_cell2) { c.add = w.S2; c.ldadd = 4 * R; }
        {
            SkJob& j = k3.job[n3++];
            j = job_lstm(c);
            j.nseg = 2;
            if (s2_in_cell2) { j.nseg = 3; j.seg[2] = seg_nt(w, PK_L2_H2H, s.h2, R, p.l2_h2h_w, R, R); j.bias[2] = p.l2_h2h_b; }
            j.seg[0] = seg_nt(w, PK_L2_I2H, s.pre1 ? s.h1o : h1_new, R, p.l2_i2h_w, R, R); j.bias[0] = p.l2_i2h_b;
            j.seg[1] = seg_nt(w, PK_L2_A2H, fused_attn ? w.AFU : s.af, R, p.l2_a2h_w, R, R); j.bias[1] = p.l2_a2h_b;
            if (fused_attn) {   // af = c / s while it is staged; the n-tiles store af and normalise alpha between them
                SkSeg& g = j.seg[1];
                g.row_scale = w.ATS; g.scaled_out = s.af; g.ld_out = R; g.ex = s.alpha; g.ex_ld = d.K; g.ex_K = d.K;
            }
        }
        if (!s.pre1 && h1_alias) {                           // the new h1 returns to the in-place state
            SkJob& j = k3.job[n3++];
            j = SkJob{};
            j.epi = SK_EPI_COPY; j.M = 1; j.N = B * R; j.C = s.h1o; j.seg[0].A = h1_new;
        }
        k3.njobs = n3;
        XG_TRY(xgk_skinny(st, k3, sk_mode(w)));
        return XG_OK;
    }
    if (!s.xt) return XG_EINVAL;                 // the token gather exists on the packed path only
    if (R % 8 == 0) {
        SkArgs k1{};
        k1.njobs = 2;
        k1.job[0] = job_store(B, A, s.P, A, false);
        k1.job[0].nseg = 2;
        k1.job[0].seg[0] = seg_nt(s.h1, R, p.h2a_w, 2 * R, R);
        k1.job[0].seg[1] = seg_nt(s.h2, R, p.h2a_w + R, 2 * R, R);
        k1.job[0].bias[0] = p.h2a_b;
        k1.job[1] = job_lstm(a);
        if (s.pre1) {
            k1.job[1].nseg = 1;
            k1.job[1].seg[0] = seg_nt(s.h1, R, p.l1_h2h_w, R, R); k1.job[1].bias[0] = p.l1_h2h_b;
            XG_TRY(xgk_skinny(st, k1, sk_mode(w)));
        } else {
            // rollout form: [p || POS gate] first (the gate feeds cell 1), then cell 1 with all three products
            SkJob cell1 = k1.job[1];
            k1.job[1] = job_store(B, R, s.gp, R, false);
            k1.job[1].epi = SK_EPI_GATE; k1.job[1].nseg = 1;
            k1.job[1].seg[0] = seg_nt(s.xt, E, p.dgate_w, E, E); k1.job[1].bias[0] = p.dgate_b;
            k1.job[1].gate_t = s.pos; k1.job[1].ldt = R; k1.job[1].gate_y = s.posg; k1.job[1].ldy = R;
            k1.job[1].drop = xg_make_drop(&run, XG_SITE_DGATE, s.t);
            XG_TRY(xgk_skinny(st, k1, sk_mode(w)));
            SkArgs k1b{};
            k1b.njobs = 1;
            k1b.job[0] = cell1;
            k1b.job[0].nseg = 3;
            k1b.job[0].seg[0] = seg_nt(s.xt, E, p.l1_i2h_w, E, E); k1b.job[0].bias[0] = p.l1_i2h_b;
            k1b.job[0].seg[1] = seg_nt(s.posg, R, p.l1_a2h_w, R, R); k1b.job[0].bias[1] = p.l1_a2h_b;
            k1b.job[0].seg[2] = seg_nt(s.h1, R, p.l1_h2h_w, R, R); k1b.job[0].bias[2] = p.l1_h2h_b;
            XG_TRY(xgk_skinny(st, k1b, sk_mode(w)));
        }
        XG_TRY(xgk_attn_fwd(st, s.P, vproj, V, p.a2w_w, s.alpha, s.af, B, d.K, R, A));
        SkArgs k2{};
        k2.njobs = 1;
        k2.job[0] = job_lstm(c);
        k2.job[0].nseg = 3;
        k2.job[0].seg[0] = seg_nt(s.h1o, R, p.l2_i2h_w, R, R); k2.job[0].bias[0] = p.l2_i2h_b;
        k2.job[0].seg[1] = seg_nt(s.af, R, p.l2_a2h_w, R, R); k2.job[0].bias[1] = p.l2_a2h_b;
        k2.job[0].seg[2] = seg_nt(s.h2, R, p.l2_h2h_w, R, R); k2.job[0].bias[2] = p.l2_h2h_b;
        XG_TRY(xgk_skinny(st, k2, sk_mode(w)));
        return XG_OK;
    }
    // generic path (R not a multiple of 8): plain GEMMs + pointwise cell kernels
    if (!s.pre1) {
        XG_TRY(xgk_linear(st, w.gm, B, R, E, s.xt, E, p.dgate_w, p.dgate_b, s.gp, R, true));
        XG_TRY(xgk_gate_fwd(st, s.gp, R, s.pos, R, 0, s.posg, R, B, R, xg_make_drop(&run, XG_SITE_DGATE, s.t), B, 1 << 30, 1, B));
    }
    XG_TRY(xgk_gemm(st, w.gm, false, true, B, A, R, s.h1, R, p.h2a_w, 2 * R, s.P, A, p.h2a_b, false, false));
    XG_TRY(xgk_gemm(st, w.gm, false, true, B, A, R, s.h2, R, p.h2a_w + R, 2 * R, s.P, A, nullptr, false, true));
    XG_TRY(xgk_attn_fwd(st, s.P, vproj, V, p.a2w_w, s.alpha, s.af, B, d.K, R, A));
    XG_TRY(xgk_linear(st, w.gm, B, 4 * R, R, s.h1, R, p.l1_h2h_w, p.l1_h2h_b, w.S, 4 * R));
    if (!s.pre1) {
        XG_TRY(xgk_linear(st, w.gm, B, 4 * R, E, s.xt, E, p.l1_i2h_w, p.l1_i2h_b, w.S, 4 * R, false, true));
        XG_TRY(xgk_linear(st, w.gm, B, 4 * R, R, s.posg, R, p.l1_a2h_w, p.l1_a2h_b, w.S, 4 * R, false, true));
    }
    a.s = w.S; a.lds_ = 4 * R;
    XG_TRY(xgk_lstm_fwd(st, a));
    XG_TRY(xgk_linear(st, w.gm, B, 4 * R, R, s.h1o, R, p.l2_i2h_w, p.l2_i2h_b, w.S2, 4 * R));
    XG_TRY(xgk_linear(st, w.gm, B, 4 * R, R, s.af, R, p.l2_a2h_w, p.l2_a2h_b, w.S2, 4 * R, false, true));
    XG_TRY(xgk_linear(st, w.gm, B, 4 * R, R, s.h2, R, p.l2_h2h_w, p.l2_h2h_b, w.S2, 4 * R, false, true));
    c.s = w.S2; c.lds_ = 4 * R;
    XG_TRY(xgk_lstm_fwd(st, c));
    return XG_OK;
}

// teacher-forced decoder, states into w.H1/C1/H2/C2 (SAModel.py:85-112)
// everything of the teacher-forced decoder that depends only on the tokens (SAModel.py:105, sub_modules.py:682 and
// the xt / pos' products of lstm_1): runs on the auxiliary stream under the CG encoder
int decoder_tokens_xe(hipStream_t sx, const XgDims& d, const XgParams& p, const XgBatch& x, const XgRun& run, Ws& w) {
    const int B = d.B, R = d.R, E = d.E, T = d.T, TB = T * B;
    XG_TRY(xgk_embed_gather(sx, p.embed_w, E, x.seq, /*inner=*/B, /*s_inner=*/T, /*s_outer=*/1, TB, d.V, w.Xe, E));
    XG_TRY(xgk_linear(sx, w.gm, TB, R, E, w.Xe, E, p.dgate_w, p.dgate_b, w.GP, R, true));                 // :682 gate, all steps
    XG_TRY(xgk_gate_fwd(sx, w.GP, R, x.pos_feats, R, B, w.POSG, R, TB, R, xg_make_drop(&run, XG_SITE_DGATE, 0),
                        /*step=row/B*/ B, 1 << 30, /*b=row%B*/ 1, B));
    XG_TRY(xgk_linear(sx, w.gm, TB, 4 * R, E, w.Xe, E, p.l1_i2h_w, p.l1_i2h_b, w.PRE1, 4 * R));
    XG_TRY(cvt16(sx, w, w.POSG, (size_t)TB * R));
    XG_TRY(lin16(sx, w.gm, TB, 4 * R, R, w.POSG, m16(w, w.POSG), R, p.l1_a2h_w, w16(w, W16_L1_A2H), p.l1_a2h_b, w.PRE1, 4 * R, false, true));
    return XG_OK;
}

// teacher-forced decoder, states into w.H1/C1/H2/C2 (SAModel.py:85-112).  The vocabulary product of the first half of
// the steps is issued on the auxiliary stream as soon as those steps are done; *logit_rows_done receives the number of
// (t,b) rows whose logits are already under way there.
// early_loss (fused loss path): the cross-entropy rows of those steps follow their logits on the auxiliary stream, under
// the remaining steps (w.sums must have been zeroed on the main stream before).
int decoder_fwd_xe(Streams& ss, const XgDims& d, const XgParams& p, const XgBatch& x, const XgRun& run, Ws& w,
                   int* logit_rows_done, bool early_loss = false) {
    hipStream_t st = ss.main;
    const int B = d.B, R = d.R, A = d.A, E = d.E, T = d.T;
    const size_t BR = (size_t)B * R;
    XG_TRY(init_and_vproj(ss, d, p, x.feat_mask, w));
    XG_TRY(zero_dsync(st, w));
    XG_TRY(ss.join());                                                                              // token-side products
    const int th = (ss.overlap() && T >= 4) ? T / 2 : 0;          // the early product starts behind step T / 2 (split points swept in rounds 3-5: flat)
    // ... as a background product, with the stand-alone attention in its half-CU form beside it: the 128-VGPR attention
    // needs an EMPTY CU and waited for the whole persistent product (255 us: the chain simply stopped).  6.16 -> 6.10 ms.
    const bool fwd_bg = th > 0 && w.gm == 0;
    *logit_rows_done = 0;
    if (run.prof_event0 && hipEventRecord(static_cast<hipEvent_t>(run.prof_event0), st) != hipSuccess) return XG_EHIP;
    for (int t = 0; t < T; ++t) {
        StepIO s{};
        s.xt = w.Xe + (size_t)t * B * E; s.posg = w.POSG + t * BR; s.pre1 = w.PRE1 + (size_t)t * B * 4 * R;
        s.mask = x.seq_mask + t; s.ldm = T;
        s.h1 = w.H1 + t * BR; s.c1 = w.C1 + t * BR; s.h2 = w.H2 + t * BR; s.c2 = w.C2 + t * BR;
        s.h1o = w.H1 + (t + 1) * BR; s.c1o = w.C1 + (t + 1) * BR; s.h2o = w.H2 + (t + 1) * BR; s.c2o = w.C2 + (t + 1) * BR;
        s.P = w.P + (size_t)t * B * A; s.alpha = w.ALPHA + (size_t)t * B * d.K; s.af = w.AF + t * BR;
        s.g1 = w.G1 + (size_t)t * B * 4 * R; s.g2 = w.G2 + (size_t)t * B * 4 * R; s.t = t;
        s.half_attn = fwd_bg;
        XG_TRY(core_step(st, d, p, run, w, w.Venc, w.vproj, s));
        if (th > 0 && t == th - 1) {
            XG_TRY(ss.fork());
            XG_TRY(cvt16(ss.aux, w, w.H2 + BR, (size_t)th * B * R));
            XG_TRY(lin16(ss.aux, w.gm | (fwd_bg ? XGK_GEMM_BG : 0), th * B, d.V, R, w.H2 + BR, m16(w, w.H2 + BR), R, p.logit_w,
                         w16(w, W16_LOGIT), p.logit_b, w.LOGITS, d.V));
            *logit_rows_done = th * B;
            if (early_loss)
                XG_TRY(xgk_xent_fwd(ss.aux, w.LOGITS, d.V, x.seq, x.seq_mask, nullptr, B, T, d.V, 1, w.LSE, w.sums, 0, th * B, false));
        }
    }
    if (run.prof_event1 && hipEventRecord(static_cast<hipEvent_t>(run.prof_event1), st) != hipSuccess) return XG_EHIP;
    return XG_OK;
}

// classifier hidden + logits for the stacked outputs H2[1..T] (SAModel.py:109-110)
int heads_fwd_logits(Streams& ss, const XgDims& d, const XgParams& p, const XgRun& run, Ws& w, int rows, int rows_done) {
    hipStream_t st = ss.main;
    const int B = d.B, R = d.R;
    const float* Hout = w.H2 + (size_t)B * R;
    XG_TRY(cvt16(st, w, Hout + (size_t)rows_done * R, (size_t)(rows - rows_done) * R));
    XG_TRY(lin16(st, w.gm, rows - rows_done, d.V, R, Hout + (size_t)rows_done * R, m16(w, Hout + (size_t)rows_done * R), R, p.logit_w,
                 w16(w, W16_LOGIT), p.logit_b, w.LOGITS + (size_t)rows_done * d.V, d.V));
    XG_TRY(xgk_linear(st, w.gm, rows, d.H, R, Hout, R, p.cls0_w, p.cls0_b, w.HC, d.H, true));
    XG_TRY(xgk_gate_fwd(st, w.HC, d.H, nullptr, 0, 0, nullptr, 0, rows, d.H, xg_make_drop(&run, XG_SITE_CLS, 0), B, 1 << 30,
                        1, B));
    XG_TRY(xgk_linear(st, w.gm, rows, d.C, d.H, w.HC, d.H, p.cls3_w, p.cls3_b, w.CL, d.C));
    return ss.join();                         // first-half logits from the auxiliary stream
}

// Shared reverse-time pass.  On entry: DH2OUT (TB,R) holds d(loss)/d(h2'_t) from the heads.
// mask: element (b,t) at mask[b*ldm + t*tstride].  tokens likewise for the embedding scatter.
int decoder_bwd_core(Streams& ss, const XgDims& d, const XgParams& p, const XgParams& g, const XgBatch& x,
                     const XgRun& run, Ws& w, const float* mask, int ldm, int mask_tstride, const int64_t* tok,
                     int tok_bstride, int tok_tstride) {
    hipStream_t st = ss.main;
    const int B = d.B, K = d.K, R = d.R, A = d.A, E = d.E, T = d.T, TB = T * B, N = B * K;
    (void)TB;
    const size_t BR = (size_t)B * R;
    // The reverse-time recurrence splits into two chains.  Chain 2 (cell 2 + attention: dh2, dc2) never reads anything
    // chain 1 (cell 1: dh1, dc1) produces, and only chain 2's products (dAF, dE) feed the encoder backward: it runs on
    // the main stream, 3 launches per step.  Chain 1 (1 launch per step) follows ONE STEP BEHIND on the second auxiliary
    // stream: what cell 1's output at step t-1 receives from chain 2 -- ds2[t-1] W_i2h2 and dp[t] W_h2a[:, :R] -- are two
    // more segments of the product that carries dh1 back (ds1[t] W_h2h1), so nothing is batched after the loop and the
    // chain fills the part of chain 2's time in which the vocabulary head's weight gradients are already done.  Its
    // results only feed parameter gradients.
    // The pointwise LSTM backward of step t-1 runs in the epilogue of the product that completes its dh.
    const bool fuse = R % 4 == 0;
    int cur = 0;
    if (w.zeroed) {
        XG_TRY(ss.join2());                      // the zero block (zero_backward_block, second auxiliary stream)
    } else {
        for (int j = 0; j < 4; ++j) ZERO(w.dst[0][j], BR);
        ZERO(w.DAF, (size_t)T * BR);             // the dAF products accumulate (split-K across workgroups)
        XG_TRY(zero_tickets(st, w, SK_MAX_JOBS));
    }
    auto cell2_bwd = [&](int t, int c) {          // backward of cell 2 at step t, reading the carried state of parity c
        LstmBwdArgs a{};
        a.gates = w.G2 + (size_t)t * B * 4 * R; a.ldg = 4 * R;
        a.c_prev = w.C2 + t * BR; a.ldcp = R; a.c_out = w.C2 + (t + 1) * BR; a.ldco = R;
        a.mask = mask + (size_t)t * mask_tstride; a.ldm = ldm;
        a.dh_out = w.dst[c][2]; a.lddh = R; a.dh_add = w.DH2OUT + t * BR; a.lddha = R;
        a.dc_out = w.dst[c][3]; a.lddc = R;
        a.ds = w.DS2 + (size_t)t * B * 4 * R; a.ldds = 4 * R;
        a.dc_prev = w.dst[c ^ 1][3]; a.lddcp = R; a.dh_prev = w.dst[c ^ 1][2]; a.lddhp = R;
        a.B = B; a.R = R; a.order = XG_ORDER_IFOG; a.mask_mode = XG_MASK_HOLD;
        a.drop = xg_make_drop(&run, XG_SITE_L2, t);
        return a;
    };
    hipStream_t s1 = ss.aux2;                 // chain 1 and what depends on it
    int cur1 = 0;
    auto cell1_bwd = [&](int t, int c, const float* dh_add) {
        LstmBwdArgs a{};
        a.gates = w.G1 + (size_t)t * B * 4 * R; a.ldg = 4 * R;
        a.c_prev = w.C1 + t * BR; a.ldcp = R; a.c_out = w.C1 + (t + 1) * BR; a.ldco = R;
        a.mask = mask + (size_t)t * mask_tstride; a.ldm = ldm;
        a.dh_out = w.dst[c][0]; a.lddh = R; a.dh_add = dh_add; a.lddha = dh_add ? R : 0; a.dc_out = w.dst[c][1]; a.lddc = R;
        a.ds = w.DS1 + (size_t)t * B * 4 * R; a.ldds = 4 * R;
        a.dc_prev = w.dst[c ^ 1][1]; a.lddcp = R; a.dh_prev = w.dst[c ^ 1][0]; a.lddhp = R;
        a.B = B; a.R = R; a.order = XG_ORDER_IFOG; a.mask_mode = XG_MASK_HOLD;
        a.drop = xg_make_drop(&run, XG_SITE_L1, t);
        return a;
    };
    // what cell 1's output at step t receives from chain 2, as segments:  ds2[t] W_i2h2 (+ dp[t+1] W_h2a[:, :R])
    auto from_chain2 = [&](SkJob& j, int first, int t) {
        j.seg[first] = seg_nn(w, PKB_L2_I2H, w.DS2 + (size_t)t * B * 4 * R, 4 * R, p.l2_i2h_w, R, 4 * R);
        int n = first + 1;
        if (t + 1 < T) j.seg[n++] = seg_nn(w, PKB_H2A1, w.DP + (size_t)(t + 1) * B * A, A, p.h2a_w, 2 * R, A);
        j.nseg = n;
    };
    // one step of chain 1, enqueued when chain 2 has finished step t (ds2[t-1], dp[t] exist)
    auto chain1_step = [&](int t) -> int {
        float* dh1p = w.dst[cur1 ^ 1][0];
        float* ds1 = w.DS1 + (size_t)t * B * 4 * R;
        if (!fuse || t == T - 1) {               // stand-alone cell backward: its dh contribution from chain 2 is materialised
            SkArgs sk{};
            sk.njobs = 1;
            sk.job[0] = job_store(B, R, w.DH1X + t * BR, R, false);
            from_chain2(sk.job[0], 0, t);
            XG_TRY(xgk_skinny(s1, sk, sk_mode(w)));
            XG_TRY(xgk_lstm_bwd(s1, cell1_bwd(t, cur1, w.DH1X + t * BR)));
        }
        SkArgs sk{};
        sk.njobs = 1;
        SkJob& j = sk.job[0];
        if (fuse && t > 0) j = job_lstm_bwd(cell1_bwd(t - 1, cur1 ^ 1, nullptr), dh1p, R);
        else j = job_store(B, R, dh1p, R, true);
        j.nseg = 1;
        j.seg[0] = seg_nn(w, PKB_L1_H2H, ds1, 4 * R, p.l1_h2h_w, R, 4 * R);
        if (fuse && t > 0) from_chain2(j, 1, t - 1);
        if (t == 0) {                            // the attention query of step 0 read the INITIAL h1
            j.seg[1] = seg_nn(w, PKB_H2A1, w.DP, A, p.h2a_w, 2 * R, A);
            j.nseg = 2;
        }
        allow_split(sk, 0, w); j.tickets = w.tickets + 2 * SKPART_INTS;       // chain 1 runs beside chain 2 / the encoder: own scratch
        // Chain 1 has slack (one launch per step against chain 2's three) and only feeds parameter gradients: it must not crowd
        // chain 2.  Launched one step behind with the full 8-way split (512 workgroups) it took the wave slots the attention
        // backward needed beside a background product (attention 40 us in situ against 12.7 alone).  So: a capped split
        // (round 4: 2-way, 128 workgroups), default wave priority, and its launches in batches of c1_lag steps (measured on MI355X,
        // tools/ubench/c1_sweep.sh: 6.08 -> 5.93 ms per iteration).
        // (round 5, lean kernel + the launcher's one-workgroup-per-CU split rule: a 4-way split -- 256 workgroups -- is the better
        //  cap again, 5.47-5.48 -> 5.44-5.45 ms; tools/ubench/caps_iter.sh)
        j.ksplit_cap = 4; j.low_prio = 1;
        XG_TRY(xgk_skinny(s1, sk, sk_mode(w)));
        cur1 ^= 1;
        return XG_OK;
    };
    // Batched weight gradients over a range of steps [t0, t1) (rows t0*B .. t1*B of the stacked per-step buffers; every
    // product accumulates into the zeroed gradient buffer, so a range split is exact up to summation order).  The steps
    // the reverse-time loop has already left are done BESIDE the loop on the auxiliary stream, behind the vocabulary
    // head's products: the loop alone leaves most of the chip idle, and whatever runs under it does not compete with
    // the encoder's backward afterwards.
    const int wgm = w.gm;
    auto wgrads_chain2 = [=, &w](hipStream_t sq, int t0, int t1) -> int {
        const int rows = (t1 - t0) * B;
        if (rows <= 0) return XG_OK;
        const size_t r0 = (size_t)t0 * B;
        const float* ds2 = w.DS2 + r0 * 4 * R;
        const float* dp = w.DP + r0 * A;
        const float *h1n = w.H1 + BR + r0 * R, *h1 = w.H1 + r0 * R, *h2 = w.H2 + r0 * R, *af = w.AF + r0 * R;
        // (bf16 mode: mirrors of this range's operands -- each is read by two or three of the products below)
        XG_TRY(cvt16(sq, w, ds2, (size_t)rows * 4 * R)); XG_TRY(cvt16(sq, w, dp, (size_t)rows * A));
        XG_TRY(cvt16(sq, w, h1, (size_t)(rows + B) * R)); XG_TRY(cvt16(sq, w, h2, (size_t)rows * R)); XG_TRY(cvt16(sq, w, af, (size_t)rows * R));
        XG_TRY(tn16(sq, wgm, rows, 4 * R, R, ds2, m16(w, ds2), 4 * R, h1n, m16(w, h1n), R, g.l2_i2h_w, R, g.l2_i2h_b, g.l2_a2h_b, g.l2_h2h_b));
        XG_TRY(tn16(sq, wgm, rows, 4 * R, R, ds2, m16(w, ds2), 4 * R, af, m16(w, af), R, g.l2_a2h_w, R));
        XG_TRY(tn16(sq, wgm, rows, 4 * R, R, ds2, m16(w, ds2), 4 * R, h2, m16(w, h2), R, g.l2_h2h_w, R));
        XG_TRY(tn16(sq, wgm, rows, A, R, dp, m16(w, dp), A, h1, m16(w, h1), R, g.h2a_w, 2 * R, g.h2a_b));
        XG_TRY(tn16(sq, wgm, rows, A, R, dp, m16(w, dp), A, h2, m16(w, h2), R, g.h2a_w + R, 2 * R));
        return XG_OK;
    };
    auto wgrads_chain1 = [=, &w](hipStream_t sq, int t0, int t1) -> int {
        const int rows = (t1 - t0) * B;
        if (rows <= 0) return XG_OK;
        const size_t r0 = (size_t)t0 * B;
        const float* ds1 = w.DS1 + r0 * 4 * R;
        const float *h1 = w.H1 + r0 * R, *posg = w.POSG + r0 * R;
        XG_TRY(cvt16(sq, w, ds1, (size_t)rows * 4 * R)); XG_TRY(cvt16(sq, w, h1, (size_t)rows * R)); XG_TRY(cvt16(sq, w, posg, (size_t)rows * R));
        XG_TRY(tn16(sq, wgm, rows, 4 * R, R, ds1, m16(w, ds1), 4 * R, h1, m16(w, h1), R, g.l1_h2h_w, R, g.l1_i2h_b, g.l1_a2h_b, g.l1_h2h_b));
        XG_TRY(tn16(sq, wgm, rows, 4 * R, E, ds1, m16(w, ds1), 4 * R, w.Xe + r0 * E, nullptr, E, g.l1_i2h_w, E));
        XG_TRY(tn16(sq, wgm, rows, 4 * R, R, ds1, m16(w, ds1), 4 * R, posg, m16(w, posg), R, g.l1_a2h_w, R));
        // input side of cell 1: pos' gate, embedding
        XG_TRY(nn16(sq, wgm, rows, R, 4 * R, ds1, m16(w, ds1), 4 * R, p.l1_a2h_w, w16(w, W16_L1_A2H), R, w.DPOSG + r0 * R, R, false));
        XG_TRY(nn16(sq, wgm, rows, E, 4 * R, ds1, m16(w, ds1), 4 * R, p.l1_i2h_w, w16(w, W16_L1_I2H), E, w.DXe + r0 * E, E, false));
        return XG_OK;
    };
    // (two ranges; fp32 products only: beside the bf16 GEMMs the loop loses more than the products gain, hidden-1024 iteration 8.50 -> 8.72 ms)
    const int wg_chunks = (ss.overlap() && T >= 8 && w.gm == 0) ? 2 : 1;
    int wg_hi = T, wg_mark = -1;                 // steps [wg_hi, T) already have their weight gradients enqueued
    int c1_next = T - 1;                         // the next step chain 1 has to do
    for (int t = T - 1; t >= 0; --t) {
        // dH of the early steps comes from the auxiliary stream; fused, step t-1's cell backward runs inside step t
        if (t == ss.dh_split_step - (fuse ? 0 : 1)) XG_TRY(ss.wait_mark(ss.dh_mark));
        float* dh2p = w.dst[cur ^ 1][2];
        float* ds2 = w.DS2 + (size_t)t * B * 4 * R;
        float* dp = w.DP + (size_t)t * B * A;
        float* daf = w.DAF + t * BR;
        if (!fuse || t == T - 1) XG_TRY(xgk_lstm_bwd(st, cell2_bwd(t, cur)));
        {   // s2 = h1' Wi + af Wa + h2 Wh : the two data gradients chain 2 needs now
            SkArgs sk{};
            sk.njobs = 2;
            sk.job[0] = job_store(B, R, daf, R, true);   sk.job[0].nseg = 1; sk.job[0].seg[0] = seg_nn(w, PKB_L2_A2H, ds2, 4 * R, p.l2_a2h_w, R, 4 * R);
            sk.job[1] = job_store(B, R, dh2p, R, true);  sk.job[1].nseg = 1; sk.job[1].seg[0] = seg_nn(w, PKB_L2_H2H, ds2, 4 * R, p.l2_h2h_w, R, 4 * R);
            allow_split(sk, 0, w); allow_split(sk, 1, w);
            XG_TRY(xgk_skinny(st, sk, sk_mode(w)));
        }
        XG_TRY(xgk_attn_bwd(st, daf, R, w.P + (size_t)t * B * A, w.vproj, w.Venc, p.a2w_w, w.ALPHA + (size_t)t * B * K,
                            w.DE + (size_t)t * B * K, dp, B, K, R, A));
        {   // into step t-1: dh2 += dp Wh2a[:, R:]  (+ cell 2's backward at t-1 on the completed dh2)
            SkArgs sk{};
            sk.njobs = 1;
            if (fuse && t > 0) sk.job[0] = job_lstm_bwd(cell2_bwd(t - 1, cur ^ 1), dh2p, R);
            else sk.job[0] = job_store(B, R, dh2p, R, true);
            sk.job[0].nseg = 1;
            sk.job[0].seg[0] = seg_nn(w, PKB_H2A2, dp, A, p.h2a_w + R, 2 * R, A);
            allow_split(sk, 0, w);
            XG_TRY(xgk_skinny(st, sk, sk_mode(w)));
        }
        cur ^= 1;
        // chain 2 has finished step t: chain 1 may do it.  One event per c1_lag steps (an event record between two dependent
        // launches of the main chain costs it ~6 us; chain 1 has slack: its launch is shorter than chain 2's three)
        constexpr int c1_lag = 7;
        bool boundary = false;                // t == ceil(k T / chunks) for some k in 1 .. chunks - 1
        for (int k = 1; k < wg_chunks; ++k) boundary = boundary || t == (k * T + wg_chunks - 1) / wg_chunks;
        const int lag = c1_lag < 1 ? 1 : c1_lag;
        if (t == 0 || (boundary && t < wg_hi) || (T - 1 - t) % lag == lag - 1) {
            XG_TRY(ss.fork2());
            for (int tt = c1_next; tt >= t; --tt) XG_TRY(chain1_step(tt));
            c1_next = t - 1;
        }
        if (boundary && t > 0 && t < wg_hi) {
            // DS2 / DP rows of steps >= t exist on main, DS1 rows of steps >= t on the second side chain
            XG_TRY(ss.fork());
            XG_TRY(wgrads_chain2(ss.aux, t, wg_hi));
            XG_TRY(ss.chain2_into_aux());
            XG_TRY(wgrads_chain1(ss.aux, t, wg_hi));
            wg_mark = ss.mark();              // the last one covers the earlier ones (same stream)
            if (wg_mark == -2) return XG_EHIP;
            wg_hi = t;
        }
    }
    const int cur2 = cur;
    XG_TRY(ss.fork());                        // chain 2 is complete: DS2, DP, DAF, DE
    hipStream_t sx = ss.aux;                  // parameter gradients that only need chain 2
    // ---- after the loop.  Main chain (the encoder backward waits for it): dVproj -> dV.  Everything else is a
    // parameter gradient and goes to the auxiliary stream, under the encoder's recurrent backward.
    // (dV first, as a plain store: accumulating on top of the product below it would read every element back)
    XG_TRY(xgk_attn_post_dV(st, w.P, w.vproj, p.a2w_w, w.DE, w.DVPROJ, g.a2w_w, w.ALPHA, w.DAF, R, (int64_t)BR, w.DV, T, B, K, A, R));
    XG_TRY(cvt16(st, w, w.DVPROJ, (size_t)N * A));
    XG_TRY(nn16(st, w.gm, N, R, A, w.DVPROJ, m16(w, w.DVPROJ), A, p.v2a_w, w16(w, W16_V2A), R, w.DV, R, true));
    // gradients wrt the initial state -> img_embed_* (init_hidden; vbar is detached: SAModel.py:59-62)
    const int cur1_end = cur1, wg_hi_end = wg_hi, wg_mark_end = wg_mark;
    auto tail = [=, &w, &ss]() -> int {
    const int cur1 = cur1_end, wg_hi = wg_hi_end, wg_mark = wg_mark_end;
    {
        float* gst[4] = {w.dst[cur1][0], w.dst[cur1][1], w.dst[cur2][2], w.dst[cur2][3]};
        float* gw[4] = {g.ih1_w, g.ic1_w, g.ih2_w, g.ic2_w};
        float* gb[4] = {g.ih1_b, g.ic1_b, g.ih2_b, g.ic2_b};
        for (int j = 0; j < 4; ++j) {
            hipStream_t sj = j < 2 ? s1 : sx;     // h1 / c1 come out of chain 1
            XG_TRY(gemm_tn_cs(sj, w.gm, B, R, R, gst[j], R, w.vbar, R, gw[j], R, gb[j]));
        }
    }
    // batched weight gradients over the steps the loop has not handed out yet
    XG_TRY(wgrads_chain2(sx, 0, wg_hi));      // (the cells' and h2a's bias gradients ride in these products: gemm_tn_cs)
    XG_TRY(wgrads_chain1(s1, 0, wg_hi));
    XG_TRY(ss.wait_mark2(wg_mark));           // DPOSG / DXe rows of the steps handed out in the loop
    XG_TRY(xgk_gate_bwd(s1, w.DPOSG, R, w.GP, R, x.pos_feats, R, B, w.DGP, R, nullptr, 0, false, TB, R,
                        xg_make_drop(&run, XG_SITE_DGATE, 0)));
    XG_TRY(gemm_tn_cs(s1, w.gm, TB, R, E, w.DGP, R, w.Xe, E, g.dgate_w, E, g.dgate_b));
    XG_TRY(gemm_nn(s1, w.gm, TB, E, R, w.DGP, R, p.dgate_w, E, w.DXe, E, true));
    XG_TRY(xgk_embed_scatter_add(s1, g.embed_w, E, tok, B, tok_bstride, tok_tstride, TB, d.V, w.DXe, E));
    // the hoisted projection's parameter gradients need dVproj (main stream, above)
    XG_TRY(ss.fork());
    if (w.gm == 1) {
        // v2a.bias: the column sums of dVproj cancel almost completely (every video's softmax gradients sum to zero across its
        // frames), so what is left of a sum over bf16-ROUNDED elements is mostly rounding (cosine 0.91 against the fp32 oracle at
        // the full configs[4] size): this one bias gradient keeps its own pass over the fp32 values
        XG_TRY(tn16(sx, w.gm, N, A, R, w.DVPROJ, m16(w, w.DVPROJ), A, w.Venc, m16(w, w.Venc), R, g.v2a_w, R));
        XG_TRY(xgk_colsum3(sx, w.DVPROJ, A, N, A, g.v2a_b, nullptr, nullptr));
    } else {
        XG_TRY(tn16(sx, w.gm, N, A, R, w.DVPROJ, m16(w, w.DVPROJ), A, w.Venc, m16(w, w.Venc), R, g.v2a_w, R, g.v2a_b));
    }
    XG_TRY(ss.chain2_into_aux());             // aux now also covers the second side chain
    // everything but two_spatial_encoder.* is final once the auxiliary stream gets here (it has waited for main above)
    if (ss.grad_event && hipEventRecord(ss.grad_event, sx) != hipSuccess) return XG_EHIP;
    return XG_OK;
    };
    return tail();
}

// heads backward from dlogits (rows,V) in w.LOGITS and dcl (rows,C) in w.DCL -> DH2OUT, head param grads
// Start of a full backward pass: clear the zero block beside whatever the main stream does first (the loss backward and
// the first data-gradient product); decoder_bwd_core joins it.
int zero_backward_block(Streams& ss, Ws& w) {
    w.zeroed = false;
    if (!ss.overlap()) return XG_OK;
    XG_TRY(ss.fork2());
    if (hipMemsetAsync(w.zblock, 0, w.zbytes, ss.aux2) != hipSuccess) return XG_EHIP;
    w.zeroed = true;
    return XG_OK;
}

// xent (fused loss path): w.LOGITS still holds the LOGITS; dlogits = coef (softmax - onehot) is formed here, in place, the
// late steps' rows first on the main stream and the early steps' rows on the auxiliary stream beside the first product.
struct XentBwd { const int64_t* seq; const float* mask; const float* dloss_dev; };
int heads_bwd(Streams& ss, const XgDims& d, const XgParams& p, const XgParams& g, const XgRun& run, Ws& w, int rows,
              bool have_cls, const XentBwd* xent = nullptr) {
    // bf16 mode: the vocabulary products read a bf16 copy of dlogits (215 MB in fp32): the fused loss path's xent kernel writes
    // it in the same pass; every other producer of dlogits is followed by one conversion pass here
    hipStream_t st = ss.main;
    const int B = d.B, R = d.R, TB = d.T * B;
    const float* Hout = w.H2 + (size_t)B * R;
    XG_TRY(zero_backward_block(ss, w));       // (every caller continues with decoder_bwd_core, which joins it)
    if (rows < TB) ZERO(w.DH2OUT + (size_t)rows * R, (size_t)(TB - rows) * R);
    // dH = dlogits * W: the reverse-time loop starts from the LAST step, so the rows of the late steps go first on the
    // main stream and the early steps' rows are produced on the auxiliary stream while the loop is already running.
    const int th = (ss.overlap() && !have_cls && rows == TB && d.T >= 4) ? d.T / 2 : 0;
    const int r0 = th * B;
    ss.dh_split_step = th; ss.dh_mark = -1;
    unsigned short* dl16 = const_cast<unsigned short*>(m16(w, w.LOGITS));
    if (xent) XG_TRY(xgk_xent_bwd(st, w.LOGITS, d.V, xent->seq, xent->mask, nullptr, B, d.T, d.V, 1, w.LSE, w.sums,
                                  xent->dloss_dev, 1.0f, r0, rows - r0, dl16));
    else XG_TRY(cvt16(st, w, w.LOGITS, (size_t)rows * d.V));
    XG_TRY(ss.fork());                        // dlogits of the late rows is final
    if (xent && r0 > 0) XG_TRY(xgk_xent_bwd(ss.aux, w.LOGITS, d.V, xent->seq, xent->mask, nullptr, B, d.T, d.V, 1, w.LSE,
                                            w.sums, xent->dloss_dev, 1.0f, 0, r0, dl16));
    XG_TRY(cvt16(st, w, Hout, (size_t)rows * R));         // (the rollouts' forward does not mirror H2; 11 MB)
    const unsigned short* hout16 = m16(w, Hout);
    XG_TRY(nn16(st, w.gm, rows - r0, R, d.V, w.LOGITS + (size_t)r0 * d.V, m16(w, w.LOGITS + (size_t)r0 * d.V), d.V, p.logit_w,
                w16(w, W16_LOGIT), R, w.DH2OUT + (size_t)r0 * R, R, false));
    // everything below runs BESIDE the reverse-time loop: behind the product above (which the loop waits for and which
    // therefore gets the whole chip), and as background products (XGK_GEMM_BG: half of every CU stays free for the loop)
    const int bgm = w.gm | (ss.overlap() && d.K <= 48 ? XGK_GEMM_BG : 0);   // (the attention backward is a half-CU kernel up to 48 frames)
    XG_TRY(ss.fork());
    if (th > 0) {
        XG_TRY(nn16(ss.aux, bgm, r0, R, d.V, w.LOGITS, m16(w, w.LOGITS), d.V, p.logit_w, w16(w, W16_LOGIT), R, w.DH2OUT, R, false));
        ss.dh_mark = ss.mark();
        if (ss.dh_mark == -2) return XG_EHIP;
    }
    // dW_logit / db: parameter gradients, under the loop as well
    auto dwl = [=, &ss, &w, &g]() -> int {
        XG_TRY(tn16(ss.aux, bgm, rows, d.V, R, w.LOGITS, m16(w, w.LOGITS), d.V, Hout, hout16, R, g.logit_w, R, g.logit_b));
        // XgRun.grad_event_head: the vocabulary head's gradients are final, long before anything else, AND logit.* is not read
        // any more in this backward (the product above was its last reader) -- a caller may all-reduce those gradients and
        // even update logit.* from here on
        if (ss.grad_event_head) {
            XG_TRY(ss.fork());
            if (hipEventRecord(ss.grad_event_head, ss.aux) != hipSuccess) return XG_EHIP;
        }
        return XG_OK;
    };
    XG_TRY(dwl());
    if (have_cls) {
        XG_TRY(gemm_tn_cs(st, w.gm, rows, d.C, d.H, w.DCL, d.C, w.HC, d.H, g.cls3_w, d.H, g.cls3_b));
        XG_TRY(gemm_nn(st, w.gm, rows, d.H, d.C, w.DCL, d.C, p.cls3_w, d.H, w.DHC, d.H, false));
        XG_TRY(xgk_relu_drop_bwd(st, w.DHC, w.HC, (int64_t)rows * d.H, xg_make_drop(&run, XG_SITE_CLS, 0)));
        XG_TRY(gemm_tn_cs(st, w.gm, rows, d.H, R, w.DHC, d.H, Hout, R, g.cls0_w, R, g.cls0_b));
        XG_TRY(gemm_nn(st, w.gm, rows, R, d.H, w.DHC, d.H, p.cls0_w, R, w.DH2OUT, R, true));
    }
    return XG_OK;
}

__global__ void losses_kernel(const float* s, float wc, float* out) {
    // s[0..1] = xe sums, s[2..3] = cls sums
    const float lx = s[0] / s[1];
    const float lc = s[3] > 0.f ? s[2] / s[3] : 0.f;
    out[0] = lx + wc * lc; out[1] = lx; out[2] = lc;
}

int check(const XgDims* d, const void* ws, size_t ws_bytes, Ws* w) {
    if (!dims_ok(d) || !ws) return XG_EINVAL;
    *w = carve(*d, const_cast<void*>(ws));
    if (ws_bytes < w->bytes) {
        // a workspace sized by xg_workspace_bytes_mode for an arithmetic mode without bf16 mirrors: no mirror region (the bf16
        // mode then converts every operand on the fly, as it does wherever a mirror is not valid)
        if (ws_bytes < w->core_bytes) return XG_EWORKSPACE;
        w->sh16 = nullptr;
        w->bytes = w->core_bytes;
    }
    if ((uintptr_t)ws % 256 != 0) return XG_EINVAL;
    return XG_OK;
}

}  // namespace

// ================================================================== C ABI
static const char* kParamNames[] = {
    "two_spatial_encoder.visual_emb_rgb.0.weight", "two_spatial_encoder.visual_emb_rgb.0.bias",
    "two_spatial_encoder.visual_emb_rgb.1.weight", "two_spatial_encoder.visual_emb_rgb.1.bias",
    "two_spatial_encoder.visual_emb_opfl.0.weight", "two_spatial_encoder.visual_emb_opfl.0.bias",
    "two_spatial_encoder.visual_emb_opfl.1.weight", "two_spatial_encoder.visual_emb_opfl.1.bias",
    "two_spatial_encoder.lstmcell_rgb.weight_ih", "two_spatial_encoder.lstmcell_rgb.weight_hh",
    "two_spatial_encoder.lstmcell_rgb.bias_ih", "two_spatial_encoder.lstmcell_rgb.bias_hh",
    "two_spatial_encoder.lstmcell_opfl.weight_ih", "two_spatial_encoder.lstmcell_opfl.weight_hh",
    "two_spatial_encoder.lstmcell_opfl.bias_ih", "two_spatial_encoder.lstmcell_opfl.bias_hh",
    "two_spatial_encoder.gate_rgb.gate.0.weight", "two_spatial_encoder.gate_rgb.gate.0.bias",
    "two_spatial_encoder.gate_opfl.gate.0.weight", "two_spatial_encoder.gate_opfl.gate.0.bias",
    "two_spatial_encoder.fusion.late_fusion.0.weight", "two_spatial_encoder.fusion.late_fusion.0.bias",
    "img_embed_h_1.weight", "img_embed_h_1.bias", "img_embed_c_1.weight", "img_embed_c_1.bias",
    "img_embed_h_2.weight", "img_embed_h_2.bias", "img_embed_c_2.weight", "img_embed_c_2.bias",
    "lstmcore.gate.gate.0.weight", "lstmcore.gate.gate.0.bias",
    "lstmcore.lstm_1.i2h.weight", "lstmcore.lstm_1.i2h.bias", "lstmcore.lstm_1.a2h.weight", "lstmcore.lstm_1.a2h.bias",
    "lstmcore.lstm_1.h2h.weight", "lstmcore.lstm_1.h2h.bias",
    "lstmcore.lstm_2.i2h.weight", "lstmcore.lstm_2.i2h.bias", "lstmcore.lstm_2.a2h.weight", "lstmcore.lstm_2.a2h.bias",
    "lstmcore.lstm_2.h2h.weight", "lstmcore.lstm_2.h2h.bias",
    "lstmcore.v2a.weight", "lstmcore.v2a.bias", "lstmcore.h2a.weight", "lstmcore.h2a.bias",
    "lstmcore.a2w.weight", "lstmcore.a2w.bias",
    "embed.weight", "logit.weight", "logit.bias",
    "classifer.0.weight", "classifer.0.bias", "classifer.3.weight", "classifer.3.bias"};
static_assert(sizeof(kParamNames) / sizeof(kParamNames[0]) == sizeof(XgParams) / sizeof(float*),
              "XgParams field count must match the state_dict name table");

extern "C" int xg_version(void) { return XG_VERSION; }
extern "C" int xg_abi_check(int version, size_t sz_dims, size_t sz_params, size_t sz_bn, size_t sz_batch, size_t sz_run) {
    return (version == XG_VERSION && sz_dims == sizeof(XgDims) && sz_params == sizeof(XgParams) && sz_bn == sizeof(XgBnState) &&
            sz_batch == sizeof(XgBatch) && sz_run == sizeof(XgRun)) ? XG_OK : XG_EINVAL;
}
extern "C" const char* xg_strerror(int code) {
    switch (code) {
        case XG_OK: return "ok";
        case XG_EINVAL: return "invalid argument (dimension, null or misaligned pointer)";
        case XG_EARCH: return "device is not gfx950";
        case XG_EHIP: return "HIP runtime / kernel launch error";
        case XG_EWORKSPACE: return "workspace too small (see xg_workspace_bytes)";
        default: return "unknown error";
    }
}
extern "C" int xg_param_count(void) { return (int)(sizeof(kParamNames) / sizeof(kParamNames[0])); }
extern "C" const char* xg_param_name(int i) { return (i >= 0 && i < xg_param_count()) ? kParamNames[i] : nullptr; }
extern "C" int xg_param_numel(const XgDims* d, int i, int64_t* numel) {
    if (!dims_ok(d) || !numel || i < 0 || i >= xg_param_count()) return XG_EINVAL;
    const int64_t R = d->R, A = d->A, E = d->E, V = d->V, C = d->C, H = d->H;
    const int64_t n[] = {R * d->F1, R, R, R, R * d->F2, R, R, R,
                         4 * R * R, 4 * R * R, 4 * R, 4 * R, 4 * R * R, 4 * R * R, 4 * R, 4 * R,
                         R * R, R, R * R, R, R * 2 * R, R,
                         R * R, R, R * R, R, R * R, R, R * R, R,
                         R * E, R,
                         4 * R * E, 4 * R, 4 * R * R, 4 * R, 4 * R * R, 4 * R,
                         4 * R * R, 4 * R, 4 * R * R, 4 * R, 4 * R * R, 4 * R,
                         A * R, A, A * 2 * R, A, A, 1,
                         V * E, V * R, V, H * R, H, C * H, C};
    static_assert(sizeof(n) / sizeof(n[0]) == sizeof(XgParams) / sizeof(float*), "numel table size");
    *numel = n[i];
    return XG_OK;
}
#ifdef XG_DIAG
// diag library only: reads AND clears the dataflow step kernel's time-out flag (a bounded spin that gave up: the step's results are
// wrong); host-synchronous.  tools/dstep_check.py fails on a non-zero value.
extern "C" int xg_debug_dstep_err(void* stream, const XgDims* d, void* ws, size_t ws_bytes, int* out) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!out) return XG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int32_t* word = w.dsync + xgk_dstep_err_word();
    if (hipMemcpyAsync(out, word, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return XG_EHIP;
    if (hipMemsetAsync(word, 0, sizeof(int), st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return XG_EHIP;
    return XG_OK;
}
#endif
extern "C" size_t xg_workspace_bytes_mode(const XgDims* d, int gemm_mode) {
    if (!dims_ok(d)) return 0;
    const Ws w = carve(*d, nullptr);
    return gemm_mode == 1 ? w.bytes : w.core_bytes;
}
extern "C" size_t xg_workspace_bytes(const XgDims* d) {
    if (!dims_ok(d)) return 0;
    return carve(*d, nullptr).bytes;
}

extern "C" int xg_workspace_init(void* stream, void* ws, size_t ws_bytes) {
    if (!ws || (uintptr_t)ws % 256 != 0) return XG_EINVAL;
    return hipMemsetAsync(ws, 0, ws_bytes, (hipStream_t)stream) == hipSuccess ? XG_OK : XG_EHIP;
}

extern "C" int xg_encoder_fwd(void* stream, const XgDims* d, const XgParams* p, const XgBnState* bn, const XgBatch* x,
                              const XgRun* run, void* ws, size_t ws_bytes, float* V) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !x || !run || !V || !x->feats_rgb || !x->feats_opfl || !x->feat_mask) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    attach_packed(w, *d, run);
    hipStream_t st = (hipStream_t)stream;
    Streams es(st, run);
    XG_TRY(encoder_fwd(st, *d, *p, bn, *x, *run, w, &es));
    if (hipMemcpyAsync(V, w.Venc, sizeof(float) * (size_t)d->B * d->K * d->R, hipMemcpyDeviceToDevice, st) != hipSuccess) return XG_EHIP;
    return XG_OK;
}
extern "C" int xg_encoder_bwd(void* stream, const XgDims* d, const XgParams* p, const XgParams* g, const XgBatch* x,
                              const XgRun* run, void* ws, size_t ws_bytes, const float* dV) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !g || !x || !run || !dV) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    attach_packed(w, *d, run);
    Streams ss((hipStream_t)stream, run);
    XG_TRY(encoder_bwd(ss, *d, *p, *g, *x, *run, w, dV));
    return ss.join();
}
extern "C" int xg_init_hidden(void* stream, const XgDims* d, const XgParams* p, const float* V, const float* feat_mask,
                              void* ws, size_t ws_bytes, float* state) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !V || !feat_mask || !state) return XG_EINVAL;
    const size_t BR = (size_t)d->B * d->R;
    return init_hidden((hipStream_t)stream, *d, *p, V, feat_mask, w, state, state + BR, state + 2 * BR, state + 3 * BR);
}
extern "C" int xg_aux_destroy(void* aux) {
    XgAux* a = static_cast<XgAux*>(aux);
    if (!a) return XG_OK;
    if (a->magic != XG_AUX_MAGIC) return XG_EINVAL;
    for (int i = 0; i < XG_NEV; ++i) if (a->ev[i]) (void)hipEventDestroy(a->ev[i]);
    if (a->s) (void)hipStreamDestroy(a->s);
    if (a->s2) (void)hipStreamDestroy(a->s2);
    a->magic = 0;
    delete a;
    return XG_OK;
}
extern "C" int xg_aux_create(void** aux) {
    if (!aux) return XG_EINVAL;
    *aux = nullptr;
    XgAux* a = new (std::nothrow) XgAux();
    if (!a) return XG_EHIP;
    a->magic = XG_AUX_MAGIC;
    for (int i = 0; i < XG_NEV; ++i) a->ev[i] = nullptr;
    bool ok = hipGetDevice(&a->device) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&a->s, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&a->s2, hipStreamNonBlocking) == hipSuccess;
    // The ring / mark events only order launches of THIS device's streams against each other: kernels publish their results at
    // agent scope when they end, so the event's own system-scope fence (an L2 write-back + invalidate in front of whatever the
    // waiting stream runs next) buys nothing here and is switched off (round 6: 5.45 -> 5.40 ms per iteration, tools/r6/xe_ab.sh;
    // XG_AUX_EVFLAGS=<flags> of the diag build overrides).  Nothing the host reads is published through these events.
    unsigned evflags = hipEventDisableTiming | hipEventDisableSystemFence;
    if (const char* e = xg_diag_env("XG_AUX_EVFLAGS")) evflags = (unsigned)strtoul(e, nullptr, 0);
    for (int i = 0; ok && i < XG_NEV; ++i) ok = hipEventCreateWithFlags(&a->ev[i], evflags) == hipSuccess;
    if (!ok) { xg_aux_destroy(a); return XG_EHIP; }
    *aux = a;
    return XG_OK;
}
extern "C" int xg_vproj(void* stream, const XgDims* d, const XgParams* p, const float* V, float* vproj, const XgRun* run) {
    if (!dims_ok(d) || !p || !V || !vproj) return XG_EINVAL;
    const int mode = run && (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;     // same arithmetic as the rollouts' own v2a(V)
    return xgk_linear((hipStream_t)stream, mode, d->B * d->K, d->A, d->R, V, d->R, p->v2a_w, p->v2a_b, vproj, d->A);
}

extern "C" int xg_step_fwd(void* stream, const XgDims* d, const XgParams* p, const int64_t* tokens, const float* xt_mask,
                           const float* V, const float* vproj, const float* pos_feats, const XgRun* run, int step,
                           void* ws, size_t ws_bytes, float* state, float* logp, float* alpha) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !tokens || !V || !vproj || !pos_feats || !run || !state) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, R = d->R, E = d->E;
    const size_t BR = (size_t)B * R;
    if ((uintptr_t)state % 16) return XG_EINVAL;
    attach_packed(w, *d, run);
    StepIO s{};
    s.pos = pos_feats; s.gp = w.GP; s.posg = w.POSG; s.pre1 = nullptr; s.mask = xt_mask; s.ldm = 1;
    s.h1o = state; s.c1o = state + BR; s.h2o = state + 2 * BR; s.c2o = state + 3 * BR;
    s.P = w.P; s.alpha = alpha ? alpha : w.ALPHA; s.af = w.AF; s.g1 = nullptr; s.g2 = nullptr; s.t = step;
    if (run->save) {
        // a following xg_step_bwd needs the OLD state (the step overwrites it in place), the activated gates and alpha
        if (hipMemcpyAsync(w.H1, state, sizeof(float) * BR, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(w.C1, state + BR, sizeof(float) * BR, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(w.H2, state + 2 * BR, sizeof(float) * BR, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(w.C2, state + 3 * BR, sizeof(float) * BR, hipMemcpyDeviceToDevice, st) != hipSuccess) return XG_EHIP;
        s.g1 = w.G1; s.g2 = w.G2; s.alpha = w.ALPHA;
    }
    if (step_packed(w, *d)) {
        // packed weights: the embedding rows are gathered inside the products and the state is updated in place
        s.xt = nullptr; s.tok = tokens;
        s.h1 = state; s.c1 = state + BR; s.h2 = state + 2 * BR; s.c2 = state + 3 * BR;
    } else {
        // the step reads the OLD state while cell 1 already writes the new h1: work from a copy (same launch as the embedding rows)
        XG_TRY(xgk_step_prep(st, p->embed_w, E, tokens, d->V, w.Xe, B, state, w.state_tmp, (int64_t)(4 * BR)));
        s.xt = w.Xe;
        s.h1 = w.state_tmp; s.c1 = w.state_tmp + BR; s.h2 = w.state_tmp + 2 * BR; s.c2 = w.state_tmp + 3 * BR;
    }
    XG_TRY(core_step(st, *d, *p, *run, w, V, vproj, s));
    if (run->save && alpha &&
        hipMemcpyAsync(alpha, w.ALPHA, sizeof(float) * (size_t)B * d->K, hipMemcpyDeviceToDevice, st) != hipSuccess) return XG_EHIP;
    if (logp) {
        XG_TRY(xgk_linear(st, w.gm, B, d->V, R, state + 2 * BR, R, p->logit_w, p->logit_b, w.LOGITS, d->V));
        XG_TRY(xgk_log_softmax(st, w.LOGITS, d->V, logp, d->V, B, d->V, 1, 1, false));
    }
    return XG_OK;
}

// Backward of ONE decoder step (LSTMCore_two_layer_gate.forward, caption_src/sub_modules.py:671-687 with the cell :750-770
// and the gate :42-47), the counterpart of xg_step_fwd run with XgRun.save = 1 on the same workspace.  Plain launches
// (pointwise kernels + GEMMs): this is the single-step building block, not the benchmarked path -- whole sequences go
// through xg_backward_xe / xg_rollout_bwd, which batch the weight gradients over all steps.
extern "C" int xg_step_bwd(void* stream, const XgDims* d, const XgParams* p, const XgParams* g, const int64_t* tokens,
                           const float* xt_mask, const float* V, const float* vproj, const float* pos_feats, const XgRun* run,
                           int step, void* ws, size_t ws_bytes, const float* state_new, const float* dstate_new,
                           float* dstate, float* dV, float* dvproj, float* dpos) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !g || !tokens || !V || !vproj || !pos_feats || !run || !state_new || !dstate_new || !dstate) return XG_EINVAL;
    w.gm = 0;                                   // single-step products are small: exact fp32
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, K = d->K, R = d->R, A = d->A, E = d->E;
    const size_t BR = (size_t)B * R;
    const float* h1n = state_new;               // new state (after the step)
    const float* c1n = state_new + BR;
    const float* c2n = state_new + 3 * BR;
    const float *h1o = w.H1, *c1o = w.C1, *h2o = w.H2, *c2o = w.C2;      // old state saved by xg_step_fwd
    float *dh1 = dstate, *dc1 = dstate + BR, *dh2 = dstate + 2 * BR, *dc2 = dstate + 3 * BR;
    float* dh1n = w.dst[0][0];                  // total gradient wrt h1' = incoming + what cell 2 sends down
    float* daf = w.DAF;
    XG_TRY(xgk_embed_gather(st, p->embed_w, E, tokens, B, 1, 0, B, d->V, w.Xe, E));
    // ---- cell 2                                                                                          :684
    LstmBwdArgs c2{};
    c2.gates = w.G2; c2.ldg = 4 * R; c2.c_prev = c2o; c2.ldcp = R; c2.c_out = c2n; c2.ldco = R;
    c2.mask = xt_mask; c2.ldm = 1;
    c2.dh_out = dstate_new + 2 * BR; c2.lddh = R; c2.dh_add = nullptr; c2.dc_out = dstate_new + 3 * BR; c2.lddc = R;
    c2.ds = w.DS2; c2.ldds = 4 * R; c2.dc_prev = dc2; c2.lddcp = R; c2.dh_prev = dh2; c2.lddhp = R;
    c2.B = B; c2.R = R; c2.order = XG_ORDER_IFOG; c2.mask_mode = XG_MASK_HOLD; c2.drop = xg_make_drop(run, XG_SITE_L2, step);
    XG_TRY(xgk_lstm_bwd(st, c2));
    if (hipMemcpyAsync(dh1n, dstate_new, sizeof(float) * BR, hipMemcpyDeviceToDevice, st) != hipSuccess) return XG_EHIP;
    XG_TRY(gemm_nn(st, 0, B, R, 4 * R, w.DS2, 4 * R, p->l2_i2h_w, R, dh1n, R, true));
    XG_TRY(gemm_nn(st, 0, B, R, 4 * R, w.DS2, 4 * R, p->l2_a2h_w, R, daf, R, false));
    XG_TRY(gemm_nn(st, 0, B, R, 4 * R, w.DS2, 4 * R, p->l2_h2h_w, R, dh2, R, true));
    XG_TRY(gemm_tn(st, 0, B, 4 * R, R, w.DS2, 4 * R, h1n, R, g->l2_i2h_w, R));
    XG_TRY(gemm_tn(st, 0, B, 4 * R, R, w.DS2, 4 * R, w.AF, R, g->l2_a2h_w, R));
    XG_TRY(gemm_tn(st, 0, B, 4 * R, R, w.DS2, 4 * R, h2o, R, g->l2_h2h_w, R));
    XG_TRY(xgk_colsum3(st, w.DS2, 4 * R, B, 4 * R, g->l2_i2h_b, g->l2_a2h_b, g->l2_h2h_b));
    // ---- attention                                                                                       :677-680
    XG_TRY(xgk_attn_bwd(st, daf, R, w.P, vproj, V, p->a2w_w, w.ALPHA, w.DE, w.DP, B, K, R, A));
    XG_TRY(xgk_attn_bwd_post(st, w.P, vproj, p->a2w_w, w.DE, w.DVPROJ, g->a2w_w, 1, B, K, A));
    if (dvproj) XG_TRY(xgk_axpy(st, dvproj, w.DVPROJ, 1.0f, (int64_t)B * K * A));
    if (dV) XG_TRY(xgk_attn_dV(st, w.ALPHA, daf, R, (int64_t)BR, dV, 1, B, K, R, true));
    XG_TRY(gemm_nn(st, 0, B, R, A, w.DP, A, p->h2a_w + R, 2 * R, dh2, R, true));
    XG_TRY(gemm_tn(st, 0, B, A, R, w.DP, A, h1o, R, g->h2a_w, 2 * R));
    XG_TRY(gemm_tn(st, 0, B, A, R, w.DP, A, h2o, R, g->h2a_w + R, 2 * R));
    XG_TRY(xgk_colsum(st, w.DP, A, B, A, g->h2a_b));
    // ---- cell 1                                                                                          :683
    LstmBwdArgs c1{};
    c1.gates = w.G1; c1.ldg = 4 * R; c1.c_prev = c1o; c1.ldcp = R; c1.c_out = c1n; c1.ldco = R;
    c1.mask = xt_mask; c1.ldm = 1;
    c1.dh_out = dh1n; c1.lddh = R; c1.dh_add = nullptr; c1.dc_out = dstate_new + BR; c1.lddc = R;
    c1.ds = w.DS1; c1.ldds = 4 * R; c1.dc_prev = dc1; c1.lddcp = R; c1.dh_prev = dh1; c1.lddhp = R;
    c1.B = B; c1.R = R; c1.order = XG_ORDER_IFOG; c1.mask_mode = XG_MASK_HOLD; c1.drop = xg_make_drop(run, XG_SITE_L1, step);
    XG_TRY(xgk_lstm_bwd(st, c1));
    XG_TRY(gemm_nn(st, 0, B, R, 4 * R, w.DS1, 4 * R, p->l1_h2h_w, R, dh1, R, true));
    XG_TRY(gemm_nn(st, 0, B, R, A, w.DP, A, p->h2a_w, 2 * R, dh1, R, true));          // the attention query read the old h1
    XG_TRY(gemm_tn(st, 0, B, 4 * R, R, w.DS1, 4 * R, h1o, R, g->l1_h2h_w, R));
    XG_TRY(gemm_tn(st, 0, B, 4 * R, E, w.DS1, 4 * R, w.Xe, E, g->l1_i2h_w, E));
    XG_TRY(gemm_tn(st, 0, B, 4 * R, R, w.DS1, 4 * R, w.POSG, R, g->l1_a2h_w, R));
    XG_TRY(xgk_colsum3(st, w.DS1, 4 * R, B, 4 * R, g->l1_i2h_b, g->l1_a2h_b, g->l1_h2h_b));
    // ---- POS gate and the token side                                                                      :682, SAModel.py:105
    XG_TRY(gemm_nn(st, 0, B, R, 4 * R, w.DS1, 4 * R, p->l1_a2h_w, R, w.DPOSG, R, false));
    XG_TRY(gemm_nn(st, 0, B, E, 4 * R, w.DS1, 4 * R, p->l1_i2h_w, E, w.DXe, E, false));
    XG_TRY(xgk_gate_bwd(st, w.DPOSG, R, w.GP, R, pos_feats, R, 0, w.DGP, R, dpos, R, true, B, R, xg_make_drop(run, XG_SITE_DGATE, step)));
    XG_TRY(gemm_tn(st, 0, B, R, E, w.DGP, R, w.Xe, E, g->dgate_w, E));
    XG_TRY(xgk_colsum(st, w.DGP, R, B, R, g->dgate_b));
    XG_TRY(gemm_nn(st, 0, B, E, R, w.DGP, R, p->dgate_w, E, w.DXe, E, true));
    XG_TRY(xgk_embed_scatter_add(st, g->embed_w, E, tokens, B, 1, 0, B, d->V, w.DXe, E));
    return XG_OK;
}

extern "C" int xg_forward_xe(void* stream, const XgDims* d, const XgParams* p, const XgBnState* bn, const XgBatch* x,
                             const XgRun* run, void* ws, size_t ws_bytes, float* logp, float* cat_logp) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !x || !run || !logp || !x->seq || !x->seq_mask || !x->pos_feats) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    attach_packed(w, *d, run);
    hipStream_t st = (hipStream_t)stream;
    const int TB = d->T * d->B;
    Streams ss(st, run);
    int rows_done = 0;
    XG_TRY(ss.fork());
    XG_TRY(decoder_tokens_xe(ss.aux, *d, *p, *x, *run, w));
    XG_TRY(encoder_fwd(st, *d, *p, bn, *x, *run, w, &ss));
    XG_TRY(decoder_fwd_xe(ss, *d, *p, *x, *run, w, &rows_done));
    XG_TRY(heads_fwd_logits(ss, *d, *p, *run, w, TB, rows_done));
    XG_TRY(xgk_log_softmax(st, w.LOGITS, d->V, logp, d->V, TB, d->V, d->B, d->T, true));
    if (cat_logp) XG_TRY(xgk_log_softmax(st, w.CL, d->C, cat_logp, d->C, TB, d->C, d->B, d->T, true));
    return XG_OK;
}

extern "C" int xg_backward_xe(void* stream, const XgDims* d, const XgParams* p, const XgParams* g, const XgBatch* x,
                              const XgRun* run, void* ws, size_t ws_bytes, const float* dlogp, const float* dcat_logp) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !g || !x || !run || !x->seq || !x->seq_mask) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    attach_packed(w, *d, run);
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, T = d->T, TB = T * B;
    // log-softmax backward needs logp = logits - lse: recompute lse rows from the saved logits (in place)
    if (dlogp) {
        XG_TRY(xgk_log_softmax(st, w.LOGITS, d->V, w.LOGITS, d->V, TB, d->V, 1, 1, false));
        XG_TRY(xgk_log_softmax_bwd(st, dlogp, w.LOGITS, d->V, w.LOGITS, d->V, TB, d->V, B, T, 2));
    } else {
        ZERO(w.LOGITS, (size_t)TB * d->V);
    }
    if (dcat_logp) {
        XG_TRY(xgk_log_softmax(st, w.CL, d->C, w.CL, d->C, TB, d->C, 1, 1, false));
        XG_TRY(xgk_log_softmax_bwd(st, dcat_logp, w.CL, d->C, w.DCL, d->C, TB, d->C, B, T, 2));
    }
    Streams ss(st, run);
    XG_TRY(heads_bwd(ss, *d, *p, *g, *run, w, TB, dcat_logp != nullptr));
    XG_TRY(decoder_bwd_core(ss, *d, *p, *g, *x, *run, w, x->seq_mask, T, 1, x->seq, T, 1));
    XG_TRY(encoder_bwd(ss, *d, *p, *g, *x, *run, w, w.DV));
    return ss.join();
}

extern "C" int xg_forward_ss(void* stream, const XgDims* d, const XgParams* p, const XgBnState* bn, const XgBatch* x,
                             const XgRun* run, float ss_prob, const float* u_sel, const float* u_tok, void* ws,
                             size_t ws_bytes, float* logp, float* cat_logp) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !x || !run || !logp || !x->seq || !x->seq_mask || !x->pos_feats) return XG_EINVAL;
    const bool ss = run->train && ss_prob > 0.f;
    if (ss && (!u_sel || !u_tok)) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    attach_packed(w, *d, run);
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, R = d->R, E = d->E, A = d->A, T = d->T, TB = T * B;
    const size_t BR = (size_t)B * R;
    Streams es(st, run);
    XG_TRY(encoder_fwd(st, *d, *p, bn, *x, *run, w, &es));
    XG_TRY(init_and_vproj(es, *d, *p, x->feat_mask, w));
    XG_TRY(zero_dsync(st, w));
    int64_t* sampled = reinterpret_cast<int64_t*>(w.DXe);       // scratch (free until the backward pass)
    for (int t = 0; t < T; ++t) {
        int64_t* tok = w.TOK + (size_t)t * B;
        const bool draw = ss && t >= 1;
        if (draw)     // draw for every row from the previous step's distribution, keep it where the coin says so (:93-98)
            XG_TRY(xgk_choose(st, w.LOGITS + (size_t)(t - 1) * B * d->V, B, d->V, XG_ROLLOUT_SAMPLE, u_tok + (size_t)t * B,
                              nullptr, 0, 1.0f, sampled, w.TOKLP));
        XG_TRY(xgk_ss_select(st, x->seq, T, t, B, draw ? u_sel + (size_t)t * B : nullptr, ss_prob, draw ? sampled : nullptr, tok));
        float* xt = w.Xe + (size_t)t * B * E;
        XG_TRY(xgk_embed_gather(st, p->embed_w, E, tok, B, 1, 0, B, d->V, xt, E));
        StepIO s{};
        s.xt = xt; s.pos = x->pos_feats; s.gp = w.GP + t * BR; s.posg = w.POSG + t * BR; s.pre1 = nullptr;
        s.mask = x->seq_mask + t; s.ldm = T;
        s.h1 = w.H1 + t * BR; s.c1 = w.C1 + t * BR; s.h2 = w.H2 + t * BR; s.c2 = w.C2 + t * BR;
        s.h1o = w.H1 + (t + 1) * BR; s.c1o = w.C1 + (t + 1) * BR; s.h2o = w.H2 + (t + 1) * BR; s.c2o = w.C2 + (t + 1) * BR;
        s.P = w.P + (size_t)t * B * A; s.alpha = w.ALPHA + (size_t)t * B * d->K; s.af = w.AF + t * BR;
        s.g1 = w.G1 + (size_t)t * B * 4 * R; s.g2 = w.G2 + (size_t)t * B * 4 * R; s.t = t;
        XG_TRY(core_step(st, *d, *p, *run, w, w.Venc, w.vproj, s));
        float* lg = w.LOGITS + (size_t)t * B * d->V;
        XG_TRY(xgk_linear(st, w.gm, B, d->V, R, s.h2o, R, p->logit_w, p->logit_b, lg, d->V));
        XG_TRY(xgk_log_softmax(st, lg, d->V, lg, d->V, B, d->V, 1, 1, false));        // LOGITS now holds log-probs (time-major)
    }
    // (T*B,V) time-major log-probs -> (B,T,V); classifier head batched over T as in xg_forward_xe
    XG_TRY(xgk_log_softmax(st, w.LOGITS, d->V, logp, d->V, TB, d->V, B, T, true));    // log_softmax of log-probs = identity
    const float* Hout = w.H2 + BR;
    XG_TRY(xgk_linear(st, w.gm, TB, d->H, R, Hout, R, p->cls0_w, p->cls0_b, w.HC, d->H, true));
    XG_TRY(xgk_gate_fwd(st, w.HC, d->H, nullptr, 0, 0, nullptr, 0, TB, d->H, xg_make_drop(run, XG_SITE_CLS, 0), B, 1 << 30, 1, B));
    XG_TRY(xgk_linear(st, w.gm, TB, d->C, d->H, w.HC, d->H, p->cls3_w, p->cls3_b, w.CL, d->C));
    if (cat_logp) XG_TRY(xgk_log_softmax(st, w.CL, d->C, cat_logp, d->C, TB, d->C, B, T, true));
    return XG_OK;
}

extern "C" int xg_backward_ss(void* stream, const XgDims* d, const XgParams* p, const XgParams* g, const XgBatch* x,
                              const XgRun* run, void* ws, size_t ws_bytes, const float* dlogp, const float* dcat_logp) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !g || !x || !run || !x->seq || !x->seq_mask) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    attach_packed(w, *d, run);
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, T = d->T, TB = T * B;
    if (dlogp) {   // LOGITS already holds the time-major log-probs
        XG_TRY(xgk_log_softmax_bwd(st, dlogp, w.LOGITS, d->V, w.LOGITS, d->V, TB, d->V, B, T, 2));
    } else {
        ZERO(w.LOGITS, (size_t)TB * d->V);
    }
    if (dcat_logp) {
        XG_TRY(xgk_log_softmax(st, w.CL, d->C, w.CL, d->C, TB, d->C, 1, 1, false));
        XG_TRY(xgk_log_softmax_bwd(st, dcat_logp, w.CL, d->C, w.DCL, d->C, TB, d->C, B, T, 2));
    }
    Streams ss(st, run);
    XG_TRY(heads_bwd(ss, *d, *p, *g, *run, w, TB, dcat_logp != nullptr));
    XG_TRY(decoder_bwd_core(ss, *d, *p, *g, *x, *run, w, x->seq_mask, T, 1, w.TOK, 1, B));
    XG_TRY(encoder_bwd(ss, *d, *p, *g, *x, *run, w, w.DV));
    return ss.join();
}

extern "C" int xg_xe_loss_fwd(void* stream, const XgDims* d, const XgParams* p, const XgBnState* bn, const XgBatch* x,
                              const int64_t* cap_classes, const float* class_mask, float weight_class, const XgRun* run,
                              void* ws, size_t ws_bytes, float* losses) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !x || !run || !losses || !x->seq || !x->seq_mask || !x->pos_feats) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    attach_packed(w, *d, run);
    hipStream_t st = (hipStream_t)stream;
    const int TB = d->T * d->B;
    Streams ss(st, run);
    int rows_done = 0;
    ZERO(w.sums, 2);                           // both halves of the cross-entropy add into it
    XG_TRY(ss.fork());
    XG_TRY(decoder_tokens_xe(ss.aux, *d, *p, *x, *run, w));
    XG_TRY(encoder_fwd(st, *d, *p, bn, *x, *run, w, &ss));
    XG_TRY(decoder_fwd_xe(ss, *d, *p, *x, *run, w, &rows_done, true));
    XG_TRY(heads_fwd_logits(ss, *d, *p, *run, w, TB, rows_done));       // (joins the auxiliary stream)
    XG_TRY(xgk_xent_fwd(st, w.LOGITS, d->V, x->seq, x->seq_mask, nullptr, d->B, d->T, d->V, 1, w.LSE, w.sums, rows_done, -1, false));
    if (cap_classes) {
        XG_TRY(xgk_xent_fwd(st, w.CL, d->C, cap_classes, x->seq_mask, class_mask, d->B, d->T, d->C, 0, w.LSEC, w.sums + 2));
    } else {
        ZERO(w.sums + 2, 2);
    }
    hipLaunchKernelGGL(losses_kernel, dim3(1), dim3(1), 0, st, w.sums, cap_classes ? weight_class : 0.f, losses);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
extern "C" int xg_xe_loss_bwd(void* stream, const XgDims* d, const XgParams* p, const XgParams* g, const XgBatch* x,
                              const int64_t* cap_classes, const float* class_mask, float weight_class,
                              const float* dloss_dev, const XgRun* run, void* ws, size_t ws_bytes) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !g || !x || !run || !x->seq || !x->seq_mask) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    attach_packed(w, *d, run);
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, T = d->T, TB = T * B;
    const XentBwd xent{x->seq, x->seq_mask, dloss_dev};
    const bool cls = cap_classes != nullptr && weight_class != 0.f;
    if (cls) {
        if (hipMemcpyAsync(w.DCL, w.CL, sizeof(float) * (size_t)TB * d->C, hipMemcpyDeviceToDevice, st) != hipSuccess) return XG_EHIP;
        XG_TRY(xgk_xent_bwd(st, w.DCL, d->C, cap_classes, x->seq_mask, class_mask, B, T, d->C, 0, w.LSEC, w.sums + 2,
                            dloss_dev, weight_class));
    }
    Streams ss(st, run);
    XG_TRY(heads_bwd(ss, *d, *p, *g, *run, w, TB, cls, &xent));
    XG_TRY(decoder_bwd_core(ss, *d, *p, *g, *x, *run, w, x->seq_mask, T, 1, x->seq, T, 1));
    XG_TRY(encoder_bwd(ss, *d, *p, *g, *x, *run, w, w.DV));
    return ss.join();
}

// One rollout over d.B rows.  split < d.B (SAMPLE mode only): rows [0, split) sample with uniforms (T, split), rows
// [split, B) decode greedily -- the SCST pair (starttrain.py:131 + myutils.py:45) as ONE batch; n_steps then has two entries.
// logits_alt (paired rollout whose sampled half will be compacted for the backward): the raw logits of rows [0, split) go there,
// (T - 1) blocks of `split` rows, instead of into this workspace -- they are two thirds of what a compaction would copy.  Only the
// tile-statistics path writes row subsets: *used_alt tells the caller whether it happened.
// shared (paired rollout over the SAME videos, xg_rollout_pair_videos): `x` holds the d1->B = B / 2 videos once; the encoder,
// v2a(V) and the initial state are computed ONCE, in the workspace `enc` of the un-repeated batch (where the backward of the
// sampled half reads them), and row-repeated into this workspace for the 2m-row decoder loop -- half the encoder work of the
// repeated batch, no repeated inputs, and BatchNorm sees the m-row batch exactly as the reference's two sample() calls do
// (two running-statistics updates with the same batch statistics, SAModel.py:169 called twice: starttrain.py:131, myutils.py:45).
static int rollout_impl(hipStream_t st, const XgDims* d, const XgParams* p, const XgBnState* bn, const XgBatch* x,
                        const XgRun* run, int mode, const float* uniforms, const int64_t* forced, float temperature, Ws& w,
                        int64_t* seq, float* seq_logp, int32_t* n_steps, int split, float* logits_alt = nullptr,
                        bool* used_alt = nullptr, const XgDims* d1 = nullptr, Ws* enc = nullptr) {
    const int B = d->B, R = d->R, E = d->E, A = d->A, T = d->T;
    const size_t BR = (size_t)B * R;
    Streams es(st, run);
    const float* pos_rows = x->pos_feats;
    if (enc) {
        const size_t B1 = d1->B, N1 = B1 * d->K;
        XG_TRY(encoder_fwd(st, *d1, *p, bn, *x, *run, *enc, &es));
        if (run->train && bn && bn->rgb_mean && bn->rgb_var && bn->opfl_mean && bn->opfl_var) {      // the second sample() call's update
            XG_TRY(xgk_bn_running(st, enc->bn_mean[0], enc->bn_var[0], bn->rgb_mean, bn->rgb_var, (int)N1, R, run->bn_momentum));
            XG_TRY(xgk_bn_running(st, enc->bn_mean[1], enc->bn_var[1], bn->opfl_mean, bn->opfl_var, (int)N1, R, run->bn_momentum));
        }
        XG_TRY(init_and_vproj(es, *d1, *p, x->feat_mask, *enc));
        CompactArgs ca{};
        auto twice = [&](float* dst, const float* src, size_t n) {
            ca.e[ca.n++] = CompactEntry{dst, src, (int64_t)n, (int64_t)n, (int64_t)n};
            ca.e[ca.n++] = CompactEntry{dst + n, src, (int64_t)n, (int64_t)n, (int64_t)n};
        };
        twice(w.Venc, enc->Venc, N1 * R); twice(w.vproj, enc->vproj, N1 * A);
        twice(w.H1, enc->H1, B1 * R); twice(w.C1, enc->C1, B1 * R); twice(w.H2, enc->H2, B1 * R); twice(w.C2, enc->C2, B1 * R);
        twice(w.DPOSG, x->pos_feats, B1 * R);          // (a backward-only buffer: this workspace never runs a backward)
        XG_TRY(xgk_compact(st, ca));
        pos_rows = w.DPOSG;
    } else {
        XG_TRY(encoder_fwd(st, *d, *p, bn, *x, *run, w, &es));
        XG_TRY(init_and_vproj(es, *d, *p, x->feat_mask, w));
    }
    XG_TRY(zero_dsync(st, w));
    if (hipMemsetAsync(w.alive, 0, sizeof(int32_t) * 4, st) != hipSuccess) return XG_EHIP;   // alive[i] = running max finishing step
    if (run->prof_event0 && hipEventRecord(static_cast<hipEvent_t>(run->prof_event0), st) != hipSuccess) return XG_EHIP;
    // Rollout steps of <= 128 rows, fp32: the vocabulary product leaves per-tile row statistics and the token choice reads those
    // (xg_heads.hip: vocab_part_kernel / roll_select_kernel) -- 10 MB of logits per step are neither written (greedy rows) nor read
    // back three times.  Rows whose logits are needed afterwards are still stored: sampled rows (the draw re-reads one tile; the
    // SCST backward the whole row), replayed rows, every row of a rollout that keeps its activations.
    static const bool no_fused_select = xg_diag_env("XG_NO_FUSED_SELECT") != nullptr;
    const bool fused_select = !no_fused_select && w.gm == 0 && xgk_vocab_select_ok(B, R, d->V, w.H2, R, p->logit_w);
    const int wr_rows = mode == XG_ROLLOUT_SAMPLE ? split : ((mode == XG_ROLLOUT_REPLAY || run->save) ? B : 0);
    const bool alt = logits_alt != nullptr && fused_select && mode == XG_ROLLOUT_SAMPLE && split < B;
    if (used_alt) *used_alt = alt;
    auto logits_of = [&](int t) { return alt ? logits_alt + (size_t)t * split * d->V : w.LOGITS + (size_t)t * B * d->V; };
    auto step_io = [&](int t) {
        StepIO s{};
        s.xt = w.Xe + (size_t)t * B * E; s.pos = pos_rows; s.gp = w.GP + t * BR; s.posg = w.POSG + t * BR; s.pre1 = nullptr;
        s.mask = w.UNF + (size_t)t * B; s.ldm = 1;
        s.h1 = w.H1 + t * BR; s.c1 = w.C1 + t * BR; s.h2 = w.H2 + t * BR; s.c2 = w.C2 + t * BR;
        s.h1o = w.H1 + (t + 1) * BR; s.c1o = w.C1 + (t + 1) * BR; s.h2o = w.H2 + (t + 1) * BR; s.c2o = w.C2 + (t + 1) * BR;
        s.P = w.P + (size_t)t * B * A; s.alpha = w.ALPHA + (size_t)t * B * d->K; s.af = w.AF + t * BR;
        s.g1 = w.G1 + (size_t)t * B * 4 * R; s.g2 = w.G2 + (size_t)t * B * 4 * R; s.t = t;
        return s;
    };
    // Steps t >= 1 on the packed fp32 path CHOOSE their tokens themselves: the choice is the prologue of the step's first launch
    // (its POS-gate tiles: xg_step.hip SEL, xg_select.h) instead of a launch of its own between the vocabulary product and the step
    // -- four dependent launches per step instead of five.  The last choice of a rollout (no step follows) keeps its launch.
    static const bool no_step_select = xg_diag_env("XG_NO_STEP_SELECT") != nullptr;
    const bool step_select = fused_select && !no_step_select && step_packed(w, *d) && xg_cdiv(d->V, xgk_vocab_tile_width(d->V)) <= 256;
    for (int t = 0; t < T; ++t) {
        int64_t* tok = w.TOK + (size_t)t * B;
        float* unf = w.UNF + (size_t)t * B;
        float* xt = w.Xe + (size_t)t * B * E;
        const float* prev_logits = t >= 1 ? logits_of(t - 1) : nullptr;
        // token choice from the previous step's raw logits + bookkeeping + embedding gather: one launch (:183-215)
        const bool in_step = step_select && t >= 1 && t + 1 < T;
        RollSelectArgs sel{};
        if (in_step)
            sel = xgk_roll_select_args(prev_logits, w.VPART, uniforms ? uniforms + (size_t)t * split : nullptr,
                                       forced ? forced + (t - 1) : nullptr, T - 1, t >= 2 ? unf - B : nullptr, p->embed_w, tok,
                                       w.TOKLP + (size_t)t * B, unf, w.LSE + (size_t)(t - 1) * B, seq, seq_logp, w.alive, xt, temperature,
                                       d->V, E, t, T, mode, split);
        else if (t >= 1 && fused_select)
            XG_TRY(xgk_roll_select(st, B, prev_logits, w.VPART, uniforms ? uniforms + (size_t)t * split : nullptr,
                                   forced ? forced + (t - 1) : nullptr, T - 1, t >= 2 ? unf - B : nullptr, p->embed_w, tok,
                                   w.TOKLP + (size_t)t * B, unf, w.LSE + (size_t)(t - 1) * B, seq, seq_logp, w.alive, xt, temperature,
                                   d->V, E, t, T, mode, split));
        else
            XG_TRY(xgk_rollout_step(st, B, prev_logits,
                                    uniforms ? uniforms + (size_t)t * split : nullptr, (forced && t >= 1) ? forced + (t - 1) : nullptr,
                                    T - 1, t >= 2 ? unf - B : nullptr, p->embed_w, tok, w.TOKLP + (size_t)t * B, unf,
                                    t >= 1 ? w.LSE + (size_t)(t - 1) * B : nullptr, seq, seq_logp, w.alive, xt, temperature, d->V, E,
                                    t, T, mode, split));
        StepIO s = step_io(t);
        if (in_step) { s.sel = &sel; s.xt = nullptr; s.tok = tok; }
        // The reference runs the core once more at t = L and discards what it computes (SAModel.py:182,217: the loop ends before
        // those logits are ever read, and no state is returned): that step feeds neither an output nor a gradient and is not run.
        if (t + 1 == T) break;
        XG_TRY(core_step(st, *d, *p, *run, w, w.Venc, w.vproj, s));
        if (t + 1 < T) {
            if (fused_select)
                XG_TRY(xgk_vocab_part(st, B, R, d->V, s.h2o, R, p->logit_w, p->logit_b, logits_of(t), wr_rows,
                                      w.VPART, temperature > 0.f ? temperature : 1.0f));
            else
                XG_TRY(xgk_linear(st, w.gm, B, d->V, R, s.h2o, R, p->logit_w, p->logit_b, w.LOGITS + (size_t)t * B * d->V, d->V));
        }
    }
    // the steps that chose their own tokens gathered their embedding rows inside the products: the (T, B, E) copy the backward (and a
    // compaction) reads is made here for all of them at once, off the token's path
    if (step_select && T >= 3)
        XG_TRY(xgk_embed_gather(st, p->embed_w, E, w.TOK + B, (T - 2) * B, 1, 0, (T - 2) * B, d->V, w.Xe + (size_t)B * E, E));
    if (run->prof_event1 && hipEventRecord(static_cast<hipEvent_t>(run->prof_event1), st) != hipSuccess) return XG_EHIP;
    XG_TRY(xgk_rollout_finalize(st, w.alive, n_steps, T - 1, split < B ? 2 : 1));
    return XG_OK;
}

extern "C" int xg_rollout(void* stream, const XgDims* d, const XgParams* p, const XgBnState* bn, const XgBatch* x,
                          const XgRun* run, int mode, const float* uniforms, const int64_t* forced, float temperature,
                          void* ws, size_t ws_bytes, int64_t* seq, float* seq_logp, int32_t* n_steps) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !x || !run || !seq || !seq_logp || !n_steps || !x->pos_feats || d->T < 2) return XG_EINVAL;
    if (mode == XG_ROLLOUT_SAMPLE && (!uniforms || !(temperature > 0.f))) return XG_EINVAL;
    if (mode == XG_ROLLOUT_REPLAY && !forced) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    attach_packed(w, *d, run);
    return rollout_impl((hipStream_t)stream, d, p, bn, x, run, mode, uniforms, forced, temperature, w, seq, seq_logp, n_steps, d->B);
}

extern "C" int xg_rollout_pair(void* stream, const XgDims* d2, const XgParams* p, const XgBnState* bn, const XgBatch* x2,
                               const XgRun* run, int n_sample, const float* uniforms, float temperature, void* ws2,
                               size_t ws2_bytes, int64_t* seq, float* seq_logp, int32_t* n_steps) {
    Ws w; XG_TRY(check(d2, ws2, ws2_bytes, &w));
    if (!p || !x2 || !run || !seq || !seq_logp || !n_steps || !x2->pos_feats || d2->T < 2) return XG_EINVAL;
    if (n_sample <= 0 || n_sample >= d2->B || !uniforms || !(temperature > 0.f)) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    attach_packed(w, *d2, run);
    return rollout_impl((hipStream_t)stream, d2, p, bn, x2, run, XG_ROLLOUT_SAMPLE, uniforms, nullptr, temperature, w, seq,
                        seq_logp, n_steps, n_sample);
}

// Everything xg_rollout_bwd reads, for the FIRST d1->B rows of a rollout that ran over d2->B >= d1->B rows: encoder-side
// tensors are row prefixes, decoder-side tensors are per-step blocks (pitch B2 -> B1).
static int compact_impl(hipStream_t st, const XgDims* d2, const Ws& a, const XgDims* d1, Ws& b, bool skip_logits, bool skip_encoder = false) {
    if (d1->B > d2->B || d1->K != d2->K || d1->R != d2->R || d1->A != d2->A || d1->E != d2->E || d1->V != d2->V ||
        d1->T != d2->T || d1->F1 != d2->F1 || d1->F2 != d2->F2 || d1->C != d2->C || d1->H != d2->H) return XG_EINVAL;
    const size_t B1 = d1->B, B2 = d2->B, K = d1->K, R = d1->R, A = d1->A, E = d1->E, V = d1->V, T = d1->T, N1 = B1 * K;
    CompactArgs ca{};
    auto prefix = [&](void* dst, const void* src, size_t nfloats) -> int {
        if (ca.n >= XG_COMPACT_MAX) return XG_EINVAL;
        ca.e[ca.n++] = CompactEntry{static_cast<float*>(dst), static_cast<const float*>(src), (int64_t)nfloats, (int64_t)nfloats, (int64_t)nfloats};
        return XG_OK;
    };
    auto blocks = [&](void* dst, const void* src, size_t width_floats, size_t nblocks) -> int {   // block = B rows x width
        if (ca.n >= XG_COMPACT_MAX) return XG_EINVAL;
        ca.e[ca.n++] = CompactEntry{static_cast<float*>(dst), static_cast<const float*>(src), (int64_t)(B1 * width_floats),
                                    (int64_t)(B2 * width_floats), (int64_t)(B1 * width_floats * nblocks)};
        return XG_OK;
    };
    for (int m = 0; m < 2 && !skip_encoder; ++m) {    // (skip_encoder: the encoder ran in `b` itself, xg_rollout_pair_videos)
        XG_TRY(prefix(b.Z[m], a.Z[m], N1 * R)); XG_TRY(prefix(b.X[m], a.X[m], N1 * R)); XG_TRY(prefix(b.PRE[m], a.PRE[m], N1 * 4 * R));
        XG_TRY(prefix(b.Hs[m], a.Hs[m], N1 * R)); XG_TRY(prefix(b.Cs[m], a.Cs[m], N1 * R)); XG_TRY(prefix(b.G[m], a.G[m], N1 * 4 * R));
        XG_TRY(prefix(b.GG[m], a.GG[m], N1 * R)); XG_TRY(prefix(b.Hprev[m], a.Hprev[m], N1 * R));
        XG_TRY(prefix(b.bn_mean[m], a.bn_mean[m], R)); XG_TRY(prefix(b.bn_var[m], a.bn_var[m], R));
    }
    if (!skip_encoder) {
        XG_TRY(prefix(b.Y, a.Y, N1 * 2 * R)); XG_TRY(prefix(b.Venc, a.Venc, N1 * R)); XG_TRY(prefix(b.vbar, a.vbar, B1 * R));
        XG_TRY(prefix(b.vproj, a.vproj, N1 * A));
    }
    XG_TRY(blocks(b.Xe, a.Xe, E, T)); XG_TRY(blocks(b.GP, a.GP, R, T)); XG_TRY(blocks(b.POSG, a.POSG, R, T));
    XG_TRY(blocks(b.H1, a.H1, R, T + 1)); XG_TRY(blocks(b.C1, a.C1, R, T + 1));
    XG_TRY(blocks(b.H2, a.H2, R, T + 1)); XG_TRY(blocks(b.C2, a.C2, R, T + 1));
    XG_TRY(blocks(b.G1, a.G1, 4 * R, T)); XG_TRY(blocks(b.G2, a.G2, 4 * R, T));
    XG_TRY(blocks(b.P, a.P, A, T)); XG_TRY(blocks(b.ALPHA, a.ALPHA, K, T)); XG_TRY(blocks(b.AF, a.AF, R, T));
    if (!skip_logits) XG_TRY(blocks(b.LOGITS, a.LOGITS, V, T - 1));     // (skipped: the rollout wrote them into `b` itself)
    XG_TRY(blocks(b.LSE, a.LSE, 1, T));
    XG_TRY(blocks(b.TOK, a.TOK, 2, T));                       // int64 = 2 floats wide
    XG_TRY(blocks(b.TOKLP, a.TOKLP, 1, T)); XG_TRY(blocks(b.UNF, a.UNF, 1, T));
    XG_TRY(xgk_compact(st, ca));
    ZERO(b.zeroBR, B1 * R);                                   // the encoder's initial state (read by its backward)
    return XG_OK;
}

extern "C" int xg_rollout_compact(void* stream, const XgDims* d2, const void* ws2, size_t ws2_bytes, const XgDims* d1,
                                  void* ws1, size_t ws1_bytes) {
    Ws a, b;
    XG_TRY(check(d2, const_cast<void*>(ws2), ws2_bytes, &a));
    XG_TRY(check(d1, ws1, ws1_bytes, &b));
    return compact_impl((hipStream_t)stream, d2, a, d1, b, false);
}

// xg_rollout_pair + xg_rollout_compact in one call: the sampled rows' logits -- 154 MB of the 230 MB a compaction copies at
// B = 64, L = 30, V = 20000 -- are written into the compacted workspace by the rollout itself (when its token choice runs over
// tile statistics: <= 128 rows, fp32; otherwise the two steps run as they are).
extern "C" int xg_rollout_pair_compact(void* stream, const XgDims* d2, const XgParams* p, const XgBnState* bn, const XgBatch* x2,
                                       const XgRun* run, int n_sample, const float* uniforms, float temperature, void* ws2,
                                       size_t ws2_bytes, const XgDims* d1, void* ws1, size_t ws1_bytes, int64_t* seq,
                                       float* seq_logp, int32_t* n_steps) {
    Ws w, b;
    XG_TRY(check(d2, ws2, ws2_bytes, &w));
    XG_TRY(check(d1, ws1, ws1_bytes, &b));
    if (!p || !x2 || !run || !seq || !seq_logp || !n_steps || !x2->pos_feats || d2->T < 2) return XG_EINVAL;
    if (n_sample <= 0 || n_sample >= d2->B || n_sample != d1->B || !uniforms || !(temperature > 0.f)) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    attach_packed(w, *d2, run);
    bool used_alt = false;
    XG_TRY(rollout_impl((hipStream_t)stream, d2, p, bn, x2, run, XG_ROLLOUT_SAMPLE, uniforms, nullptr, temperature, w, seq, seq_logp,
                        n_steps, n_sample, b.LOGITS, &used_alt));
    return compact_impl((hipStream_t)stream, d2, w, d1, b, used_alt);
}

extern "C" int xg_rollout_pair_videos(void* stream, const XgDims* d2, const XgParams* p, const XgBnState* bn, const XgBatch* x1,
                                      const XgRun* run, const float* uniforms, float temperature, void* ws2, size_t ws2_bytes,
                                      const XgDims* d1, void* ws1, size_t ws1_bytes, int compact, int64_t* seq, float* seq_logp,
                                      int32_t* n_steps) {
    Ws w, b;
    XG_TRY(check(d2, ws2, ws2_bytes, &w));
    XG_TRY(check(d1, ws1, ws1_bytes, &b));
    if (!p || !x1 || !run || !seq || !seq_logp || !n_steps || !x1->pos_feats || !x1->feats_rgb || !x1->feats_opfl || !x1->feat_mask ||
        d2->T < 2) return XG_EINVAL;
    if (d2->B != 2 * d1->B || !uniforms || !(temperature > 0.f) || ws1 == ws2) return XG_EINVAL;
    if (d1->K != d2->K || d1->R != d2->R || d1->A != d2->A || d1->E != d2->E || d1->V != d2->V || d1->T != d2->T ||
        d1->F1 != d2->F1 || d1->F2 != d2->F2 || d1->C != d2->C || d1->H != d2->H) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    b.gm = w.gm;
    attach_packed(w, *d2, run);
    attach_packed(b, *d1, run);
    bool used_alt = false;
    XG_TRY(rollout_impl((hipStream_t)stream, d2, p, bn, x1, run, XG_ROLLOUT_SAMPLE, uniforms, nullptr, temperature, w, seq, seq_logp,
                        n_steps, d1->B, compact ? b.LOGITS : nullptr, &used_alt, d1, &b));
    if (!compact) return XG_OK;
    return compact_impl((hipStream_t)stream, d2, w, d1, b, used_alt, true);
}

extern "C" int xg_rollout_bwd(void* stream, const XgDims* d, const XgParams* p, const XgParams* g, const XgBatch* x,
                              const XgRun* run, void* ws, size_t ws_bytes, const float* dseq_logp) {
    Ws w; XG_TRY(check(d, ws, ws_bytes, &w));
    if (!p || !g || !x || !run || !dseq_logp || d->T < 2) return XG_EINVAL;
    w.gm = (run->gemm_mode == 1 || run->gemm_mode == 3) ? run->gemm_mode : 0;
    attach_packed(w, *d, run);
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, T = d->T;
    // dlogits of step t-1's output from the token drawn at step t (SAModel.py:195)
    // LOGITS holds raw logits, LSE their log-sum-exp; every step in one launch
    XG_TRY(xgk_rollout_dlogits_lse(st, w.LOGITS, w.LSE, w.TOK + B, dseq_logp, T - 1, B, d->V, T - 1));
    {   // bf16 mode: this workspace may be the compacted half of a paired rollout (xg_rollout_compact copies no mirrors): the
        // encoder-side operands the backward reads through their bf16 mirrors are converted again
        const size_t NR = (size_t)B * d->K * d->R;
        for (int m = 0; m < 2; ++m) { XG_TRY(cvt16(st, w, w.X[m], NR)); XG_TRY(cvt16(st, w, w.Hs[m], NR)); }
        XG_TRY(cvt16(st, w, w.Y, 2 * NR)); XG_TRY(cvt16(st, w, w.Venc, NR));
    }
    Streams ss(st, run);
    // the rollout ran T - 1 core steps (the reference's last one is dead, rollout_impl): the reverse-time pass covers those.
    // All per-step buffers are time-major, so the first T - 1 blocks of the T-step workspace are the (T - 1)-step problem.
    // (What that relies on: LOGITS / H2 / the saved gates, states, p, alpha, af, xt and the gate values are (T, B, .) blocks with
    //  t outermost -- carve_workspace -- so rows [0, (T - 1) B) of each are exactly what a (T - 1)-step workspace would hold, and
    //  heads_bwd's split of dH into an early and a late half plus its background product see rows == dT.T * B as in the
    //  teacher-forced backward: the same code path, covered by the SCST gradient tests at configs[2]'s full size.)
    XgDims dT = *d;
    dT.T = T - 1;
    XG_TRY(heads_bwd(ss, dT, *p, *g, *run, w, (T - 1) * B, false));
    XG_TRY(decoder_bwd_core(ss, dT, *p, *g, *x, *run, w, w.UNF, 1, B, w.TOK, 1, B));
    XG_TRY(encoder_bwd(ss, *d, *p, *g, *x, *run, w, w.DV));
    return ss.join();
}
