// Skinny (M <= a few hundred rows) fp32 MFMA GEMM for the recurrent steps of the decoder / encoder.
//
// The per-timestep products are (B x K) x (K x N) with B = 128: too few output tiles to fill 256 CUs
// with an ordinary tiled GEMM, and each 32x32 MFMA tile needs K/2 dependent 64-cycle MFMAs.  So the
// parallelism comes from K: one workgroup = one 32x32 output tile, its 8 waves split the reduction
// dimension 8-way.  Each wave streams ITS OWN 32-deep chunks of the A / W rows with fully coalesced
// 128-B row loads (8 lanes per line), parks them in a wave-private LDS image ([32][36] floats, the
// conflict-free b128 fragment layout), and feeds the MFMAs from there -- fragment-shaped loads straight
// to registers touch 32 cache lines per instruction and thrash the 32 KB L1 (measured 5x slower).  No
// workgroup barrier inside the K loop; the 8 partial tiles are reduced through LDS at the end.  Several independent products ("jobs") ride in ONE launch so a decoder step is a handful of
// launches, and the LSTM cell arithmetic (reference caption_src/sub_modules.py:752-767) runs in the
// epilogue of the product that feeds it: tiles of the cell products are laid out as 8 hidden units
// x 4 gates so one tile holds everything a unit's cell update needs.
//
//   job:   C[M,N] (+)= sum_s A_s[M,K_s] * op(B_s) + sum biases        (up to 3 K-segments)
//   op(B): (N,K) row-major "k-contiguous" (nn.Linear weight, forward)  or  (K,N) row-major (data gradient)
#include "xg_common.h"
#include "xg_kernels.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SKW = 8;            // waves per workgroup (K split)
constexpr int SKT = SKW * 64;     // threads
constexpr int CK = 32;            // k-chunk depth staged per wave
constexpr int LDR = CK + 4;       // LDS row stride (floats): 9 16-B slots -> conflict-free b128 fragment reads
constexpr int OPF = 32 * LDR;     // floats of one staged operand chunk

// A staged chunk is 32 rows x 32 k (k-contiguous operand) or 32 k x 32 n (n-contiguous operand); either way lane l
// moves 16 B pieces (i*8 + l/8, 4*(l%8)) for i = 0..3 -- every 8 lanes read one full 128-B line -- and the chunk
// lands in LDS as [i*8 + l/8][4*(l%8)..+3] with row stride LDR.  Out-of-range ROWS are clamped to a valid row
// (their products land in output rows / columns that are never stored), so only the k tail needs predication.
struct ChunkPtr {
    const float* p[4];     // this lane's four piece pointers at chunk 0
    int step;              // floats to advance per chunk
};

template <bool VEC>
__device__ __forceinline__ void ld_chunk(const ChunkPtr& cp, int c, bool tail, int valid /*floats valid from the piece start*/,
                                         int tail_stride /* validity shrink per i (n-contig: 8 rows per i) */, f32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float* src = cp.p[i] + (size_t)c * cp.step;
        if (!tail) {
            if (VEC) v[i] = *reinterpret_cast<const f32x4*>(src);
            else { v[i][0] = src[0]; v[i][1] = src[1]; v[i][2] = src[2]; v[i][3] = src[3]; }
        } else {
            const int ok = valid - i * tail_stride;   // k-contig: #valid floats from this piece; n-contig: #valid k rows
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (tail_stride == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (q < ok) t[q] = src[q];
            } else if (ok > 0) {
                if (VEC) t = *reinterpret_cast<const f32x4*>(src);
                else { t[0] = src[0]; t[1] = src[1]; t[2] = src[2]; t[3] = src[3]; }
            }
            v[i] = t;
        }
    }
}
__device__ __forceinline__ void st_chunk(float* __restrict__ lds, int lane, const f32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(lds + (i * 8 + (lane >> 3)) * LDR + ((lane & 7) << 2)) = v[i];
}

#ifdef SK_TRACE
// In-kernel phase stamps (tools/sk_trace_run.py): 8 slots per workgroup of the launches whose grid matches the filter set through
// xg_debug_sk_trace_filter (jobs, grid x).  Slot 7 = the LAST wave's end of the K loop (atomic max), the others are wave 0's view.
__device__ long long sk_trace_buf[4096 * 8];
__device__ int sk_trace_sel[3] = {1, 256, 0};
#define SK_TRACE_ON() ((int)gridDim.y == sk_trace_sel[0] && (int)gridDim.x == sk_trace_sel[1] && (sk_trace_sel[2] == 0 || (int)blockDim.x == sk_trace_sel[2]))
#define SK_STAMP(i) do { if (threadIdx.x == 0 && SK_TRACE_ON()) sk_trace_buf[(blockIdx.x + blockIdx.y * gridDim.x) * 8 + (i)] = wall_clock64(); } while (0)
#define SK_STAMP_MAX(i) do { if ((threadIdx.x & 63) == 0 && SK_TRACE_ON()) atomicMax((unsigned long long*)&sk_trace_buf[(blockIdx.x + blockIdx.y * gridDim.x) * 8 + (i)], (unsigned long long)wall_clock64()); } while (0)
#else
#define SK_STAMP(i) do {} while (0)
#define SK_STAMP_MAX(i) do {} while (0)
#endif

// ---- bf16 arithmetic (XgRun.gemm_mode = 1, BASELINE.json configs[4]): the staged chunk is rounded to bf16 on its way
// into LDS ([32][40] bf16 images: 80-B rows, conflict-free b128 fragments) and the products run on
// v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate); accumulation, reduction and the cell arithmetic stay fp32.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDH = CK + 8;       // bf16 row stride
__device__ __forceinline__ unsigned bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void st_chunk_bf16(unsigned short* __restrict__ lds, int lane, const f32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));      // v_cvt_pk_bf16_f32: RNE, one instruction per pair
        bf16x2_t lo, hi;
        lo[0] = (__bf16)v[i][0]; lo[1] = (__bf16)v[i][1]; hi[0] = (__bf16)v[i][2]; hi[1] = (__bf16)v[i][3];
        uint2 pk;
        pk.x = __builtin_bit_cast(unsigned, lo); pk.y = __builtin_bit_cast(unsigned, hi);
        *reinterpret_cast<uint2*>(lds + (i * 8 + (lane >> 3)) * LDH + ((lane & 7) << 2)) = pk;
    }
}

// ---- split-bf16 arithmetic (XgRun.gemm_mode = 3): an fp32 value is the EXACT sum of three bf16 planes (8 significand bits
// each: hi = x truncated, mid = (x - hi) truncated, lo = the rest, rounded), and a product keeps the six plane products whose
// weight is above fp32 round-off (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi) on v_mfma_f32_32x32x16_bf16: 12 MFMAs of 32
// cycles per 32-deep chunk instead of 16 of 64 -- 0.375 of the exact-fp32 matrix time at fp32-class accuracy.  The weights stay
// the fp32 packed tiles (split in registers as they arrive); the activations are split while they are staged into three LDS
// plane images.  Accumulation, reductions and every epilogue are fp32 as in the other modes.
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    p0 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);                      // {hi16(a), hi16(b)}: two bf16, a in the low half
    const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);      // exact
    const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
    p1 = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    const float sa = ra - __uint_as_float(va & 0xFFFF0000u), sb = rb - __uint_as_float(vb & 0xFFFF0000u);    // exact
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t lo;
    lo[0] = (__bf16)sa; lo[1] = (__bf16)sb;                                // v_cvt_pk_bf16_f32 (round to nearest even)
    p2 = __builtin_bit_cast(unsigned, lo);
}
constexpr int PLH = 32 * LDH;     // bf16 elements of one staged plane image
// A chunk -> three plane images [3][32][LDH] (same index map as st_chunk_bf16)
__device__ __forceinline__ void st_chunk_split3(unsigned short* __restrict__ lds, int lane, const f32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned a0, a1, a2, b0, b1, b2;
        split3_pair(v[i][0], v[i][1], a0, a1, a2);
        split3_pair(v[i][2], v[i][3], b0, b1, b2);
        unsigned short* q = lds + (i * 8 + (lane >> 3)) * LDH + ((lane & 7) << 2);
        *reinterpret_cast<uint2*>(q) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(q + PLH) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(q + 2 * PLH) = make_uint2(a2, b2);
    }
}
// the 8 fp32 weights a lane holds for one 16-deep MFMA block (two f32x4 pieces) -> three bf16x8 operands
__device__ __forceinline__ void split3_b(const f32x4& x, const f32x4& y, bf16x8 (&pl)[3]) {
    unsigned w[3][4];
    split3_pair(x[0], x[1], w[0][0], w[1][0], w[2][0]);
    split3_pair(x[2], x[3], w[0][1], w[1][1], w[2][1]);
    split3_pair(y[0], y[1], w[0][2], w[1][2], w[2][2]);
    split3_pair(y[2], y[3], w[0][3], w[1][3], w[2][3]);
#pragma unroll
    for (int p = 0; p < 3; ++p) { const uint4 u = {w[p][0], w[p][1], w[p][2], w[p][3]}; pl[p] = __builtin_bit_cast(bf16x8, u); }
}

// ---- epilogues shared by the two kernels.  red = [SKW][32][RS] partial tiles in LDS.
// LSTM cell epilogue operands: every load is unconditional (absent operands read a valid dummy address and are dropped by
// a select in the epilogue): a load under a branch makes the compiler wait for it on the spot, and 19 serialized L2 round
// trips were 2 us of prologue.
struct LstmPre {
    float b0[4], b1[4], b2[4], ad[4], cp, hp, mk;
    int eb, ej;
    bool on;
};
template <int NW>
__device__ __forceinline__ LstmPre lstm_prefetch(const SkJob& job, int m0, int tn) {
    LstmPre p;
    p.cp = 0.f; p.hp = 0.f; p.mk = 1.f;
    const int em = threadIdx.x >> 3, eu = threadIdx.x & 7;
    p.eb = m0 + em; p.ej = tn * 8 + eu;
    p.on = job.epi == SK_EPI_LSTM && threadIdx.x < 256 && p.eb < job.M && p.ej < job.R;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) { p.b0[gi] = 0.f; p.b1[gi] = 0.f; p.b2[gi] = 0.f; p.ad[gi] = 0.f; }
    if (p.on) {
        const int R = job.R;
        const float* dummy = job.c_prev + (size_t)p.eb * job.ldcp + p.ej;
        const float* mkp = job.mask ? job.mask + (size_t)p.eb * job.ldm : dummy;
        const float* hpp = job.mask_mode == XG_MASK_HOLD ? job.h_prev + (size_t)p.eb * job.ldhp + p.ej : dummy;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int col = gi * R + p.ej;
            p.b0[gi] = *(job.bias[0] ? job.bias[0] + col : dummy);
            p.b1[gi] = *(job.bias[1] ? job.bias[1] + col : dummy);
            p.b2[gi] = *(job.bias[2] ? job.bias[2] + col : dummy);
            p.ad[gi] = *(job.add ? job.add + (size_t)p.eb * job.ldadd + col : dummy);
        }
        p.cp = *dummy;
        p.mk = *mkp;
        p.hp = *hpp;
    }
    return p;
}

// pointwise LSTM backward of one (row b, unit j) given dh = d(loss)/d(h') before the optional add term (same arithmetic as
// xg_pointwise.hip:lstm_bwd_body)
// (the cell's own operands, requested ahead of the value they are combined with: lstmb_load, then lstmb_finish)
struct LstmbOps { float add, ig, fg, og, gg, cp, cn, mk, dc; };
__device__ __forceinline__ LstmbOps lstmb_load(const SkJob& job, int b, int j) {
    const int R = job.R;
    LstmbOps o;
    o.add = job.add ? job.add[(size_t)b * job.ldadd + j] : 0.0f;
    const float* g = job.gates + (size_t)b * job.ldg;
    o.ig = g[j]; o.fg = g[R + j];
    o.og = job.order == XG_ORDER_IFOG ? g[2 * R + j] : g[3 * R + j];
    o.gg = job.order == XG_ORDER_IFOG ? g[3 * R + j] : g[2 * R + j];
    o.cp = job.c_prev[(size_t)b * job.ldcp + j];
    o.cn = job.c_out[(size_t)b * job.ldco + j];
    o.mk = job.mask ? job.mask[(size_t)b * job.ldm] : 1.0f;
    o.dc = job.dc_in ? job.dc_in[(size_t)b * job.lddci + j] : 0.0f;
    return o;
}
__device__ __forceinline__ void lstmb_finish(const SkJob& job, int b, int j, float v, const LstmbOps& o) {
    const int R = job.R;
    if (job.add) v += o.add;
    const float ig = o.ig, fg = o.fg, og = o.og, gg = o.gg, cp = o.cp, cn = o.cn, mk = o.mk;
    float dh = v * xg_keep(job.drop, (uint32_t)(b * R + j));
    float dc = o.dc;
    float dht, dct, dcp;
    const float tc = xg_tanh(cn);
    if (job.mask_mode == XG_MASK_HOLD) {
        if (job.dh_hold) job.dh_hold[(size_t)b * job.lddhh + j] = (1.0f - mk) * dh;
        dht = mk * dh;
        dc += dht * og * (1.0f - tc * tc);
        dcp = (1.0f - mk) * dc;
        dct = mk * dc;
    } else {
        dht = mk * dh;
        dct = mk * dc + dht * og * (1.0f - tc * tc);
        dcp = 0.0f;
    }
    const float d_o = dht * tc;
    dcp += dct * fg;
    const float d_f = dct * cp, d_i = dct * gg, d_g = dct * ig;
    float* ds = job.ds + (size_t)b * job.ldds;
    ds[j] = d_i * ig * (1.0f - ig);
    ds[R + j] = d_f * fg * (1.0f - fg);
    const float dso = d_o * og * (1.0f - og), dsg = d_g * (1.0f - gg * gg);
    if (job.order == XG_ORDER_IFOG) { ds[2 * R + j] = dso; ds[3 * R + j] = dsg; }
    else                            { ds[2 * R + j] = dsg; ds[3 * R + j] = dso; }
    job.dc_prev[(size_t)b * job.lddcp + j] = dcp;
}
__device__ __forceinline__ void lstmb_point(const SkJob& job, int b, int j, float v) { lstmb_finish(job, b, j, v, lstmb_load(job, b, j)); }

template <int RS, int NW>
__device__ __forceinline__ void sk_epilogue(const SkJob& job, const float* __restrict__ redp, int m0, int n0, const LstmPre& pre) {
    const float (*red)[32][RS] = reinterpret_cast<const float (*)[32][RS]>(redp);
    const int R = job.R;
    if (job.epi == SK_EPI_STORE) {
#pragma unroll
        for (int e = 0; e < 1024 / (NW * 64); ++e) {
            const int idx = threadIdx.x + NW * 64 * e;
            const int m = idx >> 5, c = idx & 31;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[w][m][c];
            const int row = m0 + m;
            // n0 = 32 tn; cell tiling: column = gate (c / 8) of hidden unit 8 tn + c % 8
            const int unit = (n0 >> 2) + (c & 7);
            const int col = job.cell_cols ? (unit < R ? (c >> 3) * R + unit : job.N) : n0 + c;
            if (row < job.M && col < job.N) {
                if (job.bias[0]) v += job.bias[0][col];
                if (job.bias[1]) v += job.bias[1][col];
                if (job.bias[2]) v += job.bias[2][col];
                float* dst = job.C + (size_t)row * job.ldc + col;
                if (job.accumulate) v += *dst;
                if (job.relu) v = fmaxf(v, 0.f);
                *dst = v;
            }
        }
    } else if (job.epi == SK_EPI_LSTMB) {
        // pointwise LSTM backward of the step whose dh this product completes (same arithmetic as lstm_bwd_body)
#pragma unroll
        for (int e = 0; e < 1024 / (NW * 64); ++e) {
            const int idx = threadIdx.x + NW * 64 * e;
            const int m = idx >> 5, c = idx & 31;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[w][m][c];
            const int b = m0 + m, j = n0 + c;
            if (b < job.M && j < job.N) {
                if (job.accumulate) v += job.C[(size_t)b * job.ldc + j];
                lstmb_point(job, b, j, v);
            }
        }
    } else if (job.epi == SK_EPI_GATE) {
#pragma unroll
        for (int e = 0; e < 1024 / (NW * 64); ++e) {
            const int idx = threadIdx.x + NW * 64 * e;
            const int m = idx >> 5, c = idx & 31;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[w][m][c];
            const int row = m0 + m, col = n0 + c;
            if (row < job.M && col < job.N) {
                if (job.bias[0]) v += job.bias[0][col];
                const float g = fmaxf(v, 0.f) * xg_keep(job.drop, (uint32_t)(row * job.N + col));
                job.C[(size_t)row * job.ldc + col] = g;
                const float tv = job.gate_t[(size_t)row * job.ldt + col];
                job.gate_y[(size_t)row * job.ldy + col] = g * tv + tv;
            }
        }
    } else {
        // LSTM cell epilogue: thread -> (row em, unit eu); its four gate pre-activations sit at columns eu + 8*gate
        const int em_ = threadIdx.x >> 3, eu_ = threadIdx.x & 7;
        if (pre.on) {
            const int b = pre.eb, j = pre.ej;
            float s4[4];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                float v = (job.bias[0] ? pre.b0[gi] : 0.f) + (job.bias[1] ? pre.b1[gi] : 0.f) + (job.bias[2] ? pre.b2[gi] : 0.f) +
                          (job.add ? pre.ad[gi] : 0.f);
#pragma unroll
                for (int w = 0; w < NW; ++w) v += red[w][em_][gi * 8 + eu_];
                s4[gi] = v;
            }
            const float so = job.order == XG_ORDER_IFOG ? s4[2] : s4[3];
            const float sg_ = job.order == XG_ORDER_IFOG ? s4[3] : s4[2];
            const float ig = xg_sigmoid(s4[0]), fg = xg_sigmoid(s4[1]), og = xg_sigmoid(so), gg = xg_tanh(sg_);
            const float cp = pre.cp, mk = job.mask ? pre.mk : 1.0f;
            float cn = fg * cp + ig * gg, hn;
            if (job.mask_mode == XG_MASK_HOLD) {
                cn = cn * mk + cp * (1.0f - mk);
                hn = og * xg_tanh(cn);
                hn = hn * mk + pre.hp * (1.0f - mk);
            } else {
                hn = og * xg_tanh(cn) * mk;
                cn = cn * mk;
            }
            hn *= xg_keep(job.drop, (uint32_t)(b * R + j));
            if (job.gates) {
                float* g = job.gates + (size_t)b * job.ldg;
                g[j] = ig; g[R + j] = fg;
                if (job.order == XG_ORDER_IFOG) { g[2 * R + j] = og; g[3 * R + j] = gg; }
                else                            { g[2 * R + j] = gg; g[3 * R + j] = og; }
            }
            job.c_out[(size_t)b * job.ldco + j] = cn;
            job.h_out[(size_t)b * job.ldho + j] = hn;
        }
    }
}

// Split-K epilogue (see SkJob.ksplit_ok).  STORE: this part's tile goes into C with atomics.  LSTMB (round 6): no atomics and no
// ticket.  Parts 0 .. ksplit - 2 PUBLISH their partial tile as 8-byte {tag = 1, value} granules (one agent-scope store each:
// cdna_hip_programming.md guideline 16, form R2 -- the data is the flag) and are done; the workgroups of the LAST part are the
// highest block indices of the launch (tile decode of skf_kernel), i.e. dispatched behind every publisher, and each of their
// threads collects its elements' ksplit - 1 granules (requested together with the cell's own operands, re-read until every tag is
// there), adds them in part order -- the sum no longer depends on the arrival order: the chains' dh is bit-reproducible -- puts
// the granules back to zero and runs the cell's pointwise backward.  Before: every part added its tile into C with atomics, waited
// for them, took a ticket (a returning atomic), and the last arriver re-read C: three dependent memory round trips behind the K
// loop of every part (6.6 us of the reverse-time step, in-kernel stamps of round 5) against one for the last part only.
typedef __attribute__((address_space(1))) unsigned long long sk_gu64;
template <int RS, int NW>
__device__ __forceinline__ void sk_epilogue_split(const SkJob& job, float* __restrict__ smem, int m0, int n0, int kp, int tile_id) {
    const float (*red)[32][RS] = reinterpret_cast<const float (*)[32][RS]>(smem);
    if (job.epi == SK_EPI_LSTMB) {
        constexpr int EPT = 1024 / (NW * 64);
        const int ks = job.ksplit;
        sk_gu64* slab = (sk_gu64*)(unsigned long long*)job.tickets + (size_t)tile_id * (ks - 1) * 1024;
        float v[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int idx = threadIdx.x + NW * 64 * e;
            const int m = idx >> 5, c = idx & 31;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red[w][m][c];
            v[e] = s;
        }
        if (kp != ks - 1) {                                       // a publishing part: one 8-byte store per element, nothing to wait for
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                __hip_atomic_store(slab + (size_t)kp * 1024 + threadIdx.x + NW * 64 * e,
                                   (1ull << 32) | (unsigned long long)__float_as_uint(v[e]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        // the finishing part: the cell's operands and what C already holds (earlier launches) are requested first, then the other
        // parts' tiles (all of them per pass: one round trip when the publishers are through, which they usually are)
        LstmbOps ops[EPT];
        float cacc[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int idx = threadIdx.x + NW * 64 * e;
            const int b = min(m0 + (idx >> 5), job.M - 1), j = min(n0 + (idx & 31), job.N - 1);      // (clamped: never stored)
            ops[e] = lstmb_load(job, b, j);
            cacc[e] = job.accumulate ? job.C[(size_t)b * job.ldc + j] : 0.f;
        }
        constexpr int PMAX = 3;                                   // ksplit <= 4 for LSTMB jobs (xgk_skinny): 24 registers of granules at four elements per thread
        unsigned long long g[PMAX][EPT];
        for (unsigned spins = 0;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int p = 0; p < PMAX; ++p) {
                if (p < ks - 1) {
#pragma unroll
                    for (int e = 0; e < EPT; ++e) {
                        g[p][e] = __hip_atomic_load(slab + (size_t)p * 1024 + threadIdx.x + NW * 64 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = ok && (g[p][e] >> 32) == 1ull;
                    }
                }
            }
            if (ok) break;
            if (spins > 16) __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) v[e] += cacc[e];
#pragma unroll
        for (int p = 0; p < PMAX; ++p) {
            if (p < ks - 1) {
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    v[e] += __uint_as_float((unsigned)g[p][e]);
                    __hip_atomic_store(slab + (size_t)p * 1024 + threadIdx.x + NW * 64 * e, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int idx = threadIdx.x + NW * 64 * e;
            const int b = m0 + (idx >> 5), j = n0 + (idx & 31);
            if (b < job.M && j < job.N) lstmb_finish(job, b, j, v[e], ops[e]);
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 1024 / (NW * 64); ++e) {
        const int idx = threadIdx.x + NW * 64 * e;
        const int m = idx >> 5, c = idx & 31;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][m][c];
        const int row = m0 + m, col = n0 + c;
        if (row < job.M && col < job.N) {
            if (kp == 0 && job.epi == SK_EPI_STORE) {
                if (job.bias[0]) v += job.bias[0][col];
                if (job.bias[1]) v += job.bias[1][col];
                if (job.bias[2]) v += job.bias[2][col];
            }
            atomicAdd(job.C + (size_t)row * job.ldc + col, v);
        }
    }
}

// 4 waves per SIMD = two 512-thread workgroups per CU (2 x 74 KB of LDS fit): launches with more than 256 tiles (the
// encoder's two cells, [p || cell 1]) then run in one round.  Needs <= 128 VGPRs.
template <bool VEC, int PREC>     // PREC 0: fp32 MFMA, 1: bf16 MFMA
__global__ void __launch_bounds__(SKT) __attribute__((amdgpu_waves_per_eu(4, 4))) sk_kernel(SkArgs args) {
    XG_CHAIN_PRIO();
    SK_STAMP(0);
    // wave-private staging (A chunk + B chunk per wave), re-used as the [SKW][32][32] reduction buffer
    __shared__ __attribute__((aligned(16))) float smem[SKW * 2 * OPF];
    float (*red)[32][32] = reinterpret_cast<float (*)[32][32]>(smem);
    // ---- which job / tile (XCD-aware: the m-tiles that share a weight slice stay on one XCD's L2)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int ji = 0;
#pragma unroll
    for (int j = 1; j < SK_MAX_JOBS; ++j) if (j < args.njobs && bid >= args.tile0[j]) ji = j;
    const SkJob& job = args.job[ji];
    const int tile = bid - args.tile0[ji];
    const int ntm = (job.M + 31) >> 5;
    const int tm = tile % ntm, tn = tile / ntm;
    const int m0 = tm * 32, n0 = tn * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int R = job.R;
    const bool lstm = job.epi == SK_EPI_LSTM || job.cell_cols;      // cell tiling of the weight rows
    float* As = smem + wave * 2 * OPF;
    float* Bs = As + OPF;
    const int lrow = lane >> 3, lcol = (lane & 7) << 2;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // ---- LSTM epilogue operands are requested NOW so their (cold-L2) latency hides under the K loop
    const LstmPre pre = lstm_prefetch<SKW>(job, m0, tn);

    // ---- K loop: this wave's share of the 32-deep chunks of every segment
    int nc_total = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) if (s < job.nseg) nc_total += (job.seg[s].K + CK - 1) / CK;
    const int wc0 = (wave * nc_total) / SKW, wc1 = ((wave + 1) * nc_total) / SKW;
    int seg_start = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s >= job.nseg) break;
        const SkSeg sg = job.seg[s];
        const int nc = (sg.K + CK - 1) / CK;
        const int c0 = max(wc0, seg_start) - seg_start, c1 = min(wc1, seg_start + nc) - seg_start;
        seg_start += nc;
        if (c0 >= c1) continue;
        const bool bn = sg.b_ncontig != 0;
        ChunkPtr pa, pb;
        pa.step = CK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 8 + lrow;
            pa.p[i] = sg.A + (size_t)min(m0 + r, job.M - 1) * sg.lda + lcol;
            if (bn) {          // piece = k row (i*8 + lrow) of the chunk, 4 consecutive n
                int n = n0 + lcol;
                if (n + 3 >= job.N) n = max(job.N - 4, 0);     // clamped columns are never stored (N >= 4 on this path)
                pb.p[i] = sg.B + (size_t)r * sg.ldb + n;
            } else {
                int wrow;
                if (lstm) wrow = (r >> 3) * R + min(tn * 8 + (r & 7), R - 1);
                else wrow = min(n0 + r, job.N - 1);
                pb.p[i] = sg.B + (size_t)wrow * sg.ldb + lcol;
            }
        }
        pb.step = bn ? CK * sg.ldb : CK;
        const int nfull = sg.K / CK;                         // chunks < nfull need no k predication
        // validity of this lane's pieces inside the (single) tail chunk
        const int ktail = sg.K - nfull * CK;
        const int va = ktail - lcol;                         // k-contig: valid floats from the piece start
        const int vb = bn ? ktail - lrow : ktail - lcol;     // n-contig: valid k rows below this piece's row
        f32x4 ra[4], rb[4];
        if (s == 0) SK_STAMP(1);
        ld_chunk<VEC>(pa, c0, c0 >= nfull, va, 0, ra);
        ld_chunk<VEC>(pb, c0, c0 >= nfull, vb, bn ? 8 : 0, rb);
        for (int c = c0; c < c1; ++c) {
            if (PREC == 1) {
                st_chunk_bf16(reinterpret_cast<unsigned short*>(As), lane, ra);
                st_chunk_bf16(reinterpret_cast<unsigned short*>(Bs), lane, rb);
            } else {
                st_chunk(As, lane, ra);
                st_chunk(Bs, lane, rb);
            }
            if (s == 0 && c == c0) SK_STAMP(2);
            if (c + 1 < c1) {   // next chunk's global loads fly while this chunk's MFMAs run
                ld_chunk<VEC>(pa, c + 1, c + 1 >= nfull, va, 0, ra);
                ld_chunk<VEC>(pb, c + 1, c + 1 >= nfull, vb, bn ? 8 : 0, rb);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (PREC == 1) {
                const unsigned short* Ah = reinterpret_cast<const unsigned short*>(As);
                const unsigned short* Bh = reinterpret_cast<const unsigned short*>(Bs);
#pragma unroll
                for (int kb = 0; kb < CK / 16; ++kb) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ah + l31 * LDH + kb * 16 + half * 8);
                    bf16x8 b;
                    if (bn) {
                        const unsigned short* q = Bh + (kb * 16 + half * 8) * LDH + l31;
#pragma unroll
                        for (int i = 0; i < 8; ++i) b[i] = (short)q[i * LDH];
                    } else {
                        b = *reinterpret_cast<const bf16x8*>(Bh + l31 * LDH + kb * 16 + half * 8);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                }
            } else
#pragma unroll
            for (int kb = 0; kb < CK / 8; ++kb) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(As + l31 * LDR + kb * 8 + half * 4);
                f32x4 b;
                if (bn) {
                    const float* q = Bs + (kb * 8 + half * 4) * LDR + l31;
                    b[0] = q[0]; b[1] = q[LDR]; b[2] = q[2 * LDR]; b[3] = q[3 * LDR];
                } else {
                    b = *reinterpret_cast<const f32x4*>(Bs + l31 * LDR + kb * 8 + half * 4);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    SK_STAMP(3);
    __syncthreads();   // every wave is done with its staging area before it becomes the reduction buffer
    // ---- reduce the SKW partial tiles through LDS
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
    __syncthreads();
    SK_STAMP(4);

    sk_epilogue<32, SKW>(job, smem, m0, n0, pre);
    SK_STAMP(5);
}

// ---- ZERO job: one tile = 4096 floats
template <int NW>
__device__ __forceinline__ void zero_tile(const SkJob& job, int tile) {
    const size_t n = (size_t)job.M * job.N;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 1024 / (NW * 64); ++u) {
        const size_t i = ((size_t)tile * 1024 + u * (NW * 64) + threadIdx.x) * 4;
        if (i + 3 < n) *reinterpret_cast<f32x4*>(job.C + i) = z;
        else for (size_t j = i; j < n; ++j) job.C[j] = 0.f;
    }
}

// ---- COPY job: one tile = 4096 floats
template <int NW>
__device__ __forceinline__ void copy_tile(const SkJob& job, int tile) {
    const size_t n = (size_t)job.M * job.N;
    const float* src = job.seg[0].A;
#pragma unroll
    for (int u = 0; u < 1024 / (NW * 64); ++u) {
        const size_t i = ((size_t)tile * 1024 + u * (NW * 64) + threadIdx.x) * 4;
        if (i + 3 < n) *reinterpret_cast<f32x4*>(job.C + i) = *reinterpret_cast<const f32x4*>(src + i);
        else for (size_t j = i; j < n; ++j) job.C[j] = src[j];
    }
}

// ---- ATTN job: see SK_EPI_ATTN in xg_kernels.h.  Reference: caption_src/sub_modules.py:678-680.
//   EVERY wave of the workgroup takes part (NWV = 4 or 8).  Round 4: the 8-wave launches used to let waves 4..7 return before
//   the first workgroup barrier; beside split-bf16 cell tiles in the same launch that form produced run-to-run differences in
//   the accumulated context (one video's row, ~1e-4, 4 of 5 runs; tools/step_mode_check.py) -- never seen with the fp32 tiles,
//   but nothing in the code tied it to the arithmetic, so no wave leaves a workgroup ahead of a barrier any more.
//   wave w scores rows k0 + w, k0 + w + NWV, ... of this half (whole q rows, lane = 16 B); both halves use e_0 as the
//   softmax shift (the second half scores frame 0 once more: same arithmetic, same bits), so their unnormalised sums simply
//   add.  exp argument clamped at 80 (only reachable when a frame outweighs the first by e^80; keeps that finite).
constexpr int ATT_ROWS = 64;       // rows per half <= 64 (K <= 128)
template <int NI, int NWV>         // float4 groups of A per lane: A <= 256 NI; waves of the workgroup (all of them take part)
__device__ __forceinline__ void attn_part(const SkJob& job, int tile, float* smem) {
    const int b = tile >> 1, part = tile & 1;
    const int K = job.attn_K, A = job.attn_A, R = job.R;
    const int kh = (K + 1) >> 1, k0 = part * kh, k1 = min(K, k0 + kh), nrow = k1 - k0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* pb = job.attn_p + (size_t)b * A;
    const float* qb = job.attn_q + (size_t)b * K * A;
    const float* Vb = job.attn_v + (size_t)b * K * R;
    float* se = smem;                  // [0, ATT_ROWS): scores of this half; [ATT_ROWS]: e_0; then ex
    float* sx = smem + ATT_ROWS + 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 pr[NI], wr[NI], q[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int a = lane * 4 + 256 * i;
        pr[i] = a < A ? *reinterpret_cast<const f32x4*>(pb + a) : z;
        wr[i] = a < A ? *reinterpret_cast<const f32x4*>(job.attn_w + a) : z;
    }
    // the second half's reference row (frame 0) rides on its last wave, which has the fewest rows of its own
    const bool ref_here = part == 1 && wave == NWV - 1;
    const int own = k0 + wave < k1 ? (k1 - k0 - wave + NWV - 1) / NWV : 0;
    const int cnt = own + (ref_here ? 1 : 0);
    for (int j = 0; j < cnt; ++j) {
        const float* qk = qb + (size_t)(j < own ? k0 + wave + NWV * j : 0) * A;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int a = lane * 4 + 256 * i;
            q[i] = a < A ? *reinterpret_cast<const f32x4*>(qk + a) : z;
        }
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            acc += wr[i][0] * xg_tanh(pr[i][0] + q[i][0]) + wr[i][1] * xg_tanh(pr[i][1] + q[i][1]) +
                   wr[i][2] * xg_tanh(pr[i][2] + q[i][2]) + wr[i][3] * xg_tanh(pr[i][3] + q[i][3]);
        acc = wave_sum(acc);
        if (lane == 0) se[j < own ? wave + NWV * j : ATT_ROWS] = acc;
    }
    __syncthreads();
    const float eref = part == 0 ? se[0] : se[ATT_ROWS];
    if (wave == 0) {
        const float ex = lane < nrow ? __expf(fminf(se[lane] - eref, 80.0f)) : 0.f;
        const float ssum = wave_sum(ex);
        if (lane < nrow) { sx[lane] = ex; job.attn_ex[(size_t)b * K + k0 + lane] = ex; }
        if (lane == 0 && nrow > 0) atomicAdd(job.attn_s + b, ssum);
    }
    __syncthreads();
    // unnormalised context of this half: thread -> two adjacent columns
    for (int c = threadIdx.x * 2; c < R; c += NWV * 128) {
        float ax = 0.f, ay = 0.f;
        // four rows per trip, written out: the four loads are requested first, then the four weights come out of LDS, then the
        // multiply-adds as plain v_fmac_f32, PINNED.  Why (docs/pkfma_hazard.md): left to the SLP vectorizer the two accumulators
        // become one v_pk_fma_f32 per frame with the weight broadcast by operand select, and the form `op_sel:[0,1,0]` (the LOW
        // lane takes the HIGH register of the weight pair) now and then loses its low-lane product in lanes 48-63 of the wave --
        // one frame's term missing from up to 16 even context columns of one video -- but only while split-bf16 cell tiles share
        // the CU (128-row launches; never beside fp32 / plain bf16 tiles, never in the stand-alone reproducer).  Round 6 bisection
        // (diag_pkfma_bisect.inc): the same four FMAs as op_sel_hi:[1,0,1] only, as packed FMAs without operand select, or as
        // these scalar ones never differ.  Guards: this pinned form, -fno-slp-vectorize (__graft_entry__.FLAGS), and
        // tests/test_abi_cpu.py, which disassembles the product library and fails on any packed-fp32 `op_sel:`.
        const float* vp = Vb + (size_t)k0 * R + c;
        int r = 0;
        for (; r + 4 <= nrow; r += 4) {
            const float2 v0 = *reinterpret_cast<const float2*>(vp + (size_t)(r + 0) * R);
            const float2 v1 = *reinterpret_cast<const float2*>(vp + (size_t)(r + 1) * R);
            const float2 v2 = *reinterpret_cast<const float2*>(vp + (size_t)(r + 2) * R);
            const float2 v3 = *reinterpret_cast<const float2*>(vp + (size_t)(r + 3) * R);
            const float s0 = sx[r], s1 = sx[r + 1], s2 = sx[r + 2], s3 = sx[r + 3];
#if defined(SKF_PK_ASM)   // diagnosis builds: the packed forms of docs/pkfma_hazard.md
#include "diag_pkfma_bisect.inc"
#else
            // plain v_fmac_f32, pinned (the compiler pairs ax / ay into v_pk_fma_f32 with operand-select modifiers otherwise)
            asm volatile("v_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %4" : "+v"(ax), "+v"(ay) : "v"(s0), "v"(v0.x), "v"(v0.y));
            asm volatile("v_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %4" : "+v"(ax), "+v"(ay) : "v"(s1), "v"(v1.x), "v"(v1.y));
            asm volatile("v_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %4" : "+v"(ax), "+v"(ay) : "v"(s2), "v"(v2.x), "v"(v2.y));
            asm volatile("v_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %4" : "+v"(ax), "+v"(ay) : "v"(s3), "v"(v3.x), "v"(v3.y));
#endif
        }
        for (; r < nrow; ++r) {
            const float2 v = *reinterpret_cast<const float2*>(vp + (size_t)r * R);
            ax += sx[r] * v.x; ay += sx[r] * v.y;
        }
        atomicAdd(job.attn_c + (size_t)b * R + c, ax);
        atomicAdd(job.attn_c + (size_t)b * R + c + 1, ay);
    }
}

// ================================================================================================
// Fast kernel: every segment's B operand is a PACKED weight (xg_pack.hip), fp32.
//   * B goes global -> VGPR: four 1 KB-coalesced wave loads per 32 x 32 tile, used as MFMA operands as they are (no LDS
//     write, no LDS read, no shuffle).  The LDS-staged kernel above spends more LDS-pipe time per chunk than the 16 MFMAs
//     it feeds: with no global loads at all its K loop is 64 % MFMA-busy.
//   * A (the activations: 32 rows of this m-tile) still goes through the wave-private LDS image; the k order inside a
//     32-deep chunk is the packed one (lane half h owns k in [16 h, 16 h + 16)).  Optional row gather (embedding lookup).
//   * the job is blockIdx.y.  Round 5: the job's 64-byte head and the 48-byte hot part of each segment arrive through ONE
//     round of wide scalar loads at kernel entry (s_load_dwordx16 + 3 x (x8 + x4)), the gathered segment's row indices are
//     requested right behind them, the epilogue's input block (bias / cell-state pointers, 112 B) behind the first operand
//     request, its output block behind the K loop.  Before, every field was read where it was first used -- ~15 dependent
//     scalar-cache round trips between kernel entry and the first operand request (in-kernel stamps: 2.1-7.6 us per tile,
//     profiles/r05_sk_trace_before.txt), more when the CU was already busy.
//   * epilogue operands of EVERY job type (bias rows, the accumulate / gate operand, the cell state) are requested before the
//     K loop with unconditional loads; the epilogue itself issues no dependent load.
//   * reduction buffer rows are 40 floats apart: conflict-free for the column-wise reads of the cell epilogue.
constexpr int RSF = 40;
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct SkHeadBlk { int epi, M, N, R, nseg, ksplit, ntm, ntn, ntiles, hflags, ldc, nck_all; float* C; int* tickets; };
struct SkSegHot { const float* A; const float* Bp; const int64_t* gather; int lda, K, nck, a_bytes, gather_max, sflags; };
struct SkEpiIn { const float* bias[3]; const float* add; const float* c_prev; const float* h_prev; const float* mask;
                 int ldadd, ldcp, ldhp, ldm, accumulate, relu, order, mask_mode; const float* gate_t; int ldt, ldy; float* gate_y; };
struct SkEpiOut { float* gates; float* c_out; float* h_out; int ldg, ldco, ldho, cell_cols; XgDrop drop; int pad_; };
static_assert(sizeof(SkHeadBlk) == 64 && sizeof(SkSegHot) == 48 && sizeof(SkEpiIn) == 112 && sizeof(SkEpiOut) == 64, "descriptor blocks");
static_assert(offsetof(SkJob, tickets) == 56 && offsetof(SkSeg, sflags) == 44 && offsetof(SkJob, bias) + 112 == offsetof(SkJob, gates) &&
              offsetof(SkJob, gate_y) + 8 == offsetof(SkJob, gates) && offsetof(SkJob, drop) + 20 <= offsetof(SkJob, gates) + 64,
              "descriptor blocks mirror SkJob's layout (xg_kernels.h)");
// The blocks are read as typed struct copies (adjacent scalar loads of one basic block are merged into s_load_dwordx8 / x4) and
// HELD right behind the copy: an empty asm statement takes every field as an in/out scalar-register operand.  Two effects: all of
// the block's loads are issued together in front of one wait, and the values are no longer "loads from kernel-argument memory" to
// the register allocator, which otherwise re-loads them wherever it runs short of scalar registers (measured: the 112-byte
// epilogue block re-read ten times inside the operand-request code, each time with its own wait).  Blocks are loaded where their
// latency hides and live only as long as they are needed: head + first segment + epilogue inputs at entry, the next segment's
// hot part behind the current segment's first operand request, the epilogue's output block behind the last one.
__device__ __forceinline__ void sk_hold(SkHeadBlk& h) {
    asm volatile("" : "+s"(h.epi), "+s"(h.M), "+s"(h.N), "+s"(h.R), "+s"(h.nseg), "+s"(h.ksplit), "+s"(h.ntm), "+s"(h.ntn), "+s"(h.ntiles),
                 "+s"(h.hflags), "+s"(h.ldc), "+s"(h.nck_all), "+s"(h.C), "+s"(h.tickets));
}
__device__ __forceinline__ void sk_hold(SkSegHot& g) {
    asm volatile("" : "+s"(g.A), "+s"(g.Bp), "+s"(g.gather), "+s"(g.lda), "+s"(g.K), "+s"(g.nck), "+s"(g.a_bytes), "+s"(g.gather_max), "+s"(g.sflags));
}
__device__ __forceinline__ void sk_hold(SkEpiIn& e) {
    asm volatile("" : "+s"(e.bias[0]), "+s"(e.bias[1]), "+s"(e.bias[2]), "+s"(e.add), "+s"(e.c_prev), "+s"(e.h_prev), "+s"(e.mask));
    asm volatile("" : "+s"(e.ldadd), "+s"(e.ldcp), "+s"(e.ldhp), "+s"(e.ldm), "+s"(e.accumulate), "+s"(e.relu), "+s"(e.order), "+s"(e.mask_mode),
                 "+s"(e.gate_t), "+s"(e.ldt), "+s"(e.ldy), "+s"(e.gate_y));
}
// the token choice's arguments (xg_select.h): one round of wide scalar loads from the tail of the kernel arguments, held -- read
// field by field where first used they were a dozen dependent scalar-cache round trips inside the gate tiles' prologue
__device__ __forceinline__ void sk_hold(RollSelectArgs& q) {
    RollStepArgs& a = q.r;
    asm volatile("" : "+s"(a.logits), "+s"(a.uniforms), "+s"(a.forced), "+s"(a.fstride), "+s"(a.unf_prev), "+s"(a.table), "+s"(a.tok), "+s"(a.tok_logp));
    asm volatile("" : "+s"(a.unf), "+s"(a.lse), "+s"(a.seq), "+s"(a.seq_logp), "+s"(a.maxf), "+s"(a.xt), "+s"(q.part));
    asm volatile("" : "+s"(a.temperature), "+s"(a.V), "+s"(a.E), "+s"(a.t), "+s"(a.T), "+s"(a.mode), "+s"(a.split), "+s"(q.ntiles), "+s"(q.tw));
}
__device__ __forceinline__ void sk_hold(SkEpiOut& e) {
    asm volatile("" : "+s"(e.gates), "+s"(e.c_out), "+s"(e.h_out), "+s"(e.ldg), "+s"(e.ldco), "+s"(e.ldho), "+s"(e.cell_cols), "+s"(e.drop.seed),
                 "+s"(e.drop.site), "+s"(e.drop.step), "+s"(e.drop.thresh), "+s"(e.drop.scale));
}

// ---- addressing of the fast kernel (round 5).  In-kernel stamps showed the launches' fixed parts to be INSTRUCTION-ISSUE bound: a
// wave ran ~300 scalar + ~320 vector instructions around its ~42 MFMAs, four waves per SIMD take turns at one scalar and one
// vector issue slot, and the later-dispatched tiles needed 4-5 us from entry to their first operand request (profiles/
// r05_sk_trace_*.txt).  So the address arithmetic is kept off the vector unit and out of 64-bit: operands come through buffer
// descriptors (uniform base in four scalar registers, per-lane 32-bit byte offset fixed for the whole segment, the chunk as the
// scalar offset: a chunk costs one s_add per operand), epilogue operands through uniform base + 32-bit lane offset (the saddr form
// of global_load / global_store), optional operands sit behind uniform branches instead of dummy-pointer selects, and the tile
// decode uses shifts the launcher precomputed (power-of-two tile counts) instead of divisions.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_rsrc(const void* p, int nbytes = 0x7FFFFFFF) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, nbytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bld16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bst16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, (int)soff, 0);
}
// (explicit global address space: the descriptor values pass through opaque register barriers below, after which the compiler no
//  longer knows that they came from kernel arguments and would use flat instructions, which count on the LDS counter too)
#define SK_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ float ldf(const float* sbase, unsigned off) {
    return *reinterpret_cast<const SK_GLOBAL float*>((const SK_GLOBAL char*)sbase + off);
}
__device__ __forceinline__ void stf(float* sbase, unsigned off, float v) {
    *reinterpret_cast<SK_GLOBAL float*>((SK_GLOBAL char*)sbase + off) = v;
}
// Epilogue operands requested before the K loop (p arrives zero-filled from the top of the kernel, i.e. from before any load was
// in flight).  LSTM jobs: a[0..3] / a[4..7] / a[8..11] = the four gates' bias terms of this thread's unit, a[12..15] = the add
// term, st = c_prev / h_prev / mask.  STORE / GATE jobs: a[0..2] = the three bias terms of this thread's column (one column per
// thread: the 2 or 4 elements it finishes are 16 or 8 rows apart), a[4 + e] = what element e is added to (accumulate) or gated with.
struct EpiPre { float a[16]; float st[3]; };
enum { EP_CP = 0, EP_HP = 1, EP_MK = 2 };
enum { EF_B0 = 1, EF_B1 = 2, EF_B2 = 4, EF_ADD = 8, EF_MASK = 16, EF_ACC = 32, EF_RELU = 64, EF_IFOG = 128, EF_HOLD = 256 };
template <int NW>
__device__ __forceinline__ void epi_prefetch(EpiPre& p, const SkHeadBlk& hd, const SkEpiIn& ei, int ef, bool cell_tiles, int m0, int n0, int tn) {
    const int M = hd.M, N = hd.N, R = hd.R;
    if (hd.epi == SK_EPI_LSTM) {
        // (the four waves that run the cell epilogue; a wave-uniform branch, no exec-masked region)
        if (NW == 8 && __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= 4) return;
        const int em = (threadIdx.x >> 3) & 31, eu = threadIdx.x & 7;
        const unsigned eb = (unsigned)min(m0 + em, M - 1), cj = (unsigned)min(tn * 8 + eu, R - 1) * 4u, r4 = (unsigned)R * 4u;
        if (ef & EF_B0) {
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) p.a[gi] = ldf(ei.bias[0], cj + gi * r4);
        }
        if (ef & EF_B1) {
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) p.a[4 + gi] = ldf(ei.bias[1], cj + gi * r4);
        }
        if (ef & EF_B2) {
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) p.a[8 + gi] = ldf(ei.bias[2], cj + gi * r4);
        }
        if (ef & EF_ADD) {
            const unsigned ro = eb * ((unsigned)ei.ldadd * 4u) + cj;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) p.a[12 + gi] = ldf(ei.add, ro + gi * r4);
        }
        p.st[EP_CP] = ldf(ei.c_prev, eb * ((unsigned)ei.ldcp * 4u) + cj);
        if (ef & EF_MASK) p.st[EP_MK] = ldf(ei.mask, eb * ((unsigned)ei.ldm * 4u));
        if (ef & EF_HOLD) p.st[EP_HP] = ldf(ei.h_prev, eb * ((unsigned)ei.ldhp * 4u) + cj);
        // (two different empty statements at the two branch ends: the optimiser otherwise merges the branches' last stores into one
        //  store through a pointer phi, which keeps part of `p` in scratch memory -- with a wait for the load in front of the store)
        asm volatile("; cell operands requested");
    } else if (hd.ksplit <= 1 && (hd.epi == SK_EPI_STORE || hd.epi == SK_EPI_GATE)) {
        const int c = threadIdx.x & 31;
        const int unit = min((n0 >> 2) + (c & 7), R - 1);
        const unsigned cb = (unsigned)(cell_tiles ? (c >> 3) * R + unit : min(n0 + c, N - 1)) * 4u;
        if (ef & EF_B0) p.a[0] = ldf(ei.bias[0], cb);
        if (ef & EF_B1) p.a[1] = ldf(ei.bias[1], cb);
        if (ef & EF_B2) p.a[2] = ldf(ei.bias[2], cb);
        if (hd.epi == SK_EPI_GATE || (ef & EF_ACC)) {
            const float* xb = hd.epi == SK_EPI_GATE ? ei.gate_t : hd.C;                       // (uniform)
            const unsigned ldx = (unsigned)(hd.epi == SK_EPI_GATE ? ei.ldt : hd.ldc) * 4u;
#pragma unroll
            for (int e = 0; e < 1024 / (NW * 64); ++e) {
                const int m = (threadIdx.x >> 5) + NW * 2 * e;
                p.a[4 + e] = ldf(xb, (unsigned)min(m0 + m, M - 1) * ldx + cb);
            }
        }
        asm volatile("; store operands requested");
    }
}

// epilogue of the fast kernel (un-split tiles): red = [NW][32][RSF] partial tiles in LDS.  Same arithmetic, in the same order,
// as sk_epilogue above.  ef: which optional operands exist / which variant runs, folded into one scalar at the top of the kernel
// (the pointers themselves are dead once their loads are requested: fewer scalar registers live across the K loop).
template <int NW>
__device__ __forceinline__ void skf_epilogue(const SkJob& job, const SkHeadBlk& hd, int ef, float* gate_y, int ldy, const SkEpiOut& eo,
                                             bool cell_tiles, const float* __restrict__ redp, int m0, int n0, int tn, const EpiPre& pre) {
    const float (*red)[32][RSF] = reinterpret_cast<const float (*)[32][RSF]>(redp);
    const int R = hd.R, M = hd.M, N = hd.N;
    if (hd.epi == SK_EPI_STORE || hd.epi == SK_EPI_GATE) {
        const bool gate = hd.epi == SK_EPI_GATE;
        const int c = threadIdx.x & 31;
        // n0 = 32 tn; cell tiling: column = gate (c / 8) of hidden unit 8 tn + c % 8
        const int unit = (n0 >> 2) + (c & 7);
        const int col = cell_tiles ? (unit < R ? (c >> 3) * R + unit : N) : n0 + c;
        const unsigned cb = (unsigned)col * 4u, ldc4 = (unsigned)hd.ldc * 4u, ldy4 = (unsigned)ldy * 4u;
#pragma unroll
        for (int e = 0; e < 1024 / (NW * 64); ++e) {
            const int m = (threadIdx.x >> 5) + NW * 2 * e;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[w][m][c];
            const int row = m0 + m;
            if (row < M && col < N) {
                if (ef & EF_B0) v += pre.a[0];
                if (!gate) {
                    if (ef & EF_B1) v += pre.a[1];
                    if (ef & EF_B2) v += pre.a[2];
                    if (ef & EF_ACC) v += pre.a[4 + e];
                    if (ef & EF_RELU) v = fmaxf(v, 0.f);
                    stf(hd.C, (unsigned)row * ldc4 + cb, v);
                } else {       // (sub_modules.py:42-47): g = dropout(relu(.)) -> C ; y = g*t + t
                    const float g = fmaxf(v, 0.f) * xg_keep(eo.drop, (uint32_t)(row * N + col));
                    stf(hd.C, (unsigned)row * ldc4 + cb, g);
                    const float tv = pre.a[4 + e];
                    stf(gate_y, (unsigned)row * ldy4 + cb, g * tv + tv);
                }
            }
        }
    } else if (hd.epi == SK_EPI_LSTMB) {
        // pointwise LSTM backward of the step whose dh this product completes (same arithmetic as lstm_bwd_body)
#pragma unroll
        for (int e = 0; e < 1024 / (NW * 64); ++e) {
            const int idx = threadIdx.x + NW * 64 * e;
            const int m = idx >> 5, c = idx & 31;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[w][m][c];
            const int b = m0 + m, j = n0 + c;
            if (b < M && j < N) {
                if (ef & EF_ACC) v += ldf(hd.C, ((unsigned)b * (unsigned)hd.ldc + (unsigned)j) * 4u);
                lstmb_point(job, b, j, v);
            }
        }
    } else {
        // LSTM cell epilogue: thread -> (row em, unit eu); its four gate pre-activations sit at columns eu + 8*gate
        const int em_ = threadIdx.x >> 3, eu_ = threadIdx.x & 7;
        const int b = m0 + em_, j = tn * 8 + eu_;
        if (threadIdx.x < 256 && b < M && j < R) {
            float s4[4];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                float v = ((ef & EF_B0) ? pre.a[gi] : 0.f) + ((ef & EF_B1) ? pre.a[4 + gi] : 0.f) + ((ef & EF_B2) ? pre.a[8 + gi] : 0.f) +
                          ((ef & EF_ADD) ? pre.a[12 + gi] : 0.f);
#pragma unroll
                for (int w = 0; w < NW; ++w) v += red[w][em_][gi * 8 + eu_];
                s4[gi] = v;
            }
            const float so = (ef & EF_IFOG) ? s4[2] : s4[3];
            const float sg_ = (ef & EF_IFOG) ? s4[3] : s4[2];
            const float ig = xg_sigmoid(s4[0]), fg = xg_sigmoid(s4[1]), og = xg_sigmoid(so), gg = xg_tanh(sg_);
            const float cp = pre.st[EP_CP], mk = (ef & EF_MASK) ? pre.st[EP_MK] : 1.0f;
            float cn = fg * cp + ig * gg, hn;
            if (ef & EF_HOLD) {
                cn = cn * mk + cp * (1.0f - mk);
                hn = og * xg_tanh(cn);
                hn = hn * mk + pre.st[EP_HP] * (1.0f - mk);
            } else {
                hn = og * xg_tanh(cn) * mk;
                cn = cn * mk;
            }
            hn *= xg_keep(eo.drop, (uint32_t)(b * R + j));
            const unsigned j4 = (unsigned)j * 4u, r4 = (unsigned)R * 4u;
            if (eo.gates) {
                const unsigned go = (unsigned)b * ((unsigned)eo.ldg * 4u) + j4;
                stf(eo.gates, go, ig); stf(eo.gates, go + r4, fg);
                stf(eo.gates, go + 2 * r4, (ef & EF_IFOG) ? og : gg);
                stf(eo.gates, go + 3 * r4, (ef & EF_IFOG) ? gg : og);
            }
            stf(eo.c_out, (unsigned)b * ((unsigned)eo.ldco * 4u) + j4, cn);
            stf(eo.h_out, (unsigned)b * ((unsigned)eo.ldho * 4u) + j4, hn);
        }
    }
}

// NW waves split K: 8 for launches of at most two tiles per CU (two waves per SIMD hide each other's LDS / load latency);
// 4 (256-thread workgroups, four per CU) for launches that carry ATTN jobs or more tiles than that, so that every tile of
// the launch is resident at once.
// PREC 1 (XgRun.gemm_mode 1): the packed tiles are bf16 ([i(2)][h(2)][n(32)][8], two 1 KB wave loads), the A chunk is
// rounded to bf16 on its way into LDS and a 32-deep chunk is two v_mfma_f32_32x32x16_bf16; accumulation / epilogues fp32.
// SCALE: the launch has a scaled operand (SkSeg.row_scale: cell 2 reading the unnormalised attention context).  Its few
// extra live values push the kernel over 128 VGPRs into scratch (48 B per lane, ~6 MB written and read back per launch), so
// only the launches that need it pay for it.
// (SCALE launches are single 256-tile cell-2 launches, one workgroup per CU: they get the 256-VGPR budget of two waves per
// SIMD instead of spilling at 128.)
// (Operands two chunks ahead of their MFMAs -- three B register sets, two A sets, two waves per SIMD -- were measured in rounds 3
// and 5 and are gone: stand-alone step unchanged, iteration 1-3 % slower beside the background products, docs/EXPERIMENTS.md.)
// SEL (round 6, rollout steps): the launch's SELECT job -- the POS gate -- first chooses the tokens of its tile's 32 rows from the
// previous step's vocabulary statistics (one wave per row: xg_select.h), then gathers its embedding rows by them; the n-tile 0
// workgroups also do the rows' bookkeeping.  Every n-tile of an m-tile repeats the choice (same data, same code: same tokens) --
// 128 KB of L2 reads per workgroup instead of a launch of its own in front of the step.
template <int NW, int PREC, bool SCALE, bool SEL = false>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(SCALE ? 2 : 4, SCALE ? 2 : 4)))
skf_kernel(SkArgs args) {
    XG_CHAIN_PRIO();
    SK_STAMP(0);
    // per wave: the staged activation chunk (fp32 image, one bf16 image, or three bf16 plane images), later the wave's partial tile
    constexpr bool SPLIT3 = PREC == 2 || PREC == 3;      // split-bf16: weight planes split in registers (2) or pre-split in memory (3)
    constexpr int WSM = SPLIT3 ? (3 * PLH) / 2 : (32 * RSF > OPF ? 32 * RSF : OPF);      // floats per wave
    static_assert(WSM >= 32 * RSF, "the reduction buffer must fit the staging area");
    constexpr int LGNW = NW == 8 ? 3 : 2;
    static_assert(NW == 8 || NW == 4, "waves per workgroup");
    __shared__ __attribute__((aligned(16))) float smem[NW * WSM];
    const SkJob& job = args.job[blockIdx.y];
    const char* jb = reinterpret_cast<const char*>(&job);
    // ---- first round of wide scalar loads: the head, the first segment's hot part and the epilogue's input block
    SkHeadBlk hd = *reinterpret_cast<const SkHeadBlk*>(jb);
    SkSegHot sg = *reinterpret_cast<const SkSegHot*>(jb + offsetof(SkJob, seg));
    SkEpiIn ei = *reinterpret_cast<const SkEpiIn*>(jb + offsetof(SkJob, bias));
    RollSelectArgs sel;
    if constexpr (SEL) { sel = args.sel; sk_hold(sel); }       // (same round of scalar loads as the job's own blocks)
    sk_hold(hd); sk_hold(sg); sk_hold(ei);
#ifdef SKF_PAD_SALU   // measurement build: what N more scalar instructions in every wave's prologue cost a launch
#define SKF_STR2(x) #x
#define SKF_STR(x) SKF_STR2(x)
    asm volatile(".rept " SKF_STR(SKF_PAD_SALU) "\n\ts_mov_b32 %0, %0\n\t.endr" : "+s"(hd.nck_all));
#endif
    int ef = (ei.bias[0] ? EF_B0 : 0) | (ei.bias[1] ? EF_B1 : 0) | (ei.bias[2] ? EF_B2 : 0) | (ei.add ? EF_ADD : 0) | (ei.mask ? EF_MASK : 0) |
                   (ei.accumulate ? EF_ACC : 0) | (ei.relu ? EF_RELU : 0) | (ei.order == XG_ORDER_IFOG ? EF_IFOG : 0) |
                   (ei.mask_mode == XG_MASK_HOLD ? EF_HOLD : 0);
    asm volatile("" : "+s"(ef));       // (ONE scalar tested bit by bit: left alone the compiler keeps a 64-bit lane mask per condition)
    if (hd.epi == SK_EPI_ZERO) {
        if ((size_t)blockIdx.x * 4096 < (size_t)job.M * job.N) zero_tile<NW>(job, blockIdx.x);
        return;
    }
    if (hd.epi == SK_EPI_COPY) {
        if ((size_t)blockIdx.x * 4096 < (size_t)job.M * job.N) copy_tile<NW>(job, blockIdx.x);
        return;
    }
    if (hd.epi == SK_EPI_ATTN) {
        if ((int)blockIdx.x < 2 * job.M) {           // (every wave of the workgroup scores its share of the rows)
            if (job.attn_A <= 1536) attn_part<6, NW>(job, blockIdx.x, smem); else attn_part<8, NW>(job, blockIdx.x, smem);
        }
        return;
    }
    if ((int)blockIdx.x >= hd.ntiles) return;
    if (hd.hflags & SKH_LOW_PRIO) __builtin_amdgcn_s_setprio(0);
    const int ntm = hd.ntm, ks = hd.ksplit;      // (ksplit >= 1: xgk_skinny)
    const bool cell_tiles = (hd.hflags & SKH_CELL_TILES) != 0;
    const int lg_ks = (hd.hflags >> SKH_LGKS_SHIFT) & 15;      // (the cross-workgroup split is a power of two)
    // A split launch's parts are whole ranges of block indices, part 0 first: the workgroups of the LAST part -- the ones that wait
    // for the others' tiles (sk_epilogue_split) -- are dispatched behind every workgroup they wait for.
    const int nt1 = hd.ntiles >> lg_ks;                        // tiles of one part
    int bid = blockIdx.x, kp = 0;
    if (ks > 1) { kp = bid / nt1; bid -= kp * nt1; }
    {   // XCD-aware: the m-tiles that share a weight slice stay on one XCD's L2 (block b runs on XCD b % 8)
        const int q = nt1 >> 3, r = nt1 & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm, tn;
    if (hd.hflags & SKH_POW2_NTM) {       // (every hot launch: 1, 2 or 4 m-tiles) -- no integer division in the decode
        const int lg_ntm = (hd.hflags >> SKH_LGNTM_SHIFT) & 15;
        tm = bid & (ntm - 1); tn = bid >> lg_ntm;
    } else {
        tm = bid % ntm; tn = bid / ntm;
    }
    const int m0 = tm * 32, n0 = tn * 32;
    // (the wave index is the same for every lane: saying so keeps the chunk ranges, loop counters and branches of the K loop on the
    //  scalar unit instead of exec-masked vector control flow)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, l31 = lane & 31;
    float* As = smem + wave * WSM;
    const int lrow = lane >> 3, lcol = (lane & 7) << 2;
    // this lane's rows of the m-tile (clamped: rows beyond M are computed from a valid row and never stored)
    int rowi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowi[i] = min(m0 + i * 8 + lrow, hd.M - 1);

    // ---- the gathered segment's row indices (embedding lookup inside the product): one more dependent round trip in front of
    // that segment's operands, so they are requested first of all.  (The low dword of the int64 token only: a negative token is a
    // negative int, and the clamp keeps the row inside the table whatever the value.)
    const int gseg = ((hd.hflags >> SKH_GATHER_SHIFT) & 3) - 1;
    int gidx[4] = {0, 0, 0, 0};
    bool chosen = false;
    if constexpr (SEL) {
        if (hd.hflags & SKH_SELECT) {                         // (job-uniform: every wave of the workgroup meets the barrier)
            __shared__ int tok_s[32];
            // four rows per wave and pass, 16 lanes each (xg_select.h): NW = 8 -> one pass over the tile's 32 rows, NW = 4 -> two
#pragma unroll
            for (int r0 = 0; r0 < 32; r0 += 4 * NW) {
                const int r = r0 + wave * 4 + (lane >> 4), b = m0 + r;
                const int tk = roll_select_rows16(sel, b < hd.M ? b : -1, tn == 0 && kp == 0);
                if ((lane & 15) == 0) tok_s[r] = tk;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) gidx[i] = tok_s[rowi[i] - m0];
            chosen = true;
        }
    }
    if (gseg >= 0 && !chosen) {
        const int64_t* gp = job.seg[gseg].gather;
#pragma unroll
        for (int i = 0; i < 4; ++i) gidx[i] = __float_as_int(ldf(reinterpret_cast<const float*>(gp), (unsigned)rowi[i] * 8u));
    }
    SK_STAMP(6);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // (split-bf16: the plane registers leave no room for 19 prefetched values across the K loop at 128 VGPRs -- they would go
    //  to scratch, which costs more than it hides -- so that mode requests the cell operands behind the loop, under the reduction)
    constexpr bool LATE_PRE = SPLIT3 && !SCALE;
    EpiPre pre;
#pragma unroll
    for (int i = 0; i < 16; ++i) pre.a[i] = 0.f;
    pre.st[EP_CP] = 0.f; pre.st[EP_HP] = 0.f; pre.st[EP_MK] = 1.f;
    // A scaled operand (the unnormalised attention context): every lane holds the reciprocal scales of its four rows of the
    // m-tile (requested up front, consumed when the wave reaches the scaled segment -- no LDS image and no workgroup barrier: a
    // wave's chunks are one contiguous range of the concatenated segments, so some waves START in the scaled segment and a
    // barrier there would make them wait for the others' whole share).  The unnormalised attention weights of this m-tile's
    // videos are normalised by ALL its n-tiles, a slice each: loads now, multiply + store at the very end.
    int scaled_seg = -1;
    if (SCALE) {
#pragma unroll
        for (int s = 2; s >= 0; --s) if (s < hd.nseg && job.seg[s].row_scale) scaled_seg = s;
    }
    float exv = 0.f, exs = 1.f, rs[4] = {1.f, 1.f, 1.f, 1.f};
    float* exp_ = nullptr;
    auto scale_loads = [&]() {          // consumed at the scaled segment / at the very end
        if (scaled_seg >= 0) {
            const SkSeg& sgc = job.seg[scaled_seg];
#pragma unroll
            for (int i = 0; i < 4; ++i) rs[i] = ldf(sgc.row_scale, (unsigned)rowi[i] * 4u);
            if (sgc.ex) {
                // this tile's slice of the m-tile's 32 x ex_K weights; ex_per = elements per tile and ex_magic = ceil(2^32 / ex_K) come
                // from the launcher: idx / ex_K as one multiply-high (exact for idx < 2^16), no integer division in the kernel
                const int per = sgc.ex_per;
                const int idx = tn * per + (int)threadIdx.x;
                const int r = (int)__umulhi((unsigned)idx, (unsigned)sgc.ex_magic);
                const int row = m0 + r;
                if ((int)threadIdx.x < per && idx < 32 * sgc.ex_K && row < hd.M) {
                    exp_ = sgc.ex + (size_t)row * sgc.ex_ld + (idx - r * sgc.ex_K);
                    exv = *exp_;
                    exs = sgc.row_scale[row];
                }
            }
        }
    };
    if (SCALE) scale_loads();
    // ---- the epilogue's operands are requested now: they have the whole K loop to land
    if (!LATE_PRE) epi_prefetch<NW>(pre, hd, ei, ef, cell_tiles, m0, n0, tn);
    SkEpiOut eo;
    // this lane's 16-byte piece of a B tile, sub-piece i at + i * 1024 bytes (a bf16 tile is 2 KB: 2 pieces, an fp32 tile 4 KB: 4)
    // (PREC 3: a tile is six 1 KB pieces -- block j, plane q at piece 3 j + q -- of three pre-split bf16 planes: xg_pack.hip, dtype 2)
    constexpr int TILEB = PREC == 1 ? 2048 : (PREC == 3 ? 6144 : 4096), NPB = PREC == 1 ? 2 : (PREC == 3 ? 6 : 4);
    const unsigned vB = (unsigned)(half * 32 + l31) * 16u;
    // A wave's share of the reduction is ONE contiguous range of 32-deep chunks of the concatenated segments (round 5; before,
    // every wave took a slice of every segment: a wave paid the segment set-up and an exposed first-operand round trip per
    // segment -- with 2-3 segments of 2 chunks each that was most of its K loop).  Most waves now touch one segment only.
    const int P0 = (kp * hd.nck_all) >> lg_ks, PN = (((kp + 1) * hd.nck_all) >> lg_ks) - P0;      // this workgroup's part (cross-workgroup split)
    const int w0 = P0 + ((wave * PN) >> LGNW), w1 = P0 + (((wave + 1) * PN) >> LGNW);
    int seg_start = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s >= hd.nseg) break;
        const int nc = sg.nck;
        const int c0 = max(w0, seg_start) - seg_start, c1 = min(w1, seg_start + nc) - seg_start;
        seg_start += nc;
        const bool have = c0 < c1;
        // operands through buffer descriptors: per-lane byte offsets fixed for the segment, the chunk is the scalar offset
        // (A: the operand's true extent -- the 16-byte pieces of a k tail that reach past the END of the operand's last row come back
        //  as zeros from the descriptor's range check instead of being read; B: whole packed tiles)
        const __amdgpu_buffer_rsrc_t rB = sk_rsrc(sg.Bp), rA = sk_rsrc(sg.A, sg.a_bytes);
        const unsigned lda4 = (unsigned)sg.lda * 4u;
        unsigned vA[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = rowi[i];
            if (sg.gather) {
                int t;
                if (s == gseg) t = gidx[i];
                else t = __float_as_int(ldf(reinterpret_cast<const float*>(sg.gather), (unsigned)row * 8u));
                row = t < 0 ? 0 : (t > sg.gather_max ? sg.gather_max : t);
            }
            vA[i] = __umul24((unsigned)row, lda4) + (unsigned)lcol * 4u;       // (rows and byte pitches are below 2^24: checked by the launcher)
        }
        const int nfull = sg.K >> 5;                 // chunks >= nfull are the (single) k tail: K % 4 == 0 on this path
        // unnormalised attention context as an operand: rows scaled by 1 / s while they are staged; the tn == 0 tiles write
        // the normalised rows back (scaled_out has A's row pitch: checked by the host)
        const bool sc_seg = SCALE && (sg.sflags & SKS_SCALED);
        float rsi[4] = {1.f, 1.f, 1.f, 1.f};
        if (sc_seg) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rsi[i] = xg_rcp(rs[i]);          // (v_rcp_f32, 1 ulp: an IEEE division is ~12 instructions per row)
        }
        const bool wb_seg = SCALE && (sg.sflags & SKS_WRITEBACK);      // chunk c of the scaled rows is written back by n-tile c % ntn
        __amdgpu_buffer_rsrc_t rW = rA;
        if (wb_seg) rW = sk_rsrc(job.seg[s].scaled_out, sg.a_bytes);
        // Operands are requested one chunk ahead of their MFMAs: ping-pong B sets, one A set (reloaded right behind its LDS store).
        f32x4 ra0[4], rb0[NPB], rb1[NPB];
        if (s == 0) SK_STAMP(1);
        const unsigned sB0 = (unsigned)(tn * nc) * (unsigned)TILEB;
        auto ldB = [&](int c, f32x4 (&b)[NPB]) {
            const unsigned so = sB0 + (unsigned)c * (unsigned)TILEB;
#pragma unroll
            for (int i = 0; i < NPB; ++i) b[i] = bld16(rB, vB + i * 1024u, so);
        };
        auto ldAc = [&](int c, f32x4 (&a)[4]) {
            const unsigned so = (unsigned)c * (CK * 4u);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = bld16(rA, vA[i], so);
            if (c >= nfull) {                        // the k tail: pieces beyond K are zero
                const bool ok = sg.K - c * CK - lcol > 0;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = ok ? a[i] : z;
            }
        };
        if (have) {
            ldB(c0, rb0); ldAc(c0, ra0);
        }
        // ---- behind this segment's first operand request: the next segment's hot part, or -- behind the last one -- the
        // epilogue's output block (scalar-cache hits: the first round touched their lines' neighbours)
        SkSegHot sgn = sg;
        if (s + 1 < 3 && s + 1 < hd.nseg) {
            sgn = *reinterpret_cast<const SkSegHot*>(jb + offsetof(SkJob, seg) + (s + 1) * sizeof(SkSeg));
            sk_hold(sgn);
        } else {
            eo = *reinterpret_cast<const SkEpiOut*>(jb + offsetof(SkJob, gates));
            sk_hold(eo);
        }
        // one chunk: stage A (scaled / written back when it is the attention context), request the next chunk's operands (A into
        // the registers just stored, B into the free set: no copies), then the 16 MFMAs of this chunk
        auto chunk = [&](int c, f32x4 (&ra)[4], const f32x4 (&cur)[NPB], f32x4 (&nxt)[NPB]) {
            if (sc_seg) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ra[i] *= rsi[i];
                    if (wb_seg && c == tn && m0 + i * 8 + lrow < hd.M && c * CK + lcol < sg.K) bst16(rW, vA[i], (unsigned)c * (CK * 4u), ra[i]);
                }
            }
            if (PREC == 1) st_chunk_bf16(reinterpret_cast<unsigned short*>(As), lane, ra);
            else if (SPLIT3) st_chunk_split3(reinterpret_cast<unsigned short*>(As), lane, ra);
            else st_chunk(As, lane, ra);
            if (s == 0 && c == c0) SK_STAMP(2);
            if (c + 1 < c1) { ldB(c + 1, nxt); ldAc(c + 1, ra); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (SPLIT3) {
                // lane (column l31, half h) holds the weights of k = 16 h + 4 i + kk in piece i: block j = pieces 2j, 2j + 1 =
                // k in [16 h + 8 j, + 8), and the activation fragment is read at the same k (any k order does, if both sides agree)
                const unsigned short* Ah = reinterpret_cast<const unsigned short*>(As);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bf16x8 bp3[3], ap3[3];
                    if constexpr (PREC == 3) {
#pragma unroll
                        for (int q = 0; q < 3; ++q) bp3[q] = __builtin_bit_cast(bf16x8, cur[3 * j + q]);     // the planes as packed: no VALU on the weight side
                    } else {
                        split3_b(cur[2 * j], cur[2 * j + 1], bp3);
                    }
#pragma unroll
                    for (int q = 0; q < 3; ++q) ap3[q] = *reinterpret_cast<const bf16x8*>(Ah + q * PLH + l31 * LDH + half * 16 + j * 8);
                    // smallest terms first
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap3[0], bp3[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap3[2], bp3[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap3[1], bp3[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap3[0], bp3[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap3[1], bp3[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap3[0], bp3[0], acc, 0, 0, 0);
                }
            } else if (PREC == 1) {
                const unsigned short* Ah = reinterpret_cast<const unsigned short*>(As);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ah + l31 * LDH + i * 16 + half * 8);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, cur[i]), acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(As + l31 * LDR + half * 16 + i * 4);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], cur[i][kk], acc, 0, 0, 0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        if (have) {
            for (int c = c0; c < c1; c += 2) {
                chunk(c, ra0, rb0, rb1);
                if (c + 1 < c1) chunk(c + 1, ra0, rb1, rb0);
            }
        }
        sg = sgn;
    }
    SK_STAMP(3);
    SK_STAMP_MAX(7);
    if (LATE_PRE) epi_prefetch<NW>(pre, hd, ei, ef, cell_tiles, m0, n0, tn);
    __syncthreads();
    float (*red)[32][RSF] = reinterpret_cast<float (*)[32][RSF]>(smem);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
    __syncthreads();
    SK_STAMP(4);
    if (ks > 1) sk_epilogue_split<RSF, NW>(job, smem, m0, n0, kp, tm + ntm * tn);
    else skf_epilogue<NW>(job, hd, ef, ei.gate_y, ei.ldy, eo, cell_tiles, smem, m0, n0, tn, pre);
    if (exp_) *exp_ = exv * xg_rcp(exs);
    SK_STAMP(5);
}

}  // namespace

// Generic route for shapes the tile layout cannot take (n-contiguous B with N % 4 != 0): plain tiled GEMMs.
static int skinny_fallback(hipStream_t st, const SkJob& jb, int gemm_mode) {
    if (jb.epi != SK_EPI_STORE) return XG_EINVAL;     // LSTM / GATE jobs never use an n-contiguous B
    for (int s = 0; s < jb.nseg; ++s) {
        const SkSeg& sg = jb.seg[s];
        const bool last = s == jb.nseg - 1;
        XG_TRY(xgk_gemm(st, gemm_mode, false, !sg.b_ncontig, jb.M, jb.N, sg.K, sg.A, sg.lda, sg.B, sg.ldb, jb.C, jb.ldc,
                        s == 0 ? jb.bias[0] : nullptr, last && jb.relu && !jb.bias[1] && !jb.bias[2],
                        s > 0 || jb.accumulate));
    }
    if (jb.bias[1] || jb.bias[2]) return XG_EINVAL;
    return XG_OK;
}

#ifdef SK_TRACE
extern "C" int xg_debug_sk_trace(long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sk_trace_buf), sizeof(long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
extern "C" int xg_debug_sk_trace_filter3(int njobs, int gx, int threads) {
    const int sel[3] = {njobs, gx, threads};
    return hipMemcpyToSymbol(HIP_SYMBOL(sk_trace_sel), sel, sizeof(sel)) == hipSuccess ? 0 : -1;
}
extern "C" int xg_debug_sk_trace_filter(int njobs, int gx) {
    const int sel[3] = {njobs, gx, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(sk_trace_sel), sel, sizeof(sel)) == hipSuccess ? 0 : -1;
}
extern "C" int xg_debug_sk_trace_clear(void) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(sk_trace_buf)) != hipSuccess) return -1;
    return hipMemset(p, 0, sizeof(long long) * 4096 * 8) == hipSuccess ? 0 : -1;
}
#endif

int xgk_skinny(hipStream_t st, SkArgs& a, int gemm_mode) {
    const bool planes = (gemm_mode & XGK_SK_PLANES) != 0;        // split-bf16 with pre-split weight planes in the packed tiles (dtype 2)
    gemm_mode &= ~XGK_SK_PLANES;
    if (a.njobs <= 0 || a.njobs > SK_MAX_JOBS) return XG_EINVAL;
    {   // Jobs are independent, so their order is free: most tiles first.  The grid is (widest job) x (jobs), dispatched job by job;
        // a narrower job's surplus workgroups find out that they are surplus only after their first round of descriptor loads
        // (~1 us) and hold a workgroup slot until then -- in front of a wider job they delayed its tiles by 1.5-3.4 us (in-kernel
        // stamps, profiles/r05_sk_trace_*.txt); behind it they delay nothing.
        auto job_tiles = [](const SkJob& jb) -> long {
            if (jb.epi == SK_EPI_ZERO || jb.epi == SK_EPI_COPY) return ((long)jb.M * jb.N + 4095) / 4096;
            if (jb.epi == SK_EPI_ATTN) return 2L * jb.M;
            const long ntm = xg_cdiv(jb.M, 32);
            return ntm * ((jb.epi == SK_EPI_LSTM || jb.cell_cols) ? jb.R / 8 : xg_cdiv(jb.N, 32));
        };
        // (a SELECT job goes first whatever its width: its tiles carry the launch's longest prologue)
        auto before = [&](const SkJob& x, const SkJob& y) { return x.select != y.select ? x.select > y.select : job_tiles(x) > job_tiles(y); };
        for (int i = 1; i < a.njobs; ++i)
            for (int j = i; j > 0 && before(a.job[j], a.job[j - 1]); --j) { const SkJob t = a.job[j]; a.job[j] = a.job[j - 1]; a.job[j - 1] = t; }
    }
    bool vec = true, generic = false, packed = true, special = false;
    int tiles = 0, max_tiles = 0, max_k = 0;
    for (int j = 0; j < a.njobs; ++j) {
        SkJob& jb = a.job[j];
        const bool no_segs = jb.epi == SK_EPI_ZERO || jb.epi == SK_EPI_ATTN || jb.epi == SK_EPI_COPY;
        if (jb.M <= 0 || (!no_segs && (jb.N <= 0 || jb.nseg < 1 || jb.nseg > 3))) return XG_EINVAL;
        if (jb.epi == SK_EPI_LSTMB && (jb.N != jb.R || !jb.gates || !jb.c_prev || !jb.c_out || !jb.ds || !jb.dc_prev ||
                                       (jb.accumulate && !jb.C))) return XG_EINVAL;
        a.tile0[j] = tiles;
        if (no_segs) {                                              // fast-kernel-only job types, no matrix segments
            int nt;
            if (jb.epi == SK_EPI_ZERO || jb.epi == SK_EPI_COPY) {
                if (!jb.C || ((uintptr_t)jb.C % 16) || jb.N <= 0) return XG_EINVAL;
                if (jb.epi == SK_EPI_COPY && (!jb.seg[0].A || ((uintptr_t)jb.seg[0].A % 16))) return XG_EINVAL;
                nt = (int)(((size_t)jb.M * jb.N + 4095) / 4096);
            } else {
                if (!jb.attn_p || !jb.attn_q || !jb.attn_v || !jb.attn_w || !jb.attn_ex || !jb.attn_s || !jb.attn_c) return XG_EINVAL;
                if (jb.attn_A % 4 || jb.attn_A > 256 * 8 || jb.attn_K < 1 || jb.attn_K > 128 || jb.R < 2 || jb.R % 2) return XG_EINVAL;
                if ((((uintptr_t)jb.attn_p | (uintptr_t)jb.attn_q | (uintptr_t)jb.attn_w) % 16) || ((uintptr_t)jb.attn_v % 8)) return XG_EINVAL;
                nt = 2 * jb.M;
            }
            special = true;
            if (jb.epi == SK_EPI_ATTN) tiles += nt;              // (ZERO tiles are over at once: they do not count as occupancy)
            max_tiles = nt > max_tiles ? nt : max_tiles;
            continue;
        }
        const int ntm = xg_cdiv(jb.M, 32);
        int ntn;
        if (jb.cell_cols && jb.epi != SK_EPI_STORE) return XG_EINVAL;
        if (jb.epi == SK_EPI_LSTM || jb.cell_cols) {
            if (jb.N != 4 * jb.R || jb.R % 8 != 0) return XG_EINVAL;
            ntn = jb.R / 8;
        } else {
            ntn = xg_cdiv(jb.N, 32);
        }
        tiles += ntm * ntn;
        max_tiles = ntm * ntn > max_tiles ? ntm * ntn : max_tiles;
        {   // the fast kernel's epilogues address their operands as 32-bit byte offsets row * pitch * 4 + column
            const int pitches[] = {jb.ldc, jb.ldadd, jb.ldcp, jb.ldhp, jb.ldg, jb.ldco, jb.ldho, jb.ldt, jb.ldy, jb.ldds, jb.lddci, jb.lddcp, jb.lddhh, jb.ldm};
            for (int q : pitches) packed = packed && (double)jb.M * (q > 0 ? q : 0) * 4.0 < 4294967295.0;
        }
        for (int s = 0; s < jb.nseg; ++s) {
            SkSeg& sg = jb.seg[s];
            if (!sg.A || !sg.B || sg.K <= 0) return XG_EINVAL;
            max_k = sg.K > max_k ? sg.K : max_k;
            packed = packed && sg.Bp && sg.nck == xg_cdiv(sg.K, 32);
            // (the fast kernel's 32-bit / 24-bit address arithmetic: row index and byte pitch below 2^24, operands below 2 GB)
            packed = packed && jb.M < (1 << 24) && sg.lda < (1 << 22) && (!sg.gather || sg.gather_max < (1 << 24)) &&
                     (double)(sg.gather ? sg.gather_max + 1 : jb.M) * sg.lda * 4.0 < 2147483647.0;
            if (packed) sg.a_bytes = (int)((((int64_t)(sg.gather ? sg.gather_max + 1 : jb.M) - 1) * sg.lda + sg.K) * 4);
            if (sg.row_scale && (!sg.Bp || sg.gather || (sg.scaled_out && (sg.ld_out != sg.lda || ((uintptr_t)sg.scaled_out % 16))))) return XG_EINVAL;
            if (sg.gather && !sg.Bp) return XG_EINVAL;        // the row gather exists on the packed path only
            vec = vec && ((uintptr_t)sg.A % 16 == 0) && (sg.lda % 4 == 0) && (sg.K % 4 == 0);
            vec = vec && ((uintptr_t)sg.B % 16 == 0) && (sg.ldb % 4 == 0);
            if (sg.b_ncontig && (jb.N % 4 != 0 || jb.N < 4)) generic = true;
        }
    }
    if (generic) {
        for (int j = 0; j < a.njobs; ++j) XG_TRY(skinny_fallback(st, a.job[j], gemm_mode));
        return XG_OK;
    }
    const bool bf16 = gemm_mode == 1;      // plain-bf16 mode covers the recurrent products too
    const bool bf16x3 = gemm_mode == 3;    // split-bf16: fp32 packed tiles, three bf16 planes, 6 MFMAs per 16-deep block (fast kernel only)
    static const bool no_packed = xg_diag_env("XG_NO_PACKED") != nullptr;
    const bool fast = vec && packed && !no_packed;       // (the caller attaches bf16 tiles iff gemm_mode is 1: attach_packed)
    // cross-workgroup split-K for launches that would leave most CUs idle (every job must allow it)
    int ks = 1;
    {
        static const bool no_split = xg_diag_env("XG_NO_SPLITK") != nullptr;
        bool ok = fast && !special && !no_split;
        int min_chunks = 1 << 30;
        for (int j = 0; j < a.njobs && ok; ++j) {
            const SkJob& jb = a.job[j];
            ok = jb.ksplit_ok && jb.C && jb.accumulate && !jb.relu && (jb.epi == SK_EPI_STORE || (jb.epi == SK_EPI_LSTMB && jb.tickets)) &&
                 xg_cdiv(jb.M, 32) * xg_cdiv(jb.N, 32) <= 1024;
            for (int s = 0; s < jb.nseg; ++s) min_chunks = jb.seg[s].nck < min_chunks ? jb.seg[s].nck : min_chunks;
        }
        int cap = 8;
        for (int j = 0; j < a.njobs; ++j) if (a.job[j].ksplit_cap > 0 && a.job[j].ksplit_cap < cap) cap = a.job[j].ksplit_cap;
        // Split until the launch has one workgroup per CU; beyond that (up to two per CU) only while every part keeps a deep
        // reduction.  (Round 4 filled 512 slots whenever it could; with the lean kernel of round 5 a part's fixed cost is
        // instruction issue and the ticket phase, not latency, and FEWER, longer parts win at hidden 512 -- encoder backward
        // and the reverse-time launches 4 / 4 / 8 -> 2 / 2 / 4 parts: 5.54-5.56 -> 5.50-5.51 ms, SCST 6.04 -> 6.00 -- while the
        // 128-chunk reductions of hidden 1024 still want their two parts.)
        constexpr int target = 256, deep = 64;
        // (an LSTMB job's parts hand their tiles over through XGK_SKPART_TILES 8 KB slabs of scratch per job: SkJob.tickets)
        for (int j = 0; j < a.njobs && ok; ++j)
            if (a.job[j].epi == SK_EPI_LSTMB) {
                const int jt = xg_cdiv(a.job[j].M, 32) * xg_cdiv(a.job[j].N, 32);
                if (cap > 4) cap = 4;                            // (the finishing part holds every other part's granules at once)
                while (cap > 1 && jt * (cap - 1) > XGK_SKPART_TILES) cap >>= 1;
            }
        if (ok) while (ks < cap && min_chunks / (ks * 2) >= 4 &&
                       (tiles * ks * 2 <= target || (tiles * ks * 2 <= 512 && min_chunks / ks >= deep))) ks *= 2;
        for (int j = 0; j < a.njobs; ++j) a.job[j].ksplit = ks;
        tiles *= ks; max_tiles *= ks;
    }
    // derived fields of the job heads (what skf_kernel reads with its first round of scalar loads)
    for (int j = 0; j < a.njobs; ++j) {
        SkJob& jb = a.job[j];
        const bool cells = jb.epi == SK_EPI_LSTM || jb.cell_cols;
        jb.ntm = xg_cdiv(jb.M, 32);
        jb.ntn = cells ? jb.R / 8 : xg_cdiv(jb.N, 32);
        jb.ntiles = jb.ntm * jb.ntn * ks;
        int gseg = 0;
        bool any_scaled = false;
        jb.nck_all = 0;
        for (int q = jb.nseg - 1; q >= 0; --q) {
            SkSeg& sg = jb.seg[q];
            jb.nck_all += sg.nck;
            if (sg.ex && sg.ex_K > 0) {
                if (32 * sg.ex_K >= 65536) return XG_EINVAL;
                sg.ex_per = (32 * sg.ex_K + jb.ntn - 1) / jb.ntn;
                sg.ex_magic = (int)(unsigned)((0x100000000ull + (unsigned)sg.ex_K - 1) / (unsigned)sg.ex_K);
            }
            // (the n-tile that writes chunk c of the normalised rows back is tile c: there must be one)
            if (sg.row_scale && sg.scaled_out && sg.Bp && jb.ntn < sg.nck) return XG_EINVAL;
            sg.sflags = (sg.row_scale ? SKS_SCALED : 0) | (sg.row_scale && sg.scaled_out ? SKS_WRITEBACK : 0) | (sg.row_scale && sg.ex ? SKS_EX : 0);
            any_scaled = any_scaled || sg.row_scale;
            if (sg.gather) gseg = q + 1;
        }
        int lg_ntm = 0, lg_ks = 0;
        while ((1 << lg_ntm) < jb.ntm) ++lg_ntm;
        while ((1 << lg_ks) < ks) ++lg_ks;
        const bool pow2 = (1 << lg_ntm) == jb.ntm;
        if (jb.select && (gseg != 1 || jb.nseg != 1)) return XG_EINVAL;       // the chosen tokens feed the job's one (gathered) segment
        jb.hflags = (cells ? SKH_CELL_TILES : 0) | (jb.low_prio ? SKH_LOW_PRIO : 0) | (any_scaled ? SKH_HAS_SCALED : 0) | (gseg << SKH_GATHER_SHIFT) |
                    (jb.select ? SKH_SELECT : 0) |
                    (pow2 ? SKH_POW2_NTM : 0) | (lg_ntm << SKH_LGNTM_SHIFT) | (lg_ks << SKH_LGKS_SHIFT);
    }
    if (special && !fast) return XG_EINVAL;            // ZERO / ATTN jobs and scaled operands exist in the fast kernel only
    static const bool split_jobs = xg_diag_env("XG_SPLIT_JOBS") != nullptr;       // diagnosis: one launch per job
    static const bool split_all = xg_diag_env("XG_SPLIT_JOBS") && xg_diag_env("XG_SPLIT_JOBS")[0] == '2';   // ... ATTN / ZERO / COPY jobs too
    if (fast && split_jobs && a.njobs > 1 && (!special || split_all)) {
        for (int j = 0; j < a.njobs; ++j) {
            SkArgs one{};
            one.njobs = 1; one.job[0] = a.job[j];
            XG_TRY(xgk_skinny(st, one, gemm_mode | (planes ? XGK_SK_PLANES : 0)));
        }
        return XG_OK;
    }
    if (fast) {
        static const int force_nw = xg_diag_env("XG_SK_NW") ? atoi(xg_diag_env("XG_SK_NW")) : 0;      // diagnosis
        // A launch that carries the attention: when everything fits the chip in one round as 8-wave workgroups (<= 512) the
        // products keep their 8-way K split and the attention's rows are scored by all eight waves (36.7 vs 38.6 us per step at
        // 64 rows); beyond that 4-wave workgroups for all, which are all resident (49.4 vs 52.5 us at 128 rows).
        // (bf16 tiles, hidden 1024: 4-wave workgroups throughout -- half the register footprint per workgroup is easier to place
        //  beside the bf16 GEMMs of the other streams: 8.20 -> 8.14 ms; at hidden 512 the rule above stands: fp32 6.08 vs 6.09,
        //  bf16 4.59 vs 4.66 ms)
        // (split launches: four-wave workgroups)
        const bool nw4_rule = ks > 1 || (force_nw ? force_nw == 4 : ((bf16 && max_k >= 1024) || tiles > 2 * 256));
        const dim3 grid((max_tiles + 7) & ~7, a.njobs);
        bool scaled = false;
        for (int j = 0; j < a.njobs; ++j)
            for (int q = 0; q < a.job[j].nseg; ++q) scaled = scaled || a.job[j].seg[q].row_scale != nullptr;
        bool has_select = false;
        for (int j = 0; j < a.njobs; ++j) has_select = has_select || a.job[j].select != 0;
        if (has_select) {            // the token choice in front of the gate tiles: exact-fp32 launches without a scaled operand
            if (bf16 || bf16x3 || scaled || ks > 1 || a.sel.ntiles < 1 || a.sel.ntiles > 16 * RSW_PER || a.sel.tw > 128 || !a.sel.part || a.sel.r.E % 4) return XG_EINVAL;
            if (nw4_rule) hipLaunchKernelGGL((skf_kernel<4, 0, false, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((skf_kernel<8, 0, false, true>), grid, dim3(512), 0, st, a);
            XG_CHECK_LAUNCH();
            return XG_OK;
        }
        const bool nw4 = nw4_rule;
#define XG_SKF2(NW_, PREC_) do { \
            if (scaled) hipLaunchKernelGGL((skf_kernel<NW_, PREC_, true>), grid, dim3(NW_ * 64), 0, st, a); \
            else hipLaunchKernelGGL((skf_kernel<NW_, PREC_, false>), grid, dim3(NW_ * 64), 0, st, a); } while (0)
#define XG_SKF(PREC_) do { if (nw4) XG_SKF2(4, PREC_); else XG_SKF2(8, PREC_); } while (0)
        if (bf16) XG_SKF(1); else if (bf16x3 && planes) XG_SKF(3); else if (bf16x3) XG_SKF(2); else XG_SKF(0);
#undef XG_SKF
#undef XG_SKF2
        XG_CHECK_LAUNCH();
        return XG_OK;
    }
    for (int j = 0; j < a.njobs; ++j)
        for (int s = 0; s < a.job[j].nseg; ++s) if (a.job[j].seg[s].gather) return XG_EINVAL;
    if (vec && bf16) hipLaunchKernelGGL((sk_kernel<true, 1>), dim3(tiles), dim3(SKT), 0, st, a);
    else if (vec) hipLaunchKernelGGL((sk_kernel<true, 0>), dim3(tiles), dim3(SKT), 0, st, a);
    else if (bf16) hipLaunchKernelGGL((sk_kernel<false, 1>), dim3(tiles), dim3(SKT), 0, st, a);
    else hipLaunchKernelGGL((sk_kernel<false, 0>), dim3(tiles), dim3(SKT), 0, st, a);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
