// One decoder step as ONE launch: a dataflow kernel (reference caption_src/sub_modules.py:671-687, cell :750-770, gate :42-47).
//
// The step has three all-to-all seams -- the attention needs the whole query row p = h2a([h1;h2]), cell 1 needs the gated POS
// feature, cell 2 needs cell 1's h1' and the attention's context -- and as three launches (xg_model.hip: core_step, packed
// form) every seam costs a kernel boundary plus the ramp of the next launch (descriptor loads, cold first operands, wave skew):
// about 6 us each, 18 of 46 us per step.  A grid barrier costs as much on this chip (MI355X_MICROARCH.md: barrier-xcd 4.1-4.8 us
// at one workgroup per CU).  But none of the seams is really chip-wide: a consumer tile of m-tile `tm` (32 videos) needs the
// producer tiles of THAT m-tile only, and most of every consumer's work does not depend on the seam at all (cell 2: the h2 W_h2h
// product; cell 1: the h1 / xt products).  So the step is ONE grid whose workgroups are the step's work items in dependency
// order
//     [ POS-gate tiles | p tiles | attention (one workgroup per video) | cell-1 tiles | cell-2 tiles ]
// and a consumer waits -- in the middle of its K loop, with its independent segments already accumulated and the next
// segment's weight tiles already requested -- on a per-m-tile arrival counter of its producers:
//     cell 1 :  h1 W_h2h + xt W_i2h (+ bias)        | wait gate[tm] |  pos' W_a2h               -> LSTM epilogue -> c1', h1'
//     attn b :                                       | wait p[tm]    |  scores, softmax, context  -> alpha, af
//     cell 2 :  h2 W_h2h (+ bias)                    | wait c1[tm]   |  h1' W_i2h | wait att[tm] | af W_a2h -> epilogue -> c2', h2'
// Forward progress needs no co-residency: the hardware dispatches workgroups in blockIdx order, every wait targets producers
// with LOWER block indices, so whatever is resident and waiting has its producers resident or finished -- also when another
// kernel (a background product on a side stream) holds half of every CU.  Every spin is bounded all the same (a timeout
// raises a flag word and the workgroup carries on: wrong results, never a hung GPU).
//
// Hand-off protocol (cdna_hip_programming.md, Guideline 16, R1): producers store what another workgroup will read with
// write-through (sc1) stores, every storing wave drains (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane adds
// 1 to the m-tile's counter (relaxed, agent scope); a consumer polls that ONE word from one lane (relaxed sc1 loads + s_sleep),
// the workgroup meets at a barrier, and the produced rows are read with sc1 loads (L2-served: this CU's L1 is never consulted).
// The sync words live in the caller's workspace, are zero on entry and are left at zero by the last workgroup to finish.
#include "xg_common.h"
#include "xg_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int CK = 32;            // k-chunk depth staged per wave
constexpr int LDR = CK + 4;       // fp32 A image row stride (floats): conflict-free b128 fragment reads
constexpr int LDH = CK + 8;       // bf16 A image row stride (halfwords)
constexpr int OPF = 32 * LDR;     // floats of one staged A chunk
constexpr int RSF = 40;           // reduction buffer row stride

// sync words (ints) in DStepArgs.ctr: [kind][m-tile], m-tiles <= DS_MAXTM
constexpr int DS_MAXTM = 32;
enum { DC_P = 0, DC_GATE = 1, DC_C1 = 2, DC_ATT = 3, DC_C2 = 4, DC_DONE = 5 * DS_MAXTM, DC_ERR = DC_DONE + 1, DC_WORDS = DC_DONE + 2 };

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// -DDS_TRACE: per-workgroup time stamps (wall_clock64, 100 MHz) of the LAST launch: [start, first wait passed, second wait
// passed, K loops done, end], fetched with xg_debug_ds_trace (tools/dstep_trace.py)
#ifdef DS_TRACE
__device__ long long ds_trace_buf[4096 * 8];
#define DS_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) ds_trace_buf[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define DS_STAMP(i) do {} while (0)
#endif

__device__ __forceinline__ void st_pub(float* p, float v) { __hip_atomic_store(p, v, RLX_AGENT); }     // global_store_dword sc1

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const float* P) {
    const uint64_t a = reinterpret_cast<uint64_t>(P);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, -1, 0x00020000);
}
// 16-byte load of floats [off, off + 4) of the buffer, bypassing this CU's L1 (sc1): rows another workgroup of this launch wrote
__device__ __forceinline__ f32x4 ld_sc1(const __amdgpu_buffer_rsrc_t& rs, unsigned off_floats) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off_floats * 4u), 0, 16));
}

// 16-byte write-through (sc1) store of floats [off, off + 4) of the buffer: what another workgroup of this launch will read.
// (4-byte sc1 stores are one fabric write each: a 4 KB tile written that way took 5-6 us to drain, measured.)
__device__ __forceinline__ void st_sc1(const __amdgpu_buffer_rsrc_t& rs, unsigned off_floats, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), rs,
                                           (int)(off_floats * 4u), 0, 16);
}

// the workgroup waits until *ctr >= target (one lane polls; bounded)
__device__ __forceinline__ void wg_wait(int* ctr, int target, int* err) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(ctr, RLX_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1 << 21)) { __hip_atomic_store(err, 1, RLX_AGENT); break; }      // ~1 s: give up, flag, carry on
        }
    }
    __syncthreads();
}
// everything this workgroup stored for other workgroups is out (every storing wave drains), then ONE lane signals
__device__ __forceinline__ void wg_signal(int* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1, RLX_AGENT);
}

__device__ __forceinline__ void st_chunk(float* __restrict__ lds, int lane, const f32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(lds + (i * 8 + (lane >> 3)) * LDR + ((lane & 7) << 2)) = v[i];
}
__device__ __forceinline__ void st_chunk_bf16(unsigned short* __restrict__ lds, int lane, const f32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));      // v_cvt_pk_bf16_f32: RNE
        bf16x2_t lo, hi;
        lo[0] = (__bf16)v[i][0]; lo[1] = (__bf16)v[i][1]; hi[0] = (__bf16)v[i][2]; hi[1] = (__bf16)v[i][3];
        uint2 pk;
        pk.x = __builtin_bit_cast(unsigned, lo); pk.y = __builtin_bit_cast(unsigned, hi);
        *reinterpret_cast<uint2*>(lds + (i * 8 + (lane >> 3)) * LDH + ((lane & 7) << 2)) = pk;
    }
}

// One K segment of a 32 x 32 tile: acc += A[m0 .. m0+31, 0..K) * W_tile(tn)^T, this wave's share of the segment's 32-deep
// chunks.  A (M,K) row-major (optionally row-gathered), W as packed tiles (xg_pack.hip): B goes global -> VGPR (ping-pong
// register sets), A through this wave's private LDS image.  SC1: A is read with sc1 buffer loads (rows written by other
// workgroups of this launch).  Same arithmetic and chunk order as xg_step.hip: skf_kernel.
struct SegOp {
    const float* A; int lda, K;
    const float* Bp; int nck;
    const int64_t* gather; int gather_max;         // row m of the operand is A + clamp(gather[m]) * lda
};
template <int NW, int PREC, bool SC1>
struct SegMac {
    static constexpr int TILE = PREC == 1 ? 512 : 1024, NPB = PREC == 1 ? 2 : 4;
    f32x4 rb0[NPB];
    const float* bp;
    int c0, c1;
    // request the first chunk's weight tile (nothing it needs depends on other workgroups: issued BEFORE a wait)
    __device__ __forceinline__ void begin(const SegOp& s, int tn, int wave, int lane) {
        c0 = (wave * s.nck) / NW; c1 = ((wave + 1) * s.nck) / NW;
        bp = s.Bp + ((size_t)tn * s.nck) * TILE + (size_t)lane * 4;
        if (c0 < c1) {
#pragma unroll
            for (int i = 0; i < NPB; ++i) rb0[i] = *reinterpret_cast<const f32x4*>(bp + (size_t)c0 * TILE + i * 256);
        }
    }
    __device__ __forceinline__ void run(const SegOp& s, f32x16& acc, float* As, int m0, int M, int lane) {
        if (c0 >= c1) return;
        const int lrow = lane >> 3, lcol = (lane & 7) << 2, half = lane >> 5, l31 = lane & 31;
        unsigned aoff[4];
        const float* ap[4];
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(s.A);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = min(m0 + i * 8 + lrow, M - 1);
            if (s.gather) {
                const int64_t t = s.gather[row];
                row = (int)(t < 0 ? 0 : (t > s.gather_max ? s.gather_max : t));
            }
            aoff[i] = (unsigned)row * (unsigned)s.lda + (unsigned)lcol;
            ap[i] = s.A + (size_t)row * s.lda + lcol;
        }
        const int nfull = s.K / CK;
        f32x4 ra[4], rb1[NPB];
        auto ldA = [&](int c) {
            const bool tail = c >= nfull;
            const int kleft = s.K - c * CK - lcol;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 t;
                if (SC1) t = ld_sc1(rs, aoff[i] + (unsigned)c * CK);
                else t = *reinterpret_cast<const f32x4*>(ap[i] + (size_t)c * CK);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                ra[i] = (tail && kleft <= 0) ? z : t;
            }
        };
        ldA(c0);
        auto chunk = [&](int c, const f32x4 (&cur)[NPB], f32x4 (&nxt)[NPB]) {
            if (PREC == 1) st_chunk_bf16(reinterpret_cast<unsigned short*>(As), lane, ra);
            else st_chunk(As, lane, ra);
            if (c + 1 < c1) {
#pragma unroll
                for (int i = 0; i < NPB; ++i) nxt[i] = *reinterpret_cast<const f32x4*>(bp + (size_t)(c + 1) * TILE + i * 256);
                ldA(c + 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (PREC == 1) {
                const unsigned short* Ah = reinterpret_cast<const unsigned short*>(As);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(Ah + l31 * LDH + i * 16 + half * 8);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, cur[i]), acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(As + l31 * LDR + half * 16 + i * 4);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], cur[i][kk], acc, 0, 0, 0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        for (int c = c0; c < c1; c += 2) {
            chunk(c, rb0, rb1);
            if (c + 1 < c1) chunk(c + 1, rb1, rb0);
        }
    }
};

__device__ __forceinline__ f32x16 zero16() {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.f;
    return v;
}

// the NW partial tiles -> LDS (smem re-used: every wave is past its staging image)
template <int NW>
__device__ __forceinline__ void reduce_to_lds(const f32x16& acc, float* smem, int wave, int lane) {
    __syncthreads();
    float (*red)[32][RSF] = reinterpret_cast<float (*)[32][RSF]>(smem);
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
    __syncthreads();
}

// XCD-aware tile order inside a job of `n` workgroups: the m-tiles that share a weight slice stay on one XCD's L2
__device__ __forceinline__ int xcd_order(int i, int n) {
    const int q = n / 8, r = n % 8, xcd = i % 8, idx = i / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- LSTM cell epilogue (sub_modules.py:752-770): thread -> (row em, unit eu); its four gate pre-activations sit at columns
// eu + 8 * gate of the reduced tile.  Loads are requested up front (lstm_pre), before the K loop.
struct CellPre { float b[3][4], ad[4], cp, hp, mk; int b_, j_; bool on; };
__device__ __forceinline__ CellPre cell_pre(int B, int R, int m0, int tn, const float* bias0, const float* bias1, const float* bias2,
                                            const float* add, const float* c_prev, const float* h_prev, const float* mask, int ldm) {
    CellPre p;
    const int em = threadIdx.x >> 3, eu = threadIdx.x & 7;
    p.b_ = m0 + em; p.j_ = tn * 8 + eu;
    p.on = threadIdx.x < 256 && p.b_ < B && p.j_ < R;
    p.cp = 0.f; p.hp = 0.f; p.mk = 1.f;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) { p.b[0][gi] = 0.f; p.b[1][gi] = 0.f; p.b[2][gi] = 0.f; p.ad[gi] = 0.f; }
    if (p.on) {
        const float* dummy = c_prev + (size_t)p.b_ * R + p.j_;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int col = gi * R + p.j_;
            p.b[0][gi] = *(bias0 ? bias0 + col : dummy);
            p.b[1][gi] = *(bias1 ? bias1 + col : dummy);
            p.b[2][gi] = *(bias2 ? bias2 + col : dummy);
            p.ad[gi] = *(add ? add + (size_t)p.b_ * 4 * R + col : dummy);
        }
        p.cp = *dummy;
        p.hp = h_prev[(size_t)p.b_ * R + p.j_];
        p.mk = *(mask ? mask + (size_t)p.b_ * ldm : dummy);
    }
    return p;
}
template <int NW>
__device__ __forceinline__ void cell_epilogue(const CellPre& pre, const float* smem, int R, bool b0, bool b1, bool b2, bool has_add,
                                              bool has_mask, float* gates, float* c_out, float* h_out, const XgDrop& drop) {
    const float (*red)[32][RSF] = reinterpret_cast<const float (*)[32][RSF]>(smem);
    if (threadIdx.x >= 256) return;               // (wave-uniform: waves 4.. have no epilogue work)
    const int em = threadIdx.x >> 3, eu = threadIdx.x & 7;
    float hn = 0.f;
    if (pre.on) {
    float s4[4];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
        float v = (b0 ? pre.b[0][gi] : 0.f) + (b1 ? pre.b[1][gi] : 0.f) + (b2 ? pre.b[2][gi] : 0.f) + (has_add ? pre.ad[gi] : 0.f);
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][em][gi * 8 + eu];
        s4[gi] = v;
    }
    // decoder cell: gate order i, f, o, g; masked rows HOLD their state
    const float ig = xg_sigmoid(s4[0]), fg = xg_sigmoid(s4[1]), og = xg_sigmoid(s4[2]), gg = xg_tanh(s4[3]);
    const float cp = pre.cp, mk = has_mask ? pre.mk : 1.0f;
    float cn = fg * cp + ig * gg;
    cn = cn * mk + cp * (1.0f - mk);
    hn = og * xg_tanh(cn);
    hn = hn * mk + pre.hp * (1.0f - mk);
    const int b = pre.b_, j = pre.j_;
    hn *= xg_keep(drop, (uint32_t)(b * R + j));
    if (gates) {
        float* g = gates + (size_t)b * 4 * R;
        g[j] = ig; g[R + j] = fg; g[2 * R + j] = og; g[3 * R + j] = gg;
    }
    c_out[(size_t)b * R + j] = cn;
    }
    // the row's 8 new hidden units leave as two 16-byte write-through stores: lanes eu = 0 and eu = 4 collect their neighbours'
    f32x4 h4;
    h4[0] = hn; h4[1] = __shfl_down(hn, 1); h4[2] = __shfl_down(hn, 2); h4[3] = __shfl_down(hn, 3);
    if (pre.on && (eu & 3) == 0) st_sc1(rsrc_of(h_out), (unsigned)pre.b_ * (unsigned)R + (unsigned)pre.j_, h4);
}

// ---- temporal attention of one video by one workgroup (sub_modules.py:678-680): e_k = w . tanh(p + q_k), alpha = softmax_k(e)
// over ALL K frames (unmasked), af = sum_k alpha_k V_k.  Wave w scores frames w, w + NW, ...: whole q rows (lane = 16 bytes x NI
// column groups), the NEXT frame's row requested before the current one's tanh arithmetic (two register sets).
template <int NW, int NI>                    // NI float4 groups of A per lane: A <= 256 NI
__device__ __forceinline__ void attn_video(const DStepArgs& a, int b, float* smem, int* ctr, int* err) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = a.K, A = a.A, R = a.R;
    float* se = smem;                         // [K <= 128] scores -> weights
    float* part = smem + 128;                 // [4][R] partial contexts
    const __amdgpu_buffer_rsrc_t rp = rsrc_of(a.P);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    constexpr bool DB = NI <= 6;              // two q register sets only where they fit under 128 VGPRs
    f32x4 pr[NI], wr[NI], q0[NI], q1[DB ? NI : 1];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane * 4 + 256 * i;
        wr[i] = c < A ? *reinterpret_cast<const f32x4*>(a.a2w + c) : z;        // (w = 0 beyond A: those lanes add nothing)
    }
    // q rows through a buffer descriptor: scalar row offset + one 32-bit lane offset per column group (64-bit per-group
    // addresses cost 24 VGPRs and pushed the kernel into scratch)
    const __amdgpu_buffer_rsrc_t rq = rsrc_of(a.vproj);
    int voff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) voff[i] = min(lane * 4 + 256 * i, A - 4) * 4;          // unconditional (clamped) loads stay in flight
    auto ldq = [&](auto& q, int k) {
        const int soff = __builtin_amdgcn_readfirstlane((b * K + k) * A * 4);
#pragma unroll
        for (int i = 0; i < NI; ++i) q[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, voff[i], soff, 0));
    };
    auto score = [&](const auto& q, int k) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            s += wr[i][0] * xg_tanh(pr[i][0] + q[i][0]) + wr[i][1] * xg_tanh(pr[i][1] + q[i][1]) +
                 wr[i][2] * xg_tanh(pr[i][2] + q[i][2]) + wr[i][3] * xg_tanh(pr[i][3] + q[i][3]);
        s = wave_sum(s);
        if (lane == 0) se[k] = s;
    };
    if (wave < K) ldq(q0, wave);              // the first frame's row does not depend on p: requested before the wait
    DS_STAMP(1);
    wg_wait(ctr + DC_P * DS_MAXTM + (b >> 5), (A + 31) >> 5, err);
    DS_STAMP(2);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane * 4 + 256 * i;
        pr[i] = ld_sc1(rp, (unsigned)b * (unsigned)A + (unsigned)min(c, A - 4));
    }
    if constexpr (DB) {
        for (int k = wave; k < K; k += 2 * NW) {
            if (k + NW < K) ldq(q1, k + NW);
            score(q0, k);
            if (k + NW < K) {
                if (k + 2 * NW < K) ldq(q0, k + 2 * NW);
                score(q1, k + NW);
            }
        }
    } else {
        for (int k = wave; k < K; k += NW) {
            if (k != wave) ldq(q0, k);
            score(q0, k);
        }
    }
    __syncthreads();
    DS_STAMP(3);
    if (wave == 0) {                          // softmax over all K frames (unmasked, :679), K <= 128
        const float e0 = lane < K ? se[lane] : -3.0e38f, e1 = lane + 64 < K ? se[lane + 64] : -3.0e38f;
        const float mx = wave_max(fmaxf(e0, e1));
        const float x0 = lane < K ? __expf(e0 - mx) : 0.f, x1 = lane + 64 < K ? __expf(e1 - mx) : 0.f;
        const float inv = 1.0f / wave_sum(x0 + x1);
        if (lane < K) { se[lane] = x0 * inv; if (a.alpha) a.alpha[(size_t)b * K + lane] = x0 * inv; }
        if (lane + 64 < K) { se[lane + 64] = x1 * inv; if (a.alpha) a.alpha[(size_t)b * K + lane + 64] = x1 * inv; }
    }
    __syncthreads();
    // context: thread (column group cg, frame residue kg of 4): four frames' rows in flight at a time
    const float* Vb = a.V + (size_t)b * K * R;
    const int ncg = R >> 2;                   // float4 column groups (R % 8 == 0)
    constexpr int NT = NW * 64, NCG = NT / 4;
    for (int base = 0; base < ncg; base += NCG) {
        const int cg = min(base + (int)(threadIdx.x % NCG), ncg - 1), kg = threadIdx.x / NCG;
        f32x4 s = z;
        for (int k = kg; k < K; k += 16) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(Vb + (size_t)min(k + 4 * u, K - 1) * R + cg * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) s += (k + 4 * u < K ? se[k + 4 * u] : 0.f) * v[u];
        }
        if (base + (int)(threadIdx.x % NCG) < ncg) *reinterpret_cast<f32x4*>(part + (size_t)kg * R + cg * 4) = s;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(a.af);
    for (int c4 = threadIdx.x; c4 < ncg; c4 += NT) {
        const f32x4 v = (*reinterpret_cast<const f32x4*>(part + c4 * 4) + *reinterpret_cast<const f32x4*>(part + R + c4 * 4)) +
                        (*reinterpret_cast<const f32x4*>(part + 2 * R + c4 * 4) + *reinterpret_cast<const f32x4*>(part + 3 * R + c4 * 4));
        st_sc1(ra, (unsigned)b * (unsigned)R + (unsigned)c4 * 4u, v);
    }
    wg_signal(ctr + DC_ATT * DS_MAXTM + (b >> 5));
}

}  // namespace

// ================================================================================================ the kernel
template <int NW, int PREC>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) dstep_kernel(DStepArgs a) {
    XG_CHAIN_PRIO();
    __shared__ __attribute__((aligned(16))) float smem[NW * 32 * RSF > NW * OPF ? NW * 32 * RSF : NW * OPF];
    const int bid = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* As = smem + wave * OPF;
    const int B = a.B, R = a.R;
    const int ntm = (B + 31) >> 5;
    int* const ctr = a.ctr;
    int* const err = a.ctr + DC_ERR;
    DS_STAMP(0);

    if (bid >= a.t0_p && bid < a.t0_att) {
        // ---------------------------------------------------------------- p = h2a([h1 ; h2]) + b                      :677
        const int t = xcd_order(bid - a.t0_p, a.t0_att - a.t0_p);
        const int tm = t % ntm, tn = t / ntm, m0 = tm * 32, n0 = tn * 32;
        const int nckR = (R + 31) >> 5;
        SegOp s0{a.h1, R, R, a.pk_h2a1, nckR, nullptr, 0}, s1{a.h2, R, R, a.pk_h2a2, nckR, nullptr, 0};
        f32x16 acc = zero16();
        SegMac<NW, PREC, false> m;
        m.begin(s0, tn, wave, lane); m.run(s0, acc, As, m0, B, lane);
        m.begin(s1, tn, wave, lane); m.run(s1, acc, As, m0, B, lane);
        DS_STAMP(3);
        reduce_to_lds<NW>(acc, smem, wave, lane);
        DS_STAMP(4);
        const float (*red)[32][RSF] = reinterpret_cast<const float (*)[32][RSF]>(smem);
        if (threadIdx.x < 256) {                 // thread -> (row, 4 adjacent columns): one 16-byte write-through store
            const int mm = threadIdx.x >> 3, c4 = (threadIdx.x & 7) << 2;
            f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][mm][c4]);
#pragma unroll
            for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(&red[w][mm][c4]);
            const int row = m0 + mm, col = n0 + c4;
            if (row < B && col < a.A) {          // (A % 4 == 0)
                v += *reinterpret_cast<const f32x4*>(a.h2a_b + col);
                st_sc1(rsrc_of(a.P), (unsigned)row * (unsigned)a.A + (unsigned)col, v);
            }
        }
        DS_STAMP(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DS_STAMP(6);
        wg_signal(ctr + DC_P * DS_MAXTM + tm);
    } else if (bid < a.t0_p) {
        // ---------------------------------------------------------------- POS gate: pos' = drop(relu(W_g xt + b)) * pos + pos   :682
        const int t = xcd_order(bid, a.t0_p);
        const int tm = t % ntm, tn = t / ntm, m0 = tm * 32, n0 = tn * 32;
        SegOp s0{a.xt ? a.xt : a.embed, a.E, a.E, a.pk_dgate, (a.E + 31) >> 5, a.xt ? nullptr : a.tok, a.V1};
        f32x16 acc = zero16();
        SegMac<NW, PREC, false> m;
        m.begin(s0, tn, wave, lane); m.run(s0, acc, As, m0, B, lane);
        reduce_to_lds<NW>(acc, smem, wave, lane);
        const float (*red)[32][RSF] = reinterpret_cast<const float (*)[32][RSF]>(smem);
        if (threadIdx.x < 256) {
            const int mm = threadIdx.x >> 3, c4 = (threadIdx.x & 7) << 2;
            f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][mm][c4]);
#pragma unroll
            for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(&red[w][mm][c4]);
            const int row = m0 + mm, col = n0 + c4;
            if (row < B && col < R) {            // (R % 8 == 0)
                v += *reinterpret_cast<const f32x4*>(a.dgate_b + col);
                const f32x4 tv = *reinterpret_cast<const f32x4*>(a.pos + (size_t)row * R + col);
                f32x4 g, y;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    g[q] = fmaxf(v[q], 0.f) * xg_keep(a.drop_gate, (uint32_t)(row * R + col + q));
                    y[q] = g[q] * tv[q] + tv[q];
                }
                *reinterpret_cast<f32x4*>(a.gp + (size_t)row * R + col) = g;
                st_sc1(rsrc_of(a.posg), (unsigned)row * (unsigned)R + (unsigned)col, y);
            }
        }
        wg_signal(ctr + DC_GATE * DS_MAXTM + tm);
    } else if (bid >= a.t0_c1 && bid < a.t0_c2) {
        // ---------------------------------------------------------------- cell 1                                            :683
        const int t = xcd_order(bid - a.t0_c1, a.t0_c2 - a.t0_c1);
        const int tm = t % ntm, tn = t / ntm, m0 = tm * 32;
        const int nckR = (R + 31) >> 5;
        const bool tf = a.pre1 != nullptr;                  // teacher forcing: token side hoisted into pre1 (bias included)
        const CellPre pre = cell_pre(B, R, m0, tn, a.l1_h2h_b, tf ? nullptr : a.l1_i2h_b, tf ? nullptr : a.l1_a2h_b, a.pre1, a.c1, a.h1,
                                     a.mask, a.ldm);
        SegOp s0{a.h1, R, R, a.pk_l1_h2h, nckR, nullptr, 0};
        f32x16 acc = zero16();
        SegMac<NW, PREC, false> m;
        m.begin(s0, tn, wave, lane); m.run(s0, acc, As, m0, B, lane);
        if (!tf) {
            SegOp s1{a.xt ? a.xt : a.embed, a.E, a.E, a.pk_l1_i2h, (a.E + 31) >> 5, a.xt ? nullptr : a.tok, a.V1};
            m.begin(s1, tn, wave, lane); m.run(s1, acc, As, m0, B, lane);
            SegOp s2{a.posg, R, R, a.pk_l1_a2h, nckR, nullptr, 0};
            SegMac<NW, PREC, true> m2;
            m2.begin(s2, tn, wave, lane);                                    // weight tile requested, then wait for the gate
            DS_STAMP(1);
            wg_wait(ctr + DC_GATE * DS_MAXTM + tm, (R + 31) >> 5, err);
            DS_STAMP(2);
            m2.run(s2, acc, As, m0, B, lane);
        }
        DS_STAMP(3);
        reduce_to_lds<NW>(acc, smem, wave, lane);
        cell_epilogue<NW>(pre, smem, R, true, !tf, !tf, tf, a.mask != nullptr, a.g1, a.c1o, a.h1w, a.drop_l1);
        wg_signal(ctr + DC_C1 * DS_MAXTM + tm);
    } else if (bid < a.t0_c1) {
        // ---------------------------------------------------------------- temporal attention of ONE video                :678-680
        attn_video<NW, 6>(a, bid - a.t0_att, smem, ctr, err);       // (A <= 1536: xgk_dstep_ok)
    } else {
        // ---------------------------------------------------------------- cell 2                                            :684
        const int t = xcd_order(bid - a.t0_c2, a.total - a.t0_c2);
        const int tm = t % ntm, tn = t / ntm, m0 = tm * 32;
        const int nckR = (R + 31) >> 5;
        const CellPre pre = cell_pre(B, R, m0, tn, a.l2_h2h_b, a.l2_i2h_b, a.l2_a2h_b, nullptr, a.c2, a.h2, a.mask, a.ldm);
        SegOp s0{a.h2, R, R, a.pk_l2_h2h, nckR, nullptr, 0};
        f32x16 acc = zero16();
        SegMac<NW, PREC, false> m;
        m.begin(s0, tn, wave, lane); m.run(s0, acc, As, m0, B, lane);
        SegOp s1{a.h1w, R, R, a.pk_l2_i2h, nckR, nullptr, 0}, s2{a.af, R, R, a.pk_l2_a2h, nckR, nullptr, 0};
        SegMac<NW, PREC, true> m2;
        m2.begin(s1, tn, wave, lane);
        DS_STAMP(1);
        wg_wait(ctr + DC_C1 * DS_MAXTM + tm, R >> 3, err);
        DS_STAMP(2);
        m2.run(s1, acc, As, m0, B, lane);
        m2.begin(s2, tn, wave, lane);
        DS_STAMP(3);
        wg_wait(ctr + DC_ATT * DS_MAXTM + tm, min(32, B - m0), err);
        DS_STAMP(4);
        m2.run(s2, acc, As, m0, B, lane);
        DS_STAMP(5);
        reduce_to_lds<NW>(acc, smem, wave, lane);
        cell_epilogue<NW>(pre, smem, R, true, true, true, false, a.mask != nullptr, a.g2, a.c2o, a.h2w, a.drop_l2);
        if (a.copy_back) {
            // state updated in place (xg_step_fwd): the new h1 / h2 rows of this m-tile go back into the state once EVERY reader
            // of the old rows is through -- the last cell-2 tile of the m-tile to arrive knows (p, cell 1 and attention of the
            // m-tile are behind the waits above, the other cell-2 tiles behind their tickets)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* flag = reinterpret_cast<int*>(smem);
            if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(ctr + DC_C2 * DS_MAXTM + tm, 1, RLX_AGENT);
            __syncthreads();
            if (*flag == (R >> 3) - 1) {
                const __amdgpu_buffer_rsrc_t r1 = rsrc_of(a.h1w), r2 = rsrc_of(a.h2w);
                const int rows = min(32, B - m0);
                for (int i = threadIdx.x; i < rows * (R >> 2); i += NW * 64) {
                    const unsigned off = (unsigned)m0 * (unsigned)R + (unsigned)i * 4u;
                    *reinterpret_cast<f32x4*>(a.h1o + off) = ld_sc1(r1, off);
                    *reinterpret_cast<f32x4*>(a.h2o + off) = ld_sc1(r2, off);
                }
            }
        }
    }
    // ---- the last workgroup to finish leaves every sync word at zero for the next launch
    __syncthreads();
    DS_STAMP(7);
    if (threadIdx.x == 0) {
        const int tkt = __hip_atomic_fetch_add(ctr + DC_DONE, 1, RLX_AGENT);
        if (tkt == a.total - 1)
            for (int i = 0; i <= DC_DONE; ++i) __hip_atomic_store(ctr + i, 0, RLX_AGENT);
    }
}

#ifdef DS_TRACE
extern "C" int xg_debug_ds_trace(long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ds_trace_buf), sizeof(long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif

static_assert(sizeof(int) * DC_WORDS <= XGK_DSTEP_SYNC_BYTES, "sync words exceed the workspace reservation");

int xgk_dstep_err_word() { return DC_ERR; }      // index of the time-out flag among the sync words (diagnosis entry point below)

bool xgk_dstep_ok(const XgDims& d) {
    return d.R % 8 == 0 && d.E % 4 == 0 && d.A % 4 == 0 && d.A <= 1536 && d.K <= 128 && (d.B + 31) / 32 <= DS_MAXTM &&
           d.R <= 2048;      // (the attention's partial contexts: 4 R floats of LDS)
}

int xgk_dstep(hipStream_t st, DStepArgs& a, int gemm_mode) {
    const int ntm = xg_cdiv(a.B, 32);
    const bool tf = a.pre1 != nullptr;
    const int n_p = ntm * xg_cdiv(a.A, 32), n_gate = tf ? 0 : ntm * xg_cdiv(a.R, 32), n_cell = ntm * (a.R / 8);
    // dispatch order = dependency order, the critical chain first: gate | p | attention | cell 1 | cell 2
    a.t0_p = n_gate;
    a.t0_att = a.t0_p + n_p;
    a.t0_c1 = a.t0_att + a.B;
    a.t0_c2 = a.t0_c1 + n_cell;
    a.total = a.t0_c2 + n_cell;
    if (!a.ctr || ((uintptr_t)a.ctr % 4)) return XG_EINVAL;
    if (gemm_mode == 1) hipLaunchKernelGGL((dstep_kernel<8, 1>), dim3(a.total), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((dstep_kernel<8, 0>), dim3(a.total), dim3(512), 0, st, a);
    XG_CHECK_LAUNCH();
    return XG_OK;
}
