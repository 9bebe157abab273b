// Token choice of one rollout step from the per-tile row statistics of the vocabulary product (xg_heads.hip: vocab_part16_kernel /
// vocab_part_kernel), shared by its two homes:
//   * roll_select_kernel (xg_heads.hip): one workgroup per row, its own launch -- the last choice of a rollout, and every choice
//     when the step's first launch cannot take it;
//   * skf_kernel's SELECT prologue (xg_step.hip, round 6): one WAVE per row inside the POS-gate tiles of the NEXT step's first
//     launch (roll_select_wave below) -- the choice then costs no launch of its own: a rollout step is four dependent launches
//     instead of five (SAModel.py:183-215 folded into sub_modules.py:682's product).
// Reference: caption_src/SAModel.py:183-215 (sample_max / multinomial branch, unfinished bookkeeping, early exit).
#pragma once
#include "xg_common.h"

struct RollStepArgs {
    const float* logits;      // (B,V) raw logits of step t-1, null at t = 0
    const float* uniforms;    // (B) for SAMPLE
    const int64_t* forced;    // element b at forced[b * fstride], for REPLAY
    int64_t fstride;
    const float* unf_prev;    // (B) unfinished after step t-1 (t >= 2)
    const float* table;       // embedding (V,E)
    int64_t* tok; float* tok_logp; float* unf; float* lse;      // (B) each, step-t slices
    int64_t* seq; float* seq_logp;                               // (B,Tm1)
    int32_t* maxf;            // running max over rows of the step at which the row finished (T if it never does)
    float* xt;                // (B,E) out
    float temperature;
    int V, E, t, T, mode;
    int split;                // rows >= split of a SAMPLE rollout decode greedily and report into maxf[1] (paired SCST rollout)
};
struct RollSelectArgs {
    RollStepArgs r;                       // (r.logits = the rows stored by the vocabulary product; r.t >= 1)
    const float* part; int ntiles;        // (B, rs_pitch(ntiles), 4): max, sum exp(x - max), sum exp((x - max) / T), argmax column (int bits)
    int tw;                               // columns per tile statistic (32: vocab_part_kernel, 80: vocab_part16_kernel)
};
// Layout of a row's tile statistics: tile j sits in slot (j % per) * 16 + j / per, per = ceil(ntiles / 16) -- the 16 lanes of a DPP
// row, lane l owning the CONTIGUOUS tiles [l per, l per + per) (column order, as the draw needs), then read 16 adjacent 16-byte slots
// per load instruction (one 256-byte run per token row) instead of 16 slots 16 per bytes apart.  Row pitch = 16 per slots.
__host__ __device__ __forceinline__ int rs_per(int ntiles) { return (ntiles + 15) >> 4; }
__host__ __device__ __forceinline__ int rs_pitch(int ntiles) { return rs_per(ntiles) << 4; }
__host__ __device__ __forceinline__ int rs_slot(int j, int per) { return (j % per) * 16 + j / per; }
static_assert(sizeof(RollSelectArgs) <= 176, "RollSelectArgs rides in SkArgs (kernel-argument budget, xg_kernels.h)");

// the per-row bookkeeping of a rollout step once its token is chosen (one thread): unfinished &= it > 0 ; it *= unfinished ;
// append (SAModel.py:200-210), the running "first step at which every row is finished" (:211-215)
__device__ __forceinline__ void roll_bookkeep(const RollStepArgs& a, int b, int mode, int64_t tk, float lp, float lse, int32_t* maxf) {
    const float u = (a.t == 1 ? 1.0f : a.unf_prev[b]) * (tk > 0 ? 1.0f : 0.0f);
    if (mode != XG_ROLLOUT_REPLAY) {
        const bool was = a.t == 1 ? true : a.unf_prev[b] > 0.f;
        if (was && u == 0.f) atomicMax(maxf, a.t);                 // this row finishes at step t
        else if (u > 0.f && a.t == a.T - 1) atomicMax(maxf, a.T);  // never finished
    } else if (a.t == a.T - 1) {
        atomicMax(maxf, a.T);
    }
    a.unf[b] = u;
    a.lse[b] = lse;
    a.tok[b] = tk;                                                   // xt = embed(it) uses the raw draw (:198)
    a.tok_logp[b] = lp;
    a.seq[(size_t)b * (a.T - 1) + (a.t - 1)] = mode == XG_ROLLOUT_REPLAY ? tk : (u > 0.f ? tk : 0);
    a.seq_logp[(size_t)b * (a.T - 1) + (a.t - 1)] = lp;
}

// ---- DPP helpers over a ROW of 16 lanes (one token row per DPP row: four rows per wave, no LDS shuffles)
template <int CTRL>
__device__ __forceinline__ int rs_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ float rs_dpp_f(float v) { return __int_as_float(rs_dpp_i<CTRL>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ double rs_dpp_d(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)rs_dpp_i<CTRL>((int)(unsigned)u), hi = (unsigned)rs_dpp_i<CTRL>((int)(unsigned)(u >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// all-reduce within the row: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float rs_row_max(float v) {
    v = fmaxf(v, rs_dpp_f<0xB1>(v)); v = fmaxf(v, rs_dpp_f<0x4E>(v)); v = fmaxf(v, rs_dpp_f<0x141>(v)); return fmaxf(v, rs_dpp_f<0x140>(v));
}
__device__ __forceinline__ int rs_row_min(int v) {
    v = min(v, rs_dpp_i<0xB1>(v)); v = min(v, rs_dpp_i<0x4E>(v)); v = min(v, rs_dpp_i<0x141>(v)); return min(v, rs_dpp_i<0x140>(v));
}
__device__ __forceinline__ int rs_row_max(int v) {
    v = max(v, rs_dpp_i<0xB1>(v)); v = max(v, rs_dpp_i<0x4E>(v)); v = max(v, rs_dpp_i<0x141>(v)); return max(v, rs_dpp_i<0x140>(v));
}
__device__ __forceinline__ float rs_row_sum(float v) {
    v += rs_dpp_f<0xB1>(v); v += rs_dpp_f<0x4E>(v); v += rs_dpp_f<0x141>(v); return v + rs_dpp_f<0x140>(v);
}
__device__ __forceinline__ double rs_row_sum(double v) {
    v += rs_dpp_d<0xB1>(v); v += rs_dpp_d<0x4E>(v); v += rs_dpp_d<0x141>(v); return v + rs_dpp_d<0x140>(v);
}
// inclusive scan within the row: row_shr 1, 2, 4, 8 (zeros shift in)
__device__ __forceinline__ double rs_row_scan(double v) {
    v += rs_dpp_d<0x111>(v); v += rs_dpp_d<0x112>(v); v += rs_dpp_d<0x114>(v); return v + rs_dpp_d<0x118>(v);
}
__device__ __forceinline__ float rs_row_scan(float v) {
    v += rs_dpp_f<0x111>(v); v += rs_dpp_f<0x112>(v); v += rs_dpp_f<0x114>(v); return v + rs_dpp_f<0x118>(v);
}

// One WAVE chooses the tokens of FOUR rows at once, 16 lanes (one DPP row) per token row: every reduction and scan is a handful
// of DPP moves inside the row, the four rows ride the same instructions, and the wave pays the two dependent memory round trips
// (statistics; the drawn tile) once.  (First form, one wave per row with 64-lane shuffles: 24 ds_bpermute chains per row -- the
// prologue cost 15 us, more than the launch it replaced.)  Lane l of a row owns the contiguous tiles [l per, l per + per), per =
// ceil(ntiles / 16) <= RSW_PER: tile order = column order, as the inverse-CDF draw needs.  Same arithmetic as roll_select_kernel up
// to the grouping of the sums.  b = the lane's token row (the same for the 16 lanes of a DPP row), < 0: absent.  Returns the row's
// token in every lane of the row.  write: the rows' bookkeeping too.
constexpr int RSW_PER = 16;          // ntiles <= 256
__device__ __forceinline__ int roll_select_rows16(const RollSelectArgs& q, int b, bool write) {
    typedef float rs_f32x4 __attribute__((ext_vector_type(4)));
    const RollStepArgs& a = q.r;
    const int lane = threadIdx.x & 63, l = lane & 15;
    const int nt = q.ntiles, per = rs_per(nt);
    const bool here = b >= 0;
    const int br = here ? b : 0;
    const bool second = br >= a.split;
    const int mode = (second && a.mode == XG_ROLLOUT_SAMPLE) ? XG_ROLLOUT_GREEDY : a.mode;
    const bool smp = mode == XG_ROLLOUT_SAMPLE, rpl = mode == XG_ROLLOUT_REPLAY;
    const float invt = 1.0f / a.temperature;
    // ---- stage 1: the row's statistics, its uniform / forced token / previous mask requested
    const rs_f32x4* pr = reinterpret_cast<const rs_f32x4*>(q.part) + (size_t)br * (per << 4) + l;
    rs_f32x4 pv[RSW_PER];
#pragma unroll
    for (int i = 0; i < RSW_PER; ++i) {
        const int j = l * per + i;
        const rs_f32x4 z = {-INFINITY, 0.f, 0.f, 0.f};
        pv[i] = (i < per && j < nt) ? pr[i << 4] : z;           // slot (i, l): rs_slot(j, per)
    }
    const float uni = smp ? a.uniforms[br] : 0.f;
    const float unf_prev = a.t == 1 ? 1.0f : a.unf_prev[br];
    int64_t forced = 0;
    if (rpl) forced = a.forced[(size_t)br * a.fstride];
    // ---- stage 2: maximum / argmax, log-sum-exp, the draw's tile
    float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < RSW_PER; ++i) {
        const int c = __float_as_int(pv[i][3]);
        if (pv[i][0] > best || (pv[i][0] == best && c < bi)) { best = pv[i][0]; bi = c; }     // (absent tiles: -inf, never taken)
    }
    const float mx = rs_row_max(best);
    bi = rs_row_min(best == mx ? bi : 0x7fffffff);
    const bool any_smp = __any(smp);
    float c1 = 0.f; double ct = 0.0;
#pragma unroll
    for (int i = 0; i < RSW_PER; ++i) {
        const float d = pv[i][0] - mx;                           // (-inf for absent tiles: exp -> 0, times the zero sums)
        c1 += pv[i][1] * __expf(d);
        if (any_smp) {
            const float wt = smp ? pv[i][2] * __expf(d * invt) : 0.f;
            pv[i][2] = wt;
            ct += (double)wt;
        }
    }
    const float lse = mx + logf(rs_row_sum(c1));
    int tk = bi; float xtk = mx;
    int v0 = 0; float base = 0.f, tf = 0.f;
    if (any_smp) {
        // inverse CDF over w_v = exp((logit_v - max) / temperature) (:190-194): inclusive scan of the lanes' sums in double inside
        // the row; the first lane whose running sum passes the target walks its tiles' sums to the tile that holds it
        const double inc = rs_row_scan(ct);
        const double tot = rs_row_sum(ct);
        const double target = (double)uni * tot;
        const unsigned long long pass = __ballot(inc > target);
        const unsigned mine = (unsigned)(pass >> (lane & 48)) & 0xFFFFu;
        const int owner = mine ? __ffs((int)mine) - 1 : 15;
        double run = inc - ct;
        int jt = min(nt, l * per + per) - 1;                     // (rounding: the last tile of the share if nothing passes)
        if (jt < l * per) jt = nt - 1;
        bool found = false;
#pragma unroll
        for (int i = 0; i < RSW_PER; ++i) {
            if (!found && i < per && l * per + i < nt) {
                if (run + (double)pv[i][2] > target) { jt = l * per + i; found = true; }
                else run += (double)pv[i][2];
            }
        }
        const bool own = l == owner;
        jt = rs_row_max(own ? jt : -1);
        base = (float)rs_row_sum(own ? run : 0.0); tf = (float)target;
        v0 = jt * q.tw;
    }
    // ---- stage 3: the drawn tile's logits (tw <= 128: 8 per lane, column order) / the replayed token's logit requested
    constexpr int XP = 8;
    float x[XP];
    const int nvalid = min(q.tw, a.V - v0);
    const int xper = (q.tw + 15) >> 4;                           // 5 for 80-column tiles, 2 for 32
    if (any_smp) {
        const float* xrow = a.logits + (size_t)br * a.V + v0;
#pragma unroll
        for (int k = 0; k < XP; ++k) { const int c = l * xper + k; x[k] = (smp && k < xper && c < nvalid) ? xrow[c] : -INFINITY; }
    }
    float xr = 0.f;
    if (rpl) {
        forced = forced < 0 ? 0 : (forced >= a.V ? a.V - 1 : forced);
        tk = (int)forced;
        xr = a.logits[(size_t)br * a.V + tk];
    }
    // ---- stage 4: the draw inside its tile
    if (any_smp) {
        float e[XP], es = 0.f;
#pragma unroll
        for (int k = 0; k < XP; ++k) { e[k] = (smp && k < xper && l * xper + k < nvalid) ? __expf((x[k] - mx) * invt) : 0.f; es += e[k]; }
        const float inc = rs_row_scan(es);
        const unsigned long long pass = __ballot(smp && base + inc > tf);
        const unsigned mine = (unsigned)(pass >> (lane & 48)) & 0xFFFFu;
        // (nothing passes -- rounding at the very end of the distribution --: the tile's last valid column)
        const int owner = mine ? __ffs((int)mine) - 1 : min(15, (nvalid - 1) / xper);
        float runf = base + (inc - es);
        int pk = min(xper, nvalid - l * xper) - 1; if (pk < 0) pk = 0;
        bool found = false;
#pragma unroll
        for (int k = 0; k < XP; ++k) {
            if (!found && k < xper && l * xper + k < nvalid) {
                runf += e[k];
                if (runf > tf) { pk = k; found = true; }
            }
        }
        float xp = x[0];
#pragma unroll
        for (int k = 1; k < XP; ++k) xp = pk == k ? x[k] : xp;
        const bool own = smp && l == owner;
        const int tks = rs_row_max(own ? v0 + l * xper + pk : -1);
        const float xps = rs_row_max(own ? xp : -INFINITY);
        if (smp) { tk = tks; xtk = xps; }
    }
    if (rpl) xtk = xr;
    if (write) {
        if (here && l == 0) {
            // (roll_bookkeep's arithmetic with the previous mask already in a register: requested in stage 1)
            const float u = unf_prev * (tk > 0 ? 1.0f : 0.0f);
            int32_t* maxf = a.maxf + (second ? 1 : 0);
            if (mode != XG_ROLLOUT_REPLAY) {
                const bool was = unf_prev > 0.f;
                if (was && u == 0.f) atomicMax(maxf, a.t);
                else if (u > 0.f && a.t == a.T - 1) atomicMax(maxf, a.T);
            } else if (a.t == a.T - 1) {
                atomicMax(maxf, a.T);
            }
            const float lp = xtk - lse;
            a.unf[br] = u; a.lse[br] = lse; a.tok[br] = (int64_t)tk; a.tok_logp[br] = lp;
            a.seq[(size_t)br * (a.T - 1) + (a.t - 1)] = mode == XG_ROLLOUT_REPLAY ? (int64_t)tk : (u > 0.f ? (int64_t)tk : 0);
            a.seq_logp[(size_t)br * (a.T - 1) + (a.t - 1)] = lp;
        }
        // (the embedding rows the backward reads are gathered once per rollout, behind its last step: rollout_impl)
    }
    return tk;
}
