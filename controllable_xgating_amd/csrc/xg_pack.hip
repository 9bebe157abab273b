// Recurrent weights re-tiled into MFMA-fragment order ("packed weights").
//
// The per-timestep products are skinny (M = batch rows <= a few hundred): every workgroup streams a 32-column slice of a
// weight matrix exactly once per launch.  Staging that slice through LDS costs more LDS-pipe time than the fp32 MFMAs
// it feeds (measured: the K loop of the LDS-staged kernel runs at 64 % MFMA-busy with NO global loads at all), so the
// weights are kept a second time in the order the matrix cores consume them: a tile of 32 output columns x 32 k is 1024
// contiguous floats [i(4)][h(2)][n(32)][4], element (n, k = 16 h + 4 i + q), and the B operand of
// v_mfma_f32_32x32x2_f32 goes global -> VGPR with four fully coalesced 1 KB wave loads per tile, no LDS, no shuffles.
// (The MFMA sums over k in any order as long as A uses the same one: lane half h owns k in [16 h, 16 h + 16).)
//
// dtype 1 = bf16 tiles (XgRun.gemm_mode 1, BASELINE.json configs[4]): the same 32 x 32 tiles, rounded to bf16 ONCE here
// (round-to-nearest-even) instead of on every pass of every step, as [i(2)][h(2)][n(32)][8]: element (n, k = 16 i + 8 h + q),
// the B operand of v_mfma_f32_32x32x16_bf16, two 1 KB wave loads per tile; half the weight bytes per step.
//
// dtype 2 = three bf16 planes per weight (XgRun.gemm_mode 3, split-bf16; round 5): x = p0 + p1 + p2 exactly (p0 = x truncated to
// bf16, p1 = (x - p0) truncated, p2 = the rest rounded: the split xg_step.hip used to do in registers on every pass of every
// step -- ~160 vector instructions per lane per 32-deep chunk, the largest single cost of those launches), as six 1 KB pieces
// per tile [j(2)][q(3)][h(2)][n(32)][8]: plane q of element (n, k = 16 h + 8 j + e).  6 bytes per weight instead of 4.
//
// The shadow is caller-owned (xg_packed_bytes / xg_pack_weights), refreshed after every optimizer step
// (reference update: caption_src/starttrain.py:136-137) and handed to the entry points through XgRun.packed.
//   NT entries (forward, y = x W^T, W (N,K) row-major):  h2a[:, :R], h2a[:, R:], decoder gate, lstm_1.{i2h,a2h,h2h},
//       lstm_2.{i2h,a2h,h2h}, encoder W_hh x2.  The 4R-row cell matrices use the cell tiling: tile tn holds hidden units
//       8 tn .. 8 tn + 7 of all four gates (row r = 8 gate + unit), so one tile carries a unit's whole cell update.
//   NN entries (backward, dx = dy W, W (Kc,N) row-major): lstm_2.{a2h,h2h,i2h}, h2a[:, R:], h2a[:, :R], lstm_1.h2h, encoder W_hh x2.
#include "xg_common.h"
#include "xg_kernels.h"

namespace {

struct PackDesc { const float* src; int N, K, sn, sk, cell_R; };   // src(n,k) = src[n*sn + k*sk]; cell_R > 0: cell tiling

void describe(const XgDims& d, const XgParams& p, PackDesc* e) {
    const int R = d.R, A = d.A, E = d.E;
    e[PK_H2A1] = {p.h2a_w, A, R, 2 * R, 1, 0};
    e[PK_H2A2] = {p.h2a_w + R, A, R, 2 * R, 1, 0};
    e[PK_DGATE] = {p.dgate_w, R, E, E, 1, 0};
    e[PK_L1_I2H] = {p.l1_i2h_w, 4 * R, E, E, 1, R};
    e[PK_L1_A2H] = {p.l1_a2h_w, 4 * R, R, R, 1, R};
    e[PK_L1_H2H] = {p.l1_h2h_w, 4 * R, R, R, 1, R};
    e[PK_L2_I2H] = {p.l2_i2h_w, 4 * R, R, R, 1, R};
    e[PK_L2_A2H] = {p.l2_a2h_w, 4 * R, R, R, 1, R};
    e[PK_L2_H2H] = {p.l2_h2h_w, 4 * R, R, R, 1, R};
    e[PK_ENC_RGB] = {p.lstm_rgb_whh, 4 * R, R, R, 1, R};
    e[PK_ENC_OPFL] = {p.lstm_opfl_whh, 4 * R, R, R, 1, R};
    // data-gradient direction: output column n = input feature, reduction k = weight row
    e[PKB_L2_A2H] = {p.l2_a2h_w, R, 4 * R, 1, R, 0};
    e[PKB_L2_H2H] = {p.l2_h2h_w, R, 4 * R, 1, R, 0};
    e[PKB_H2A2] = {p.h2a_w + R, R, A, 1, 2 * R, 0};
    e[PKB_L1_H2H] = {p.l1_h2h_w, R, 4 * R, 1, R, 0};
    e[PKB_ENC_RGB] = {p.lstm_rgb_whh, R, 4 * R, 1, R, 0};
    e[PKB_ENC_OPFL] = {p.lstm_opfl_whh, R, 4 * R, 1, R, 0};
    e[PKB_L2_I2H] = {p.l2_i2h_w, R, 4 * R, 1, R, 0};
    e[PKB_H2A1] = {p.h2a_w, R, A, 1, 2 * R, 0};
}

inline size_t entry_floats(const PackDesc& e) {
    const size_t ntn = e.cell_R ? (size_t)e.cell_R / 8 : (size_t)xg_cdiv(e.N, 32);
    return ntn * (size_t)xg_cdiv(e.K, 32) * 1024;
}

struct PackArgs { PackDesc e[PK_COUNT]; float* dst[PK_COUNT]; int tile0[PK_COUNT + 1]; int first, last, bf16; };

// one workgroup (256 threads) per packed tile: reads 32 x 32 source elements (coalesced along the source's unit-stride
// dimension), writes 4 KB contiguous
__global__ void __launch_bounds__(256) pack_kernel(PackArgs a) {
    __shared__ float t[32][33];
    int ei = a.first;
    for (int i = a.first + 1; i < a.last; ++i) if ((int)blockIdx.x >= a.tile0[i]) ei = i;
    const PackDesc e = a.e[ei];
    const int tile = blockIdx.x - a.tile0[ei];
    const int nck = (e.K + 31) >> 5;
    const int tn = tile / nck, kc = tile - tn * nck;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // (a, b) = (fast, slow) source coordinates: k is unit-stride for NT entries, n for NN entries
        const int slow = ty + 8 * r, fast = tx;
        const int nn = e.sk == 1 ? slow : fast, kk = e.sk == 1 ? fast : slow;
        const int k = kc * 32 + kk;
        int n;
        bool ok;
        if (e.cell_R) { n = (nn >> 3) * e.cell_R + tn * 8 + (nn & 7); ok = true; }
        else { n = tn * 32 + nn; ok = n < e.N; }
        float v = 0.f;
        if (ok && k < e.K) v = e.src[(size_t)n * e.sn + (size_t)k * e.sk];
        t[nn][kk] = v;
    }
    __syncthreads();
    if (a.bf16 == 2) {
        unsigned short* dst = reinterpret_cast<unsigned short*>(a.dst[ei]) + (size_t)tile * 3072;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = threadIdx.x + 256 * r;                      // [j(2)][h(2)][n(32)][e(8)] of the tile's 1024 elements
            const int e = o & 7, nn = (o >> 3) & 31, h = (o >> 8) & 1, j = o >> 9;
            // the same split as xg_step.hip: split3_pair (two truncations, then round to nearest even)
            const float x = t[nn][16 * h + 8 * j + e];
            const unsigned u0 = __float_as_uint(x) & 0xFFFF0000u;
            const float r1 = x - __uint_as_float(u0);                 // exact
            const unsigned u1 = __float_as_uint(r1) & 0xFFFF0000u;
            const float r2 = r1 - __uint_as_float(u1);                // exact
            unsigned u2 = __float_as_uint(r2);
            u2 += 0x7FFFu + ((u2 >> 16) & 1u);
            const int at = h * 256 + nn * 8 + e;
            dst[(j * 3 + 0) * 512 + at] = (unsigned short)(u0 >> 16);
            dst[(j * 3 + 1) * 512 + at] = (unsigned short)(u1 >> 16);
            dst[(j * 3 + 2) * 512 + at] = (unsigned short)(u2 >> 16);
        }
        return;
    }
    if (a.bf16) {
        unsigned short* dst = reinterpret_cast<unsigned short*>(a.dst[ei]) + (size_t)tile * 1024;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = threadIdx.x + 256 * r;                      // [i(2)][h(2)][n(32)][q(8)]
            const int q = o & 7, nn = (o >> 3) & 31, h = (o >> 8) & 1, i = o >> 9;
            unsigned u = __float_as_uint(t[nn][16 * i + 8 * h + q]);
            u += 0x7FFFu + ((u >> 16) & 1u);                          // round to nearest even
            dst[o] = (unsigned short)(u >> 16);
        }
        return;
    }
    float* dst = a.dst[ei] + (size_t)tile * 1024;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = threadIdx.x + 256 * r;                          // [i(4)][h(2)][n(32)][q(4)]
        const int q = o & 3, nn = (o >> 2) & 31, h = (o >> 7) & 1, i = o >> 8;
        dst[o] = t[nn][16 * h + 4 * i + q];
    }
}

// ---- plain bf16 copies of the large products' weights (W16_*), one launch: segment table, 8 elements per thread
struct W16Desc { const float* src; size_t n; };
void describe_w16(const XgDims& d, const XgParams& p, W16Desc* e) {
    const size_t R = d.R, A = d.A, E = d.E, V = d.V;
    e[W16_LOGIT] = {p.logit_w, V * R};
    e[W16_EMB_RGB] = {p.emb_rgb_w, R * d.F1}; e[W16_EMB_OPFL] = {p.emb_opfl_w, R * d.F2};
    e[W16_WIH_RGB] = {p.lstm_rgb_wih, 4 * R * R}; e[W16_WIH_OPFL] = {p.lstm_opfl_wih, 4 * R * R};
    e[W16_GATE_RGB] = {p.gate_rgb_w, R * R}; e[W16_GATE_OPFL] = {p.gate_opfl_w, R * R};
    e[W16_FUSION] = {p.fusion_w, R * 2 * R}; e[W16_V2A] = {p.v2a_w, A * R};
    e[W16_DGATE] = {p.dgate_w, R * E}; e[W16_L1_I2H] = {p.l1_i2h_w, 4 * R * E}; e[W16_L1_A2H] = {p.l1_a2h_w, 4 * R * R};
}
inline size_t w16_bytes(size_t n) { return (n * 2 + 255) & ~(size_t)255; }
struct CvtArgs { const float* src[W16_COUNT]; unsigned short* dst[W16_COUNT]; unsigned n8[W16_COUNT]; unsigned wg0[W16_COUNT + 1]; };
__global__ void __launch_bounds__(256) cvt_multi_kernel(CvtArgs a) {
    int e = 0;
    for (int i = 1; i < W16_COUNT; ++i) if (blockIdx.x >= a.wg0[i]) e = i;
    const unsigned i8 = (blockIdx.x - a.wg0[e]) * 256u + threadIdx.x;
    if (i8 >= a.n8[e]) return;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x4_t x = *reinterpret_cast<const f32x4_t*>(a.src[e] + (size_t)i8 * 8), y = *reinterpret_cast<const f32x4_t*>(a.src[e] + (size_t)i8 * 8 + 4);
    bf16x2_t p0, p1, p2, p3;
    p0[0] = (__bf16)x[0]; p0[1] = (__bf16)x[1]; p1[0] = (__bf16)x[2]; p1[1] = (__bf16)x[3];
    p2[0] = (__bf16)y[0]; p2[1] = (__bf16)y[1]; p3[0] = (__bf16)y[2]; p3[1] = (__bf16)y[3];
    *reinterpret_cast<uint4*>(a.dst[e] + (size_t)i8 * 8) = uint4{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1),
                                                                 __builtin_bit_cast(unsigned, p2), __builtin_bit_cast(unsigned, p3)};
}

}  // namespace

size_t xgk_packed_total_bytes(const XgDims& d, int dtype) {
    size_t n = ((xgk_packed_floats(d) * (dtype == 1 ? 2 : (dtype == 2 ? 6 : 4))) + 255) & ~(size_t)255;
    if (dtype == 1) {
        XgParams p{};
        W16Desc e[W16_COUNT];
        describe_w16(d, p, e);
        for (int i = 0; i < W16_COUNT; ++i) n += w16_bytes(e[i].n);
    }
    return n;
}

size_t xgk_packed_floats(const XgDims& d) {
    XgParams p{};
    PackDesc e[PK_COUNT];
    describe(d, p, e);
    size_t n = 0;
    for (int i = 0; i < PK_COUNT; ++i) n += entry_floats(e[i]);
    return n;
}

bool xgk_packed_view(const XgDims& d, const void* packed, int dtype, PackedView* v) {
    if (!packed || d.R % 8 != 0 || ((uintptr_t)packed % 16) != 0 || dtype < 0 || dtype > 2) return false;
    XgParams p{};
    PackDesc e[PK_COUNT];
    describe(d, p, e);
    const char* base = static_cast<const char*>(packed);
    const size_t esz = dtype == 1 ? 2 : (dtype == 2 ? 6 : 4);
    size_t off = 0;
    for (int i = 0; i < PK_COUNT; ++i) {
        v->m[i] = reinterpret_cast<const float*>(base + off);
        v->nck[i] = xg_cdiv(e[i].K, 32);
        off += entry_floats(e[i]) * esz;
    }
    v->dtype = dtype;
    for (int i = 0; i < W16_COUNT; ++i) v->w16[i] = nullptr;
    if (dtype == 1 && d.V > 1 && d.F1 > 0 && d.F2 > 0) {       // (the element counts of the weight copies need the full dims)
        W16Desc we[W16_COUNT];
        describe_w16(d, p, we);
        off = (off + 255) & ~(size_t)255;
        for (int i = 0; i < W16_COUNT; ++i) {
            // 16-byte loadable rows only (the GEMM falls back to the fp32 weight otherwise: xgk_gemm_bf16x checks pitch % 8)
            v->w16[i] = reinterpret_cast<const unsigned short*>(base + off);
            off += w16_bytes(we[i].n);
        }
    }
    return true;
}

extern "C" size_t xg_packed_bytes(const XgDims* d, int dtype) {
    if (!d || d->R <= 0 || d->A <= 0 || d->E <= 0 || d->R % 8 != 0 || dtype < 0 || dtype > 2) return 0;
    if (dtype == 1 && (d->V <= 1 || d->F1 <= 0 || d->F2 <= 0)) return 0;
    return xgk_packed_total_bytes(*d, dtype);
}

static int pack_part(void* stream, const XgDims* d, const XgParams* p, void* packed, size_t packed_bytes, int dtype, int with_backward, int part);
extern "C" int xg_pack_weights(void* stream, const XgDims* d, const XgParams* p, void* packed, size_t packed_bytes,
                               int dtype, int with_backward) {
    return pack_part(stream, d, p, packed, packed_bytes, dtype, with_backward, 0);
}
// part 1: everything but the CG encoder's matrices; part 2: those only (bf16 tiles, dtype 1: the plain bf16 weight copies are
// split the same way -- the encoder's embeddings, input-side LSTM matrices, cross gates and fusion belong to part 2).  The decoder's matrices are
// final -- updated -- long before the encoder's (train.ClipAdam(overlap=True)): a caller can refresh their tiles under the
// encoder's backward and only the encoder's four tiles at the head of the next iteration.
extern "C" int xg_pack_weights_part(void* stream, const XgDims* d, const XgParams* p, void* packed, size_t packed_bytes,
                                    int dtype, int with_backward, int part) {
    if (part < 0 || part > 2) return XG_EINVAL;
    return pack_part(stream, d, p, packed, packed_bytes, dtype, with_backward, part);
}
static int pack_part(void* stream, const XgDims* d, const XgParams* p, void* packed, size_t packed_bytes, int dtype, int with_backward, int part) {
    if (!d || !p || !packed || d->R <= 0 || d->A <= 0 || d->E <= 0) return XG_EINVAL;
    if (d->R % 8 != 0 || ((uintptr_t)packed % 16) != 0 || dtype < 0 || dtype > 2) return XG_EINVAL;
    if (dtype == 1 && (d->V <= 1 || d->F1 <= 0 || d->F2 <= 0)) return XG_EINVAL;
    if (packed_bytes < xgk_packed_total_bytes(*d, dtype)) return XG_EWORKSPACE;
    PackDesc all[PK_COUNT];
    describe(*d, *p, all);
    PackedView v;
    if (!xgk_packed_view(*d, packed, dtype, &v)) return XG_EINVAL;
    PackArgs a{};
    const int last = with_backward ? PK_COUNT : PKB_L2_A2H;
    a.bf16 = dtype;                                // 0 fp32 tiles, 1 bf16 tiles, 2 three bf16 planes
    int tiles = 0, n = 0;
    for (int i = 0; i < last; ++i) {
        const bool enc = i == PK_ENC_RGB || i == PK_ENC_OPFL || i == PKB_ENC_RGB || i == PKB_ENC_OPFL;
        if ((part == 1 && enc) || (part == 2 && !enc)) continue;
        if (!all[i].src) return XG_EINVAL;
        a.e[n] = all[i];                         // (entries compacted: the kernel walks [first, last) of ITS arrays)
        a.dst[n] = const_cast<float*>(v.m[i]);
        a.tile0[n] = tiles;
        tiles += (int)(entry_floats(all[i]) / 1024);
        ++n;
    }
    a.first = 0; a.last = n;
    a.tile0[n] = tiles;
    if (tiles > 0) hipLaunchKernelGGL(pack_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
    XG_CHECK_LAUNCH();
    if (dtype == 1) {                            // the plain bf16 copies of the large products' weights
        W16Desc we[W16_COUNT];
        describe_w16(*d, *p, we);
        CvtArgs c{};
        unsigned wgs = 0;
        for (int i = 0; i < W16_COUNT; ++i) {
            if (!we[i].src || ((uintptr_t)we[i].src % 16) || we[i].n % 8) return XG_EINVAL;
            const bool enc = i == W16_EMB_RGB || i == W16_EMB_OPFL || i == W16_WIH_RGB || i == W16_WIH_OPFL || i == W16_GATE_RGB ||
                             i == W16_GATE_OPFL || i == W16_FUSION;            // two_spatial_encoder.*
            const bool skip = (part == 1 && enc) || (part == 2 && !enc);
            c.src[i] = we[i].src; c.dst[i] = const_cast<unsigned short*>(v.w16[i]); c.n8[i] = skip ? 0u : (unsigned)(we[i].n / 8);
            c.wg0[i] = wgs;
            wgs += (c.n8[i] + 255) / 256;
        }
        c.wg0[W16_COUNT] = wgs;
        if (wgs > 0) hipLaunchKernelGGL(cvt_multi_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, c);
        XG_CHECK_LAUNCH();
    }
    return XG_OK;
}
