// Tile geometry + LDS tile loaders shared by the implicit-GEMM conv (fwd/dgrad) and wgrad kernels.
#pragma once
#include "common.cuh"

// A workgroup owns TI images x TH rows x TW cols of OUTPUT pixels (power-of-two extents).
struct ConvGeom {
    int n_img;          // images in the launch
    int H, W;           // spatial dims of the tensor the halo loader reads
    int Ho, Wo;         // output-space dims (conv output; for wgrad: dims of dz)
    int ks;             // filter extent (1, 3, or 4 for the space-to-depth stem)
    int stride;         // 1 or 2 (forward stride; the zero-insert loader implies an effective stride of 1)
    int pad;            // leading pad (1 for 3x3, 0 for 1x1, 2 for the 4x4 s2d stem)
    int zins;           // 1: read the input as if zeros were inserted between its pixels (transposed stride-2)
    int tw_log2, th_log2, ti_log2;
    int tiles_x, tiles_y, n_groups;
    int hh, hw;         // halo tile extent in input pixels
};

__host__ inline void mil_geom_set(ConvGeom& g, int tw, int th, int ti);
#include <cstdlib>
__host__ inline bool mil_geom_wide() { const char* e = mil_ab_env("MIL_GEOM_WIDE"); return e && e[0] == '1'; }
__host__ inline void mil_geom_tiles(ConvGeom& g, int tile_px_log2) {
    // choose TW x TH x TI = 2^tile_px_log2 output pixels
    int tw, th;
    if (tile_px_log2 == 8) {          // 256 px
        if (g.Wo > 8 || g.Ho > 8) {
            // 16x16 tiles of one image, or 8x8 tiles of four: whichever wastes less on maps that are not a multiple of
            // 16 (the live driver's 300x300 tiles give 75/38/19/10-pixel maps).  Cost per image ~ tiles x (256 MFMA
            // pixels + staged halo pixels) / images per tile.
            const long c16 = (long)((g.Wo + 15) >> 4) * ((g.Ho + 15) >> 4) * (256 + 18 * 18);
            const long c8 = (long)((g.Wo + 7) >> 3) * ((g.Ho + 7) >> 3) * (256 + 4 * 10 * 10) / 4;
#ifndef MIL_GEOM_KS_MAX
#define MIL_GEOM_KS_MAX 3
#endif
            if (c8 < c16 && g.stride == 1 && !g.zins && g.ks <= MIL_GEOM_KS_MAX) { tw = 3; th = 3; } else { tw = 4; th = 4; }
            // experiment (MIL_GEOM_WIDE=1): 64-pixel-wide maps as 4 rows x 64 columns — a tile's rows are then ONE contiguous
            // run of the NHWC tensor (24 KB of a 24-channel fp32 map instead of sixteen 1.5 KB segments)
            if (g.Wo == 64 && g.stride == 1 && !g.zins && g.ks == 3 && mil_geom_wide()) { tw = 6; th = 2; }
        }
        else if (g.Wo > 4 || g.Ho > 4) { tw = 3; th = 3; }
        else { tw = 2; th = 2; }
    } else if (tile_px_log2 == 7) {   // 128 px
        if (g.Wo > 8 || g.Ho > 8) { tw = 4; th = 3; }
        else if (g.Wo > 4 || g.Ho > 4) { tw = 3; th = 3; }
        else { tw = 2; th = 2; }
    } else {                           // 64 px
        if (g.Wo > 4 || g.Ho > 4) { tw = 3; th = 3; }
        else { tw = 2; th = 2; }
    }
    mil_geom_set(g, tw, th, tile_px_log2 - tw - th);
}

__host__ inline void mil_geom_set(ConvGeom& g, int tw, int th, int ti) {
    g.tw_log2 = tw; g.th_log2 = th; g.ti_log2 = ti;
    g.tiles_x = (g.Wo + (1 << tw) - 1) >> tw;
    g.tiles_y = (g.Ho + (1 << th) - 1) >> th;
    g.n_groups = (g.n_img + (1 << g.ti_log2) - 1) >> g.ti_log2;
    int s = g.zins ? 1 : g.stride;
    g.hh = ((1 << th) - 1) * s + g.ks;
    g.hw = ((1 << tw) - 1) * s + g.ks;
}

struct TileOrigin { int img0, oy0, ox0; };

__device__ __forceinline__ TileOrigin mil_tile_origin(const ConvGeom& g, int tile) {
    TileOrigin o;
    int tx = tile % g.tiles_x; tile /= g.tiles_x;
    int ty = tile % g.tiles_y; int grp = tile / g.tiles_y;
    o.img0 = grp << g.ti_log2; o.oy0 = ty << g.th_log2; o.ox0 = tx << g.tw_log2;
    return o;
}

// Input halo tile -> LDS, [ti][hy][hx] pixels of CINP channels at pitch PIXB; out-of-image pixels are
// zero (the conv's zero padding).  One wave per halo row: a row's pixels are contiguous in NHWC memory.
template <typename T, int CINP>
__device__ __forceinline__ void mil_load_halo(char* lds, const typename T::elem* __restrict__ x,
                                              const ConvGeom& g, const TileOrigin& o, int tid, int nthreads) {
    constexpr int ESZ = T::ESZ;
    constexpr int PIXB = mil_pix_pitch(CINP, ESZ);
    constexpr int N16 = CINP * ESZ / 16;
    const int wave = tid >> 6, lane = tid & 63, nwaves = nthreads >> 6;
    const int s = g.zins ? 1 : g.stride;
    const int iy0 = o.oy0 * s - g.pad, ix0 = o.ox0 * s - g.pad;
    const int rows = g.hh << g.ti_log2;
    const int ppr = g.hw * N16;
    for (int row = wave; row < rows; row += nwaves) {
        const int ti = row / g.hh, hy = row - ti * g.hh;
        const int img = o.img0 + ti;
        int iy = iy0 + hy;
        bool row_ok = img < g.n_img && iy >= 0;
        if (g.zins) { row_ok = row_ok && !(iy & 1); iy >>= 1; }
        row_ok = row_ok && iy < g.H;
        const char* src_row = reinterpret_cast<const char*>(x) + ((size_t)img * g.H + iy) * g.W * (CINP * ESZ);
        char* dst_row = lds + (size_t)row * g.hw * PIXB;
        if constexpr (T::SPLIT) {               // 8 floats -> 16 bytes of the record's hi plane + 16 bytes of its lo plane
            constexpr int N8 = CINP / 8;
            for (int piece = lane; piece < g.hw * N8; piece += 64) {
                const int hx = piece / N8, j = piece - hx * N8;
                int ix = ix0 + hx;
                bool ok = row_ok && ix >= 0;
                if (g.zins) { ok = ok && !(ix & 1); ix >>= 1; }
                ok = ok && ix < g.W;
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (ok) load8<F32>(reinterpret_cast<const float*>(src_row + (size_t)ix * (CINP * 4) + j * 32), v);
                bf16x8_t hi, lo;
                mil_split8(v, hi, lo);
                *reinterpret_cast<bf16x8_t*>(dst_row + hx * PIXB + j * 16) = hi;
                *reinterpret_cast<bf16x8_t*>(dst_row + hx * PIXB + CINP * 2 + j * 16) = lo;
            }
            continue;
        }
        for (int piece = lane; piece < ppr; piece += 64) {
            const int hx = piece / N16, j = piece - hx * N16;
            int ix = ix0 + hx;
            bool ok = row_ok && ix >= 0;
            if (g.zins) { ok = ok && !(ix & 1); ix >>= 1; }
            ok = ok && ix < g.W;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (ok) v = *reinterpret_cast<const uint4*>(src_row + (size_t)ix * (CINP * ESZ) + j * 16);
            *reinterpret_cast<uint4*>(dst_row + hx * PIXB + j * 16) = v;
        }
    }
}

// Output-space tile (no halo) -> LDS, [tile pixel][CP] at pitch PIXZ; pixels outside the image are zero.
template <typename T, int CP>
__device__ __forceinline__ void mil_load_otile(char* lds, const typename T::elem* __restrict__ z,
                                               const ConvGeom& g, const TileOrigin& o, int tid, int nthreads,
                                               int tile_px) {
    constexpr int ESZ = T::ESZ;
    constexpr int PIXZ = mil_pix_pitch(CP, ESZ);
    constexpr int N16 = CP * ESZ / 16;
    const int tw_mask = (1 << g.tw_log2) - 1, th_mask = (1 << g.th_log2) - 1;
    if constexpr (T::SPLIT) {
        constexpr int N8 = CP / 8;
        for (int idx = tid; idx < tile_px * N8; idx += nthreads) {
            const int tp = idx / N8, j = idx - tp * N8;
            const int ox = o.ox0 + (tp & tw_mask);
            const int oy = o.oy0 + ((tp >> g.tw_log2) & th_mask);
            const int img = o.img0 + (tp >> (g.tw_log2 + g.th_log2));
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (img < g.n_img && oy < g.Ho && ox < g.Wo)
                load8<F32>(reinterpret_cast<const float*>(z) + (((size_t)img * g.Ho + oy) * g.Wo + ox) * CP + j * 8, v);
            bf16x8_t hi, lo;
            mil_split8(v, hi, lo);
            *reinterpret_cast<bf16x8_t*>(lds + tp * PIXZ + j * 16) = hi;
            *reinterpret_cast<bf16x8_t*>(lds + tp * PIXZ + CP * 2 + j * 16) = lo;
        }
        return;
    }
    for (int idx = tid; idx < tile_px * N16; idx += nthreads) {
        const int tp = idx / N16, j = idx - tp * N16;
        const int ox = o.ox0 + (tp & tw_mask);
        const int oy = o.oy0 + ((tp >> g.tw_log2) & th_mask);
        const int img = o.img0 + (tp >> (g.tw_log2 + g.th_log2));
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (img < g.n_img && oy < g.Ho && ox < g.Wo)
            v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(z) +
                    (((size_t)img * g.Ho + oy) * g.Wo + ox) * (CP * ESZ) + j * 16);
        *reinterpret_cast<uint4*>(lds + tp * PIXZ + j * 16) = v;
    }
}

// Byte offset of tile pixel tp's (top-left tap) position inside the halo tile.
template <int PIXB>
__device__ __forceinline__ int mil_pix_base(const ConvGeom& g, int tp, int s_eff) {
    const int tx = tp & ((1 << g.tw_log2) - 1);
    const int ty = (tp >> g.tw_log2) & ((1 << g.th_log2) - 1);
    const int ti = tp >> (g.tw_log2 + g.th_log2);
    return ((ti * g.hh + ty * s_eff) * g.hw + tx * s_eff) * PIXB;
}

// ---- split (issue-early / write-late) forms of the two loaders: global -> registers now, registers -> LDS
// after the next barrier, so a persistent workgroup keeps the NEXT tile's loads in flight while it computes.
// Flat piece index = tid + 256*i; NP is a compile-time bound on pieces per thread (guarded by the real count).
template <typename T, int CINP, int NP>
__device__ __forceinline__ void mil_halo_fetch(uint4 (&r)[NP], const typename T::elem* __restrict__ x, const ConvGeom& g,
                                               const TileOrigin& o, int tid) {
    constexpr int ESZ = T::ESZ;
    constexpr int N16 = CINP * ESZ / 16;
    const int s = g.zins ? 1 : g.stride;
    const int iy0 = o.oy0 * s - g.pad, ix0 = o.ox0 * s - g.pad;
    const int ppr = g.hw * N16;
    const int total = (g.hh << g.ti_log2) * ppr;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = tid + 256 * i;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (idx < total) {
            const int row = idx / ppr, piece = idx - row * ppr;
            const int ti = row / g.hh, hy = row - ti * g.hh;
            const int hx = piece / N16, j = piece - hx * N16;
            const int img = o.img0 + ti;
            int iy = iy0 + hy, ix = ix0 + hx;
            bool ok = img < g.n_img && iy >= 0 && ix >= 0;
            if (g.zins) { ok = ok && !((iy | ix) & 1); iy >>= 1; ix >>= 1; }
            ok = ok && iy < g.H && ix < g.W;
            if (ok) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) +
                            (((size_t)img * g.H + iy) * g.W + ix) * (CINP * ESZ) + j * 16);
        }
        r[i] = v;
    }
}

template <typename T, int CINP, int NP>
__device__ __forceinline__ void mil_halo_commit(const uint4 (&r)[NP], char* lds, const ConvGeom& g, int tid) {
    constexpr int ESZ = T::ESZ;
    constexpr int PIXB = mil_pix_pitch(CINP, ESZ);
    constexpr int N16 = CINP * ESZ / 16;
    const int total = (g.hh * g.hw << g.ti_log2) * N16;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = tid + 256 * i;
        if (idx < total) {
            const int px = idx / N16, j = idx - px * N16;
            *reinterpret_cast<uint4*>(lds + px * PIXB + j * 16) = r[i];
        }
    }
}

template <typename T, int CP, int NP>
__device__ __forceinline__ void mil_otile_fetch(uint4 (&r)[NP], const typename T::elem* __restrict__ z, const ConvGeom& g,
                                                const TileOrigin& o, int tid, int tile_px) {
    constexpr int ESZ = T::ESZ;
    constexpr int N16 = CP * ESZ / 16;
    const int tw_mask = (1 << g.tw_log2) - 1, th_mask = (1 << g.th_log2) - 1;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = tid + 256 * i;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (idx < tile_px * N16) {
            const int tp = idx / N16, j = idx - tp * N16;
            const int ox = o.ox0 + (tp & tw_mask);
            const int oy = o.oy0 + ((tp >> g.tw_log2) & th_mask);
            const int img = o.img0 + (tp >> (g.tw_log2 + g.th_log2));
            if (img < g.n_img && oy < g.Ho && ox < g.Wo)
                v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(z) +
                        (((size_t)img * g.Ho + oy) * g.Wo + ox) * (CP * ESZ) + j * 16);
        }
        r[i] = v;
    }
}

template <typename T, int CP, int NP>
__device__ __forceinline__ void mil_otile_commit(const uint4 (&r)[NP], char* lds, int tid, int tile_px) {
    constexpr int ESZ = T::ESZ;
    constexpr int PIXZ = mil_pix_pitch(CP, ESZ);
    constexpr int N16 = CP * ESZ / 16;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = tid + 256 * i;
        if (idx < tile_px * N16) {
            const int tp = idx / N16, j = idx - tp * N16;
            *reinterpret_cast<uint4*>(lds + tp * PIXZ + j * 16) = r[i];
        }
    }
}

// pieces-per-thread bound for a halo of at most 400 pixels (18x18, 19x19 and 4 x 10x10 tiles)
__host__ __device__ constexpr int mil_halo_np(int cinp, int esz) { return (400 * (cinp * esz / 16) + 255) / 256; }
// same for 128-px tiles: 10x18, 2 x 10x10 and 8 x 6x6 halos (<= 288 pixels)
__host__ __device__ constexpr int mil_halo_px_max(int mtw) { return mtw == 4 ? 400 : 288; }
__host__ __device__ constexpr int mil_halo_np_mtw(int cinp, int esz, int mtw) { return (mil_halo_px_max(mtw) * (cinp * esz / 16) + 255) / 256; }

// ---- stride-2 transposed conv (3x3, pad 1) by output parity class ---------------------------------------------
// dx(2i+py, 2j+px) only sees the taps (ky,kx) of the zero-insert form whose source coordinate is even:
// py=0 -> ky=1 (dz row i);  py=1 -> ky=0 (row i), ky=2 (row i+1); same for columns.  Class c = py*2+px therefore
// is a 1-, 2-, 2- or 4-tap conv over the COMPACT dz map; the 1x1/s2 projection's transposed conv lands on class 0
// only (second source).  K-groups of a class: (tap, 8-channel group) in tap order, then the projection's groups.
#define MIL_PACK_DGRAD_S2 3
struct S2Group { int src, ky, kx, di, dj, cg; bool valid; };
__host__ __device__ inline S2Group mil_s2_group(int c, int q, int CG) {
    S2Group gr{0, 0, 0, 0, 0, 0, false};
    const int py = c >> 1, px = c & 1;
    const int nkx = px ? 2 : 1, ntap = (py ? 2 : 1) * nkx;
    if (q < ntap * CG) {
        const int t = q / CG;
        gr.cg = q - t * CG;
        const int iy = t / nkx, ix = t - iy * nkx;
        gr.ky = py ? 2 * iy : 1; gr.kx = px ? 2 * ix : 1;
        gr.di = (py + gr.ky - 1) >> 1; gr.dj = (px + gr.kx - 1) >> 1;
        gr.valid = true;
    } else if (c == 0 && q < 2 * CG) {
        gr.src = 1; gr.cg = q - CG; gr.valid = true;
    }
    return gr;
}
__host__ __device__ constexpr int mil_s2_steps(int c, int CG) { return c == 3 ? CG : (CG + 1) / 2; }
__host__ __device__ constexpr int mil_s2_nsteps(int CG) { return 3 * ((CG + 1) / 2) + CG; }
// packed element (global k-step s, lane, e, column tile nt) of the class-ordered filter: w1 [cout][cin][3][3] is the
// forward conv's weight, wproj [cout][cin] the 1x1 projection's (or null); k = its output channel, n = its input channel
__host__ __device__ inline float mil_s2_pack_value(const float* w1, const float* wproj, int s, int lane, int e, int nt,
                                                   int cout, int cin, int CG) {
    int c = 0;
    while (s >= mil_s2_steps(c, CG)) { s -= mil_s2_steps(c, CG); ++c; }
    const S2Group gr = mil_s2_group(c, 4 * s + (lane >> 4), CG);
    const int kin = gr.cg * 8 + e, nout = nt * 16 + (lane & 15);
    if (!gr.valid || kin >= cout || nout >= cin) return 0.f;
    if (gr.src == 1) return wproj ? wproj[(size_t)kin * cin + nout] : 0.f;
    return w1[((size_t)kin * cin + nout) * 9 + (2 - gr.ky) * 3 + (2 - gr.kx)];
}

// ---- K-packed order of a 3x3 filter over 20 channels ("K20") ----------------------------------------------------
// 9 taps x 20 channels are 45 four-channel pieces.  The standard order spends 27 eight-channel k-groups on them (7
// k-steps: the third group of every tap is half padding).  Here the half groups of two horizontally neighbouring taps
// share a k-group:
//   q = 0..17 : tap q/2, channels 8*(q&1) .. +7
//   q = 18..20: filter row ty = q-18: elements 0-3 = tap (ty,0) channels 16-19, elements 4-7 = tap (ty,1) channels 16-19
//   q = 21..23: filter row ty = q-21: elements 0-3 = tap (ty,2) channels 16-19, elements 4-7 = zero weights
// 24 k-groups = 6 k-steps (-14 % MFMAs and fragment reads).  The kernels that use it keep the LDS pixel record as
// [ch 0-15][ch 16-19][ch 16-19 of the NEXT pixel of the tile row] (the 48-byte pitch of a 24-channel record), so every
// k-group is still ONE aligned 16-byte read at (tap offset) + {0, 16, 32}.
#define MIL_K20_STEPS 6
struct K20Elem { int tap, ch; };              // tap < 0: a zero weight
__host__ __device__ inline K20Elem mil_k20_elem(int q, int e) {
    if (q < 18) return K20Elem{q >> 1, 8 * (q & 1) + e};
    if (q < 21) return K20Elem{(q - 18) * 3 + (e >> 2), 16 + (e & 3)};
    if (q < 24 && e < 4) return K20Elem{(q - 21) * 3 + 2, 16 + e};
    return K20Elem{-1, 0};
}
// LDS byte offset of k-group q's 16 bytes relative to the record of the pixel under the filter's top-left tap
__host__ __device__ constexpr int mil_k20_off(int q, int row_pitch, int pix_pitch) {
    return q < 18 ? ((q >> 1) / 3) * row_pitch + ((q >> 1) % 3) * pix_pitch + (q & 1) * 16
         : q < 21 ? (q - 18) * row_pitch + 32
         : q < 24 ? (q - 21) * row_pitch + 2 * pix_pitch + 32 : 0;
}
// packed buffers of these filters carry the K20 k-steps behind the standard ones (mode 0: forward, K side = cin; mode 1:
// data gradient, K side = cout)
__host__ __device__ constexpr bool mil_pack_has_k20(int mode, int cout, int cin, int ks) {
    return ks == 3 && ((mode == 0 && cin == 20) || (mode == 1 && cout == 20));
}

// ---- K-packed order of the stem's 4x4 filter over the 12 real space-to-depth channels ("SK6") ----------------------------
// A space-to-depth record is [c0: 4 ch][c1: 4 ch][c2: 4 ch][4 padding channels]; the standard order spends 32 eight-channel
// k-groups (8 k-steps) on 16 taps x 16 channels, a quarter of them padding.  Here the c2 halves of two horizontally
// neighbouring taps share a k-group — the kernel keeps [c2 of the NEXT pixel] in the record's padding bytes, so the group is
// still one aligned 16-byte read:
//   q = 6*ty + g:  g = 0..3: tap (ty, g), channels 0-7;   g = 4: taps (ty,0),(ty,1) channels 8-11;   g = 5: taps (ty,2),(ty,3)
// 24 k-groups = 6 k-steps (-25 % MFMAs and fragment reads).  The packed stem filter of a 20-channel stem carries these six
// k-steps BEHIND the eight standard ones.
#define MIL_SK6_STEPS 6
__host__ __device__ inline K20Elem mil_sk6_elem(int q, int e) {
    const int ty = q / 6, g = q - ty * 6;
    if (g < 4) return K20Elem{ty * 4 + g, e};
    return K20Elem{ty * 4 + (g == 4 ? 0 : 2) + (e >> 2), 8 + (e & 3)};
}
__host__ __device__ constexpr int mil_sk6_off(int q, int row_pitch, int pix_pitch) {
    return (q / 6) * row_pitch + ((q % 6) < 4 ? (q % 6) : ((q % 6) == 4 ? 0 : 2)) * pix_pitch + ((q % 6) >= 4 ? 16 : 0);
}
__host__ __device__ constexpr bool mil_pack_has_sk6(int mode, int cout) { return mode == 2 && cout <= 32; }      // MIL_PACK_STEM, two column tiles

// Fixed-order sum over slabs for the weight-gradient reductions: thread group gq of MIL_RED_GROUPS sums slabs gq,
// gq+G, gq+2G, ... (8 independent loads in flight per thread, added in index order), group 0 then adds the G partial
// sums in order.  The tree depends only on (nslab, G): bitwise reproducible run to run.
#define MIL_RED_GROUPS 32
__device__ __forceinline__ float mil_slab_partial(const float* __restrict__ slab, size_t slab_elems, size_t src, int gq, int nslab) {
    float s = 0.f;
    int i = gq;
    for (; i + 7 * MIL_RED_GROUPS < nslab; i += 8 * MIL_RED_GROUPS) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = slab[(size_t)(i + k * MIL_RED_GROUPS) * slab_elems + src];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; i < nslab; i += MIL_RED_GROUPS) s += slab[(size_t)i * slab_elems + src];
    return s;
}
