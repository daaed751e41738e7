// Fused stem forward: fp32 NCHW tiles -> space-to-depth (kept for the backward) -> 7x7/s2 conv + bias +
// LeakyReLU -> MaxPool2d(3,2,1) + winner records, in ONE pass (reference ops gbm/model.py:24-26,51-53).
//
// The unfused chain writes and re-reads the s2d tensor (1.07 GB at T=2048, 256x256) and the stem activation
// (1.6 GB + 1.6 GB) through HBM; here the fp32 input is read once, the s2d tile and the stem tile live in LDS,
// and only what later kernels need is written: the s2d tensor (stem weight gradient), the pooled map and the
// 1-byte winner records.  Arithmetic is the same as the unfused kernels' (same packed filter, same k order,
// bias-initialised accumulators, bf16 rounding of the stem activation before the pool), so results are
// bit-identical to stem_s2d -> conv_igemm -> maxpool_fwd.
//
// Tile = 8x16 pooled pixels of one image  <- 17x33 stem pixels <- 20x36 s2d pixels (loaded as 20x38 so that the
// fp32 input is fetched in aligned 16-byte pieces) <- 40x76 input pixels x 3 colours.  Workgroups are persistent;
// each XCD walks its own contiguous range of tiles so that neighbouring tiles (which share halo rows/columns)
// are in flight on the same L2.
#include "pf_common.cuh"
#include "stamp.cuh"
#include <type_traits>

#ifndef MIL_STEM_FWD_LOOKAHEAD
#define MIL_STEM_FWD_LOOKAHEAD 2      // pixel fragments read this many (k-step, row tile) steps ahead of their MFMAs; 0 = compiler order
#endif

struct StemFwdArgs {
    const float* x;            // [n,3,H,W] (FROM_XS: null)
    const __bf16* xs_in;       // FROM_XS: the input already as bf16 space-to-depth NHWC [n,H2,W2,16] (mil_tile_preprocess_s2d)
    const void* w;             // MIL_PACK_STEM fragments [8][NT][64][8] (bf16; MIL_DT_F32S: [hi | lo] pairs)
    const float* bias;         // [NT*16]
    __bf16* xs;                // [n,H2,W2,16] (bf16 path only; may be null)
    void* pool;                // [n,Ho,Wo,COUTP] bf16, or fp32 (MIL_DT_F32S)
    uint8_t* widx;             // [n,Ho,Wo,COUTP]
    int n_img, H, W, H2, W2, Ho, Wo, tiles_x, tiles_y, ntiles;
    float slope;
    unsigned long long* stamp;      // MIL_STAMP diagnostic build only
};

// Tile = PH x 16 pooled pixels.  PH = 8 (bf16): 17x33 stem pixels <- 20x38 s2d pixels.  PH = 4 (split precision): 9x33 <- 12x38,
// which, with the filter streamed from L1/L2 instead of staged, is 70 KB of LDS: TWO 4-wave workgroups per CU, each running its
// convert / store / pool phases under the other's MFMAs (the 8x16 form was one 8-wave workgroup per CU on 156 KB, every wave of
// the CU in the same phase: gemm 39 % of a tile in the phase stamps).  Measured: 2.00 -> 1.89 ms per launch (2048 tiles of 256x256).
#ifndef MIL_STEM_X3_PH
#define MIL_STEM_X3_PH 4
#endif
__host__ __device__ constexpr int sf_ph(bool x3) { return x3 ? MIL_STEM_X3_PH : 8; }
__host__ __device__ constexpr bool sf_wstream(bool x3) { return x3 && MIL_STEM_X3_PH < 8; }
constexpr int SF_SW = 33;                           // stem tile width
constexpr int SF_XW = 38, SF_NPAIR = 19;            // s2d tile width; a "pair" = 2 s2d pixels = 4 input columns
__host__ __device__ constexpr int sf_sh(int ph) { return 2 * ph + 1; }      // stem tile rows
__host__ __device__ constexpr int sf_xh(int ph) { return 2 * ph + 4; }      // s2d tile rows
// s2d pixel record in LDS: 16 ch bf16 = 32 B at an odd 16-B slot pitch (48); split precision (X3): [hi 32 B][lo 32 B] at pitch 80
__host__ __device__ constexpr int sf_xpix(bool x3) { return x3 ? 80 : 48; }
__host__ __device__ constexpr int sf_xbytes(bool x3) { return sf_xh(sf_ph(x3)) * SF_XW * sf_xpix(x3); }  // 36480 (bf16, 20 rows) / 36480 (split, 12 rows)
__host__ __device__ constexpr int sf_nitem(int ph) { return sf_xh(ph) * SF_NPAIR * 3; }      // (row, pair, colour) load items
__host__ __device__ constexpr int sf_nstem(int ph) { return sf_sh(ph) * SF_SW; }             // 561 / 297
__host__ __device__ constexpr int sf_mtiles(int ph) { return (sf_nstem(ph) + 15) / 16; }     // 16-pixel row tiles of the stem tile: 36 / 19

template <int NT, bool X3 = false>
__host__ __device__ constexpr int sf_lds_bytes() {
    return sf_xbytes(X3) + ((sf_nstem(sf_ph(X3)) * mil_pix_pitch(mil_nt_to_cp(NT), X3 ? 4 : 2) + 15) & ~15) +
           (sf_wstream(X3) ? 0 : 8 * NT * 64 * (X3 ? 32 : 16)) + 256;   // + dump slot
}

// NW = waves per workgroup.  The kernel is VALU-bound (pool compare/select, activation, conversions) and at 4 waves per
// workgroup its 238 VGPRs leave two waves per SIMD, which keep the vector pipe only half busy; with 8 waves every
// per-wave quantity halves (5 row tiles, 3 load items, 2 pool items) and four waves per SIMD fit on the same LDS tiles.
// X3 (MIL_DT_F32S: fp32 tensors, bf16x3 split products): the s2d tile holds hi and lo bf16 planes, every (filter, pixel)
// fragment pair costs three MFMAs, the stem tile and the pooled output are fp32.  156 KB of LDS: one 8-wave workgroup per CU.
// FROM_XS (bf16 only): the tiles arrive as the bf16 space-to-depth tensor itself — the s2d tile is a plain halo copy in
// 16-byte pieces (1.07 GB read per 2048 tiles of 256x256 instead of 1.6 GB of fp32 in 304-byte plane segments), everything
// behind it is the same code on the same LDS bytes: bit-identical pooled map and winner records.
template <int NT, int NW, bool X3 = false, bool FROM_XS = false>
__global__ __launch_bounds__(64 * NW, X3 ? 2 : (NT <= 2 ? (NW == 8 ? 4 : (NW == 6 ? 3 : 2)) : 1)) void stem_fwd_fused_kernel(StemFwdArgs a) {
    static_assert(!(X3 && FROM_XS), "the space-to-depth feed is bf16");
    using T = typename std::conditional<X3, F32S, BF16>::type;
    constexpr int PH = sf_ph(X3);                             // pooled rows per tile
    constexpr bool WSTREAM = sf_wstream(X3);                  // filter fragments from L1/L2, not from LDS
    // the streamed branch of wfrag() is written for the folded split-precision form only (column tile 1 = [wh ; wl]) and for the
    // pipelined loop (the MIL_STEM_FWD_LOOKAHEAD == 0 loop reads ldsW, which is not staged when the filter streams)
    static_assert(!WSTREAM || (X3 && NT == 2), "streamed filter fragments: folded split-precision form only");
    static_assert(!WSTREAM || MIL_STEM_FWD_LOOKAHEAD > 0, "streamed filter fragments need the pipelined MFMA loop");
    constexpr int SF_SH = sf_sh(PH), SF_XH = sf_xh(PH), SF_NITEM = sf_nitem(PH), SF_NSTEM = sf_nstem(PH), SF_MTILES = sf_mtiles(PH);
    constexpr int SF_XPIX = sf_xpix(X3), SF_XBYTES = sf_xbytes(X3);
    constexpr int OESZ = X3 ? 4 : 2;                          // bytes per element of the stem tile / pooled output
    constexpr int FRAGB = X3 ? 32 : 16;
    constexpr int COUTP = mil_nt_to_cp(NT);
    constexpr int SPIX = mil_pix_pitch(COUTP, OESZ);
    constexpr int KSTEPS = 8;
    constexpr int NG4 = COUTP / 4;
    constexpr int NTHR = 64 * NW;
    constexpr int SF_NLOAD = ((FROM_XS ? SF_XH * SF_XW * 2 : SF_NITEM) + NTHR - 1) / NTHR;      // FROM_XS: items = 16-byte halves of the tile's s2d records
    constexpr int SF_MT = (SF_MTILES + NW - 1) / NW;          // row tiles per wave
    constexpr int NXS = (1024 + NTHR - 1) / NTHR;             // 16-byte pieces of the tile's own 16x32 s2d pixels per thread
    constexpr int NPOOL = (PH * 16 * NG4 + NTHR - 1) / NTHR;  // (pooled pixel, 4-channel group) items per thread
    constexpr bool LAST_PARTIAL = (COUTP % 16) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    char* ldsX = smem;
    char* ldsS = smem + SF_XBYTES;
    char* ldsW = ldsS + ((SF_NSTEM * SPIX + 15) & ~15);
    // LDS writes of unused slots (the last partial rounds of the tables below) go to a dump area instead of being
    // branched around: a divergent branch per store costs more than the store.
    const int dump = sf_lds_bytes<NT, X3>() - 256;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    {
        if constexpr (!WSTREAM) mil_stage_filter(ldsW, a.w, KSTEPS * NT * 64 * FRAGB, tid, NTHR);
        for (int i = tid * 16; i < SF_XBYTES; i += NTHR * 16)          // channels 12..15 of every s2d pixel stay zero
            *reinterpret_cast<uint4*>(ldsX + i) = make_uint4(0, 0, 0, 0);
    }
    const int H = a.H, W = a.W, H2 = a.H2, W2 = a.W2, Ho = a.Ho, Wo = a.Wo;
    const __amdgpu_buffer_rsrc_t rs_x = FROM_XS ? mil_rsrc(a.xs_in, (unsigned)((size_t)a.n_img * H2 * W2 * 32))
                                                : mil_rsrc(a.x, (unsigned)((size_t)a.n_img * 3 * H * W * 4));
    const __amdgpu_buffer_rsrc_t rs_xs = mil_rsrc(a.xs, (unsigned)((size_t)a.n_img * H2 * W2 * 32));
    const __amdgpu_buffer_rsrc_t rs_p = mil_rsrc(a.pool, (unsigned)((size_t)a.n_img * Ho * Wo * COUTP * OESZ));
    const __amdgpu_buffer_rsrc_t rs_i = mil_rsrc(a.widx, (unsigned)((size_t)a.n_img * Ho * Wo * COUTP));

    // ---- tile-invariant tables --------------------------------------------------------------------
    int l_lds[SF_NLOAD], l_rel[SF_NLOAD];       // input -> s2d tile; l_lds = LDS offset | row << 18 | pair << 24 (row 31 = unused)
#pragma unroll
    for (int i = 0; i < SF_NLOAD; ++i) {
        const int idx = tid + NTHR * i;
        l_lds[i] = dump | (31 << 18); l_rel[i] = 0;
        if constexpr (FROM_XS) {             // (row, s2d column, half): "pair" field = the column
            if (idx < SF_XH * SF_XW * 2) {
                const int half = idx & 1, col = (idx >> 1) % SF_XW, row = (idx >> 1) / SF_XW;
                l_lds[i] = ((row * SF_XW + col) * SF_XPIX + half * 16) | (row << 18) | (col << 24);
                l_rel[i] = (row * W2 + col) * 32 + half * 16;
            }
        } else
        if (idx < SF_NITEM) {
            const int pair = idx % SF_NPAIR, t = idx / SF_NPAIR;
            const int c = t % 3, row = t / 3;
            l_lds[i] = ((row * SF_XW + 2 * pair) * SF_XPIX + c * 8) | (row << 18) | (pair << 24);
            l_rel[i] = ((c * H + 2 * row) * W + 4 * pair) * 4;
        }
    }
    // s2d tile interior -> xs tensor: 16-B piece id = tid + NTHR*i -> (row (NTHR/64)*i + tid>>6, col (tid>>1)&31, half tid&1)
    const int x_row0 = tid >> 6, x_col = (tid >> 1) & 31;
    const int x_lds0 = ((x_row0 + 3) * SF_XW + x_col + 4) * SF_XPIX + (tid & 1) * 16;
    const int x_rel0 = (x_row0 * W2 + x_col) * 32 + (tid & 1) * 16;
    int pixbase[SF_MT], sdst[SF_MT];                             // MFMA row tiles of the stem tile
#pragma unroll
    for (int m = 0; m < SF_MT; ++m) {
        const int tp = (wave + NW * m) * 16 + r;
        const bool ok = tp < SF_NSTEM;
        const int sy = tp / SF_SW, sx = tp - sy * SF_SW;
        // k-group q = 4*step + gq is tap 2*step + (gq>>1), channel group gq&1: the lane-dependent part of the tap offset
        // ((gq>>1) pixels + (gq&1) pieces) is folded in here, the step-dependent part is a compile-time immediate below
        pixbase[m] = (ok ? (sy * SF_XW + sx + 1) * SF_XPIX : 0) + (gq >> 1) * SF_XPIX + (gq & 1) * 16;
        sdst[m] = (ok ? SF_XBYTES + tp * SPIX : dump) + gq * 4 * OESZ;
    }
    int p_lds[NPOOL], p_rel[NPOOL];          // pooled pixels x 4-channel groups; p_lds = LDS offset | py << 20 | px << 24 (py 15 = unused)
#pragma unroll
    for (int i = 0; i < NPOOL; ++i) {
        const int id = tid + NTHR * i;
        p_lds[i] = 15 << 20; p_rel[i] = 0;
        if (id < PH * 16 * NG4) {
            const int c4 = id % NG4, pp = id / NG4, py = pp >> 4, px = pp & 15;
            p_lds[i] = (((2 * py) * SF_SW + 2 * px) * SPIX + c4 * 4 * OESZ) | (py << 20) | (px << 24);
            p_rel[i] = (py * Wo + px) * COUTP + c4 * 4;
        }
    }
    f32x4_t bias_r[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias_r[nt][i] = a.bias ? a.bias[nt * 16 + gq * 4 + i] : 0.f;
    if constexpr (X3 && NT == 2 && MIL_STEM_FWD_LOOKAHEAD > 0) {      // folded column tile 1: rows 4-15 accumulate w_lo * x_hi of rows 0-3 — they
#pragma unroll                                                       // start at zero whatever bias_pad[20..] holds (the public C ABI does not promise zeros there)
        for (int i = 0; i < 4; ++i) bias_r[1][i] = gq == 0 ? bias_r[1][i] : 0.f;
    }

    // ---- tile walk: XCD x (= blockIdx & 7) owns tiles [x*per, (x+1)*per) ---------------------------
    const int G8 = gridDim.x >> 3, per = (a.ntiles + 7) >> 3;
    const int t_begin = (blockIdx.x & 7) * per;
    const int t_end = min(t_begin + per, a.ntiles);
    int tile = t_begin + (blockIdx.x >> 3);

    u32x4_t r0[SF_NLOAD], r1[FROM_XS ? 1 : SF_NLOAD];
    auto fetch = [&](int t) {
        const int tx = t % a.tiles_x, q = t / a.tiles_x, ty = q % a.tiles_y, img = q / a.tiles_y;
        const int y0 = 2 * PH * ty - 3, c0 = 64 * tx - 8;        // first s2d row / first input column of the tile
        if constexpr (FROM_XS) {
            const int x0 = 32 * tx - 4;                          // first s2d column of the tile
            const int base = ((img * H2 + y0) * W2 + x0) * 32;
#pragma unroll
            for (int i = 0; i < SF_NLOAD; ++i) {
                const int row = (l_lds[i] >> 18) & 31, col = l_lds[i] >> 24;
                const bool ok = (unsigned)(y0 + row) < (unsigned)H2 && (unsigned)(x0 + col) < (unsigned)W2;
                r0[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(base + l_rel[i]) : MIL_OOB, 0, 0);
            }
            return;
        }
        const int base = (((img * 3) * H + 2 * y0) * W + c0) * 4;
#pragma unroll
        for (int i = 0; i < SF_NLOAD; ++i) {
            const int row = (l_lds[i] >> 18) & 31, pair = l_lds[i] >> 24;
            const bool ok = (unsigned)(y0 + row) < (unsigned)H2 && (unsigned)(c0 + 4 * pair) < (unsigned)W;      // row 31 is never inside
            const unsigned off = ok ? (unsigned)(base + l_rel[i]) : MIL_OOB;
            r0[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
            r1[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off + (unsigned)(W * 4), 0, 0);
        }
    };
    if (tile < t_end) fetch(tile);
    __syncthreads();

    MIL_STAMP_DECL(7)
    for (; tile < t_end; tile += G8) {
        const int tx = tile % a.tiles_x, tq = tile / a.tiles_x, ty = tq % a.tiles_y, img = tq / a.tiles_y;
        MIL_STAMP_BEGIN()
        // ---- s2d tile: fp32 -> bf16, channel = c*4 + dy*2 + dx --------------------------------------
        if constexpr (FROM_XS) {
#pragma unroll
            for (int i = 0; i < SF_NLOAD; ++i) *reinterpret_cast<u32x4_t*>(smem + (l_lds[i] & 0x3FFFF)) = r0[i];
        } else
#pragma unroll
        for (int i = 0; i < SF_NLOAD; ++i) {
            const f32x4_t v0 = __builtin_bit_cast(f32x4_t, r0[i]), v1 = __builtin_bit_cast(f32x4_t, r1[i]);
            const float fa[4] = {v0[0], v0[1], v1[0], v1[1]}, fb[4] = {v0[2], v0[3], v1[2], v1[3]};
            bf16x4_t pa, pb, qa, qb;
            if constexpr (X3) {
                mil_split4(f32x4_t{fa[0], fa[1], fa[2], fa[3]}, pa, qa);
                mil_split4(f32x4_t{fb[0], fb[1], fb[2], fb[3]}, pb, qb);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { pa[j] = (__bf16)fa[j]; pb[j] = (__bf16)fb[j]; }
            }
            char* dst = smem + (l_lds[i] & 0x3FFFF);
            *reinterpret_cast<bf16x4_t*>(dst) = pa;
            *reinterpret_cast<bf16x4_t*>(dst + SF_XPIX) = pb;
            if constexpr (X3) {                                  // lo plane: 32 bytes behind the hi plane of the same pixel
                *reinterpret_cast<bf16x4_t*>(dst + 32) = qa;
                *reinterpret_cast<bf16x4_t*>(dst + SF_XPIX + 32) = qb;
            }
        }
        MIL_STAMP_MARK(0)
        __syncthreads();
        MIL_STAMP_MARK(1)
        if (tile + G8 < t_end) fetch(tile + G8);
        MIL_STAMP_MARK(2)
        // ---- the tile's own 16x32 s2d pixels go to the xs tensor (when the caller keeps one) ------------
        if (!X3 && !FROM_XS && a.xs) {
            const int xbase = ((img * H2 + 16 * ty) * W2 + 32 * tx) * 32;
            const int ylim = H2 - 16 * ty, xlim = W2 - 32 * tx;
#pragma unroll
            for (int i = 0; i < NXS; ++i) {
                constexpr int RPI = NTHR / 64;                   // rows covered per round
                const bool ok = x_row0 + RPI * i < (ylim < 16 ? ylim : 16) && x_col < xlim;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(ldsX + x_lds0 + i * (RPI * SF_XW * SF_XPIX));
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_xs, ok ? (unsigned)(xbase + x_rel0 + i * (RPI * W2 * 32)) : MIL_OOB, 0, 0);
            }
        }
        // ---- 4x4 s1 implicit GEMM over the s2d tile, D[channel][pixel] ------------------------------
        f32x4_t acc[SF_MT][NT];
#pragma unroll
        for (int m = 0; m < SF_MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = bias_r[nt];
#ifdef MIL_EXP_STEMF_NO_CONV
#elif MIL_STEM_FWD_LOOKAHEAD > 0
        {
            // The (k-step, row tile) loop flattened into one software pipeline: the pixel fragment of step j+LA is read
            // LA steps before the two MFMAs that consume it (a ring of LA+1 fragments), the filter fragments one k-step
            // ahead, and scheduling fences keep that order.  Left alone, hipcc reads each fragment right in front of its
            // MFMA pair behind an lgkmcnt(0): 72 LDS round trips per tile and wave, two thirds of this kernel's time.
            constexpr int TOT = KSTEPS * SF_MT, LA = MIL_STEM_FWD_LOOKAHEAD, R = LA + 1;
            constexpr int WD = WSTREAM ? 2 : 1, WR = WD + 1;      // filter fragments: k-steps ahead (streamed: an L2 round trip against 25 MFMAs per k-step) / ring slots
            Frag8<T> ring[R], wq[WR][NT];
            // X3, 20 channels: column tile 1 has four real rows (channels 16-19), so its rows 4-7 carry the LO halves of the same
            // channels (lanes r = 4..7 read lane r-4's lo half): [wh ; wl] x xh, then wh x xl — two MFMAs instead of three; the
            // epilogue adds rows 4-7 onto rows 0-3
            constexpr bool FOLD = X3 && NT == 2;
            const int w1h = (FOLD && r >= 4 && r < 8) ? (64 + lane - 4) * FRAGB + 16 : (64 + lane) * FRAGB;
            const __amdgpu_buffer_rsrc_t rs_w = mil_rsrc(a.w, KSTEPS * NT * 64 * FRAGB);
            auto wfrag = [&](int sl, int nt) {
                if constexpr (WSTREAM) {             // (FOLD form only) 16 bytes per lane and half, the k-step in the scalar offset
                    Frag8<T> f;
                    f.h = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(nt == 1 ? w1h : lane * FRAGB), sl * NT * 64 * FRAGB, 0));
                    f.l = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(nt == 1 ? (64 + lane) * FRAGB : lane * FRAGB + 16), sl * NT * 64 * FRAGB, 0));
                    return f;
                } else
                if constexpr (FOLD) {
                    if (nt == 1) {
                        Frag8<T> f;
                        f.h = *reinterpret_cast<const bf16x8_t*>(ldsW + sl * NT * 64 * FRAGB + w1h);
                        f.l = *reinterpret_cast<const bf16x8_t*>(ldsW + ((sl * NT + 1) * 64 + lane) * FRAGB);      // the hi half: rows 4-15 are zero weights
                        return f;
                    }
                }
                return lds_frag<T>(ldsW + ((sl * NT + nt) * 64 + lane) * FRAGB);
            };
#pragma unroll
            for (int k = 0; k < WD; ++k)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wq[k][nt] = wfrag(k, nt);
#pragma unroll
            for (int j = 0; j < LA; ++j)
                ring[j % R] = lds_pix_frag<T, 32>(ldsX + pixbase[j % SF_MT] + (((j / SF_MT) >> 1) * SF_XW + 2 * ((j / SF_MT) & 1)) * SF_XPIX);
#pragma unroll
            for (int j = 0; j < TOT; ++j) {
                const int sl = j / SF_MT, m = j % SF_MT;
                if (j + LA < TOT) {
                    const int jn = j + LA, sn = jn / SF_MT;
                    ring[jn % R] = lds_pix_frag<T, 32>(ldsX + pixbase[jn % SF_MT] + ((sn >> 1) * SF_XW + 2 * (sn & 1)) * SF_XPIX);
                }
                if (m == 0 && sl + WD < KSTEPS) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) wq[(sl + WD) % WR][nt] = wfrag(sl + WD, nt);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (FOLD) {
                    acc[m][0] = mma8(wq[sl % WR][0], ring[j % R], acc[m][0]);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[sl % WR][1].h, ring[j % R].h, acc[m][1], 0, 0, 0);      // [wh ; wl] x xh
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[sl % WR][1].l, ring[j % R].l, acc[m][1], 0, 0, 0);      // wh x xl
                } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mma8(wq[sl % WR][nt], ring[j % R], acc[m][nt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
#pragma unroll
        for (int sl = 0; sl < KSTEPS; ++sl) {
            Frag8<T> wf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = lds_frag<T>(ldsW + ((sl * NT + nt) * 64 + lane) * FRAGB);
#pragma unroll
            for (int m = 0; m < SF_MT; ++m) {
                const Frag8<T> xf = lds_pix_frag<T, 32>(ldsX + pixbase[m] + ((sl >> 1) * SF_XW + 2 * (sl & 1)) * SF_XPIX);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mma8(wf[nt], xf, acc[m][nt]);
            }
        }
#endif
        MIL_STAMP_MARK(3)
        // Stem pixels outside the image are the pool's -inf padding: written as such, so that the pool phase below
        // needs no per-tap bounds tests (only tiles on the image border have any).
        const int sy0 = 2 * PH * ty - 1, sx0 = 32 * tx - 1;       // image coordinates of stem-tile pixel (0,0)
        const bool border = sy0 < 0 || sx0 < 0 || sy0 + SF_SH > H2 || sx0 + SF_SW > W2;
#pragma unroll
        for (int m = 0; m < SF_MT; ++m) {
            bool inside = true;
            if (border) {                                     // rare: recompute the pixel's tile coordinates instead of keeping a table
                const int tp = (wave + NW * m) * 16 + r, sy = (tp * 1986) >> 16, sx = tp - sy * SF_SW;      // tp / 33 for tp < 1024
                inside = (unsigned)(sy0 + sy) < (unsigned)H2 && (unsigned)(sx0 + sx) < (unsigned)W2;
            }
            if constexpr (X3 && NT == 2 && MIL_STEM_FWD_LOOKAHEAD > 0) {      // column tile 1: rows 4-7 (lane group gq == 1) hold wl*xh of rows 0-3
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t0 = acc[m][1][i], t1 = t0;
                    if (i == 0) mil_swap16<true>(t0, t1); else mil_swap16<false>(t0, t1);      // t1 of lane group 0 = group 1's value
                    acc[m][1][i] = gq == 0 ? acc[m][1][i] + t1 : 0.f;        // channels 20-23 are padding: exact zeros
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (LAST_PARTIAL && nt == NT - 1 && gq >= 2) continue;        // channels COUTP.. do not exist
                if constexpr (X3) {
                    f32x4_t o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float v = acc[m][nt][i]; o[i] = fmaxf(v, v * a.slope); }
                    u32x4_t ou = __builtin_bit_cast(u32x4_t, o);
                    if (border) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) ou[i] = inside ? ou[i] : 0xFF800000u;      // -inf
                    }
                    *reinterpret_cast<u32x4_t*>(smem + sdst[m] + nt * 64) = ou;
                } else {
                bf16x4_t o;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float v = acc[m][nt][i]; o[i] = (__bf16)fmaxf(v, v * a.slope); }
                u32x2_t ou = __builtin_bit_cast(u32x2_t, o);
                if (border) { ou[0] = inside ? ou[0] : 0xFF80FF80u; ou[1] = inside ? ou[1] : 0xFF80FF80u; }
                *reinterpret_cast<u32x2_t*>(smem + sdst[m] + nt * 32) = ou;
                }
            }
        }
        MIL_STAMP_MARK(4)
        __syncthreads();
        MIL_STAMP_MARK(5)
        // ---- 3x3 s2 max-pool of the stem tile: first maximum in (ky,kx) scan order wins ---------------------
#ifndef MIL_EXP_STEMF_NO_POOL
        {
            const int obase = ((img * Ho + PH * ty) * Wo + 16 * tx) * COUTP;
            const int ylim = Ho - PH * ty, xlim = Wo - 16 * tx;
#pragma unroll
            for (int it = 0; it < NPOOL; ++it) {
                if (wave * 64 + NTHR * it >= PH * 16 * NG4) continue;      // a wave whose 64 items of this round all lie behind the last one (wave-uniform)
                const int py = (p_lds[it] >> 20) & 15, px = p_lds[it] >> 24;
                const char* src = ldsS + (p_lds[it] & 0xFFFFF);
                float best[4];
                unsigned bi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { best[j] = -INFINITY; bi[j] = 0; }
                if constexpr (X3) {
                    f32x4_t t[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) t[k] = *reinterpret_cast<const f32x4_t*>(src + ((k / 3) * SF_SW + (k % 3)) * SPIX);
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (t[k][j] > best[j]) { best[j] = t[k][j]; bi[j] = k; }
                        }
                    }
                } else {
                u32x2_t t[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) t[k] = *reinterpret_cast<const u32x2_t*>(src + ((k / 3) * SF_SW + (k % 3)) * SPIX);
#pragma unroll
                for (int k = 0; k < 9; ++k) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned d = t[k][j >> 1];
                        const float v = __uint_as_float((j & 1) ? (d & 0xFFFF0000u) : (d << 16));
                        if (v > best[j]) { best[j] = v; bi[j] = k; }
                    }
                }
                }
                const bool ok = py < ylim && px < xlim;                 // py 15 (unused slot) is never inside: ylim <= 8
                const unsigned eoff = (unsigned)(obase + p_rel[it]);
                if constexpr (X3) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{best[0], best[1], best[2], best[3]}), rs_p,
                                                           ok ? eoff * 4u : MIL_OOB, 0, 0);
                } else {
                    u32x2_t ov;
                    ov[0] = (__float_as_uint(best[0]) >> 16) | (__float_as_uint(best[1]) & 0xFFFF0000u);
                    ov[1] = (__float_as_uint(best[2]) >> 16) | (__float_as_uint(best[3]) & 0xFFFF0000u);
                    __builtin_amdgcn_raw_buffer_store_b64(ov, rs_p, ok ? eoff * 2u : MIL_OOB, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) bi[j] |= (best[j] > 0.f) ? 0u : 16u;
                const unsigned rec = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
                __builtin_amdgcn_raw_buffer_store_b32(rec, rs_i, ok ? eoff : MIL_OOB, 0, 0);
            }
        }
#endif
        MIL_STAMP_MARK(6)
    }
    MIL_STAMP_STORE(a.stamp, NW)
}


// =====================================================================================================================
// Round 5: the 20-channel stem with the max-pool taken IN REGISTERS, straight from the fp32 accumulators.
//
// The kernel above stores the activated stem tile to LDS (bf16), meets at a barrier and pools it with nine
// extract / compare / select / select steps per pooled element: 82 % of its vector instructions and two thirds of its time
// (round-4 phase stamps: pool 32 %, stem-tile store 22 %, gemm 26 %).  The reference pools the fp32 activations
// (gbm/model.py:51-53; LeakyReLU is monotonic, so pool(lrelu(v)) = lrelu(pool(v))): nothing requires the bf16 round trip.
//
// Here a wave owns two pooled rows x 16 pooled columns of the tile and arranges its MFMA pixel tiles so that the pooling
// window of a pooled pixel lies in ONE lane:
//   * D[channel][pixel] accumulators as before (A = filter fragment, B = 16 pixels), but the 16 pixel lanes of an MFMA are the
//     16 EVEN stem columns 2*px (px = the lane's pooled column) or the 16 ODD columns 2*px+1 of one stem row, so the window's
//     columns 2*px and 2*px+1 are two registers of the lane, and its left column 2*px-1 is the odd tile of the lane to the
//     left: one `v_max_f32_dpp row_shr:1`;
//   * lane 0's left column belongs to no tile of the row.  One extra "edge" MFMA tile per wave computes column 2*px0-1 for
//     the wave's five stem rows (pixel lane = row), and `v_max_f32_dpp row_shl:rho bank_mask:0x1` drops row rho's value into
//     lane 0 (the other three lanes of the bank read lanes that a select has set to the sentinel);
//   * the three stem rows of a pooled row are registers of the same lane: one v_max3_f32.
// The winner (first maximum in scan order up to fp32 rounding — see below) rides in the low four mantissa bits of the
// value: every accumulator gets a position code (stem row & 3, stem column & 3) by one v_and_or before the maxima, the code
// of the maximum says which of the nine taps won (a 16-entry table per lane parity / pooled-row parity in a VGPR pair).
// Per pooled element that is ~6 vector instructions for value + winner instead of 36, no LDS round trip, no second barrier.
//
// Numerics.  Values that agree in their upper 28 bits compare by position code: the recorded winner may then be a tap whose
// value lies within 2^-19 (relative) of the maximum, and the pooled value carries the code in bits that a bf16 store rounds
// away (split precision: the nibble is replaced by its midpoint: <= 5e-7 relative, a sixteenth of a bf16x3 product's own
// representation error).  The result is closer to the reference than the kernel above, which pooled the bf16-ROUNDED
// activations (ties on 8 significant bits, broken by scan order).  Out-of-image stem pixels take part as -3e38.
//
// Work per wave and tile: 5 stem rows x (even, odd) + 1 edge tile = 11 pixel tiles x 8 k-steps (the middle stem row of the tile
// boundary is computed by both neighbouring waves: 20 rows per 8x16-pooled-pixel tile instead of 17).  The filter fragments
// are streamed from L1/L2 (16 KB bf16 / 32 KB split, two k-steps ahead), LDS holds only the s2d tile — double-buffered in
// bf16 (one barrier per tile), single in split precision (hi + lo planes: 61 KB; two barriers).
constexpr int SP_XH = 20;                                   // s2d rows of an 8x16-pooled-pixel tile
template <bool X3> __host__ __device__ constexpr int sp_xbytes() { return SP_XH * SF_XW * sf_xpix(X3); }      // 36480 / 60800
#ifndef MIL_SP_PRIO
#define MIL_SP_PRIO 0                 // >0: the vector-instruction phases (maxima, decode, convert) at this priority (measured round 5 at equal repetition counts: no difference beyond the 3 % run-to-run spread)
#endif
#ifndef MIL_SP_WLDS
#define MIL_SP_WLDS 1                 // bf16: filter fragments staged in LDS (16 KB) and ONE s2d buffer; 0: streamed from L1/L2, two buffers
#endif
template <bool X3> __host__ __device__ constexpr bool sp_wlds() { return !X3 && MIL_SP_WLDS; }
template <bool X3> __host__ __device__ constexpr int sp_nbuf() { return (X3 || sp_wlds<X3>()) ? 1 : 2; }
template <bool X3> __host__ __device__ constexpr int sp_lds_bytes() { return 64 + sp_nbuf<X3>() * (sp_xbytes<X3>() + 256) + 160 + (sp_wlds<X3>() ? MIL_SK6_STEPS * 2 * 64 * 16 : 0); }      // spare + tiles + the bias vector (+ the filter)

__device__ __forceinline__ float sp_key(float v, unsigned code) {           // low nibble := position code
    return __uint_as_float((__float_as_uint(v) & ~15u) | code);
}
// h[i] = max(e[i], o[i], o[i] of the lane to the left) for the eight registers of one stem row (in place in e)
__device__ __forceinline__ void sp_hmax8(float (&e)[8], const float (&o)[8]) {
    asm volatile(
        "v_max_f32 %0, %0, %8\n\tv_max_f32 %1, %1, %9\n\tv_max_f32 %2, %2, %10\n\tv_max_f32 %3, %3, %11\n\t"
        "v_max_f32 %4, %4, %12\n\tv_max_f32 %5, %5, %13\n\tv_max_f32 %6, %6, %14\n\tv_max_f32 %7, %7, %15\n\t"
        "v_max_f32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %9, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %2, %10, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %3, %11, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %4, %12, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %5, %13, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %6, %14, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %7, %15, %7 row_shr:1 row_mask:0xf bank_mask:0xf"
        : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7])
        : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]), "v"(o[6]), "v"(o[7]));
}
// lane 0 of every 16-lane row: h = max(h, edge value of stem row RHO) — the edge registers hold row rho in pixel lane rho
template <int RHO>
__device__ __forceinline__ void sp_edge8(float (&h)[8], const float (&ek)[8], float sent, unsigned long long mask) {
    float t0, t1, t2, t3, t4, t5, t6, t7;
#define SP_CND_ "v_cndmask_b32_e64 %8, %24, %16, %25\n\tv_cndmask_b32_e64 %9, %24, %17, %25\n\tv_cndmask_b32_e64 %10, %24, %18, %25\n\t" \
                "v_cndmask_b32_e64 %11, %24, %19, %25\n\tv_cndmask_b32_e64 %12, %24, %20, %25\n\tv_cndmask_b32_e64 %13, %24, %21, %25\n\t" \
                "v_cndmask_b32_e64 %14, %24, %22, %25\n\tv_cndmask_b32_e64 %15, %24, %23, %25\n\t"
#define SP_OPS_ : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]), \
                  "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) \
                : "v"(ek[0]), "v"(ek[1]), "v"(ek[2]), "v"(ek[3]), "v"(ek[4]), "v"(ek[5]), "v"(ek[6]), "v"(ek[7]), "v"(sent), "s"(mask)
    if constexpr (RHO == 0) {
        asm volatile(SP_CND_
            "v_max_f32 %0, %0, %8\n\tv_max_f32 %1, %1, %9\n\tv_max_f32 %2, %2, %10\n\tv_max_f32 %3, %3, %11\n\t"
            "v_max_f32 %4, %4, %12\n\tv_max_f32 %5, %5, %13\n\tv_max_f32 %6, %6, %14\n\tv_max_f32 %7, %7, %15"
            SP_OPS_);
    } else {
#define SP_DPP_(n, RS) "v_max_f32_dpp %" #n ", %" #RS ", %" #n " row_shl:%c26 row_mask:0xf bank_mask:0x1\n\t"
        asm volatile(SP_CND_
            SP_DPP_(0, 8) SP_DPP_(1, 9) SP_DPP_(2, 10) SP_DPP_(3, 11) SP_DPP_(4, 12) SP_DPP_(5, 13) SP_DPP_(6, 14)
            "v_max_f32_dpp %7, %15, %7 row_shl:%c26 row_mask:0xf bank_mask:0x1"
            SP_OPS_, "n"(RHO));
#undef SP_DPP_
    }
#undef SP_CND_
#undef SP_OPS_
}
__device__ __forceinline__ float sp_max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float sp_lrelu(float v, float slope) {            // max(v, slope * v), 0 <= slope < 1
    float t, d;
    asm("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(slope), "v"(v));
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(v), "v"(t));
    return d;
}

template <bool X3, bool FROM_XS>
__global__ __launch_bounds__(256, 2) void stem_fwd_pool_kernel(StemFwdArgs a) {
    static_assert(!(X3 && FROM_XS), "the space-to-depth feed is bf16");
    using T = typename std::conditional<X3, F32S, BF16>::type;
    constexpr int NT = 2, COUTP = 24, NTHR = 256, KSTEPS = MIL_SK6_STEPS, KSTEPS_STD = 8, PH = 8;      // the SK6 order of geom.cuh: 6 k-steps
    constexpr int XPIX = sf_xpix(X3), XBYTES = sp_xbytes<X3>(), NBUF = sp_nbuf<X3>(), BUFSTRIDE = XBYTES + 256;
    constexpr int SPARE = 64;                                 // in front of the tile: the "next pixel" slot of the pixel left of column 0
    constexpr int FRAGB = X3 ? 32 : 16, OESZ = X3 ? 4 : 2;
    constexpr int NITEM = SP_XH * SF_NPAIR * 3;
    constexpr int NLOAD = ((FROM_XS ? SP_XH * SF_XW * 2 : NITEM) + NTHR - 1) / NTHR;
    constexpr int NXS = 1024 / NTHR;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    MIL_POISON(smem_raw);
    char* smem = smem_raw;
    constexpr int dump = XBYTES + SPARE;                      // behind each buffer: where the unused table slots write
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid * 16; i < NBUF * BUFSTRIDE; i += NTHR * 16)         // channels 12..15 of every s2d pixel stay zero
        *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);
    smem += SPARE;
    const int H = a.H, W = a.W, H2 = a.H2, W2 = a.W2, Ho = a.Ho, Wo = a.Wo;
    const __amdgpu_buffer_rsrc_t rs_x = FROM_XS ? mil_rsrc(a.xs_in, (unsigned)((size_t)a.n_img * H2 * W2 * 32))
                                                : mil_rsrc(a.x, (unsigned)((size_t)a.n_img * 3 * H * W * 4));
    const __amdgpu_buffer_rsrc_t rs_xs = mil_rsrc(a.xs, (unsigned)((size_t)a.n_img * H2 * W2 * 32));
    const __amdgpu_buffer_rsrc_t rs_p = mil_rsrc(a.pool, (unsigned)((size_t)a.n_img * Ho * Wo * COUTP * OESZ));
    const __amdgpu_buffer_rsrc_t rs_i = mil_rsrc(a.widx, (unsigned)((size_t)a.n_img * Ho * Wo * COUTP));
    constexpr int W_OFF = KSTEPS_STD * NT * 64 * FRAGB;       // the SK6 k-steps sit behind the eight standard ones
    const __amdgpu_buffer_rsrc_t rs_w = mil_rsrc(static_cast<const char*>(a.w) + W_OFF, KSTEPS * NT * 64 * FRAGB);

    // ---- tile-invariant tables (input -> s2d tile: as in the kernel above) --------------------------
    int l_lds[NLOAD], l_rel[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int idx = tid + NTHR * i;
        l_lds[i] = dump | (31 << 18); l_rel[i] = 0;           // bits 0-15: LDS offset; bit 16: this piece carries the c2 channels (written twice)
        if constexpr (FROM_XS) {
            if (idx < SP_XH * SF_XW * 2) {
                const int half = idx & 1, col = (idx >> 1) % SF_XW, row = (idx >> 1) / SF_XW;
                l_lds[i] = ((row * SF_XW + col) * XPIX + half * 16) | (half << 16) | (row << 18) | (col << 24);
                l_rel[i] = (row * W2 + col) * 32 + half * 16;
            }
        } else
        if (idx < NITEM) {
            const int pair = idx % SF_NPAIR, t = idx / SF_NPAIR;
            const int c = t % 3, row = t / 3;
            l_lds[i] = ((row * SF_XW + 2 * pair) * XPIX + c * 8) | ((c == 2 ? 1 : 0) << 16) | (row << 18) | (pair << 24);
            l_rel[i] = ((c * H + 2 * row) * W + 4 * pair) * 4;
        }
    }
    const int x_row0 = tid >> 6, x_col = (tid >> 1) & 31;
    const int x_lds0 = ((x_row0 + 3) * SF_XW + x_col + 4) * XPIX + (tid & 1) * 16;
    const int x_rel0 = (x_row0 * W2 + x_col) * 32 + (tid & 1) * 16;

    // ---- per-lane constants of the pooling layout -------------------------------------------------------
    // stem pixel (row 4*wave-1+rho, column 2*r+par) of the tile reads s2d pixels (4*wave+rho+ty, 2*r+par+2+tx); a lane's k-group
    // is tap (sl>>1, 2*(sl&1) + (gq>>1)), channel half gq&1
    // (SK6 order: the lane group's k-group of k-step s is q = 4*s + gq, at mil_sk6_off(q) from the record under tap (0,0))
    const int lane_base = (4 * wave * SF_XW + 2 * r + 2) * XPIX;
    const int rE = r < 4 ? r : 4;                               // edge tile: pixel lane = stem row of the wave (lanes 5.. repeat row 4)
    const int laneE_delta = ((4 * wave + rE) * SF_XW + 1) * XPIX - lane_base;
    int kaddr[KSTEPS];                                          // lane_base + the k-group's offset, per k-step
#pragma unroll
    for (int sl = 0; sl < KSTEPS; ++sl) {
        int o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (gq == k) o = mil_sk6_off(4 * sl + k, SF_XW * XPIX, XPIX);
        kaddr[sl] = lane_base + o;
    }
    // position codes: (stem row & 3) << 2 | (stem column & 3); tile origins are multiples of 4 in both directions
    const unsigned cc_even = 2u * (r & 1), cc_odd = cc_even + 1u;
    const unsigned cc_edge = ((unsigned)((rE + 3) & 3) << 2) | 3u;
    // code -> tap ky*3+kx tables, one 64-bit word per (pooled row of the wave, lane parity): kept in LDS (read once per pooled row)
    // the bias is added to the POOLED value (max(v) + b = max(v + b)): the accumulators start at zero and the 32 floats wait in LDS
    char* ldsB = smem + NBUF * BUFSTRIDE;
    if (tid < 32) reinterpret_cast<float*>(ldsB)[tid] = a.bias ? a.bias[tid] : 0.f;
    if (tid >= 64 && tid < 68) {
        const int p = (tid >> 1) & 1, odd = tid & 1;
        unsigned long long lut = 0ull;
        for (int code = 0; code < 16; ++code) {
            const int ky = ((code >> 2) + (p ? 3 : 1)) & 3, kx = ((code & 3) + (odd ? 3 : 1)) & 3;
            lut |= ((ky < 3 && kx < 3) ? (unsigned long long)(ky * 3 + kx) : 0ull) << (4 * code);
        }
        reinterpret_cast<unsigned long long*>(ldsB + 128)[p * 2 + odd] = lut;
    }
    constexpr bool WLDS = sp_wlds<X3>();
    char* ldsW = ldsB + 160;
    if constexpr (WLDS) mil_stage_filter(ldsW, static_cast<const char*>(a.w) + W_OFF, KSTEPS * NT * 64 * FRAGB, tid, NTHR);
    float sentv = -3.0e38f;
    asm volatile("" : "+v"(sentv));                             // a VGPR: v_cndmask_b32_e64 has one constant-bus slot, taken by the mask
    const float slope = a.slope;

    // ---- tile walk: XCD x (= blockIdx & 7) owns tiles [x*per, (x+1)*per) ---------------------------
    const int G8 = gridDim.x >> 3, per = (a.ntiles + 7) >> 3;
    const int t_begin = (blockIdx.x & 7) * per;
    const int t_end = min(t_begin + per, a.ntiles);
    int tile = t_begin + (blockIdx.x >> 3);

    u32x4_t r0[NLOAD], r1[FROM_XS ? 1 : NLOAD];
    auto fetch = [&](int t) {
        const int tx = t % a.tiles_x, q = t / a.tiles_x, ty = q % a.tiles_y, img = q / a.tiles_y;
        const int y0 = 2 * PH * ty - 3, c0 = 64 * tx - 8;
        if constexpr (FROM_XS) {
            const int x0 = 32 * tx - 4;
            const int base = ((img * H2 + y0) * W2 + x0) * 32;
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                const int row = (l_lds[i] >> 18) & 31, col = l_lds[i] >> 24;
                const bool ok = (unsigned)(y0 + row) < (unsigned)H2 && (unsigned)(x0 + col) < (unsigned)W2;
                r0[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(base + l_rel[i]) : MIL_OOB, 0, 0);
            }
            return;
        }
        const int base = (((img * 3) * H + 2 * y0) * W + c0) * 4;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int row = (l_lds[i] >> 18) & 31, pair = l_lds[i] >> 24;
            const bool ok = (unsigned)(y0 + row) < (unsigned)H2 && (unsigned)(c0 + 4 * pair) < (unsigned)W;
            const unsigned off = ok ? (unsigned)(base + l_rel[i]) : MIL_OOB;
            r0[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
            r1[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off + (unsigned)(W * 4), 0, 0);
        }
    };
    if (tile < t_end) fetch(tile);
    __syncthreads();
    int buf = 0;
    MIL_STAMP_DECL(5)
    for (; tile < t_end; tile += G8) {
        const int tx = tile % a.tiles_x, tq = tile / a.tiles_x, ty = tq % a.tiles_y, img = tq / a.tiles_y;
        char* ldsX = smem + buf * BUFSTRIDE;
        MIL_STAMP_BEGIN()
        // ---- s2d tile: fp32 -> bf16 (hi / lo planes in split precision), channel = c*4 + dy*2 + dx ---------------------
        if constexpr (FROM_XS) {
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                // half 0 = [c0 c1]: 16 bytes as they are; half 1 = [c2 | padding]: its c2 also goes into the padding bytes of the
                // pixel to the left (that pixel's "c2 of the next pixel"), and only 8 bytes into its own record
                char* dst = ldsX + (l_lds[i] & 0xFFFF);
                if (tid & 1) {
                    const u32x2_t c2 = u32x2_t{r0[i][0], r0[i][1]};
                    const bool first = (l_lds[i] >> 24) == 0;          // column 0: its left neighbour lies outside the tile
                    *reinterpret_cast<u32x2_t*>(dst) = c2;
                    *reinterpret_cast<u32x2_t*>(first ? ldsX + dump : dst - XPIX + 8) = c2;
                } else {
                    *reinterpret_cast<u32x4_t*>(dst) = r0[i];
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const f32x4_t v0 = __builtin_bit_cast(f32x4_t, r0[i]), v1 = __builtin_bit_cast(f32x4_t, r1[i]);
            const float fa[4] = {v0[0], v0[1], v1[0], v1[1]}, fb[4] = {v0[2], v0[3], v1[2], v1[3]};
            bf16x4_t pa, pb, qa, qb;
            if constexpr (X3) {
                mil_split4(f32x4_t{fa[0], fa[1], fa[2], fa[3]}, pa, qa);
                mil_split4(f32x4_t{fb[0], fb[1], fb[2], fb[3]}, pb, qb);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { pa[j] = (__bf16)fa[j]; pb[j] = (__bf16)fb[j]; }
            }
            char* dst = ldsX + (l_lds[i] & 0xFFFF);
            *reinterpret_cast<bf16x4_t*>(dst) = pa;
            *reinterpret_cast<bf16x4_t*>(dst + XPIX) = pb;
            // colour 2 (s2d channels 8-11) a second time: behind the c2 of the pixel to the left ("c2 of the next pixel", SK6 order)
            const bool c2 = (l_lds[i] >> 16) & 1;
            char* dupb = c2 ? dst + 8 : ldsX + dump;                   // pixel 2*pair  <- c2 of pixel 2*pair + 1
            char* dupa = (c2 && (l_lds[i] >> 24) != 0) ? dst - XPIX + 8 : ldsX + dump;      // pixel 2*pair - 1 <- c2 of pixel 2*pair (pair 0: outside the tile)
            *reinterpret_cast<bf16x4_t*>(dupb) = pb;
            *reinterpret_cast<bf16x4_t*>(dupa) = pa;
            if constexpr (X3) {
                *reinterpret_cast<bf16x4_t*>(dst + 32) = qa;
                *reinterpret_cast<bf16x4_t*>(dst + XPIX + 32) = qb;
                *reinterpret_cast<bf16x4_t*>(dupb + 32) = qb;
                *reinterpret_cast<bf16x4_t*>(dupa + 32) = qa;
            }
        }
        MIL_STAMP_MARK(0)
        __syncthreads();
        MIL_STAMP_MARK(1)
        if (MIL_SP_PRIO) __builtin_amdgcn_s_setprio(0);
        const bool has_next = tile + G8 < t_end;
        if (!X3 && !FROM_XS && a.xs) {       // the tile's own 16x32 s2d pixels go to the xs tensor (when the caller keeps one)
            const int xbase = ((img * H2 + 16 * ty) * W2 + 32 * tx) * 32;
            const int ylim = H2 - 16 * ty, xlim = W2 - 32 * tx;
#pragma unroll
            for (int i = 0; i < NXS; ++i) {
                constexpr int RPI = NTHR / 64;
                const bool ok = x_row0 + RPI * i < (ylim < 16 ? ylim : 16) && x_col < xlim;
                u32x4_t v = *reinterpret_cast<const u32x4_t*>(ldsX + x_lds0 + i * (RPI * SF_XW * XPIX));
                if (tid & 1) { v[2] = 0u; v[3] = 0u; }             // the record's padding channels hold the next pixel's c2 in LDS only
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_xs, ok ? (unsigned)(xbase + x_rel0 + i * (RPI * W2 * 32)) : MIL_OOB, 0, 0);
            }
        }
        // ---- 4x4 s1 implicit GEMM over the s2d tile, D[channel][pixel]: 5 rows x (even, odd) + the edge tile ------------------
        f32x4_t acc[11][NT];                 // pixel tile m = 2 * rho + parity (rho = stem row of the wave), m = 10: the edge tile
#pragma unroll
        for (int m = 0; m < 11; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        {
            #ifndef MIL_SP_WD
#define MIL_SP_WD 2
#endif
#ifndef MIL_SP_LA
#define MIL_SP_LA 2
#endif
#ifndef MIL_SP_EXP
#define MIL_SP_EXP 0                  // ablations (development): 1 = no MFMA loop, 2 = no horizontal maxima, 4 = no decode / stores
#endif
            constexpr int MT = 11, TOT = (MIL_SP_EXP & 1) ? 0 : KSTEPS * MT, LA = X3 ? 1 : MIL_SP_LA, R = LA + 1;
            constexpr int WD = X3 ? 1 : MIL_SP_WD, WR = WD + 1;           // filter fragments: k-steps ahead / ring slots
            // the step that issues the last filter loads: the next tile's input is requested behind them (vector loads return
            // in order).  Split precision has no registers for it inside the loop (two operand planes): requested behind the loop.
            constexpr int FETCH_AT = X3 ? -1 : (WLDS ? 0 : (KSTEPS - 1 - WD) * MT);
            Frag8<T> ring[R], wq[WR][NT];
#ifndef MIL_SP_X3_FOLD
#define MIL_SP_X3_FOLD 1              // split precision, column tile 1 (channels 16-19): [wh ; wl] x xh, then wh x xl (two MFMAs + a lane exchange in the
#endif                                // epilogue) instead of three MFMAs; measured 1395 vs 1425 us per launch (round 5)
            constexpr bool FOLD = X3 && MIL_SP_X3_FOLD;
            const int w1h = (FOLD && r >= 4 && r < 8) ? (64 + lane - 4) * FRAGB + 16 : (64 + lane) * FRAGB;
            auto wfrag = [&](int sl, int nt) {
                Frag8<T> f;
                if constexpr (X3) {       // column tile 1: rows 4-7 carry the LO halves of channels 16-19 ([wh ; wl] x xh, then wh x xl)
                    f.h = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(nt == 1 ? w1h : lane * FRAGB), sl * NT * 64 * FRAGB, 0));
                    f.l = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(nt == 1 ? (64 + lane) * FRAGB + (FOLD ? 0 : 16) : lane * FRAGB + 16), sl * NT * 64 * FRAGB, 0));
                } else if constexpr (WLDS) {
                    f.v = *reinterpret_cast<const bf16x8_t*>(ldsW + ((sl * NT + nt) * 64 + lane) * FRAGB);
                } else {
                    f.v = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(lane * FRAGB), (sl * NT + nt) * 64 * FRAGB, 0));
                }
                return f;
            };
            auto xaddr = [&](int j) -> const char* {
                const int sl = j / MT, m = j % MT;
                return m < 10 ? ldsX + kaddr[sl] + ((m >> 1) * SF_XW + (m & 1)) * XPIX : ldsX + kaddr[sl] + laneE_delta;
            };
#pragma unroll
            for (int k = 0; k < WD; ++k)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wq[k][nt] = wfrag(k, nt);
#pragma unroll
            for (int j = 0; j < LA; ++j) ring[j % R] = lds_pix_frag<T, 32>(xaddr(j));
#pragma unroll
            for (int j = 0; j < TOT; ++j) {
                const int sl = j / MT, m = j % MT;
                if (j + LA < TOT) ring[(j + LA) % R] = lds_pix_frag<T, 32>(xaddr(j + LA));
                if (m == 0 && sl + WD < KSTEPS) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) wq[(sl + WD) % WR][nt] = wfrag(sl + WD, nt);
                }
                if (j == FETCH_AT) {
                    if (has_next) fetch(tile + G8);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (FOLD) {
                    acc[m][0] = mma8(wq[sl % WR][0], ring[j % R], acc[m][0]);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[sl % WR][1].h, ring[j % R].h, acc[m][1], 0, 0, 0);      // [wh ; wl] x xh
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[sl % WR][1].l, ring[j % R].l, acc[m][1], 0, 0, 0);      // wh x xl
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mma8(wq[sl % WR][nt], ring[j % R], acc[m][nt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (X3) {
            if (has_next) fetch(tile + G8);
        }
        if (MIL_SP_PRIO) __builtin_amdgcn_s_setprio(MIL_SP_PRIO);
        // the hand-written vector instructions below read MFMA results: the compiler pads its own, not those inside asm
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        MIL_STAMP_MARK(2)
        if constexpr (X3 && MIL_SP_X3_FOLD) {      // column tile 1: rows 4-7 (lane group 1) hold w_lo * x_hi of rows 0-3 — add them, zero the padding channels
#pragma unroll
            for (int m = 0; m < 11; ++m) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t0 = acc[m][1][i], t1 = t0;
                    mil_swap16<false>(t0, t1);                         // t1 of lane group 0 = group 1's value
                    acc[m][1][i] = gq == 0 ? acc[m][1][i] + t1 : 0.f;
                }
            }
        }
        // ---- position codes, horizontal maxima per stem row, vertical maximum per pooled row ------------------------------
        const int S0 = 2 * PH * ty + 4 * wave - 1;               // image row of the wave's stem row rho = 0
        const bool xedge = 32 * tx + 32 > W2;                    // (rare) stem columns of this tile beyond the image
        const bool ev_in = 32 * tx + 2 * r < W2, od_in = 32 * tx + 2 * r + 1 < W2;
        float ek[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ek[q] = sp_key(acc[10][q >> 2][q & 3], cc_edge);
        float h[5][8];
        auto hrow = [&](auto RHO) {                              // compile-time row index: h[][] stays in registers
            constexpr int rho = decltype(RHO)::value;
            constexpr unsigned rc = (unsigned)((rho + 3) & 3) << 2;
            const unsigned ce = rc | cc_even, co = rc | cc_odd;
            float o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                h[rho][q] = sp_key(acc[2 * rho][q >> 2][q & 3], ce);
                o[q] = sp_key(acc[2 * rho + 1][q >> 2][q & 3], co);
            }
            if (xedge) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { h[rho][q] = ev_in ? h[rho][q] : sentv; o[q] = od_in ? o[q] : sentv; }
            }
            if (!(MIL_SP_EXP & 2)) sp_hmax8(h[rho], o);
            if (tx > 0 && !(MIL_SP_EXP & 2)) {                   // column -1 of the image is padding: nothing to add
                constexpr unsigned long long mask = rho == 0 ? 0x0001000100010001ull : 0x1111111111111111ull << (rho & 3);
                sp_edge8<rho>(h[rho], ek, sentv, mask);
            }
            if ((unsigned)(S0 + rho) >= (unsigned)H2) {          // a stem row outside the image (the pool's padding)
#pragma unroll
                for (int q = 0; q < 8; ++q) h[rho][q] = sentv;
            }
        };
        hrow(std::integral_constant<int, 0>{}); hrow(std::integral_constant<int, 1>{}); hrow(std::integral_constant<int, 2>{});
        hrow(std::integral_constant<int, 3>{}); hrow(std::integral_constant<int, 4>{});
        MIL_STAMP_MARK(3)
        // ---- winner decode, + bias, LeakyReLU, store ------------------------------------------------------------------------
        const f32x4_t bias0 = *reinterpret_cast<const f32x4_t*>(ldsB + gq * 16), bias1 = *reinterpret_cast<const f32x4_t*>(ldsB + 64 + gq * 16);
#pragma unroll
        for (int p = 0; p < ((MIL_SP_EXP & 4) ? 0 : 2); ++p) {
            const int py = PH * ty + 2 * wave + p, px = 16 * tx + r;
            const bool ok = py < Ho && px < Wo;
            const unsigned pix = (unsigned)((img * Ho + py) * Wo + px);
            float y[8];
            unsigned rec[2] = {0u, 0u};
            const unsigned long long lut = *reinterpret_cast<const unsigned long long*>(ldsB + 128 + (p * 2 + (r & 1)) * 8);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float K = sp_max3(h[2 * p][q], h[2 * p + 1][q], h[2 * p + 2][q]);
                const unsigned kb = __float_as_uint(K);
                const unsigned k = (unsigned)(lut >> ((kb << 2) & 60u)) & 15u;
                const float v = (X3 ? __uint_as_float((kb & ~15u) | 8u) : K) + (q < 4 ? bias0[q & 3] : bias1[q & 3]);
                rec[q >> 2] |= (k | ((__float_as_uint(v) >> 27) & 16u)) << (8 * (q & 3));
                y[q] = sp_lrelu(v, slope);
            }
            const bool ok1 = ok && gq < 2;                        // column tile 1: channels 16-19 (gq 0), padding 20-23 (gq 1)
            if (gq != 0) {
#pragma unroll
                for (int q = 4; q < 8; ++q) y[q] = 0.f;
                rec[1] = 0x10101010u;                             // what a zero activation records
            }
            if constexpr (X3) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{y[0], y[1], y[2], y[3]}), rs_p,
                                                       ok ? pix * 96u + (unsigned)gq * 16u : MIL_OOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{y[4], y[5], y[6], y[7]}), rs_p,
                                                       ok1 ? pix * 96u + 64u + (unsigned)gq * 16u : MIL_OOB, 0, 0);
            } else {
                bf16x4_t o0, o1;
#pragma unroll
                for (int i = 0; i < 4; ++i) { o0[i] = (__bf16)y[i]; o1[i] = (__bf16)y[4 + i]; }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, o0), rs_p, ok ? pix * 48u + (unsigned)gq * 8u : MIL_OOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, o1), rs_p, ok1 ? pix * 48u + 32u + (unsigned)gq * 8u : MIL_OOB, 0, 0);
            }
            __builtin_amdgcn_raw_buffer_store_b32(rec[0], rs_i, ok ? pix * 24u + (unsigned)gq * 4u : MIL_OOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(rec[1], rs_i, ok1 ? pix * 24u + 16u + (unsigned)gq * 4u : MIL_OOB, 0, 0);
        }
        if (MIL_SP_EXP & 4) {                                    // ablation: keep the maxima alive without the decode / stores
            float sacc = 0.f;
#pragma unroll
            for (int rho = 0; rho < 5; ++rho)
#pragma unroll
                for (int q = 0; q < 8; ++q) sacc += h[rho][q];
            if (sacc == 1234.5f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sacc), rs_i, 0, 0, 0);
        }
        MIL_STAMP_MARK(4)
        if constexpr (NBUF == 1) __syncthreads();                // every wave is done reading the s2d tile
        else buf ^= 1;
    }
    MIL_STAMP_STORE(a.stamp, 4)
}

template <bool X3, bool FROM_XS>
static int launch_stem_fwd_pool(StemFwdArgs a, hipStream_t st) {
    constexpr int COUTP = 24, OESZ = X3 ? 4 : 2;
    const int lds = sp_lds_bytes<X3>();
    a.tiles_y = (a.Ho + 7) / 8;
    auto kern = stem_fwd_pool_kernel<X3, FROM_XS>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return MIL_ERR_LAUNCH;
    size_t per_img = FROM_XS ? 0 : (size_t)3 * a.H * a.W * 4;
    const size_t xs_img = X3 ? 0 : (size_t)a.H2 * a.W2 * 32, p_img = (size_t)a.Ho * a.Wo * COUTP * OESZ;
    if (xs_img > per_img) per_img = xs_img;
    if (p_img > per_img) per_img = p_img;
    const int chunk = mil_imgs_under_2g(per_img);
    const int n_total = a.n_img;
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        StemFwdArgs b = a;
        b.n_img = n_total - i0 < chunk ? n_total - i0 : chunk;
        b.x = a.x ? a.x + (size_t)i0 * 3 * a.H * a.W : nullptr;
        b.xs_in = a.xs_in ? a.xs_in + (size_t)i0 * a.H2 * a.W2 * 16 : nullptr;
        b.xs = a.xs ? a.xs + (size_t)i0 * a.H2 * a.W2 * 16 : nullptr;
        b.pool = static_cast<char*>(a.pool) + (size_t)i0 * a.Ho * a.Wo * COUTP * OESZ;
        b.widx = a.widx + (size_t)i0 * a.Ho * a.Wo * COUTP;
        b.ntiles = b.n_img * a.tiles_y * a.tiles_x;
        int grid = (b.ntiles + 7) & ~7;
        int per_cu = mil_resident_per_cu(kern, lds, 2);
        if (const char* e = mil_ab_env("MIL_STEM_WGS")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;      // development builds only
        const int cap = mil_num_cus() * per_cu;
        if (grid > cap) grid = cap;
#ifdef MIL_STAMP
        static MilStampBuf sb;
        b.stamp = sb.get((size_t)grid * 4 * 7);
#endif
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, b);
        MIL_CHECK_LAUNCH();
#ifdef MIL_STAMP
        static const char* const ph[5] = {"convert", "barrier-x", "gemm", "maxima", "decode-store"};
        sb.report(X3 ? "stem_fwd_pool_kernel<x3>" : "stem_fwd_pool_kernel", grid, 4, 5, ph, st);
#endif
    }
    return MIL_OK;
}

#include "stem_walk.cuh"

template <int NT, bool X3 = false, bool FROM_XS = false>
static int launch_stem_fwd(StemFwdArgs a, hipStream_t st) {
    constexpr int COUTP = mil_nt_to_cp(NT);
    constexpr int OESZ = X3 ? 4 : 2;
    const int lds = sf_lds_bytes<NT, X3>();
    // measured (24 channels, bf16): 1120 us with 4 waves per workgroup at 238 VGPRs, 1428 us with 8 waves squeezed into 128
    // VGPRs (15 spilled): the 8-wave form serves the split-precision kernel, whose 156 KB of LDS leave one workgroup per CU
    // (two waves per SIMD at up to 256 VGPRs)
#ifndef MIL_STEM_FWD_WAVES
#define MIL_STEM_FWD_WAVES 4
#endif
    constexpr int NW = X3 ? (sf_wstream(true) ? 4 : 8) : (NT <= 2 ? MIL_STEM_FWD_WAVES : 4);
    a.tiles_y = (a.Ho + sf_ph(X3) - 1) / sf_ph(X3);
    auto kern = stem_fwd_fused_kernel<NT, NW, X3, FROM_XS>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return MIL_ERR_LAUNCH;
    // every tensor is addressed with 32-bit offsets below 2 GiB: split the launch by images
    size_t per_img = FROM_XS ? 0 : (size_t)3 * a.H * a.W * 4;
    const size_t xs_img = X3 ? 0 : (size_t)a.H2 * a.W2 * 32, p_img = (size_t)a.Ho * a.Wo * COUTP * OESZ;
    if (xs_img > per_img) per_img = xs_img;
    if (p_img > per_img) per_img = p_img;
    const int chunk = mil_imgs_under_2g(per_img);
    const int n_total = a.n_img;
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        StemFwdArgs b = a;
        b.n_img = n_total - i0 < chunk ? n_total - i0 : chunk;
        b.x = a.x ? a.x + (size_t)i0 * 3 * a.H * a.W : nullptr;
        b.xs_in = a.xs_in ? a.xs_in + (size_t)i0 * a.H2 * a.W2 * 16 : nullptr;
        b.xs = a.xs ? a.xs + (size_t)i0 * a.H2 * a.W2 * 16 : nullptr;
        b.pool = static_cast<char*>(a.pool) + (size_t)i0 * a.Ho * a.Wo * COUTP * OESZ;
        b.widx = a.widx + (size_t)i0 * a.Ho * a.Wo * COUTP;
        b.ntiles = b.n_img * a.tiles_y * a.tiles_x;
        int grid = (b.ntiles + 7) & ~7;
        const int cap = mil_num_cus() * ((NT <= 2 && (!X3 || sf_wstream(true))) ? 2 : 1);
        if (grid > cap) grid = cap;
#ifdef MIL_STAMP
        static MilStampBuf sb;
        b.stamp = sb.get((size_t)grid * NW * 9);
#endif
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, st, b);
        MIL_CHECK_LAUNCH();
#ifdef MIL_STAMP
        static const char* const ph[7] = {"convert", "barrier-x", "fetch-issue", "gemm", "stem-store", "barrier-s", "pool"};
        sb.report(X3 ? "stem_fwd_fused_kernel<x3>" : "stem_fwd_fused_kernel", grid, NW, 7, ph, st);
#endif
    }
    return MIL_OK;
}

// xs = s2d(x); pool, widx = maxpool(lrelu(conv7x7s2(x) + bias)).  bf16 only; needs H even, W a multiple of 4 and a
// 16-byte aligned x (otherwise MIL_ERR_UNSUPPORTED: the caller runs mil_stem_s2d / mil_conv_igemm / mil_maxpool_fwd).
// xs may be null: no space-to-depth copy is kept (1.07 GB less written at 2048 tiles of 256x256) and the backward
// rebuilds its s2d tiles from x itself (mil_stem_bwd_fused_nchw).
extern "C" int mil_stem_fwd_fused(const float* x_nchw, const void* wpack, const float* bias_pad, void* xs, void* pool,
                                  uint8_t* widx, int n_img, int H, int W, int cout_p, float slope, int dtype, void* stream) {
    if (!x_nchw || !wpack || !pool || !widx || n_img < 0 || H <= 0 || W <= 0) return MIL_ERR_ARG;
    if ((dtype != MIL_DT_BF16 && dtype != MIL_DT_F32S) || (H & 1) || (W & 3) || (reinterpret_cast<uintptr_t>(x_nchw) & 15) || slope < 0.f || slope >= 1.f)
        return MIL_ERR_UNSUPPORTED;
    if (cout_p != 24 && cout_p != 64) return MIL_ERR_UNSUPPORTED;
    if (dtype == MIL_DT_F32S && (cout_p != 24 || xs)) return MIL_ERR_UNSUPPORTED;      // split precision: no s2d copy (the backward reads x), 20-channel stem
    if (n_img == 0) return MIL_OK;
    StemFwdArgs a{};
    a.x = x_nchw; a.w = wpack; a.bias = bias_pad; a.xs = (__bf16*)xs; a.pool = pool; a.widx = widx;
    a.n_img = n_img; a.H = H; a.W = W; a.H2 = H / 2; a.W2 = W / 2;
    a.Ho = (a.H2 - 1) / 2 + 1; a.Wo = (a.W2 - 1) / 2 + 1;
    a.tiles_y = (a.Ho + 7) / 8; a.tiles_x = (a.Wo + 15) / 16;
    a.slope = slope;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#ifndef MIL_STEM_FWD_POOL
#define MIL_STEM_FWD_POOL 1           // 0: the round-2..4 kernel (stem tile through LDS, bf16-rounded activations pooled)
#endif
    if (MIL_STEM_FWD_POOL && cout_p == 24) {
        if (mil_stem_walk_wanted(a, mil_num_cus() * 2))
            return dtype == MIL_DT_F32S ? launch_stem_fwd_walk<true>(a, st) : launch_stem_fwd_walk<false>(a, st);
        return dtype == MIL_DT_F32S ? launch_stem_fwd_pool<true, false>(a, st) : launch_stem_fwd_pool<false, false>(a, st);
    }
    if (dtype == MIL_DT_F32S) return launch_stem_fwd<2, true>(a, st);
    return cout_p == 24 ? launch_stem_fwd<2>(a, st) : launch_stem_fwd<4>(a, st);
}

// The same pass fed by the bf16 space-to-depth tensor xs [n,H2,W2,16] (mil_tile_preprocess_s2d's output, or mil_stem_s2d's):
// pool / widx are bit-identical to mil_stem_fwd_fused on the fp32 tiles xs was made from.  bf16 only; H2, W2 = dims of xs.
extern "C" int mil_stem_fwd_fused_xs(const void* xs, const void* wpack, const float* bias_pad, void* pool, uint8_t* widx,
                                     int n_img, int H2, int W2, int cout_p, float slope, int dtype, void* stream) {
    if (!xs || !wpack || !pool || !widx || n_img < 0 || H2 <= 0 || W2 <= 0) return MIL_ERR_ARG;
    if (dtype != MIL_DT_BF16 || (reinterpret_cast<uintptr_t>(xs) & 15) || slope < 0.f || slope >= 1.f) return MIL_ERR_UNSUPPORTED;
    if (cout_p != 24 && cout_p != 64) return MIL_ERR_UNSUPPORTED;
    if (n_img == 0) return MIL_OK;
    StemFwdArgs a{};
    a.xs_in = (const __bf16*)xs; a.w = wpack; a.bias = bias_pad; a.pool = pool; a.widx = widx;
    a.n_img = n_img; a.H = 2 * H2; a.W = 2 * W2; a.H2 = H2; a.W2 = W2;
    a.Ho = (H2 - 1) / 2 + 1; a.Wo = (W2 - 1) / 2 + 1;
    a.tiles_y = (a.Ho + 7) / 8; a.tiles_x = (a.Wo + 15) / 16;
    a.slope = slope;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (MIL_STEM_FWD_POOL && cout_p == 24) return launch_stem_fwd_pool<false, true>(a, st);
    return cout_p == 24 ? launch_stem_fwd<2, false, true>(a, st) : launch_stem_fwd<4, false, true>(a, st);
}
