// Forward of a whole identity-shortcut residual block in one pass (bf16 path, 24- and 40-channel stages):
//   o1  = lrelu(conv3x3(x) + b1)                  nnBlocks.py:176-177
//   out = lrelu(conv3x3(o1) + b2 + x)             nnBlocks.py:179-189 (identity shortcut, in-place add)
// The two persistent conv launches this replaces are at the device-copy bandwidth on the 64x64 maps (the forward
// convs move 2 and 3 tensors per launch); fused, the block input is read once (it is both conv1's operand and the
// residual), the mid activation goes from the accumulators to an LDS tile that conv2 reads directly, and only what
// the backward needs is written: o1 and out.  5 tensor passes become 3.
//
// Tile = 16x16 output pixels of one image <- 18x18 mid pixels (conv1 is recomputed on the one-pixel ring, +27% of
// its MFMA work, which is not the bottleneck) <- 20x20 input pixels.  Mid pixels outside the image are conv2's zero
// padding and are written as zeros.  Structure otherwise as conv_igemm_pf_kernel: persistent workgroups, filters
// resident in LDS, register prefetch of the next input halo, D[channel][pixel] accumulators, paired 16-byte epilogue.
#include "pf_common.cuh"
#include <cstdlib>

#ifndef MIL_BLOCK_LOOKAHEAD
#define MIL_BLOCK_LOOKAHEAD 2      // pixel fragments read this many (k-step, row tile) steps ahead of their MFMAs; 0 = compiler order
#endif

struct BlockFwdArgs {
    const __bf16* x;        // [n,H,W,CP]
    const __bf16* w1;       // MIL_PACK_FWD fragments [KSTEPS][NT][64][8]
    const __bf16* w2;
    const float* b1;        // [NT*16]
    const float* b2;
    __bf16* o1;             // [n,H,W,CP]
    __bf16* y;              // [n,H,W,CP]
    ConvGeom g;             // 16x16 tiles; halo described as a 5x5 / pad 2 window (20x20 input pixels)
    int lds_o_off, lds_w_off, lds_dump_off;
    float slope;
};

// NW waves per workgroup (4 or 8).  With 8 waves every per-wave quantity halves (3 + 2 row tiles of accumulators, 3 halo
// pieces in flight), the kernel fits 128 VGPRs and a CU holds 16 waves instead of 8 on the same LDS tiles and filters:
// kept as an experiment (MIL_BLOCK_WAVES=8): it measured no faster, see mil_block_waves().
template <int CP, int NT, int NW>
__global__ __launch_bounds__(64 * NW, (CP <= 24 ? 2 : 1) * (NW == 8 ? 2 : 1)) void conv_block_fwd_kernel(BlockFwdArgs a, int ntiles, unsigned bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int PIXB = mil_pix_pitch(CP, 2);
    constexpr int CG = CP / 8;
    constexpr int KSTEPS = (9 * CG + 3) / 4;
    constexpr int NTHR = 64 * NW;
    constexpr int NPX = (400 * CG + NTHR - 1) / NTHR;
    constexpr int MT1 = 24 / NW;                             // NW waves x MT1 row tiles x 16 = 384 >= 324 mid pixels
    constexpr int MT2 = 16 / NW;                             // 256 output pixels
    constexpr int NPAIR = MT2 / 2;
    constexpr int NPC = (256 * CG + NTHR - 1) / NTHR;
    constexpr bool LAST_PARTIAL = (CP % 16) != 0;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    char* ldsX = smem;
    char* ldsO = smem + a.lds_o_off;
    char* ldsW1 = smem + a.lds_w_off;
    char* ldsW2 = ldsW1 + KSTEPS * NT * 64 * 16;
    mil_stage_filter(ldsW1, a.w1, KSTEPS * NT * 64 * 16, tid, NTHR);
    mil_stage_filter(ldsW2, a.w2, KSTEPS * NT * 64 * 16, tid, NTHR);
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, bytes);
    const __amdgpu_buffer_rsrc_t rs_o = mil_rsrc(a.o1, bytes);
    const __amdgpu_buffer_rsrc_t rs_y = mil_rsrc(a.y, bytes);
    const int H = g.H, W = g.W;

    // ---- tile-invariant tables --------------------------------------------------------------------
    HaloTables<NPX> ht;                                      // 20x20 input halo pieces: id = tid + NTHR*i -> (row, col, piece)
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int idx = tid + NTHR * i;
        ht.pos[i] = -1; ht.lds[i] = a.lds_dump_off; ht.rel[i] = 0;
        if (idx < 400 * CG) {
            const int px = idx / CG, j = idx - px * CG, hy = px / 20, hx = px - hy * 20;
            ht.pos[i] = (hy << 10) | hx;
            ht.lds[i] = px * PIXB + j * 16;
            ht.rel[i] = (hy * W + hx) * (CP * 2) + j * 16;
        }
    }
    int toff1[KSTEPS], toff2[KSTEPS];
#pragma unroll
    for (int sl = 0; sl < KSTEPS; ++sl) {
        const int q = 4 * sl + gq;
        int tap = q / CG, cg = q - tap * CG;
        if (tap >= 9) { tap = 0; cg = 0; }
        toff1[sl] = ((tap / 3) * 20 + (tap % 3)) * PIXB + cg * 16;
        toff2[sl] = ((tap / 3) * 18 + (tap % 3)) * PIXB + cg * 16;
    }
    int pixbase1[MT1], sdst1[MT1];                           // conv1: mid pixel (py,px) reads input halo (py+ky, px+kx)
#pragma unroll
    for (int i = 0; i < MT1; ++i) {
        const int tp = (wave + NW * i) * 16 + r;
        const bool ok = tp < 324;
        const int py = tp / 18, px = tp - py * 18;
        pixbase1[i] = ok ? (py * 20 + px) * PIXB : 0;
        sdst1[i] = ok ? a.lds_o_off + tp * PIXB + gq * 8 : a.lds_dump_off;
    }
    int pixbase2[MT2];                                       // conv2: output pixel (ty,tx) reads mid (ty+ky, tx+kx)
#pragma unroll
    for (int m = 0; m < MT2; ++m) {
        const int tp = (wave * MT2 + m) * 16 + r;
        pixbase2[m] = ((tp >> 4) * 18 + (tp & 15)) * PIXB;
    }
    int o_rel[NPAIR], o_pos[NPAIR], xres[NPAIR];             // paired epilogue: pixel (2p + (gq&1), r) of this wave's row tiles
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
        const int tp = (wave * MT2 + 2 * p + (gq & 1)) * 16 + r;
        const int ty = tp >> 4, tx = tp & 15;
        o_rel[p] = (ty * W + tx) * (CP * 2) + (gq >> 1) * 16;
        o_pos[p] = (ty << 10) | tx;
        xres[p] = ((ty + 2) * 20 + tx + 2) * PIXB + (gq >> 1) * 16;          // the residual = centre of the input halo tile
    }
    // mid-tile centre pieces -> o1 tensor: piece id = tid + 256*i -> (pixel id / CG, piece id % CG)
    int c_lds[NPC], c_rel[NPC], c_pos[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int id = tid + NTHR * i, px = id / CG, j = id - px * CG;
        const int ty = px >> 4, tx = px & 15;
        const bool used = id < 256 * CG;
        c_lds[i] = used ? ((ty + 1) * 18 + tx + 1) * PIXB + j * 16 : 0;
        c_rel[i] = (ty * W + tx) * (CP * 2) + j * 16;
        c_pos[i] = used ? (ty << 10) | tx : (1023 << 10);          // an unused slot is never inside the image
    }
    const bool last_ok = !LAST_PARTIAL || (gq >> 1) == 0;
    f32x4_t b1r[NT], b2r[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b1r[nt][i] = a.b1 ? a.b1[nt * 16 + gq * 4 + i] : 0.f;
            b2r[nt][i] = a.b2 ? a.b2[nt * 16 + gq * 4 + i] : 0.f;
        }

    TileWalker cur, nxt;
    const int bid = mil_xcd_block_id();
    cur.init(g, bid, gridDim.x);
    nxt = cur; nxt.advance();
    u32x4_t rx[NPX];
    auto fetch = [&](const TileOrigin& o) {
        const int iy0 = o.oy0 - 2, ix0 = o.ox0 - 2;
        const int base = ((o.img0 * H + iy0) * W + ix0) * (CP * 2);
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int p = ht.pos[i];
            const bool ok = p >= 0 && (unsigned)(iy0 + (p >> 10)) < (unsigned)H && (unsigned)(ix0 + (p & 1023)) < (unsigned)W;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(base + ht.rel[i]) : MIL_OOB, 0, 0);
        }
    };
    if (bid < ntiles) fetch(cur.origin(g));
    const int G = gridDim.x;
    for (int tile = bid; tile < ntiles; tile += G) {
        const TileOrigin o = cur.origin(g);
        __syncthreads();                       // previous tile: residual reads of ldsX and conv2's reads of ldsO are done
        mil_commit_halo_all<NPX>(rx, ldsX, ht);
        __syncthreads();
        if (tile + G < ntiles) fetch(nxt.origin(g));
        cur = nxt; nxt.advance();

        // ---- conv1 on the 18x18 mid tile -> LDS ------------------------------------------------------
        {
            f32x4_t acc[MT1][NT];
#pragma unroll
            for (int i = 0; i < MT1; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[i][nt] = b1r[nt];
#if MIL_BLOCK_LOOKAHEAD > 0
            mil_conv_ring<NT, MT1, KSTEPS, MIL_BLOCK_LOOKAHEAD>(acc, ldsW1, lane,
                [&](int sl, int i) { return ldsX + pixbase1[i] + toff1[sl]; });
#else
#pragma unroll
            for (int sl = 0; sl < KSTEPS; ++sl) {
                Frag8<BF16> wf[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wf[nt] = lds_frag<BF16>(ldsW1 + ((sl * NT + nt) * 64 + lane) * 16);
#pragma unroll
                for (int i = 0; i < MT1; ++i) {
#ifndef MIL_BLOCK_KEEP_DUMMY
                    // row tiles 21..23 lie entirely behind the 324 mid pixels (their results go to the dump slot): skipped
                    // under a scalar (wave-uniform) branch — 3 of 24 row tiles of conv1's MFMA work
                    if (i == MT1 - 1 && (wave + NW * i) * 16 >= 324) continue;
#endif
                    const Frag8<BF16> xf = lds_frag<BF16>(ldsX + pixbase1[i] + toff1[sl]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[i][nt] = mma8(wf[nt], xf, acc[i][nt]);
                }
            }
#endif
            // mid pixels outside the image are conv2's zero padding (only tiles on the image border have any)
            const int my0 = o.oy0 - 1, mx0 = o.ox0 - 1;
            const bool border = my0 < 0 || mx0 < 0 || my0 + 18 > H || mx0 + 18 > W;
#pragma unroll
            for (int i = 0; i < MT1; ++i) {
                bool inside = true;
                if (border) {
                    const int tp = (wave + NW * i) * 16 + r, py = (tp * 3641) >> 16, px = tp - py * 18;      // tp / 18 for tp < 1024
                    inside = (unsigned)(my0 + py) < (unsigned)H && (unsigned)(mx0 + px) < (unsigned)W;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bf16x4_t ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float v = acc[i][nt][e]; ov[e] = (__bf16)fmaxf(v, v * a.slope); }
                    u32x2_t ou = __builtin_bit_cast(u32x2_t, ov);
                    if (border) { ou[0] = inside ? ou[0] : 0u; ou[1] = inside ? ou[1] : 0u; }
                    const int dst = (LAST_PARTIAL && nt == NT - 1 && gq >= 2) ? a.lds_dump_off : sdst1[i] + nt * 32;
                    *reinterpret_cast<u32x2_t*>(smem + dst) = ou;
                }
            }
        }
        __syncthreads();                       // mid tile visible

        // ---- the tile's own 16x16 mid pixels -> o1 tensor (kept for the backward) ------------------------
        const int obase = ((o.img0 * H + o.oy0) * W + o.ox0) * (CP * 2);
        const int ylim = H - o.oy0, xlim = W - o.ox0;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const bool ok = (c_pos[i] >> 10) < ylim && (c_pos[i] & 1023) < xlim;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(ldsO + c_lds[i]);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, ok ? (unsigned)(obase + c_rel[i]) : MIL_OOB, 0, 0);
        }
        // ---- conv2 + residual + LeakyReLU ------------------------------------------------------------------
        f32x4_t acc[MT2][NT];
#pragma unroll
        for (int m = 0; m < MT2; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = b2r[nt];
#if MIL_BLOCK_LOOKAHEAD > 0
        mil_conv_ring<NT, MT2, KSTEPS, MIL_BLOCK_LOOKAHEAD>(acc, ldsW2, lane,
            [&](int sl, int m) { return ldsO + pixbase2[m] + toff2[sl]; });
#else
#pragma unroll
        for (int sl = 0; sl < KSTEPS; ++sl) {
            Frag8<BF16> wf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = lds_frag<BF16>(ldsW2 + ((sl * NT + nt) * 64 + lane) * 16);
#pragma unroll
            for (int m = 0; m < MT2; ++m) {
                const Frag8<BF16> of = lds_frag<BF16>(ldsO + pixbase2[m] + toff2[sl]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mma8(wf[nt], of, acc[m][nt]);
            }
        }
#endif
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
            const bool ok = (o_pos[p] >> 10) < ylim && (o_pos[p] & 1023) < xlim;
            const unsigned ooff = ok ? (unsigned)(obase + o_rel[p]) : MIL_OOB;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float lo = acc[2 * p][nt][i], hi = acc[2 * p + 1][nt][i];
                    if (i == 0) mil_swap16<true>(lo, hi); else mil_swap16<false>(lo, hi);
                    v[i] = lo;
                    v[4 + i] = hi;
                }
                const bool chan_ok = !(LAST_PARTIAL && nt == NT - 1) || last_ok;
                const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(ldsX + (chan_ok ? xres[p] + nt * 32 : 0));
                bf16x8_t ov;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float s = v[i] + (float)t[i]; ov[i] = (__bf16)fmaxf(s, s * a.slope); }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ov), rs_y, chan_ok ? ooff + nt * 32 : MIL_OOB, 0, 0);
            }
        }
    }
}

static int mil_block_waves() {
    // measured on the 64x64x24 maps: 434 us with 4 waves per workgroup, 446 us with 8 (16 resident waves per CU at 127
    // VGPRs) — more resident waves do not help: LDS operand bandwidth (~450 KB per tile) and the MFMA pipe set the pace
    static const int v = [] { const char* e = mil_ab_env("MIL_BLOCK_WAVES"); return (e && atoi(e) == 8) ? 8 : 4; }();
    return v;
}

template <int CP, int NT>
static int launch_block_fwd(BlockFwdArgs a, hipStream_t st) {
    constexpr int PIXB = mil_pix_pitch(CP, 2), CG = CP / 8;
    constexpr int KSTEPS = (9 * CG + 3) / 4;
    ConvGeom& g = a.g;
    g.tw_log2 = 4; g.th_log2 = 4; g.ti_log2 = 0;
    g.tiles_x = (g.W + 15) >> 4; g.tiles_y = (g.H + 15) >> 4; g.n_groups = g.n_img;
    g.hh = 20; g.hw = 20;
    const int x_bytes = 400 * PIXB, o_bytes = (324 * PIXB + 15) & ~15, w_bytes = 2 * KSTEPS * NT * 64 * 16;
    a.lds_o_off = x_bytes; a.lds_w_off = x_bytes + o_bytes; a.lds_dump_off = x_bytes + o_bytes + w_bytes;
    const int lds = a.lds_dump_off + 64;
    if (lds > 160 * 1024) return MIL_ERR_UNSUPPORTED;
#ifdef MIL_AB_SWITCHES
    const int nw = mil_block_waves();
    auto kern = nw == 8 ? conv_block_fwd_kernel<CP, NT, 8> : conv_block_fwd_kernel<CP, NT, 4>;
#else
    constexpr int nw = 4;              // the 8-wave form (A/B builds only) measured no faster and spills at 128 VGPRs
    auto kern = conv_block_fwd_kernel<CP, NT, 4>;
#endif
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return MIL_ERR_LAUNCH;
    static thread_local int occ_lds = -1, occ_n = 1;
    static thread_local const void* occ_k = nullptr;
    if (occ_k != reinterpret_cast<const void*>(kern) || occ_lds != lds) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 64 * nw, (size_t)lds) != hipSuccess || n < 1) n = 1;
        occ_k = reinterpret_cast<const void*>(kern); occ_lds = lds; occ_n = n;
    }
    const int per_cu = occ_n > 4 ? 4 : occ_n;
    const size_t img = (size_t)g.H * g.W * CP * 2;
    const int chunk = mil_imgs_under_2g(img);
    const int n_total = g.n_img;
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        const int n = (n_total - i0 < chunk) ? n_total - i0 : chunk;
        BlockFwdArgs c = a;
        c.g.n_img = n; c.g.n_groups = n;
        c.x = a.x + (size_t)i0 * (img / 2); c.o1 = a.o1 + (size_t)i0 * (img / 2); c.y = a.y + (size_t)i0 * (img / 2);
        const int ntiles = n * g.tiles_y * g.tiles_x;
        int grid = mil_num_cus() * per_cu;
        if (grid > ntiles) grid = ntiles;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, st, c, ntiles, (unsigned)(img * n));
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

// ---- the same block as a ROW WALK on 64-pixel-wide maps (the layer-1 maps of 256 x 256 tiles), bf16, 24 channels -------------
// A workgroup owns a whole image and walks down it four rows (256 pixels) at a time, the tile spanning the full width.  The input
// and the mid activation live in two 8-row LDS rings (row pitch 65 records: column 64 of a row IS column -1 of the next, one
// shared zero record).  A step commits the four NEW input rows 4s .. 4s+3 (12 KB, contiguous in the NHWC tensor), computes mid
// rows 4s-1 .. 4s+2 from input rows 4s-2 .. 4s+3, copies them to o1, and computes output rows 4s-4 .. 4s-1 from mid rows
// 4s-5 .. 4s with the residual read from the input ring.  Against the 16 x 16 tiles: the input is staged once (1.0 x instead of
// 400 / 256 halo pixels per tile), conv1 runs on 16 row tiles per step instead of 21 (of 24) — at one prologue step per image
// (mid rows -1 .. 2) and one epilogue step whose conv1 sees one real row.  Same arithmetic per output element: bit-identical.
// Ring rows wrap: the row part of a fragment offset is ((base + m + ky) & 7) * ROW, rebuilt per step (28 mads).
// 78.8 KB of LDS: two 4-wave workgroups per CU.  Whole images are the unit of work: launch_block_fwd takes this form when the
// images fill the resident workgroups evenly enough (mil_block_strip_wanted; MIL_BLOCK_STRIP = 0 / 1 is a TEST knob).
//
// SW x R = 64 x 4 (the 64 x 64 maps of 256 x 256 tiles) or 128 x 2 (the 128 x 128 maps of 512 x 512 tiles): 256 pixels per step
// either way; rings of 2R rows.  Row tile m of a wave = (row m % R2, column block wave + 4 * (m / R2)) with R2 = rows per
// column block = R (64 wide) or 2 (128 wide: two column blocks per wave), so that row tiles 2p, 2p+1 are always two rows of
// the same columns (the epilogue's permlane pairs).
#ifndef MIL_STRIP_LA
#define MIL_STRIP_LA 2                // pixel fragments read this many (k-step, row tile) steps ahead of their MFMAs
#endif
template <int SW, int R> struct StripCfg {
    static constexpr int RP = SW + 1, ROW = RP * 48, NR = 2 * R, PLANE = (NR * RP + 1) * 48;
    static constexpr int W_BYTES = 7 * 2 * 64 * 16;
    static constexpr int LDS = 2 * PLANE + 2 * W_BYTES + 64;
};

template <int SW, int R>
__global__ __launch_bounds__(256, 2) void conv_block_strip_kernel(BlockFwdArgs a, int n_img, unsigned bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    using Cfg = StripCfg<SW, R>;
    constexpr int NT = 2, NTHR = 256, KSTEPS = 7, CG = 3, MT = 4;
    static_assert(SW * R == 256 && (SW == 64 || SW == 128), "256 pixels per step");
    constexpr int PIXB = 48, RP = Cfg::RP, ROW = Cfg::ROW, NR = Cfg::NR, RMASK = NR - 1;
    constexpr int PLANE = Cfg::PLANE, W_BYTES = Cfg::W_BYTES;
    constexpr int OFF_X = 0, OFF_O = PLANE, OFF_W1 = 2 * PLANE, OFF_W2 = OFF_W1 + W_BYTES, OFF_DUMP = OFF_W2 + W_BYTES;
    constexpr int NCB = SW / 64;                                     // column blocks of 16 pixels per wave
    auto m_row = [](int m) { return NCB == 1 ? m : (m & 1); };
    auto m_cb = [](int m) { return NCB == 1 ? 0 : (m >> 1); };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    char* ldsX = smem + OFF_X;
    char* ldsO = smem + OFF_O;
    const int H = a.g.H;
    const int S = (H + R - 1) / R + 1;                               // steps per image: one per R rows + the last output rows
    mil_stage_filter(smem + OFF_W1, a.w1, W_BYTES, tid, NTHR);
    mil_stage_filter(smem + OFF_W2, a.w2, W_BYTES, tid, NTHR);
    // the zero columns of both rings: records 0, RP, .., NR * RP (nothing ever writes them again)
    if (tid < 2 * (NR + 1) * 3) {
        const int pl = tid / ((NR + 1) * 3), rem = tid - pl * ((NR + 1) * 3), rec = rem / 3, pc = rem - rec * 3;
        *reinterpret_cast<u32x4_t*>(smem + pl * PLANE + rec * ROW + pc * 16) = u32x4_t{0u, 0u, 0u, 0u};
    }
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, bytes);
    const __amdgpu_buffer_rsrc_t rs_o = mil_rsrc(a.o1, bytes);
    const __amdgpu_buffer_rsrc_t rs_y = mil_rsrc(a.y, bytes);

    // R rows = 256 pixels x 3 pieces (16 B = eight channels): flat id = tid + 256*i = byte offset / 16 inside the row group
    constexpr int NPX = 3;
    int p_lds[NPX], p_row[NPX];
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int idx = tid + NTHR * i, px = idx / CG, j = idx - px * CG;
        p_row[i] = px / SW;
        p_lds[i] = ((px % SW) + 1) * PIXB + j * 16;                  // inside its ring row
    }
    // k-group q = 4*sl + gq = (tap, 8-channel group): column / channel part of the fragment offset, filter row in the low bits
    int kq[KSTEPS];
#pragma unroll
    for (int sl = 0; sl < KSTEPS; ++sl) {
        const int q = 4 * sl + gq;
        int tap = q / CG, cg = q - tap * CG;
        if (tap >= 9) { tap = 0; cg = 0; }                           // zero weights: any finite operand
        kq[sl] = ((tap % 3) * PIXB + cg * 16) | (tap / 3);
    }
    const int col0 = wave * 16 + r;                                  // this lane's pixel column in column block 0 (+ 64 in block 1)
    const int pb = col0 * PIXB;                                      // record under the top-left tap: column col - 1 = record col
    const bool last_ok = (gq >> 1) == 0;                             // column tile 1 holds channels 16-23 only
    f32x4_t b1r[NT], b2r[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b1r[nt][i] = a.b1 ? a.b1[nt * 16 + gq * 4 + i] : 0.f;
            b2r[nt][i] = a.b2 ? a.b2[nt * 16 + gq * 4 + i] : 0.f;
        }
    int koff[MT][KSTEPS];
    u32x4_t rx[NPX];
    // input rows R*s .. R*s+R-1 of image img (rows beyond the image: zeros = the bottom padding); no branch around the loads
    auto fetch = [&](int img, int s) {
        const int y0 = R * s;
        const int base = (img * H + y0) * (SW * PIXB);
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const bool ok = y0 + p_row[i] < H;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(base + (tid + NTHR * i) * 16) : MIL_OOB, 0, 0);
        }
    };
    const int G = gridDim.x;
    int img = blockIdx.x, s = 0;
    if (img < n_img) fetch(img, 0);
    while (img < n_img) {
        const int nb = (s & 1) * R;                                  // ring rows of the new input rows
        __syncthreads();                       // previous step: every read of both rings is done
#pragma unroll
        for (int i = 0; i < NPX; ++i) *reinterpret_cast<u32x4_t*>(ldsX + (nb + p_row[i]) * ROW + p_lds[i]) = rx[i];
        if (s == 0) {
            // input rows -2, -1 (ring rows NR-2, NR-1) are the top padding: 2 * RP + 1 records
            for (int id = tid; id < (2 * RP + 1) * 3; id += NTHR) *reinterpret_cast<u32x4_t*>(ldsX + (NR - 2) * ROW + id * 16) = u32x4_t{0u, 0u, 0u, 0u};
        }
        __syncthreads();                       // new input rows visible
        int ns = s + 1, nimg = img;
        if (ns == S) { ns = 0; nimg += G; }
        if (nimg < n_img) fetch(nimg, ns);
        // conv row m_row, filter row ky: input rows R*s-2+m_row+ky (ring row = row & RMASK) for conv1, mid rows R*s-R-1+m_row+ky
        // (ring row = (row + R - 1) & RMASK) for conv2 — the same ring rows
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int sl = 0; sl < KSTEPS; ++sl)
                koff[m][sl] = ((nb + NR - 2 + m_row(m) + (kq[sl] & 3)) & RMASK) * ROW + (kq[sl] & ~3) + m_cb(m) * (64 * PIXB);
        const int ibase = img * H * (SW * PIXB);

        // ---- conv1: mid rows R*s-1 .. R*s+R-2 -> the mid ring's rows (nb + R - 2 + m_row) & RMASK -------------------------------
        {
            f32x4_t acc[MT][NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = b1r[nt];
            mil_conv_ring<NT, MT, KSTEPS, MIL_STRIP_LA>(acc, smem + OFF_W1, lane, [&](int sl, int m) { return ldsX + pb + koff[m][sl]; });
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool inside = (unsigned)(R * s - 1 + m_row(m)) < (unsigned)H;     // wave-uniform: a mid row outside the image is conv2's zero padding
                const int sdst = ((nb + R - 2 + m_row(m)) & RMASK) * ROW + (col0 + m_cb(m) * 64 + 1) * PIXB + gq * 8;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bf16x4_t ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float v = acc[m][nt][e]; ov[e] = (__bf16)fmaxf(v, v * a.slope); }
                    u32x2_t ou = __builtin_bit_cast(u32x2_t, ov);
                    ou[0] = inside ? ou[0] : 0u; ou[1] = inside ? ou[1] : 0u;
                    const int dst = (nt == NT - 1 && gq >= 2) ? OFF_DUMP - OFF_O : sdst + nt * 32;
                    *reinterpret_cast<u32x2_t*>(ldsO + dst) = ou;
                }
            }
        }
        __syncthreads();                       // new mid rows visible

        // ---- the new mid rows -> o1 tensor (kept for the backward): 12 KB, contiguous -----------------------------------------
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int jy = R * s - 1 + p_row[i];
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(ldsO + ((nb + R - 2 + p_row[i]) & RMASK) * ROW + p_lds[i]);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, (unsigned)jy < (unsigned)H ? (unsigned)(ibase + (R * s - 1) * (SW * PIXB) + (tid + NTHR * i) * 16) : MIL_OOB, 0, 0);
        }
        if (s > 0) {
            // ---- conv2 + residual (input ring) + LeakyReLU -> y rows R*s-R .. R*s-1 ---------------------------------------------
            f32x4_t acc[MT][NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = b2r[nt];
            mil_conv_ring<NT, MT, KSTEPS, MIL_STRIP_LA>(acc, smem + OFF_W2, lane, [&](int sl, int m) { return ldsO + pb + koff[m][sl]; });
#pragma unroll
            for (int p = 0; p < MT / 2; ++p) {
                // after the swap a lane holds 8 channels of pixel (row rr, column col) of row tiles 2p, 2p+1
                const int rr = m_row(2 * p) + (gq & 1), col = col0 + m_cb(2 * p) * 64, ey = R * s - R + rr;
                const unsigned ooff = ey < H ? (unsigned)(ibase + (ey * SW + col) * PIXB + (gq >> 1) * 16) : MIL_OOB;
                const int xres = ((nb + R + rr) & RMASK) * ROW + (col + 1) * PIXB + (gq >> 1) * 16;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float lo = acc[2 * p][nt][i], hi = acc[2 * p + 1][nt][i];
                        if (i == 0) mil_swap16<true>(lo, hi); else mil_swap16<false>(lo, hi);
                        v[i] = lo;
                        v[4 + i] = hi;
                    }
                    const bool chan_ok = nt < NT - 1 || last_ok;
                    const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(ldsX + (chan_ok ? xres + nt * 32 : 0));
                    bf16x8_t ov;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const float q = v[i] + (float)t[i]; ov[i] = (__bf16)fmaxf(q, q * a.slope); }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ov), rs_y, (chan_ok && ooff != MIL_OOB) ? ooff + nt * 32 : MIL_OOB, 0, 0);
                }
            }
        }
        img = nimg; s = ns;
    }
}

// The row-walk forms take 64-pixel-wide (bf16: also 128-pixel-wide) maps when whole images fill the resident workgroups well: their unit of work is an image
// (`steps` steps of `strip_cost` each), the tiled forms' a tile (`tile_cost` each; costs in k cycles, measured).
// MIL_BLOCK_STRIP (a TEST knob, read per call: "0" never, "1" whenever the map is 64 wide) lets the tests compare the two forms
// bit for bit on small inputs.
static bool mil_block_strip_wanted(bool width_ok, int n_img, long tiles_per_img, int steps, int tile_cost, int strip_cost, int grid_cap) {
    if (!width_ok) return false;
    const char* e = getenv("MIL_BLOCK_STRIP");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    const long cost_tile = (n_img * tiles_per_img + grid_cap - 1) / grid_cap * tile_cost;
    const long cost_strip = (long)((n_img + grid_cap - 1) / grid_cap) * steps * strip_cost;
    return cost_strip < cost_tile;
}

template <int SW, int R>
static int launch_block_strip(BlockFwdArgs a, hipStream_t st) {
    const ConvGeom& g = a.g;
    constexpr int lds = StripCfg<SW, R>::LDS;
    auto kern = conv_block_strip_kernel<SW, R>;
    static std::atomic<unsigned long long> attr_set{0};
    if (mil_device_needs(attr_set)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MIL_ERR_LAUNCH;
        mil_device_done(attr_set);
    }
    const int per_cu = mil_resident_per_cu(kern, lds, 2, 256);
    const size_t img = (size_t)g.H * g.W * 48;
    const int chunk = mil_imgs_under_2g(img);
    for (int i0 = 0; i0 < g.n_img; i0 += chunk) {
        const int n = (g.n_img - i0 < chunk) ? g.n_img - i0 : chunk;
        BlockFwdArgs c = a;
        c.x = a.x + (size_t)i0 * (img / 2); c.o1 = a.o1 + (size_t)i0 * (img / 2); c.y = a.y + (size_t)i0 * (img / 2);
        int grid = mil_num_cus() * per_cu;
        if (grid > n) grid = n;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, c, n, (unsigned)(img * n));
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

#include <type_traits>
#include "conv_block_fwd_x3.cuh"

// o1 = lrelu(conv3x3(x)+b1), y = lrelu(conv3x3(o1)+b2+x) for an identity-shortcut block; x/o1/y [n,H,W,cp].
// bf16, cp in {24, 40}, H and W >= 16; MIL_DT_F32S (fp32 tensors, split products), cp = 24 (20 real channels), H >= 8 and
// W >= 16; otherwise MIL_ERR_UNSUPPORTED (caller: two mil_conv_igemm calls).
extern "C" int mil_conv_block_fwd(const void* x, const void* wpack1, const float* bias1, const void* wpack2, const float* bias2,
                                  void* o1, void* y, int n_img, int H, int W, int cp, float slope, int dtype, void* stream) {
    if (!x || !wpack1 || !wpack2 || !o1 || !y || n_img < 0 || H <= 0 || W <= 0) return MIL_ERR_ARG;
    if (dtype == MIL_DT_F32S) {
        if (cp != 24 || H < 8 || W < 16 || H >= 1024 || W >= 1024 || slope < 0.f || slope >= 1.f) return MIL_ERR_UNSUPPORTED;
        if (n_img == 0) return MIL_OK;
        BlockFwdX3Args b{};
        b.x = (const float*)x; b.w1 = (const char*)wpack1; b.w2 = (const char*)wpack2; b.b1 = bias1; b.b2 = bias2;
        b.o1 = (float*)o1; b.y = (float*)y; b.slope = slope; b.apx = cp * 4;
        b.g.n_img = n_img; b.g.H = H; b.g.W = W; b.g.Ho = H; b.g.Wo = W; b.g.ks = 5; b.g.stride = 1; b.g.pad = 2; b.g.zins = 0;
        return launch_block_fwd_x3(b, reinterpret_cast<hipStream_t>(stream));
    }
    if (dtype != MIL_DT_BF16 || H < 16 || W < 16 || H >= 1024 || W >= 1024 || slope < 0.f || slope >= 1.f) return MIL_ERR_UNSUPPORTED;
    if (n_img == 0) return MIL_OK;
    BlockFwdArgs a{};
    a.x = (const __bf16*)x; a.w1 = (const __bf16*)wpack1; a.w2 = (const __bf16*)wpack2; a.b1 = bias1; a.b2 = bias2;
    a.o1 = (__bf16*)o1; a.y = (__bf16*)y; a.slope = slope;
    a.g.n_img = n_img; a.g.H = H; a.g.W = W; a.g.Ho = H; a.g.Wo = W; a.g.ks = 5; a.g.stride = 1; a.g.pad = 2; a.g.zins = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (cp == 24) {
        const long tpi = (long)((H + 15) >> 4) * ((W + 15) >> 4);
        if (W == 64 && mil_block_strip_wanted(true, n_img, tpi, ((H + 3) >> 2) + 1, 11, 9, mil_num_cus() * 2)) return launch_block_strip<64, 4>(a, st);
        if (W == 128 && mil_block_strip_wanted(true, n_img, tpi, ((H + 1) >> 1) + 1, 11, 9, mil_num_cus() * 2)) return launch_block_strip<128, 2>(a, st);
        return launch_block_fwd<24, 2>(a, st);
    }
    if (cp == 40) return launch_block_fwd<40, 3>(a, st);
    return MIL_ERR_UNSUPPORTED;
}
