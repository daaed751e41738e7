// The fused stem backward (stem_bwd_fused_kernel<FROM_X>, above: max-pool backward + LeakyReLU backward + the 7x7 weight / bias
// gradient in one pass — autograd of gbm/model.py:24-26,51-53) as a ROW WALK on 256-pixel-wide tiles, bf16.  Included by
// conv_wgrad.hip; same gather, same GEMM rows and k order per pixel.  (The partial sums of a workgroup cover other pixels than a
// tile walk's, so dW / db agree with the tiled form to fp32 summation order, not bit for bit.)
//
// The tiled kernel stages a 19 x 19 s2d halo per 16 x 16 stem pixels (1.41 x) out of 152-byte row segments of three colour
// planes: a wave's load instruction touches six or seven such segments, and ISSUING the nine loads per thread is 19 % of its
// tile time (stamps: 147 cycles per load instruction); the 81 pooling windows of a tile are 1.27 x its own 64.  (With the
// windows decoded once at the commit in both forms the row walk is 2-3 % ahead: 769 against 788 us per launch.)  Here a workgroup
// owns a whole IMAGE and walks down two stem rows (256 pixels, the GEMM's K per step as before) at a time:
//   * s2d rows in a 6-row LDS ring: a step converts the two NEW rows (four image rows x three colours, each ONE contiguous 1 KB
//     load instruction per wave) and re-uses three; every input byte fetched once;
//   * pooling windows in a 3-row ring: one new pooled row (64 windows x 3 pieces: 3 KB contiguous) per step, DECODED as it is
//     committed (masked gradient as fp32): the gather's four blocks around a window read the decoded values;
//   * columns -4 .. -1 / 128 .. 129 of an s2d row and window column 64 are padding: zeroed once, never written;
//   * the row part of an A-operand address is ((2s + dy + ty) mod 6) * ROWB, rebuilt per step (6 values per lane).
// Steps per image: H2/2 + 1 (step 0 only loads).  Whole images are the unit of work (mil_stem_walk_wanted_bwd); MIL_STEM_WALK
// = 0 / 1 is the TEST knob it shares with the forward.  Split precision keeps the tiled kernel: five 134-pixel rows of [hi | lo]
// records (54 KB) and its 29 KB gradient tile do not fit twice per CU.
#pragma once

constexpr int SBW_XW = 134, SBW_NRING = 6, SBW_WW = 65;
constexpr int SBW_ROWB = SBW_XW * 48;
constexpr int SBW_XBYTES = SBW_NRING * SBW_ROWB;                       // 38592
constexpr int SBW_ZBYTES = 256 * 48;                                    // 12288
constexpr int SBW_GBYTES = 3 * SBW_WW * 96, SBW_IBYTES = 3 * SBW_WW * 24;      // 18720 (decoded fp32 gradients), 4680
constexpr int SBW_LDS = SBW_XBYTES + SBW_ZBYTES + SBW_GBYTES + SBW_IBYTES + 64;

__global__ __launch_bounds__(256, 2) void stem_bwd_walk_kernel(StemBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int NT = 2, KS = 4, COUTP = 24;
    constexpr int PIXB = 48, PIXZ = 48, PIXG = 96, ROWB = SBW_ROWB, NRING = SBW_NRING, WW = SBW_WW;
    constexpr int NPC = 3, MT = KS * KS * NPC / 4, MW = (MT + 3) / 4;      // 12 row tiles of (tap, four s2d channels), three per wave
    static_assert(MT == 4 * MW, "every wave owns MW full row tiles");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ldsX = smem;
    char* ldsZ = smem + SBW_XBYTES;
    char* ldsG = ldsZ + SBW_ZBYTES;
    char* ldsI = ldsG + SBW_GBYTES;
    constexpr int dump = SBW_XBYTES + SBW_ZBYTES + SBW_GBYTES + SBW_IBYTES;
    for (int i = tid * 16; i < SBW_LDS; i += 256 * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);      // padding columns / windows: zero once
    const int H = a.H, W = a.W, H2 = a.g.H, Hp = a.Hp, Wp = a.Wp;
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rs_g = mil_rsrc(a.gp, a.gp_bytes);
    const __amdgpu_buffer_rsrc_t rs_i = mil_rsrc(a.widx, a.wi_bytes);

    // ---- load items: (s2d row 0..1 of the step, pair of s2d pixels, colour): 384 = one and a half per thread -----------------
    constexpr int NL = 2;
    int l_col[NL], l_rel[NL], l_row[NL];
    bool l_used[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = tid + 256 * i;
        const int pair = idx & 63, t = idx >> 6, c = t % 3, row = t / 3;
        l_used[i] = idx < 2 * 64 * 3;
        l_col[i] = (4 + 2 * pair) * PIXB + c * 8;
        l_rel[i] = ((c * H + 2 * row) * W + 4 * pair) * 4;
        l_row[i] = row;
    }
    u32x4_t lr0[NL], lr1[NL];
    // s2d rows 2s-1, 2s of image img (rows outside the image: zeros)
    auto fetch_x = [&](int img, int s) {
        const int y0 = 2 * s - 1;
        const int base = ((img * 3) * H + 2 * y0) * W * 4;       // negative for s = 0: its valid row's sums are not
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const bool ok = l_used[i] && (unsigned)(y0 + l_row[i]) < (unsigned)H2;
            const unsigned off = ok ? (unsigned)(base + l_rel[i]) : MIL_OOB;
            lr0[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
            lr1[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? off + (unsigned)(W * 4) : MIL_OOB, 0, 0);
        }
    };
    // pooled row s of the image: 64 windows x 3 pieces (gradient piece = 8 channels, winner piece = 8 bytes)
    const int w_win = tid / 3, w_j = tid - w_win * 3;
    const bool w_used = tid < 64 * 3;
    u32x4_t rgp;
    u32x2_t rwi;
    auto fetch_win = [&](int img, int s) {
        const bool ok = w_used && s < Hp && w_win < Wp;
        const int pix = (img * Hp + s) * Wp + w_win;
        const unsigned goff = ok ? (unsigned)(pix * a.gpx + w_j * 16) : MIL_OOB;
        const u32x2_t lo = __builtin_amdgcn_raw_buffer_load_b64(rs_g, goff, 0, 0);
        const u32x2_t hi = __builtin_amdgcn_raw_buffer_load_b64(rs_g, (!ok || (a.gpx != 48 && w_j == 2)) ? MIL_OOB : goff + 8, 0, 0);      // dense: channels 20-23 do not exist
        rgp = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
        rwi = __builtin_amdgcn_raw_buffer_load_b64(rs_i, ok ? (unsigned)(pix * COUTP + w_j * 8) : MIL_OOB, 0, 0);
    };

    // per-lane tr-read offsets of this wave's row tiles: row piece P = 4*mt + (lane&3) = (tap, four s2d channels)
    int tcol[MW], tty[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int P = 4 * (wave + 4 * i) + (lane & 3);
        const int tap = P / NPC, c3 = P - tap * NPC;
        tcol[i] = (tap % KS) * PIXB + c3 * 8;
        tty[i] = tap / KS;
    }
    float bsum[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x4_t acc[MW][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < MW; ++i) acc[i][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, gq = lane >> 4;
    // the lane's two pixels of a 32-pixel k-step: columns c0 + 8*gq + q4 (+ 4); s2d column of stem column c under tap tx = c + 2 + tx
    const int wcol0 = (8 * gq + q4 + 2) * PIXB, wcol1 = wcol0 + 4 * PIXB;
    // dz builder: thread -> (2x2 pixel block b_x of the step's two rows, 6-channel group)
    const int b_x = tid >> 2, bc6 = tid & 3;

    const int S = H2 / 2 + 1;
    const int G = gridDim.x;
    int img = blockIdx.x, s = 0;
    if (img < a.g.n_img) { fetch_x(img, 0); fetch_win(img, 0); }
    MIL_STAMP_DECL(7)
    while (img < a.g.n_img) {
        MIL_STAMP_BEGIN()
        __syncthreads();                         // previous step's MFMA loop and gather are done with the rings and the dz tile
        MIL_STAMP_MARK(0)
        // ---- commit: the two new s2d rows (ring rows (2s+3) % 6, (2s+4) % 6) and pooled row s (ring row s % 3) -----------------
        {
            const int rb = (2 * s + 3) % NRING;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const f32x4_t v0 = __builtin_bit_cast(f32x4_t, lr0[i]), v1 = __builtin_bit_cast(f32x4_t, lr1[i]);
                bf16x4_t pa, pb;
                pa[0] = (__bf16)v0[0]; pa[1] = (__bf16)v0[1]; pa[2] = (__bf16)v1[0]; pa[3] = (__bf16)v1[1];
                pb[0] = (__bf16)v0[2]; pb[1] = (__bf16)v0[3]; pb[2] = (__bf16)v1[2]; pb[3] = (__bf16)v1[3];
                int rr = rb + l_row[i];
                rr = rr >= NRING ? rr - NRING : rr;
                const int d0 = l_used[i] ? rr * ROWB + l_col[i] : dump;
                *reinterpret_cast<bf16x4_t*>(ldsX + d0) = pa;
                *reinterpret_cast<bf16x4_t*>(ldsX + (l_used[i] ? d0 + PIXB : dump + 8)) = pb;
            }
            if (s == 0) {
                // s2d row -2 (ring row 2) is the top padding of the first stem row: zeros
                for (int id = tid; id < ROWB / 16; id += 256) *reinterpret_cast<uint4*>(ldsX + 2 * ROWB + id * 16) = make_uint4(0, 0, 0, 0);
            }
            if (w_used) {
                // a window's record is decoded ONCE here — the masked gradient g * lrelu'(winner) as fp32 — instead of by each of the
                // four 2x2 blocks around it in the gather (24 of a gather thread's decodes per step: a third of its instructions)
                const int wr = s % 3;
                f32x4_t g0, g1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned wb = (rwi[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    const unsigned gw = rgp[j >> 1];
                    const float gj = __uint_as_float((j & 1) ? (gw & 0xffff0000u) : (gw << 16));
                    const float gm = (wb & 16u) ? gj * a.slope : gj;
                    if (j < 4) g0[j] = gm; else g1[j - 4] = gm;
                }
                char* gd = ldsG + (wr * WW + w_win) * PIXG + w_j * 32;
                *reinterpret_cast<f32x4_t*>(gd) = g0;
                *reinterpret_cast<f32x4_t*>(gd + 16) = g1;
                *reinterpret_cast<u32x2_t*>(ldsI + (wr * WW + w_win) * COUTP + w_j * 8) = rwi;
            }
        }
        MIL_STAMP_MARK(1)
        __syncthreads();
        MIL_STAMP_MARK(2)
        int ns = s + 1, nimg = img;
        if (ns == S) { ns = 0; nimg += G; }
        if (nimg < a.g.n_img) { fetch_x(nimg, ns); fetch_win(nimg, ns); }
        MIL_STAMP_MARK(3)
        if (s > 0) {
            const int k = s - 1;                 // stem rows 2k, 2k+1; 2x2 blocks (k, b_x); windows rows k, k+1
            // ---- dz tile = lrelu'(stem) * maxpool^T(g): gather over the 4 windows that cover a 2x2 block (as the tiled kernel) ----
            {
                float gsum[2][2][6];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                        for (int j = 0; j < 6; ++j) gsum[dy][dx][j] = 0.f;
#pragma unroll
                for (int wy = 0; wy < 2; ++wy) {
                    const int wr = (k + wy) % 3;
#pragma unroll
                    for (int wx = 0; wx < 2; ++wx) {
                        const int win = wr * WW + b_x + wx;
                        const unsigned short* wi = reinterpret_cast<const unsigned short*>(ldsI + win * COUTP + bc6 * 6);
                        const f32x2_t* gp = reinterpret_cast<const f32x2_t*>(ldsG + win * PIXG + bc6 * 24);
                        const unsigned wpk[3] = {wi[0], wi[1], wi[2]};
                        const f32x2_t gpk[3] = {gp[0], gp[1], gp[2]};
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            const float gm = gpk[j >> 1][j & 1];
                            const unsigned t = (wpk[j >> 1] >> (8 * (j & 1))) & 15u;
#pragma unroll
                            for (int dy = 0; dy < 2; ++dy) {
                                const int ky = dy + 1 - 2 * wy;          // tap row of pixel 2k+dy inside window k+wy
                                if (ky < 0 || ky > 2) continue;
#pragma unroll
                                for (int dx = 0; dx < 2; ++dx) {
                                    const int kx = dx + 1 - 2 * wx;
                                    if (kx < 0 || kx > 2) continue;
                                    gsum[dy][dx][j] += (t == (unsigned)(ky * 3 + kx)) ? gm : 0.f;
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int tp = dy * 128 + 2 * b_x + dx;
                        unsigned* dst = reinterpret_cast<unsigned*>(ldsZ + tp * PIXZ + bc6 * 12);
#pragma unroll
                        for (int kk = 0; kk < 3; ++kk) {
                            bf16x2_t pr;
                            pr[0] = (__bf16)gsum[dy][dx][2 * kk]; pr[1] = (__bf16)gsum[dy][dx][2 * kk + 1];
                            dst[kk] = __builtin_bit_cast(unsigned, pr);
                            bsum[2 * kk] += (float)pr[0]; bsum[2 * kk + 1] += (float)pr[1];      // the rounded values, as the MFMA loop sees them
                        }
                    }
            }
            MIL_STAMP_MARK(4)
            __syncthreads();
            MIL_STAMP_MARK(5)
            // ---- weight gradient: rows (tap, s2d channel), cols stem channel, K = the step's 256 pixels, one k-step ahead ------------
            {
                // A-operand row offsets: stem row 2k+dy reads s2d row 2s-4+dy+ty = ring row (2s + dy + ty) % 6
                int roff[2][MW];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int i = 0; i < MW; ++i) {
                        int rr = (2 * s) % NRING + dy + tty[i];
                        rr = rr >= NRING ? rr - NRING : rr;
                        roff[dy][i] = rr * ROWB + tcol[i];
                    }
                bf16x8_t bc[NT], ac[MW], bn[NT], an[MW];
                auto load = [&](int k32, bf16x8_t (&bf)[NT], bf16x8_t (&af)[MW]) {
                    const int dy = k32 >> 7, c0 = (k32 & 127) * PIXB;
                    const char* z0 = ldsZ + (k32 + 8 * gq + q4) * PIXZ + p4 * 8;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bf[nt] = mil_tr_pair(z0 + nt * 32, z0 + nt * 32 + 4 * PIXZ);
#pragma unroll
                    for (int i = 0; i < MW; ++i) af[i] = mil_tr_pair(ldsX + roff[dy][i] + c0 + wcol0, ldsX + roff[dy][i] + c0 + wcol1);
                };
                load(0, bc, ac);
#pragma unroll
                for (int k32 = 0; k32 < 256; k32 += 32) {
                    if (k32 + 32 < 256) load(k32 + 32, bn, an);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MW; ++i)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[i], bc[nt], acc[i][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bc[nt] = bn[nt];
#pragma unroll
                    for (int i = 0; i < MW; ++i) ac[i] = an[i];
                }
            }
            MIL_STAMP_MARK(6)
        }
        img = nimg; s = ns;
    }
    MIL_STAMP_STORE(a.stamp, 4)

    constexpr int SLAB_COLS = NT * 16;
    constexpr size_t SLAB_ELEMS = (size_t)(MT + 1) * 16 * SLAB_COLS;
    float* slab = a.slab + (size_t)blockIdx.x * SLAB_ELEMS;
    const int col = lane & 15;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int mt = wave + 4 * i;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                slab[(size_t)(mt * 16 + gq * 4 + e) * SLAB_COLS + nt * 16 + col] = acc[i][nt][e];
    }
    // bias sums -> slab row MT*16: 64 pixel-block threads per channel, added in thread order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 6; ++j) red[(tid >> 2) * 24 + bc6 * 6 + j] = bsum[j];
    __syncthreads();
    if (tid < 24) {
        float v = 0.f;
        for (int b = 0; b < 64; ++b) v += red[b * 24 + tid];
        slab[(size_t)MT * 16 * SLAB_COLS + tid] = v;
    }
}

// bf16, from the fp32 tiles, 256-pixel-wide tiles, whole images filling the resident workgroups evenly (cost: rounds x steps x
// time per step against rounds x tiles x time per tile); MIL_STEM_WALK = 0 / 1 forces either form (TEST knob, read per call).
static bool mil_stem_walk_wanted_bwd(int n_img, int H2, int W2, bool from_x, bool bf16, int tiles_per_img, int grid_cap) {
    if (!from_x || !bf16 || W2 != 128 || (H2 & 1)) return false;
    const char* e = getenv("MIL_STEM_WALK");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    // measured (windows decoded once in both forms): 6.3 k cycles per 16 x 16 tile, 6.0 k per two-row step
    const long cost_tile = ((long)n_img * tiles_per_img + 2 * grid_cap - 1) / (2 * grid_cap) * 2 * 63;     // the tiled form runs two rounds of the resident set
    const long cost_walk = (long)((n_img + grid_cap - 1) / grid_cap) * (H2 / 2 + 1) * 60;
    return cost_walk < cost_tile;
}
