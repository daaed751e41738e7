// 3x3 stride-1 convolution (forward or data gradient: the packed filter decides) for the 40-channel stage in SPLIT PRECISION
// (MIL_DT_F32S: fp32 tensors, bf16x3 products) with the FILTER STREAMED, not staged.  Included by conv_igemm.hip.
//
// The filter-resident persistent kernel (conv_igemm_pf_kernel<F32S,40,..>) keeps 74 KB of [hi | lo] fragments in LDS next to a
// 57 KB halo tile in both planes: one 8-wave workgroup per CU, every wave marching through commit / barrier / MFMA phases with
// the others (0.28 ms per conv for 0.11 ms of matrix work).  Here LDS holds only the halo planes of a 16x16-pixel tile (52 KB)
// and every wave reads the packed fragments of a k-step from L1/L2 into registers one k-step ahead (2 KB per wave-load and
// column tile; four row tiles per wave re-use each fragment for twelve MFMAs): TWO to THREE independent 4-wave workgroups per
// CU, each SIMD alternating between their waves.  Epilogue per (row tile, column tile) and lane: four consecutive channels of a
// pixel = one 16-byte fp32 piece: out = mask( lrelu?( acc + bias? + res? ) ), the contract of mil_conv_igemm.
#pragma once
#include "stamp.cuh"
#ifndef MIL_STREAM_FULL_EPI
#define MIL_STREAM_FULL_EPI 0
#endif

struct StreamX3Args {
    const float* x;         // [n,H,W,C]
    const char* w;          // packed MIL_DT_F32S fragments [KSTEPS][NT][64][32 B] (MIL_PACK_FWD or MIL_PACK_DGRAD)
    const float* bias;      // [NT*16] or null
    const float* res;       // [n,H,W,C] or null
    const float* act;       // [n,H,W,C] or null
    float* y;               // [n,H,W,C]
    ConvGeom g;             // 16x16 tiles of one image
    int lrelu;
    float slope;
    unsigned long long* stamp;      // MIL_STAMP diagnostic build only
};

template <int C, int NT, bool RES, bool ACT>
__global__ __launch_bounds__(256, 2) void conv_stream_x3_kernel(StreamX3Args a, int ntiles, unsigned bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int CG = C / 8, NTHR = 256, PPP = C / 4;               // 16-byte pieces (four fp32 channels) per pixel
    constexpr int PIXB = mil_pix_pitch(C, 2);
    constexpr int HW = 18, PLANE = HW * HW * PIXB;                   // hi plane, then lo plane
    constexpr int KSTEPS = (9 * CG + 3) / 4;
    constexpr int MTW = 4;                                           // row tiles per wave: 16 rows of 16 pixels / 4 waves
    constexpr int NPX = (HW * HW * PPP + NTHR - 1) / NTHR;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    const int dumpo = 2 * PLANE;                                     // 64-byte dump slot behind the planes
    const int H = g.H, W = g.W;
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, bytes);
    const __amdgpu_buffer_rsrc_t rs_res = mil_rsrc(a.res, RES ? bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_act = mil_rsrc(a.act, ACT ? bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_y = mil_rsrc(a.y, bytes);
    const __amdgpu_buffer_rsrc_t rs_w = mil_rsrc(a.w, KSTEPS * NT * 2048);

    // halo pieces: flat id = tid + 256*i -> (pixel id/PPP, piece id%PPP); h_pk = LDS offset (16 bits) | hx << 16 | hy << 21, < 0 unused
    // (+ piece << 26: the global offset is recomputed per tile — five vector instructions per piece against 36 MFMAs — to keep
    // the register budget of the loop below)
    int h_pk[NPX];
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int idx = tid + NTHR * i;
        const int px = idx / PPP, j = idx - px * PPP;
        const int hy = px / HW, hx = px - hy * HW;
        const bool used = px < HW * HW;
        h_pk[i] = used ? (px * PIXB + j * 8) | (hx << 16) | (hy << 21) | (j << 26) : (int)0x80000000u;
    }
    int koff[KSTEPS];                                                // per-lane fragment offset from the top-left tap's record
#pragma unroll
    for (int sl = 0; sl < KSTEPS; ++sl) {
        const int q = 4 * sl + gq;
        int tap = q / CG, cg = q - tap * CG;
        if (tap >= 9) { tap = 0; cg = 0; }                           // zero weights: any finite record
        koff[sl] = ((tap / 3) * HW + (tap % 3)) * PIXB + cg * 16;
    }
    const int pixbase = (wave * MTW * HW + r) * PIXB;                // tile row 4*wave [+m], column r: + m * HW * PIXB
    const int o_rel = (wave * MTW * W + r) * (C * 4) + gq * 16;      // + m * W * C*4 + nt * 64
    const __amdgpu_buffer_rsrc_t rs_b = mil_rsrc(a.bias, a.bias ? NT * 64 : 0);        // read per tile in the epilogue (12 registers less in the loop)

    TileWalker cur, nxt;
    const int bid = mil_xcd_block_id();
    cur.init(g, bid, gridDim.x);
    nxt = cur; nxt.advance();
    u32x4_t rx[NPX];
    auto fetch = [&](const TileOrigin& o) {
        const int iy0 = o.oy0 - 1, ix0 = o.ox0 - 1;
        const int base = ((o.img0 * H + iy0) * W + ix0) * (C * 4);     // may be negative; valid lanes' sums are not
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            int p = h_pk[i];
            asm volatile("" : "+v"(p));
            const int hy = (p >> 21) & 31, hx = (p >> 16) & 31, j = (p >> 26) & 15;
            const bool ok = (p >= 0) & ((unsigned)(iy0 + hy) < (unsigned)H) & ((unsigned)(ix0 + hx) < (unsigned)W);
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(base + (hy * W + hx) * (C * 4) + j * 16) : MIL_OOB, 0, 0);
        }
    };
    if (bid < ntiles) fetch(cur.origin(g));
    const int G = gridDim.x;
    MIL_STAMP_DECL(6)
    for (int tile = bid; tile < ntiles; tile += G) {
        const TileOrigin o = cur.origin(g);
        MIL_STAMP_BEGIN()
        __syncthreads();                       // the previous tile's fragment reads are done
        MIL_STAMP_MARK(0)
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            int p = h_pk[i];
            asm volatile("" : "+v"(p));
            const f32x4_t v = __builtin_bit_cast(f32x4_t, rx[i]);
            bf16x4_t h, l;
            mil_split4(v, h, l);
            const int l0 = p >= 0 ? (p & 0xFFFF) : dumpo;
            *reinterpret_cast<bf16x4_t*>(smem + l0) = h;
            *reinterpret_cast<bf16x4_t*>(smem + (p >= 0 ? l0 + PLANE : dumpo + 8)) = l;
        }
        MIL_STAMP_MARK(1)
        __syncthreads();                       // halo planes visible
        MIL_STAMP_MARK(2)

        f32x4_t acc[MTW][NT];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        {
            // the (k-step, row tile) loop flattened: pixel fragments one row-tile step ahead, the filter
            // fragments of k-step sl+1 requested from L1/L2 at the start of k-step sl (four row tiles = 36 MFMAs to land)
            constexpr int TOT = KSTEPS * MTW, LA = 1, R = LA + 1;      // one row-tile step (9 MFMAs) covers an LDS read; two spill
#ifndef MIL_STREAM_WD
#define MIL_STREAM_WD 1
#endif
            constexpr int WD = MIL_STREAM_WD, WR = WD + 1;             // filter fragments: k-steps ahead / ring slots
            Frag8<F32S> wq[WR][NT], ring[R];
            // The last column tile of a 40-channel filter holds channels 32-39: rows 8-15 idle (zero weights).  Lanes of rows 8-15 read
            // the LO half of row r-8 instead, so ONE fragment F = [w_hi ; w_lo] serves both planes — F x x_hi and F x x_lo, two MFMAs
            // for what took three (rows 8-15 also collect w_lo*x_lo, the 2^-18 term the three-product form drops) — and the epilogue
            // adds rows 8-15 (lanes 32-63) onto rows 0-7.  One filter load less per k-step, too.
            constexpr bool FOLD = (C % 16) == 8;
            const unsigned wfold = (unsigned)((r >= 8 ? lane - 8 : lane) * 32 + (r >= 8 ? 16 : 0));
            auto fetch_w = [&](int sl) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (FOLD && nt == NT - 1) {
                        wq[sl % WR][nt].h = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wfold, (sl * NT + nt) * 2048, 0));
                        continue;
                    }
                    wq[sl % WR][nt].h = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(lane * 32), (sl * NT + nt) * 2048, 0));
                    wq[sl % WR][nt].l = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(lane * 32 + 16), (sl * NT + nt) * 2048, 0));
                }
            };
            auto xfrag = [&](int j) {
                const char* p = smem + pixbase + koff[j / MTW] + (j % MTW) * (HW * PIXB);
                Frag8<F32S> f;
                f.h = *reinterpret_cast<const bf16x8_t*>(p);
                f.l = *reinterpret_cast<const bf16x8_t*>(p + PLANE);
                return f;
            };
#pragma unroll
            for (int k = 0; k < WD; ++k) fetch_w(k);
#pragma unroll
            for (int j = 0; j < LA; ++j) ring[j % R] = xfrag(j);
#pragma unroll
            for (int j = 0; j < TOT; ++j) {
                const int sl = j / MTW, m = j % MTW;
                if (j + LA < TOT) ring[(j + LA) % R] = xfrag(j + LA);
                if (m == 0 && sl + WD < KSTEPS) fetch_w(sl + WD);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (FOLD && nt == NT - 1) {
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[sl % WR][nt].h, ring[j % R].l, acc[m][nt], 0, 0, 0);
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[sl % WR][nt].h, ring[j % R].h, acc[m][nt], 0, 0, 0);
                    } else
                    acc[m][nt] = mma8(wq[sl % WR][nt], ring[j % R], acc[m][nt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (FOLD) {                  // rows 8-15 of the folded tile (lanes 32-63) onto rows 0-7
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[m][NT - 1][i] += __shfl_down(acc[m][NT - 1][i], 32, 64);
            }
        }
        MIL_STAMP_MARK(3)
        // ---- epilogue operands (16 bytes each), requested BEFORE the next tile's halo: loads return in order, so the epilogue waits
        // for its own pieces only while the 13 halo pieces stay in flight behind them (requested after the halo, every tile paid two
        // full memory round trips here: 43 % of the tile time).  The operand registers of the loop are free by now.
        const int obase = ((o.img0 * H + o.oy0) * W + o.ox0) * (C * 4);
        const int ylim = H - o.oy0 - wave * MTW, xok = r < W - o.ox0;
        f32x4_t bias_r[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bias_r[nt] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (unsigned)(nt * 64 + gq * 16), 0, 0));      // no bias: zeros
        constexpr int MH = (RES && ACT && !MIL_STREAM_FULL_EPI) ? MTW / 2 : MTW;             // both operands: 96 registers beside the halo pieces do not fit -> two halves
        u32x4_t rr[RES ? MH : 1][NT], ra[ACT ? MH : 1][NT];
        auto epi_off = [&](int m, int nt) {
            return (m < ylim && xok && nt * 16 + gq * 4 < C) ? (unsigned)(obase + o_rel + m * W * (C * 4) + nt * 64) : MIL_OOB;
        };
        auto epi_load = [&](int m0) {
#pragma unroll
            for (int mm = 0; mm < MH; ++mm)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const unsigned oo = epi_off(m0 + mm, nt);
                    if constexpr (RES) rr[mm][nt] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, oo, 0, 0);
                    if constexpr (ACT) ra[mm][nt] = __builtin_amdgcn_raw_buffer_load_b128(rs_act, oo, 0, 0);
                }
        };
        auto epi_store = [&](int m0) {
#pragma unroll
            for (int mm = 0; mm < MH; ++mm)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4_t v = acc[m0 + mm][nt];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += bias_r[nt][i];
                    if constexpr (RES) {
                        const f32x4_t t = __builtin_bit_cast(f32x4_t, rr[mm][nt]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += t[i];
                    }
                    if (a.lrelu) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], v[i] * a.slope);
                    }
                    if constexpr (ACT) {
                        const f32x4_t t = __builtin_bit_cast(f32x4_t, ra[mm][nt]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] *= (t[i] > 0.f ? 1.f : a.slope);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs_y, epi_off(m0 + mm, nt), 0, 0);
                }
        };
        epi_load(0);
        if (tile + G < ntiles) fetch(nxt.origin(g));
        cur = nxt; nxt.advance();
        MIL_STAMP_MARK(4)
        epi_store(0);
        if constexpr (MH < MTW) { epi_load(MH); epi_store(MH); }
        MIL_STAMP_MARK(5)
    }
    MIL_STAMP_STORE(a.stamp, 4)
}

// 3x3 stride-1 pad-1 C -> C conv on fp32 tensors with split products, maps of at least 16x16: MIL_ERR_UNSUPPORTED otherwise.
template <int C, int NT>
static int launch_stream_x3(StreamX3Args a, hipStream_t st) {
    constexpr int PIXB = mil_pix_pitch(C, 2), lds = 2 * 18 * 18 * PIXB + 64;
    ConvGeom& g = a.g;
    g.tw_log2 = 4; g.th_log2 = 4; g.ti_log2 = 0;
    g.tiles_x = (g.W + 15) >> 4; g.tiles_y = (g.H + 15) >> 4; g.n_groups = g.n_img;
    g.hh = 18; g.hw = 18;
    using Kern = void (*)(StreamX3Args, int, unsigned);
    const Kern kerns[4] = {conv_stream_x3_kernel<C, NT, false, false>, conv_stream_x3_kernel<C, NT, true, false>,
                           conv_stream_x3_kernel<C, NT, false, true>, conv_stream_x3_kernel<C, NT, true, true>};
    const Kern kern = kerns[(a.res ? 1 : 0) + (a.act ? 2 : 0)];
    static std::atomic<unsigned long long> attr_set{0};
    if (mil_device_needs(attr_set)) {
        for (const Kern k : kerns)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MIL_ERR_LAUNCH;
        mil_device_done(attr_set);
    }
    const int per_cu = mil_resident_per_cu(kern, lds, 3, 256);
    const size_t img = (size_t)g.H * g.W * C * 4;
    int chunk = mil_imgs_under_2g(img);
    const int n_total = g.n_img;
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        const int n = (n_total - i0 < chunk) ? n_total - i0 : chunk;
        StreamX3Args c = a;
        c.g.n_img = n; c.g.n_groups = n;
        c.x = a.x + (size_t)i0 * (img / 4); c.y = a.y + (size_t)i0 * (img / 4);
        if (a.res) c.res = a.res + (size_t)i0 * (img / 4);
        if (a.act) c.act = a.act + (size_t)i0 * (img / 4);
        const int ntiles = n * g.tiles_y * g.tiles_x;
        int grid = mil_num_cus() * per_cu;
        if (grid > ntiles) grid = ntiles;
#ifdef MIL_STAMP
        static MilStampBuf sb;
        c.stamp = sb.get((size_t)grid * 4 * 8);
#endif
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, c, ntiles, (unsigned)(img * n));
        MIL_CHECK_LAUNCH();
#ifdef MIL_STAMP
        static const char* const ph[6] = {"barrier-top", "commit", "barrier-x", "gemm", "fetch-issue", "epilogue"};
        sb.report("conv_stream_x3_kernel", grid, 4, 6, ph, st);
#endif
    }
    return MIL_OK;
}
