// Forward of a stage-entry block's two stride-2 convs in one pass, SPLIT PRECISION (MIL_DT_F32S: fp32 tensors, bf16x3 products):
//   y1 = lrelu(conv3x3_s2(x) + b1)      (nnBlocks.py:176-177 with stride 2)
//   y2 = conv1x1_s2(x)                  (the bias-free projection shortcut, gbm/model.py:38-40)
// Included by conv_s2_entry.hip.  The generic kernel these launches ran on before (conv_igemm_kernel<F32S,..,1>: 64-pixel
// tiles, synchronous loads, filter chunks through LDS) kept the matrix pipe 6-8 % busy: 0.45 / 0.31 ms per conv.
//
// A [hi | lo] filter pair (3x3 + 1x1) is 43 KB (20 -> 40 channels) / 139 KB (40 -> 64): next to a stride-2 halo tile in both
// planes it leaves one workgroup per CU or does not fit at all.  So the FILTER IS NOT STAGED: every wave streams the packed
// fragments of a k-step straight from L1/L2 into registers one k-step ahead (2 KB per wave-load and column tile, the fragment
// index in the scalar offset of the buffer load — the conv_resident_kernel idea), and LDS holds only the input halo planes:
// 54 KB (20 channels, 8x16 output pixels) / 46 KB (40 channels, 8x8): THREE 4-wave workgroups per CU.
//   * 20 input channels use the K20 order of geom.cuh (6 k-steps instead of 7; records [ch 0-15][ch 16-19][ch 16-19 of the
//     next pixel]); the 1x1 projection is the centre tap in the standard order (its third k-group reads [16-19 | next 16-19]
//     against zero weights for "channels 20-23").
//   * a lane of the D[channel][pixel] accumulators holds four consecutive output channels of a pixel = one 16-byte fp32 store.
#pragma once

struct S2EntryX3Args {
    const float* x;         // [n,H,W,xpx/4]
    const char* w3;         // MIL_PACK_FWD fragments of the 3x3 filter (MIL_DT_F32S): [7 + 6 | 12 k-steps][NT][64][32 B]
    const char* wp;         // MIL_PACK_FWD fragments of the 1x1 filter: [K2][NT][64][32 B]
    const float* bias;      // [NT*16] or null
    float* y1;              // [n,Ho,Wo,COUTP]
    float* y2;
    ConvGeom g;             // output tiles: 8x16 (TPX 128) or 8x8 (TPX 64) of one image
    float slope;
    int xpx;                // bytes per pixel of x
};

template <int CINP, int NT, int TPX>
__global__ __launch_bounds__(256, CINP >= 64 ? 1 : 2) void conv_s2_entry_x3_kernel(S2EntryX3Args a, int ntiles, unsigned x_bytes, unsigned y_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr bool K20 = CINP == 24;
    constexpr int CG = CINP / 8, COUTP = mil_nt_to_cp(NT), NTHR = 256;
    constexpr int PIXB = K20 ? 48 : mil_pix_pitch(CINP, 2);
    constexpr int TW = TPX == 128 ? 16 : 8, TH = 8;
    constexpr int HH = 2 * TH + 1, HWD = 2 * TW + 1, ROWB = HWD * PIXB;
    constexpr int SPARE = 16;
    constexpr int PLANE = SPARE + HH * HWD * PIXB;                   // hi plane, then lo plane
    constexpr int K1 = K20 ? MIL_K20_STEPS : (9 * CG + 3) / 4, K1_OFF = K20 ? 7 : 0;      // K20 section behind the 7 standard k-steps
    constexpr int K2 = (CG + 3) / 4;
    constexpr int MTW = TPX / 64;                                     // row tiles per wave
    constexpr int PPP = K20 ? 5 : CINP / 4;                           // 16-byte pieces (four fp32 channels) fetched per pixel
    constexpr int NPX = (HH * HWD * PPP + NTHR - 1) / NTHR;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    char* ldsX = smem + SPARE;
    const int dumpo = 2 * PLANE - SPARE;                              // dump slot (64 B) behind the planes, relative to ldsX
    const int H = g.H, W = g.W, Ho = g.Ho, Wo = g.Wo, XPX = a.xpx;
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, x_bytes);
    const __amdgpu_buffer_rsrc_t rs_y1 = mil_rsrc(a.y1, y_bytes);
    const __amdgpu_buffer_rsrc_t rs_y2 = mil_rsrc(a.y2, y_bytes);
    const __amdgpu_buffer_rsrc_t rs_w3 = mil_rsrc(a.w3, (K1_OFF + K1) * NT * 2048);
    const __amdgpu_buffer_rsrc_t rs_wp = mil_rsrc(a.wp, K2 * NT * 2048);
    if (K20 && tid < 2) *reinterpret_cast<u32x2_t*>(ldsX + tid * PLANE + (HH * HWD - 1) * PIXB + 40) = u32x2_t{0u, 0u};      // never-committed "next pixel" slot

    // ---- tile-invariant tables ------------------------------------------------------------------------------------------
    int h_pos[NPX], h_lds[NPX], h_rel[NPX];                           // halo pieces: flat id = tid + 256*i -> (pixel id/PPP, piece id%PPP)
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int idx = tid + NTHR * i;
        const int px = idx / PPP, j = idx - px * PPP;
        const int hy = px / HWD, hx = px - hy * HWD;
        const bool used = px < HH * HWD;
        h_pos[i] = used ? (j << 20) | (hy << 10) | hx : (int)0x80000000u;
        h_lds[i] = used ? px * PIXB + j * 8 : dumpo;
        h_rel[i] = used ? ((hy + 1) * W + hx + 1) * XPX + j * 16 : (int)MIL_OOB;      // relative to one row and one column before the halo origin
    }
    int k3[K1], k1o[K2];                                               // per-lane fragment offsets from the top-left tap's record
#pragma unroll
    for (int sl = 0; sl < K1; ++sl) {
        const int q = 4 * sl + gq;
        if constexpr (K20) {
            int o = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (gq == k) o = mil_k20_off(4 * sl + k, ROWB, PIXB);
            k3[sl] = o;
        } else {
            int tap = q / CG, cg = q - tap * CG;
            if (tap >= 9) { tap = 0; cg = 0; }                         // zero weights: any finite record
            k3[sl] = ((tap / 3) * HWD + (tap % 3)) * PIXB + cg * 16;
        }
    }
#pragma unroll
    for (int sl = 0; sl < K2; ++sl) {
        const int q = 4 * sl + gq;
        k1o[sl] = (HWD + 1) * PIXB + (q < CG ? q : 0) * 16;           // centre tap of the 3x3 window = the 1x1/s2 sample
    }
    int pixbase[MTW], o_rel[MTW], o_pos[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int tp = (wave * MTW + m) * 16 + r;
        const int tx = tp % TW, ty = tp / TW;
        pixbase[m] = (2 * ty * HWD + 2 * tx) * PIXB;
        o_rel[m] = (ty * Wo + tx) * (COUTP * 4) + gq * 16;
        o_pos[m] = (ty << 10) | tx;
    }
    f32x4_t bias_r[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias_r[nt][i] = a.bias ? a.bias[nt * 16 + gq * 4 + i] : 0.f;
    auto split4 = [](const f32x4_t& v, u32x2_t& hi, u32x2_t& lo) {
        bf16x4_t h, l;
        mil_split4(v, h, l);
        hi = __builtin_bit_cast(u32x2_t, h);
        lo = __builtin_bit_cast(u32x2_t, l);
    };
    // one GEMM: filter fragments of k-step sl+1 requested from L1/L2 while the MFMAs of k-step sl run; pixel fragments two
    // row-tile steps ahead
    auto gemm = [&](f32x4_t (&acc)[MTW][NT], const __amdgpu_buffer_rsrc_t& rs_w, int sl0, auto nk_c, const auto& koff) {
        constexpr int NK = decltype(nk_c)::value;
        Frag8<F32S> wq[2][NT], xq[2][MTW];
        auto fetch_w = [&](int sl) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                wq[sl & 1][nt].h = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(lane * 32), ((sl0 + sl) * NT + nt) * 2048, 0));
                wq[sl & 1][nt].l = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(lane * 32 + 16), ((sl0 + sl) * NT + nt) * 2048, 0));
            }
        };
        auto fetch_x = [&](int sl) {
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const char* p = ldsX + pixbase[m] + koff[sl];
                xq[sl & 1][m].h = *reinterpret_cast<const bf16x8_t*>(p);
                xq[sl & 1][m].l = *reinterpret_cast<const bf16x8_t*>(p + PLANE);
            }
        };
        fetch_w(0);
        fetch_x(0);
#pragma unroll
        for (int sl = 0; sl < NK; ++sl) {
            if (sl + 1 < NK) { fetch_w(sl + 1); fetch_x(sl + 1); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mma8(wq[sl & 1][nt], xq[sl & 1][m], acc[m][nt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    TileWalker cur, nxt;
    const int bid = mil_xcd_block_id();
    cur.init(g, bid, gridDim.x);
    nxt = cur; nxt.advance();
    u32x4_t rx[NPX];
    auto fetch = [&](const TileOrigin& o) {
        const int iy0 = 2 * o.oy0 - 1, ix0 = 2 * o.ox0 - 1;            // input pixel of halo (0,0): pad 1
        const int base = ((o.img0 * H + iy0 - 1) * W + ix0 - 1) * XPX;
        if (iy0 >= 1 && ix0 >= 1 && iy0 + HH <= H && ix0 + HWD <= W) { // interior tile (wave-uniform): base in the scalar offset
#pragma unroll
            for (int i = 0; i < NPX; ++i) rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)h_rel[i], base, 0);
            return;
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            int p = h_pos[i];
            asm volatile("" : "+v"(p));
            const int hy = (p >> 10) & 1023, hx = p & 1023;
            const bool ok = (p >= 0) & ((unsigned)(iy0 + hy) < (unsigned)H) & ((unsigned)(ix0 + hx) < (unsigned)W);
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(base + h_rel[i]) : MIL_OOB, 0, 0);
        }
    };
    if (bid < ntiles) fetch(cur.origin(g));
    const int G = gridDim.x;
    for (int tile = bid; tile < ntiles; tile += G) {
        const TileOrigin o = cur.origin(g);
        __syncthreads();                       // the previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            int p = h_pos[i];
            asm volatile("" : "+v"(p));
            u32x2_t hi, lo;
            split4(__builtin_bit_cast(f32x4_t, rx[i]), hi, lo);
            const int l0 = h_lds[i];
            *reinterpret_cast<u32x2_t*>(ldsX + l0) = hi;
            *reinterpret_cast<u32x2_t*>(ldsX + (l0 == dumpo ? l0 + 8 : l0 + PLANE)) = lo;
            if constexpr (K20) {               // channels 16-19: also the previous pixel's "next pixel" slot
                const bool dup = p >= 0 && ((p >> 20) & 7) == 4;
                const int l1 = dup ? l0 - 40 : dumpo;
                *reinterpret_cast<u32x2_t*>(ldsX + l1) = hi;
                *reinterpret_cast<u32x2_t*>(ldsX + (dup ? l1 + PLANE : l1 + 8)) = lo;
            }
        }
        __syncthreads();                       // input halo visible
        if (tile + G < ntiles) fetch(nxt.origin(g));
        cur = nxt; nxt.advance();

        f32x4_t acc1[MTW][NT], acc2[MTW][NT];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { acc1[m][nt] = bias_r[nt]; acc2[m][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        gemm(acc1, rs_w3, K1_OFF, std::integral_constant<int, K1>{}, k3);
        gemm(acc2, rs_wp, 0, std::integral_constant<int, K2>{}, k1o);

        const int obase = ((o.img0 * Ho + o.oy0) * Wo + o.ox0) * (COUTP * 4);
        const int ylim = Ho - o.oy0, xlim = Wo - o.ox0;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const bool ok = (o_pos[m] >> 10) < ylim && (o_pos[m] & 1023) < xlim;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool chan_ok = nt * 16 + gq * 4 < COUTP;
                const unsigned off = (ok && chan_ok) ? (unsigned)(obase + o_rel[m] + nt * 64) : MIL_OOB;
                f32x4_t v = acc1[m][nt];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], v[i] * a.slope);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs_y1, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc2[m][nt]), rs_y2, off, 0, 0);
            }
        }
    }
}

template <int CINP, int NT, int TPX>
static int launch_s2_entry_x3(S2EntryX3Args a, hipStream_t st) {
    constexpr bool K20 = CINP == 24;
    constexpr int PIXB = K20 ? 48 : mil_pix_pitch(CINP, 2), COUTP = mil_nt_to_cp(NT);
    constexpr int TW = TPX == 128 ? 16 : 8, HH = 17, HWD = 2 * TW + 1;
    constexpr int lds = 2 * (16 + HH * HWD * PIXB) + 64;
    ConvGeom& g = a.g;
    g.tw_log2 = TPX == 128 ? 4 : 3; g.th_log2 = 3; g.ti_log2 = 0;
    g.tiles_x = (g.Wo + TW - 1) / TW; g.tiles_y = (g.Ho + 7) >> 3; g.n_groups = g.n_img;
    g.hh = HH; g.hw = HWD;
    auto kern = conv_s2_entry_x3_kernel<CINP, NT, TPX>;
    static std::atomic<unsigned long long> attr_set{0};
    if (mil_device_needs(attr_set)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MIL_ERR_LAUNCH;
        mil_device_done(attr_set);
    }
    const int per_cu = mil_resident_per_cu(kern, lds, 3, 256);
    const size_t x_img = (size_t)g.H * g.W * a.xpx, y_img = (size_t)g.Ho * g.Wo * COUTP * 4;
    int chunk = mil_imgs_under_2g(x_img > y_img ? x_img : y_img);
    const int n_total = g.n_img;
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        const int n = (n_total - i0 < chunk) ? n_total - i0 : chunk;
        S2EntryX3Args c = a;
        c.g.n_img = n; c.g.n_groups = n;
        c.x = a.x + (size_t)i0 * (x_img / 4);
        c.y1 = a.y1 + (size_t)i0 * (y_img / 4);
        c.y2 = a.y2 + (size_t)i0 * (y_img / 4);
        const int ntiles = n * g.tiles_y * g.tiles_x;
        int grid = mil_num_cus() * per_cu;
        if (grid > ntiles) grid = ntiles;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, c, ntiles, (unsigned)(x_img * n), (unsigned)(y_img * n));
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}
