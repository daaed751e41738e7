// Forward of a stage-entry block's two stride-2 convs in one pass (bf16 path):
//   o1    = lrelu(conv3x3_s2(x) + b1)      (nnBlocks.py:176-177 with stride 2)
//   short = conv1x1_s2(x)                  (the bias-free projection shortcut, gbm/model.py:38-40; nnBlocks.py:182-183)
// Both read the same block input; run separately each of them fetches the full-resolution tensor from HBM (the
// projection to use a quarter of it).  Here the input halo tile is staged once per 128-pixel output tile and feeds
// two accumulator sets: the 3x3 filter over all taps, the 1x1 filter over the centre tap.  Persistent workgroups,
// register prefetch of the next halo, paired 16-byte register epilogue — the structure of conv_igemm_pf_kernel.
#include "pf_common.cuh"

struct S2EntryArgs {
    const __bf16* x;        // [n,H,W,CINP]
    const __bf16* w1;       // MIL_PACK_FWD fragments of the 3x3 filter  [ceil(9*CG/4)][NT][64][8]
    const __bf16* wp;       // MIL_PACK_FWD fragments of the 1x1 filter  [ceil(CG/4)][NT][64][8]
    const float* bias;      // [NT*16] bias of the 3x3 conv, or null
    __bf16* y1;             // [n,Ho,Wo,COUTP]
    __bf16* y2;             // [n,Ho,Wo,COUTP]
    ConvGeom g;
    int lds_w_off;
    float slope;
};

// MTW = 16-pixel row tiles per wave: 2 (128-pixel output tiles) or 1 (64-pixel tiles, for the 64 -> 80 channel entry whose
// halo tile and filters would not fit in LDS otherwise)
template <int CINP, int NT, int MTW = 2>
__global__ __launch_bounds__(256, CINP <= 24 ? 2 : 1) void conv_s2_entry_kernel(S2EntryArgs a, int ntiles, unsigned x_bytes, unsigned y_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int PIXB = mil_pix_pitch(CINP, 2);
    constexpr int CG = CINP / 8;
    constexpr int COUTP = mil_nt_to_cp(NT);
    constexpr int K1 = (9 * CG + 3) / 4, K2 = (CG + 3) / 4;
    constexpr int NPX = (324 * MTW * CG + 255) / 256;           // halos of 128-px tiles at stride 2: 17x33, 2 x 17x17, 8 x 9x9 (64-px: 17x17, 4 x 9x9)
    constexpr bool LAST_PARTIAL = (COUTP % 16) != 0;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    char* ldsA = smem;
    char* ldsW = smem + a.lds_w_off;
    mil_stage_filter(ldsW, a.w1, K1 * NT * 64 * 16, tid, 256);
    mil_stage_filter(ldsW + K1 * NT * 64 * 16, a.wp, K2 * NT * 64 * 16, tid, 256);
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, x_bytes);
    const __amdgpu_buffer_rsrc_t rs_y1 = mil_rsrc(a.y1, y_bytes);
    const __amdgpu_buffer_rsrc_t rs_y2 = mil_rsrc(a.y2, y_bytes);
    const int TW = 1 << g.tw_log2, TH = 1 << g.th_log2;

    HaloTables<NPX> ht;
    mil_build_halo_tables<CINP, NPX>(ht, g, tid);
    int toff[K1], toff2[K2];
#pragma unroll
    for (int sl = 0; sl < K1; ++sl) {
        const int q = 4 * sl + gq;
        int tap = q / CG, cg = q - tap * CG;
        if (tap >= 9) { tap = 0; cg = 0; }
        toff[sl] = ((tap / 3) * g.hw + (tap % 3)) * PIXB + cg * 16;
    }
#pragma unroll
    for (int sl = 0; sl < K2; ++sl) {
        const int q = 4 * sl + gq;
        toff2[sl] = (g.hw + 1) * PIXB + (q < CG ? q : 0) * 16;     // centre tap of the 3x3 window = the 1x1/s2 sample
    }
    int pixbase[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int tp = (wave * MTW + m) * 16 + r;
        const int tx = tp & (TW - 1), ty = (tp >> g.tw_log2) & (TH - 1), ti = tp >> (g.tw_log2 + g.th_log2);
        pixbase[m] = ((ti * g.hh + ty * 2) * g.hw + tx * 2) * PIXB;
    }
    // after the swap between row tiles 0 and 1 a lane holds channels 16*nt + 8*(gq>>1) .. +8 of pixel ((gq&1), r)
    // (MTW = 1: no partner row tile; a lane stores its own 4 channels 16*nt + 4*gq .. of pixel r)
    int o_rel, o_pos;
    {
        const int tp = MTW == 2 ? (wave * 2 + (gq & 1)) * 16 + r : wave * 16 + r;
        const int tx = tp & (TW - 1), ty = (tp >> g.tw_log2) & (TH - 1), ti = tp >> (g.tw_log2 + g.th_log2);
        o_rel = ((ti * g.Ho + ty) * g.Wo + tx) * (COUTP * 2) + (MTW == 2 ? (gq >> 1) * 16 : gq * 8);
        o_pos = (ti << 20) | (ty << 10) | tx;
    }
    const bool last_ok = !LAST_PARTIAL || (gq >> 1) == 0;
    f32x4_t bias_r[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias_r[nt][i] = a.bias ? a.bias[nt * 16 + gq * 4 + i] : 0.f;

    TileWalker cur, nxt;
    const int bid = mil_xcd_block_id();
    cur.init(g, bid, gridDim.x);
    nxt = cur; nxt.advance();
    u32x4_t rx[NPX];
    if (bid < ntiles) mil_fetch_halo<CINP, NPX>(rx, rs_x, ht, g, cur.origin(g));
    const int G = gridDim.x;
    for (int tile = bid; tile < ntiles; tile += G) {
        __syncthreads();
        mil_commit_halo<NPX>(rx, ldsA, ht);
        const TileOrigin o = cur.origin(g);
        __syncthreads();
        if (tile + G < ntiles) mil_fetch_halo<CINP, NPX>(rx, rs_x, ht, g, nxt.origin(g));
        cur = nxt; nxt.advance();

        f32x4_t acc1[MTW][NT], acc2[MTW][NT];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { acc1[m][nt] = bias_r[nt]; acc2[m][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#ifndef MIL_S2_ENTRY_NO_PIPE
        // both GEMMs as flattened software pipelines with fragments read two steps ahead (mil_conv_ring, pf_common.cuh)
        mil_conv_ring<NT, MTW, K1, 2>(acc1, ldsW, lane, [&](int sl, int m) { return ldsA + pixbase[m] + toff[sl]; });
        mil_conv_ring<NT, MTW, K2, 2>(acc2, ldsW + K1 * NT * 64 * 16, lane, [&](int sl, int m) { return ldsA + pixbase[m] + toff2[sl]; });
#else
#pragma unroll
        for (int sl = 0; sl < K1; ++sl) {
            Frag8<BF16> wf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = lds_frag<BF16>(ldsW + ((sl * NT + nt) * 64 + lane) * 16);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const Frag8<BF16> xf = lds_frag<BF16>(ldsA + pixbase[m] + toff[sl]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc1[m][nt] = mma8(wf[nt], xf, acc1[m][nt]);
            }
        }
#pragma unroll
        for (int sl = 0; sl < K2; ++sl) {
            Frag8<BF16> wf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = lds_frag<BF16>(ldsW + (((K1 + sl) * NT + nt) * 64 + lane) * 16);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const Frag8<BF16> xf = lds_frag<BF16>(ldsA + pixbase[m] + toff2[sl]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc2[m][nt] = mma8(wf[nt], xf, acc2[m][nt]);
            }
        }
#endif
        const int obase = ((o.img0 * g.Ho + o.oy0) * g.Wo + o.ox0) * (COUTP * 2);
        const bool ok = (o_pos >> 20) < g.n_img - o.img0 && ((o_pos >> 10) & 1023) < g.Ho - o.oy0 && (o_pos & 1023) < g.Wo - o.ox0;
        const unsigned ooff = ok ? (unsigned)(obase + o_rel) : MIL_OOB;
        if constexpr (MTW == 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bf16x4_t ov, ou;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float v = acc1[0][nt][i]; ov[i] = (__bf16)fmaxf(v, v * a.slope); ou[i] = (__bf16)acc2[0][nt][i]; }
                const unsigned off = (LAST_PARTIAL && nt == NT - 1 && gq >= 2) ? MIL_OOB : ooff + nt * 32;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, ov), rs_y1, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, ou), rs_y2, off, 0, 0);
            }
        } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v[8], u[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo = acc1[0][nt][i], hi = acc1[1][nt][i];
                if (i == 0) mil_swap16<true>(lo, hi); else mil_swap16<false>(lo, hi);
                v[i] = lo; v[4 + i] = hi;
                float lo2 = acc2[0][nt][i], hi2 = acc2[1][nt][i];
                if (i == 0) mil_swap16<true>(lo2, hi2); else mil_swap16<false>(lo2, hi2);
                u[i] = lo2; u[4 + i] = hi2;
            }
            bf16x8_t ov, ou;
#pragma unroll
            for (int i = 0; i < 8; ++i) { ov[i] = (__bf16)fmaxf(v[i], v[i] * a.slope); ou[i] = (__bf16)u[i]; }
            const unsigned off = (LAST_PARTIAL && nt == NT - 1 && !last_ok) ? MIL_OOB : ooff + nt * 32;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ov), rs_y1, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ou), rs_y2, off, 0, 0);
        }
        }
    }
}

template <int CINP, int NT, int MTW = 2>
static int launch_s2_entry(S2EntryArgs a, hipStream_t st) {
    constexpr int CG = CINP / 8, PIXB = mil_pix_pitch(CINP, 2), COUTP = mil_nt_to_cp(NT);
    constexpr int K1 = (9 * CG + 3) / 4, K2 = (CG + 3) / 4;
    mil_geom_tiles(a.g, MTW == 2 ? 7 : 6);
    const int halo_px = (a.g.hh * a.g.hw) << a.g.ti_log2;
    if (halo_px > 324 * MTW || a.g.hh >= 1024 || a.g.hw >= 1024) return MIL_ERR_UNSUPPORTED;
    const int a_bytes = (halo_px * PIXB + 15) & ~15;
    const int lds = a_bytes + (K1 + K2) * NT * 64 * 16;
    if (lds > 160 * 1024) return MIL_ERR_UNSUPPORTED;
    a.lds_w_off = a_bytes;
    auto kern = conv_s2_entry_kernel<CINP, NT, MTW>;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return MIL_ERR_LAUNCH;
    // the resident set by registers AND LDS: a grid sized by LDS alone (3 per CU at 178 VGPRs) ran its last third of
    // workgroups as a second round behind the 2 per CU that fit (217 us against 148 us of traffic at copy rate)
    const int per_cu = mil_resident_per_cu(kern, lds, 4);
    const size_t x_img = (size_t)a.g.H * a.g.W * CINP * 2, y_img = (size_t)a.g.Ho * a.g.Wo * COUTP * 2;
    int chunk = mil_imgs_under_2g(x_img > y_img ? x_img : y_img);
    if (chunk >= 16) chunk &= ~15;
    const int n_total = a.g.n_img;
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        const int n = (n_total - i0 < chunk) ? n_total - i0 : chunk;
        S2EntryArgs c = a;
        c.g.n_img = n;
        c.g.n_groups = (n + (1 << c.g.ti_log2) - 1) >> c.g.ti_log2;
        c.x = a.x + (size_t)i0 * (x_img / 2);
        c.y1 = a.y1 + (size_t)i0 * (y_img / 2);
        c.y2 = a.y2 + (size_t)i0 * (y_img / 2);
        const int ntiles = c.g.n_groups * c.g.tiles_y * c.g.tiles_x;
        int grid = mil_num_cus() * per_cu;
        if (grid > ntiles) grid = ntiles;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, c, ntiles, (unsigned)(x_img * n), (unsigned)(y_img * n));
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

#include <type_traits>
#include "conv_s2_entry_x3.cuh"

// y1 = lrelu(conv3x3_s2(x) + bias), y2 = conv1x1_s2(x); x [n,H,W,cin_p], y1/y2 [n,(H-1)/2+1,(W-1)/2+1,cout_p].
// bf16, (cin_p,cout_p) in {(24,40),(40,64),(64,80)}; MIL_DT_F32S (fp32 tensors, split products): the same three;
// otherwise MIL_ERR_UNSUPPORTED (caller: two mil_conv_igemm calls).
extern "C" int mil_conv_s2_entry(const void* x, const void* wpack3, const float* bias_pad, const void* wpack1, void* y1, void* y2,
                                 int n_img, int H, int W, int cin_p, int cout_p, float slope, int dtype, void* stream) {
    if (!x || !wpack3 || !wpack1 || !y1 || !y2 || n_img < 0 || H <= 0 || W <= 0) return MIL_ERR_ARG;
    if (dtype == MIL_DT_F32S) {
        if (slope < 0.f || slope >= 1.f || H >= 1024 || W >= 1024) return MIL_ERR_UNSUPPORTED;
        if (!((cin_p == 24 && cout_p == 40) || (cin_p == 40 && cout_p == 64) || (cin_p == 64 && cout_p == 80))) return MIL_ERR_UNSUPPORTED;
        if (n_img == 0) return MIL_OK;
        S2EntryX3Args b{};
        b.x = (const float*)x; b.w3 = (const char*)wpack3; b.wp = (const char*)wpack1; b.bias = bias_pad;
        b.y1 = (float*)y1; b.y2 = (float*)y2; b.slope = slope; b.xpx = cin_p * 4;
        b.g.n_img = n_img; b.g.H = H; b.g.W = W; b.g.Ho = (H - 1) / 2 + 1; b.g.Wo = (W - 1) / 2 + 1;
        b.g.ks = 3; b.g.stride = 2; b.g.pad = 1; b.g.zins = 0;
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        if (cin_p == 64) return launch_s2_entry_x3<64, 5, 64>(b, st);      // 83 KB of halo planes: one workgroup per CU (against two generic launches, 0.23 ms)
        return cin_p == 24 ? launch_s2_entry_x3<24, 3, 128>(b, st) : launch_s2_entry_x3<40, 4, 64>(b, st);
    }
    if (dtype != MIL_DT_BF16 || slope < 0.f || slope >= 1.f) return MIL_ERR_UNSUPPORTED;
    if (n_img == 0) return MIL_OK;
    S2EntryArgs a{};
    a.x = (const __bf16*)x; a.w1 = (const __bf16*)wpack3; a.wp = (const __bf16*)wpack1; a.bias = bias_pad;
    a.y1 = (__bf16*)y1; a.y2 = (__bf16*)y2; a.slope = slope;
    a.g.n_img = n_img; a.g.H = H; a.g.W = W; a.g.Ho = (H - 1) / 2 + 1; a.g.Wo = (W - 1) / 2 + 1;
    a.g.ks = 3; a.g.stride = 2; a.g.pad = 1; a.g.zins = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (cin_p == 24 && cout_p == 40) return launch_s2_entry<24, 3>(a, st);
    if (cin_p == 40 && cout_p == 64) return launch_s2_entry<40, 4>(a, st);
    if (cin_p == 64 && cout_p == 80) return launch_s2_entry<64, 5, 1>(a, st);
    return MIL_ERR_UNSUPPORTED;
}
