// Data gradient of a stage-entry block's input (bf16 path): the transposed 3x3 stride-2 conv of dz1 PLUS the
// transposed 1x1 stride-2 projection of dz, times the lrelu' mask of the block input — one pass
// (reference: autograd of nnBlocks.py:175-189 for the blocks built at gbm/model.py:37-41).
//
// The zero-insert form (conv_igemm with zero_insert=1) stages a halo tile that is 3/4 zeros, multiplies all 9
// taps against it, and needs a second launch plus a full-resolution temporary for the projection.  Here the 256
// output pixels of a tile are split by parity class (py,px) = (y&1, x&1): each class is a 1/2/2/4-tap conv over
// the COMPACT dz map (geom.cuh: mil_s2_group), so the LDS tile is 81 pixels instead of 324, the MFMA work drops
// from 9 to 2.25 taps per output pixel, and the projection is 5..10 more K-groups of class (0,0) read from a
// second compact tile.  Wave w owns row tile w (16 pixels) of every class; the paired 16-byte register epilogue
// of the persistent conv kernel then stores two horizontally adjacent output pixels per lane pair.
#include "pf_common.cuh"
#ifndef MIL_DGRAD_S2_X3_NW8
#define MIL_DGRAD_S2_X3_NW8 1
#endif

template <typename T>
struct DgradS2Args {
    const typename T::elem* dz1;      // [n,h,w,CZ]   gradient of the 3x3/s2 conv's output
    const typename T::elem* dz2;      // [n,h,w,CZ]   gradient of the block output (input of the projection's transposed conv), or null
    const typename T::elem* w;        // MIL_PACK_DGRAD_S2 fragments [mil_s2_nsteps][NT][64][8] (F32S: [hi | lo] pairs)
    const typename T::elem* act;      // [n,H,W,CXP]  block input (lrelu' mask), or null
    typename T::elem* y;              // [n,H,W,CXP]
    ConvGeom g;             // output tiling: Ho=H, Wo=W (dx), H=h, W=w (dz)
    int ch, cw;             // compact halo extent per image: TH/2+1, TW/2+1
    int lds_z2_off, lds_w_off;
    float slope;
    int ypx;                // bytes per pixel of y: CXP*ESZ, or 40 / 80 when a 20-channel output is written dense (MIL_DT_BF16_DGRAD / MIL_DT_F32S_DGRAD)
    unsigned act_bytes;
};

// T = BF16, or F32S (MIL_DT_F32S: fp32 tensors, bf16x3 split products — the compact tiles hold [hi | lo] planes, three MFMAs per
// fragment pair, fp32 mask / output; the 40 -> 24 channel entry only: the larger filters do not fit LDS beside two compact tiles).
// STREAM (split precision): the parity-class filter is NOT staged in LDS — every wave reads the packed fragments of a k-step
// from L1/L2 into registers two k-steps ahead (2 KB per wave-load and column tile, the fragment index in the scalar offset of
// the buffer load).  A [hi | lo] filter of 57 / 123 / 205 KB (40 -> 24, 64 -> 40, 80 -> 64 channels) beside two compact
// tiles left ONE 4-wave workgroup per CU (one wave per SIMD, matrix pipe 16 % busy: 0.52 ms) or did not fit at all (the two
// larger entries ran the zero-insert forms: 0.65 + 0.35 ms for a quarter of useful MFMAs); with only the compact tiles in
// LDS (29-54 KB) two to four workgroups are resident (80 channels: one, for its registers); the 64 -> 40 and 80 -> 64 channel
// entries run this form (the 40 -> 24 entry measured 0.60 ms streamed: it keeps the staged filter).
// NW = 8 (split precision, 40 -> 24 channels, maps of at least 16x32): 512-pixel output tiles (16 rows x 32 columns, compact
// tiles of 9x17 pixels) on eight waves that share ONE staged filter — two waves per SIMD instead of the one the 4-wave form
// is left with (57 KB filter + 252 VGPRs: matrix pipe 0.14 busy, 0.52 ms).
template <typename T, int CZ, int NT, bool STREAM = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NW == 8 || (STREAM && CZ < 80) || (CZ <= 40 && !T::SPLIT)) ? 2 : 1) void conv_dgrad_s2_kernel(DgradS2Args<T> a, int ntiles, unsigned z_bytes, unsigned y_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int ESZ = T::ESZ, FRAGB = 8 * ESZ;
    constexpr int CG = CZ / 8;
    constexpr int N16 = CZ * ESZ / 16;                       // 16-byte pieces per dz pixel
    constexpr int JB = T::SPLIT ? 8 : 16;
    constexpr int PIXZ = mil_pix_pitch(CZ, ESZ);
    constexpr int CXP = mil_nt_to_cp(NT);
    constexpr int NS = mil_s2_nsteps(CG);
    constexpr int NTHR = 64 * NW;
    constexpr int NPZ = ((NW == 8 ? 153 : 144) * N16 + NTHR - 1) / NTHR;      // <= 144 compact halo pixels (16 images of 3x3); NW 8: 9x17
    constexpr bool LAST_PARTIAL = (CXP % 16) != 0;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    char* ldsZ = smem;
    char* ldsW = smem + a.lds_w_off;
    if constexpr (!STREAM) mil_stage_filter(ldsW, a.w, NS * NT * 64 * FRAGB, tid, NTHR);
    const __amdgpu_buffer_rsrc_t rs_w = mil_rsrc(a.w, NS * NT * 64 * FRAGB);
    const __amdgpu_buffer_rsrc_t rs_z1 = mil_rsrc(a.dz1, z_bytes);
    const __amdgpu_buffer_rsrc_t rs_z2 = mil_rsrc(a.dz2, a.dz2 ? z_bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_act = mil_rsrc(a.act, a.act ? a.act_bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_y = mil_rsrc(a.y, y_bytes);
    const int CH = a.ch, CW = a.cw, h = g.H, w = g.W, H = g.Ho, W = g.Wo;
    const int TW = 1 << g.tw_log2, TH = 1 << g.th_log2;

    // ---- tile-invariant tables ---------------------------------------------------------------------
    int z_pos[NPZ], z_lds[NPZ], z_rel[NPZ];                  // compact halo pieces (same for both sources)
    {
        const int total = ((CH * CW) << g.ti_log2) * N16;
#pragma unroll
        for (int i = 0; i < NPZ; ++i) {
            const int idx = tid + NTHR * i;
            z_pos[i] = -1; z_lds[i] = 0; z_rel[i] = 0;
            if (idx < total) {
                const int p = idx / N16, j = idx - p * N16;
                const int cx = p % CW, t = p / CW, cy = t % CH, ti = t / CH;
                z_pos[i] = (ti << 20) | (cy << 10) | cx;
                z_lds[i] = p * PIXZ + j * JB;
                z_rel[i] = ((ti * h + cy) * w + cx) * (CZ * ESZ) + j * 16;
            }
        }
    }
    // class pixel of this lane: index wave*16 + r of the 64 (ti, i, j) positions; the same for all four classes
    const int cp = wave * 16 + r;
    const int cj = cp & (TW / 2 - 1), ci = (cp >> (g.tw_log2 - 1)) & (TH / 2 - 1), cti = cp >> (g.tw_log2 + g.th_log2 - 2);
    const int pixbase = ((cti * CH + ci) * CW + cj) * PIXZ;
    int toff[NS];
    {
        int s = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int sl = 0; sl < mil_s2_steps(c, CG); ++sl, ++s) {
                const S2Group gr = mil_s2_group(c, 4 * sl + gq, CG);
                toff[s] = gr.valid ? gr.src * a.lds_z2_off + (gr.di * CW + gr.dj) * PIXZ + gr.cg * 16 : 0;
            }
        }
    }
    // epilogue: after one v_permlane16_swap per accumulator register between classes 2p and 2p+1 a lane holds 8
    // consecutive channels (16*nt + 8*(gq>>1) ...) of output pixel (2i + p, 2j + (gq&1))
    int o_rel[2], o_pos[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int oy = 2 * ci + p, ox = 2 * cj + (gq & 1);
        o_rel[p] = (cti * H + oy) * W + ox;                      // in pixels: act is read at CXP*2 bytes per pixel, y written at a.ypx
        o_pos[p] = (cti << 20) | (oy << 10) | ox;
    }
    const bool last_ok = !LAST_PARTIAL || (gq >> 1) == 0;

    TileWalker cur, nxt;
    const int bid = mil_xcd_block_id();
    cur.init(g, bid, gridDim.x);
    nxt = cur; nxt.advance();
    auto fetch_z = [&](u32x4_t (&r1)[NPZ], u32x4_t (&r2)[NPZ], const TileOrigin& o) {
        const int i0 = o.oy0 >> 1, j0 = o.ox0 >> 1;
        const int base = ((o.img0 * h + i0) * w + j0) * (CZ * ESZ);
        const int ylim = h - i0, xlim = w - j0, ilim = g.n_img - o.img0;
#pragma unroll
        for (int i = 0; i < NPZ; ++i) {
            const int p = z_pos[i];
            const bool ok = p >= 0 && (p >> 20) < ilim && ((p >> 10) & 1023) < ylim && (p & 1023) < xlim;
            const unsigned off = ok ? (unsigned)(base + z_rel[i]) : MIL_OOB;
            r1[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_z1, off, 0, 0);
            r2[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_z2, off, 0, 0);
        }
    };
    constexpr int NE = T::SPLIT ? 2 : 1;                     // 16-byte loads per 8 channels of the mask operand
    auto fetch_epi = [&](const TileOrigin& o, unsigned (&ooff)[2], u32x4_t (&ract)[2][NT][NE]) {
        const int obase = (o.img0 * H + o.oy0) * W + o.ox0;
        const int ylim = H - o.oy0, xlim = W - o.ox0, ilim = g.n_img - o.img0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool ok = (o_pos[p] >> 20) < ilim && ((o_pos[p] >> 10) & 1023) < ylim && (o_pos[p] & 1023) < xlim;
            const int opix = obase + o_rel[p];
            ooff[p] = ok ? (unsigned)(opix * a.ypx + (gq >> 1) * 8 * ESZ) : MIL_OOB;
            const unsigned aoff = ok ? (unsigned)(opix * (CXP * ESZ) + (gq >> 1) * 8 * ESZ) : MIL_OOB;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned off = (LAST_PARTIAL && nt == NT - 1 && !last_ok) ? MIL_OOB : aoff + nt * 16 * ESZ;
                if (a.act) {
                    ract[p][nt][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_act, off, 0, 0);
                    if constexpr (T::SPLIT) ract[p][nt][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_act, off == MIL_OOB ? MIL_OOB : off + 16, 0, 0);
                }
            }
        }
    };

    // STREAM: no register prefetch of the next tile (its 18-24 pieces and the mask operands do not fit beside the streamed filter
    // fragments: the loads are issued at the top of the tile and behind the MFMA loop; the other resident workgroups cover them)
    // (80 channels, streamed: ONE wave per SIMD — 64 accumulator + 64 fragment registers beside 25 tap offsets do not fit 256 —
    // so the next tile IS prefetched in registers, as in the staged form)
    constexpr bool NOPF = STREAM && CZ < 80;
    u32x4_t rz1[NPZ], rz2[NPZ];
    unsigned ooff_n[2];
    u32x4_t ract_n[2][NT][NE];
    if (!NOPF && bid < ntiles) {
        fetch_z(rz1, rz2, cur.origin(g));
        fetch_epi(cur.origin(g), ooff_n, ract_n);
    }
    const int G = gridDim.x;
    for (int tile = bid; tile < ntiles; tile += G) {
        const TileOrigin o_cur = cur.origin(g);
        __syncthreads();                       // every wave has finished reading the compact tiles of the previous tile
        if constexpr (NOPF) fetch_z(rz1, rz2, o_cur);
#pragma unroll
        for (int i = 0; i < NPZ; ++i) {
            if (z_pos[i] >= 0) {
                mil_commit_piece<T, CZ * 2>(ldsZ + z_lds[i], rz1[i]);
                mil_commit_piece<T, CZ * 2>(ldsZ + a.lds_z2_off + z_lds[i], rz2[i]);
            }
        }
        unsigned ooff[2];
        u32x4_t ract[2][NT][NE];
        if constexpr (!NOPF) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            ooff[p] = ooff_n[p];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < NE; ++e) ract[p][nt][e] = ract_n[p][nt][e];
        }
        }
        __syncthreads();
        if constexpr (!NOPF) {
        if (tile + G < ntiles) {
            fetch_z(rz1, rz2, nxt.origin(g));
            fetch_epi(nxt.origin(g), ooff_n, ract_n);
        }
        }
        cur = nxt; nxt.advance();

        f32x4_t acc[4][NT];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[c][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#ifndef MIL_DGRAD_S2_NO_PIPE
        {
            // one k-step ahead (the steps of the four parity classes flattened into one sequence): the fragments of step
            // s+1 are read before the MFMAs of step s and scheduling fences keep that order (see mil_conv_ring)
            auto cls = [](int st) { int c = 0; while (st >= mil_s2_steps(c, CG)) { st -= mil_s2_steps(c, CG); ++c; } return c; };
            constexpr int WD = (STREAM && NT < 4) ? 2 : 1, WR = WD + 1;      // filter fragments in flight: k-steps ahead / ring slots (8 VGPRs per split fragment)
            Frag8<T> xq[2], wq[WR][NT];
            auto wfrag = [&](int st, int nt) {
                if constexpr (STREAM) {                              // split precision only: [hi | lo] = two 16-byte loads per lane
                    Frag8<T> f;
                    f.h = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(lane * FRAGB), (st * NT + nt) * 64 * FRAGB, 0));
                    f.l = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(lane * FRAGB + 16), (st * NT + nt) * 64 * FRAGB, 0));
                    return f;
                } else {
                    return lds_frag<T>(ldsW + ((st * NT + nt) * 64 + lane) * FRAGB);
                }
            };
            xq[0] = lds_pix_frag<T, CZ * 2>(ldsZ + pixbase + toff[0]);
#pragma unroll
            for (int k = 0; k < WD && k < NS; ++k)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wq[k % WR][nt] = wfrag(k, nt);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                if (st + 1 < NS) xq[(st + 1) & 1] = lds_pix_frag<T, CZ * 2>(ldsZ + pixbase + toff[st + 1 < NS ? st + 1 : st]);
                if (st + WD < NS) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) wq[(st + WD) % WR][nt] = wfrag(st + WD, nt);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[cls(st)][nt] = mma8(wq[st % WR][nt], xq[st & 1], acc[cls(st)][nt]);          // D[channel][pixel]
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
        {
            int s = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int sl = 0; sl < mil_s2_steps(c, CG); ++sl, ++s) {
                    const Frag8<T> xf = lds_pix_frag<T, CZ * 2>(ldsZ + pixbase + toff[s]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const Frag8<T> wf = lds_frag<T>(ldsW + ((s * NT + nt) * 64 + lane) * FRAGB);
                        acc[c][nt] = mma8(wf, xf, acc[c][nt]);          // D[channel][pixel]
                    }
                }
            }
        }
#endif
        if constexpr (NOPF) fetch_epi(o_cur, ooff, ract);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
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
                const unsigned off = (LAST_PARTIAL && nt == NT - 1 && !last_ok) ? MIL_OOB : ooff[p] + nt * 16 * ESZ;
                if constexpr (T::SPLIT) {
                    if (a.act) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const f32x4_t t = __builtin_bit_cast(f32x4_t, ract[p][nt][e]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[4 * e + i] *= (t[i] > 0.f ? 1.f : a.slope);
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[0], v[1], v[2], v[3]}), rs_y, off, 0, 0);
                    // dense 20-channel output (a.ypx = 80): channels 20-23 of the last column tile do not exist
                    const bool skip2 = off == MIL_OOB || (LAST_PARTIAL && nt == NT - 1 && a.ypx != CXP * ESZ);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[4], v[5], v[6], v[7]}), rs_y, skip2 ? MIL_OOB : off + 16, 0, 0);
                } else {
                if (a.act) {
                    const bf16x8_t t = __builtin_bit_cast(bf16x8_t, ract[p][nt][0]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= ((float)t[i] > 0.f ? 1.f : a.slope);
                }
                bf16x8_t ov;
#pragma unroll
                for (int i = 0; i < 8; ++i) ov[i] = (__bf16)v[i];
                const u32x4_t ou = __builtin_bit_cast(u32x4_t, ov);
                if (LAST_PARTIAL && nt == NT - 1 && a.ypx != CXP * 2) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{ou[0], ou[1]}, rs_y, off, 0, 0);      // dense: channels 16-19 only
                else __builtin_amdgcn_raw_buffer_store_b128(ou, rs_y, off, 0, 0);
                }
            }
        }
    }
}

template <typename T, int CZ, int NT, bool STREAM = false, int NW = 4>
static int launch_dgrad_s2(DgradS2Args<T> a, hipStream_t st) {
    constexpr int ESZ = T::ESZ;
    constexpr int CG = CZ / 8, PIXZ = mil_pix_pitch(CZ, ESZ), CXP = mil_nt_to_cp(NT);
    if constexpr (NW == 8) mil_geom_set(a.g, 5, 4, 0);         // 16 x 32 output pixels of one image
    else
    mil_geom_tiles(a.g, 8);
    if (a.g.tw_log2 < 1 || a.g.th_log2 < 1) return MIL_ERR_UNSUPPORTED;
    a.ch = (1 << a.g.th_log2) / 2 + 1; a.cw = (1 << a.g.tw_log2) / 2 + 1;
    const int npx = (a.ch * a.cw) << a.g.ti_log2;
    if (npx > (NW == 8 ? 153 : 144)) return MIL_ERR_UNSUPPORTED;
    const int z_bytes = (npx * PIXZ + 15) & ~15;
    const int w_bytes = STREAM ? 0 : mil_s2_nsteps(CG) * NT * 64 * 8 * ESZ;
    a.lds_z2_off = z_bytes; a.lds_w_off = 2 * z_bytes;
    const int lds = 2 * z_bytes + w_bytes;
    if (lds > 160 * 1024) return MIL_ERR_UNSUPPORTED;
    auto kern = conv_dgrad_s2_kernel<T, CZ, NT, STREAM, NW>;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return MIL_ERR_LAUNCH;
    const int per_cu = mil_resident_per_cu(kern, lds, 4, 64 * NW);      // by registers AND LDS (see conv_s2_entry.hip)
    const size_t z_img = (size_t)a.g.H * a.g.W * CZ * ESZ, act_img = (size_t)a.g.Ho * a.g.Wo * CXP * ESZ, y_img = (size_t)a.g.Ho * a.g.Wo * a.ypx;
    int chunk = mil_imgs_under_2g(z_img > act_img ? z_img : act_img);
    if (chunk >= 16) chunk &= ~15;
    const int n_total = a.g.n_img;
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        const int n = (n_total - i0 < chunk) ? n_total - i0 : chunk;
        DgradS2Args<T> c = a;
        c.g.n_img = n;
        c.g.n_groups = (n + (1 << c.g.ti_log2) - 1) >> c.g.ti_log2;
        c.dz1 = a.dz1 + (size_t)i0 * (z_img / ESZ);
        if (a.dz2) c.dz2 = a.dz2 + (size_t)i0 * (z_img / ESZ);
        if (a.act) c.act = a.act + (size_t)i0 * (act_img / ESZ);
        c.act_bytes = (unsigned)(act_img * n);
        c.y = a.y + (size_t)i0 * (y_img / ESZ);
        const int ntiles = c.g.n_groups * c.g.tiles_y * c.g.tiles_x;
        int grid = mil_num_cus() * per_cu;
        if (grid > ntiles) grid = ntiles;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, st, c, ntiles, (unsigned)(z_img * n), (unsigned)(y_img * n));
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

// y[n,H,W,cx_p] = lrelu'(act) * ( conv3x3_s2^T(dz1) + conv1x1_s2^T(dz2) ), with wpack from
// mil_pack_conv_weights(mode MIL_PACK_DGRAD_S2: w = the 3x3 weight, bias argument = the 1x1 projection weight or null).
// dz1/dz2 [n,h,w,cz_p] with h = (H-1)/2+1, w = (W-1)/2+1.  bf16, (cz_p,cx_p) in {(40,24),(64,40),(80,64)}; with
// MIL_DT_BF16_DGRAD and (40,24) y is [n,H,W,20] (dense gradient layout of the 20-channel layer; act stays padded).
extern "C" int mil_conv_dgrad_s2(const void* dz1, const void* dz2, const void* wpack, const void* act, void* y, int n_img,
                                 int h, int w, int cz_p, int H, int W, int cx_p, float slope, int dtype, void* stream) {
    if (!dz1 || !wpack || !y || n_img < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return MIL_ERR_ARG;
    if ((dtype != MIL_DT_BF16 && dtype != MIL_DT_BF16_DGRAD && dtype != MIL_DT_F32S && dtype != MIL_DT_F32S_DGRAD) || h != (H - 1) / 2 + 1 || w != (W - 1) / 2 + 1 || slope < 0.f || slope >= 1.f) return MIL_ERR_UNSUPPORTED;
    if (n_img == 0) return MIL_OK;
    if (dtype == MIL_DT_F32S || dtype == MIL_DT_F32S_DGRAD) {      // fp32 tensors, bf16x3 products: all three entries, the filter streamed
        // (80 -> 64 channels: 64 accumulator + 64 fragment registers beside 25 per-lane tap offsets spill 58 VGPRs at two waves per
        // SIMD, so that entry runs ONE wave per SIMD with the next tile prefetched in registers — against 0.34 ms on the
        // zero-insert form of the generic kernel)
        if (!((cz_p == 40 && cx_p == 24) || (cz_p == 64 && cx_p == 40) || (cz_p == 80 && cx_p == 64))) return MIL_ERR_UNSUPPORTED;
        if (dtype == MIL_DT_F32S_DGRAD && cz_p != 40) return MIL_ERR_UNSUPPORTED;
        DgradS2Args<F32S> b{};
        b.dz1 = (const float*)dz1; b.dz2 = (const float*)dz2; b.w = (const float*)wpack; b.act = (const float*)act; b.y = (float*)y;
        b.g.n_img = n_img; b.g.H = h; b.g.W = w; b.g.Ho = H; b.g.Wo = W; b.g.ks = 3; b.g.stride = 1; b.g.pad = 1; b.g.zins = 1;
        b.slope = slope; b.ypx = dtype == MIL_DT_F32S_DGRAD ? 80 : cx_p * 4;      // y [n,H,W,20] dense fp32, or padded
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        if (cz_p == 40) {               // staged filter: 0.52 ms against 0.60 ms streamed; eight waves on one staged filter where the map allows
            if (MIL_DGRAD_S2_X3_NW8 && H >= 16 && W >= 32) return launch_dgrad_s2<F32S, 40, 2, false, 8>(b, st);
            return launch_dgrad_s2<F32S, 40, 2>(b, st);
        }
        if (cz_p == 64) return launch_dgrad_s2<F32S, 64, 3, true>(b, st);
        return launch_dgrad_s2<F32S, 80, 4, true>(b, st);
    }
    DgradS2Args<BF16> a{};
    a.dz1 = (const __bf16*)dz1; a.dz2 = (const __bf16*)dz2; a.w = (const __bf16*)wpack; a.act = (const __bf16*)act; a.y = (__bf16*)y;
    a.g.n_img = n_img; a.g.H = h; a.g.W = w; a.g.Ho = H; a.g.Wo = W; a.g.ks = 3; a.g.stride = 1; a.g.pad = 1; a.g.zins = 1;
    a.slope = slope;
    a.ypx = cx_p * 2;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIL_DT_BF16_DGRAD) {            // y [n,H,W,20] dense: the 40 -> 20 channel entry only
        if (cz_p != 40 || cx_p != 24) return MIL_ERR_UNSUPPORTED;
        a.ypx = 40;
    }
    if (cz_p == 40 && cx_p == 24) return launch_dgrad_s2<BF16, 40, 2>(a, st);
    if (cz_p == 64 && cx_p == 40) return launch_dgrad_s2<BF16, 64, 3>(a, st);
    if (cz_p == 80 && cx_p == 64) return launch_dgrad_s2<BF16, 80, 4>(a, st);
    return MIL_ERR_UNSUPPORTED;
}
