// The 20-channel fused stem forward (stem_fwd_pool_kernel, above) as a ROW WALK on 256-pixel-wide tiles — gbm/model.py:24-26,51-53.
// Included by stem_fused.hip; same packed filter, same SK6 k order, same in-register pooling and winner codes: bit-identical
// pooled map and winner records (tested against the tiled kernel).
//
// The tiled kernel gives a wave two pooled rows x 16 pooled columns of an 8 x 16 tile: five stem rows (the boundary row of two
// waves is computed by both: 20 rows per 17) + an edge tile = 11 MFMA pixel tiles for 32 pooled pixels, and a 20 x 38 s2d halo
// per 128 pooled pixels (5.9 s2d pixels converted per pooled pixel; the input is fetched 1.5 x).  Here a workgroup owns a whole
// IMAGE: wave w owns pooled columns 16w .. 16w+15 and the workgroup walks down two pooled rows (four stem rows) per step:
//   * the horizontal maxima of a step's LAST stem row are exactly what the next step needs of its first window row, and the
//     lane <-> column mapping does not move: they stay in eight registers per lane — 4 stem rows + the edge tile = 9 pixel
//     tiles per step and wave instead of 11;
//   * the s2d rows live in an LDS RING (8 rows bf16 / 7 rows of [hi | lo] records in split precision): a step converts the four
//     NEW s2d rows (eight image rows x three colours, each 1 KB contiguous) and re-uses three — 4.0 s2d pixels per pooled
//     pixel, every input byte fetched once;
//   * columns -4 .. -1 and 128 .. 129 of a ring row are the zero padding: never written after the kernel's start;
//   * ring rows wrap: the row part of a fragment address is a per-step scalar (seven of them), added to a per-lane constant.
// Steps per image: Ho/2 + 1 (step 0 only loads s2d rows -3 .. 0).  Whole images are the unit of work: the launcher takes this
// form when the images fill the resident workgroups evenly (mil_stem_walk_wanted; MIL_STEM_WALK = 0 / 1 is a TEST knob).
#pragma once

constexpr int SWK_XW = 134;                       // s2d columns -4 .. 129 of an image row
template <bool X3> __host__ __device__ constexpr int swk_nring() { return X3 ? 7 : 8; }
template <bool X3> __host__ __device__ constexpr int swk_rowb() { return SWK_XW * sf_xpix(X3); }            // 6432 / 10720
template <bool X3> __host__ __device__ constexpr int swk_lds_bytes() {
    return 64 + swk_nring<X3>() * swk_rowb<X3>() + 256 + 160 + (X3 ? 0 : MIL_SK6_STEPS * 2 * 64 * 16);     // 64224 / 75520
}

template <bool X3>
__global__ __launch_bounds__(256, 2) void stem_fwd_walk_kernel(StemFwdArgs a) {
    using T = typename std::conditional<X3, F32S, BF16>::type;
    constexpr int NT = 2, COUTP = 24, NTHR = 256, KSTEPS = MIL_SK6_STEPS, KSTEPS_STD = 8;
    constexpr int XPIX = sf_xpix(X3), ROWB = swk_rowb<X3>(), NRING = swk_nring<X3>(), XBYTES = NRING * ROWB;
    constexpr int SPARE = 64, FRAGB = X3 ? 32 : 16, OESZ = X3 ? 4 : 2;
    constexpr int NLOAD = 3;                                  // 4 s2d rows x 64 pairs x 3 colours = 768 load items
    constexpr int MT = 9;                                     // pixel tiles per wave and step: 4 stem rows x (even, odd) + the edge tile
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    MIL_POISON(smem_raw);
    char* smem = smem_raw;
    constexpr int dump = XBYTES;                              // behind the ring: where the unused second copies go
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid * 16; i < SPARE + XBYTES + 256; i += NTHR * 16)          // padding columns, "next pixel" slots: zero once
        *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);
    smem += SPARE;
    const int H = a.H, W = a.W, H2 = a.H2, Ho = a.Ho, Wo = a.Wo;
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, (unsigned)((size_t)a.n_img * 3 * H * W * 4));
    const __amdgpu_buffer_rsrc_t rs_p = mil_rsrc(a.pool, (unsigned)((size_t)a.n_img * Ho * Wo * COUTP * OESZ));
    const __amdgpu_buffer_rsrc_t rs_i = mil_rsrc(a.widx, (unsigned)((size_t)a.n_img * Ho * Wo * COUTP));
    constexpr int W_OFF = KSTEPS_STD * NT * 64 * FRAGB;       // the SK6 k-steps sit behind the eight standard ones
    const __amdgpu_buffer_rsrc_t rs_w = mil_rsrc(static_cast<const char*>(a.w) + W_OFF, KSTEPS * NT * 64 * FRAGB);

    // ---- load items: (s2d row 0..3 of the step, pair of s2d pixels = four image columns, colour) ----------------------------
    int l_col[NLOAD], l_rel[NLOAD], l_row[NLOAD];
    bool l_c2[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int idx = tid + NTHR * i;
        const int pair = idx & 63, t = idx >> 6, c = t % 3, row = t / 3;
        l_col[i] = (4 + 2 * pair) * XPIX + c * 8;
        l_rel[i] = ((c * H + 2 * row) * W + 4 * pair) * 4;
        l_row[i] = row;
        l_c2[i] = c == 2;
    }
    // ---- per-lane constants of the pooling layout ------------------------------------------------------------------------
    // stem pixel (row, column C = 32*wave + 2*r + par) reads s2d pixels (row - 2 + ty, C - 2 + tx) = ring column C + 2 + tx;
    // the lane group's k-group of k-step sl is q = 4*sl + gq of the SK6 order: filter row q / 6, column / channel part below
    const int lane_col = (32 * wave + 2 * r + 2) * XPIX;
    const int rhoE = r < 1 ? 1 : (r < 4 ? r : 4);               // edge tile: pixel lane = stem row rho of the step (lane 0 repeats row 1, lanes 5.. row 4)
    const int edge_col = (32 * wave + 1) * XPIX;                // stem column 32*wave - 1
    int kcol[KSTEPS], kcolE[KSTEPS], kty[KSTEPS];
#pragma unroll
    for (int sl = 0; sl < KSTEPS; ++sl) {
        int o = 0, ty = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (gq == k) { o = mil_sk6_off(4 * sl + k, 0, XPIX); ty = (4 * sl + k) / 6; }
        kcol[sl] = lane_col + o; kcolE[sl] = edge_col + o; kty[sl] = ty;
    }
    const unsigned cc_even = 2u * (r & 1), cc_odd = cc_even + 1u;
    const unsigned cc_edge = ((unsigned)((rhoE + 3) & 3) << 2) | 3u;
    char* ldsB = smem + XBYTES + 256;
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
    constexpr bool WLDS = !X3;
    char* ldsW = ldsB + 160;
    if constexpr (WLDS) mil_stage_filter(ldsW, static_cast<const char*>(a.w) + W_OFF, KSTEPS * NT * 64 * FRAGB, tid, NTHR);
    float sentv = -3.0e38f;
    asm volatile("" : "+v"(sentv));
    const float slope = a.slope;
    const int S = (Ho + 1) / 2 + 1;                             // steps per image

    u32x4_t r0[NLOAD], r1[NLOAD];
    // s2d rows 4s-3 .. 4s of image img (rows outside the image: zeros)
    auto fetch = [&](int img, int s) {
        const int y0 = 4 * s - 3;
        const int base = ((img * 3) * H + 2 * y0) * W * 4;       // negative for s = 0: its valid rows' sums are not
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const bool ok = (unsigned)(y0 + l_row[i]) < (unsigned)H2;
            const unsigned off = ok ? (unsigned)(base + l_rel[i]) : MIL_OOB;
            r0[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
            r1[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? off + (unsigned)(W * 4) : MIL_OOB, 0, 0);
        }
    };
    float h0[8];                                                // horizontal maxima of the previous step's last stem row
#pragma unroll
    for (int q = 0; q < 8; ++q) h0[q] = sentv;
    const int G = gridDim.x;
    int img = blockIdx.x, s = 0;
    if (img < a.n_img) fetch(img, 0);
    __syncthreads();
    MIL_STAMP_DECL(5)
    while (img < a.n_img) {
        const int base4 = (4 * s) % NRING;                       // ring row of s2d row 4s-3
        char* ldsX = smem;
        MIL_STAMP_BEGIN()
        // ---- the four new s2d rows: fp32 -> bf16 (hi / lo halves in split precision), channel = c*4 + dy*2 + dx ----------------
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
            int rr = base4 + l_row[i];
            rr = rr >= NRING ? rr - NRING : rr;
            char* dst = ldsX + rr * ROWB + l_col[i];
            *reinterpret_cast<bf16x4_t*>(dst) = pa;
            *reinterpret_cast<bf16x4_t*>(dst + XPIX) = pb;
            // colour 2 (s2d channels 8-11) a second time: behind the c2 of the pixel to the left ("c2 of the next pixel", SK6 order)
            char* dupb = l_c2[i] ? dst + 8 : ldsX + dump;              // pixel 2*pair     <- c2 of pixel 2*pair + 1
            char* dupa = l_c2[i] ? dst - XPIX + 8 : ldsX + dump + 64;  // pixel 2*pair - 1 <- c2 of pixel 2*pair (pair 0: the padding column -1)
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
        __syncthreads();                       // new rows visible
        MIL_STAMP_MARK(1)
        int ns = s + 1, nimg = img;
        if (ns == S) { ns = 0; nimg += G; }
        const bool has_next = nimg < a.n_img;
        if (s == 0) {
            if (has_next) fetch(nimg, ns);
#pragma unroll
            for (int q = 0; q < 8; ++q) h0[q] = sentv;           // stem row -1 of the image: the pool's padding
        } else {
            const int t = s - 1;                                 // pooled rows 2t, 2t+1 <- stem rows 4t-1 (h0), 4t .. 4t+3
            // ring row of (stem row index ri = 0..3 of the step, filter row ty): s2d row 4s-6+ri+ty
            int srow[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                int rr = base4 + (NRING - 3) + k;
                rr = rr >= 2 * NRING ? rr - 2 * NRING : (rr >= NRING ? rr - NRING : rr);
                srow[k] = rr * ROWB;
            }
            int koff[4][KSTEPS], koffE[KSTEPS];
#pragma unroll
            for (int sl = 0; sl < KSTEPS; ++sl) {
                const int ty0 = (4 * sl) / 6, ty1 = (4 * sl + 3) / 6;      // the k-step's filter rows: the same for all lanes, or two
#pragma unroll
                for (int ri = 0; ri < 4; ++ri)
                    koff[ri][sl] = (ty0 == ty1 ? srow[ri + ty0] : (kty[sl] == ty0 ? srow[ri + ty0] : srow[ri + ty1])) + kcol[sl];
                int e = srow[0 + ty0];
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    const int v = ty0 == ty1 ? srow[ri + ty0] : (kty[sl] == ty0 ? srow[ri + ty0] : srow[ri + ty1]);
                    e = rhoE == ri + 1 ? v : e;
                }
                koffE[sl] = e + kcolE[sl];
            }
            f32x4_t acc[MT][NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            {
                constexpr int TOT = KSTEPS * MT, LA = X3 ? 1 : MIL_SP_LA, R = LA + 1;
                constexpr int WD = X3 ? 1 : MIL_SP_WD, WR = WD + 1;
                constexpr int FETCH_AT = X3 ? -1 : 0;
                Frag8<T> ring[R], wq[WR][NT];
                constexpr bool FOLD = X3 && MIL_SP_X3_FOLD;
                const int w1h = (FOLD && r >= 4 && r < 8) ? (64 + lane - 4) * FRAGB + 16 : (64 + lane) * FRAGB;
                auto wfrag = [&](int sl, int nt) {
                    Frag8<T> f;
                    if constexpr (X3) {
                        f.h = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(nt == 1 ? w1h : lane * FRAGB), sl * NT * 64 * FRAGB, 0));
                        f.l = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(nt == 1 ? (64 + lane) * FRAGB + (FOLD ? 0 : 16) : lane * FRAGB + 16), sl * NT * 64 * FRAGB, 0));
                    } else {
                        f.v = *reinterpret_cast<const bf16x8_t*>(ldsW + ((sl * NT + nt) * 64 + lane) * FRAGB);
                    }
                    return f;
                };
                auto xaddr = [&](int j) -> const char* {
                    const int sl = j / MT, m = j % MT;
                    return m < 8 ? ldsX + koff[m >> 1][sl] + (m & 1) * XPIX : ldsX + koffE[sl];
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
                        if (has_next) fetch(nimg, ns);
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
                if (has_next) fetch(nimg, ns);
            }
            // the hand-written vector instructions below read MFMA results: the compiler pads its own, not those inside asm
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            MIL_STAMP_MARK(2)
            if constexpr (X3 && MIL_SP_X3_FOLD) {      // column tile 1: rows 4-7 (lane group 1) hold w_lo * x_hi of rows 0-3 — add them, zero the padding channels
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t0 = acc[m][1][i], t1 = t0;
                        mil_swap16<false>(t0, t1);
                        acc[m][1][i] = gq == 0 ? acc[m][1][i] + t1 : 0.f;
                    }
                }
            }
            // ---- position codes, horizontal maxima per stem row, vertical maximum per pooled row ------------------------------
            const int S0 = 4 * t - 1;                                // image row of stem row rho = 0
            float ek[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ek[q] = sp_key(acc[8][q >> 2][q & 3], cc_edge);
            float h[5][8];
#pragma unroll
            for (int q = 0; q < 8; ++q) h[0][q] = h0[q];
            auto hrow = [&](auto RHO) {
                constexpr int rho = decltype(RHO)::value;
                constexpr unsigned rc = (unsigned)((rho + 3) & 3) << 2;
                const unsigned ce = rc | cc_even, co = rc | cc_odd;
                float o[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    h[rho][q] = sp_key(acc[2 * (rho - 1)][q >> 2][q & 3], ce);
                    o[q] = sp_key(acc[2 * (rho - 1) + 1][q >> 2][q & 3], co);
                }
                sp_hmax8(h[rho], o);
                if (wave > 0) {                                      // column -1 of the image is padding: nothing to add
                    constexpr unsigned long long mask = 0x1111111111111111ull << (rho & 3);
                    sp_edge8<rho>(h[rho], ek, sentv, mask);
                }
                if ((unsigned)(S0 + rho) >= (unsigned)H2) {          // a stem row outside the image (the pool's padding)
#pragma unroll
                    for (int q = 0; q < 8; ++q) h[rho][q] = sentv;
                }
            };
            hrow(std::integral_constant<int, 1>{}); hrow(std::integral_constant<int, 2>{});
            hrow(std::integral_constant<int, 3>{}); hrow(std::integral_constant<int, 4>{});
            MIL_STAMP_MARK(3)
            // ---- winner decode, + bias, LeakyReLU, store ------------------------------------------------------------------------
            const f32x4_t bias0 = *reinterpret_cast<const f32x4_t*>(ldsB + gq * 16), bias1 = *reinterpret_cast<const f32x4_t*>(ldsB + 64 + gq * 16);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int py = 2 * t + p, px = 16 * wave + r;
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
#pragma unroll
            for (int q = 0; q < 8; ++q) h0[q] = h[4][q];            // the next step's window row above
            MIL_STAMP_MARK(4)
        }
        __syncthreads();                       // every wave is done reading the ring: the next step's rows may land
        img = nimg; s = ns;
    }
    MIL_STAMP_STORE(a.stamp, 4)
}

// Row walk when the tile is 256 pixels wide, no space-to-depth copy is kept, and whole images fill the resident workgroups
// evenly enough (cost: rounds x steps x time per step against rounds x tiles x time per tile).  MIL_STEM_WALK = 0 / 1 (read per call) is a
// TEST knob that forces either form.
static bool mil_stem_walk_wanted(const StemFwdArgs& a, int grid_cap) {
    if (a.W != 256 || a.xs || a.xs_in || (a.H & 1) || a.Wo != 64) return false;
    const char* e = getenv("MIL_STEM_WALK");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    const long tiles = (long)a.n_img * ((a.Ho + 7) / 8) * 4;
    const long cost_tile = (tiles + grid_cap - 1) / grid_cap * 6;           // measured: 6.5 / 11.0 us per tile, 5.4 / 9.0 us per step (bf16 / split)
    const long cost_walk = (long)((a.n_img + grid_cap - 1) / grid_cap) * ((a.Ho + 1) / 2 + 1) * 5;
    return cost_walk < cost_tile;
}

template <bool X3>
static int launch_stem_fwd_walk(StemFwdArgs a, hipStream_t st) {
    constexpr int COUTP = 24, OESZ = X3 ? 4 : 2;
    const int lds = swk_lds_bytes<X3>();
    auto kern = stem_fwd_walk_kernel<X3>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return MIL_ERR_LAUNCH;
    size_t per_img = (size_t)3 * a.H * a.W * 4;
    const size_t p_img = (size_t)a.Ho * a.Wo * COUTP * OESZ;
    if (p_img > per_img) per_img = p_img;
    const int chunk = mil_imgs_under_2g(per_img);
    const int n_total = a.n_img;
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        StemFwdArgs b = a;
        b.n_img = n_total - i0 < chunk ? n_total - i0 : chunk;
        b.x = a.x + (size_t)i0 * 3 * a.H * a.W;
        b.pool = static_cast<char*>(a.pool) + (size_t)i0 * a.Ho * a.Wo * COUTP * OESZ;
        b.widx = a.widx + (size_t)i0 * a.Ho * a.Wo * COUTP;
        int grid = mil_num_cus() * mil_resident_per_cu(kern, lds, 2);
        if (grid > b.n_img) grid = b.n_img;
#ifdef MIL_STAMP
        static MilStampBuf sb;
        b.stamp = sb.get((size_t)grid * 4 * 7);
#endif
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, b);
        MIL_CHECK_LAUNCH();
#ifdef MIL_STAMP
        static const char* const ph[5] = {"convert", "barrier-x", "gemm", "maxima", "decode-store"};
        sb.report(X3 ? "stem_fwd_walk_kernel<x3>" : "stem_fwd_walk_kernel", grid, 4, 5, ph, st);
#endif
    }
    return MIL_OK;
}
