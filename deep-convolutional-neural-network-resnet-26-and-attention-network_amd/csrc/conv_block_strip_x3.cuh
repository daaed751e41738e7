// Forward of a whole identity-shortcut residual block of the 20-channel stage, SPLIT PRECISION, on 64-pixel-wide maps (the layer-1
// maps of 256 x 256 tiles), as a ROW WALK — nnBlocks.py:175-189:  o1 = lrelu(conv3x3(x) + b1),  y = lrelu(conv3x3(o1) + b2 + x).
// Included by conv_block_fwd.hip behind conv_block_fwd_x3.cuh (same arguments, same K20 filter section, same arithmetic per
// output element, the residual as hi + lo from LDS in both: bit-identical results).
//
// Why a second form.  conv_block_fwd_x3_kernel cuts an image into 16 x 8 tiles: every tile stages a 20 x 12 input halo (1.875 x
// its own pixels, re-fetched from HBM because fp32 halos of 512 resident workgroups do not fit the L2s: 1.41 GB fetched for a
// 0.67 GB input) and recomputes conv1 on an 18 x 10 mid tile (12 MFMA row tiles for 8 of output).  Here a workgroup owns a whole
// IMAGE and walks down its rows two at a time, the tile spanning the full width:
//   * the input and the mid activation live in two 4-row RINGS of [hi | lo] planes; a step loads the two NEW input rows
//     (12 KB, contiguous in the NHWC tensor), computes the two new mid rows from ring rows (old, old, new, new) and the two
//     output rows above them — every input pixel is fetched ONCE (1.0 x), conv1 is computed once per pixel (2 + 2 row tiles
//     per wave and step instead of 3 + 2), the commit splits 128 pixels per step instead of 240;
//   * the left / right zero padding is ONE shared record per row (row pitch 65 records: column 64 of a row is column -1 of
//     the next), the top / bottom padding is a zeroed ring row (input) and a mid row forced to zero by its `inside` test;
//   * ring rows wrap: a fragment's row offset is ((base + m + ky) & 3) * ROW, rebuilt per step from a 6-entry per-lane table
//     (12 mads), so the MFMA loop itself is the one of the tiled kernel.
// 81.7 KB of LDS (rings 2 x 25 KB, the two compact filters 31.5 KB): two 4-wave workgroups per CU as before.
// A step costs an image prologue (step 0 computes mid rows -1 and 0 only: +3 % of conv1) and whole images are the unit of
// work, so the launcher takes this form only when the images fill the resident workgroups evenly enough (launch_block_fwd_x3).
#pragma once
#ifndef MIL_STRIP_X3_LA
#define MIL_STRIP_X3_LA 2             // pixel fragments read this many (k-step, row tile) steps ahead of their MFMAs
#endif

__global__ __launch_bounds__(256, 2) void conv_block_strip_x3_kernel(BlockFwdX3Args a, int n_img, unsigned bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int NT = 2, NTHR = 256, KSTEPS = MIL_K20_STEPS, KSTEPS_STD = 7;
    constexpr int PIXB = 48, SW = 64, RP = SW + 1, ROW = RP * PIXB;  // 3120 bytes per ring row and plane
    constexpr int PLANE = (4 * RP + 1) * PIXB;                       // 12528: records 0, 65, 130, 195, 260 are the zero columns
    constexpr int WSTEP = 2048 + 512 + 64;
    constexpr int W_BYTES = KSTEPS * WSTEP;                          // 15744
    constexpr int OFF_X = 0, OFF_O = 2 * PLANE, OFF_W1 = OFF_O + 2 * PLANE, OFF_W2 = OFF_W1 + W_BYTES, OFF_DUMP = OFF_W2 + W_BYTES;
    constexpr int MT = 2;                                            // row tiles per wave and conv: ring row m, columns 16*wave ..
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    char* ldsX = smem + OFF_X;                                       // hi plane; lo plane PLANE behind
    char* ldsO = smem + OFF_O;
    const int H = a.g.H, APX = a.apx;
    const int S = ((H + 1) >> 1) + 1;                                // steps per image: the prologue + one per row pair

    // ---- filters: K20 section of each packed buffer -> compact LDS form (as conv_block_fwd_x3_kernel) ------------------------
    {
        const __amdgpu_buffer_rsrc_t rw1 = mil_rsrc(a.w1, (KSTEPS_STD + KSTEPS) * NT * 64 * 32);
        const __amdgpu_buffer_rsrc_t rw2 = mil_rsrc(a.w2, (KSTEPS_STD + KSTEPS) * NT * 64 * 32);
        constexpr int K20_OFF = KSTEPS_STD * NT * 64 * 32;
        for (int id = tid; id < 2 * KSTEPS * 164; id += NTHR) {
            const int f = id / (KSTEPS * 164), rem = id - f * (KSTEPS * 164);
            const int sl = rem / 164, p = rem - sl * 164;
            char* dst = smem + (f ? OFF_W2 : OFF_W1) + sl * WSTEP;
            u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
            int doff;
            if (p < 128) {
                doff = p * 16;
                v = __builtin_amdgcn_raw_buffer_load_b128(f ? rw2 : rw1, (unsigned)(K20_OFF + (sl * NT + 0) * 2048 + p * 16), 0, 0);
            } else if (p < 160) {
                const int e = (p - 128) >> 1, half = (p - 128) & 1;
                doff = 2048 + e * 32 + half * 16;
                v = __builtin_amdgcn_raw_buffer_load_b128(f ? rw2 : rw1, (unsigned)(K20_OFF + (sl * NT + 1) * 2048 + ((e >> 2) * 16 + (e & 3)) * 32 + half * 16), 0, 0);
            } else {
                doff = 2560 + (p - 160) * 16;
            }
            *reinterpret_cast<u32x4_t*>(dst + doff) = v;
        }
        // the zero columns of both rings (4 planes x 5 records x 48 B = 60 pieces); the commits only ever touch their bytes
        // 40-47 (the "next pixel" copy of column 0, which is what a tap on column -1 must see there)
        if (tid < 60) {
            const int pl = tid / 15, rem = tid - pl * 15, rec = rem / 3, pc = rem - rec * 3;
            *reinterpret_cast<u32x4_t*>(smem + pl * PLANE + rec * ROW + pc * 16) = u32x4_t{0u, 0u, 0u, 0u};
        } else if (tid >= 64 && tid < 80) {
            // ... and the "next pixel" slot of column 63 of every ring row: its next pixel is the zero column, which no commit
            // writes; the zero-weight half of k-groups 21-23 reads it (0 x whatever the previous kernel left there, NaNs included)
            const int pl = (tid - 64) >> 2, rr = (tid - 64) & 3;
            *reinterpret_cast<u32x2_t*>(smem + pl * PLANE + (rr * RP + SW) * PIXB + 40) = u32x2_t{0u, 0u};
        }
    }
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, bytes);
    const __amdgpu_buffer_rsrc_t rs_o = mil_rsrc(a.o1, bytes);
    const __amdgpu_buffer_rsrc_t rs_y = mil_rsrc(a.y, bytes);

    // ---- step-invariant tables --------------------------------------------------------------------------------------------
    // the two new input rows = 128 pixels x 5 pieces (16 B = four fp32 channels): flat id = tid + 256*i; the third slot of the
    // upper half of the threads is unused (offset MIL_OOB: no request, zeros; committed to the dump slot) — no branch around a
    // load: hipcc merges a conditionally loaded register through copies behind an s_waitcnt vmcnt(0), which turns the prefetch
    // into a blocking load (measured: 2.4 k cycles of "fetch issue" per step)
    constexpr int NPX = 3;
    int h_lds[NPX], h_rel[NPX];
    bool h_row1[NPX], h_dup[NPX];
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int idx = tid + NTHR * i;
        const bool used = idx < 2 * SW * 5;
        const int px = idx / 5, j = idx - px * 5;
        const int row = px >> 6, col = px & 63;
        h_lds[i] = used ? (row * RP + col + 1) * PIXB + j * 8 : -1;
        h_rel[i] = used ? px * APX + j * 16 : (int)MIL_OOB;
        h_row1[i] = row != 0;
        h_dup[i] = used && j == 4;
    }
    // K20 k-group q = 4*sl + gq: column / channel part of its offset, its filter row in the two low bits
    int kq[KSTEPS];
#pragma unroll
    for (int sl = 0; sl < KSTEPS; ++sl) {
        int v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = 4 * sl + k;
            const int ky = q < 18 ? (q >> 1) / 3 : q < 21 ? q - 18 : q - 21;
            if (gq == k) v = mil_k20_off(q, 0, PIXB) | ky;
        }
        kq[sl] = v;
    }
    const int col = wave * 16 + r;                                   // this lane's pixel column in both convs
    const int pb = col * PIXB;                                       // record under the top-left tap: column col - 1 = record col
    const int hsel = gq >> 1;                                        // conv2 epilogue pair: pixel (row gq & 1, col), channels 8*hsel.. (+16..19 for hsel 0)
    const bool last_ok = hsel == 0;
    const int wb0 = lane * 32;
    const int wb1a = r < 4 ? 2048 + (gq * 4 + r) * 32 : (r < 8 ? 2048 + (gq * 4 + r - 4) * 32 + 16 : 2576);
    const int wb1b = r < 4 ? 2048 + (gq * 4 + r) * 32 : 2576;
    f32x4_t b1r[NT], b2r[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b1r[nt][i] = a.b1 ? a.b1[nt * 16 + gq * 4 + i] : 0.f;
            b2r[nt][i] = a.b2 ? a.b2[nt * 16 + gq * 4 + i] : 0.f;
        }
    auto split4 = [](const f32x4_t& v, u32x2_t& hi, u32x2_t& lo) {
        bf16x4_t h, l;
        mil_split4(v, h, l);
        hi = __builtin_bit_cast(u32x2_t, h);
        lo = __builtin_bit_cast(u32x2_t, l);
    };
    int koff[MT][KSTEPS];                                            // per step: fragment offsets of ring row m
    auto conv = [&](f32x4_t (&acc)[MT][NT], const char* ldsW, const char* plane0) {
        constexpr int TOT = KSTEPS * MT, LA = MIL_STRIP_X3_LA, R = LA + 1;
        Frag8<F32S> ring[R], wq[2][NT];
        auto pfrag = [&](int j) {
            const char* p = plane0 + pb + koff[j % MT][j / MT];
            Frag8<F32S> f;
            f.h = *reinterpret_cast<const bf16x8_t*>(p);
            f.l = *reinterpret_cast<const bf16x8_t*>(p + PLANE);
            return f;
        };
        auto wfrag = [&](int sl, int nt) {
            Frag8<F32S> f;
            if (nt == 0) { f = lds_frag<F32S>(ldsW + sl * WSTEP + wb0); }
            else { f.h = *reinterpret_cast<const bf16x8_t*>(ldsW + sl * WSTEP + wb1a); f.l = *reinterpret_cast<const bf16x8_t*>(ldsW + sl * WSTEP + wb1b); }
            return f;
        };
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wq[0][nt] = wfrag(0, nt);
#pragma unroll
        for (int j = 0; j < LA; ++j) ring[j % R] = pfrag(j);
#pragma unroll
        for (int j = 0; j < TOT; ++j) {
            const int sl = j / MT, m = j % MT;
            if (j + LA < TOT) ring[(j + LA) % R] = pfrag(j + LA);
            if (m == 0 && sl + 1 < KSTEPS) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wq[(sl + 1) & 1][nt] = wfrag(sl + 1, nt);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[m][0] = mma8(wq[sl & 1][0], ring[j % R], acc[m][0]);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[sl & 1][1].h, ring[j % R].h, acc[m][1], 0, 0, 0);      // [wh ; wl] x xh
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[sl & 1][1].l, ring[j % R].l, acc[m][1], 0, 0, 0);      // wh x xl
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    u32x4_t rx[NPX];
    // input rows 2s, 2s+1 of image img (rows beyond the image: zeros = the bottom padding)
    auto fetch = [&](int img, int s) {
        const int y0 = 2 * s;
        const int base = (img * H + y0) * SW * APX;
        const bool ok0 = y0 < H, ok1 = y0 + 1 < H;
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const bool ok = (h_row1[i] ? ok1 : ok0) && h_rel[i] >= 0;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(base + h_rel[i]) : MIL_OOB, 0, 0);
        }
    };
    const int G = gridDim.x;
    int img = blockIdx.x, s = 0;
    if (img < n_img) fetch(img, 0);
    MIL_STAMP_DECL(9)
    while (img < n_img) {
        const int nb = (s & 1) * 2;                                  // ring rows the new input rows / new mid rows go to
        MIL_STAMP_BEGIN()
        __syncthreads();                       // previous step: conv2's reads of the mid ring and conv1's of the input ring are done
        MIL_STAMP_MARK(0)
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            u32x2_t hi, lo;
            split4(__builtin_bit_cast(f32x4_t, rx[i]), hi, lo);
            const bool used = h_lds[i] >= 0;
            const int d0 = used ? nb * ROW + h_lds[i] : OFF_DUMP, d1 = h_dup[i] ? d0 - 40 : OFF_DUMP;
            *reinterpret_cast<u32x2_t*>(ldsX + d0) = hi;
            *reinterpret_cast<u32x2_t*>(ldsX + (used ? d0 + PLANE : d0 + 8)) = lo;
            *reinterpret_cast<u32x2_t*>(ldsX + d1) = hi;            // channels 16-19: also the previous record's "next pixel" slot
            *reinterpret_cast<u32x2_t*>(ldsX + (h_dup[i] ? d1 + PLANE : d1 + 8)) = lo;
        }
        if (s == 0) {
            // input row -1 (ring row 3) is the top padding: zeros, records 195..260 of both planes (66 x 48 B = 198 pieces)
            for (int id = tid; id < 2 * 198; id += NTHR) {
                const int pl = id >= 198, p = id - pl * 198;
                *reinterpret_cast<u32x4_t*>(ldsX + pl * PLANE + 3 * ROW + p * 16) = u32x4_t{0u, 0u, 0u, 0u};
            }
        }
        MIL_STAMP_MARK(1)
        __syncthreads();                       // new input rows visible
        MIL_STAMP_MARK(2)
        int ns = s + 1, nimg = img;
        if (ns == S) { ns = 0; nimg += G; }
        if (nimg < n_img) fetch(nimg, ns);
        MIL_STAMP_MARK(3)
        // ring row of (conv row m, filter row ky): input rows 2s-2 .. 2s+1 sit in ring rows (2s+2 .. 2s+5) & 3, mid rows 2s-3 .. 2s
        // (one row later in their ring) in the same ring rows
        {
            const int base = nb ^ 2;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int sl = 0; sl < KSTEPS; ++sl)
                    koff[m][sl] = ((base + m + (kq[sl] & 3)) & 3) * ROW + (kq[sl] & ~3);
        }
        const int ibase = img * H * SW * APX;

        // ---- conv1: mid rows 2s-1, 2s -> the mid ring's rows nb, nb+1 (+ exact fp32 to o1) -----------------------------------------
        {
            f32x4_t acc[MT][NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = b1r[nt];
            conv(acc, smem + OFF_W1, ldsX);
            MIL_STAMP_MARK(4)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int jy = 2 * s - 1 + m;
                const bool inside = (unsigned)jy < (unsigned)H;      // wave-uniform: a mid row outside the image is conv2's zero padding
                const int sdst = ((nb + m) * RP + col + 1) * PIXB;
#ifdef MIL_EXP_STRIP_NO_O1
                const unsigned ooff = MIL_OOB;
#else
                const unsigned ooff = inside ? (unsigned)(ibase + (jy * SW + col) * APX + gq * 16) : MIL_OOB;
#endif
#pragma unroll
                for (int e = 0; e < 4; ++e) {            // column tile 1: rows 4-7 (the lane group gq == 1) hold wl*xh of rows 0-3
                    float t0 = acc[m][1][e], t1 = t0;
                    if (e == 0) mil_swap16<true>(t0, t1); else mil_swap16<false>(t0, t1);
                    acc[m][1][e] += t1;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float t = acc[m][nt][e]; v[e] = inside ? fmaxf(t, t * a.slope) : 0.f; }
                    u32x2_t hi, lo;
                    split4(v, hi, lo);
                    const bool real = nt == 0 || gq == 0;                // channels 16-19 sit in lanes gq == 0 of column tile 1
                    const int d0 = real ? sdst + nt * 32 + gq * 8 : OFF_DUMP - OFF_O;
                    *reinterpret_cast<u32x2_t*>(ldsO + d0) = hi;
                    *reinterpret_cast<u32x2_t*>(ldsO + (real ? d0 + PLANE : d0 + 8)) = lo;
                    if (nt == 1) {                                       // + the previous record's "next pixel" slot
                        const int d1 = gq == 0 ? sdst - 8 : OFF_DUMP - OFF_O;
                        *reinterpret_cast<u32x2_t*>(ldsO + d1) = hi;
                        *reinterpret_cast<u32x2_t*>(ldsO + (gq == 0 ? d1 + PLANE : d1 + 8)) = lo;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs_o,
                                                           (ooff == MIL_OOB || !real) ? MIL_OOB : ooff + nt * 64, 0, 0);
                    if (nt == 1 && APX == 96)                            // padded layout: the pixel's four padding channels
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{0u, 0u, 0u, 0u}, rs_o, (ooff == MIL_OOB || gq != 0) ? MIL_OOB : ooff + 80, 0, 0);
                }
            }
        }
        if (s > 0) {
            // output rows 2s-2, 2s-1 of this lane's pixel.  Their residual = input rows that are still in the input ring (rows
            // (nb ^ 2), + 1), as hi + lo: x to 2^-18 relative, the precision the products see.  (Re-read from the tensor as exact
            // fp32 they had left the L2 two steps after their fetch: 1.54 GB fetched per launch for a 0.81 GB input.)
            const int ey = 2 * s - 2 + (gq & 1);
            const unsigned eoff = ey < H ? (unsigned)(ibase + (ey * SW + col) * APX + hsel * 32) : MIL_OOB;
            const char* xrec = ldsX + ((nb ^ 2) + (gq & 1)) * ROW + (col + 1) * PIXB;
            const bf16x8_t xh0 = *reinterpret_cast<const bf16x8_t*>(xrec + hsel * 16), xl0 = *reinterpret_cast<const bf16x8_t*>(xrec + PLANE + hsel * 16);
            const bf16x4_t xh1 = *reinterpret_cast<const bf16x4_t*>(xrec + 32), xl1 = *reinterpret_cast<const bf16x4_t*>(xrec + PLANE + 32);
            MIL_STAMP_MARK(5)
            __syncthreads();                   // new mid rows visible
            MIL_STAMP_MARK(6)

            // ---- conv2 + residual + LeakyReLU -> y rows 2s-2, 2s-1 -----------------------------------------------------------------
            f32x4_t acc[MT][NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = b2r[nt];
            conv(acc, smem + OFF_W2, ldsO);
            MIL_STAMP_MARK(7)
            float v[8], u[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo = acc[0][0][i], hi = acc[1][0][i];
                if (i == 0) mil_swap16<true>(lo, hi); else mil_swap16<false>(lo, hi);
                v[i] = lo; v[4 + i] = hi;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo = acc[0][1][i], hi = acc[1][1][i];
                if (i == 0) mil_swap16<true>(lo, hi); else mil_swap16<false>(lo, hi);
                u[i] = lo + hi;                                    // rows 0-3 + rows 4-7 (wl x o1_hi) of the same pixel
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float q = v[i] + ((float)xh0[i] + (float)xl0[i]); v[i] = fmaxf(q, q * a.slope);
                q = v[4 + i] + ((float)xh0[4 + i] + (float)xl0[4 + i]); v[4 + i] = fmaxf(q, q * a.slope);
                q = u[i] + ((float)xh1[i] + (float)xl1[i]); u[i] = fmaxf(q, q * a.slope);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[0], v[1], v[2], v[3]}), rs_y, eoff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[4], v[5], v[6], v[7]}), rs_y, eoff == MIL_OOB ? MIL_OOB : eoff + 16, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{u[0], u[1], u[2], u[3]}), rs_y,
                                                   (eoff == MIL_OOB || !last_ok) ? MIL_OOB : eoff + 64, 0, 0);
            if (APX == 96)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{0u, 0u, 0u, 0u}, rs_y, (eoff == MIL_OOB || !last_ok) ? MIL_OOB : eoff + 80, 0, 0);
            MIL_STAMP_MARK(8)
        }
        img = nimg; s = ns;
    }
    MIL_STAMP_STORE(a.stamp, 4)
}

constexpr int MIL_STRIP_X3_LDS = 4 * ((4 * 65 + 1) * 48) + 2 * (MIL_K20_STEPS * (2048 + 512 + 64)) + 64;
