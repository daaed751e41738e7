// Fused backward of a 20 -> 20 channel 3x3 stride-1 conv on 16x16 tiles, SPLIT PRECISION (MIL_DT_F32S: fp32 tensors in HBM,
// bf16x3 products) — the layer-1 kernel of the path that meets the 1e-3 gate on the logits (nnBlocks.py:169-171's autograd:
// dx = (conv^T(dz, W) + addend) * lrelu'(x),  dW, db).  Included by conv_bwd_fused.hip (BwdFusedArgs, the slab reduction).
//
// Same pass as conv_bwd_fused16_kernel (compile-time geometry, K20 order, bias sums through a ones channel), re-shaped for
// operands that are twice as large and cost three MFMAs per fragment pair:
//   * LDS holds hi = bf16(v) and lo = bf16(v - hi) as two PLANES with the bf16 kernel's record layout each (dz halo: [ch 0-15]
//     [ch 16-19][ch 16-19 of the next pixel], 48-byte pitch, conflict-free for 16-lane b128 reads; x tile: [24 ch], channel 23
//     := 1 in the hi plane, 0 in the lo plane), so every fragment address of the bf16 kernel works for both planes with one
//     more immediate.  31 KB halo + 24 KB x tile + 24 KB filter = 80 KB: TWO workgroups per CU.
//   * 256 threads (four waves, one per SIMD), two waves per SIMD coming from two INDEPENDENT workgroups: while one sits in its
//     commit / barrier phase the other owns the matrix pipe — the 8-wave single-workgroup form of the generic kernel
//     (conv_bwd_fused_kernel<F32S,24,..>) marched all its waves through every phase together.  Each wave owns four row tiles
//     of the data gradient (24 MFMAs per k-step for 12 fragment reads) and three of the twelve weight-gradient row tiles (no
//     idle half as with eight waves), with 256 VGPRs to keep one step of operands ahead without spilling.
//   * tensors: dz / addend / dx at a RUN-TIME pixel stride a.gpx = 96 bytes (padded 24 channels) or 80 (dense 20 fp32
//     channels, MIL_DT_F32S_DGRAD: five 16-byte pieces per pixel and no padding traffic); x at a.xpx (96).
//   * a fetched 16-byte piece is four fp32 channels; it is split when it is committed to LDS (8 bytes into each plane).
#pragma once
#include "stamp.cuh"

template <bool ADD, bool MASK>
__global__ __launch_bounds__(256, 2) void conv_bwd_fused16x3_kernel(BwdFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int NTX = 2, KS = 3, NW = 4;
    constexpr int PIXB = 48, PIXX = 48, HW = 18, ROWB = HW * PIXB;
    constexpr int KSTEPS_STD = 7, KSTEPS = MIL_K20_STEPS, MTW = 4, MT = 12, MW = 3, NTHR = 256;
    constexpr int HALO0 = 16;                                     // spare bytes in front of each halo plane (pixel 0's back-copy)
    constexpr int A_PLANE = HALO0 + HW * HW * PIXB;               // 15568: hi plane, then lo plane
    constexpr int X_PLANE = 256 * PIXX;                           // 12288
    constexpr int NPH = (HW * HW * 5 + NTHR - 1) / NTHR;          // 7 halo pieces (16 B = four fp32 channels) per thread
    constexpr int NPXT = 5;                                       // x-tile pieces per thread (256 px * 5 / 256)
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    char* ldsA = smem + HALO0;                                    // hi plane records; lo plane A_PLANE behind
    char* ldsW = smem + a.lds_w_off;
    char* ldsX = smem + a.lds_x_off;                              // hi plane; lo plane X_PLANE behind
    const int dump = a.lds_dump_off;
    // the K20 k-steps sit behind the standard ones; a packed F32S fragment is 32 bytes per lane ([hi 8][lo 8])
    mil_stage_filter(ldsW, reinterpret_cast<const char*>(a.w) + KSTEPS_STD * NTX * 64 * 32, KSTEPS * NTX * 64 * 32, tid, NTHR);
    // static parts of the LDS images: channels 20-23 of every x record (hi: 0,0,0,1 — the ones channel of the bias sums; lo:
    // zeros) and the last halo record's "next pixel" slot, which no commit writes (zero weights read it: must be finite)
    *reinterpret_cast<u32x2_t*>(ldsX + tid * PIXX + 40) = u32x2_t{0u, 0x3f800000u};
    *reinterpret_cast<u32x2_t*>(ldsX + X_PLANE + tid * PIXX + 40) = u32x2_t{0u, 0u};
    if (tid < 2) *reinterpret_cast<u32x2_t*>(ldsA + tid * A_PLANE + (HW * HW - 1) * PIXB + 40) = u32x2_t{0u, 0u};

    const __amdgpu_buffer_rsrc_t rs_z = mil_rsrc(a.dz, a.z_bytes);
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rs_add = mil_rsrc(a.addend, a.addend ? a.g_bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_dx = mil_rsrc(a.dx, a.g_bytes);
    const int H = g.H, W = g.W, GPX = a.gpx, XPX = a.xpx;

    // ---- halo pieces of this thread: flat piece id = tid + 256*s -> (halo pixel id/5, piece id%5): consecutive lanes read
    // consecutive 16-byte pieces (80 contiguous bytes per pixel).  h_pos = j<<20 | hy<<10 | hx, negative when unused.
    // (tables in registers — this kernel has 256 VGPRs to spend —: a fetch is a bounds test and one add per piece, and a tile whose
    // halo lies inside the image takes neither: the tile's base rides in the scalar offset of the buffer load)
    // h_pk = LDS offset of the piece relative to ldsA (14 bits) | hx << 14 | hy << 19 | (piece == 4) << 24; negative when unused
    int h_pk[NPH], h_rel[NPH];
#pragma unroll
    for (int i = 0; i < NPH; ++i) {
        const int idx = tid + NTHR * i;
        const int px = idx / 5, j = idx - px * 5;
        const int hy = px / HW, hx = px - hy * HW;
        const bool used = px < HW * HW;
        h_pk[i] = used ? (px * PIXB + j * 8) | (hx << 14) | (hy << 19) | ((j == 4) << 24) : (int)0x80000000u;
        h_rel[i] = used ? ((hy + 1) * W + hx + 1) * GPX + j * 16 : (int)MIL_OOB;      // relative to the pixel one row and one column before the halo origin: never negative
    }
    u32x4_t rz[NPH];
    auto fetch_halo = [&](const TileOrigin& o) {
        const int iy0 = o.oy0 - 1, ix0 = o.ox0 - 1;
        const int base = ((o.img0 * H + iy0 - 1) * W + ix0 - 1) * GPX;         // may be negative; valid lanes' sums are not
        if (iy0 >= 1 && ix0 >= 1 && iy0 + HW <= H && ix0 + HW <= W) {          // interior tile (wave-uniform): no per-piece test, base in the scalar offset
#pragma unroll
            for (int i = 0; i < NPH; ++i) rz[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, (unsigned)h_rel[i], base, 0);
            return;
        }
#pragma unroll
        for (int i = 0; i < NPH; ++i) {
            int p = h_pk[i];
            asm volatile("" : "+v"(p));
            const int hy = (p >> 19) & 31, hx = (p >> 14) & 31;
            const bool ok = (p >= 0) & ((unsigned)(iy0 + hy) < (unsigned)H) & ((unsigned)(ix0 + hx) < (unsigned)W);
            rz[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, ok ? (unsigned)(base + h_rel[i]) : MIL_OOB, 0, 0);
        }
    };
    // ---- x-tile pieces: flat piece id = tid + 256*s -> (tile pixel id/5, piece id%5); the padding piece of a 96-byte record
    // is never read
    int x_pk[NPXT], x_rel[NPXT];                                     // x_pk = LDS offset (14 bits) | tx << 14 | ty << 18
#pragma unroll
    for (int i = 0; i < NPXT; ++i) {
        const int idx = tid + NTHR * i;
        const int px = idx / 5, j = idx - px * 5;
        x_pk[i] = (px * PIXX + j * 8) | ((px & 15) << 14) | ((px >> 4) << 18);
        x_rel[i] = ((px >> 4) * W + (px & 15)) * XPX + j * 16;
    }
    u32x4_t rxt[NPXT];
    auto fetch_x = [&](const TileOrigin& o) {
        const int base = ((o.img0 * H + o.oy0) * W + o.ox0) * XPX;
        const int ylim = H - o.oy0, xlim = W - o.ox0;
        if (ylim >= 16 && xlim >= 16) {                                 // whole tile inside the image (wave-uniform)
#pragma unroll
            for (int i = 0; i < NPXT; ++i) rxt[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)x_rel[i], base, 0);
            return;
        }
#pragma unroll
        for (int i = 0; i < NPXT; ++i) {
            int p = x_pk[i];
            asm volatile("" : "+v"(p));
            const bool ok = ((p >> 18) < ylim) & (((p >> 14) & 15) < xlim);
            rxt[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(base + x_rel[i]) : MIL_OOB, 0, 0);
        }
    };
    auto split4 = [](const u32x4_t& rr, u32x2_t& hi, u32x2_t& lo) {
        const f32x4_t v = __builtin_bit_cast(f32x4_t, rr);
        bf16x4_t h, l;
        mil_split4(v, h, l);
        hi = __builtin_bit_cast(u32x2_t, h);
        lo = __builtin_bit_cast(u32x2_t, l);
    };

    // ---- data gradient: fragment address = per-lane base + immediate (K20 order, geom.cuh) ------------------------------
    auto koff = [](int q) { return mil_k20_off(q, ROWB, PIXB); };
    const int bA = (wave * MTW * HW + r) * PIXB + 16 * gq;         // pixel (tile row 4*wave [+m], col r), lane-group part
    int zb[KSTEPS];
#pragma unroll
    for (int sl = 0; sl < KSTEPS; ++sl) {
        const int o0 = koff(4 * sl);
        const int d1 = koff(4 * sl + 1) - o0 - 16, d2 = koff(4 * sl + 2) - o0 - 32, d3 = koff(4 * sl + 3) - o0 - 48;
        zb[sl] = bA + (gq == 1 ? d1 : gq == 2 ? d2 : gq == 3 ? d3 : 0);
    }
    // ---- epilogue pairs: after the permlane swap a lane holds 8 consecutive channels of pixel (4*wave + 2p + (gq&1), r) ----
    constexpr int NPAIR = MTW / 2;
    const int hsel = gq >> 1;                                      // 0: channels 0-7 and 16-19, 1: channels 8-15
    const bool last_ok = hsel == 0;
    const int c_off = hsel * 32;                                   // byte offset of the lane's first channel in an fp32 pixel
    // ---- weight gradient: row pieces (tap', four dz channels) of row tiles mt = wave + 4*i; K = the 256 centre pixels
    const int q4 = (lane & 15) >> 2, p4 = lane & 3;
    int zw[MW][2];
    {
        const int kl0 = 8 * gq + q4, kl1 = kl0 + 4;               // pixel of this lane inside a 32-pixel k-step
        const int wpl0 = ((kl0 >> 4) * HW + (kl0 & 15)) * PIXB, wpl1 = ((kl1 >> 4) * HW + (kl1 & 15)) * PIXB;
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            int P = 4 * (wave + NW * i) + p4;                      // rows tap'*20 + co in four-row pieces
            if (P >= KS * KS * 5) P = 0;                           // rows that do not exist: finite data, never reduced
            const int tap = P / 5, c4 = P - tap * 5;
            const int wt = ((tap / KS) * HW + (tap % KS)) * PIXB + c4 * 8;
            zw[i][0] = wpl0 + wt; zw[i][1] = wpl1 + wt;
        }
    }
    const int xb = (8 * gq + q4) * PIXX + p4 * 8;
    // column tile 1 of the weight gradient = x channels 16-23: four real channels, three padding, the ones channel (bias sums) —
    // its columns 8-15 are idle.  The lanes that read the 4-channel piece of columns 8-11 (p4 == 2) read the LO plane's channels
    // 16-19 instead: z_hi x [x_hi | x_lo] and z_lo x [x_hi | x_lo] are two MFMAs for what took three (z_lo*x_hi, z_hi*x_lo,
    // z_hi*x_hi); columns 8-11 are added onto 0-3 when the slab is written.  (They also carry z_lo*x_lo, the 2^-18 term the
    // three-product form drops.)
    const int xb1 = p4 == 2 ? X_PLANE + (8 * gq + q4) * PIXX + 32 : xb + 32;
    f32x4_t wacc[MW][NTX];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int nt = 0; nt < NTX; ++nt) wacc[i][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    TileWalker cur, nxt;
    const int bid = mil_xcd_block_id();
    cur.init(g, bid, gridDim.x);
    nxt = cur; nxt.advance();
    unsigned goff[NPAIR], goff_n[NPAIR];
    u32x4_t radd[NPAIR][3];
    auto set_goff = [&](const TileOrigin& o) {
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
            const int e_ty = wave * MTW + 2 * p + (gq & 1);
            const bool ok = e_ty < H - o.oy0 && r < W - o.ox0;
            goff_n[p] = ok ? (unsigned)(((o.img0 * H + o.oy0 + e_ty) * W + o.ox0 + r) * GPX + c_off) : MIL_OOB;
        }
    };
    auto fetch_add = [&]() {
        if constexpr (ADD) {
#pragma unroll
            for (int p = 0; p < NPAIR; ++p) {
                const unsigned o0 = goff_n[p];
                radd[p][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_add, o0, 0, 0);
                radd[p][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_add, o0 == MIL_OOB ? MIL_OOB : o0 + 16, 0, 0);
                radd[p][2] = __builtin_amdgcn_raw_buffer_load_b128(rs_add, (o0 == MIL_OOB || !last_ok) ? MIL_OOB : o0 + 64, 0, 0);
            }
        }
    };
    if (bid < a.ntiles) {
        const TileOrigin o0 = cur.origin(g);
        fetch_halo(o0); fetch_x(o0); set_goff(o0); fetch_add();
    }

    MIL_STAMP_DECL(8)
    for (int tile = bid; tile < a.ntiles; tile += gridDim.x) {
        MIL_STAMP_BEGIN()
        __syncthreads();                       // every wave has left the previous tile's loops: the images may be overwritten
        MIL_STAMP_MARK(0)
        // ---- commit: dz halo (both planes; piece 4 also into the previous pixel's "next pixel" slot) and x tile -------------
#pragma unroll
        for (int i = 0; i < NPH; ++i) {
            int p = h_pk[i];
            asm volatile("" : "+v"(p));                                // keeps the values derived from p out of loop-long registers
            u32x2_t hi, lo;
            split4(rz[i], hi, lo);
            const int l0 = p >= 0 ? (p & 0x3FFF) : dump - HALO0;
            const int l1 = (p >= 0 && ((p >> 24) & 1)) ? l0 - 40 : dump - HALO0;
            *reinterpret_cast<u32x2_t*>(ldsA + l0) = hi;
            *reinterpret_cast<u32x2_t*>(ldsA + A_PLANE + l0) = lo;
            *reinterpret_cast<u32x2_t*>(ldsA + l1) = hi;
            *reinterpret_cast<u32x2_t*>(ldsA + (l1 == dump - HALO0 ? l1 + 8 : l1 + A_PLANE)) = lo;
        }
#pragma unroll
        for (int i = 0; i < NPXT; ++i) {
            u32x2_t hi, lo;
            split4(rxt[i], hi, lo);
            int xp = x_pk[i];
            asm volatile("" : "+v"(xp));
            *reinterpret_cast<u32x2_t*>(ldsX + (xp & 0x3FFF)) = hi;
            *reinterpret_cast<u32x2_t*>(ldsX + X_PLANE + (xp & 0x3FFF)) = lo;
        }
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) goff[p] = goff_n[p];
        MIL_STAMP_MARK(1)
        __syncthreads();                       // dz halo and x tile visible
        MIL_STAMP_MARK(2)
        const bool more = tile + (int)gridDim.x < a.ntiles;
        const TileOrigin o_next = nxt.origin(g);
        if (more) { fetch_halo(o_next); fetch_x(o_next); }
        cur = nxt; nxt.advance();
        MIL_STAMP_MARK(3)

        // ---- data gradient D[cx][pixel]: a (k-step, row tile) pipeline, fragment reads two row tiles ahead -----------------
        f32x4_t acc[MTW][NTX];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) acc[m][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        {
            constexpr int TOT = KSTEPS * MTW, LA = 2, R = LA + 1;
            // column tile 1 holds channels 16-19 only (rows 0-3 of its filter fragments): rows 4-7 of its hi fragment are loaded
            // with the LO weights of the same channels (lanes 4..7 of a lane group read lane - 4's lo half), so ONE MFMA against
            // the pixels' hi plane yields wh*zh (rows 0-3) and wl*zh (rows 4-7); the second multiplies wh by the lo plane.  Five
            // MFMAs per fragment pair instead of six; the epilogue adds rows 4-7 to rows 0-3 (they meet in the permlane swap).
            Frag8<F32S> ring[R], wq[2][NTX];
            const int w1a = (r >= 4 && r < 8) ? (lane - 4) * 32 + 16 : lane * 32;
            auto wfrag = [&](int sl, int nt) {
                const char* p = ldsW + (sl * NTX + nt) * 64 * 32;
                Frag8<F32S> f;
                if (nt == 0) { f = lds_frag<F32S>(p + lane * 32); }
                else { f.h = *reinterpret_cast<const bf16x8_t*>(p + w1a); f.l = *reinterpret_cast<const bf16x8_t*>(p + lane * 32); }
                return f;
            };
            auto zfrag = [&](int j) {
                const int sl = j / MTW, m = j % MTW;
                const char* p = ldsA + zb[sl] + koff(4 * sl) + m * ROWB;
                Frag8<F32S> f;
                f.h = *reinterpret_cast<const bf16x8_t*>(p);
                f.l = *reinterpret_cast<const bf16x8_t*>(p + A_PLANE);
                return f;
            };
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) wq[0][nt] = wfrag(0, nt);
#pragma unroll
            for (int j = 0; j < LA; ++j) ring[j % R] = zfrag(j);
#pragma unroll
            for (int j = 0; j < TOT; ++j) {
                const int sl = j / MTW, m = j % MTW;
                if (j + LA < TOT) ring[(j + LA) % R] = zfrag(j + LA);
                if (m == 0 && sl + 1 < KSTEPS) {
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt) wq[(sl + 1) & 1][nt] = wfrag(sl + 1, nt);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[m][0] = mma8(wq[sl & 1][0], ring[j % R], acc[m][0]);
                acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[sl & 1][1].h, ring[j % R].h, acc[m][1], 0, 0, 0);      // [wh ; wl] x zh
                acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[sl & 1][1].l, ring[j % R].l, acc[m][1], 0, 0, 0);      // wh x zl
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        MIL_STAMP_MARK(4)
        // ---- epilogue from registers: channels 0-15 as 8 per lane, channels 16-19 as 4 per lane (lanes with hsel == 0) -------
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
            const int e_ty = wave * MTW + 2 * p + (gq & 1);
            const char* xrec = ldsX + (e_ty * 16 + r) * PIXX;      // own pixel's record in the hi plane (sign source of the mask)
            float v[8], u[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo = acc[2 * p][0][i], hi = acc[2 * p + 1][0][i];
                if (i == 0) mil_swap16<true>(lo, hi); else mil_swap16<false>(lo, hi);
                v[i] = lo; v[4 + i] = hi;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo = acc[2 * p][1][i], hi = acc[2 * p + 1][1][i];
                if (i == 0) mil_swap16<true>(lo, hi); else mil_swap16<false>(lo, hi);
                u[i] = lo + hi;                                    // rows 0-3 (wh products) + rows 4-7 (wl x zh) of the same pixel
            }
            if constexpr (ADD) {
                const f32x4_t t0 = __builtin_bit_cast(f32x4_t, radd[p][0]), t1 = __builtin_bit_cast(f32x4_t, radd[p][1]);
                const f32x4_t t2 = __builtin_bit_cast(f32x4_t, radd[p][2]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] += t0[i]; v[4 + i] += t1[i]; u[i] += t2[i]; }
            }
            if constexpr (MASK) {
                const bf16x8_t m0 = *reinterpret_cast<const bf16x8_t*>(xrec + hsel * 16);
                const bf16x4_t m1 = *reinterpret_cast<const bf16x4_t*>(xrec + 32);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] *= ((float)m0[i] > 0.f ? 1.f : a.slope);
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] *= ((float)m1[i] > 0.f ? 1.f : a.slope);
            }
            const unsigned o0 = goff[p];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[0], v[1], v[2], v[3]}), rs_dx, o0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[4], v[5], v[6], v[7]}), rs_dx, o0 == MIL_OOB ? MIL_OOB : o0 + 16, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{u[0], u[1], u[2], u[3]}), rs_dx,
                                                   (o0 == MIL_OOB || !last_ok) ? MIL_OOB : o0 + 64, 0, 0);
            if (GPX == 96)                   // padded layout: the four padding channels of the pixel are zeros
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{0u, 0u, 0u, 0u}, rs_dx, (o0 == MIL_OOB || !last_ok) ? MIL_OOB : o0 + 80, 0, 0);
        }
        MIL_STAMP_MARK(5)
        if (more) { set_goff(o_next); fetch_add(); }      // next tile's addend: its registers are free now
        MIL_STAMP_MARK(6)

        // ---- weight gradient, one k-step (32 pixels = two tile rows) ahead: dW' += zl*xh + zh*xl + zh*xh ---------------------
        {
            bf16x8_t xc[2][NTX], zc[2][MW], xn[2][NTX], zn[2][MW];
            auto loadw = [&](int kk, bf16x8_t (&xf)[2][NTX], bf16x8_t (&zf)[2][MW]) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    if (pl == 0) xf[0][1] = mil_tr_pair(ldsX + xb1 + kk * 32 * PIXX, ldsX + xb1 + kk * 32 * PIXX + 4 * PIXX);      // the folded fragment
                    xf[pl][0] = mil_tr_pair(ldsX + pl * X_PLANE + xb + kk * 32 * PIXX, ldsX + pl * X_PLANE + xb + kk * 32 * PIXX + 4 * PIXX);
#pragma unroll
                    for (int i = 0; i < MW; ++i)
                        zf[pl][i] = mil_tr_pair(ldsA + pl * A_PLANE + zw[i][0] + kk * 2 * ROWB, ldsA + pl * A_PLANE + zw[i][1] + kk * 2 * ROWB);
                }
            };
            loadw(0, xc, zc);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                if (kk + 1 < 8) loadw(kk + 1, xn, zn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt) {
                        wacc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zc[1][i], xc[0][nt], wacc[i][nt], 0, 0, 0);
                        if (nt == 0) wacc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zc[0][i], xc[1][nt], wacc[i][nt], 0, 0, 0);
                        wacc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zc[0][i], xc[0][nt], wacc[i][nt], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt) if (pl == 0 || nt == 0) xc[pl][nt] = xn[pl][nt];
#pragma unroll
                    for (int i = 0; i < MW; ++i) zc[pl][i] = zn[pl][i];
                }
            }
        }
        MIL_STAMP_MARK(7)
    }
    MIL_STAMP_STORE(a.stamp, NW)
    // ---- partial sums -> slab: rows tap'*20 + co, cols ci (col 23 of the centre-tap rows = bias sums) --------------------
    constexpr int SLAB_COLS = NTX * 16;
    constexpr size_t SLAB_ELEMS = (size_t)(14 + 1) * 16 * SLAB_COLS;      // the launcher's slab pitch (generic kernel's row count)
    float* slab = a.slab + (size_t)blockIdx.x * SLAB_ELEMS;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int mt = wave + NW * i;
#pragma unroll
        for (int nt = 0; nt < NTX; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = wacc[i][nt][e];
                if (nt == 1) {                                   // folded column tile: columns 8-11 (the x_lo products) onto columns 0-3
                    const float up = __shfl_down(v, 8, 16);
                    v = r < 4 ? v + up : (r < 8 ? v : 0.f);
                }
                slab[(size_t)(mt * 16 + gq * 4 + e) * SLAB_COLS + nt * 16 + r] = v;
            }
    }
}
