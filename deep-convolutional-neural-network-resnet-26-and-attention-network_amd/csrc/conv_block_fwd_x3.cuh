// Forward of a whole identity-shortcut residual block of the 20-channel stage in one pass, SPLIT PRECISION (MIL_DT_F32S: fp32
// tensors, bf16x3 products) — nnBlocks.py:175-189:  o1 = lrelu(conv3x3(x) + b1),  y = lrelu(conv3x3(o1) + b2 + x).
// Included by conv_block_fwd.hip.  The two persistent conv launches it replaces move 5 fp32 tensor passes; this one moves 3
// (x in; o1 and y out: what the backward needs).
//
// Shape of the kernel, and why it differs from conv_block_fwd_kernel (bf16):
//   * operands live in LDS as hi = bf16(v) / lo = bf16(v - hi) PLANES (48-byte records in the K20 layout of geom.cuh:
//     [ch 0-15][ch 16-19][ch 16-19 of the next pixel]); a conv is 6 k-steps x 3 MFMAs per fragment pair.
//   * two [hi | lo] filters do not fit beside the tiles twice per CU as they are packed (2 x 24.6 KB).  Of a 20-channel filter's
//     second column tile (channels 16-31) only four rows exist: it is staged as 16 lane entries + one zero entry that the other
//     48 lanes read (a broadcast) — 15.7 KB per filter.  With 16 x 8 output tiles (20 x 12 input halo, 18 x 10 mid tile) the
//     workgroup needs 72 KB: TWO 4-wave workgroups per CU, each wave owning 3 of the 12 mid row tiles and 2 of the 8 output
//     row tiles (no idle waves), and one workgroup's commit / barrier phases run under the other's MFMA loops.
//   * the mid activation goes from the accumulators to the LDS planes (split there) AND, for the tile's own pixels, straight to
//     the o1 tensor as exact fp32 (a lane holds four consecutive channels of a pixel = one 16-byte store); the residual is
//     the centre of the input halo, hi + lo (x to 2^-18 relative; round 5: the exact fp32 re-read it replaced missed L2).
//   * x / o1 / y are addressed at a RUN-TIME pixel stride a.apx: 96 bytes (24 padded channels) or 80 (dense 20 channels).
#pragma once
#include "stamp.cuh"
#ifndef MIL_BFX3_STORE_AUX
#define MIL_BFX3_STORE_AUX 0            // cache policy of the o1 / y stores (gfx940 encoding: 1 = sc0, 2 = nt, 16 = sc1).  Measured (round 5, 2048 tiles): plain 0.84 ms per launch, nt 1.12, sc1 1.75 — write-through stores do not buy the x halo more L2, they stall the epilogue
#endif

struct BlockFwdX3Args {
    const float* x;         // [n,H,W,apx/4]
    const char* w1;         // MIL_PACK_FWD fragments of MIL_DT_F32S: [7 + 6 k-steps][2][64][32 B]; the K20 section is used
    const char* w2;
    const float* b1;        // [32]
    const float* b2;
    float* o1;
    float* y;
    ConvGeom g;             // 16 x 8 tiles of one image
    float slope;
    int apx;                // bytes per pixel of x / o1 / y
    unsigned long long* stamp;      // MIL_STAMP diagnostic build only (else null)
};

__global__ __launch_bounds__(256, 2) void conv_block_fwd_x3_kernel(BlockFwdX3Args a, int ntiles, unsigned bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int NT = 2, NW = 4, NTHR = 256, KSTEPS = MIL_K20_STEPS, KSTEPS_STD = 7;
    constexpr int PIXB = 48, XW = 20, XH = 12, OW = 18, OH = 10, ROWX = XW * PIXB, ROWO = OW * PIXB;
    constexpr int SPARE = 16;                                        // in front of every plane: pixel 0's "next pixel" back-copy
    constexpr int X_PLANE = SPARE + XW * XH * PIXB;                  // 11536
    constexpr int O_PLANE = SPARE + OW * OH * PIXB;                  // 8656
    constexpr int WSTEP = 2048 + 512 + 64;                           // per k-step: column tile 0 (64 lanes x 32 B), 16 entries of tile 1, zeros
    constexpr int W_BYTES = KSTEPS * WSTEP;                          // 15744
    constexpr int OFF_X = 0, OFF_O = 2 * X_PLANE, OFF_W1 = OFF_O + 2 * O_PLANE, OFF_W2 = OFF_W1 + W_BYTES, OFF_DUMP = OFF_W2 + W_BYTES;
    constexpr int MT1 = 3, MT2 = 2;                                  // row tiles per wave: 12 mid (180 px), 8 output (128 px)
    constexpr int NPX = (XW * XH * 5 + NTHR - 1) / NTHR;             // 5 halo pieces (16 B = four fp32 channels) per thread
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    char* ldsX = smem + OFF_X + SPARE;                               // hi plane records; lo plane X_PLANE behind
    char* ldsO = smem + OFF_O + SPARE;
    const int H = g.H, W = g.W, APX = a.apx;

    // ---- filters: K20 section of each packed buffer -> compact LDS form ---------------------------------------------------
    {
        const __amdgpu_buffer_rsrc_t rw1 = mil_rsrc(a.w1, (KSTEPS_STD + KSTEPS) * NT * 64 * 32);
        const __amdgpu_buffer_rsrc_t rw2 = mil_rsrc(a.w2, (KSTEPS_STD + KSTEPS) * NT * 64 * 32);
        constexpr int K20_OFF = KSTEPS_STD * NT * 64 * 32;
        // per filter and k-step: 128 pieces of column tile 0, 32 pieces of tile 1 (lanes with r < 4), 4 zero pieces
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
                const int e = (p - 128) >> 1, half = (p - 128) & 1;          // entry e = gq*4 + r (r < 4) <- lane gq*16 + r
                doff = 2048 + e * 32 + half * 16;
                v = __builtin_amdgcn_raw_buffer_load_b128(f ? rw2 : rw1, (unsigned)(K20_OFF + (sl * NT + 1) * 2048 + ((e >> 2) * 16 + (e & 3)) * 32 + half * 16), 0, 0);
            } else {
                doff = 2560 + (p - 160) * 16;
            }
            *reinterpret_cast<u32x4_t*>(dst + doff) = v;
        }
        // "next pixel" slots no commit writes (zero weights read them: must be finite): last record of every plane
        if (tid < 2) *reinterpret_cast<u32x2_t*>(ldsX + tid * X_PLANE + (XW * XH - 1) * PIXB + 40) = u32x2_t{0u, 0u};
        else if (tid < 4) *reinterpret_cast<u32x2_t*>(ldsO + (tid - 2) * O_PLANE + (OW * OH - 1) * PIXB + 40) = u32x2_t{0u, 0u};
    }
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, bytes);
    const __amdgpu_buffer_rsrc_t rs_o = mil_rsrc(a.o1, bytes);
    const __amdgpu_buffer_rsrc_t rs_y = mil_rsrc(a.y, bytes);

    // ---- tile-invariant tables -------------------------------------------------------------------------------------------
    // input halo pieces: flat id = tid + 256*i -> (halo pixel id/5, piece id%5): consecutive lanes read consecutive pieces
    int h_pos[NPX], h_lds[NPX], h_rel[NPX];
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int idx = tid + NTHR * i;
        const int px = idx / 5, j = idx - px * 5;
        const int hy = px / XW, hx = px - hy * XW;
        const bool used = px < XW * XH;
        h_pos[i] = used ? (j << 20) | (hy << 10) | hx : (int)0x80000000u;
        h_lds[i] = used ? px * PIXB + j * 8 : OFF_DUMP - (OFF_X + SPARE);
        h_rel[i] = used ? ((hy + 1) * W + hx + 1) * APX + j * 16 : (int)MIL_OOB;      // relative to one row and one column before the halo origin
    }
    // fragment offsets: K20 k-group q = 4*sl + gq relative to the record under the filter's top-left tap
    int kx[KSTEPS], ko[KSTEPS];                                      // in the input halo (row pitch 20) / in the mid tile (18)
#pragma unroll
    for (int sl = 0; sl < KSTEPS; ++sl) {
        int qx = 0, qo = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (gq == k) { qx = mil_k20_off(4 * sl + k, ROWX, PIXB); qo = mil_k20_off(4 * sl + k, ROWO, PIXB); } }
        kx[sl] = qx; ko[sl] = qo;
    }
    int pb1[MT1], sdst[MT1], mpos[MT1];                              // conv1: mid pixel tp = (wave + 4*i)*16 + r
#pragma unroll
    for (int i = 0; i < MT1; ++i) {
        const int tp = (wave + NW * i) * 16 + r;
        const bool ok = tp < OW * OH;
        const int py = tp / OW, px = tp - py * OW;
        pb1[i] = ok ? (py * XW + px) * PIXB : 0;
        sdst[i] = ok ? tp * PIXB : -1;
        mpos[i] = ok ? (py << 10) | px : (int)0x80000000u;
    }
    const int pb2 = (wave * MT2 * OW + r) * PIXB;                   // conv2: output pixel (2*wave [+m], r) reads mid (ty+ky, tx+kx)
    const int e_ty = wave * MT2 + (gq & 1), hsel = gq >> 1;          // epilogue pair: pixel (e_ty, r), channels 8*hsel.. (+16..19 for hsel 0)
    const bool last_ok = hsel == 0;
    // column tile 1 (channels 16-19 = rows 0-3): its "hi" fragment carries the lo weights of the same channels in rows 4-7, so
    // one MFMA against the pixels' hi plane gives wh*xh (rows 0-3) and wl*xh (rows 4-7); its second fragment is wh alone, for
    // the pixels' lo plane: five MFMAs per fragment pair instead of six, rows 4-7 added to rows 0-3 in the epilogues.
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
    // one conv as a flattened (k-step, row tile) pipeline: pixel fragments LA row tiles ahead, the next k-step's filter fragments
    // at the start of the current one
    auto conv = [&](auto& acc, auto mt_c, const char* ldsW, const char* plane0, int plane_step, auto paddr) {
        constexpr int MT = decltype(mt_c)::value;
        constexpr int TOT = KSTEPS * MT, LA = 2, R = LA + 1;
        Frag8<F32S> ring[R], wq[2][NT];
        auto pfrag = [&](int j) {
            const char* p = plane0 + paddr(j / MT, j % MT);
            Frag8<F32S> f;
            f.h = *reinterpret_cast<const bf16x8_t*>(p);
            f.l = *reinterpret_cast<const bf16x8_t*>(p + plane_step);
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

    TileWalker cur, nxt;
    const int bid = mil_xcd_block_id();
    cur.init(g, bid, gridDim.x);
    nxt = cur; nxt.advance();
    u32x4_t rx[NPX];
    auto fetch = [&](const TileOrigin& o) {
        const int iy0 = o.oy0 - 2, ix0 = o.ox0 - 2;
        const int base = ((o.img0 * H + iy0 - 1) * W + ix0 - 1) * APX;         // may be negative; valid lanes' sums are not
        if (iy0 >= 1 && ix0 >= 1 && iy0 + XH <= H && ix0 + XW <= W) {          // interior tile (wave-uniform): base in the scalar offset
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
    MIL_STAMP_DECL(9)
    for (int tile = bid; tile < ntiles; tile += G) {
        const TileOrigin o = cur.origin(g);
        MIL_STAMP_BEGIN()
        __syncthreads();                       // previous tile: conv2's reads of the mid tile and conv1's of the halo are done
        MIL_STAMP_MARK(0)
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            int p = h_pos[i];
            asm volatile("" : "+v"(p));
            u32x2_t hi, lo;
            split4(__builtin_bit_cast(f32x4_t, rx[i]), hi, lo);
            const int l0 = h_lds[i];
            const bool dup = p >= 0 && ((p >> 20) & 7) == 4;            // channels 16-19: also the previous pixel's "next pixel" slot
            const int l1 = dup ? l0 - 40 : OFF_DUMP - (OFF_X + SPARE);
            *reinterpret_cast<u32x2_t*>(ldsX + l0) = hi;
            *reinterpret_cast<u32x2_t*>(ldsX + X_PLANE + l0) = lo;
            *reinterpret_cast<u32x2_t*>(ldsX + l1) = hi;
            *reinterpret_cast<u32x2_t*>(ldsX + (dup ? l1 + X_PLANE : l1 + 8)) = lo;
        }
        MIL_STAMP_MARK(1)
        __syncthreads();                       // input halo visible
        MIL_STAMP_MARK(2)
        if (tile + G < ntiles) fetch(nxt.origin(g));
        cur = nxt; nxt.advance();
        MIL_STAMP_MARK(3)
        const int obase = ((o.img0 * H + o.oy0) * W + o.ox0) * APX;
        const int ylim = H - o.oy0, xlim = W - o.ox0;

        // ---- conv1 on the 18 x 10 mid tile -> LDS planes (+ the tile's own pixels -> o1, exact fp32) ------------------------
        {
            f32x4_t acc[MT1][NT];
#pragma unroll
            for (int i = 0; i < MT1; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[i][nt] = b1r[nt];
            conv(acc, std::integral_constant<int, MT1>{}, smem + OFF_W1, ldsX, X_PLANE, [&](int sl, int i) { return pb1[i] + kx[sl]; });
            MIL_STAMP_MARK(4)
            const int my0 = o.oy0 - 1, mx0 = o.ox0 - 1;
#pragma unroll
            for (int i = 0; i < MT1; ++i) {
                const int p = mpos[i];
                const int py = (p >> 10) & 1023, px = p & 1023;
                // mid pixels outside the image are conv2's zero padding
                const bool inside = (p >= 0) & ((unsigned)(my0 + py) < (unsigned)H) & ((unsigned)(mx0 + px) < (unsigned)W);
                const bool own = inside & (py >= 1) & (py <= 8) & (px >= 1) & (px <= 16);
                const unsigned ooff = own ? (unsigned)(obase + ((py - 1) * W + px - 1) * APX + gq * 16) : MIL_OOB;
#pragma unroll
                for (int e = 0; e < 4; ++e) {            // column tile 1: rows 4-7 (the lane group gq == 1) hold wl*xh of rows 0-3
                    float t0 = acc[i][1][e], t1 = t0;
                    if (e == 0) mil_swap16<true>(t0, t1); else mil_swap16<false>(t0, t1);      // t1 of lane group 0 = group 1's value
                    acc[i][1][e] += t1;                  // (meaningful in the lanes of group 0 only; the others are never stored)
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float t = acc[i][nt][e]; v[e] = inside ? fmaxf(t, t * a.slope) : 0.f; }
                    u32x2_t hi, lo;
                    split4(v, hi, lo);
                    const bool real = nt == 0 || gq == 0;                // channels 16-19 sit in lanes gq == 0 of column tile 1
                    const int d0 = (sdst[i] >= 0 && real) ? sdst[i] + nt * 32 + gq * 8 : OFF_DUMP - (OFF_O + SPARE);
                    *reinterpret_cast<u32x2_t*>(ldsO + d0) = hi;
                    *reinterpret_cast<u32x2_t*>(ldsO + (d0 == OFF_DUMP - (OFF_O + SPARE) ? d0 + 8 : d0 + O_PLANE)) = lo;
                    if (nt == 1) {                                       // + the previous pixel's "next pixel" slot
                        const int d1 = (sdst[i] >= 0 && gq == 0) ? sdst[i] - 8 : OFF_DUMP - (OFF_O + SPARE);
                        *reinterpret_cast<u32x2_t*>(ldsO + d1) = hi;
                        *reinterpret_cast<u32x2_t*>(ldsO + (d1 == OFF_DUMP - (OFF_O + SPARE) ? d1 + 8 : d1 + O_PLANE)) = lo;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs_o,
                                                           (ooff == MIL_OOB || !real) ? MIL_OOB : ooff + nt * 64, 0, MIL_BFX3_STORE_AUX);
                    if (nt == 1 && APX == 96)                            // padded layout: the pixel's four padding channels
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{0u, 0u, 0u, 0u}, rs_o, (ooff == MIL_OOB || gq != 0) ? MIL_OOB : ooff + 80, 0, 0);
                }
            }
        }
        // residual: the tile's own x pixels = the centre of the input halo in LDS, as hi + lo (x to 2^-18 relative: the precision
        // the products see) in the epilogue layout — the same expression as the row-walk kernel's, so the two forms agree bit for
        // bit and a result does not depend on which of them a launch size selects.  (Until round 5 the pixels were re-read from the
        // tensor as exact fp32; at this footprint that re-read missed L2.)
        const bool e_ok = e_ty < ylim && r < xlim;
        const unsigned eoff = e_ok ? (unsigned)(obase + (e_ty * W + r) * APX + hsel * 32) : MIL_OOB;
        const char* xrec = ldsX + ((e_ty + 2) * XW + r + 2) * PIXB;
        MIL_STAMP_MARK(5)
        __syncthreads();                       // mid tile visible
        MIL_STAMP_MARK(6)

        // ---- conv2 + residual + LeakyReLU -> y --------------------------------------------------------------------------------
        {
            f32x4_t acc[MT2][NT];
#pragma unroll
            for (int m = 0; m < MT2; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = b2r[nt];
            conv(acc, std::integral_constant<int, MT2>{}, smem + OFF_W2, ldsO, O_PLANE, [&](int sl, int m) { return pb2 + m * ROWO + ko[sl]; });
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
            const bf16x8_t xh0 = *reinterpret_cast<const bf16x8_t*>(xrec + hsel * 16), xl0 = *reinterpret_cast<const bf16x8_t*>(xrec + X_PLANE + hsel * 16);
            const bf16x4_t xh1 = *reinterpret_cast<const bf16x4_t*>(xrec + 32), xl1 = *reinterpret_cast<const bf16x4_t*>(xrec + X_PLANE + 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = v[i] + ((float)xh0[i] + (float)xl0[i]); v[i] = fmaxf(s, s * a.slope);
                s = v[4 + i] + ((float)xh0[4 + i] + (float)xl0[4 + i]); v[4 + i] = fmaxf(s, s * a.slope);
                s = u[i] + ((float)xh1[i] + (float)xl1[i]); u[i] = fmaxf(s, s * a.slope);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[0], v[1], v[2], v[3]}), rs_y, eoff, 0, MIL_BFX3_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[4], v[5], v[6], v[7]}), rs_y, eoff == MIL_OOB ? MIL_OOB : eoff + 16, 0, MIL_BFX3_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{u[0], u[1], u[2], u[3]}), rs_y,
                                                   (eoff == MIL_OOB || !last_ok) ? MIL_OOB : eoff + 64, 0, MIL_BFX3_STORE_AUX);
            if (APX == 96)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{0u, 0u, 0u, 0u}, rs_y, (eoff == MIL_OOB || !last_ok) ? MIL_OOB : eoff + 80, 0, 0);
        }
        MIL_STAMP_MARK(8)
    }
    MIL_STAMP_STORE(a.stamp, NW)
}

#include "conv_block_strip_x3.cuh"

static int launch_block_strip_x3(BlockFwdX3Args a, hipStream_t st) {
    const ConvGeom& g = a.g;
    constexpr int lds = MIL_STRIP_X3_LDS;
    auto kern = conv_block_strip_x3_kernel;
    static std::atomic<unsigned long long> attr_set{0};
    if (mil_device_needs(attr_set)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MIL_ERR_LAUNCH;
        mil_device_done(attr_set);
    }
    const int per_cu = mil_resident_per_cu(kern, lds, 2, 256);
    const size_t img = (size_t)g.H * g.W * a.apx;
    const int chunk = mil_imgs_under_2g(img);
    for (int i0 = 0; i0 < g.n_img; i0 += chunk) {
        const int n = (g.n_img - i0 < chunk) ? g.n_img - i0 : chunk;
        BlockFwdX3Args c = a;
        c.x = a.x + (size_t)i0 * (img / 4); c.o1 = a.o1 + (size_t)i0 * (img / 4); c.y = a.y + (size_t)i0 * (img / 4);
        int grid = mil_num_cus() * per_cu;
        if (grid > n) grid = n;
#ifdef MIL_STAMP
        static MilStampBuf sb;
        c.stamp = sb.get((size_t)grid * 4 * 11);
#endif
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, c, n, (unsigned)(img * n));
        MIL_CHECK_LAUNCH();
#ifdef MIL_STAMP
        static const char* const ph[9] = {"barrier-top", "commit", "barrier-x", "fetch-issue", "conv1", "conv1-epilogue", "barrier-mid", "conv2", "conv2-epilogue"};
        sb.report("conv_block_strip_x3_kernel", grid, 4, 9, ph, st);
#endif
    }
    return MIL_OK;
}

static int launch_block_fwd_x3(BlockFwdX3Args a, hipStream_t st) {
    ConvGeom& g = a.g;
    if (mil_block_strip_wanted(g.W == 64, g.n_img, (long)((g.H + 7) >> 3) * 4, ((g.H + 1) >> 1) + 1, 6, 5, mil_num_cus() * 2)) return launch_block_strip_x3(a, st);
    g.tw_log2 = 4; g.th_log2 = 3; g.ti_log2 = 0;
    g.tiles_x = (g.W + 15) >> 4; g.tiles_y = (g.H + 7) >> 3; g.n_groups = g.n_img;
    g.hh = 12; g.hw = 20;
    constexpr int X_PLANE = 16 + 20 * 12 * 48, O_PLANE = 16 + 18 * 10 * 48, W_BYTES = MIL_K20_STEPS * (2048 + 512 + 64);
    constexpr int lds = 2 * X_PLANE + 2 * O_PLANE + 2 * W_BYTES + 64;
    auto kern = conv_block_fwd_x3_kernel;
    static std::atomic<unsigned long long> attr_set{0};
    if (mil_device_needs(attr_set)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return MIL_ERR_LAUNCH;
        mil_device_done(attr_set);
    }
    const int per_cu = mil_resident_per_cu(kern, lds, 2, 256);
    const size_t img = (size_t)g.H * g.W * a.apx;
    int chunk = mil_imgs_under_2g(img);
    const int n_total = g.n_img;
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        const int n = (n_total - i0 < chunk) ? n_total - i0 : chunk;
        BlockFwdX3Args c = a;
        c.g.n_img = n; c.g.n_groups = n;
        c.x = a.x + (size_t)i0 * (img / 4); c.o1 = a.o1 + (size_t)i0 * (img / 4); c.y = a.y + (size_t)i0 * (img / 4);
        const int ntiles = n * g.tiles_y * g.tiles_x;
        int grid = mil_num_cus() * per_cu;
        if (grid > ntiles) grid = ntiles;
#ifdef MIL_STAMP
        static MilStampBuf sb;
        c.stamp = sb.get((size_t)grid * 4 * 11);
#endif
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, c, ntiles, (unsigned)(img * n));
        MIL_CHECK_LAUNCH();
#ifdef MIL_STAMP
        static const char* const ph[9] = {"barrier-top", "commit", "barrier-x", "fetch-issue", "conv1", "conv1-epilogue", "barrier-mid", "conv2", "conv2-epilogue"};
        sb.report("conv_block_fwd_x3_kernel", grid, 4, 9, ph, st);
#endif
    }
    return MIL_OK;
}
