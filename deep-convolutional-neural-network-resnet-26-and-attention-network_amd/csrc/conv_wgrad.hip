// Weight- and bias-gradient of the tile-encoder convolutions on CDNA4 MFMA (autograd of
// nnBlocks.py:169-171, gbm/model.py:24, gbm/model.py:38-40 in the reference).
//
//   dW[(tap,ci)][co] = sum over output pixels p of  x[p (+) tap][ci] * dz[p][co]
//   GEMM view:  M = taps x input channels (row groups of 8 channels, two per 16-row MFMA tile),
//               N = output channels,  K = output pixels.
//   Both operands are K(pixel)-major in NHWC memory but the MFMA wants 8 consecutive k per lane, so
//   the bf16 path reads both LDS tiles with ds_read_b64_tr_b16 (hardware transpose); the exact-f32
//   path uses v_mfma_f32_16x16x4_f32, whose one-element-per-lane operands need no transpose.
//   Each workgroup walks a strided set of spatial tiles, keeps its partial dW in registers and writes
//   ONE fp32 slab; a second kernel sums the slabs in fixed order (deterministic, no float atomics).
//   db comes from an all-ones A fragment on wave 0 (column sums of dz on the same MFMA stream).
#include "pf_common.cuh"
#include "reduce.cuh"
#include "stamp.cuh"

template <typename T>
struct WgradArgs {
    const typename T::elem* x;
    const typename T::elem* dz;
    float* slab;            // [gridDim.x][(MT+1)*16][NT*16]
    ConvGeom g;
    int ntiles;
    int tile_px;
    int lds_z_off;
    unsigned x_bytes, z_bytes;   // buffer-descriptor sizes (prefetch-pipelined path)
    const typename T::elem* dz2; // PROJ: gradient of the 1x1/s2 projection's output (same shape as dz)
    int lds_z2_off;
    unsigned long long* stamp;   // MIL_STAMP diagnostic build only
};

__device__ __forceinline__ bf16x8_t tr_pair(const char* p0, const char* p1) {
    typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

// NW = waves per workgroup: 4, or 8 for the persistent bf16 form of the 64/80-channel layers, whose accumulators
// (up to 9 row tiles x 5 column tiles per wave) otherwise leave ONE wave per SIMD: eight waves on the same LDS tiles
// halve every per-wave quantity and give each SIMD a second wave to overlap with.
// PROJ (stage-entry blocks, 3x3/s2 conv + 1x1/s2 projection on the same input): the projection's weight gradient
// dWp[ci][co] = sum_q x[2q][ci] * dz2[q][co] has the 3x3 conv's CENTRE-TAP rows as its A operand, so the row tiles that
// hold those rows get a second accumulator set fed by a second dz tile — the block input is read once for both filters
// instead of once per weight-gradient launch.  Its rows are appended to the slab behind the bias tile.
// which instantiations take the explicit one-step-ahead operand prefetch (second operand register set): the persistent
// bf16 forms below 64 input channels; the 64/80-channel forms sit at 232-256 VGPRs already
#ifndef MIL_WGRAD_X3_WIDE_PF
#define MIL_WGRAD_X3_WIDE_PF 1      // split precision, 64 channels: register prefetch of the next 128-pixel tile (80 channels: 26 VGPRs spilled)
#endif
#ifndef MIL_WGRAD_X3_HALF40
#define MIL_WGRAD_X3_HALF40 1       // split precision, 40 -> 40 channels: 128-pixel tiles on TWO 4-wave workgroups per CU
#endif
// Split precision, 3x3 stride-1 40 -> 40 channels: 128-pixel tiles (58 KB of LDS) on 4-wave workgroups, two per CU, instead of
// 256-pixel tiles on one 8-wave workgroup (123 KB): the phase stamps had 45 % of a tile outside the MFMA loop (request burst,
// commit, barriers) with every wave of the CU in the same phase; two independent workgroups run those phases under each
// other's MFMAs.
__host__ __device__ constexpr bool mil_wgrad_x3_half(bool split, int ks, int cinp, int nt, int msplit, bool proj) {
    return MIL_WGRAD_X3_HALF40 && split && ks == 3 && cinp == 40 && nt == 3 && msplit == 1 && !proj;
}
__host__ __device__ constexpr int mil_wgrad_halo_max(int cinp, bool proj, bool split = false, bool half = false) {
    if ((split && cinp >= 64) || half) return 200;  // 128-pixel tiles: 18x10 pixels, or two 10x10 images
#ifdef MIL_WGRAD_PAIR24_64PX
    return 400;
#else
    return (proj && cinp <= 24) ? 576 : 400;
#endif
}
__host__ __device__ constexpr int mil_wgrad_tile_max(int cinp, bool split, bool half = false) { return ((split && cinp >= 64) || half) ? 128 : 256; }
#ifdef MIL_WGRAD_NO_PIPE
#define MIL_WGRAD_PIPE(BF, PF, CINP, NW, PROJ) false
#else
#define MIL_WGRAD_PIPE(BF, PF, CINP, NW, PROJ) ((BF) && (PF) && (CINP) < 64)
#endif
template <typename T, int KS, int CINP, int NT, int MSPLIT, bool PF, int NW = 4, bool PROJ = false>
#ifndef MIL_WGRAD_X3_WAVES
#define MIL_WGRAD_X3_WAVES 2        // split precision, 8-wave workgroups: waves per SIMD the register budget is held to (4 = 128 VGPRs spilled 22)
#endif
__global__ __launch_bounds__(64 * NW, NW == 8 ? ((T::SPLIT && CINP <= 24 && NT <= 2) ? MIL_WGRAD_X3_WAVES : 2) : ((PF && CINP > 40) ? 1 : ((PF && T::SPLIT) ? 2 : 0))) void wgrad_kernel(WgradArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int ESZ = T::ESZ;
    constexpr int PIXB = mil_pix_pitch(CINP, ESZ);
    constexpr int COUTP = mil_nt_to_cp(NT);
    constexpr int PIXZ = mil_pix_pitch(COUTP, ESZ);
    constexpr int CG = CINP / 8;
    constexpr int RG = KS * KS * CG;            // row groups of 8 input channels
    constexpr int MT = (RG + 1) / 2;            // 16-row MFMA tiles over (tap, ci)
    constexpr int MT_S = (MT + MSPLIT - 1) / MSPLIT;
    constexpr int NTHR = 64 * NW;
    constexpr int MW = (MT_S + NW - 1) / NW;    // tiles per wave
    constexpr int PM0 = 2 * CG, PM1 = (5 * CG - 1) / 2, PMN = PROJ ? PM1 - PM0 + 1 : 0;    // row tiles holding centre-tap rows
    constexpr bool WPIPE = MIL_WGRAD_PIPE(T::DT == MIL_DT_BF16, PF, CINP, NW, PROJ);
    // Split precision, 24 or 40 output channels: the last column tile holds eight real columns, columns 8-15 idle.  The lanes that
    // read the pieces of columns 8-15 (p >= 2) read the LO plane's pieces of the tile's eight channels instead, so x_lo * [dz_hi |
    // dz_lo] and x_hi * [dz_hi | dz_lo] are two MFMAs for what took three (columns 8-15 also collect x_lo*dz_lo, the 2^-18 term the
    // three-product form drops); they are added onto columns 0-7 when the slab is written.
#ifndef MIL_WGRAD_ZFOLD
#define MIL_WGRAD_ZFOLD 1
#endif
    constexpr bool ZFOLD = MIL_WGRAD_ZFOLD && T::SPLIT && (COUTP % 16) == 8;
    constexpr int ZFOLD_D = COUTP * 2 - 16;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    // wave id through readfirstlane: provably wave-uniform, so branches on it are scalar branches (an MFMA or a
    // ds_read_b64_tr_b16 inside an EXEC-masked region would still execute / need all lanes)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.y;
    const bool bias_wave = (wave == 0 && split == 0);
    char* ldsX = smem;
    char* ldsZ = smem + a.lds_z_off;

    // per-lane byte offset of this lane's (tap, channel) rows inside a halo pixel, per owned m-tile
    int toff[MW];
    bool mvalid[MW];
    int proj_i = -1, proj_mt = 0;               // the owned row tile (at most one per wave) that also feeds the projection
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int ml = wave + NW * i;
        const int mt = split * MT_S + ml;
        mvalid[i] = (ml < MT_S) && (mt < MT);
        if (PROJ && mvalid[i] && mt >= PM0 && mt <= PM1) { proj_i = i; proj_mt = mt; }
        int rg, sub;
        if constexpr (T::TR16) { const int p = lane & 3; rg = 2 * mt + (p >> 1); sub = (p & 1) * 8; }
        else { const int row = lane & 15; rg = 2 * mt + (row >> 3); sub = (row & 7) * 4; }
        if (rg >= RG) rg = 0;                    // rows past the filter: finite duplicates, never read back
        const int tap = rg / CG, cg = rg - tap * CG;
        const int ky = tap / KS, kx = tap - ky * KS;
        toff[i] = (ky * g.hw + kx) * PIXB + cg * T::CGB + sub;
    }

    f32x4_t acc[MW][NT];
    f32x4_t accb[NT], accp[PROJ ? NT : 1];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        accb[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if constexpr (PROJ) accp[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MW; ++i) acc[i][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // PF: software pipeline — the next tile's global loads are issued (into registers) before this tile's
    // MFMA loop and written to LDS after the loop's barrier, so HBM latency hides under compute.  Addressing
    // through buffer descriptors + tile-invariant tables (pf_common.cuh) keeps the per-tile VALU work small.
    // halo pixels the register prefetch is sized for: 400, or 576 for the paired 24-channel stage entry (128-pixel tiles at
    // stride 2 stage 17x33 = 561 pixels; 64-pixel tiles left two k-steps of MFMAs per pair of barriers: 223 us for 148 us of traffic)
    constexpr bool HALF = NW == 4 && mil_wgrad_x3_half(T::SPLIT, KS, CINP, NT, MSPLIT, PROJ);
    constexpr int NPX = PF ? (mil_wgrad_halo_max(CINP, PROJ, T::SPLIT, HALF) * (CINP * ESZ / 16) + NTHR - 1) / NTHR : 1;
    constexpr int NPZ = PF ? (mil_wgrad_tile_max(CINP, T::SPLIT, HALF) * (COUTP * ESZ / 16) + NTHR - 1) / NTHR : 1;
    u32x4_t rx[NPX], rz[NPZ], rz2[PROJ ? NPZ : 1];
    char* ldsZ2 = smem + a.lds_z2_off;
    __amdgpu_buffer_rsrc_t rs_z2;
    HaloTables<NPX> ht;
    OtileTables<NPZ> zt;
    TileWalker cur, nxt;
    __amdgpu_buffer_rsrc_t rs_x, rs_z;
    const int bid = mil_xcd_block_id();
    if constexpr (PF) {
        rs_x = mil_rsrc(a.x, a.x_bytes);
        rs_z = mil_rsrc(a.dz, a.z_bytes);
        if constexpr (PROJ) rs_z2 = mil_rsrc(a.dz2, a.z_bytes);
        mil_build_halo_tables<CINP, NPX, NTHR, T>(ht, g, tid);
        mil_build_otile_tables<COUTP, NPZ, NTHR, T>(zt, g, tid, a.tile_px);
        cur.init(g, bid, gridDim.x);
        nxt = cur; nxt.advance();
        if (bid < a.ntiles) {
            mil_fetch_halo<CINP, NPX, T>(rx, rs_x, ht, g, cur.origin(g));
            mil_fetch_otile<COUTP, NPZ, T>(rz, rs_z, zt, g, cur.origin(g));
            if constexpr (PROJ) mil_fetch_otile<COUTP, NPZ, T>(rz2, rs_z2, zt, g, cur.origin(g));
        }
    }
    MIL_STAMP_DECL(5)
    for (int tile = bid; tile < a.ntiles; tile += gridDim.x) {
        MIL_STAMP_BEGIN()
        __syncthreads();
        MIL_STAMP_MARK(0)
        if constexpr (PF) {
            mil_commit_halo<NPX, T, CINP>(rx, ldsX, ht);
            mil_commit_otile<NPZ, T, COUTP>(rz, ldsZ, zt);
            if constexpr (PROJ) mil_commit_otile<NPZ, T, COUTP>(rz2, ldsZ2, zt);
            MIL_STAMP_MARK(1)
            if (tile + (int)gridDim.x < a.ntiles) {
                mil_fetch_halo<CINP, NPX, T>(rx, rs_x, ht, g, nxt.origin(g));
                mil_fetch_otile<COUTP, NPZ, T>(rz, rs_z, zt, g, nxt.origin(g));
                if constexpr (PROJ) mil_fetch_otile<COUTP, NPZ, T>(rz2, rs_z2, zt, g, nxt.origin(g));
            }
            cur = nxt; nxt.advance();
        } else {
            const TileOrigin o = mil_tile_origin(g, tile);
            mil_load_halo<T, CINP>(ldsX, a.x, g, o, tid, NTHR);
            mil_load_otile<T, COUTP>(ldsZ, a.dz, g, o, tid, NTHR, a.tile_px);
            MIL_STAMP_MARK(1)
        }
        MIL_STAMP_MARK(2)
        __syncthreads();
        MIL_STAMP_MARK(3)
        if constexpr (T::TR16) {
            const int q4 = (lane & 15) >> 2, p = lane & 3, gq = lane >> 4;
            bf16x8_t ones;
#pragma unroll
            for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
            const int wpl0 = mil_pix_base<PIXB>(g, 8 * gq + q4, g.stride), wpl1 = mil_pix_base<PIXB>(g, 8 * gq + q4 + 4, g.stride);
            if constexpr (WPIPE) {
                // One 32-pixel k-step ahead, two k-steps per loop trip on ping-pong operand sets (tile_px is a multiple of
                // 64): the transposed reads of step k+1 are issued before the MFMAs of step k, and the wave's row tiles are
                // processed without validity branches (a row tile that does not exist reads row group 0 into accumulators
                // that are never stored).  The compiler-ordered loop below waits for every fragment right in front of its
                // MFMAs; at the one or two waves per SIMD these kernels run at, that made them LDS-latency loops.
                bf16x8_t bA[NT], b2A[PROJ ? NT : 1], aA[MW], bB[NT], b2B[PROJ ? NT : 1], aB[MW];
                auto load = [&](int k32, bf16x8_t (&bf)[NT], bf16x8_t (&bf2)[PROJ ? NT : 1], bf16x8_t (&af)[MW]) {
                    const int kb = mil_pix_base<PIXB>(g, k32, g.stride);
                    const char* z0 = ldsZ + (k32 + 8 * gq + q4) * PIXZ + p * 8;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bf[nt] = tr_pair(z0 + nt * 32, z0 + 4 * PIXZ + nt * 32);
                    if constexpr (PROJ) {
                        const char* y0 = ldsZ2 + (k32 + 8 * gq + q4) * PIXZ + p * 8;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bf2[nt] = tr_pair(y0 + nt * 32, y0 + 4 * PIXZ + nt * 32);
                    }
#pragma unroll
                    for (int i = 0; i < MW; ++i) af[i] = tr_pair(ldsX + kb + wpl0 + toff[i], ldsX + kb + wpl1 + toff[i]);
                };
                auto mfma = [&](const bf16x8_t (&bf)[NT], const bf16x8_t (&bf2)[PROJ ? NT : 1], const bf16x8_t (&af)[MW]) {
#pragma unroll
                    for (int i = 0; i < MW; ++i) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[nt], acc[i][nt], 0, 0, 0);
                        if constexpr (PROJ) {
                            if (i == proj_i) {          // wave-uniform
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
                                    accp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf2[nt], accp[nt], 0, 0, 0);
                            }
                        }
                    }
                    if (bias_wave) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            accb[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bf[nt], accb[nt], 0, 0, 0);
                    }
                };
                load(0, bA, b2A, aA);
                for (int k32 = 0; k32 < a.tile_px; k32 += 64) {
                    load(k32 + 32, bB, b2B, aB);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma(bA, b2A, aA);
                    __builtin_amdgcn_sched_barrier(0);
                    if (k32 + 64 < a.tile_px) load(k32 + 64, bA, b2A, aA);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma(bB, b2B, aB);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            for (int k32 = 0; k32 < a.tile_px; k32 += 32) {
                // halo offset of pixel k32 + lane part: additive (disjoint bit fields), k32 part is wave-uniform
                const int kb = mil_pix_base<PIXB>(g, k32, g.stride);
                const int pb0 = kb + wpl0, pb1 = kb + wpl1;
                const char* z0 = ldsZ + (k32 + 8 * gq + q4) * PIXZ + p * 8;
                const char* z1 = z0 + 4 * PIXZ;
                bf16x8_t bf[NT], bf2[PROJ ? NT : 1];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[nt] = tr_pair(z0 + nt * 32, z1 + nt * 32);
                if constexpr (T::SPLIT) {        // MIL_DT_F32S: dW += x_lo*dz_hi + x_hi*dz_lo + x_hi*dz_hi from the hi/lo planes of both tiles
                    bf16x8_t bl[NT], bl2[PROJ ? NT : 1];
                    const int zf = (ZFOLD && p >= 2) ? ZFOLD_D : 0;      // the folded fragment of the last column tile (see ZFOLD)
                    if constexpr (ZFOLD) bf[NT - 1] = tr_pair(z0 + (NT - 1) * 32 + zf, z1 + (NT - 1) * 32 + zf);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        if (!(ZFOLD && nt == NT - 1)) bl[nt] = tr_pair(z0 + COUTP * 2 + nt * 32, z1 + COUTP * 2 + nt * 32);
                    if constexpr (PROJ) {
                        if (proj_i >= 0) {       // wave-uniform: hi and lo planes of the projection's dz tile
                            const char* y0 = ldsZ2 + (k32 + 8 * gq + q4) * PIXZ + p * 8;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                const int yf = (ZFOLD && nt == NT - 1) ? zf : 0;
                                bf2[nt] = tr_pair(y0 + nt * 32 + yf, y0 + 4 * PIXZ + nt * 32 + yf);
                                if (!(ZFOLD && nt == NT - 1)) bl2[nt] = tr_pair(y0 + COUTP * 2 + nt * 32, y0 + 4 * PIXZ + COUTP * 2 + nt * 32);
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < MW; ++i) {
                        if (mvalid[i]) {         // wave-uniform
                            const bf16x8_t af = tr_pair(ldsX + pb0 + toff[i], ldsX + pb1 + toff[i]);
                            const bf16x8_t al = tr_pair(ldsX + pb0 + CINP * 2 + toff[i], ldsX + pb1 + CINP * 2 + toff[i]);
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bf[nt], acc[i][nt], 0, 0, 0);
                                if (!(ZFOLD && nt == NT - 1)) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bl[nt], acc[i][nt], 0, 0, 0);
                                acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[nt], acc[i][nt], 0, 0, 0);
                            }
                            if constexpr (PROJ) {
                                if (i == proj_i) {
#pragma unroll
                                    for (int nt = 0; nt < NT; ++nt) {
                                        accp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bf2[nt], accp[nt], 0, 0, 0);
                                        if (!(ZFOLD && nt == NT - 1)) accp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bl2[nt], accp[nt], 0, 0, 0);
                                        accp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf2[nt], accp[nt], 0, 0, 0);
                                    }
                                }
                            }
                        }
                    }
                    if (bias_wave) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            if (!(ZFOLD && nt == NT - 1)) accb[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bl[nt], accb[nt], 0, 0, 0);
                            accb[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bf[nt], accb[nt], 0, 0, 0);
                        }
                    }
                    continue;
                }
                if constexpr (PROJ) {
                    if (proj_i >= 0) {           // wave-uniform
                        const char* y0 = ldsZ2 + (k32 + 8 * gq + q4) * PIXZ + p * 8;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bf2[nt] = tr_pair(y0 + nt * 32, y0 + 4 * PIXZ + nt * 32);
                    }
                }
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    if (mvalid[i]) {             // wave-uniform
                        const bf16x8_t af = tr_pair(ldsX + pb0 + toff[i], ldsX + pb1 + toff[i]);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[nt], acc[i][nt], 0, 0, 0);
                        if constexpr (PROJ) {
                            if (i == proj_i) {
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
                                    accp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf2[nt], accp[nt], 0, 0, 0);
                            }
                        }
                    }
                }
                if (bias_wave) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        accb[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bf[nt], accb[nt], 0, 0, 0);
                }
            }
            }
        } else {
            const int gq = lane >> 4, col = lane & 15;
            for (int k4 = 0; k4 < a.tile_px; k4 += 4) {
                const int tp = k4 + gq;
                const int pb = mil_pix_base<PIXB>(g, tp, g.stride);
                float bf[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bf[nt] = *reinterpret_cast<const float*>(ldsZ + tp * PIXZ + (nt * 16 + col) * 4);
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    if (mvalid[i]) {
                        const float af = *reinterpret_cast<const float*>(ldsX + pb + toff[i]);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[nt], acc[i][nt], 0, 0, 0);
                    }
                }
                if (bias_wave) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        accb[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, bf[nt], accb[nt], 0, 0, 0);
                }
            }
        }
        MIL_STAMP_MARK(4)
    }
    MIL_STAMP_STORE(a.stamp, NW)

    // one slab per workgroup column; row = (tap*CINP + ci), col = co; bias sums live in tile MT
    constexpr int SLAB_COLS = NT * 16;
    constexpr size_t SLAB_ELEMS = (size_t)(MT + 1 + PMN) * 16 * SLAB_COLS;
    float* slab = a.slab + (size_t)blockIdx.x * SLAB_ELEMS;
    const int gq = lane >> 4, col = lane & 15;
    // ZFOLD: columns 8-15 of the last column tile (the dz_lo products) onto columns 0-7
    auto unfold = [&](float v, int nt) {
        if (ZFOLD && nt == NT - 1) {
            const float up = __shfl_down(v, 8, 16);
            return col < 8 ? v + up : 0.f;
        }
        return v;
    };
    if constexpr (PROJ) {
        if (proj_i >= 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    slab[(size_t)((MT + 1 + proj_mt - PM0) * 16 + gq * 4 + e) * SLAB_COLS + nt * 16 + col] = unfold(accp[nt][e], nt);
        }
    }
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        if (!mvalid[i]) continue;
        const int mt = split * MT_S + wave + NW * i;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                slab[(size_t)(mt * 16 + gq * 4 + e) * SLAB_COLS + nt * 16 + col] = unfold(acc[i][nt][e], nt);
    }
    if (bias_wave) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                slab[(size_t)(MT * 16 + gq * 4 + e) * SLAB_COLS + nt * 16 + col] = unfold(accb[nt][e], nt);
    }
}

// The slab reductions themselves live in reduce.cuh (shared with conv_bwd_fused.hip and the batched launch).
static thread_local MilDeferState g_defer{nullptr, 0, 0};
MilDeferState& mil_defer_state() { return g_defer; }

__global__ __launch_bounds__(32 * MIL_RED_GROUPS) void wgrad_reduce_job_kernel(MilReduceJob j) {
    __shared__ f32x4_t part[MIL_RED_GROUPS][32];
    mil_reduce_job_block(j, blockIdx.x, part);
}

// Every recorded reduction of a backward pass in ONE launch: block -> (job, block inside the job) through the jobs'
// block0 prefix, one lane per job (at most 64 jobs per launch; the host entry point splits longer tables).
__global__ __launch_bounds__(32 * MIL_RED_GROUPS) void wgrad_reduce_all_kernel(const MilReduceJob* __restrict__ jobs, int njobs) {
    __shared__ f32x4_t part[MIL_RED_GROUPS][32];
    __shared__ int which;
    const int b = blockIdx.x;
    if (threadIdx.x < 64) {
        bool mine = false;
        if ((int)threadIdx.x < njobs) {
            const int b0 = jobs[threadIdx.x].block0, nb = jobs[threadIdx.x].n_blocks;
            mine = b >= b0 && b < b0 + nb;
        }
        const unsigned long long m = __ballot(mine);
        if (threadIdx.x == 0) which = m ? __ffsll((long long)m) - 1 : -1;
    }
    __syncthreads();
    const int k = which;
    if (k < 0) return;
    const MilReduceJob j = jobs[k];
    mil_reduce_job_block(j, b - j.block0, part);
}

extern "C" int mil_reduce_job_bytes(void) { return (int)sizeof(MilReduceJob); }

extern "C" int mil_reduce_defer_begin(void* jobs_host, int max_jobs) {
    if (!jobs_host || max_jobs <= 0) return MIL_ERR_ARG;
    g_defer.jobs = static_cast<MilReduceJob*>(jobs_host); g_defer.cap = max_jobs; g_defer.n = 0;
    return MIL_OK;
}

extern "C" int mil_reduce_defer_end(int* njobs) {
    if (njobs) *njobs = g_defer.n;
    g_defer.jobs = nullptr; g_defer.cap = 0; g_defer.n = 0;
    return MIL_OK;
}

// jobs_dev: device copy of the first `njobs` records of a table filled between defer_begin / defer_end.
extern "C" int mil_wgrad_reduce_all(const void* jobs_dev, const void* jobs_host, int njobs, void* stream) {
    if (njobs < 0 || (njobs > 0 && (!jobs_dev || !jobs_host))) return MIL_ERR_ARG;
    const MilReduceJob* hj = static_cast<const MilReduceJob*>(jobs_host);
    const MilReduceJob* dj = static_cast<const MilReduceJob*>(jobs_dev);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int k0 = 0; k0 < njobs; k0 += 64) {               // 64 jobs per launch (one lane per job in the lookup)
        const int n = njobs - k0 < 64 ? njobs - k0 : 64;
        const int first = hj[k0].block0, last = hj[k0 + n - 1].block0 + hj[k0 + n - 1].n_blocks;
        if (last <= first) continue;
        if (k0 != 0) return MIL_ERR_UNSUPPORTED;           // block0 prefixes are relative to job 0: tables are <= 64 jobs
        hipLaunchKernelGGL(wgrad_reduce_all_kernel, dim3(last - first), dim3(32 * MIL_RED_GROUPS), 0, st, dj + k0, n);
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

// ---------------------------------------------------------------------------------------------
struct WgradPlan { int grid_x; int msplit; size_t slab_elems; int slab_cols; int mt; int lds; int tile_px_log2; };

template <typename T, int KS, int CINP, int NT, int MSPLIT>
static int plan_wgrad(ConvGeom& g, WgradPlan& pl, int* lds_z_off, bool proj = false) {
    constexpr int ESZ = T::ESZ;
    constexpr int PIXB = mil_pix_pitch(CINP, ESZ);
    constexpr int PIXZ = mil_pix_pitch(mil_nt_to_cp(NT), ESZ);
    constexpr int RG = KS * KS * (CINP / 8);
    constexpr int MT = (RG + 1) / 2;
    // 256-px tiles when the halo fits comfortably, else 64-px tiles (stride-2 layers, f32 wide layers)
    constexpr bool PF_OK = T::TR16 && (!T::SPLIT || CINP < 64 || (MIL_WGRAD_X3_WIDE_PF && (CINP == 64 || (CINP == 80 && MSPLIT >= 3))));            // bf16, and fp32 with split-precision products: the register-prefetch pipeline
    constexpr bool HALF = mil_wgrad_x3_half(T::SPLIT, KS, CINP, NT, MSPLIT, false);
    const bool half = HALF && !proj;
    const int hmax = mil_wgrad_halo_max(CINP, proj, T::SPLIT, half);
    // split precision, 64 / 80 channels: 128-pixel tiles on eight-wave workgroups (64 channels with the register prefetch of the next tile:
    // 0.89 -> 0.64 ms per five launches; the stamps had 40 % of a tile in the synchronous load) — four 32-pixel k-steps
    // per pair of barriers and two waves per SIMD, where the 64-pixel tiles of the 4-wave form ran the matrix pipe 19 % busy
    constexpr bool WIDE_X3 = T::SPLIT && CINP >= 64;
    for (int lg = 8; lg >= 6; --lg) {
        if (lg == 8 && ((WIDE_X3 && PF_OK) || half)) continue;          // its prefetch registers are sized for 128-pixel tiles
        if (lg == 7 && hmax == 400 && !WIDE_X3) continue;  // 128-pixel tiles only where the prefetch registers are sized for their halo
        mil_geom_tiles(g, lg);
        const int xb = ((((g.hh * g.hw) << g.ti_log2) * PIXB) + 15) & ~15;
        const int zb = (1 << lg) * PIXZ * (proj ? 2 : 1);
        const bool halo_fits_regs = ((g.hh * g.hw) << g.ti_log2) <= hmax;
        if ((xb + zb <= 150 * 1024 && (halo_fits_regs || !PF_OK)) || lg == 6) {
            if (xb + zb > 160 * 1024) return MIL_ERR_UNSUPPORTED;
            pl.lds = xb + zb; *lds_z_off = xb; pl.tile_px_log2 = lg;
            break;
        }
    }
    pl.msplit = MSPLIT; pl.mt = MT; pl.slab_cols = NT * 16;
    {
        constexpr int CGc = CINP / 8;
        const int pmn = proj ? (5 * CGc - 1) / 2 - 2 * CGc + 1 : 0;
        pl.slab_elems = (size_t)(MT + 1 + pmn) * 16 * NT * 16;
    }
    pl.grid_x = 0;                                           // set by run_wgrad once the kernel (and its residency) is known
    return MIL_OK;
}

template <typename T, int KS, int CINP, int NT, int MSPLIT, bool PROJ = false>
static int run_wgrad(const void* x, const void* dz, float* dw, float* db, void* ws, size_t ws_bytes, ConvGeom g,
                     int cout, int cin, int stem_mode, int accumulate, bool query, size_t* need, hipStream_t stream,
                     const void* dz2 = nullptr, float* dw1 = nullptr) {
    WgradPlan pl{};
    int lds_z_off = 0;
    int rc = plan_wgrad<T, KS, CINP, NT, MSPLIT>(g, pl, &lds_z_off, PROJ);
    if (rc != MIL_OK) return rc;
    WgradArgs<T> a{};
    a.x = (const typename T::elem*)x; a.dz = (const typename T::elem*)dz; a.slab = (float*)ws; a.g = g;
    a.ntiles = g.n_groups * g.tiles_y * g.tiles_x; a.tile_px = 1 << pl.tile_px_log2; a.lds_z_off = lds_z_off;
    a.dz2 = (const typename T::elem*)dz2;
    a.lds_z2_off = lds_z_off + (1 << pl.tile_px_log2) * mil_pix_pitch(mil_nt_to_cp(NT), T::ESZ);
    // register-prefetch pipeline for the bf16 path when the halo is small enough for its register budget
    // (split precision at >= 64 channels: the doubled prefetch registers spill — 130-250 VGPRs — so those keep the plain loader)
    constexpr bool PF_OK = T::TR16 && (!T::SPLIT || CINP < 64 || (MIL_WGRAD_X3_WIDE_PF && (CINP == 64 || (CINP == 80 && MSPLIT >= 3))));
    const size_t xb_total = (size_t)g.n_img * g.H * g.W * CINP * T::ESZ;
    const size_t zb_total = (size_t)g.n_img * g.Ho * g.Wo * mil_nt_to_cp(NT) * T::ESZ;
    // buffer descriptors address < 2 GiB: a larger tensor is walked in image chunks, one launch and one set of slabs per chunk
    // (the stem's fp32 space-to-depth input is 2.1 GB at 2048 tiles of 256x256)
    const size_t x_img = (size_t)g.H * g.W * CINP * T::ESZ, z_img = (size_t)g.Ho * g.Wo * mil_nt_to_cp(NT) * T::ESZ;
    constexpr bool HALF = mil_wgrad_x3_half(T::SPLIT, KS, CINP, NT, MSPLIT, PROJ);
    const bool pf = PF_OK && (((g.hh * g.hw) << g.ti_log2) <= mil_wgrad_halo_max(CINP, PROJ, T::SPLIT, HALF)) && (1 << pl.tile_px_log2) <= mil_wgrad_tile_max(CINP, T::SPLIT, HALF) &&
                    g.hh < 1024 && g.hw < 1024;
    int chunk = g.n_img > 0 ? g.n_img : 1;
    if (pf && (xb_total > mil_buffer_limit() || zb_total > mil_buffer_limit())) {
        chunk = mil_imgs_under_2g(x_img > z_img ? x_img : z_img);
        if (chunk >= 16) chunk &= ~15;                            // keep image groups (<= 16 images per tile) intact
        if (chunk < (1 << g.ti_log2)) return MIL_ERR_UNSUPPORTED;
    }
    const int nchunk = g.n_img > 0 ? (g.n_img + chunk - 1) / chunk : 1;
    // 8-wave workgroups where the 4-wave form holds a single wave per SIMD (persistent bf16 form, >= 64 input channels)
    // (split precision: eight waves too — the fragment reads of its three-product loop are not software-pipelined, and twice
    // the waves per SIMD hide their latency instead)
    // (24 channels / 24 columns at four waves per SIMD; 40 channels or columns at two — 240-390 VGPRs would spill at four —, where
    // the 4-wave form's 397 registers leave ONE wave per SIMD)
    constexpr int NW = HALF ? 4 : ((PF_OK && (CINP >= 64 || T::SPLIT)) ? 8 : 4);
    constexpr int NW0 = (T::SPLIT && CINP >= 64) ? 8 : 4;         // waves of the form without register prefetch
    const int nthr = pf ? 64 * NW : 64 * NW0;
    if (PROJ && !pf) return MIL_ERR_UNSUPPORTED;                 // the paired form exists for the persistent bf16 kernel only
    auto kern = pf ? wgrad_kernel<T, KS, CINP, NT, MSPLIT, PF_OK, NW, PROJ && PF_OK> : wgrad_kernel<T, KS, CINP, NT, MSPLIT, false, NW0>;
    if (pl.lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pl.lds) != hipSuccess)
            return MIL_ERR_LAUNCH;
    }
    {   // one fp32 slab per workgroup: no more workgroups than are resident at once, by registers AND LDS (the 64/80-channel
        // slabs are 150-235 KB — a second round of workgroups would only double the slab traffic, which already rivals
        // the activation traffic — and a partial second round leaves CUs idle behind the stragglers)
        const int per_cu = mil_resident_per_cu(kern, pl.lds, 2, nthr);
        int gx = mil_num_cus() * per_cu / MSPLIT;
        // (fewer, longer-running workgroups to write fewer slabs do not pay: the 64/80-channel launches took 1.7x / 3.2x as long on
        // half / a quarter of the workgroups — their time is tiles per workgroup x a load round trip per tile, not slab traffic)
        const int tiles_chunk = (((chunk < g.n_img ? chunk : g.n_img) + (1 << g.ti_log2) - 1) >> g.ti_log2) * g.tiles_y * g.tiles_x;
        if (gx > tiles_chunk) gx = tiles_chunk;
        if (gx < 1) gx = 1;
        pl.grid_x = gx;
    }
    const size_t bytes = pl.slab_elems * pl.grid_x * nchunk * sizeof(float);
    if (query) { *need = bytes; return MIL_OK; }
    if (ws_bytes < bytes || !ws) return MIL_ERR_ARG;
    for (int c = 0; c < nchunk; ++c) {          // a workgroup without tiles (short last chunk) still writes its all-zero slab
        WgradArgs<T> b = a;
        const int i0 = c * chunk, n = g.n_img - i0 < chunk ? g.n_img - i0 : chunk;
        b.g.n_img = n;
        b.g.n_groups = (n + (1 << g.ti_log2) - 1) >> g.ti_log2;
        b.ntiles = b.g.n_groups * g.tiles_y * g.tiles_x;
        b.x = a.x + (size_t)i0 * (x_img / T::ESZ);
        b.dz = a.dz + (size_t)i0 * (z_img / T::ESZ);
        if (a.dz2) b.dz2 = a.dz2 + (size_t)i0 * (z_img / T::ESZ);
        b.x_bytes = (unsigned)(x_img * n); b.z_bytes = (unsigned)(z_img * n);
        b.slab = a.slab + (size_t)c * pl.grid_x * pl.slab_elems;
#ifdef MIL_STAMP
        static MilStampBuf sb;
        const int nwv = nthr / 64;
        b.stamp = (T::SPLIT && MSPLIT == 1) ? sb.get((size_t)pl.grid_x * nwv * 7) : nullptr;
#endif
        hipLaunchKernelGGL(kern, dim3(pl.grid_x, MSPLIT), dim3(nthr), pl.lds, stream, b);
        MIL_CHECK_LAUNCH();
#ifdef MIL_STAMP
        if (b.stamp) {
            static const char* const ph[5] = {"barrier-top", "commit", "fetch-issue", "barrier-x", "gemm"};
            char nm[64];
            snprintf(nm, sizeof nm, "wgrad_kernel<F32S,%d,%d,%d>", KS, CINP, NT);
            sb.report(nm, pl.grid_x, nwv, 5, ph, stream);
        }
#endif
    }
    pl.grid_x *= nchunk;                         // slabs to reduce
    const int n_rows = KS * KS * CINP;
    {
        MilReduceJob j{};
        j.slab = (const float*)ws; j.nslab = pl.grid_x; j.slab_elems = pl.slab_elems; j.slab_cols = pl.slab_cols; j.n_rows = n_rows;
        j.dw = dw; j.db = db; j.cout = cout; j.cin = cin; j.ks = KS; j.kind = 0; j.cinp = CINP; j.stem_mode = stem_mode;
        j.bias_row = pl.mt * 16; j.accumulate = accumulate;
        mil_reduce_or_defer(j, stream);
        MIL_CHECK_LAUNCH();
        if constexpr (PROJ) {        // the projection's rows sit behind the bias tile: rows = input channel, one tap
            MilReduceJob j1 = j;
            j1.slab = (const float*)ws + (size_t)(pl.mt + 1) * 16 * pl.slab_cols;
            j1.n_rows = CINP; j1.dw = dw1; j1.db = nullptr; j1.ks = 1; j1.stem_mode = 0; j1.bias_row = 0;
            mil_reduce_or_defer(j1, stream);
            MIL_CHECK_LAUNCH();
        }
    }
    return MIL_OK;
}

// 3x3/s2 conv + 1x1/s2 projection of a stage-entry block: both weight gradients from one pass over the block input.
template <typename T>
static int dispatch_wgrad_pair(const void* x, const void* dz1, const void* dz2, float* dw3, float* db3, float* dw1, void* ws,
                               size_t ws_bytes, const ConvGeom& g, int cout, int cin, int accumulate, bool query, size_t* need,
                               hipStream_t st) {
    if constexpr (T::DT != MIL_DT_BF16 && T::DT != MIL_DT_F32S) return MIL_ERR_UNSUPPORTED;
    const int cinp = mil_cpad(cin), coutp = mil_cpad(cout);
    if constexpr (T::SPLIT) {                    // split precision: the 20 -> 40 channel entry (the larger ones sit at their register cap)
        if (cinp == 24 && coutp == 40)
            return run_wgrad<T, 3, 24, 3, 1, true>(x, dz1, dw3, db3, ws, ws_bytes, g, cout, cin, 0, accumulate, query, need, st, dz2, dw1);
        return MIL_ERR_UNSUPPORTED;
    } else {
#define MIL_WGP(CI, NTV, MS) return run_wgrad<T, 3, CI, NTV, MS, true>(x, dz1, dw3, db3, ws, ws_bytes, g, cout, cin, 0, accumulate, query, need, st, dz2, dw1)
    if (cinp == 24 && coutp == 40) MIL_WGP(24, 3, 1);
    if (cinp == 40 && coutp == 64) MIL_WGP(40, 4, 1);
    // (64 -> 80: NOT paired.  The paired instantiation spilled 59 VGPRs at its 256-register cap and measured 124 us per launch
    // against 56 + 25 us for the two separate weight-gradient launches (round 4, 256x256 tiles; 306 vs 138 + 54 us at 300x300).)
#ifdef MIL_PAIR64
    if (cinp == 64 && coutp == 80) MIL_WGP(64, 5, 2);
#endif
#undef MIL_WGP
    return MIL_ERR_UNSUPPORTED;
    }
}

template <typename T>
static int dispatch_wgrad(const void* x, const void* dz, float* dw, float* db, void* ws, size_t ws_bytes,
                          const ConvGeom& g, int cout, int cin, int stem_mode, int accumulate, bool query, size_t* need,
                          hipStream_t st) {
    const int cinp = stem_mode ? 16 : mil_cpad(cin), coutp = mil_cpad(cout);
#define MIL_WG(KSV, CI, NTV, MS) return run_wgrad<T, KSV, CI, NTV, MS>(x, dz, dw, db, ws, ws_bytes, g, cout, cin, stem_mode, accumulate, query, need, st)
    if (g.ks == 4 && cinp == 16 && coutp == 24) MIL_WG(4, 16, 2, 1);
    if (g.ks == 4 && cinp == 16 && coutp == 64) MIL_WG(4, 16, 4, 1);     // alt_resnet stem
    if (g.ks == 3) {
        if (cinp == 24 && coutp == 24) MIL_WG(3, 24, 2, 1);
        if (cinp == 40 && coutp == 40) MIL_WG(3, 40, 3, 1);
        if (cinp == 64 && coutp == 64) MIL_WG(3, 64, 4, 1);
        if (cinp == 80 && coutp == 80) { if constexpr (T::SPLIT) { MIL_WG(3, 80, 5, 3); } else { MIL_WG(3, 80, 5, 2); } }
        if (cinp == 24 && coutp == 40) MIL_WG(3, 24, 3, 1);
        if (cinp == 40 && coutp == 64) {
            // split precision: the rows in two grid halves (23 row tiles x 4 column tiles of accumulators beside the doubled
            // prefetch registers spilled 30 VGPRs in one)
            if constexpr (T::SPLIT) { MIL_WG(3, 40, 4, 2); } else { MIL_WG(3, 40, 4, 1); }
        }
        if (cinp == 64 && coutp == 80) MIL_WG(3, 64, 5, 2);
    }
    if (g.ks == 1) {
        if (cinp == 24 && coutp == 40) MIL_WG(1, 24, 3, 1);
        if (cinp == 40 && coutp == 64) MIL_WG(1, 40, 4, 1);
        if (cinp == 64 && coutp == 80) MIL_WG(1, 64, 5, 1);
    }
#undef MIL_WG
    return MIL_ERR_UNSUPPORTED;
}

static int wgrad_entry(const void* x, const void* dz, float* dw, float* db, void* ws, size_t ws_bytes, int n_img,
                       int H, int W, int cin, int Ho, int Wo, int cout, int ks, int stride, int pad, int stem_mode,
                       int accumulate, int dtype, bool query, size_t* need, void* stream) {
    if (n_img < 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return MIL_ERR_ARG;
    ConvGeom g{};
    g.n_img = n_img; g.H = H; g.W = W; g.Ho = Ho; g.Wo = Wo; g.ks = ks; g.stride = stride; g.pad = pad; g.zins = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIL_DT_BF16) return dispatch_wgrad<BF16>(x, dz, dw, db, ws, ws_bytes, g, cout, cin, stem_mode, accumulate, query, need, st);
    if (dtype == MIL_DT_F32) return dispatch_wgrad<F32>(x, dz, dw, db, ws, ws_bytes, g, cout, cin, stem_mode, accumulate, query, need, st);
    if (dtype == MIL_DT_F32S) return dispatch_wgrad<F32S>(x, dz, dw, db, ws, ws_bytes, g, cout, cin, stem_mode, accumulate, query, need, st);
    return MIL_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------
// Fused backward of the stem (gbm/model.py:24-26,51-53): max-pool backward + LeakyReLU backward + the
// 7x7 conv's weight/bias gradient in ONE pass.  The dz tile that the weight-gradient MFMA loop reads from
// LDS is not loaded from HBM: it is built in LDS from the pooled-output gradient and the winner records of
// the (TH/2+1)x(TW/2+1) pooling windows that cover the tile (deterministic gather, one thread per 2x2 pixel
// block as in maxpool_bwd_kernel).  The 4x-larger d(stem output) tensor is never written or read.
struct StemBwdArgs {
    const __bf16* xs;        // [n,H2,W2,16] space-to-depth input (FROM_X = false)
    const float* x;          // [n,3,H,W] the fp32 tiles themselves (FROM_X = true: the s2d tile is rebuilt in LDS, no copy kept)
    int H, W;                // input dims (FROM_X)
    unsigned x_bytes;
    int lds_dump_off;
    const __bf16* gp;        // [n,Hp,Wp,24] gradient of the pooled output ([n,Hp,Wp,20] when gpx == 40: MIL_DT_BF16_DGRAD; fp32 when gpx == 96: MIL_DT_F32S)
    int gpx;                 // bytes per pixel of gp: 48 / 40 (bf16 padded / dense) or 96 / 80 (fp32 padded / dense: MIL_DT_F32S_DGRAD)
    const uint8_t* widx;     // [n,Hp,Wp,24] winner tap (bits 0-3) + "winner <= 0" (bit 4)
    float* slab;
    ConvGeom g;              // geometry of the stem conv as executed (ks 4, stride 1, pad 2, Ho=H2, Wo=W2)
    int Hp, Wp;
    int ntiles;
    int lds_z_off, lds_g_off, lds_i_off;
    unsigned xs_bytes, gp_bytes, wi_bytes;
    float slope;
    unsigned long long* stamp;      // MIL_STAMP diagnostic build only
};

#ifndef MIL_STEM_BWD_WAVES
#define MIL_STEM_BWD_WAVES 2       // measured (us per launch, 2048 tiles): no prefetch 1285 @2 waves/SIMD, 1083 @3 (30 spilled VGPRs), 2102 @4;
                                   // one-step-ahead prefetch (MIL_STEM_BWD_PIPE) 1025 @2 (193 VGPRs, no spill), 1237 @3 (spills)
#endif
#ifndef MIL_STEM_BWD_PIPE
#define MIL_STEM_BWD_PIPE 1
#endif
// X3 (MIL_DT_F32S, FROM_X only): fp32 pooled gradient; the s2d tile and the dz tile hold hi and lo bf16 planes ([hi | lo] per
// pixel record), the weight-gradient GEMM takes x_lo*dz_hi + x_hi*dz_lo + x_hi*dz_hi, the bias sums are the un-rounded fp32 values.
// NPW = pooled-window pieces per thread: 1 when the tile's windows x 3 pieces fit 256 threads (16 x 16 tiles of one image: 81
// windows = 243 pieces — every map of at least 9 pixels), else 2.  An unused second slot is three more loads per thread and tile
// with an out-of-range offset: no memory traffic, but a quarter of this kernel's load instructions, and issuing them is what
// its "fetch issue" phase (23 % of a tile) pays for.
template <bool FROM_X, bool X3 = false, int NPW = 2>
__global__ __launch_bounds__(256, MIL_STEM_BWD_WAVES) void stem_bwd_fused_kernel(StemBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    static_assert(!X3 || FROM_X, "the split-precision form reads the fp32 tiles");
    constexpr int CINP = 16, NT = 2, KS = 4, COUTP = 24;
    constexpr int PIXB = mil_pix_pitch(CINP, X3 ? 4 : 2);   // 48; X3: 80 = [hi 32 B][lo 32 B] + pad
    constexpr int PIXZ = mil_pix_pitch(COUTP, X3 ? 4 : 2);  // dz tile: 48; X3: 112 = [hi 48 B][lo 48 B] + pad
    constexpr int PIXG = 96;                                // pooled-gradient tile: 24 fp32 per window — the DECODED gradient g * lrelu'(winner), see the commit
    // GEMM rows = (tap, s2d channel) in four-channel pieces (one ds_read_b64_tr_b16 each): 16 taps x 3 real pieces = 48 pieces
    // = 12 row tiles; the padding piece (channels 12-15) of a pixel record is never read
    constexpr int NPC = 3, MT = KS * KS * NPC / 4, MW = (MT + 3) / 4;
    constexpr int NPX = mil_halo_np(CINP, 2);
    constexpr int NL = 3;                                   // FROM_X load items per thread (<= 400 halo px: <= 210 pairs x 3 colours)
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ldsX = smem;
    char* ldsZ = smem + a.lds_z_off;
    char* ldsG = smem + a.lds_g_off;
    char* ldsI = smem + a.lds_i_off;
    const int TW = 1 << g.tw_log2, TH = 1 << g.th_log2;
    const int WH = TH / 2 + 1, WW = TW / 2 + 1;             // pooling windows per image of the tile
    const int nwin = (WH * WW) << g.ti_log2;

    const __amdgpu_buffer_rsrc_t rs_x = FROM_X ? mil_rsrc(a.x, a.x_bytes) : mil_rsrc(a.xs, a.xs_bytes);
    const __amdgpu_buffer_rsrc_t rs_g = mil_rsrc(a.gp, a.gp_bytes);
    const __amdgpu_buffer_rsrc_t rs_i = mil_rsrc(a.widx, a.wi_bytes);
    HaloTables<NPX> ht;
    if constexpr (!FROM_X) mil_build_halo_tables<CINP, NPX>(ht, g, tid);
    // FROM_X: load item = (image of the tile, halo row, PAIR of s2d pixels, colour) = input rows 2r, 2r+1 x 4 columns of one
    // colour plane (two 16-byte loads) -> s2d channels 4c..4c+3 of two neighbouring pixels (as stem_fwd_fused_kernel)
    //   l_pos = ti<<20 | hy<<10 | pair (-1: unused);  l_lds = LDS offset of the first pixel's piece | "second pixel is
    //   outside the halo tile" << 20;  l_rel = byte offset of the first load relative to the tile's first load
    int l_pos[NL], l_lds[NL], l_rel[NL];
    if constexpr (FROM_X) {
        const int npair = (g.hw + 1) >> 1, rows = g.hh << g.ti_log2;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + 256 * i;
            l_pos[i] = -1; l_lds[i] = a.lds_dump_off | (1 << 20); l_rel[i] = 0;
            if (idx < rows * npair * 3) {
                const int pair = idx % npair, t = idx / npair;
                const int c = t % 3, row = t / 3;
                const int ti = row / g.hh, hy = row - ti * g.hh;
                l_pos[i] = (ti << 20) | (hy << 10) | pair;
                l_lds[i] = ((row * g.hw + 2 * pair) * PIXB + c * 8) | ((2 * pair + 1 >= g.hw) ? 1 << 20 : 0);
                l_rel[i] = (((ti * 3 + c) * a.H + 2 * hy) * a.W + 4 * pair) * 4;
            }
        }
    }
    u32x4_t lr0[NL], lr1[NL];
    auto fetch_x = [&](const TileOrigin& o) {
        const int y0 = o.oy0 - g.pad, c0 = 2 * (o.ox0 - g.pad);          // first s2d row / first input column of the halo tile
        const int base = (((o.img0 * 3) * a.H + 2 * y0) * a.W + c0) * 4;       // may be negative; valid lanes are not
        const int ilim = g.n_img - o.img0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int p = l_pos[i];
            const bool ok = (p >= 0) & ((p >> 20) < ilim) & ((unsigned)(y0 + ((p >> 10) & 1023)) < (unsigned)g.H) &
                            ((unsigned)(c0 + 4 * (p & 1023)) < (unsigned)a.W);
            const unsigned off = ok ? (unsigned)(base + l_rel[i]) : MIL_OOB;
            lr0[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
            lr1[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off + (unsigned)(a.W * 4), 0, 0);
        }
    };
    auto commit_x = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const f32x4_t v0 = __builtin_bit_cast(f32x4_t, lr0[i]), v1 = __builtin_bit_cast(f32x4_t, lr1[i]);
            const float fa[4] = {v0[0], v0[1], v1[0], v1[1]}, fb[4] = {v0[2], v0[3], v1[2], v1[3]};
            bf16x4_t pa, pb, qa, qb;
            if constexpr (X3) {
                mil_split4(f32x4_t{fa[0], fa[1], fa[2], fa[3]}, pa, qa);
                mil_split4(f32x4_t{fb[0], fb[1], fb[2], fb[3]}, pb, qb);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { pa[j] = (__bf16)fa[j]; pb[j] = (__bf16)fb[j]; }
            }
            const int d0 = l_lds[i] & 0xFFFFF;
            const int d1 = (l_lds[i] >> 20) ? a.lds_dump_off : d0 + PIXB;
            *reinterpret_cast<bf16x4_t*>(ldsX + d0) = pa;
            *reinterpret_cast<bf16x4_t*>(ldsX + d1) = pb;
            if constexpr (X3) {                              // lo plane 32 bytes behind the hi plane (the dump slot has room for both)
                *reinterpret_cast<bf16x4_t*>(ldsX + d0 + 32) = qa;
                *reinterpret_cast<bf16x4_t*>(ldsX + d1 + 32) = qb;
            }
        }
    };
    // pooled-window pieces: item = window*3 + j; gradient piece = 16 B (8 channels), winner piece = 8 B
    int w_pos[NPW], w_rel[NPW], w_lds[NPW];                  // w_rel = pooled pixel (relative) * 4 + piece
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int idx = tid + 256 * i;
        w_pos[i] = -1; w_rel[i] = 0; w_lds[i] = 0;
        if (idx < nwin * 3) {
            const int win = idx / 3, j = idx - win * 3;
            const int ti = win / (WH * WW), rem = win - ti * (WH * WW);
            const int wy = rem / WW, wx = rem - wy * WW;
            w_pos[i] = (ti << 20) | (wy << 10) | wx;
            w_rel[i] = (((ti * a.Hp + wy) * a.Wp + wx) << 2) | j;
            w_lds[i] = win * COUTP + j * 8;
        }
    }
    auto fetch_win = [&](u32x4_t (&rg)[NPW], u32x4_t (&rg2)[X3 ? NPW : 1], u32x2_t (&ri)[NPW], const TileOrigin& o) {
        const int py0 = o.oy0 >> 1, px0 = o.ox0 >> 1;
        const int base = (o.img0 * a.Hp + py0) * a.Wp + px0;
        const int ylim = a.Hp - py0, xlim = a.Wp - px0, ilim = g.n_img - o.img0;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int p = w_pos[i];
            const bool ok = p >= 0 && (p >> 20) < ilim && ((p >> 10) & 1023) < ylim && (p & 1023) < xlim;
            const int pix = base + (w_rel[i] >> 2), j = w_rel[i] & 3;
            if constexpr (X3) {                               // eight fp32 channels = two 16-byte loads
                const unsigned goff = ok ? (unsigned)(pix * a.gpx + j * 32) : MIL_OOB;
                rg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, goff, 0, 0);
                rg2[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (ok && !(a.gpx != 96 && j == 2)) ? goff + 16 : MIL_OOB, 0, 0);      // dense: channels 20-23 do not exist
            } else {
            const unsigned goff = ok ? (unsigned)(pix * a.gpx + j * 16) : MIL_OOB;
            const u32x2_t lo = __builtin_amdgcn_raw_buffer_load_b64(rs_g, goff, 0, 0);
            const u32x2_t hi = __builtin_amdgcn_raw_buffer_load_b64(rs_g, (a.gpx != 48 && j == 2) ? MIL_OOB : goff + 8, 0, 0);      // dense: channels 20-23 do not exist
            rg[i] = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
            }
            ri[i] = __builtin_amdgcn_raw_buffer_load_b64(rs_i, ok ? (unsigned)(pix * COUTP + j * 8) : MIL_OOB, 0, 0);
        }
    };

    // per-lane tr-read offsets of this wave's row tiles: row piece P = 4*mt + (lane&3) = (tap, four s2d channels)
    int toff[MW];
    bool mvalid[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int mt = wave + 4 * i;
        mvalid[i] = mt < MT;
        const int P = mvalid[i] ? 4 * mt + (lane & 3) : 0;
        const int tap = P / NPC, c3 = P - tap * NPC;
        toff[i] = ((tap / KS) * g.hw + (tap % KS)) * PIXB + c3 * 8;
    }
    // The bias gradient sum_p dz[p][co] needs no MFMA: the thread that builds (2x2 pixel block, 6 channels) of the dz tile
    // keeps the running sums of what it wrote (6 registers), reduced over the workgroup in a fixed order at the end.
    float bsum[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x4_t acc[MW][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < MW; ++i) acc[i][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, gq = lane >> 4;
    const int wpl0 = mil_pix_base<PIXB>(g, 8 * gq + q4, 1), wpl1 = mil_pix_base<PIXB>(g, 8 * gq + q4 + 4, 1);
    // dz builder: thread -> (2x2 pixel block, 6-channel group) of the tile: 64 blocks x 4 groups = all 256 threads
    const int bitem = tid >> 2, bc6 = tid & 3;
    const int bW = TW / 2, bH = TH / 2;
    const int b_ti = bitem / (bW * bH), b_rem = bitem - b_ti * (bW * bH);
    const int b_y = b_rem / bW, b_x = b_rem - b_y * bW;

    TileWalker cur, nxt;
    const int bid = mil_xcd_block_id();
    cur.init(g, bid, gridDim.x);
    nxt = cur; nxt.advance();
    u32x4_t rx[FROM_X ? 1 : NPX], rgp[NPW], rgp2[X3 ? NPW : 1];
    u32x2_t rwi[NPW];
    if (bid < a.ntiles) {
        if constexpr (FROM_X) fetch_x(cur.origin(g)); else mil_fetch_halo<CINP, NPX>(rx, rs_x, ht, g, cur.origin(g));
        fetch_win(rgp, rgp2, rwi, cur.origin(g));
    }
    MIL_STAMP_DECL(7)
    for (int tile = bid; tile < a.ntiles; tile += gridDim.x) {
        MIL_STAMP_BEGIN()
        __syncthreads();                         // previous tile's MFMA loop is done with ldsX / ldsZ
        MIL_STAMP_MARK(0)
        if constexpr (FROM_X) commit_x(); else mil_commit_halo<NPX>(rx, ldsX, ht);
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            if (w_pos[i] >= 0) {
                // a window's record is decoded ONCE here (the masked gradient as fp32) instead of by each of the four 2x2 blocks
                // around it in the gather (round 5: 24 decodes per gather thread and tile, a third of its instructions in bf16)
                f32x4_t g0, g1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned wb = (rwi[i][j >> 2] >> (8 * (j & 3))) & 0xffu;
                    float gj;
                    if constexpr (X3) gj = __uint_as_float(j < 4 ? rgp[i][j] : rgp2[i][j - 4]);
                    else { const unsigned gw = rgp[i][j >> 1]; gj = __uint_as_float((j & 1) ? (gw & 0xffff0000u) : (gw << 16)); }
                    const float gm = (wb & 16u) ? gj * a.slope : gj;
                    if (j < 4) g0[j] = gm; else g1[j - 4] = gm;
                }
                *reinterpret_cast<f32x4_t*>(ldsG + w_lds[i] * 4) = g0;
                *reinterpret_cast<f32x4_t*>(ldsG + w_lds[i] * 4 + 16) = g1;
                *reinterpret_cast<u32x2_t*>(ldsI + w_lds[i]) = rwi[i];
            }
        }
        MIL_STAMP_MARK(1)
        __syncthreads();
        MIL_STAMP_MARK(2)
        if (tile + (int)gridDim.x < a.ntiles) {
            if constexpr (FROM_X) fetch_x(nxt.origin(g)); else mil_fetch_halo<CINP, NPX>(rx, rs_x, ht, g, nxt.origin(g));
            fetch_win(rgp, rgp2, rwi, nxt.origin(g));
        }
        cur = nxt; nxt.advance();
        MIL_STAMP_MARK(3)

#ifndef MIL_EXP_STEM_NO_GATHER
        // ---- dz tile = lrelu'(stem) * maxpool^T(g): gather over the 4 windows that cover a 2x2 block --------
        // A window's gradient goes to exactly one pixel (its recorded winner tap); per (window, channel) the tap and
        // the masked gradient are decoded once, then tested against the (at most 4) taps this block's pixels have in
        // that window: 9 (window, pixel) pairs in all.
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
#pragma unroll
                for (int wx = 0; wx < 2; ++wx) {
                    const int win = (b_ti * WH + b_y + wy) * WW + b_x + wx;
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
                            const int ky = dy + 1 - 2 * wy;          // tap row of pixel 2*b_y+dy inside window b_y+wy
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
                    const int tp = (b_ti << (g.tw_log2 + g.th_log2)) + ((2 * b_y + dy) << g.tw_log2) + 2 * b_x + dx;
                    unsigned* dst = reinterpret_cast<unsigned*>(ldsZ + tp * PIXZ + bc6 * 12);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        bf16x2_t pr, lo;
                        if constexpr (X3) mil_split2(gsum[dy][dx][2 * k], gsum[dy][dx][2 * k + 1], pr, lo);
                        else { pr[0] = (__bf16)gsum[dy][dx][2 * k]; pr[1] = (__bf16)gsum[dy][dx][2 * k + 1]; }
                        dst[k] = __builtin_bit_cast(unsigned, pr);
                        if constexpr (X3) {                      // lo plane of the dz record; the bias sums take the fp32 values themselves
                            dst[12 + k] = __builtin_bit_cast(unsigned, lo);
                            bsum[2 * k] += gsum[dy][dx][2 * k]; bsum[2 * k + 1] += gsum[dy][dx][2 * k + 1];
                        } else {
                        bsum[2 * k] += (float)pr[0]; bsum[2 * k + 1] += (float)pr[1];      // the rounded values, as the MFMA loop sees them
                        }
                    }
                }
        }
#endif
        MIL_STAMP_MARK(4)
        __syncthreads();
        MIL_STAMP_MARK(5)

        // ---- weight gradient: rows (tap, s2d channel), cols stem channel, K = the tile's 256 pixels ---------
#ifndef MIL_EXP_STEM_NO_MFMA
        // One 32-pixel k-step ahead (MIL_STEM_BWD_PIPE): the transposed reads of step k+1 are issued before the MFMAs of
        // step k and scheduling fences keep that order.  Left to itself hipcc reads each row tile's fragment right in
        // front of its two MFMAs behind an lgkmcnt(0): five LDS round trips per k-step, forty per tile — the whole tile
        // time of this kernel.  Every wave owns MW = 3 row tiles that all exist (MT = 12), so there is no validity branch.
#if MIL_STEM_BWD_PIPE
        {
            static_assert(MT == 4 * MW, "every wave owns MW full row tiles");
            constexpr int NL2 = X3 ? 2 : 1;                  // operand planes: hi (+ lo)
            // X3: column tile 1 has four real columns (stem channels 16-19), so its columns 4-7 carry the LO halves of the same
            // channels (the lanes that read the 4-channel piece of channels 20-23 read the lo plane's piece of 16-19 instead):
            // x_hi * [dz_hi | dz_lo] then x_lo * dz_hi — two MFMAs instead of three; columns 4-7 are added onto 0-3 at the end.
            // bf[1][1] holds that folded fragment (there is no separate lo fragment of column tile 1).
            const int fold1 = (X3 && p4 == 1) ? 40 : 0;
            bf16x8_t bc[NL2][NT], ac[NL2][MW], bn[NL2][NT], an[NL2][MW];
            auto load = [&](int k32, bf16x8_t (&bf)[NL2][NT], bf16x8_t (&af)[NL2][MW]) {
                const int kb = mil_pix_base<PIXB>(g, k32, 1);
                const char* z0 = ldsZ + (k32 + 8 * gq + q4) * PIXZ + p4 * 8;
#pragma unroll
                for (int pl = 0; pl < NL2; ++pl) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int zo = (X3 && pl == 1 && nt == 1) ? 32 + fold1 : pl * 48 + nt * 32;
                        bf[pl][nt] = mil_tr_pair(z0 + zo, z0 + zo + 4 * PIXZ);
                    }
#pragma unroll
                    for (int i = 0; i < MW; ++i) af[pl][i] = mil_tr_pair(ldsX + kb + wpl0 + toff[i] + pl * 32, ldsX + kb + wpl1 + toff[i] + pl * 32);
                }
            };
            load(0, bc, ac);
#pragma unroll
            for (int k32 = 0; k32 < 256; k32 += 32) {
                if (k32 + 32 < 256) load(k32 + 32, bn, an);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if constexpr (X3) {
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[1][i], bc[0][nt], acc[i][nt], 0, 0, 0);
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[0][i], bc[1][nt], acc[i][nt], 0, 0, 0);      // nt 1: the folded fragment = x_hi * [dz_hi | dz_lo]
                            if (nt == 1) continue;
                        }
                        acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[0][i], bc[0][nt], acc[i][nt], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pl = 0; pl < NL2; ++pl) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bc[pl][nt] = bn[pl][nt];
#pragma unroll
                    for (int i = 0; i < MW; ++i) ac[pl][i] = an[pl][i];
                }
            }
        }
#else
#pragma unroll 2
        for (int k32 = 0; k32 < 256; k32 += 32) {
            const int kb = mil_pix_base<PIXB>(g, k32, 1);
            const int pb0 = kb + wpl0, pb1 = kb + wpl1;
            const char* z0 = ldsZ + (k32 + 8 * gq + q4) * PIXZ + p4 * 8;
            const char* z1 = z0 + 4 * PIXZ;
            bf16x8_t bf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = mil_tr_pair(z0 + nt * 32, z1 + nt * 32);
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                if (mvalid[i]) {
                    const bf16x8_t af = mil_tr_pair(ldsX + pb0 + toff[i], ldsX + pb1 + toff[i]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[nt], acc[i][nt], 0, 0, 0);
                }
            }
        }
#endif
#endif      // MIL_EXP_STEM_NO_MFMA
        MIL_STAMP_MARK(6)
    }
    MIL_STAMP_STORE(a.stamp, 4)

    constexpr int SLAB_COLS = NT * 16;
    constexpr size_t SLAB_ELEMS = (size_t)(MT + 1) * 16 * SLAB_COLS;
    float* slab = a.slab + (size_t)blockIdx.x * SLAB_ELEMS;
    const int col = lane & 15;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        if (!mvalid[i]) continue;
        const int mt = wave + 4 * i;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[i][nt][e];
                if (X3 && nt == 1 && MIL_STEM_BWD_PIPE) {      // folded column tile: columns 4-7 (the lo products) onto columns 0-3
                    const float up = __shfl_down(v, 4, 16);
                    v = col < 4 ? v + up : 0.f;
                }
                slab[(size_t)(mt * 16 + gq * 4 + e) * SLAB_COLS + nt * 16 + col] = v;
            }
    }
    // bias sums -> slab row MT*16: 64 pixel-block threads per channel, added in thread order
    __syncthreads();                             // the last tile's MFMA loop is done with the LDS tiles
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

#include "stem_bwd_walk.cuh"

// xs != null: the bf16 space-to-depth copy [n,H2,W2,16] is the conv input; else x = the fp32 tiles [n,3,H,W] (H = 2*H2,
// W = 2*W2, W % 4 == 0) and the kernel rebuilds its s2d tiles itself.  Inputs beyond the 2 GiB reach of a buffer
// descriptor are walked in image chunks, later chunks accumulating into dW/db.
static int stem_bwd_entry(const void* xs, const float* x, const void* gp, const uint8_t* widx, float* dw, float* db, void* ws,
                          size_t ws_bytes, int n, int H2, int W2, float slope, int accumulate, int dtype, bool from_x, bool query,
                          size_t* need, void* stream) {
    const bool x3 = dtype == MIL_DT_F32S || dtype == MIL_DT_F32S_DGRAD;      // fp32 g_pool, split-precision products: from the fp32 tiles only
    if (dtype != MIL_DT_BF16 && dtype != MIL_DT_BF16_DGRAD && !(x3 && from_x)) return MIL_ERR_UNSUPPORTED;
    const int gpx = x3 ? (dtype == MIL_DT_F32S_DGRAD ? 80 : 96) : dtype == MIL_DT_BF16_DGRAD ? 40 : 48;   // g_pool [n,Hp,Wp,20] dense or [n,Hp,Wp,24]
    if (n <= 0 || H2 <= 0 || W2 <= 0) return MIL_ERR_ARG;
    if (from_x && (W2 & 1)) return MIL_ERR_UNSUPPORTED;      // 16-byte input pieces: W % 4 == 0
    const int PIXB = mil_pix_pitch(16, x3 ? 4 : 2), PIXZ = mil_pix_pitch(24, x3 ? 4 : 2), PIXG = 96;
    constexpr int MT = 12;                                  // 16 taps x 3 four-channel row pieces / 4 (see the kernel)
    StemBwdArgs a{};
    ConvGeom& g = a.g;
    g.n_img = n; g.H = H2; g.W = W2; g.Ho = H2; g.Wo = W2; g.ks = 4; g.stride = 1; g.pad = 2; g.zins = 0;
    mil_geom_tiles(g, 8);
#ifndef MIL_STEM_BWD_WIDE
#define MIL_STEM_BWD_WIDE 0           // 1: 32 x 8 tiles instead of 16 x 16 where the map allows (35 x 11 halo: 280-byte row segments instead of 152)
#endif
    if (MIL_STEM_BWD_WIDE && g.tw_log2 == 4 && g.th_log2 == 4 && g.ti_log2 == 0 && W2 >= 32) mil_geom_set(g, 5, 3, 0);
    a.Hp = (H2 - 1) / 2 + 1; a.Wp = (W2 - 1) / 2 + 1;
    a.H = 2 * H2; a.W = 2 * W2;
    const int halo_px = (g.hh * g.hw) << g.ti_log2;
    const int nwin = (((1 << g.th_log2) / 2 + 1) * ((1 << g.tw_log2) / 2 + 1)) << g.ti_log2;
    if (halo_px > 400 || nwin * 3 > 512 || g.hh >= 1024 || g.hw >= 1024) return MIL_ERR_UNSUPPORTED;
    if (from_x && ((g.hh << g.ti_log2) * ((g.hw + 1) >> 1) * 3 > 3 * 256)) return MIL_ERR_UNSUPPORTED;
    // images per launch: every tensor of a launch must stay under 2 GiB; whole tiles of images per chunk
    const size_t in_img = from_x ? (size_t)12 * a.H * a.W : (size_t)H2 * W2 * 32, gp_img = (size_t)a.Hp * a.Wp * gpx;
    a.gpx = gpx;
    int chunk = mil_imgs_under_2g(in_img > gp_img ? in_img : gp_img);
    if (chunk >= (1 << g.ti_log2)) chunk &= ~((1 << g.ti_log2) - 1); else return MIL_ERR_UNSUPPORTED;
    if (chunk > n) chunk = n;
    const int xb = (halo_px * PIXB + 15) & ~15, zb = 256 * PIXZ, gb = (nwin * PIXG + 15) & ~15, ib = (nwin * 24 + 15) & ~15;
    const int lds = xb + zb + gb + ib + 48;                 // + dump slot (FROM_X: second pixel of a pair behind an odd-width halo; hi + lo)
    const int groups = (chunk + (1 << g.ti_log2) - 1) >> g.ti_log2;
    const int ntiles_max = groups * g.tiles_y * g.tiles_x;
#ifdef MIL_EXP_STEM_BWD_NPW2                                // A/B build: always two window slots per thread (the form until round 5)
    const bool one = false;
#else
    const bool one = nwin * 3 <= 256;                       // one window piece per thread
#endif
    auto kern = x3 ? (one ? stem_bwd_fused_kernel<true, true, 1> : stem_bwd_fused_kernel<true, true, 2>)
              : from_x ? (one ? stem_bwd_fused_kernel<true, false, 1> : stem_bwd_fused_kernel<true, false, 2>)
                       : (one ? stem_bwd_fused_kernel<false, false, 1> : stem_bwd_fused_kernel<false, false, 2>);
    int grid = mil_num_cus() * mil_resident_per_cu(kern, lds, 4) * 2;          // two rounds of the resident set
    if (grid > ntiles_max) grid = ntiles_max;
    const size_t slab_elems = (size_t)(MT + 1) * 16 * 32;
    const size_t bytes = slab_elems * grid * sizeof(float);
    if (query) { *need = bytes; return MIL_OK; }
    if (!ws || ws_bytes < bytes) return MIL_ERR_ARG;
    a.slab = (float*)ws;
    a.lds_z_off = xb; a.lds_g_off = xb + zb; a.lds_i_off = xb + zb + gb; a.lds_dump_off = xb + zb + gb + ib; a.slope = slope;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // the row-walk form (stem_bwd_walk.cuh): one workgroup per image, no more slabs than the tiled form's workspace holds
    const bool walk = mil_stem_walk_wanted_bwd(n, H2, W2, from_x, !x3, g.tiles_y * g.tiles_x, mil_num_cus() * 2);
    int walk_grid = 0;
    if (walk) {
        auto wk = stem_bwd_walk_kernel;
        static std::atomic<unsigned long long> attr_set{0};
        if (mil_device_needs(attr_set)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(wk), hipFuncAttributeMaxDynamicSharedMemorySize, SBW_LDS) != hipSuccess) return MIL_ERR_LAUNCH;
            mil_device_done(attr_set);
        }
        walk_grid = mil_num_cus() * mil_resident_per_cu(wk, SBW_LDS, 2);
        if (walk_grid > grid) walk_grid = grid;
    }
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int nc = n - i0 < chunk ? n - i0 : chunk;
        StemBwdArgs c = a;
        c.g.n_img = nc; c.g.n_groups = (nc + (1 << g.ti_log2) - 1) >> g.ti_log2;
        c.ntiles = c.g.n_groups * g.tiles_y * g.tiles_x;
        if (from_x) { c.x = x + (size_t)i0 * 3 * a.H * a.W; c.x_bytes = (unsigned)((size_t)nc * 12 * a.H * a.W); }
        else { c.xs = (const __bf16*)xs + (size_t)i0 * H2 * W2 * 16; c.xs_bytes = (unsigned)((size_t)nc * H2 * W2 * 32); }
        c.gp = (const __bf16*)gp + (size_t)i0 * a.Hp * a.Wp * (gpx / 2); c.gp_bytes = (unsigned)((size_t)nc * gp_img);
        c.widx = widx + (size_t)i0 * a.Hp * a.Wp * 24; c.wi_bytes = (unsigned)((size_t)nc * a.Hp * a.Wp * 24);
        const int gr = walk ? (walk_grid < nc ? walk_grid : nc) : (grid < c.ntiles ? grid : c.ntiles);
#ifdef MIL_STAMP
        static MilStampBuf sb;
        c.stamp = sb.get((size_t)gr * 4 * 9);
#endif
        if (walk) hipLaunchKernelGGL(stem_bwd_walk_kernel, dim3(gr), dim3(256), SBW_LDS, st, c);
        else hipLaunchKernelGGL(kern, dim3(gr), dim3(256), lds, st, c);
        MIL_CHECK_LAUNCH();
#ifdef MIL_STAMP
        static const char* const ph[7] = {"barrier-top", "commit", "barrier-x", "fetch-issue", "gather", "barrier-z", "gemm"};
        sb.report(walk ? "stem_bwd_walk_kernel" : x3 ? "stem_bwd_fused_kernel<x3>" : "stem_bwd_fused_kernel", gr, 4, 7, ph, st);
#endif
        MilReduceJob j{};
        j.slab = (const float*)ws; j.nslab = gr; j.slab_elems = slab_elems; j.slab_cols = 32; j.n_rows = 16 * 12;
        j.dw = dw; j.db = db; j.cout = 20; j.cin = 3; j.ks = 7; j.kind = 0; j.cinp = 12; j.stem_mode = 1;      // rows tap*12 + s2d channel
        j.bias_row = MT * 16;        // the row behind the weight rows: the kernel's VALU bias sums
        j.accumulate = (i0 > 0) ? 1 : accumulate;
        mil_reduce_or_defer(j, st, /*may_defer=*/chunk >= n);      // a split launch re-uses the slabs per chunk
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

extern "C" int mil_stem_bwd_fused_workspace(size_t* bytes, int n, int H2, int W2, int dtype) {
    if (!bytes) return MIL_ERR_ARG;
    return stem_bwd_entry(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, n, H2, W2, 0.1f, 0, dtype, false, true, bytes, nullptr);
}

extern "C" int mil_stem_bwd_fused(const void* xs, const void* g_pool, const uint8_t* widx, float* dw, float* db,
                                  void* workspace, size_t workspace_bytes, int n, int H2, int W2, float slope,
                                  int accumulate, int dtype, void* stream) {
    if (!xs || !g_pool || !widx || !dw || !db) return MIL_ERR_ARG;
    size_t need = 0;
    return stem_bwd_entry(xs, nullptr, g_pool, widx, dw, db, workspace, workspace_bytes, n, H2, W2, slope, accumulate, dtype, false, false,
                          &need, stream);
}

extern "C" int mil_stem_bwd_fused_nchw_workspace(size_t* bytes, int n, int H, int W, int dtype) {
    if (!bytes) return MIL_ERR_ARG;
    if (H <= 0 || W <= 0 || (H & 1) || (W & 3)) return MIL_ERR_UNSUPPORTED;
    return stem_bwd_entry(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, n, H / 2, W / 2, 0.1f, 0, dtype, true, true, bytes, nullptr);
}

extern "C" int mil_stem_bwd_fused_nchw(const float* x, const void* g_pool, const uint8_t* widx, float* dw, float* db,
                                       void* workspace, size_t workspace_bytes, int n, int H, int W, float slope,
                                       int accumulate, int dtype, void* stream) {
    if (!x || !g_pool || !widx || !dw || !db) return MIL_ERR_ARG;
    if (H <= 0 || W <= 0 || (H & 1) || (W & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return MIL_ERR_UNSUPPORTED;
    size_t need = 0;
    return stem_bwd_entry(nullptr, x, g_pool, widx, dw, db, workspace, workspace_bytes, n, H / 2, W / 2, slope, accumulate, dtype, true, false,
                          &need, stream);
}

static int wgrad_pair_entry(const void* x, const void* dz1, const void* dz2, float* dw3, float* db3, float* dw1, void* ws,
                            size_t ws_bytes, int n_img, int H, int W, int cin, int Ho, int Wo, int cout, int accumulate,
                            int dtype, bool query, size_t* need, void* stream) {
    if (n_img < 0 || H <= 0 || W <= 0 || Ho != (H - 1) / 2 + 1 || Wo != (W - 1) / 2 + 1) return MIL_ERR_ARG;
    if (dtype != MIL_DT_BF16 && dtype != MIL_DT_F32S) return MIL_ERR_UNSUPPORTED;
    ConvGeom g{};
    g.n_img = n_img; g.H = H; g.W = W; g.Ho = Ho; g.Wo = Wo; g.ks = 3; g.stride = 2; g.pad = 1; g.zins = 0;
    if (dtype == MIL_DT_F32S)
        return dispatch_wgrad_pair<F32S>(x, dz1, dz2, dw3, db3, dw1, ws, ws_bytes, g, cout, cin, accumulate, query, need,
                                         reinterpret_cast<hipStream_t>(stream));
    return dispatch_wgrad_pair<BF16>(x, dz1, dz2, dw3, db3, dw1, ws, ws_bytes, g, cout, cin, accumulate, query, need,
                                     reinterpret_cast<hipStream_t>(stream));
}

extern "C" int mil_conv_wgrad_pair_workspace(size_t* bytes, int n_img, int H, int W, int cin, int Ho, int Wo, int cout, int dtype) {
    if (!bytes) return MIL_ERR_ARG;
    return wgrad_pair_entry(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, n_img, H, W, cin, Ho, Wo, cout, 0,
                            dtype, true, bytes, nullptr);
}

// Weight gradients of a stage-entry block's two stride-2 convs from ONE pass over the block input x [n,H,W,cpad(cin)]:
// dw3 [cout,cin,3,3], db3 [cout] from dz1, and dw1 [cout,cin,1,1] from dz2 (both [n,Ho,Wo,cpad(cout)]).  bf16 only.
extern "C" int mil_conv_wgrad_pair(const void* x, const void* dz1, const void* dz2, float* dw3, float* db3, float* dw1,
                                   void* workspace, size_t workspace_bytes, int n_img, int H, int W, int cin, int Ho, int Wo,
                                   int cout, int accumulate, int dtype, void* stream) {
    if (!x || !dz1 || !dz2 || !dw3 || !dw1 || !workspace) return MIL_ERR_ARG;
    size_t need = 0;
    return wgrad_pair_entry(x, dz1, dz2, dw3, db3, dw1, workspace, workspace_bytes, n_img, H, W, cin, Ho, Wo, cout, accumulate,
                            dtype, false, &need, stream);
}

extern "C" int mil_conv_wgrad_workspace(size_t* bytes, int n_img, int H, int W, int cin, int Ho, int Wo, int cout,
                                        int ks, int stride, int pad, int stem_mode, int dtype) {
    if (!bytes) return MIL_ERR_ARG;
    return wgrad_entry(nullptr, nullptr, nullptr, nullptr, nullptr, 0, n_img, H, W, cin, Ho, Wo, cout, ks, stride, pad,
                       stem_mode, 0, dtype, true, bytes, nullptr);
}

extern "C" int mil_conv_wgrad(const void* x, const void* dz, float* dw, float* db, void* workspace,
                              size_t workspace_bytes, int n_img, int H, int W, int cin, int Ho, int Wo, int cout,
                              int ks, int stride, int pad, int stem_mode, int accumulate, int dtype, void* stream) {
    if (!x || !dz || !dw) return MIL_ERR_ARG;
    size_t need = 0;
    return wgrad_entry(x, dz, dw, db, workspace, workspace_bytes, n_img, H, W, cin, Ho, Wo, cout, ks, stride, pad,
                       stem_mode, accumulate, dtype, false, &need, stream);
}
