// NHWC implicit-GEMM convolution on CDNA4 MFMA: forward conv and data-gradient (dgrad) of the
// residual-block convolutions of the tile encoder (reference: nnBlocks.py:169-171 conv1/conv2,
// gbm/model.py:24 stem, gbm/model.py:38-40 1x1 projection shortcut).
//
//   GEMM view:  M = output pixels (a TI x TH x TW tile per workgroup),  N = output channels
//               (NT 16-wide MFMA column tiles),  K = taps x input channels, flattened in groups of 8
//               channels so that one 16x16x32 MFMA consumes 4 (tap, channel-group) pairs.
//   A operand:  the input halo tile, staged ONCE in LDS and re-read for every tap (no im2col in HBM).
//   B operand:  weights pre-packed on device in MFMA B-fragment order (pack.hip), staged in LDS.
//   Epilogue:   accumulators -> LDS (fp32) -> per-pixel 8-channel rows:
//               v = acc + bias + residual;  v = lrelu(v);  v *= lrelu'(act);  16-B NHWC stores.
//   dgrad:      the same kernel run over dz with transposed+flipped packed weights; stride-2 dgrad
//               reads dz through the zero-insert loader (transposed convolution).
#include "geom.cuh"
#include "pf_common.cuh"

template <typename T>
struct ConvArgs {
    const typename T::elem* x;
    const typename T::elem* w;      // packed [nsteps][NT][64][8]
    const float* bias;              // padded to NT*16, or null
    const typename T::elem* res;    // residual / addend in output layout, or null
    const typename T::elem* act;    // saved activation for the lrelu' mask, or null
    typename T::elem* y;
    ConvGeom g;
    int nsteps, kc;                 // k-steps total / per weight chunk held in LDS
    int lds_w_off;                  // byte offset of the weight chunk in LDS
    int lds_a2_off, lds_dump_rel;   // persistent kernels: offset of the second halo buffer (0 = none); dump slot relative to a halo buffer
    int apply_lrelu;
    float slope;
};

template <typename T, int CINP, int NT, int MTW>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int ESZ = T::ESZ;
    constexpr int PIXB = mil_pix_pitch(CINP, ESZ);
    constexpr int CG = CINP / 8;
    constexpr int FRAGB = 8 * ESZ;
    constexpr int COUTP = mil_nt_to_cp(NT);
    constexpr int NGRP = COUTP / 8;
    constexpr int TILE_PX = 64 * MTW;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    // wave id through readfirstlane: provably wave-uniform, so branches on it are scalar branches (an MFMA or a
    // ds_read_b64_tr_b16 inside an EXEC-masked region would still execute / need all lanes)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;

    const TileOrigin o = mil_tile_origin(g, blockIdx.x);
    char* ldsA = smem;
    char* ldsW = smem + a.lds_w_off;
    mil_load_halo<T, CINP>(ldsA, a.x, g, o, tid, 256);

    const int s_eff = g.zins ? 1 : g.stride;
    int pixbase[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) pixbase[m] = mil_pix_base<PIXB>(g, (wave * MTW + m) * 16 + r, s_eff);

    f32x4_t acc[MTW][NT];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int ntaps = g.ks * g.ks;
    for (int s0 = 0; s0 < a.nsteps; s0 += a.kc) {
        __syncthreads();
        const int cs = min(a.kc, a.nsteps - s0);
        mil_stage_filter(ldsW, reinterpret_cast<const char*>(a.w) + (size_t)s0 * NT * 64 * FRAGB, cs * NT * 64 * FRAGB, tid, 256);
        __syncthreads();
        for (int sl = 0; sl < cs; ++sl) {
            const int q = 4 * (s0 + sl) + gq;
            int tap = q / CG;
            int cg = q - tap * CG;
            if (tap >= ntaps) { tap = 0; cg = 0; }      // K padding: weights there are zero
            const int ky = tap / g.ks, kx = tap - ky * g.ks;
            const int toff = (ky * g.hw + kx) * PIXB + cg * T::CGB;
            Frag8<T> bf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = lds_frag<T>(ldsW + ((sl * NT + nt) * 64 + lane) * FRAGB);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const Frag8<T> af = lds_pix_frag<T, CINP * 2>(ldsA + pixbase[m] + toff);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mma8(af, bf[nt], acc[m][nt]);
            }
        }
    }
    __syncthreads();

    // accumulators (col = lane&15, row = 4*(lane>>4)+i) -> LDS [tile pixel][NT*16] fp32
    float* epi = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                epi[((wave * MTW + m) * 16 + gq * 4 + i) * (NT * 16) + nt * 16 + r] = acc[m][nt][i];
    __syncthreads();

    const int tw_mask = (1 << g.tw_log2) - 1, th_mask = (1 << g.th_log2) - 1;
    for (int idx = tid; idx < TILE_PX * NGRP; idx += 256) {
        const int tp = idx / NGRP, c8 = idx - tp * NGRP;
        const int ox = o.ox0 + (tp & tw_mask);
        const int oy = o.oy0 + ((tp >> g.tw_log2) & th_mask);
        const int img = o.img0 + (tp >> (g.tw_log2 + g.th_log2));
        if (img >= g.n_img || oy >= g.Ho || ox >= g.Wo) continue;
        float v[8];
        {
            const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(epi + tp * (NT * 16) + c8 * 8);
            const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(epi + tp * (NT * 16) + c8 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
        }
        if (a.bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += a.bias[c8 * 8 + j];
        }
        const size_t off = (((size_t)img * g.Ho + oy) * g.Wo + ox) * COUTP + c8 * 8;
        if (a.res) {
            float rv[8];
            load8<T>(a.res + off, rv);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rv[j];
        }
        if (a.apply_lrelu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = lrelu(v[j], a.slope);
        }
        if (a.act) {
            float av[8];
            load8<T>(a.act + off, av);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= lrelu_grad(av[j], a.slope);
        }
        store8<T>(a.y + off, v);
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent, software-pipelined form for the narrow (HBM-bound) layers (bf16 path).
//   * each workgroup stages the packed filter in LDS ONCE and walks a strided list of 256-px tiles;
//   * while tile i's MFMA loop runs, the loads of tile i+1's halo and of tile i's epilogue operands
//     (residual, activation mask) are already in flight into registers (issue-early / write-late);
//   * the MFMA operands are swapped (A = filter fragment, B = pixel fragment) so the accumulator holds
//     D[channel][pixel]: every lane owns 4 consecutive channels of one pixel and the whole epilogue
//     (residual, LeakyReLU, mask, bf16 store) runs from registers with 8-byte accesses — no LDS round
//     trip; the accumulators start from the bias instead of zero;
//   * everything that does not depend on the tile (piece -> halo coordinate, k-step -> tap offset,
//     lane -> output pixel) is computed once per workgroup, the tile walk runs on the scalar unit, and
//     memory goes through buffer descriptors with out-of-range offsets as the predicate (pf_common.cuh):
//     the per-tile instruction stream is loads, MFMAs and the element-wise epilogue, which is what keeps
//     the kernel on the HBM roof rather than on the vector-issue roof.
// FLAGS: -1 = epilogue options read from the arguments at run time; otherwise a bit mask fixed at compile time
// (1 residual/addend, 2 lrelu' mask, 4 LeakyReLU): the unused operand prefetch registers and branches disappear, which
// for the 24-channel layers is the difference between two and three resident waves per SIMD.
#ifndef MIL_PF_WAVES_24
#define MIL_PF_WAVES_24 2       // (3 waves per SIMD = 168 VGPRs spilled 13 in the residual / mask variants; the whole-block forward and the fused backward have taken the hot launches)
#endif
// which instantiations get the explicit one-step-ahead operand prefetch (it costs a second operand register set)
#ifndef MIL_PF_PIPE
// measured: no gain on the 64/80-channel forms (56.0/52.5 -> 56.8/50.2 us per launch); the 24/40-channel forms sit at their
// VGPR cap (3-4 waves per SIMD) and a second operand set spills -> off
#define MIL_PF_PIPE(CINP, NT, MTW, NW) false
#endif
// NW = waves per workgroup: 4, or 8 for the layers whose resident filter leaves room for only ONE workgroup per CU
// (64 channels: 72 KB of filter) — eight waves on the same LDS tiles give every SIMD a second wave to overlap with.
// T = BF16, or F32S: the same pipeline on fp32 tensors with split-precision products (MIL_DT_F32S) — the halo pieces are
// split into hi/lo bf16 planes when they are committed to LDS, every (filter, pixel) fragment pair costs three MFMAs, and
// the register epilogue loads / stores its 8 channels per lane as two 16-byte accesses.  At most two waves per SIMD there
// (the prefetch and epilogue register sets double).
template <typename T> struct Epi8;          // 8 consecutive channels of one pixel, as fetched: residual / mask operands
template <> struct Epi8<BF16> { u32x4_t v[1]; };
template <> struct Epi8<F32S> { u32x4_t v[2]; };

// NTALL > NT: OUTPUT-COLUMN SPLIT.  The workgroups of grid row blockIdx.y compute only the NT column tiles
// [blockIdx.y*NT, +NT) of the layer's NTALL and keep only that slice of the filter resident — for the filters that do not
// fit LDS whole (split precision, 64 channels: 147 KB of [hi | lo] fragments; a half is 74 KB next to a 128-pixel halo tile).
// Every grid row walks all tiles, so the input is read once per row (the second read mostly from L2 / Infinity Cache: the
// rows run side by side), the output once.
template <typename T, int CINP, int NT, int KS, int MTW, int FLAGS = -1, int NW = 4, int NTALL = NT>
// Waves per SIMD the 8-wave 24/40-channel forms are compiled for.  The forward variants that BASELINE configurations launch
// (FLAGS 4 / 5: bias + LeakyReLU, + residual) stay at 4 (128 VGPRs; the residual form spills 6 of them): measured round 4 on
// MI355X, 113 us per launch at 4 against 139 us at 3 waves per SIMD with no spill (256x256 tiles; 136 vs 179 us at 300x300) — two
// resident workgroups beat a spill-free single one.  The data-gradient variants (FLAGS -1, 2, 3: 12-27 spills at 128) are compiled
// for 3 waves per SIMD (168 VGPRs, no spill): the fused backward kernels have taken their launches at every BASELINE size.
#ifndef MIL_PF40_EU
#define MIL_PF40_EU(FLAGS) (((FLAGS) == 4 || (FLAGS) == 5) ? 4 : 3)
#endif
__global__ __launch_bounds__(64 * NW, T::SPLIT ? (NW == 8 ? 2 : (CINP <= 24 ? 2 : 1)) : (NW == 8 ? (CINP <= 40 ? MIL_PF40_EU(FLAGS) : 2) : ((MTW == 2 && CINP <= 24) ? 4 : (CINP <= 24 && FLAGS >= 0 && FLAGS != 3) ? MIL_PF_WAVES_24 : (CINP <= 40 ? 2 : 1))))
void conv_igemm_pf_kernel(ConvArgs<T> a, int ntiles, unsigned x_bytes, unsigned y_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int ESZ = T::ESZ;
    constexpr int FRAGB = 8 * ESZ;              // bytes of one packed filter fragment per lane
#ifndef MIL_PF_PIPE_X3
// split precision: the one-step-ahead operand sets where they fit the 256-VGPR budget of two waves per SIMD
// (measured with everything on: 24 channels spill 47-145 VGPRs, the 40-channel res+mask variant 64; the 40-channel 8-wave forms
// with one epilogue operand fit at 230-252, the 64-channel column-split form at 178-194)
// Only the 8-wave forms: the 4-wave 64-channel 1x1 zero-insert form (256 VGPRs + 216 AGPRs either way) went from 0.21 to
// 0.53 ms with it.
#define MIL_PF_PIPE_X3(CINP, NT, MTW, NW, FLAGS) ((NW) == 8 && ((CINP) == 64 || ((CINP) == 40 && (FLAGS) >= 0 && (FLAGS) != 3)))
#endif
    constexpr bool PIPE = T::SPLIT ? MIL_PF_PIPE_X3(CINP, NT, MTW, NW, FLAGS) : MIL_PF_PIPE(CINP, NT, MTW, NW);
    constexpr int PIXB = mil_pix_pitch(CINP, ESZ);
    constexpr int CG = CINP / 8;
    constexpr int COUTP = mil_nt_to_cp(NTALL);
    constexpr int NTHR = 64 * NW;
    constexpr int NPX = (mil_halo_px_max(NW * MTW == 16 ? 4 : 2) * (CINP * ESZ / 16) + NTHR - 1) / NTHR;
    constexpr int KSTEPS = (KS * KS * CG + 3) / 4;
    constexpr bool LAST_PARTIAL = (COUTP % 16) != 0;        // the last column tile holds only 8 channels
    static_assert(NTALL % NT == 0, "column split: equal grid rows");
    const int nt0 = NTALL == NT ? 0 : (int)blockIdx.y * NT; // first column tile of this grid row
    // the layer's LAST column tile (8 channels when COUTP % 16 != 0) is the last tile of the last grid row
    const bool last_row = NTALL == NT || nt0 + NT == NTALL;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    char* ldsA = smem;
    char* ldsW = smem + a.lds_w_off;
    const bool has_res = FLAGS < 0 ? (a.res != nullptr) : (FLAGS & 1) != 0;
    const bool has_act = FLAGS < 0 ? (a.act != nullptr) : (FLAGS & 2) != 0;
    const bool do_lrelu = FLAGS < 0 ? (a.apply_lrelu != 0) : (FLAGS & 4) != 0;

    if constexpr (NTALL == NT) {
        mil_stage_filter(ldsW, a.w, KSTEPS * NT * 64 * FRAGB, tid, NTHR);
    } else {                                 // this row's NT column tiles of every k-step: KSTEPS runs of NT*64*FRAGB bytes
        constexpr int RUN = NT * 64 * FRAGB;
        const char* src = reinterpret_cast<const char*>(a.w) + (size_t)nt0 * 64 * FRAGB;
        for (int i = tid * 16; i < KSTEPS * RUN; i += NTHR * 16) {
            const int ks = i / RUN, r_ = i - ks * RUN;
            *reinterpret_cast<u32x4_t*>(ldsW + i) = *reinterpret_cast<const u32x4_t*>(src + (size_t)ks * (NTALL * 64 * FRAGB) + r_);
        }
    }
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, x_bytes);
    const __amdgpu_buffer_rsrc_t rs_res = mil_rsrc(a.res, a.res ? y_bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_act = mil_rsrc(a.act, a.act ? y_bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_y = mil_rsrc(a.y, y_bytes);
    const int s_eff = g.zins ? 1 : g.stride;
    const int TW = 1 << g.tw_log2, TH = 1 << g.th_log2;

    // ---- tile-invariant tables ---------------------------------------------------------------------
    HaloTables<NPX> ht;
    mil_build_halo_tables<CINP, NPX, NTHR, T>(ht, g, tid);
    mil_halo_tables_use_dump<NPX>(ht, a.lds_dump_rel);                           // spare bytes behind each halo buffer
    int toff[KSTEPS];
#pragma unroll
    for (int sl = 0; sl < KSTEPS; ++sl) {
        const int q = 4 * sl + gq;
        int tap = q / CG, cg = q - tap * CG;
        if (tap >= KS * KS) { tap = 0; cg = 0; }
        const int ky = tap / KS, kx = tap - ky * KS;
        toff[sl] = (ky * g.hw + kx) * PIXB + cg * 16;
    }
    int pixbase[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int tp = (wave * MTW + m) * 16 + r;
        const int tx = tp & (TW - 1), ty = (tp >> g.tw_log2) & (TH - 1), ti = tp >> (g.tw_log2 + g.th_log2);
        pixbase[m] = ((ti * g.hh + ty * s_eff) * g.hw + tx * s_eff) * PIXB;
    }
    // Epilogue layout.  After the MFMA loop a lane holds 4 consecutive channels of pixel (m, r) for each row
    // tile m.  One v_permlane16_swap per accumulator register between row tiles 2p and 2p+1 turns that into
    // 8 consecutive channels of ONE pixel per lane — pixel (2p + (gq&1), r), channels 16*nt + 8*(gq>>1) ... —
    // so residual / mask loads and the output store are 16-byte accesses (half the VMEM instructions).
    // With ONE row tile per wave (MTW == 1: the 8-wave form on 128-pixel tiles) there is nothing to pair with: the lane
    // keeps its 4 channels of pixel r and the epilogue runs on 8-byte accesses.
    constexpr bool PAIRED = MTW >= 2;
    constexpr int NPAIR = PAIRED ? MTW / 2 : 1;
    int o_rel[NPAIR], o_pos[NPAIR];
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
        const int tp = PAIRED ? (wave * MTW + 2 * p + (gq & 1)) * 16 + r : wave * 16 + r;
        const int tx = tp & (TW - 1), ty = (tp >> g.tw_log2) & (TH - 1), ti = tp >> (g.tw_log2 + g.th_log2);
        o_rel[p] = ((ti * g.Ho + ty) * g.Wo + tx) * (COUTP * ESZ) + (PAIRED ? (gq >> 1) * 8 : gq * 4) * ESZ + nt0 * 16 * ESZ;
        o_pos[p] = (ti << 20) | (ty << 10) | tx;
    }
    // channels of the last column tile this lane owns exist
    const bool last_ok = !LAST_PARTIAL || !last_row || (PAIRED ? (gq >> 1) == 0 : gq * 4 < COUTP - (NTALL - 1) * 16);
    f32x4_t bias_r[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias_r[nt][i] = a.bias ? a.bias[(nt0 + nt) * 16 + gq * 4 + i] : 0.f;

    TileWalker cur, nxt;
    const int bid = mil_xcd_block_id();
    cur.init(g, bid, gridDim.x);
    nxt = cur; nxt.advance();
    // epilogue operand offsets + loads of one tile (16 bytes = 8 channels per lane per (row-tile pair, column tile))
    auto fetch_epi = [&](const TileOrigin& o, unsigned (&ooff)[NPAIR], Epi8<T> (&rres)[NPAIR][NT], Epi8<T> (&ract)[NPAIR][NT]) {
        const int obase = ((o.img0 * g.Ho + o.oy0) * g.Wo + o.ox0) * (COUTP * ESZ);
        const int ylim = g.Ho - o.oy0, xlim = g.Wo - o.ox0, ilim = g.n_img - o.img0;
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
            const bool ok = (o_pos[p] >> 20) < ilim && ((o_pos[p] >> 10) & 1023) < ylim && (o_pos[p] & 1023) < xlim;
            ooff[p] = ok ? (unsigned)(obase + o_rel[p]) : MIL_OOB;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned off = (LAST_PARTIAL && nt == NT - 1 && !last_ok) ? MIL_OOB : ooff[p] + nt * 16 * ESZ;
                if constexpr (PAIRED) {
                    if (has_res) rres[p][nt].v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, off, 0, 0);
                    if (has_act) ract[p][nt].v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_act, off, 0, 0);
                    if constexpr (T::SPLIT) {           // fp32: channels 4-7 of the lane's eight
                        const unsigned off2 = off == MIL_OOB ? MIL_OOB : off + 16;
                        if (has_res) rres[p][nt].v[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, off2, 0, 0);
                        if (has_act) ract[p][nt].v[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_act, off2, 0, 0);
                    }
                } else if constexpr (T::SPLIT) {        // four fp32 channels per lane
                    if (has_res) rres[p][nt].v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, off, 0, 0);
                    if (has_act) ract[p][nt].v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_act, off, 0, 0);
                } else {
                    if (has_res) { const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rs_res, off, 0, 0); rres[p][nt].v[0][0] = t[0]; rres[p][nt].v[0][1] = t[1]; }
                    if (has_act) { const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rs_act, off, 0, 0); ract[p][nt].v[0][0] = t[0]; ract[p][nt].v[0][1] = t[1]; }
                }
            }
        }
    };

    // EPI_AHEAD: request the epilogue operands a whole tile ahead (costs one more register set; the 40/64-channel
    // instantiations have no room for it and request them at the start of their own tile instead).
    // DEPTH: how many tiles ahead the halo loads run.  A load round trip under load is ~2 us whether it hits
    // L2 or HBM, about one tile time, so the narrowest layers keep TWO tiles of halo loads in flight (two
    // register sets, tile loop unrolled by two).
    constexpr bool EPI_AHEAD = CINP <= 24 && !T::SPLIT;
#ifndef MIL_PF_DEPTH40
#define MIL_PF_DEPTH40 1
#endif
    // measured twice: a second tile of halo loads in flight is 8% slower on the 24-channel layers and no faster on the
    // 40-channel ones (108 -> 107 us plain, 142 -> 230 us with both epilogue operands: the second register set spills)
#ifndef MIL_PF_DEPTH80
#define MIL_PF_DEPTH80 1
#endif
    constexpr int DEPTH = (CINP == 40 && NW == 4) ? MIL_PF_DEPTH40 : (CINP == 80 ? MIL_PF_DEPTH80 : 1);
    const int G = gridDim.x;
    TileWalker nx2 = nxt;
    nx2.advance();
    u32x4_t rxA[NPX], rxB[DEPTH == 2 ? NPX : 1];
    // EPI_AHEAD: a second register set, filled a whole tile ahead.  Otherwise ONE set, refilled for the next tile as soon
    // as this tile's epilogue has consumed it (the loads then fly under the next tile's barriers and MFMA loop instead of
    // being issued at its start and waited for right after its MFMA loop).
    unsigned ooff_n[NPAIR];
    Epi8<T> rres_n[NPAIR][NT], ract_n[NPAIR][NT];
    if (bid < ntiles) {
        mil_fetch_halo<CINP, NPX, T>(rxA, rs_x, ht, g, cur.origin(g));
        fetch_epi(cur.origin(g), ooff_n, rres_n, ract_n);
    }
    if constexpr (DEPTH == 2) {
        if (bid + G < ntiles) mil_fetch_halo<CINP, NPX, T>(rxB, rs_x, ht, g, nxt.origin(g));
    }

    // Two halo buffers (when LDS allows): the next tile's halo is committed into the buffer the tile BEFORE the current
    // one used, which every wave has left by the time this wave is past the current tile's only barrier — so the
    // "everyone has finished reading" barrier of the single-buffer form disappears (one barrier per tile, not two).
    const int buf_step = a.lds_a2_off;          // 0: single buffer
    int buf = 0;
    auto do_tile = [&](u32x4_t (&rx)[NPX], int tile) {
        if (buf_step == 0) __syncthreads();    // single buffer: every wave has finished reading ldsA for the previous tile
        char* ldsA_t = ldsA + buf;
        buf = buf_step - buf;
        mil_commit_halo_all<NPX, T, CINP>(rx, ldsA_t, ht);
        // this tile's epilogue operands were requested one tile ago; take them over before re-issuing
        unsigned ooff_l[NPAIR];
        Epi8<T> rres_l[NPAIR][NT], ract_l[NPAIR][NT];
        if constexpr (EPI_AHEAD) {
#pragma unroll
            for (int p = 0; p < NPAIR; ++p) {
                ooff_l[p] = ooff_n[p];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { rres_l[p][nt] = rres_n[p][nt]; ract_l[p][nt] = ract_n[p][nt]; }
            }
        }
        auto& ooff = *(EPI_AHEAD ? &ooff_l : &ooff_n);
        auto& rres = *(EPI_AHEAD ? &rres_l : &rres_n);
        auto& ract = *(EPI_AHEAD ? &ract_l : &ract_n);
        __syncthreads();
        // issue-early: refill the register set just written to LDS with the halo of the tile DEPTH ahead
        if (tile + DEPTH * G < ntiles) mil_fetch_halo<CINP, NPX, T>(rx, rs_x, ht, g, (DEPTH == 2 ? nx2 : nxt).origin(g));
        if constexpr (EPI_AHEAD) {
            if (tile + G < ntiles) fetch_epi(nxt.origin(g), ooff_n, rres_n, ract_n);
        }
        cur = nxt; nxt = nx2; nx2.advance();

        f32x4_t acc[MTW][NT];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = bias_r[nt];
        if constexpr (PIPE) {
            // one k-step ahead: the fragment reads of step sl+1 are issued before the MFMAs of step sl and scheduling fences
            // keep that order (left alone, hipcc issues every read right in front of its MFMAs behind an lgkmcnt(0))
            Frag8<T> wc[NT], xc[MTW], wn[NT], xn[MTW];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wc[nt] = lds_frag<T>(ldsW + (nt * 64 + lane) * FRAGB);
#pragma unroll
            for (int m = 0; m < MTW; ++m) xc[m] = lds_pix_frag<T, CINP * 2>(ldsA_t + pixbase[m] + toff[0]);
#pragma unroll
            for (int sl = 0; sl < KSTEPS; ++sl) {
                if (sl + 1 < KSTEPS) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) wn[nt] = lds_frag<T>(ldsW + (((sl + 1) * NT + nt) * 64 + lane) * FRAGB);
#pragma unroll
                    for (int m = 0; m < MTW; ++m) xn[m] = lds_pix_frag<T, CINP * 2>(ldsA_t + pixbase[m] + toff[sl + 1 < KSTEPS ? sl + 1 : sl]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mma8(wc[nt], xc[m], acc[m][nt]);   // D[channel][pixel]
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wc[nt] = wn[nt];
#pragma unroll
                for (int m = 0; m < MTW; ++m) xc[m] = xn[m];
            }
        } else {
#pragma unroll
        for (int sl = 0; sl < KSTEPS; ++sl) {
            Frag8<T> wf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = lds_frag<T>(ldsW + ((sl * NT + nt) * 64 + lane) * FRAGB);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const Frag8<T> xf = lds_pix_frag<T, CINP * 2>(ldsA_t + pixbase[m] + toff[sl]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mma8(wf[nt], xf, acc[m][nt]);   // D[channel][pixel]
            }
        }

        }

        // register epilogue on 8 channels per lane (see "Epilogue layout" above); 4 channels per lane when unpaired
        if constexpr (!PAIRED) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[0][nt][i];
                auto four = [](const Epi8<T>& e, float (&f)[4]) {
                    if constexpr (T::SPLIT) {
                        const f32x4_t t = __builtin_bit_cast(f32x4_t, e.v[0]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) f[i] = t[i];
                    } else {
                        const bf16x4_t t = __builtin_bit_cast(bf16x4_t, u32x2_t{e.v[0][0], e.v[0][1]});
#pragma unroll
                        for (int i = 0; i < 4; ++i) f[i] = (float)t[i];
                    }
                };
                if (has_res) {
                    float rv[4];
                    four(rres[0][nt], rv);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += rv[i];
                }
                if (do_lrelu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], v[i] * a.slope);
                }
                if (has_act) {
                    float av[4];
                    four(ract[0][nt], av);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] *= (av[i] > 0.f ? 1.f : a.slope);
                }
                const unsigned off = (LAST_PARTIAL && nt == NT - 1 && !last_ok) ? MIL_OOB : ooff[0] + nt * 16 * ESZ;
                if constexpr (T::SPLIT) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[0], v[1], v[2], v[3]}), rs_y, off, 0, 0);
                } else {
                    bf16x4_t ov;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ov[i] = (__bf16)v[i];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, ov), rs_y, off, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
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
                auto eight = [](const Epi8<T>& e, float (&f)[8]) {
                    if constexpr (T::SPLIT) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x4_t t = __builtin_bit_cast(f32x4_t, e.v[h]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) f[4 * h + i] = t[i];
                        }
                    } else {
                        const bf16x8_t t = __builtin_bit_cast(bf16x8_t, e.v[0]);
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] = (float)t[i];
                    }
                };
                if (has_res) {
                    float rv[8];
                    eight(rres[p][nt], rv);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += rv[i];
                }
                if (do_lrelu) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], v[i] * a.slope);      // 0 < slope < 1
                }
                if (has_act) {
                    float av[8];
                    eight(ract[p][nt], av);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= (av[i] > 0.f ? 1.f : a.slope);
                }
                const unsigned off = (LAST_PARTIAL && nt == NT - 1 && !last_ok) ? MIL_OOB : ooff[p] + nt * 16 * ESZ;
                if constexpr (T::SPLIT) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[0], v[1], v[2], v[3]}), rs_y, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[4], v[5], v[6], v[7]}), rs_y, off == MIL_OOB ? MIL_OOB : off + 16, 0, 0);
                } else {
                    bf16x8_t ov;
#pragma unroll
                    for (int i = 0; i < 8; ++i) ov[i] = (__bf16)v[i];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ov), rs_y, off, 0, 0);
                }
            }
        }
        }
        if constexpr (!EPI_AHEAD) {            // `cur` is the next tile by now
            if (tile + G < ntiles) fetch_epi(cur.origin(g), ooff_n, rres_n, ract_n);
        }
    };

    for (int tile = bid; tile < ntiles;) {
        do_tile(rxA, tile);
        tile += G;
        if constexpr (DEPTH == 2) {
            if (tile >= ntiles) break;
            do_tile(rxB, tile);
            tile += G;
        }
    }
}

template <typename T, int CINP, int NT, int MTW>
static int launch_conv(const ConvArgs<T>& a0, hipStream_t stream) {
    ConvArgs<T> a = a0;
    constexpr int ESZ = T::ESZ;
    constexpr int PIXB = mil_pix_pitch(CINP, ESZ);
    constexpr int FRAGB = 8 * ESZ;
    constexpr int TILE_PX = 64 * MTW;
    mil_geom_tiles(a.g, MTW == 4 ? 8 : 6);
    const int a_bytes = (((a.g.hh * a.g.hw) << a.g.ti_log2) * PIXB + 15) & ~15;
    const int step_bytes = NT * 64 * FRAGB;
    const int epi_bytes = TILE_PX * NT * 16 * 4;
    const int budget = 160 * 1024 - a_bytes;
    if (budget < step_bytes) return MIL_ERR_UNSUPPORTED;
    int kc = a.nsteps;
    if (kc * step_bytes > budget) kc = budget / step_bytes;
    a.kc = kc;
    a.lds_w_off = a_bytes;
    int lds = a_bytes + kc * step_bytes;
    if (lds < epi_bytes) lds = epi_bytes;
    auto kern = conv_igemm_kernel<T, CINP, NT, MTW>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MIL_ERR_LAUNCH;
    }
    const int grid = a.g.n_groups * a.g.tiles_y * a.g.tiles_x;
    if (grid <= 0) return MIL_OK;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}


// Persistent launches need enough tiles to amortise the per-workgroup filter load; MIL_PF_MIN_TILES overrides the
// threshold (the parity tests set it to 1 so that small cases run the persistent kernels too).
#include <cstdlib>
static int mil_pf_min_tiles() {
    const char* e = getenv("MIL_PF_MIN_TILES");
    return e ? atoi(e) : 512;
}

static int mil_pf_rounds() {
    static const int r = [] { const char* e = mil_ab_env("MIL_PF_ROUNDS"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
    return r;
}

static bool mil_pf_double_buffer() {
    static const bool v = [] { const char* e = mil_ab_env("MIL_PF_DBUF"); return !(e && atoi(e) == 0); }();
    return v;
}

static int mil_pf_waves64() {
    static const int v = [] { const char* e = mil_ab_env("MIL_PF_WAVES64"); return (e && atoi(e) == 4) ? 4 : 8; }();
    return v;
}

#ifndef MIL_PF_WG_PER_CU
#define MIL_PF_WG_PER_CU 4
#endif
#ifndef MIL_PF_MTW_24
#define MIL_PF_MTW_24 4
#endif
template <typename T, int CINP, int NT, int KS, int MTW, int FLAGS, int NW = 4, int NTALL = NT>
static auto conv_pf_variant() { return conv_igemm_pf_kernel<T, CINP, NT, KS, MTW, FLAGS, NW, NTALL>; }

template <typename T, int CINP, int NT, int KS, int MTW = 4, int NW = 4, int NTALL = NT>
static int launch_conv_pf_ks(const ConvArgs<T>& a0, hipStream_t stream, bool* taken) {
    ConvArgs<T> a = a0;
    constexpr int ESZ = T::ESZ;
    constexpr int PIXB = mil_pix_pitch(CINP, ESZ);
    constexpr int COUTP = mil_nt_to_cp(NTALL);
    constexpr int DUMPB = T::SPLIT ? CINP * 2 + 16 : 16;        // spare bytes behind a halo buffer for the branch-free commit (split: hi + lo)
    *taken = false;
    mil_geom_tiles(a.g, NW * MTW == 16 ? 8 : 7);
    const int halo_px = (a.g.hh * a.g.hw) << a.g.ti_log2;
    if (halo_px > mil_halo_px_max(NW * MTW == 16 ? 4 : 2)) return MIL_OK;
    const int a_bytes = ((halo_px * PIXB + 15) & ~15) + DUMPB;  // + dump slot for the branch-free halo commit
    const int w_bytes = a.nsteps * NT * 64 * 8 * ESZ;
    // a second halo buffer (one barrier per tile instead of two) when it does not cost a resident workgroup
    const bool dbuf = mil_pf_double_buffer() && (160 * 1024) / (2 * a_bytes + w_bytes) >= ((160 * 1024) / (a_bytes + w_bytes) > 2 ? 3 : (160 * 1024) / (a_bytes + w_bytes));
    const int lds = (dbuf ? 2 : 1) * a_bytes + w_bytes;
    if (lds > 160 * 1024) return MIL_OK;
    if (a.g.hh >= 1024 || a.g.hw >= 1024 || a.slope < 0.f || a.slope >= 1.f) return MIL_OK;   // max(v, slope*v) form
    if ((a.g.n_groups * a.g.tiles_y * a.g.tiles_x) < mil_pf_min_tiles()) return MIL_OK;    // too few tiles for a persistent launch
    a.kc = a.nsteps;
    a.lds_w_off = (dbuf ? 2 : 1) * a_bytes;
    a.lds_a2_off = dbuf ? a_bytes : 0;
    a.lds_dump_rel = a_bytes - DUMPB;
    auto kern = conv_igemm_pf_kernel<T, CINP, NT, KS, MTW, -1, NW, NTALL>;
    // the hot square 3x3 layers get the epilogue options as compile-time constants
    if constexpr (KS == 3 && ((CINP == 24 && NTALL == 2) || (CINP == 40 && NTALL == 3) || (CINP == 64 && NTALL == 4) || (CINP == 80 && NTALL == 5))) {
        const int fl = (a.res ? 1 : 0) | (a.act ? 2 : 0) | (a.apply_lrelu ? 4 : 0);
        if (fl == 4) kern = conv_pf_variant<T, CINP, NT, KS, MTW, 4, NW, NTALL>();
        else if (fl == 5) kern = conv_pf_variant<T, CINP, NT, KS, MTW, 5, NW, NTALL>();
        else if (fl == 2) kern = conv_pf_variant<T, CINP, NT, KS, MTW, 2, NW, NTALL>();
        else if (fl == 3) kern = conv_pf_variant<T, CINP, NT, KS, MTW, 3, NW, NTALL>();
    }
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MIL_ERR_LAUNCH;
    }
    const int per_cu = mil_resident_per_cu(kern, lds, MIL_PF_WG_PER_CU, 64 * NW) * mil_pf_rounds();   // rounds of resident workgroups
    // buffer descriptors address < 2 GiB: split the launch by images when a tensor is larger
    const size_t x_img = (size_t)a.g.H * a.g.W * CINP * ESZ, y_img = (size_t)a.g.Ho * a.g.Wo * COUTP * ESZ;
    int chunk = mil_imgs_under_2g(x_img > y_img ? x_img : y_img);
    if (chunk >= 16) chunk &= ~15;                       // keep image groups (<= 16 images per tile) intact
    const int n_total = a0.g.n_img;
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        const int n = (n_total - i0 < chunk) ? n_total - i0 : chunk;
        ConvArgs<T> c = a;
        c.g.n_img = n;
        c.g.n_groups = (n + (1 << c.g.ti_log2) - 1) >> c.g.ti_log2;
        c.x = a.x + (size_t)i0 * (x_img / ESZ);
        c.y = a.y + (size_t)i0 * (y_img / ESZ);
        if (a.res) c.res = a.res + (size_t)i0 * (y_img / ESZ);
        if (a.act) c.act = a.act + (size_t)i0 * (y_img / ESZ);
        const int ntiles = c.g.n_groups * c.g.tiles_y * c.g.tiles_x;
        int grid = mil_num_cus() * per_cu / (NTALL / NT);      // the column rows share the resident set
        if (grid < 1) grid = 1;
        if (grid > ntiles) grid = ntiles;
        hipLaunchKernelGGL(kern, dim3(grid, NTALL / NT), dim3(64 * NW), lds, stream, c, ntiles, (unsigned)(x_img * n), (unsigned)(y_img * n));
        MIL_CHECK_LAUNCH();
    }
    *taken = true;
    return MIL_OK;
}

template <typename T, int CINP, int NT>
static int launch_conv_pf(const ConvArgs<T>& a, hipStream_t stream, bool* taken) {
    *taken = false;
    if constexpr (CINP == 16 && NT == 2) { if (a.g.ks == 4) return launch_conv_pf_ks<T, 16, 2, 4>(a, stream, taken); }
    if constexpr (CINP == 24 && NT == 2) { if (a.g.ks == 3) return launch_conv_pf_ks<T, 24, 2, 3, MIL_PF_MTW_24>(a, stream, taken); }
#ifndef MIL_PF40_WAVES
#define MIL_PF40_WAVES 8        // measured in the model: 122 / 100 us per launch with 4 waves, 114 / 88 us with 8 (four waves per SIMD)
#endif
    if constexpr (CINP == 40 && NT == 3) {
#if MIL_PF40_WAVES == 8
        if (a.g.ks == 3) return launch_conv_pf_ks<T, 40, 3, 3, 2, 8>(a, stream, taken);
#else
        if (a.g.ks == 3) return launch_conv_pf_ks<T, 40, 3, 3>(a, stream, taken);
#endif
    }
    if constexpr (CINP == 40 && NT == 2) {
        if (a.g.ks == 3) return launch_conv_pf_ks<T, 40, 2, 3>(a, stream, taken);
        if (a.g.ks == 1) return launch_conv_pf_ks<T, 40, 2, 1>(a, stream, taken);
    }
    if constexpr (CINP == 64 && NT == 4) {
        // split precision: the 147 KB filter does not fit — two grid rows of 32 output channels each on 128-pixel tiles
        if constexpr (T::SPLIT) { if (a.g.ks == 3) return launch_conv_pf_ks<T, 64, 2, 3, 1, 8, 4>(a, stream, taken); }
        else
        if (a.g.ks == 3) return mil_pf_waves64() == 8 ? launch_conv_pf_ks<T, 64, 4, 3, 2, 8>(a, stream, taken) : launch_conv_pf_ks<T, 64, 4, 3>(a, stream, taken);
    }
    if constexpr (CINP == 64 && NT == 3) {
        // split precision (the 64 -> 40 channel stage-entry data gradient, zero-insert): three grid rows of one column tile
        if constexpr (T::SPLIT) { if (a.g.ks == 3) return launch_conv_pf_ks<T, 64, 1, 3, 2, 8, 3>(a, stream, taken); }
        else
        if (a.g.ks == 3) return launch_conv_pf_ks<T, 64, 3, 3>(a, stream, taken);
        if (a.g.ks == 1) return launch_conv_pf_ks<T, 64, 3, 1>(a, stream, taken);
    }
    // 80-channel layers: the whole filter (115 KB) stays resident, so the tile shrinks to 128 px
    // (eight waves with one row tile each by default: one workgroup per CU either way, but two waves per SIMD)
    if constexpr (CINP == 80 && NT == 5) {
        // (split precision: five grid rows of one 16-channel column tile each — 46 KB of [hi | lo] fragments per row on
        // 128-pixel tiles — measured 124 us per conv against 103 us on the K-chunked generic kernel: not dispatched)
        if constexpr (T::SPLIT) { return MIL_OK; }
        else
        if (a.g.ks == 3) return mil_pf_waves64() == 8 ? launch_conv_pf_ks<T, 80, 5, 3, 1, 8>(a, stream, taken) : launch_conv_pf_ks<T, 80, 5, 3, 2>(a, stream, taken);
    }
    if constexpr (CINP == 80 && NT == 4) {
        if (a.g.ks == 3) return mil_pf_waves64() == 8 ? launch_conv_pf_ks<T, 80, 4, 3, 1, 8>(a, stream, taken) : launch_conv_pf_ks<T, 80, 4, 3, 2>(a, stream, taken);
        if (a.g.ks == 1) return launch_conv_pf_ks<T, 80, 4, 1, 2>(a, stream, taken);
    }
    return MIL_OK;
}

// 256-px tiles (4 MFMA row tiles per wave) when the halo + a weight chunk fit in LDS, else 64-px tiles.
template <typename T, int CINP, int NT>
static int launch_conv_auto(const ConvArgs<T>& a, bool small_tile, hipStream_t stream) {
    if constexpr (T::DT != MIL_DT_F32) {        // bf16 and split-precision fp32: the persistent prefetch-pipelined kernel when it fits
        if (!small_tile) {
            bool taken = false;
            const int rc = launch_conv_pf<T, CINP, NT>(a, stream, &taken);
            if (rc != MIL_OK || taken) return rc;
        }
    }
    if (!small_tile) {
        const int rc = launch_conv<T, CINP, NT, 4>(a, stream);
        if (rc != MIL_ERR_UNSUPPORTED) return rc;
    }
    return launch_conv<T, CINP, NT, 1>(a, stream);
}

#include "conv_stream_x3.cuh"

template <typename T>
static int dispatch_conv(const ConvArgs<T>& a, int cin_p, int cout_p, hipStream_t stream) {
    const bool small_tile = (a.g.stride == 2 && !a.g.zins);     // stride-2 forward: the halo is 4x the tile
#define MIL_CONV_CASE(CI, NTV) return launch_conv_auto<T, CI, NTV>(a, small_tile, stream)
    if (cin_p == 16 && cout_p == 24) MIL_CONV_CASE(16, 2);
    if (cin_p == 16 && cout_p == 64) MIL_CONV_CASE(16, 4);   // alt_resnet stem (3 -> 64)
    if (cin_p == 24 && cout_p == 24) MIL_CONV_CASE(24, 2);
    if (cin_p == 40 && cout_p == 40) MIL_CONV_CASE(40, 3);
    if (cin_p == 64 && cout_p == 64) MIL_CONV_CASE(64, 4);
    if (cin_p == 80 && cout_p == 80) MIL_CONV_CASE(80, 5);
    if (cin_p == 40 && cout_p == 24) MIL_CONV_CASE(40, 2);   // dgrad of stage-entry convs
    if (cin_p == 64 && cout_p == 40) MIL_CONV_CASE(64, 3);
    if (cin_p == 80 && cout_p == 64) MIL_CONV_CASE(80, 4);
    if (cin_p == 24 && cout_p == 40) MIL_CONV_CASE(24, 3);   // stage-entry convs / projections
    if (cin_p == 40 && cout_p == 64) MIL_CONV_CASE(40, 4);
    if (cin_p == 64 && cout_p == 80) MIL_CONV_CASE(64, 5);
#undef MIL_CONV_CASE
    return MIL_ERR_UNSUPPORTED;
}

// pixel-resident kernels for the small maps of the last two stages (conv_resident.hip); MIL_ERR_UNSUPPORTED for every other shape
int mil_resident_conv(const void* x, const void* wpack, const float* bias_pad, const void* res, const void* act, void* y, int n_img,
                      int H, int W, int cp, int apply_lrelu, float slope, hipStream_t st);
int mil_resident_conv_x3(const void* x, const void* wpack, const float* bias_pad, const void* res, const void* act, void* y, int n_img,
                         int H, int W, int cp, int apply_lrelu, float slope, hipStream_t st);

extern "C" int mil_conv_igemm(const void* x, const void* wpack, const float* bias_pad, const void* res,
                              const void* act, void* y, int n_img, int H, int W, int cin_p, int Ho, int Wo,
                              int cout_p, int ks, int stride, int pad, int zero_insert, int apply_lrelu,
                              float slope, int dtype, void* stream) {
    if (!x || !wpack || !y || n_img < 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return MIL_ERR_ARG;
    if (!(ks == 1 || ks == 3 || ks == 4) || !(stride == 1 || stride == 2)) return MIL_ERR_ARG;
    ConvGeom g{};
    g.n_img = n_img; g.H = H; g.W = W; g.Ho = Ho; g.Wo = Wo; g.ks = ks; g.stride = stride; g.pad = pad;
    g.zins = zero_insert ? 1 : 0;
    const int nsteps = (ks * ks * (cin_p / 8) + 3) / 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIL_DT_BF16) {
        if (cin_p == cout_p && ks == 3 && stride == 1 && pad == 1 && !zero_insert && Ho == H && Wo == W) {
            const int rc = mil_resident_conv(x, wpack, bias_pad, res, act, y, n_img, H, W, cin_p, apply_lrelu, slope, st);
            if (rc != MIL_ERR_UNSUPPORTED) return rc;
        }
        ConvArgs<BF16> a{};
        a.x = (const __bf16*)x; a.w = (const __bf16*)wpack; a.bias = bias_pad; a.res = (const __bf16*)res;
        a.act = (const __bf16*)act; a.y = (__bf16*)y; a.g = g; a.nsteps = nsteps; a.apply_lrelu = apply_lrelu; a.slope = slope;
        return dispatch_conv<BF16>(a, cin_p, cout_p, st);
    } else if (dtype == MIL_DT_F32) {
        ConvArgs<F32> a{};
        a.x = (const float*)x; a.w = (const float*)wpack; a.bias = bias_pad; a.res = (const float*)res;
        a.act = (const float*)act; a.y = (float*)y; a.g = g; a.nsteps = nsteps; a.apply_lrelu = apply_lrelu; a.slope = slope;
        return dispatch_conv<F32>(a, cin_p, cout_p, st);
    } else if (dtype == MIL_DT_F32S) {
        if (cin_p == cout_p && ks == 3 && stride == 1 && pad == 1 && !zero_insert && Ho == H && Wo == W) {
            const int rc = mil_resident_conv_x3(x, wpack, bias_pad, res, act, y, n_img, H, W, cin_p, apply_lrelu, slope, st);
            if (rc != MIL_ERR_UNSUPPORTED) return rc;
        }
        // 40 -> 40 channels on maps of 16 pixels and more (layer 2 at every BASELINE tile size): the filter-streaming kernel
        if (cin_p == 40 && cout_p == 40 && ks == 3 && stride == 1 && pad == 1 && !zero_insert && Ho == H && Wo == W && H >= 16 && W >= 16 &&
            H < 1024 && W < 1024 && slope >= 0.f && slope < 1.f && n_img > 0 &&
            n_img * ((H + 15) >> 4) * ((W + 15) >> 4) >= mil_pf_min_tiles()) {
            StreamX3Args s{};
            s.x = (const float*)x; s.w = (const char*)wpack; s.bias = bias_pad; s.res = (const float*)res; s.act = (const float*)act;
            s.y = (float*)y; s.lrelu = apply_lrelu; s.slope = slope;
            s.g = g;
            return launch_stream_x3<40, 3>(s, st);
        }
        ConvArgs<F32S> a{};
        a.x = (const float*)x; a.w = (const float*)wpack; a.bias = bias_pad; a.res = (const float*)res;
        a.act = (const float*)act; a.y = (float*)y; a.g = g; a.nsteps = nsteps; a.apply_lrelu = apply_lrelu; a.slope = slope;
        return dispatch_conv<F32S>(a, cin_p, cout_p, st);
    }
    return MIL_ERR_ARG;
}
