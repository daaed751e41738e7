// Fused backward of one 3x3 stride-1 convolution of a residual block (bf16 path, narrow layers):
// ONE pass over dz and x produces both gradients that the reference's autograd computes for
// nnBlocks.py:169-171 —
//     dx = ( conv^T(dz, W) + addend ) * lrelu'(x)            (data gradient, fused mask/residual-grad)
//     dW[co][ci][ky][kx] = sum_p x[p+(ky,kx)-1][ci] * dz[p][co],   db[co] = sum_p dz[p][co]
// instead of a dgrad kernel (reads dz, x-as-mask) plus a wgrad kernel (reads x and dz again).
//
// Structure = the persistent prefetch-pipelined conv kernel (conv_igemm.hip) run over dz with the
// transposed+flipped packed filter, plus a second MFMA loop on the SAME LDS tiles:
//   dW'[(tap',co)][ci] = sum_q dz[q (+) tap'][co] * x[q][ci]        (q = the tile's 256 centre pixels)
// i.e. the weight gradient written with the halo on dz instead of on x (tap' is the flipped tap), so the
// dz halo tile already staged for the data gradient is its A operand and the x centre tile (needed
// anyway for the mask) is its B operand, both read with ds_read_b64_tr_b16.  Per-workgroup partial sums
// stay in registers across tiles and leave as one fp32 slab; a fixed-order reduction finishes them.
#include "pf_common.cuh"
#include "reduce.cuh"
#include <cstdlib>

struct BwdFusedArgs {
    const __bf16* dz;       // [n,H,W,CZ]
    const __bf16* w;        // dgrad-packed filter [ksteps][NTX][64][8]
    const __bf16* x;        // [n,H,W,CX]  conv input (mask source + wgrad operand)
    const __bf16* addend;   // [n,H,W,CX] or null
    __bf16* dx;             // [n,H,W,CX]
    float* slab;            // [gridDim.x][(MT+1)*16][NTX*16]
    ConvGeom g;
    int ntiles;
    int lds_w_off, lds_x_off, lds_dump_off, lds_a2_off;     // lds_a2_off: second dz-halo buffer (0 = none)
    int lds_x2_off;         // fused16 kernel: distance to the second x-tile buffer
    int apply_mask;
    float slope;
    unsigned z_bytes, x_bytes;      // byte sizes of dz and of x/addend/dx (buffer descriptors)
    unsigned g_bytes;               // fused16 kernel: byte size of addend / dx (z_bytes there is dz at the gradient stride)
    int gpx;                        // fused16 kernels: bytes per pixel of dz / addend / dx: 48 (padded) or 40 (MIL_DT_BF16_DGRAD); split precision: 96 or 80
    int xpx;                        // split-precision fused16 kernel: bytes per pixel of x (96)
    unsigned long long* stamp;      // MIL_STAMP diagnostic build only: [grid][NW][8] phase cycle sums (else null)
};

// Diagnostic build (-DMIL_STAMP): per-wave s_memtime sums of the five phases of a tile, written behind the slab.
// The stamps drain the LDS queue (lgkmcnt(0)) and pin the schedule: read the SHARES, never the run time.
#ifdef MIL_STAMP
#define MIL_ST_DECL unsigned long long st_prev = 0, st_sum[6] = {0, 0, 0, 0, 0, 0}; int st_tiles = 0; const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime();
#define MIL_ST_T(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define MIL_ST_BEGIN() MIL_ST_T(st_prev)
#define MIL_ST_MARK(i) { unsigned long long t_; MIL_ST_T(t_); st_sum[i] += t_ - st_prev; st_prev = t_; }
#else
#define MIL_ST_DECL
#define MIL_ST_BEGIN()
#define MIL_ST_MARK(i)
#endif

#ifndef MIL_BWD_WAVES
#define MIL_BWD_WAVES 2
#endif
// NW = waves per workgroup: 4 for the 24-channel layers (two workgroups per CU), 8 for the 40/64-channel layers, whose
// filter + tiles leave one workgroup per CU: eight waves share its LDS, every per-wave quantity halves (two row tiles of
// data-gradient accumulators, three to five of weight-gradient ones) and each SIMD still holds two waves.
#ifndef MIL_BWD_PIPE_MAXC
#define MIL_BWD_PIPE_MAXC 40      // explicit one-step-ahead operand prefetch for layers up to this many channels (the 64-channel
#endif                            // instantiation already sits at 256 VGPRs: the second operand set would spill)
// T = BF16, or F32S (round 3: MIL_DT_F32S — fp32 dz / x / addend / dx, bf16x3 split products).  The pointers of BwdFusedArgs
// are then float tensors behind their __bf16 type; the dz halo and the x centre tile hold [hi | lo] planes per pixel record,
// both MFMA loops take three products per fragment pair, the epilogue adds / masks / stores fp32.  24 channels only (one
// 8-wave workgroup per CU on 94 KB of LDS): the 40-channel filter, halo and x tile do not fit together.
template <typename T, int CZ, int NTX, int KS, bool ADD, bool MASK, int NW = 4>
#ifndef MIL_BWD24_EU
#define MIL_BWD24_EU 3          // waves per SIMD the 8-wave 24-channel generic form is compiled for: 3 = 168 VGPRs, no spill (at 4 = 128
                                // it spilled 18-26; the 16x16-tile kernel below takes the BASELINE sizes and the 300x300 driver size)
#endif
__global__ __launch_bounds__(64 * NW, T::SPLIT ? 2 : ((NW == 8 && CZ <= 24) ? MIL_BWD24_EU : MIL_BWD_WAVES)) void conv_bwd_fused_kernel(BwdFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int ESZ = T::ESZ, FRAGB = 8 * ESZ;
    constexpr int NE = T::SPLIT ? 2 : 1;                  // 16-byte registers per 8 channels of an epilogue operand
#ifndef MIL_BWD_X3_PIPE
#define MIL_BWD_X3_PIPE 1         // split precision: the same one-step-ahead operand sets (hi and lo planes)
#endif
    constexpr bool PIPE = (T::SPLIT ? MIL_BWD_X3_PIPE != 0 : true) && CZ <= MIL_BWD_PIPE_MAXC;
    constexpr int NL2 = T::SPLIT ? 2 : 1;                 // operand planes of the weight-gradient loop: hi (+ lo)
    constexpr int PIXB = mil_pix_pitch(CZ, ESZ);          // dz halo pixel pitch
    constexpr int CG = CZ / 8;
    constexpr int CX = mil_nt_to_cp(NTX);
    constexpr int PIXX = mil_pix_pitch(CX, ESZ);          // x centre tile pixel pitch
    constexpr int NTHR = 64 * NW;
    constexpr int MTW = 16 / NW;                          // data-gradient row tiles per wave (256-pixel tile)
    constexpr int NPX = (400 * (CZ * ESZ / 16) + NTHR - 1) / NTHR;
    constexpr int KSTEPS = (KS * KS * CG + 3) / 4;
    constexpr int RG = KS * KS * CG;                       // wgrad row groups (tap', 8 dz channels)
    constexpr int MT = (RG + 1) / 2;
    constexpr int MW = (MT + NW - 1) / NW;
    constexpr int CTAP = (KS * KS) / 2;                    // centre tap
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    // wave id through readfirstlane: provably wave-uniform, so branches on it are scalar branches (an MFMA or a
    // ds_read_b64_tr_b16 inside an EXEC-masked region would still execute / need all lanes)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    char* ldsA = smem;
    char* ldsW = smem + a.lds_w_off;
    char* ldsX = smem + a.lds_x_off;

    mil_stage_filter(ldsW, a.w, KSTEPS * NTX * 64 * FRAGB, tid, NTHR);
    const int TW = 1 << g.tw_log2, TH = 1 << g.th_log2;

    const __amdgpu_buffer_rsrc_t rs_z = mil_rsrc(a.dz, a.z_bytes);
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rs_add = mil_rsrc(a.addend, a.addend ? a.x_bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_dx = mil_rsrc(a.dx, a.x_bytes);

    // ---- tile-invariant tables (see conv_igemm_pf_kernel) -------------------------------------------
    constexpr int DUMPB = T::SPLIT ? CZ * 2 + 16 : 16;    // spare bytes behind each halo buffer (split: a hi and a lo piece)
    HaloTables<NPX> ht;
    mil_build_halo_tables<CZ, NPX, NTHR, T>(ht, g, tid);
    mil_halo_tables_use_dump<NPX>(ht, a.lds_w_off - DUMPB - a.lds_a2_off);
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
        pixbase[m] = ((ti * g.hh + ty) * g.hw + tx) * PIXB;
    }
    // paired epilogue layout (see conv_igemm_pf_kernel): after one v_permlane16_swap per accumulator register a
    // lane holds 8 consecutive channels of pixel (2p + (gq&1), r), channels 16*nt + 8*(gq>>1) ...
    constexpr int NPAIR = MTW / 2;
    int o_rel[NPAIR], o_pos[NPAIR], x_lds[NPAIR];
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
        const int tp = (wave * MTW + 2 * p + (gq & 1)) * 16 + r;
        const int tx = tp & (TW - 1), ty = (tp >> g.tw_log2) & (TH - 1), ti = tp >> (g.tw_log2 + g.th_log2);
        o_rel[p] = ((ti * g.Ho + ty) * g.Wo + tx) * (CX * ESZ) + (gq >> 1) * 8 * ESZ;
        o_pos[p] = (ti << 20) | (ty << 10) | tx;
        x_lds[p] = tp * PIXX + (gq >> 1) * 16;
    }
    constexpr bool LAST_PARTIAL = (CX % 16) != 0;
    const bool last_ok = !LAST_PARTIAL || (gq >> 1) == 0;

    // wgrad: this wave's row tiles mt = wave + 4*i; per-lane tr-read offset inside a dz halo pixel
    const int q4 = (lane & 15) >> 2, p4 = lane & 3;
    int wtoff[MW];
    bool mvalid[MW];
    int bias_i = -1;                                       // which owned row tile holds centre-tap rows
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int mt = wave + NW * i;
        mvalid[i] = mt < MT;
        int rg = 2 * mt + (p4 >> 1);
        if (rg >= RG) rg = 0;
        const int tap = rg / CG, cg = rg - tap * CG;
        const int ky = tap / KS, kx = tap - ky * KS;
        wtoff[i] = (ky * g.hw + kx) * PIXB + cg * 16 + (p4 & 1) * 8;
        const int rg_lo = 2 * mt, rg_hi = 2 * mt + 1;
        if (mvalid[i] && rg_hi >= CTAP * CG && rg_lo < (CTAP + 1) * CG) bias_i = i;
    }
    const int wpl0 = mil_pix_base<PIXB>(g, 8 * gq + q4, 1), wpl1 = mil_pix_base<PIXB>(g, 8 * gq + q4 + 4, 1);
    const int wxl0 = (8 * gq + q4) * PIXX + p4 * 8;
    f32x4_t wacc[MW][NTX];
    f32x4_t bacc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int nt = 0; nt < NTX; ++nt) wacc[i][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    TileWalker cur, nxt;
    const int bid = mil_xcd_block_id();
    cur.init(g, bid, gridDim.x);
    nxt = cur; nxt.advance();
    u32x4_t rx[NPX];
    if (bid < a.ntiles) mil_fetch_halo<CZ, NPX, T>(rx, rs_z, ht, g, cur.origin(g));
    // The tile's x (mask + wgrad operand) and addend are needed only after the data-gradient MFMA loop; their loads are
    // issued a phase early — behind the previous tile's last barrier, under its weight-gradient loop — into the registers
    // that tile has just finished with, so the ~2 us round trip is no longer waited for in the middle of the tile.
#ifndef MIL_BWD_LATE_ADD_C
#define MIL_BWD_LATE_ADD_C 64     // from this many channels on, the addend is fetched behind the data-gradient loop instead of a
#endif                            // phase early: its registers are not live across that loop (the 64-channel form spilled 15-18 VGPRs)
    constexpr bool LATE_ADD = ADD && CZ >= MIL_BWD_LATE_ADD_C;
    unsigned ooff[NPAIR];
    u32x4_t rxc[NPAIR][NTX][NE], radd[NPAIR][NTX][NE];
    auto fetch_add_late = [&]() {
#pragma unroll
        for (int p = 0; p < NPAIR; ++p)
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) {
                const unsigned off = (LAST_PARTIAL && nt == NTX - 1 && !last_ok) ? MIL_OOB : ooff[p] + nt * 16 * ESZ;
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    radd[p][nt][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_add, (e == 0 || off == MIL_OOB) ? off : off + 16, 0, 0);
            }
    };
    auto fetch_xa = [&](const TileOrigin& o) {
        const int obase = ((o.img0 * g.Ho + o.oy0) * g.Wo + o.ox0) * (CX * ESZ);
        const int ylim = g.Ho - o.oy0, xlim = g.Wo - o.ox0, ilim = g.n_img - o.img0;
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
            const bool ok = (o_pos[p] >> 20) < ilim && ((o_pos[p] >> 10) & 1023) < ylim && (o_pos[p] & 1023) < xlim;
            ooff[p] = ok ? (unsigned)(obase + o_rel[p]) : MIL_OOB;
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) {
                const unsigned off = (LAST_PARTIAL && nt == NTX - 1 && !last_ok) ? MIL_OOB : ooff[p] + nt * 16 * ESZ;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const unsigned oe = (e == 0 || off == MIL_OOB) ? off : off + 16;
                    rxc[p][nt][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, oe, 0, 0);
                    if constexpr (ADD && !LATE_ADD) radd[p][nt][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_add, oe, 0, 0);
                }
            }
        }
    };
    if (bid < a.ntiles) fetch_xa(cur.origin(g));

    // Two dz-halo buffers: the next tile's halo goes into the buffer of the tile before the current one, which every
    // wave has left once this wave is past the current tile's barriers — the "all reads are done" barrier at the top of
    // the tile disappears (two barriers per tile instead of three).  ldsX is written only after the tile's first barrier,
    // by which time every wave has finished the previous tile's weight-gradient loop.
    const int buf_step = a.lds_a2_off;
    int buf = 0;
    MIL_ST_DECL
    for (int tile = bid; tile < a.ntiles; tile += gridDim.x) {
        MIL_ST_BEGIN()
        if (buf_step == 0) __syncthreads();    // single buffer: previous tile's reads of ldsA are done
        char* ldsA_t = ldsA + buf;
        buf = buf_step - buf;
        mil_commit_halo_all<NPX, T, CZ>(rx, ldsA_t, ht);
        MIL_ST_MARK(0)                         // halo commit (incl. the wait for the prefetched loads)

        __syncthreads();                       // dz halo visible
        MIL_ST_MARK(1)                         // barrier 1
        const bool more = tile + (int)gridDim.x < a.ntiles;
        if (more) mil_fetch_halo<CZ, NPX, T>(rx, rs_z, ht, g, nxt.origin(g));
        const TileOrigin o_next = nxt.origin(g);
        cur = nxt; nxt.advance();

        // ---- data gradient: D[cx][pixel] -----------------------------------------------------------
        f32x4_t acc[MTW][NTX];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) acc[m][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if constexpr (PIPE) {
            // one k-step ahead: the fragment reads of step sl+1 are issued before the MFMAs of step sl, and scheduling
            // fences keep that order (left alone, hipcc issues each read right in front of its MFMAs and waits
            // lgkmcnt(0) per group: with two waves per SIMD the loop is then paced by LDS latency)
            Frag8<T> wc[NTX], zc[MTW], wn[NTX], zn[MTW];
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) wc[nt] = lds_frag<T>(ldsW + (nt * 64 + lane) * FRAGB);
#pragma unroll
            for (int m = 0; m < MTW; ++m) zc[m] = lds_pix_frag<T, CZ * 2>(ldsA_t + pixbase[m] + toff[0]);
#pragma unroll
            for (int sl = 0; sl < KSTEPS; ++sl) {
                if (sl + 1 < KSTEPS) {
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt) wn[nt] = lds_frag<T>(ldsW + (((sl + 1) * NTX + nt) * 64 + lane) * FRAGB);
#pragma unroll
                    for (int m = 0; m < MTW; ++m) zn[m] = lds_pix_frag<T, CZ * 2>(ldsA_t + pixbase[m] + toff[sl + 1 < KSTEPS ? sl + 1 : sl]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt) acc[m][nt] = mma8(wc[nt], zc[m], acc[m][nt]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NTX; ++nt) wc[nt] = wn[nt];
#pragma unroll
                for (int m = 0; m < MTW; ++m) zc[m] = zn[m];
            }
        } else {
#pragma unroll
        for (int sl = 0; sl < KSTEPS; ++sl) {
            Frag8<T> wf[NTX];
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) wf[nt] = lds_frag<T>(ldsW + ((sl * NTX + nt) * 64 + lane) * FRAGB);
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const Frag8<T> zf = lds_pix_frag<T, CZ * 2>(ldsA_t + pixbase[m] + toff[sl]);
#pragma unroll
                for (int nt = 0; nt < NTX; ++nt) acc[m][nt] = mma8(wf[nt], zf, acc[m][nt]);
            }
        }

        }
        MIL_ST_MARK(2)                         // next-halo issue + data-gradient MFMA loop
        if constexpr (LATE_ADD) fetch_add_late();      // lands while the x tile is written to LDS
        // x centre tile -> LDS [pixel][CX] (zeros outside the image: no contribution to dW)
#pragma unroll
        for (int p = 0; p < NPAIR; ++p)
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt)
            {
                // lanes whose channels of the last column tile do not exist write to the dump slot instead of branching
                const int dst = (LAST_PARTIAL && nt == NTX - 1 && !last_ok) ? a.lds_dump_off : a.lds_x_off + x_lds[p] + nt * 32;
                if constexpr (T::SPLIT) {       // eight fp32 channels -> 16 bytes of the record's hi plane + 16 bytes of its lo plane
                    const f32x4_t v0 = __builtin_bit_cast(f32x4_t, rxc[p][nt][0]), v1 = __builtin_bit_cast(f32x4_t, rxc[p][nt][1]);
                    const float f8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    bf16x8_t hi, lo;
                    mil_split8(f8, hi, lo);
                    *reinterpret_cast<bf16x8_t*>(smem + dst) = hi;
                    *reinterpret_cast<bf16x8_t*>(smem + dst + CX * 2) = lo;
                } else {
                    *reinterpret_cast<u32x4_t*>(smem + dst) = rxc[p][nt][0];
                }
            }

        // ---- data-gradient epilogue from registers, 8 channels per lane --------------------------------
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float lo = acc[2 * p][nt][i], hi = acc[2 * p + 1][nt][i];
                    if (i == 0) mil_swap16<true>(lo, hi); else mil_swap16<false>(lo, hi);
                    v[i] = lo;
                    v[4 + i] = hi;
                }
                const unsigned off = (LAST_PARTIAL && nt == NTX - 1 && !last_ok) ? MIL_OOB : ooff[p] + nt * 16 * ESZ;
                if constexpr (T::SPLIT) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if constexpr (ADD) {
                            const f32x4_t t = __builtin_bit_cast(f32x4_t, radd[p][nt][e]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[4 * e + i] += t[i];
                        }
                        if constexpr (MASK) {
                            const f32x4_t t = __builtin_bit_cast(f32x4_t, rxc[p][nt][e]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[4 * e + i] *= (t[i] > 0.f ? 1.f : a.slope);
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[0], v[1], v[2], v[3]}), rs_dx, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4_t{v[4], v[5], v[6], v[7]}), rs_dx, off == MIL_OOB ? MIL_OOB : off + 16, 0, 0);
                } else {
                if constexpr (ADD) {
                    const bf16x8_t t = __builtin_bit_cast(bf16x8_t, radd[p][nt][0]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += (float)t[i];
                }
                if constexpr (MASK) {
                    const bf16x8_t t = __builtin_bit_cast(bf16x8_t, rxc[p][nt][0]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= ((float)t[i] > 0.f ? 1.f : a.slope);
                }
                bf16x8_t ov;
#pragma unroll
                for (int i = 0; i < 8; ++i) ov[i] = (__bf16)v[i];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ov), rs_dx, off, 0, 0);
                }
            }
        }
        MIL_ST_MARK(3)                         // x tile write + epilogue + dx stores
        __syncthreads();                       // x centre tile visible
        MIL_ST_MARK(4)                         // barrier 2
        if (more) fetch_xa(o_next);            // next tile's x / addend: lands while the loop below runs

        // ---- weight gradient: rows (tap', dz channel), cols x channel, K = the tile's 256 pixels -------
        if constexpr (PIPE) {
            // same, one 32-pixel k-step ahead and without branches on the wave's row-tile validity (a row tile that
            // does not exist computes on row group 0 and is never stored); only the bias MFMA sits under a scalar branch
            bf16x8_t xc[NL2][NTX], zc[NL2][MW], xn[NL2][NTX], zn[NL2][MW];
            auto loadw = [&](int k32, bf16x8_t (&xf)[NL2][NTX], bf16x8_t (&zf)[NL2][MW]) {
                const int kb = mil_pix_base<PIXB>(g, k32, 1);
                const char* x0 = ldsX + k32 * PIXX + wxl0;
#pragma unroll
                for (int pl = 0; pl < NL2; ++pl) {
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt) xf[pl][nt] = mil_tr_pair(x0 + pl * (CX * 2) + nt * 32, x0 + pl * (CX * 2) + 4 * PIXX + nt * 32);
#pragma unroll
                    for (int i = 0; i < MW; ++i) zf[pl][i] = mil_tr_pair(ldsA_t + kb + wpl0 + wtoff[i] + pl * (CZ * 2), ldsA_t + kb + wpl1 + wtoff[i] + pl * (CZ * 2));
                }
            };
            loadw(0, xc, zc);
#pragma unroll
            for (int k32 = 0; k32 < 256; k32 += 32) {
                if (k32 + 32 < 256) loadw(k32 + 32, xn, zn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MW; ++i) {
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt) {
                        if constexpr (T::SPLIT) {
                            wacc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zc[1][i], xc[0][nt], wacc[i][nt], 0, 0, 0);
                            wacc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zc[0][i], xc[1][nt], wacc[i][nt], 0, 0, 0);
                        }
                        wacc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zc[0][i], xc[0][nt], wacc[i][nt], 0, 0, 0);
                    }
                    if (i == bias_i) {
                        if constexpr (T::SPLIT) bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zc[1][i], ones, bacc, 0, 0, 0);
                        bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zc[0][i], ones, bacc, 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pl = 0; pl < NL2; ++pl) {
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt) xc[pl][nt] = xn[pl][nt];
#pragma unroll
                    for (int i = 0; i < MW; ++i) zc[pl][i] = zn[pl][i];
                }
            }
        } else {
#pragma unroll 2
        for (int k32 = 0; k32 < 256; k32 += 32) {
            // pixel k = k32 + (8*gq + q4 [+4]): the halo offset is additive in the two parts (no carries between
            // their bit fields), so the lane part is tile-invariant and the k32 part is wave-uniform (scalar unit)
            const int kb = mil_pix_base<PIXB>(g, k32, 1);
            const int pb0 = kb + wpl0, pb1 = kb + wpl1;
            const char* x0 = ldsX + k32 * PIXX + wxl0;
            const char* x1 = x0 + 4 * PIXX;
            bf16x8_t xf[NTX];
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) xf[nt] = mil_tr_pair(x0 + nt * 32, x1 + nt * 32);
            if constexpr (T::SPLIT) {            // dW' += dz_lo*x_hi + dz_hi*x_lo + dz_hi*x_hi from the hi / lo planes of both tiles
                bf16x8_t xl[NTX];
#pragma unroll
                for (int nt = 0; nt < NTX; ++nt) xl[nt] = mil_tr_pair(x0 + CX * 2 + nt * 32, x1 + CX * 2 + nt * 32);
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    if (mvalid[i]) {
                        const bf16x8_t zf = mil_tr_pair(ldsA_t + pb0 + wtoff[i], ldsA_t + pb1 + wtoff[i]);
                        const bf16x8_t zl = mil_tr_pair(ldsA_t + pb0 + CZ * 2 + wtoff[i], ldsA_t + pb1 + CZ * 2 + wtoff[i]);
#pragma unroll
                        for (int nt = 0; nt < NTX; ++nt) {
                            wacc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zl, xf[nt], wacc[i][nt], 0, 0, 0);
                            wacc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, xl[nt], wacc[i][nt], 0, 0, 0);
                            wacc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, xf[nt], wacc[i][nt], 0, 0, 0);
                        }
                        if (i == bias_i) {
                            bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zl, ones, bacc, 0, 0, 0);
                            bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, ones, bacc, 0, 0, 0);
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                if (mvalid[i]) {
                    const bf16x8_t zf = mil_tr_pair(ldsA_t + pb0 + wtoff[i], ldsA_t + pb1 + wtoff[i]);
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt)
                        wacc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, xf[nt], wacc[i][nt], 0, 0, 0);
                    if (i == bias_i) bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, ones, bacc, 0, 0, 0);
                }
            }
        }
        }
        MIL_ST_MARK(5)                         // x/addend issue + weight-gradient loop
#ifdef MIL_STAMP
        ++st_tiles;
#endif
    }
#ifdef MIL_STAMP
    if (a.stamp && lane == 0) {
        unsigned long long* d = a.stamp + ((size_t)blockIdx.x * NW + wave) * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = st_sum[i];
        d[6] = (unsigned long long)st_tiles;
        d[7] = __builtin_amdgcn_s_memrealtime() - st_rt0;      // 100 MHz ticks over the whole tile loop
    }
#endif

    // ---- partial sums -> slab: rows tap'*CZ + co, cols ci; bias sums in the extra row tile ---------------
    constexpr int SLAB_COLS = NTX * 16;
    constexpr size_t SLAB_ELEMS = (size_t)(MT + 1) * 16 * SLAB_COLS;
    float* slab = a.slab + (size_t)blockIdx.x * SLAB_ELEMS;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        if (!mvalid[i]) continue;
        const int mt = wave + NW * i;
#pragma unroll
        for (int nt = 0; nt < NTX; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                slab[(size_t)(mt * 16 + gq * 4 + e) * SLAB_COLS + nt * 16 + r] = wacc[i][nt][e];
        if (i == bias_i && r == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = mt * 16 + gq * 4 + e;          // = tap'*CZ + co
                const int co = row - CTAP * CZ;
                if (co >= 0 && co < CZ) slab[(size_t)MT * 16 * SLAB_COLS + co] = bacc[e];
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// 16x16-tile form with COMPILE-TIME geometry, explicit software pipelining and no branches in the MFMA loops
// (20-channel layers; tiles of one image, halo 18x18).  What the generic kernel above loses on these maps — measured
// with the -DMIL_STAMP build: 2.4 k cycles in the data-gradient loop and 4.8 k in the weight-gradient loop per tile and
// wave for 28 + 36 MFMAs (0.45 k + 0.58 k cycles of matrix pipe) — is LDS latency: hipcc issues every fragment read
// right in front of the MFMA pair that consumes it (`s_waitcnt lgkmcnt(0)` per pair), the scalar branches on the
// wave's row-tile validity keep it from hoisting anything, and every read pays run-time address arithmetic.  Here
//   * tile shape and pitches are constants, so a fragment address is ONE per-lane base register + an immediate;
//   * both loops are written one k-step ahead: the reads of step s+1 are issued, then the MFMAs of step s, with
//     scheduling fences so the order survives (the compiler still counts the waits: builtin LDS loads);
//   * the K order is the packed one of geom.cuh ("K20"): 24 k-groups in 6 k-steps for the data gradient, 45
//     four-channel row pieces in 12 row tiles for the weight gradient (standard order: 7 k-steps, 14 row tiles); the LDS
//     record of a dz pixel is [ch 0-15][ch 16-19][ch 16-19 of the next pixel];
//   * every wave runs the same instruction stream: waves whose second row tile does not exist compute on row piece 0
//     and never store it;
//   * ONE barrier per tile: the x tile is prefetched and committed together with the dz halo (both double-buffered), the
//     mask is read back from it, so between barriers a wave runs data gradient, epilogue and weight gradient at its own
//     pace and the waves of a SIMD overlap one's MFMAs with another's epilogue instead of marching in phase;
//   * the bias gradient needs no MFMA of its own: channel CX-1 of the x tile in LDS (a padding channel) is set to 1, so
//     column CX-1 of the centre-tap rows of dW' is sum_q dz[q][co] (the reduction reads db from there);
//   * GRADIENT tensors (dz, addend, dx) are read and written in 8-byte pieces of four channels, five per pixel, at a
//     RUN-TIME pixel stride a.gpx: 48 bytes (the padded 24-channel layout every activation has) or 40 (dense: the
//     gradient chain of the 20-channel layer is produced and consumed only by kernels that know this layout —
//     MIL_DT_BF16_DGRAD —, 17 % fewer bytes on three of the four tensor passes).
#ifndef MIL_BWD16_K20
#define MIL_BWD16_K20 1
#endif
#ifndef MIL_BWD16_PRIO
#define MIL_BWD16_PRIO 0      // measured: no difference either way (366-385 us with and without on the same box)
#endif
template <bool ADD, bool MASK>
__global__ __launch_bounds__(512, 4) void conv_bwd_fused16_kernel(BwdFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int CZ = 24, NTX = 2, KS = 3, NW = 8;
    constexpr int PIXB = 48, PIXX = 48, CX = 24, HW = 18, ROWB = HW * PIXB;      // 864 B per halo row
    constexpr int KSTEPS_STD = 7;
    constexpr int NTHR = 512, MTW = 2, KSTEPS = MIL_K20_STEPS, MT = 12, MW = 2;
    constexpr int NPX = 4;                                   // 8-byte halo pieces per thread: 3 slots of pieces 0-3, 1 slot of piece 4
    constexpr int HALO0 = 16;                                // 16 spare bytes in FRONT of a halo tile (dump slot / pixel 0's back-copy)
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    char* ldsW = smem + a.lds_w_off;
    {
        // the K20 k-steps sit behind the standard ones
        mil_stage_filter(ldsW, reinterpret_cast<const char*>(a.w) + KSTEPS_STD * NTX * 64 * 16, KSTEPS * NTX * 64 * 16, tid, NTHR);
        // the last halo record's "next pixel" slot is never written by a commit: finite once and for all (zero weights read it)
        if (tid < 2) *reinterpret_cast<u32x2_t*>(smem + tid * a.lds_a2_off + HALO0 + (HW * HW - 1) * PIXB + 40) = u32x2_t{0u, 0u};
    }
    const __amdgpu_buffer_rsrc_t rs_z = mil_rsrc(a.dz, a.z_bytes);
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rs_add = mil_rsrc(a.addend, a.addend ? a.g_bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_dx = mil_rsrc(a.dx, a.g_bytes);
    const int H = g.H, W = g.W, GPX = a.gpx;                  // GPX: bytes per pixel of dz / addend / dx (48 or 40)

    // ---- halo pieces of this thread (8 bytes = four channels; the padding piece of the 48-byte layout is never read) ----
    // slots 0-2: piece id = tid + 512*s -> (pixel id>>2, piece id&3): LDS offset = h_lds0 + s*128*PIXB (an immediate)
    // slot 3   : pixel tid (< 324), piece 4 (channels 16-19): committed to its own record AND to the previous pixel's
    //            "next pixel" slot (record offset - 40; pixel 0's lands in the spare bytes)
    // h_pos = hy<<10 | hx, negative when the slot is unused (its writes go to the spare bytes)
    int h_pos[NPX];
    const int h_j8 = (tid & 3) * 8;
    const int h_lds0 = (tid >> 2) * PIXB + h_j8;
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int px = i < 3 ? (tid >> 2) + 128 * i : tid;
        const int hy = px / HW, hx = px - hy * HW;
        h_pos[i] = px < HW * HW ? (hy << 10) | hx : (int)0x80000000u;
    }
    u32x2_t rx[NPX];
    auto fetch_halo = [&](const TileOrigin& o) {
        const int iy0 = o.oy0 - 1, ix0 = o.ox0 - 1;
        const int base = ((o.img0 * H + iy0) * W + ix0) * GPX;      // may be negative; valid lanes are not
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            int p = h_pos[i];
            asm volatile("" : "+v"(p));       // keeps what is derived from p from being hoisted into loop-long registers
            const int hy = (p >> 10) & 1023, hx = p & 1023;
            const bool ok = (p >= 0) & ((unsigned)(iy0 + hy) < (unsigned)H) & ((unsigned)(ix0 + hx) < (unsigned)W);
            const int rel = (hy * W + hx) * GPX + (i < 3 ? h_j8 : 32);
            rx[i] = __builtin_amdgcn_raw_buffer_load_b64(rs_z, ok ? (unsigned)(base + rel) : MIL_OOB, 0, 0);
        }
    };

    // ---- data gradient: fragment address = per-lane base + immediate -----------------------------------------
    // k-group q = 4*sl + gq of geom.cuh's K20 order; halo offset of q relative to the pixel's top-left tap
    auto koff = [](int q) { return mil_k20_off(q, ROWB, PIXB); };
    const int bA = (wave * MTW * HW + r) * PIXB + 16 * gq;        // pixel (row 2*wave [+m], col r), lane group part
    int zb[KSTEPS];
#pragma unroll
    for (int sl = 0; sl < KSTEPS; ++sl) {
        const int o0 = koff(4 * sl);
        const int d1 = koff(4 * sl + 1) - o0 - 16, d2 = koff(4 * sl + 2) - o0 - 32, d3 = koff(4 * sl + 3) - o0 - 48;
        zb[sl] = bA + (gq == 1 ? d1 : gq == 2 ? d2 : gq == 3 ? d3 : 0);      // all-zero deltas fold to bA itself
    }
    // ---- epilogue pair: after the permlane swap a lane holds 8 consecutive channels of pixel (2*wave + (gq&1), r)
    const int e_ty = wave * MTW + (gq & 1);
    const int c_off = (gq >> 1) * 16;                              // byte offset of the lane's channels inside a pixel (x and gradients)
    const int x_lds = (e_ty * 16 + r) * PIXX + c_off;             // own 16-byte piece(s) inside an x-tile buffer
    const bool last_ok = (gq >> 1) == 0;                           // channels 24..31 do not exist
    // ---- weight gradient: row pieces (tap', four dz channels) of row tiles mt = wave, wave + 8; K = the 256 centre pixels
    const int q4 = (lane & 15) >> 2, p4 = lane & 3;
    int zw[MW][2];
    {
        const int kl0 = 8 * gq + q4, kl1 = kl0 + 4;               // pixel of this lane inside a 32-pixel k-step
        const int wpl0 = ((kl0 >> 4) * HW + (kl0 & 15)) * PIXB, wpl1 = ((kl1 >> 4) * HW + (kl1 & 15)) * PIXB;
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            int P = 4 * (wave + NW * i) + p4;                      // rows tap'*20 + co
            if (P >= KS * KS * 5) P = 0;                           // rows that do not exist: finite data, never reduced
            const int tap = P / 5, c4 = P - tap * 5;
            const int wt = ((tap / KS) * HW + (tap % KS)) * PIXB + c4 * 8;
            zw[i][0] = wpl0 + wt; zw[i][1] = wpl1 + wt;
        }
    }
    const int xb = (8 * gq + q4) * PIXX + p4 * 8;                     // inside an x-tile buffer
    f32x4_t wacc[MW][NTX];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int nt = 0; nt < NTX; ++nt) wacc[i][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    TileWalker cur, nxt;
    const int bid = mil_xcd_block_id();
    cur.init(g, bid, gridDim.x);
    nxt = cur; nxt.advance();
    if (bid < a.ntiles) fetch_halo(cur.origin(g));
    // x (mask source + weight-gradient operand) is prefetched with the halo and committed to LDS with it, so ONE barrier
    // per tile orders everything; the addend is only needed in the epilogue and is prefetched right after the previous one.
    unsigned goff = MIL_OOB, goff_n = MIL_OOB;                 // byte offset of the lane's channels in addend / dx (this / next tile)
    // a lane's second piece holds channels 16-19 only (20-23 are padding): 8 bytes everywhere
    u32x4_t rxc0, radd0;
    u32x2_t rxc1, radd1;
    auto fetch_x = [&](const TileOrigin& o) {
        const int opix = (o.img0 * H + o.oy0 + e_ty) * W + o.ox0 + r;
        const bool ok = e_ty < H - o.oy0 && r < W - o.ox0;
        const unsigned xoff = ok ? (unsigned)(opix * (CX * 2) + c_off) : MIL_OOB;
        goff_n = ok ? (unsigned)(opix * GPX + c_off) : MIL_OOB;
        rxc0 = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff, 0, 0);
        rxc1 = __builtin_amdgcn_raw_buffer_load_b64(rs_x, last_ok ? xoff + 32 : MIL_OOB, 0, 0);
    };
    auto fetch_add = [&]() {
        if constexpr (ADD) {
            radd0 = __builtin_amdgcn_raw_buffer_load_b128(rs_add, goff_n, 0, 0);
            radd1 = __builtin_amdgcn_raw_buffer_load_b64(rs_add, last_ok ? goff_n + 32 : MIL_OOB, 0, 0);
        }
    };
    if (bid < a.ntiles) { fetch_x(cur.origin(g)); fetch_add(); }

    const int buf_step = a.lds_a2_off, xbuf_step = a.lds_x2_off;
    int buf = 0, xbuf = 0;
#if MIL_BWD16_PRIO
    // the second-dispatched half of an 8-wave workgroup loses issue arbitration to the older half (priority, then age) and
    // sets the pace at every barrier (finding 3 of DESIGN.md): one static priority raise for it, no per-phase flips
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
    MIL_ST_DECL
    for (int tile = bid; tile < a.ntiles; tile += gridDim.x) {
        MIL_ST_BEGIN()
        // Two halo buffers and two x-tile buffers: a wave may commit tile t+1 while others still compute tile t (they
        // read the other buffers); it can reach tile t+2's commit only past barrier t+1, i.e. after every wave left tile t.
        char* ldsA_t = smem + buf + HALO0;
        char* ldsX_t = smem + a.lds_x_off + xbuf;
        buf = buf_step - buf;
        xbuf = xbuf_step - xbuf;
        {
            int p2 = h_pos[2], p3 = h_pos[3];
            asm volatile("" : "+v"(p2), "+v"(p3));
            *reinterpret_cast<u32x2_t*>(ldsA_t + h_lds0) = rx[0];
            *reinterpret_cast<u32x2_t*>(ldsA_t + h_lds0 + 128 * PIXB) = rx[1];
            *reinterpret_cast<u32x2_t*>(ldsA_t + (p2 >= 0 ? h_lds0 + 256 * PIXB : -16)) = rx[2];
            const int l3 = p3 >= 0 ? tid * PIXB + 32 : -16;       // an unused slot writes the spare bytes twice
            *reinterpret_cast<u32x2_t*>(ldsA_t + l3) = rx[3];
            *reinterpret_cast<u32x2_t*>(ldsA_t + (p3 >= 0 ? l3 - 40 : -8)) = rx[3];
        }
        // x centre tile -> LDS [pixel][CX] (zeros outside the image); its padding channel CX-1 := 1 (bias sums)
        *reinterpret_cast<u32x4_t*>(ldsX_t + x_lds) = rxc0;
        *reinterpret_cast<u32x4_t*>(last_ok ? ldsX_t + x_lds + 32 : smem + a.lds_dump_off) = u32x4_t{rxc1[0], rxc1[1], 0u, 0x3f800000u};
        goff = goff_n;
        MIL_ST_MARK(0)
        __syncthreads();                       // dz halo and x tile visible: the tile's only barrier
        MIL_ST_MARK(1)
        const bool more = tile + (int)gridDim.x < a.ntiles;
        if (more) { const TileOrigin o_next = nxt.origin(g); fetch_halo(o_next); fetch_x(o_next); }
        cur = nxt; nxt.advance();
        // own x pieces back from LDS for the LeakyReLU mask of the epilogue (their registers now hold the next tile's)
        u32x4_t mk0;
        u32x2_t mk1;
        if constexpr (MASK) {
            mk0 = *reinterpret_cast<const u32x4_t*>(ldsX_t + x_lds);
            mk1 = *reinterpret_cast<const u32x2_t*>(last_ok ? ldsX_t + x_lds + 32 : smem + a.lds_dump_off);
        }

        // ---- data gradient D[cx][pixel], one k-step ahead ----------------------------------------------------
        f32x4_t acc[MTW][NTX];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) acc[m][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        {
            Frag8<BF16> wc[NTX], zc[MTW], wn[NTX], zn[MTW];
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) wc[nt] = lds_frag<BF16>(ldsW + (nt * 64 + lane) * 16);
#pragma unroll
            for (int m = 0; m < MTW; ++m) zc[m] = lds_frag<BF16>(ldsA_t + zb[0] + koff(0) + m * ROWB);
#pragma unroll
            for (int sl = 0; sl < KSTEPS; ++sl) {
                if (sl + 1 < KSTEPS) {
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt) wn[nt] = lds_frag<BF16>(ldsW + (((sl + 1) * NTX + nt) * 64 + lane) * 16);
#pragma unroll
                    for (int m = 0; m < MTW; ++m) zn[m] = lds_frag<BF16>(ldsA_t + zb[sl + 1] + koff(4 * (sl + 1)) + m * ROWB);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt) acc[m][nt] = mma8(wc[nt], zc[m], acc[m][nt]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NTX; ++nt) wc[nt] = wn[nt];
#pragma unroll
                for (int m = 0; m < MTW; ++m) zc[m] = zn[m];
            }
        }
        MIL_ST_MARK(2)

        // ---- data-gradient epilogue from registers: channels 0-15 as 8 per lane, channels 16-19 as 4 per lane --------
        {
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo = acc[0][0][i], hi = acc[1][0][i];
                if (i == 0) mil_swap16<true>(lo, hi); else mil_swap16<false>(lo, hi);
                v[i] = lo;
                v[4 + i] = hi;
            }
            if constexpr (ADD) {
                const bf16x8_t t = __builtin_bit_cast(bf16x8_t, radd0);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += (float)t[i];
            }
            if constexpr (MASK) {
                const bf16x8_t t = __builtin_bit_cast(bf16x8_t, mk0);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] *= ((float)t[i] > 0.f ? 1.f : a.slope);
            }
            bf16x8_t ov;
#pragma unroll
            for (int i = 0; i < 8; ++i) ov[i] = (__bf16)v[i];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ov), rs_dx, goff, 0, 0);
        }
        {
            // second column tile: only the lanes that end up with channels 16-19 (gq>>1 == 0) store; channels 20-23 are padding
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo = acc[0][1][i], hi = acc[1][1][i];
                if (i == 0) mil_swap16<true>(lo, hi); else mil_swap16<false>(lo, hi);
                v[i] = lo;
            }
            if constexpr (ADD) {
                const bf16x4_t t = __builtin_bit_cast(bf16x4_t, radd1);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += (float)t[i];
            }
            if constexpr (MASK) {
                const bf16x4_t t = __builtin_bit_cast(bf16x4_t, mk1);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= ((float)t[i] > 0.f ? 1.f : a.slope);
            }
            bf16x4_t ov;
#pragma unroll
            for (int i = 0; i < 4; ++i) ov[i] = (__bf16)v[i];
            const u32x2_t ou = __builtin_bit_cast(u32x2_t, ov);
            const unsigned off = last_ok ? goff + 32 : MIL_OOB;
            if (GPX == 48) __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ou[0], ou[1], 0u, 0u}, rs_dx, off, 0, 0);      // + the zero padding channels
            else __builtin_amdgcn_raw_buffer_store_b64(ou, rs_dx, off, 0, 0);
        }
        MIL_ST_MARK(3)
        if (more) fetch_add();                 // next tile's addend: its registers are free now
        MIL_ST_MARK(4)

        // ---- weight gradient, one k-step (32 pixels = two tile rows) ahead ----------------------------------------
        {
            bf16x8_t xc[NTX], zc[MW], xn[NTX], zn[MW];
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) xc[nt] = mil_tr_pair(ldsX_t + xb + nt * 32, ldsX_t + xb + nt * 32 + 4 * PIXX);
#pragma unroll
            for (int i = 0; i < MW; ++i) zc[i] = mil_tr_pair(ldsA_t + zw[i][0], ldsA_t + zw[i][1]);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                if (kk + 1 < 8) {
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt)
                        xn[nt] = mil_tr_pair(ldsX_t + xb + (kk + 1) * 32 * PIXX + nt * 32, ldsX_t + xb + (kk + 1) * 32 * PIXX + nt * 32 + 4 * PIXX);
#pragma unroll
                    for (int i = 0; i < MW; ++i)
                        zn[i] = mil_tr_pair(ldsA_t + zw[i][0] + (kk + 1) * 2 * ROWB, ldsA_t + zw[i][1] + (kk + 1) * 2 * ROWB);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    // waves 4..7 own only one of the 12 row tiles: their second set of MFMAs would multiply into a
                    // tile that is never stored — skipped under a scalar (wave-uniform) branch, the reads stay
                    if (i == MW - 1 && wave + NW * i >= MT) continue;
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt)
                        wacc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zc[i], xc[nt], wacc[i][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NTX; ++nt) xc[nt] = xn[nt];
#pragma unroll
                for (int i = 0; i < MW; ++i) zc[i] = zn[i];
            }
        }
        MIL_ST_MARK(5)
#ifdef MIL_STAMP
        ++st_tiles;
#endif
    }
#ifdef MIL_STAMP
    if (a.stamp && lane == 0) {
        unsigned long long* d = a.stamp + ((size_t)blockIdx.x * NW + wave) * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = st_sum[i];
        d[6] = (unsigned long long)st_tiles;
        d[7] = __builtin_amdgcn_s_memrealtime() - st_rt0;      // 100 MHz ticks over the whole tile loop
    }
#endif
    // ---- partial sums -> slab: rows tap'*20 + co, cols ci (col CX-1 of the centre-tap rows = bias sums) ---------
    constexpr int SLAB_COLS = NTX * 16;
    constexpr size_t SLAB_ELEMS = (size_t)(14 + 1) * 16 * SLAB_COLS;      // the launcher's slab pitch (generic kernel's row count)
    float* slab = a.slab + (size_t)blockIdx.x * SLAB_ELEMS;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int mt = wave + NW * i;
        if (mt >= MT) continue;
#pragma unroll
        for (int nt = 0; nt < NTX; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                slab[(size_t)(mt * 16 + gq * 4 + e) * SLAB_COLS + nt * 16 + r] = wacc[i][nt][e];
    }
}

#include "conv_bwd_fused16_x3.cuh"

// (the slab reduction for this layout — rows tap'*CZ + co with tap' the flipped tap, cols ci — is kind 1 of reduce.cuh)

// MIL_BWD16=0 falls back to the generic kernel on 16x16 tiles too (A/B runs, bit-compatible results up to the bias
// gradient's summation path)
static bool mil_bwd16_enabled() {
    static const bool v = [] { const char* e = mil_ab_env("MIL_BWD16"); return !(e && e[0] == '0'); }();
    return v;
}

// ---------------------------------------------------------------------------------------------
template <typename T, int CZ, int NTX, int KS, int NW = 4>
static int run_bwd_fused(BwdFusedArgs a, float* dw, float* db, void* ws, size_t ws_bytes, int cout, int cin, int accumulate, bool query,
                         size_t* need, hipStream_t stream, bool dense_grads = false) {
    constexpr int ESZ = T::ESZ, EM = ESZ / 2;               // EM: BwdFusedArgs pointers are typed __bf16 — fp32 tensors advance two per element
    constexpr int PIXB = mil_pix_pitch(CZ, ESZ);
    constexpr int CX = mil_nt_to_cp(NTX);
    constexpr int PIXX = mil_pix_pitch(CX, ESZ);
    constexpr int DUMPA = T::SPLIT ? CZ * 2 + 16 : 16, DUMPX = T::SPLIT ? CX * 2 + 16 : 16;      // dump slots (split: a hi and a lo piece)
    constexpr int KSTEPS = (KS * KS * (CZ / 8) + 3) / 4;
    constexpr int RG = KS * KS * (CZ / 8);
    constexpr int MT = (RG + 1) / 2;
    mil_geom_tiles(a.g, 8);
    {   // the 8x8-tiles-of-four-images shape stages 400 halo pixels; where that (plus filter and x tile) does not fit in
        // LDS but the 16x16 shape's 324 pixels would, take 16x16
        const int px = (a.g.hh * a.g.hw) << a.g.ti_log2;
        const int need = ((px * PIXB + 15) & ~15) + DUMPA + KSTEPS * NTX * 64 * 8 * ESZ + 256 * PIXX + DUMPX;
        if (need > 160 * 1024 && a.g.tw_log2 == 3 && a.g.ti_log2 == 2 && (a.g.Wo > 8 || a.g.Ho > 8)) mil_geom_set(a.g, 4, 4, 0);
    }
    const int halo_px = (a.g.hh * a.g.hw) << a.g.ti_log2;
    if (halo_px > 400 || a.g.hh >= 1024 || a.g.hw >= 1024) return MIL_ERR_UNSUPPORTED;
    // buffer descriptors address < 2 GiB: larger launches are split by images, later chunks accumulating into dW/db
    const size_t img_bytes = (size_t)a.g.H * a.g.W * (CZ > CX ? CZ : CX) * ESZ;
    int chunk = mil_imgs_under_2g(img_bytes);
    if (chunk >= 16) chunk &= ~15;
    const int n_total = a.g.n_img;
    if (chunk < n_total) { a.g.n_img = chunk; a.g.n_groups = (chunk + (1 << a.g.ti_log2) - 1) >> a.g.ti_log2; }
    const int a_bytes = ((halo_px * PIXB + 15) & ~15) + DUMPA;   // + dump slot behind the halo for the branch-free commit
    const int w_bytes = KSTEPS * NTX * 64 * 8 * ESZ;
    const int x_bytes = 256 * PIXX;
    const bool dbuf = 2 * (2 * a_bytes + w_bytes + x_bytes + DUMPX) <= 160 * 1024;      // second halo buffer if two workgroups still fit
    // 24-channel layers on 16x16 tiles of one image: the compile-time-geometry, software-pipelined, one-barrier form
    // (two halo buffers AND two x-tile buffers: 70 KB, two workgroups per CU)
    bool t16 = false;
    if constexpr (CZ == 24 && NTX == 2 && KS == 3 && NW == 8 && !T::SPLIT)
        t16 = mil_bwd16_enabled() && dbuf && a.g.tw_log2 == 4 && a.g.th_log2 == 4 && a.g.ti_log2 == 0 && a.g.H < 1024 && a.g.W < 1024 &&
              cout == 20;      // the K20 order (and the packed filter's second section) exists for 20 dz channels
    if (dense_grads && !t16) return MIL_ERR_UNSUPPORTED;      // only the 16x16-tile kernel reads the dense gradient layout
    const int lds = (dbuf ? 2 : 1) * a_bytes + w_bytes + (t16 ? 2 : 1) * x_bytes + DUMPX;          // + dump slot for the x-tile writes
    if (lds > 160 * 1024) return MIL_ERR_UNSUPPORTED;
    const int ntiles = a.g.n_groups * a.g.tiles_y * a.g.tiles_x;
    auto kern = a.addend ? (a.apply_mask ? conv_bwd_fused_kernel<T, CZ, NTX, KS, true, true, NW> : conv_bwd_fused_kernel<T, CZ, NTX, KS, true, false, NW>)
                         : (a.apply_mask ? conv_bwd_fused_kernel<T, CZ, NTX, KS, false, true, NW> : conv_bwd_fused_kernel<T, CZ, NTX, KS, false, false, NW>);
    if constexpr (CZ == 24 && NTX == 2 && KS == 3 && NW == 8 && !T::SPLIT) {
        if (t16) kern = a.addend ? (a.apply_mask ? conv_bwd_fused16_kernel<true, true> : conv_bwd_fused16_kernel<true, false>)
                                 : (a.apply_mask ? conv_bwd_fused16_kernel<false, true> : conv_bwd_fused16_kernel<false, false>);
    }
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MIL_ERR_LAUNCH;
    }
    // the workspace query (no pointers yet) must size for the same grid the launch will use: occupancy does not depend
    // on the ADD/MASK variant's few registers, but take the minimum over the variants to be safe
    int per_cu = 3;
    {
        int o1 = mil_resident_per_cu(conv_bwd_fused_kernel<T, CZ, NTX, KS, true, true, NW>, lds, 3, 64 * NW);
        int o2 = mil_resident_per_cu(conv_bwd_fused_kernel<T, CZ, NTX, KS, false, true, NW>, lds, 3, 64 * NW);
        int o3 = mil_resident_per_cu(conv_bwd_fused_kernel<T, CZ, NTX, KS, true, false, NW>, lds, 3, 64 * NW);
        if constexpr (CZ == 24 && NTX == 2 && KS == 3 && NW == 8 && !T::SPLIT) {
            if (t16) {
                o1 = mil_resident_per_cu(conv_bwd_fused16_kernel<true, true>, lds, 3, 512);
                o2 = mil_resident_per_cu(conv_bwd_fused16_kernel<false, true>, lds, 3, 512);
                o3 = mil_resident_per_cu(conv_bwd_fused16_kernel<true, false>, lds, 3, 512);
            }
        }
        per_cu = o1 < o2 ? o1 : o2; per_cu = per_cu < o3 ? per_cu : o3;
    }
    int grid = mil_num_cus() * per_cu;
    if (grid > ntiles) grid = ntiles;
    const size_t slab_elems = (size_t)(MT + 1) * 16 * NTX * 16;
    size_t bytes = slab_elems * grid * sizeof(float);
#ifdef MIL_STAMP
    const size_t stamp_off = (bytes + 15) & ~(size_t)15;
    bytes = stamp_off + (size_t)grid * NW * 8 * sizeof(unsigned long long);
#endif
    if (query) { *need = bytes; return MIL_OK; }
    if (!ws || ws_bytes < bytes) return MIL_ERR_ARG;
#ifdef MIL_STAMP
    a.stamp = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + stamp_off);
#endif
    const int a_tot = (dbuf ? 2 : 1) * a_bytes;
    a.slab = (float*)ws; a.lds_w_off = a_tot; a.lds_x_off = a_tot + w_bytes; a.lds_dump_off = a_tot + w_bytes + x_bytes * (t16 ? 2 : 1);
    a.lds_a2_off = dbuf ? a_bytes : 0;
    a.lds_x2_off = t16 ? x_bytes : 0;
    if (grid <= 0) return MIL_OK;
    const int n_rows = KS * KS * CZ;
    const BwdFusedArgs a0 = a;
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        const int n = n_total - i0 < chunk ? n_total - i0 : chunk;
        BwdFusedArgs c = a0;
        c.g.n_img = n;
        c.g.n_groups = (n + (1 << c.g.ti_log2) - 1) >> c.g.ti_log2;
        c.ntiles = c.g.n_groups * c.g.tiles_y * c.g.tiles_x;
        const int gz = dense_grads ? 20 : CZ, gx = dense_grads ? 20 : CX;      // channels per pixel of the gradient tensors
        const size_t zo = (size_t)i0 * a.g.H * a.g.W * gz, xo = (size_t)i0 * a.g.H * a.g.W * CX, go = (size_t)i0 * a.g.H * a.g.W * gx;
        c.dz = a0.dz + zo * EM; c.x = a0.x + xo * EM; c.dx = a0.dx + go * EM;
        if (a0.addend) c.addend = a0.addend + go * EM;
        c.z_bytes = (unsigned)((size_t)n * a.g.H * a.g.W * gz * ESZ);
        c.x_bytes = (unsigned)((size_t)n * a.g.H * a.g.W * CX * ESZ);
        c.g_bytes = (unsigned)((size_t)n * a.g.H * a.g.W * gx * ESZ);
        c.gpx = gz * 2;
        int gr = grid < c.ntiles ? grid : c.ntiles;
        hipLaunchKernelGGL(kern, dim3(gr), dim3(64 * NW), lds, stream, c);
        MIL_CHECK_LAUNCH();
        {
            MilReduceJob j{};
            j.slab = (const float*)ws; j.nslab = gr; j.slab_elems = slab_elems; j.slab_cols = NTX * 16; j.n_rows = n_rows;
            j.dw = dw; j.db = db; j.cout = cout; j.cin = cin; j.ks = KS; j.kind = 1; j.cinp = CZ;
            j.bias_off = t16 ? (KS * KS / 2) * CZ * NTX * 16 + (CX - 1) : MT * 16 * NTX * 16; j.bias_stride = t16 ? NTX * 16 : 1;
            if (t16) {                           // rows tap'*20 + co (four-channel row pieces)
                j.cinp = 20; j.n_rows = KS * KS * 20; j.bias_off = (KS * KS / 2) * 20 * NTX * 16 + (CX - 1);
            }
            j.accumulate = (i0 > 0) ? 1 : accumulate;
            mil_reduce_or_defer(j, stream, /*may_defer=*/chunk >= n_total);      // a split launch re-uses the slabs per chunk
        }
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

// Split precision on 16x16 tiles of one image (layer 1 at 256x256 and 512x512 tiles): conv_bwd_fused16x3_kernel.  Returns
// MIL_ERR_UNSUPPORTED when the shape is not its own (the caller then takes the generic kernel, padded layout only).
static int run_bwd_fused16_x3(BwdFusedArgs a, float* dw, float* db, void* ws, size_t ws_bytes, int cout, int cin, int accumulate, bool query,
                              size_t* need, hipStream_t stream, bool dense_grads) {
    constexpr int KS = 3, NTX = 2, MT_PITCH = 14;                  // slab pitch = the generic kernel's row-tile count (+1 bias tile)
    if (cout != 20 || cin != 20) return MIL_ERR_UNSUPPORTED;
    mil_geom_tiles(a.g, 8);
    if (a.g.tw_log2 != 4 || a.g.th_log2 != 4 || a.g.ti_log2 != 0 || a.g.H >= 1024 || a.g.W >= 1024) return MIL_ERR_UNSUPPORTED;
    constexpr int A_PLANE = 16 + 18 * 18 * 48, X_PLANE = 256 * 48, W_BYTES = MIL_K20_STEPS * NTX * 64 * 32;
    constexpr int lds = 2 * A_PLANE + W_BYTES + 2 * X_PLANE + 64;
    const int gpx = dense_grads ? 80 : 96, xpx = 96;
    const size_t img_bytes = (size_t)a.g.H * a.g.W * 96;
    int chunk = mil_imgs_under_2g(img_bytes);
    if (chunk >= 16) chunk &= ~15;
    const int n_total = a.g.n_img;
    if (chunk < n_total) { a.g.n_img = chunk; a.g.n_groups = chunk; }
    const int ntiles = a.g.n_groups * a.g.tiles_y * a.g.tiles_x;
    auto pick = [](bool add, bool mask) {
        return add ? (mask ? conv_bwd_fused16x3_kernel<true, true> : conv_bwd_fused16x3_kernel<true, false>)
                   : (mask ? conv_bwd_fused16x3_kernel<false, true> : conv_bwd_fused16x3_kernel<false, false>);
    };
    auto kern = pick(a.addend != nullptr, a.apply_mask != 0);
    static std::atomic<unsigned long long> attr_set{0};
    if (mil_device_needs(attr_set)) {
        for (int v = 0; v < 4; ++v)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(pick(v & 1, v & 2)), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
                return MIL_ERR_LAUNCH;
        mil_device_done(attr_set);
    }
    int per_cu = 3;
    for (int v = 0; v < 4; ++v) {          // the workspace query must size for the grid any ADD/MASK variant will use
        const int o = mil_resident_per_cu(pick(v & 1, v & 2), lds, 3, 256);
        per_cu = o < per_cu ? o : per_cu;
    }
    int grid = mil_num_cus() * per_cu;
    if (grid > ntiles) grid = ntiles;
    const size_t slab_elems = (size_t)(MT_PITCH + 1) * 16 * NTX * 16;
    const size_t bytes = slab_elems * grid * sizeof(float);
    if (query) { *need = bytes; return MIL_OK; }
    if (!ws || ws_bytes < bytes) return MIL_ERR_ARG;
    a.slab = (float*)ws;
    a.lds_w_off = 2 * A_PLANE; a.lds_x_off = 2 * A_PLANE + W_BYTES; a.lds_dump_off = 2 * A_PLANE + W_BYTES + 2 * X_PLANE;
    a.lds_a2_off = 0; a.lds_x2_off = 0;
    if (grid <= 0) return MIL_OK;
    const BwdFusedArgs a0 = a;
    for (int i0 = 0; i0 < n_total; i0 += chunk) {
        const int n = n_total - i0 < chunk ? n_total - i0 : chunk;
        BwdFusedArgs c = a0;
        c.g.n_img = n; c.g.n_groups = n;
        c.ntiles = n * c.g.tiles_y * c.g.tiles_x;
        const size_t px0 = (size_t)i0 * a.g.H * a.g.W, npx = (size_t)n * a.g.H * a.g.W;
        // BwdFusedArgs pointers are typed __bf16: byte offsets / 2
        c.dz = a0.dz + px0 * gpx / 2; c.x = a0.x + px0 * xpx / 2; c.dx = a0.dx + px0 * gpx / 2;
        if (a0.addend) c.addend = a0.addend + px0 * gpx / 2;
        c.z_bytes = (unsigned)(npx * gpx); c.g_bytes = (unsigned)(npx * gpx); c.x_bytes = (unsigned)(npx * xpx);
        c.gpx = gpx; c.xpx = xpx;
        const int gr = grid < c.ntiles ? grid : c.ntiles;
#ifdef MIL_STAMP
        static MilStampBuf sb;
        c.stamp = sb.get((size_t)gr * 4 * 10);
#endif
        hipLaunchKernelGGL(kern, dim3(gr), dim3(256), lds, stream, c);
        MIL_CHECK_LAUNCH();
#ifdef MIL_STAMP
        static const char* const ph[8] = {"barrier-top", "commit", "barrier-vis", "fetch-issue", "dgrad", "epilogue", "addend-issue", "wgrad"};
        sb.report(a0.addend ? "conv_bwd_fused16x3_kernel<ADD>" : "conv_bwd_fused16x3_kernel", gr, 4, 8, ph, stream);
#endif
        MilReduceJob j{};
        j.slab = (const float*)ws; j.nslab = gr; j.slab_elems = slab_elems; j.slab_cols = NTX * 16;
        j.dw = dw; j.db = db; j.cout = cout; j.cin = cin; j.ks = KS; j.kind = 1;
        j.cinp = 20; j.n_rows = KS * KS * 20; j.bias_off = (KS * KS / 2) * 20 * NTX * 16 + 23; j.bias_stride = NTX * 16;      // rows tap'*20 + co
        j.accumulate = (i0 > 0) ? 1 : accumulate;
        mil_reduce_or_defer(j, stream, /*may_defer=*/chunk >= n_total);
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}

static int bwd_fused_entry(const void* dz, const void* wpack, const void* x, const void* addend, void* dx, float* dw,
                           float* db, void* ws, size_t ws_bytes, int n_img, int H, int W, int cout, int cin, int ks,
                           int pad, int apply_mask, int accumulate, float slope, int dtype, bool query, size_t* need, void* stream) {
    const bool dense_grads = dtype == MIL_DT_BF16_DGRAD || dtype == MIL_DT_F32S_DGRAD;
    if ((dtype != MIL_DT_BF16 && !dense_grads && dtype != MIL_DT_F32S) || ks != 3 || pad != 1) return MIL_ERR_UNSUPPORTED;
    if (n_img <= 0 || H <= 0 || W <= 0) return MIL_ERR_ARG;
    if (slope <= 0.f || slope >= 1.f) return MIL_ERR_UNSUPPORTED;
    BwdFusedArgs a{};
    a.dz = (const __bf16*)dz; a.w = (const __bf16*)wpack; a.x = (const __bf16*)x; a.addend = (const __bf16*)addend;
    a.dx = (__bf16*)dx; a.apply_mask = apply_mask; a.slope = slope;
    a.g.n_img = n_img; a.g.H = H; a.g.W = W; a.g.Ho = H; a.g.Wo = W; a.g.ks = ks; a.g.stride = 1; a.g.pad = pad; a.g.zins = 0;
    const int czp = mil_cpad(cout), cxp = mil_cpad(cin);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#ifndef MIL_BWD24_WAVES
#define MIL_BWD24_WAVES 8       // measured in the model: 397/428/401 us per launch with 4 waves per workgroup (two per SIMD), 356/397/371 us with 8 (four per SIMD, 122-128 VGPRs)
#endif
    if (dtype == MIL_DT_F32S || dtype == MIL_DT_F32S_DGRAD) {      // fp32 tensors, bf16x3 products: the 24-channel layers
        if (czp != 24 || cxp != 24) return MIL_ERR_UNSUPPORTED;
        // 16x16 tiles of one image: the compile-time-geometry kernel (two 4-wave workgroups per CU; dense or padded gradients)
        const int rc = run_bwd_fused16_x3(a, dw, db, ws, ws_bytes, cout, cin, accumulate, query, need, st, dense_grads);
        if (rc != MIL_ERR_UNSUPPORTED || dense_grads) return rc;          // only that kernel reads the dense layout
        return run_bwd_fused<F32S, 24, 2, 3, 8>(a, dw, db, ws, ws_bytes, cout, cin, accumulate, query, need, st);
    }
    if (czp == 24 && cxp == 24) return run_bwd_fused<BF16, 24, 2, 3, MIL_BWD24_WAVES>(a, dw, db, ws, ws_bytes, cout, cin, accumulate, query, need, st, dense_grads);
    if (dense_grads) return MIL_ERR_UNSUPPORTED;
    if (czp == 40 && cxp == 40) return run_bwd_fused<BF16, 40, 3, 3, 8>(a, dw, db, ws, ws_bytes, cout, cin, accumulate, query, need, st);
    if (czp == 64 && cxp == 64) return run_bwd_fused<BF16, 64, 4, 3, 8>(a, dw, db, ws, ws_bytes, cout, cin, accumulate, query, need, st);
    return MIL_ERR_UNSUPPORTED;
}

extern "C" int mil_conv_bwd_fused_workspace(size_t* bytes, int n_img, int H, int W, int cout, int cin, int ks, int pad,
                                            int dtype) {
    if (!bytes) return MIL_ERR_ARG;
    return bwd_fused_entry(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, n_img, H, W, cout, cin,
                           ks, pad, 0, 0, 0.1f, dtype, true, bytes, nullptr);
}

extern "C" int mil_conv_bwd_fused(const void* dz, const void* wpack_dgrad, const void* x, const void* addend, void* dx,
                                  float* dw, float* db, void* workspace, size_t workspace_bytes, int n_img, int H, int W,
                                  int cout, int cin, int ks, int pad, int apply_mask, int accumulate, float slope, int dtype,
                                  void* stream) {
    if (!dz || !wpack_dgrad || !x || !dx || !dw) return MIL_ERR_ARG;
    size_t need = 0;
    return bwd_fused_entry(dz, wpack_dgrad, x, addend, dx, dw, db, workspace, workspace_bytes, n_img, H, W, cout, cin, ks,
                           pad, apply_mask, accumulate, slope, dtype, false, &need, stream);
}
