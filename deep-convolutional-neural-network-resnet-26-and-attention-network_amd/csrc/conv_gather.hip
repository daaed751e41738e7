// Gather-GEMM convolution for the wide layers (alt_resnet.py widths 128/256/512: alt_resnet.py:24-33,35-66), bf16.
//
//   D[m][n] = sum over taps t, channels c of  Src[pix(m) + off(t)][c] * W[t][c][n]
//
// as ONE pipelined GEMM per workgroup (eight waves, one workgroup per CU): 256 output pixels (m) x 128 output channels (n),
// K-step = one tap x 64 input channels.  The operand images of a K-step are copied L2/HBM -> LDS by LDS-DMA (`buffer_load ...
// lds`, 16 B per lane: no staging registers, no ds_write pass) into rings of stage buffers; the copies of the next one or two
// K-steps stay in flight across the ONE barrier per K-step (counted `s_waitcnt vmcnt`, raw `s_barrier`).  Out-of-image pixels
// are out-of-range buffer offsets: the DMA writes zeros (the conv's zero padding).
//
// The tap list is data — (dy, dx, filter tap) triples with a source stride and an output stride/offset — so one kernel runs
// the stride-1 forward conv and data gradient, the stride-2 forward conv (source stride 2: 156/140/222 us on the channel-
// blocked kernel of conv_wide.hip -> 72/61/47 us), the 1x1 projections, and the stride-2 data gradient as four parity
// classes of output pixels in one launch (blockIdx.z), each with only the 1, 2, 2 or 4 taps of 9 that reach it instead of
// multiplying inserted zeros.
//
// What was measured on the way (256 tiles of 256x256, 3x3 stride 1, 128 ch @32x32 / 256 @16x16 / 512 @8x8; TFLOP/s forward):
//   conv_wide.hip (128 px x 64 ch tiles, two workgroups per CU, register prefetch, barrier pair per chunk)      817 / 886 / 735
//   this kernel, every K-step gathering its own 256 pixel rows (48 KB), copies issued in one burst               715 / 792 / 882
//     ablations at 512 ch: MFMAs + barrier only 1250-1350 (the ceiling of 32 MFMAs per wave and barrier), with the 16
//     fragment reads per K-step in two bursts 970, with the six copies per wave as well 850: an LDS-DMA instruction costs
//     its wave 60-180 cycles of issue, and bursts cost BOTH waves of a SIMD because the barrier keeps them in phase
//   + halo-resident form (one halo image per 64-channel chunk read at nine row offsets: 20.6 instead of 48 KB per K-step),
//     fragment reads of the next half K-step and the copies interleaved between the MFMAs (sched_group_barrier)  766 / 902 / 921
//   two wave groups staggered by one barrier, four barriers per K-step (reads of one group under the MFMAs of the other)
//     559 / 605 / 654: slower — kept out; sixteen waves per workgroup (64 x 32 wave tiles): 722 / 798 / 889 — kept as an A/B knob
//   + the nine taps of a chunk unrolled, the 36 swizzled fragment offsets of a lane precomputed, chunk pairs unrolled so the
//     halo buffer offset is a ds_read immediate (the loop form spent ~45 of its ~65 vector instructions per K-step beside
//     the 32 MFMAs on that address arithmetic: the kernel was instruction-issue-bound)                               913 / 1061 / 1063
//   + three ring slots (8x8-pixel tiles): the barrier in the middle of the K-step, a filter image gets a whole K-step to land  896 / 1037 / 1223
//   a persistent form (several tiles per workgroup, next tile's copies under the epilogue): 256 VGPRs + ~100 spilled — not kept
// Data gradients 705/775/709 -> 811/1052/1253.  MFMA-busy cycles of the CU-busy cycles: profiles/r03_sq_counters_alt.txt.
//
// LDS images are [row][64 ch] bf16 with the 16-byte slot index XOR-ed by (row >> 1) & 7: the sixteen lanes of a
// ds_read_b128 group then fall on sixteen different 16-byte slots of the 256-byte bank row.  LDS-DMA writes lanes linearly,
// so the swizzle is applied to the SOURCE address (pixel rows) or baked into the packed filter images.
#include "pf_common.cuh"
#include <cstdlib>
#include <type_traits>

#define GC_BM 256
#define GC_BN 128
#define GC_BK 64
#define GC_STAGES 3
#define GC_A_BYTES (GC_BM * GC_BK * 2)         // 32 KB
#define GC_B_BYTES (GC_BN * GC_BK * 2)         // 16 KB
#define GC_STAGE_BYTES (GC_A_BYTES + GC_B_BYTES)
#define GC_LDS_BYTES (GC_STAGES * GC_STAGE_BYTES)      // 144 KB
#define GC_MAX_TAPS 9

struct GConvArgs {
    const __bf16* x;            // source [n_img, Hs, Ws, cin]
    const __bf16* w;            // filter images [cout/128][ktaps][cin/64][128 rows][64] (swizzled), mil_gconv_pack_weights
    const __bf16* res;          // added before the activation, or null          [n_img, Hout, Wout, cout]
    const __bf16* act;          // multiplies by lrelu'(act) after it, or null   [n_img, Hout, Wout, cout]
    __bf16* y;                  // [n_img, Hout, Wout, cout]
    int n_img, Hs, Ws, cin;
    int Hout, Wout, cout;
    int os;                     // output pixel of grid point (gy, gx) of class z = (gy*os + c_oy[z], gx*os + c_ox[z])
    int ss;                     // source pixel of grid point (gy, gx), tap t = (gy*ss + dy[t], gx*ss + dx[t])
    int tw_log2, th_log2, ti_log2, tiles_x, tiles_y;
    int ktaps;                  // taps in the packed filter
    // blockIdx.z = class of output pixels: the whole output (one class), or the four parities of a stride-2 data gradient,
    // each with its own grid extent, output offset and tap list ((dy + 8) << 16 | (dx + 8) << 8 | filter tap)
    int ncls;
    int c_ntaps[4], c_hg[4], c_wg[4], c_oy[4], c_ox[4];
    int c_tap[4][GC_MAX_TAPS];
    int apply_relu;
    float slope;
    unsigned x_bytes, w_bytes, y_bytes;
};

template <int N>
__device__ __forceinline__ void gc_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#define GC_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// HD = 0: every K-step copies its own 256 gathered pixel rows (any tap list, any source stride): 6 copies per wave and K-step
//         into a ring of three 48 KB stages.
// HD > 0: the canonical 3x3 stride-1 tap list.  The K-steps run chunk-major (nine taps of one 64-channel chunk after another)
//         on ONE halo image per chunk — (th+2) x (tw+2) pixel rows per image of the tile, copied once (HD 64-row pieces per
//         wave) and read at nine row offsets — double-buffered: the next chunk's halo arrives one piece per wave during the
//         first HD K-steps of the current chunk.  Per K-step the copy traffic falls from 48 KB to 16 KB of filter + 4.6 KB of
//         halo.  HD = 6 (16x16-pixel tiles) leaves room for a ring of four filter images (filter image s+3 is requested during
//         K-step s: two K-steps to land), HD = 7 (8x8 pixels of four images) for three.
// The loop bodies are branch-free — copies that would run past the end get an out-of-range offset and land as zeros in a slot
// nobody reads again — so that reads, copies and MFMAs of a K-step sit in ONE basic block and can be interleaved.
// NW = waves per workgroup (8 or 16; one workgroup per CU either way).  A bare loop of 16x16x32 bf16 MFMAs on random operands
// with one barrier per 32 MFMAs per wave runs at 1.56 PFLOP/s with two waves per SIMD and at 1.83 with four (no barrier: 1.79 /
// 1.81; one wave per SIMD: 1.13): with sixteen waves the wave tile is 64 pixels x 32 channels (16 MFMAs, 12 fragment reads and
// at most 2 copies per wave and K-step) and the scheduler has four waves per SIMD to cover reads, copies and barriers with.
// Measured in this kernel: 722/798/889 TFLOP/s forward with sixteen waves against 749/866/926 with eight (stride-2 shapes equal):
// the barrier structure is not what holds it, eight stays the default (MIL_GCONV_WAVES=16 for A/B runs).
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gconv_kernel(GConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int NTW = 32 / NW;                             // 16-channel tiles per wave: 4 (wave tile 64 x 64) or 2 (64 x 32)
    constexpr int NBP = 16 / NW;                             // 1 KB pieces of a filter image per wave: 2 or 1
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;                 // wave grid: 4 (pixels) x NW/4 (channels)
    const int nb = blockIdx.y, cls = blockIdx.z;
    const int nchunks = a.cin / GC_BK;
    const int ntaps = a.c_ntaps[cls], Hg = a.c_hg[cls], Wg = a.c_wg[cls], oy_off = a.c_oy[cls], ox_off = a.c_ox[cls];
    const int KT = ntaps * nchunks;

    // ---- tile origin ------------------------------------------------------------------------------------------
    int tile = blockIdx.x;
    const int tx0 = tile % a.tiles_x; tile /= a.tiles_x;
    const int ty0 = tile % a.tiles_y; const int grp = tile / a.tiles_y;
    const int img0 = grp << a.ti_log2, gy0 = ty0 << a.th_log2, gx0 = tx0 << a.tw_log2;
    const int tw_mask = (1 << a.tw_log2) - 1, th_mask = (1 << a.th_log2) - 1;

    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = mil_rsrc(a.w, a.w_bytes);
    const int row_bytes = a.cin * 2;
    const int b_voff = lane * 16;

    f32x4_t acc[NTW][4];                                     // [channel tile][pixel tile]: D rows = channels, columns = pixels
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- K-step schedule --------------------------------------------------------------------------------------------------
    // Two fragment sets (the k32 halves of a K-step).  While the MFMAs of one half run, the fragment reads of the next half —
    // the second half of this K-step, then the first half of the NEXT one, whose stage landed a K-step early — are issued
    // between them (sched_group_barrier pins that interleave: a read in an MFMA's shadow costs the wave a few cycles, a burst
    // of reads in front of the MFMAs costs every wave of the SIMD the whole burst, because the barrier keeps them in the same
    // phase).  The K-step's copies go out the same way.
    struct Half { bf16x8_t w[NTW], x[4]; };
    const int swq = gq ^ (r >> 1);                               // filter rows: (row >> 1) & 7 == r >> 1
    auto read_w = [&](Half& h, const char* lb, int k32) {
        const char* wb = lb + (wn * (NTW * 16) + r) * 128 + ((swq ^ (k32 * 4)) * 16);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) h.w[nt] = *reinterpret_cast<const bf16x8_t*>(wb + nt * 2048);
    };
    auto mfma_half = [&](const Half& h) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h.w[nt], h.x[mt], acc[nt][mt], 0, 0, 0);
    };
    // the schedule of one half: nv copies and NTW + 4 reads spread over NTW * 4 MFMAs
    auto pin_half = [&](int nv) {
#ifndef GC_NO_PIN
        constexpr int NR = NTW + 4, MPR = (NTW * 4) / NR;       // reads, MFMAs per read: 8 / 2 or 6 / 1
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);     // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // 1 DS read
            if (i < nv) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read (LDS-DMA copy)
        }
#endif
    };
    Half f0, f1;

    if constexpr (HD == 0) {
        // ---- this lane's pixel rows of every A image: piece q = wave*NAP + i covers rows q*8 .. q*8+7, slot = lane%8 -----------
        constexpr int NAP = 32 / NW;                             // 1 KB pieces of the pixel image per wave: 4 or 2
        int a_base[NAP], a_pos[NAP];
#pragma unroll
        for (int i = 0; i < NAP; ++i) {
            const int m = (wave * NAP + i) * 8 + (lane >> 3);
            const int gx = gx0 + (m & tw_mask), gy = gy0 + ((m >> a.tw_log2) & th_mask), img = img0 + (m >> (a.tw_log2 + a.th_log2));
            const bool ok = img < a.n_img && gy < Hg && gx < Wg;
            const int sy = gy * a.ss, sx = gx * a.ss;
            const int slot = (lane & 7) ^ ((m >> 1) & 7);        // logical 16-byte piece of the 128-byte row that lands in physical slot lane%8
            a_pos[i] = ok ? (sy << 16) | sx : (0x4000 << 16);    // an invalid row fails every bounds test below
            a_base[i] = ((img * a.Hs + sy) * a.Ws + sx) * row_bytes + slot * 16;
        }
        // the tap list lives in lanes 0..8 of one register (v_readlane by the wave-uniform tap index): a table look-up in memory
        // would be a scalar load, whose counter the fragment reads share
        const int tap_tab = a.c_tap[cls][lane < ntaps ? lane : 0];
        // copies d0..d1-1 of the NAP + NBP of stage (tap t, chunk c) into ring buffer `buf`: pixel-row pieces, then filter pieces;
        // !live: out-of-range offsets (zeros land in a buffer nobody reads again)
        constexpr int ND = NAP + NBP;
        auto issue = [&](int t, int c, bool live, int buf, int d0, int d1) {
            const int tp = __builtin_amdgcn_readlane(tap_tab, t < ntaps ? t : 0);
            const int dy = (tp >> 16) - 8, dx = ((tp >> 8) & 0xFF) - 8, tw = tp & 0xFF;
            const int delta = (dy * a.Ws + dx) * row_bytes + c * (GC_BK * 2);
            const int img_off = ((nb * a.ktaps + tw) * nchunks + c) * GC_B_BYTES + wave * (NBP * 1024);
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                if (d < d0 || d >= d1) continue;
                if (d < NAP) {
                    const int sy = (a_pos[d] >> 16) + dy, sx = (a_pos[d] & 0xFFFF) + dx;
                    const bool ok = live && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, GC_LDS_PTR(smem + buf * GC_STAGE_BYTES + (wave * NAP + d) * 1024), 16,
                                                             ok ? (unsigned)(a_base[d] + delta) : MIL_OOB, 0, 0, 0);
                } else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, GC_LDS_PTR(smem + buf * GC_STAGE_BYTES + GC_A_BYTES + (wave * NBP + d - NAP) * 1024), 16,
                                                             live ? (unsigned)b_voff : MIL_OOB, live ? img_off + (d - NAP) * 1024 : 0, 0, 0);
            }
        };
        auto read_half = [&](Half& h, int buf, int k32) {
            const char* stg = smem + buf * GC_STAGE_BYTES;
            read_w(h, stg + GC_A_BYTES, k32);
            const char* xb = stg + (wm * 64 + r) * 128 + ((swq ^ (k32 * 4)) * 16);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h.x[mt] = *reinterpret_cast<const bf16x8_t*>(xb + mt * 2048);
        };
        // K-step s = tap s / nchunks, chunk s % nchunks; (ti, ci) walks two K-steps ahead
        issue(0, 0, KT > 0, 0, 0, ND);
        int ti = nchunks > 1 ? 0 : 1, ci = nchunks > 1 ? 1 : 0;
        issue(ti, ci, KT > 1, 1, 0, ND);
        if (++ci == nchunks) { ci = 0; ++ti; }
        gc_wait_vm<ND>();                                        // stage 0 landed
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_half(f0, 0, 0);
        int buf = 0;
        for (int s = 0; s < KT; ++s) {
            gc_wait_vm<0>();                                     // stage s+1 (requested a K-step ago) has landed
            __builtin_amdgcn_s_barrier();                        // ... everywhere; everyone is done reading stage s-1
            __builtin_amdgcn_sched_barrier(0);
            const int nxt = buf == 2 ? 0 : buf + 1, prv = buf == 0 ? 2 : buf - 1;
            const bool live = s + 2 < KT;
            issue(ti, ci, live, prv, 0, ND / 2);
            read_half(f1, buf, 1);
            mfma_half(f0);
            pin_half(ND / 2);
            __builtin_amdgcn_sched_barrier(0);
            issue(ti, ci, live, prv, ND / 2, ND);
            read_half(f0, nxt, 0);
            mfma_half(f1);
            pin_half(ND - ND / 2);
            __builtin_amdgcn_sched_barrier(0);
            buf = nxt;
            if (++ci == nchunks) { ci = 0; ++ti; }
        }
        gc_wait_vm<0>();
    } else {
        // ---- halo images: [2][HD * 64 rows][128 B], then the filter ring [RS][16 KB] ----------------------------------
        constexpr int RS = HD <= 6 ? 4 : 3;
        constexpr int hbytes = HD * 8192;
        constexpr int HP = HD * 8;                               // 1 KB pieces (8 rows) of a halo image
        constexpr int NHP = (HP + NW - 1) / NW;                  // ... per wave: piece j*NW + wave for j < NHP
        char* ring = smem + 2 * hbytes;
        // HD fixes the tile: 6 = 16x16 pixels of one image (18x18 halo), 7 = 8x8 pixels of four images (10x10 each): compile-time
        // divisors for the halo tables
        constexpr int hw = HD == 6 ? 18 : 10, hh = hw;
        int h_off[NHP];
#pragma unroll
        for (int j = 0; j < NHP; ++j) {
            const int hr = (j * NW + wave) * 8 + (lane >> 3);
            const int ti = hr / (hh * hw), rem = hr - ti * (hh * hw), hy = rem / hw, hx = rem - hy * hw;
            const int img = img0 + ti, sy = gy0 - 1 + hy, sx = gx0 - 1 + hx;
            const bool ok = (ti >> a.ti_log2) == 0 && img < a.n_img && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
            const int slot = (lane & 7) ^ ((hr >> 1) & 7);
            h_off[j] = ok ? ((img * a.Hs + sy) * a.Ws + sx) * row_bytes + slot * 16 : (int)MIL_OOB;
        }
        auto piece_exists = [&](int j) { return j * NW + wave < HP; };      // wave-uniform (HD 7, sixteen waves: the last round is half empty)
        auto issue_halo = [&](int c, int j) {                    // piece j of this wave of chunk c's halo
            int off = h_off[0];                                  // a select chain the optimiser may not turn into a scratch array (its loads
#pragma unroll                                                  // would count on vmcnt)
            for (int q = 1; q < NHP; ++q) { off = j == q ? h_off[q] : off; asm volatile("" : "+v"(off)); }
            const unsigned o = off == (int)MIL_OOB ? MIL_OOB : (unsigned)(off + c * (GC_BK * 2));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, GC_LDS_PTR(smem + (c & 1) * hbytes + (j * NW + wave) * 1024), 16, o, 0, 0, 0);
        };
        // the canonical list: tap t = (t/3 - 1, t%3 - 1), filter tap t
        auto issue_b = [&](int c, int t, int slot, int j) {      // piece j of this wave of the filter image of chunk c, tap t; past the end: zeros
            const bool live = c < nchunks;
            const int img_off = ((nb * 9 + t) * nchunks + c) * GC_B_BYTES + (wave * NBP + j) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, GC_LDS_PTR(ring + slot * GC_B_BYTES + (wave * NBP + j) * 1024), 16,
                                                     live ? (unsigned)b_voff : MIL_OOB, live ? img_off : 0, 0, 0);
        };
        // pixel fragments: halo row of tile pixel m at tap (0,0) = ti*hh*hw + ty*hw + tx, + the tap's row offset; the slot
        // swizzle follows the row
        int prow[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = wm * 64 + mt * 16 + r;
            prow[mt] = (m >> (a.tw_log2 + a.th_log2)) * (hh * hw) + ((m >> a.tw_log2) & th_mask) * hw + (m & tw_mask);
        }
        auto read_half = [&](Half& h, int c, int toff, int slot, int k32) {
            read_w(h, ring + slot * GC_B_BYTES, k32);
            const char* halo = smem + (c & 1) * hbytes;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = prow[mt] + toff;
                h.x[mt] = *reinterpret_cast<const bf16x8_t*>(halo + row * 128 + ((((row >> 1) & 7) ^ gq ^ (k32 * 4)) * 16));
            }
        };
#pragma unroll
        for (int j = 0; j < NHP; ++j)
            if (piece_exists(j)) issue_halo(0, j);
#pragma unroll
        for (int st = 0; st < RS - 1; ++st)
#pragma unroll
            for (int j = 0; j < NBP; ++j) issue_b(0, st, st, j);      // nine taps per chunk >= RS - 1
        gc_wait_vm<NBP * (RS - 2)>();                            // filter image 0 and, older, the first halo have landed
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_half(f0, 0, 0, 0, 0);
#ifndef GC_UNROLL_TAPS
#define GC_UNROLL_TAPS 1
#endif
        if constexpr (NW == 8 && GC_UNROLL_TAPS) {
            // The nine taps of a chunk unrolled, with the 36 swizzled fragment offsets of this lane (tap x pixel tile) computed
            // once: the loop form spends ~45 of its ~65 vector instructions per K-step beside the 32 MFMAs on that arithmetic.
            // Tap-dependent decisions (which K-steps carry a halo piece, the counted waits) become compile-time.
            int xoff[9][4];
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int row = prow[mt] + (t / 3) * hw + (t % 3);
                    xoff[t][mt] = row * 128 + ((((row >> 1) & 7) ^ gq) * 16);
                }
            auto read_half_u = [&](Half& h, int hb, int t, int slot, int k32) {      // t: compile-time; hb: halo buffer offset
                read_w(h, ring + slot * GC_B_BYTES, k32);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) h.x[mt] = *reinterpret_cast<const bf16x8_t*>(smem + hb + (xoff[t][mt] ^ (k32 * 64)));
            };
            auto issue_halo_u = [&](int c, int j) {                // j: compile-time; past the last chunk: zeros into the idle buffer
                const unsigned o = (h_off[j] == (int)MIL_OOB || c >= nchunks) ? MIL_OOB : (unsigned)(h_off[j] + c * (GC_BK * 2));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, GC_LDS_PTR(smem + (c & 1) * hbytes + (j * NW + wave) * 1024), 16, o, 0, 0, 0);
            };
            // one chunk = nine K-steps; PAR = the chunk's halo buffer (compile-time, so the buffer offset folds into the
            // ds_read immediates)
            auto chunk = [&](int c, auto par) {
                constexpr int PAR = decltype(par)::value;
                constexpr int hb = PAR * hbytes, hbn = hbytes - hb;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    // K-step s = 9c + t in ring slot s % RS (9 = 1 mod 4, 0 mod 3).  Three ring slots: ONE barrier, in the middle of the
                    // K-step:
                    //   first half : request filter image s + RS - 1 into the slot K-step s-1 read (every wave finished those reads
                    //                before the previous mid-step barrier) and, t < NHP, a piece of the next chunk's halo; read the
                    //                second-half fragments of stage s; MFMAs of the first half
                    //   middle     : this wave's pieces of filter image s+1 (and everything older) have landed — counted wait over the
                    //                copies requested since — its fragment reads are back; barrier
                    //   second half: read the first-half fragments of stage s+1; MFMAs of the second half
                    // so a copy has a whole K-step (RS = 3) or two (RS = 4) to land; with the barrier at the top of the K-step the
                    // second piece of a filter image had half a K-step at RS = 3 and the wait stalled every K-step.
                    const int slot = RS == 4 ? ((c + t) & 3) : (t % 3);
                    const int nslot = RS == 4 ? ((c + t + 1) & 3) : ((t + 1) % 3);
                    const int islot = RS == 4 ? ((c + t + 3) & 3) : ((t + 2) % 3);      // the slot K-step s-1 read
                    const int tb = (t + RS - 1) % 9, cb = c + (t + RS - 1) / 9;         // filter image s + RS - 1
                    if constexpr (RS == 3) {
                        if (t < NHP) issue_halo_u(c + 1, t);
#pragma unroll
                        for (int j = 0; j < NBP; ++j) issue_b(cb, tb, islot, j);
                        read_half_u(f1, hb, t, slot, 1);
                        mfma_half(f0);
                        pin_half((t < NHP ? 1 : 0) + NBP);
                        __builtin_amdgcn_sched_barrier(0);
                        if (t < NHP) gc_wait_vm<NBP + 1>(); else gc_wait_vm<NBP>();      // younger: this K-step's halo piece and filter image s+2
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                        read_half_u(f0, t == 8 ? hbn : hb, t == 8 ? 0 : t + 1, nslot, 0);
                        mfma_half(f1);
                        pin_half(0);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        // four ring slots: a filter image is requested two K-steps before its first read, the barrier at the top of
                        // the K-step costs nothing in latency (the mid-step form measured 4-7 % slower here: its lgkmcnt(0))
                        if (t >= 1 && t - 1 < NHP) gc_wait_vm<NBP + 1>(); else gc_wait_vm<NBP>();
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                        if (t < NHP) issue_halo_u(c + 1, t);
                        issue_b(cb, tb, islot, 0);
                        read_half_u(f1, hb, t, slot, 1);
                        mfma_half(f0);
                        pin_half(t < NHP ? 2 : 1);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (NBP == 2) issue_b(cb, tb, islot, 1);
                        read_half_u(f0, t == 8 ? hbn : hb, t == 8 ? 0 : t + 1, nslot, 0);
                        mfma_half(f1);
                        pin_half(NBP == 2 ? 1 : 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            for (int c = 0; c < nchunks; c += 2) {
                chunk(c, std::integral_constant<int, 0>{});
                if (c + 1 < nchunks) chunk(c + 1, std::integral_constant<int, 1>{});
            }
        } else {
        int slot = 0, c = 0, t = 0, kx = 0, toff = 0;            // K-step s = 9c + t; toff = (t/3)*hw + t%3
            int cb = 0, tb = RS - 1;                                 // filter image s + RS - 1
            bool prev_h = false;
            for (int s = 0; s < KT; ++s) {
                // Filter image s+1 has landed (and, older than it, the next chunk's halo when s+1 starts a chunk): the copies
                // requested after it — the previous K-step's halo piece and, RS == 4, filter image s+2 — may still be in flight
                if constexpr (RS == 4) { if (prev_h) gc_wait_vm<NBP + 1>(); else gc_wait_vm<NBP>(); }
                else gc_wait_vm<0>();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                int nslot = slot + 1; if (nslot == RS) nslot = 0;
                int islot = slot - 1; if (islot < 0) islot = RS - 1;        // the slot K-step s-1 read
                // the next K-step: tap, row offset, chunk
                int tn = t + 1, kxn = kx + 1, toffn = toff + 1, cn = c;
                if (kxn == 3) { kxn = 0; toffn += hw - 3; }
                if (tn == 9) { tn = 0; toffn = 0; ++cn; }
                prev_h = t < NHP && piece_exists(t) && c + 1 < nchunks;
                if (prev_h) issue_halo(c + 1, t);
                __builtin_amdgcn_sched_barrier(0);
                issue_b(cb, tb, islot, 0);
                read_half(f1, c, toff, slot, 1);
                mfma_half(f0);
                pin_half(1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (NBP == 2) issue_b(cb, tb, islot, 1);
                read_half(f0, cn, toffn, nslot, 0);
                mfma_half(f1);
                pin_half(NBP == 2 ? 1 : 0);
                __builtin_amdgcn_sched_barrier(0);
                slot = nslot; t = tn; kx = kxn; toff = toffn; c = cn;
                if (++tb == 9) { tb = 0; ++cb; }
            }
    }
        gc_wait_vm<0>();
    }

    // ---- epilogue straight from the accumulators: lane = pixel column r of tile mt, 4 consecutive channels per tile --
    const __amdgpu_buffer_rsrc_t rs_y = mil_rsrc(a.y, a.y_bytes);
    const __amdgpu_buffer_rsrc_t rs_r = mil_rsrc(a.res, a.res ? a.y_bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_a = mil_rsrc(a.act, a.act ? a.y_bytes : 0);
    const int cbase = (nb * GC_BN + wn * (NTW * 16) + gq * 4) * 2;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = wm * 64 + mt * 16 + r;
        const int gx = gx0 + (m & tw_mask), gy = gy0 + ((m >> a.tw_log2) & th_mask), img = img0 + (m >> (a.tw_log2 + a.th_log2));
        const bool ok = img < a.n_img && gy < Hg && gx < Wg;
        const unsigned poff = ok ? (unsigned)(((img * a.Hout + gy * a.os + oy_off) * a.Wout + gx * a.os + ox_off) * (a.cout * 2) + cbase) : MIL_OOB;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const unsigned off = ok ? poff + nt * 32 : MIL_OOB;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[nt][mt][i];
            if (a.res) {
                const u32x2_t rv = __builtin_amdgcn_raw_buffer_load_b64(rs_r, off, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += __uint_as_float((i & 1) ? (rv[i >> 1] & 0xFFFF0000u) : (rv[i >> 1] << 16));
            }
            if (a.apply_relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = lrelu(v[i], a.slope);
            }
            if (a.act) {
                const u32x2_t av = __builtin_amdgcn_raw_buffer_load_b64(rs_a, off, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= lrelu_grad(__uint_as_float((i & 1) ? (av[i >> 1] & 0xFFFF0000u) : (av[i >> 1] << 16)), a.slope);
            }
            bf16x4_t o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, o), rs_y, off, 0, 0);
        }
    }
}

// fp32 master [Cout][Cin][k][k] -> filter images [n block][tap][chunk][128 rows][64] with the slot swizzle.
// mode 0: the conv as written (rows = Cout, k = Cin); mode 1: its data gradient (rows = Cin, k = Cout, taps flipped).
__global__ void gconv_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int cout, int cin, int ks, int mode, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int kk = ks * ks;
    const int cin_x = mode ? cout : cin;                       // contraction channels of the conv as executed
    const int nchunks = cin_x / GC_BK;
    const int e = idx & 7, pslot = (idx >> 3) & 7, row = (idx >> 6) & 127;
    size_t t = idx >> 13;
    const int c = (int)(t % nchunks); t /= nchunks;
    const int tap = (int)(t % kk);
    const int nb = (int)(t / kk);
    const int slot = pslot ^ ((row >> 1) & 7);
    const int kin = c * GC_BK + slot * 8 + e, nout = nb * GC_BN + row;
    float val;
    if (!mode) val = w[((size_t)nout * cin + kin) * kk + tap];
    else val = w[((size_t)kin * cin + nout) * kk + (kk - 1 - tap)];
    out[idx] = (__bf16)val;
}

extern "C" int mil_gconv_supported(int cin, int cout, int ks, int stride) {
    return (cin % GC_BK == 0 && cout % GC_BN == 0 && (ks == 1 || ks == 3) && (stride == 1 || stride == 2)) ? 1 : 0;
}

extern "C" int mil_gconv_packed_elems(size_t* elems, int cout, int cin, int ks, int mode) {
    if (!elems) return MIL_ERR_ARG;
    const int k_x = mode ? cout : cin, n_x = mode ? cin : cout;
    if (k_x % GC_BK || n_x % GC_BN || !(ks == 1 || ks == 3)) return MIL_ERR_UNSUPPORTED;
    *elems = (size_t)n_x * k_x * ks * ks;
    return MIL_OK;
}

extern "C" int mil_gconv_pack_weights(const float* w, void* wpack, int cout, int cin, int ks, int mode, void* stream) {
    size_t total = 0;
    if (!w || !wpack) return MIL_ERR_ARG;
    const int rc = mil_gconv_packed_elems(&total, cout, cin, ks, mode);
    if (rc != MIL_OK) return rc;
    hipLaunchKernelGGL(gconv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                       (__bf16*)wpack, cout, cin, ks, mode, total);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

static int gconv_launch(GConvArgs a, hipStream_t st) {
    int Hg = 0, Wg = 0;
    for (int z = 0; z < a.ncls; ++z) { if (a.c_hg[z] > Hg) Hg = a.c_hg[z]; if (a.c_wg[z] > Wg) Wg = a.c_wg[z]; }
    if (Hg <= 0 || Wg <= 0 || a.n_img <= 0) return MIL_OK;
    // 256 grid points per workgroup: 16x16 of one image, or 8x8 of four (maps of 8 pixels and below: 4x4 of sixteen)
    int tw, th;
    if (Wg > 8 || Hg > 8) {
        const long c16 = (long)((Wg + 15) >> 4) * ((Hg + 15) >> 4) * 4, c8 = (long)((Wg + 7) >> 3) * ((Hg + 7) >> 3);
        static const bool tile8 = [] { const char* e = mil_ab_env("MIL_GCONV_TILE8"); return e && e[0] == '1'; }();      // A/B runs
        tw = th = (c8 < c16 || (tile8 && c8 <= c16)) ? 3 : 4;
    } else if (Wg > 4 || Hg > 4) tw = th = 3;
    else tw = th = 2;
    a.tw_log2 = tw; a.th_log2 = th; a.ti_log2 = 8 - tw - th;
    a.tiles_x = (Wg + (1 << tw) - 1) >> tw; a.tiles_y = (Hg + (1 << th) - 1) >> th;
    const int groups = (a.n_img + (1 << a.ti_log2) - 1) >> a.ti_log2;
    static const bool no_halo = [] { const char* e = mil_ab_env("MIL_GCONV_NO_HALO"); return e && e[0] == '1'; }();      // A/B runs: per-K-step gather everywhere
    // the canonical 3x3 stride-1 tap list runs on the halo-resident form when two halo images fit beside the filter ring
    bool canon = a.ncls == 1 && a.ss == 1 && a.os == 1 && a.c_ntaps[0] == 9 && a.ktaps == 9 && !no_halo;
    for (int t = 0; t < 9 && canon; ++t) canon = a.c_tap[0][t] == (((t / 3 - 1 + 8) << 16) | ((t % 3 - 1 + 8) << 8) | t);
    // halo-resident forms exist for the two tile shapes the choice above produces on maps of 8 pixels and more: 16x16 pixels of one
    // image (324 halo rows: six 64-row pieces per wave-set) and 8x8 pixels of four images (400 rows: seven)
    const int hd = canon ? (tw == 4 ? 6 : (tw == 3 ? 7 : 0)) : 0;
    static const int nw = [] { const char* e = mil_ab_env("MIL_GCONV_WAVES"); return e && atoi(e) == 16 ? 16 : 8; }();      // A/B runs
    static std::atomic<unsigned long long> attr_set{0};      // per device
    if (mil_device_needs(attr_set)) {
        const void* ks[6] = {reinterpret_cast<const void*>(gconv_kernel<0, 8>), reinterpret_cast<const void*>(gconv_kernel<6, 8>),
                             reinterpret_cast<const void*>(gconv_kernel<7, 8>), reinterpret_cast<const void*>(gconv_kernel<0, 16>),
                             reinterpret_cast<const void*>(gconv_kernel<6, 16>), reinterpret_cast<const void*>(gconv_kernel<7, 16>)};
        for (const void* k : ks)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MIL_ERR_LAUNCH;
        mil_device_done(attr_set);
    }
    const dim3 grid(groups * a.tiles_y * a.tiles_x, a.cout / GC_BN, a.ncls);
    const int lds = hd == 6 ? 2 * 6 * 8192 + 4 * GC_B_BYTES : (hd == 7 ? 2 * 7 * 8192 + 3 * GC_B_BYTES : GC_LDS_BYTES);
    if (nw == 16) {
        if (hd == 6) hipLaunchKernelGGL((gconv_kernel<6, 16>), grid, dim3(1024), lds, st, a);
        else if (hd == 7) hipLaunchKernelGGL((gconv_kernel<7, 16>), grid, dim3(1024), lds, st, a);
        else hipLaunchKernelGGL((gconv_kernel<0, 16>), grid, dim3(1024), lds, st, a);
    } else {
        if (hd == 6) hipLaunchKernelGGL((gconv_kernel<6, 8>), grid, dim3(512), lds, st, a);
        else if (hd == 7) hipLaunchKernelGGL((gconv_kernel<7, 8>), grid, dim3(512), lds, st, a);
        else hipLaunchKernelGGL((gconv_kernel<0, 8>), grid, dim3(512), lds, st, a);
    }
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// y = mask(relu?(conv(x) + res?)) on the gather-GEMM kernel, bf16 NHWC.  `wpack` = mil_gconv_pack_weights(mode 0) of the
// conv's weight for transposed == 0 (y = conv(x): x [n,H,W,cin] -> y [n,Ho,Wo,cout]), mode 1 for transposed == 1 (the conv's
// data gradient: x is dz [n,H,W,cin_x] on the conv's OUTPUT grid, y is dx [n,Ho,Wo,cout_x] on its input grid; `stride` is the
// conv's stride, a stride-2 gradient runs as four parity-class launches).  cin/cout are the channel counts of x / y.
extern "C" int mil_gconv(const void* x, const void* wpack, const void* res, const void* act, void* y, int n_img, int H, int W,
                         int cin, int Ho, int Wo, int cout, int ks, int stride, int pad, int transposed, int apply_relu, float slope,
                         void* stream) {
    if (!x || !wpack || !y || n_img < 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return MIL_ERR_ARG;
    if (!mil_gconv_supported(cin, cout, ks, stride) || pad != ks / 2) return MIL_ERR_UNSUPPORTED;
    const size_t x_img = (size_t)H * W * cin * 2, y_img = (size_t)Ho * Wo * cout * 2;
    const size_t wb = (size_t)cin * cout * ks * ks * 2;
    if (x_img >= mil_buffer_limit() || y_img >= mil_buffer_limit() || H >= 0x4000 || W >= 0x4000) return MIL_ERR_UNSUPPORTED;
    if (n_img == 0) return MIL_OK;
    // every tensor is addressed with 32-bit offsets through a buffer descriptor: launches above its 2 GiB reach walk image chunks
    const int chunk = mil_imgs_under_2g(x_img > y_img ? x_img : y_img);
    if (n_img > chunk) {
        for (int i0 = 0; i0 < n_img; i0 += chunk) {
            const int n = n_img - i0 < chunk ? n_img - i0 : chunk;
            const int rc = mil_gconv(static_cast<const char*>(x) + i0 * x_img, wpack, res ? static_cast<const char*>(res) + i0 * y_img : nullptr,
                                     act ? static_cast<const char*>(act) + i0 * y_img : nullptr, static_cast<char*>(y) + i0 * y_img, n, H, W, cin, Ho,
                                     Wo, cout, ks, stride, pad, transposed, apply_relu, slope, stream);
            if (rc != MIL_OK) return rc;
        }
        return MIL_OK;
    }
    const size_t xb = n_img * x_img, yb = n_img * y_img;
    GConvArgs a{};
    a.x = (const __bf16*)x; a.w = (const __bf16*)wpack; a.res = (const __bf16*)res; a.act = (const __bf16*)act; a.y = (__bf16*)y;
    a.n_img = n_img; a.Hs = H; a.Ws = W; a.cin = cin; a.Hout = Ho; a.Wout = Wo; a.cout = cout;
    a.ktaps = ks * ks; a.apply_relu = apply_relu; a.slope = slope;
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.y_bytes = (unsigned)yb;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!transposed || stride == 1) {
        // forward (source stride = the conv's stride) or stride-1 gradient (the flipped filter is in the packing)
        if (!transposed && (Ho != (H + 2 * pad - ks) / stride + 1 || Wo != (W + 2 * pad - ks) / stride + 1)) return MIL_ERR_ARG;
        if (transposed && (Ho != H || Wo != W)) return MIL_ERR_ARG;
        a.ncls = 1; a.os = 1; a.ss = transposed ? 1 : stride;
        a.c_hg[0] = Ho; a.c_wg[0] = Wo; a.c_oy[0] = a.c_ox[0] = 0; a.c_ntaps[0] = ks * ks;
        for (int t = 0; t < ks * ks; ++t) a.c_tap[0][t] = ((t / ks - pad + 8) << 16) | ((t % ks - pad + 8) << 8) | t;
        return gconv_launch(a, st);
    }
    // stride-2 data gradient: output pixel (2j + py, 2i + px) receives dz[(y + pad - ky) / 2] for the ky with y + pad - ky even:
    // four classes of output pixels in one launch, each with the 1, 2, 2 or 4 taps (of 9) that reach it
    if (H != (Ho + 2 * pad - ks) / 2 + 1 || W != (Wo + 2 * pad - ks) / 2 + 1) return MIL_ERR_ARG;
    a.ncls = 4; a.os = 2; a.ss = 1;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int z = py * 2 + px;
            a.c_hg[z] = (Ho - py + 1) / 2; a.c_wg[z] = (Wo - px + 1) / 2; a.c_oy[z] = py; a.c_ox[z] = px;
            int n = 0;
            for (int ky = 0; ky < ks; ++ky)
                for (int kx = 0; kx < ks; ++kx) {
                    const int ny = py + pad - ky, nx = px + pad - kx;        // even and >= -(ks - 1 - pad) when the tap reaches the class
                    if ((ny & 1) || (nx & 1)) continue;
                    // mode-1 packing stores forward tap q at index kk-1-q
                    a.c_tap[z][n++] = ((ny / 2 + 8) << 16) | ((nx / 2 + 8) << 8) | (ks * ks - 1 - (ky * ks + kx));
                }
            a.c_ntaps[z] = n;
        }
    return gconv_launch(a, st);
    return MIL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the wide layers in the same gather form:  dW[co][ci][tap] = sum over output pixels q of
// dz[q][co] * x[src(q) + off(tap)][ci], a GEMM whose K dimension is the pixels.  One workgroup owns one (128 output
// channels) x (64 input channels) x (all taps) block of dW — 36 accumulator tiles per wave — and a strided share of the
// pixel stages; a stage is 32 consecutive output pixels: their dz rows (contiguous memory) and, per tap, their 32 gathered x
// rows, copied by LDS-DMA into a ring of three.  Any stride and tap list (no halo): this is the form the stride-2 3x3 and the
// 1x1 projection gradients run on, which the pipelined kernel of conv_wide.hip (stride 1 only) left on its synchronous
// sibling.  Both MFMA operands are pixel-major in LDS and k (pixel)-major in the MFMA: ds_read_b64_tr_b16, with the 32-byte
// chunk index XOR-ed by row bits so that the eight rows of a half-wave read fall on eight different bank spans.
// Partial sums leave as one fp32 slab [tap][128][64] per workgroup; gwgrad_reduce_kernel adds them in fixed order.
#define GW_P 32
#define GW_Z_BYTES (GW_P * GC_BN * 2)          // 8 KB
#define GW_X_BYTES (GW_P * GC_BK * 2)          // 4 KB per tap
#define GW_STAGE_BYTES (GW_Z_BYTES + GC_MAX_TAPS * GW_X_BYTES)     // 44 KB

struct GWgradArgs {
    const __bf16* x;            // [n_img, H, W, cin]
    const __bf16* dz;           // [n_img, Ho, Wo, cout]
    float* slab;                // [pairs][gx][ntaps][128][64]
    int n_img, H, W, cin, Ho, Wo, cout;
    int stride, ntaps, nstages;
    int tap[GC_MAX_TAPS];       // (dy + 8) << 8 | (dx + 8)
    unsigned x_bytes, z_bytes;
};

__global__ __launch_bounds__(512, 1) void gwgrad_kernel(GWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 1, wc = wave >> 1;                 // wave grid: 2 (64 output channels) x 4 (16 input channels)
    const int nchunks = a.cin / GC_BK;
    const int cb = blockIdx.y / nchunks, ch = blockIdx.y - cb * nchunks;
    const int ntaps = a.ntaps;
    const int Q = a.n_img * a.Ho * a.Wo;
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rs_z = mil_rsrc(a.dz, a.z_bytes);

    // copies of this wave: one dz piece (rows 4*wave .. +3, 16 slots of 16 B each) and the x pieces j = wave + 8*i of the
    // 4*ntaps (tap j/4, rows (j%4)*8 .. +7, 8 slots each): j%4 = wave%4 for every i, so a lane gathers ONE pixel per stage
    const int zrow = wave * 4 + (lane >> 4);
    const int zslot = (lane & 15) ^ ((((zrow & 3) | (((zrow >> 3) & 1) << 2))) << 1);
    const int xrow = (wave & 3) * 8 + (lane >> 3);
    const int xslot = (lane & 7) ^ (((((xrow >> 1) & 1) | (((xrow >> 3) & 1) << 1))) << 1);
    const int nx = (4 * ntaps - wave + 7) / 8;               // x pieces of this wave: 4 or 5 for nine taps, 0 or 1 for one
    const int tap_tab = a.tap[lane < ntaps ? lane : 0];

    auto issue = [&](int st, int buf, bool live) {
        char* stg = smem + buf * GW_STAGE_BYTES;
        const int q0 = st * GW_P;
        {
            const int q = q0 + zrow;
            const bool ok = live && q < Q;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_z, GC_LDS_PTR(stg + wave * 1024), 16,
                                                     ok ? (unsigned)(q * (a.cout * 2) + cb * (GC_BN * 2) + zslot * 16) : MIL_OOB, 0, 0, 0);
        }
        const int q = q0 + xrow;
        const int img = q / (a.Ho * a.Wo), rem = q - img * (a.Ho * a.Wo), oy = rem / a.Wo, ox = rem - oy * a.Wo;
        const bool qok = live && q < Q;
        const int sy0 = oy * a.stride, sx0 = ox * a.stride;
        const int base = ((img * a.H + sy0) * a.W + sx0) * (a.cin * 2) + ch * (GC_BK * 2) + xslot * 16;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i < nx) {                                        // wave-uniform
                const int j = wave + 8 * i;
                const int tp = __builtin_amdgcn_readlane(tap_tab, j >> 2);
                const int dy = (tp >> 8) - 8, dx = (tp & 0xFF) - 8;
                const bool ok = qok && (unsigned)(sy0 + dy) < (unsigned)a.H && (unsigned)(sx0 + dx) < (unsigned)a.W;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, GC_LDS_PTR(stg + GW_Z_BYTES + j * 1024), 16,
                                                         ok ? (unsigned)(base + (dy * a.W + dx) * (a.cin * 2)) : MIL_OOB, 0, 0, 0);
            }
        }
    };
    auto wait_stage = [&](bool one_in_flight) {                // this wave's copies of the oldest stage in flight have landed
        const int n = one_in_flight ? 1 + nx : 0;
        if (n == 0) gc_wait_vm<0>(); else if (n == 1) gc_wait_vm<1>(); else if (n == 2) gc_wait_vm<2>();
        else if (n == 5) gc_wait_vm<5>(); else gc_wait_vm<6>();
    };

    f32x4_t acc[GC_MAX_TAPS][4];                             // [tap][output-channel tile]
#pragma unroll
    for (int t = 0; t < GC_MAX_TAPS; ++t)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[t][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: pixel rows 8*gq + q4 (and + 4), 8 bytes at chunk (tile) of the row, chunk swizzled by row bits
    const int q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int row0 = 8 * gq + q4, row1 = row0 + 4;
    auto zoff = [&](int row, int ct) { return row * 256 + ((((wr * 4 + ct) ^ ((row & 3) | (((row >> 3) & 1) << 2))) * 32) + p4 * 8); };
    auto xoff = [&](int row) { return row * 128 + (((wc ^ (((row >> 1) & 1) | (((row >> 3) & 1) << 1))) * 32) + p4 * 8); };
    const int x0 = xoff(row0), x1 = xoff(row1);

    // this workgroup's stages: blockIdx.x, blockIdx.x + gridDim.x, ...
    const int G = gridDim.x;
    int st = blockIdx.x;
    const int mine = st < a.nstages ? (a.nstages - st + G - 1) / G : 0;
    issue(st, 0, mine > 0);
    issue(st + G, 1, mine > 1);
    int buf = 0;
    for (int i = 0; i < mine; ++i) {
        wait_stage(i + 1 < mine);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(st + 2 * G, buf == 0 ? 2 : buf - 1, i + 2 < mine);
        const char* stg = smem + buf * GW_STAGE_BYTES;
        bf16x8_t zf[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) zf[ct] = mil_tr_pair(stg + zoff(row0, ct), stg + zoff(row1, ct));
#pragma unroll
        for (int t = 0; t < GC_MAX_TAPS; ++t) {
            if (t < ntaps) {                                     // wave-uniform
                const char* xt = stg + GW_Z_BYTES + t * GW_X_BYTES;
                const bf16x8_t xf = mil_tr_pair(xt + x0, xt + x1);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[ct], xf, acc[t][ct], 0, 0, 0);
            }
        }
        buf = buf == 2 ? 0 : buf + 1;
        st += G;
    }
    gc_wait_vm<0>();
    // slab [tap][128 co][64 ci]: D rows = output channels (ct*16 + gq*4 + e), columns = input channels (r)
    float* slab = a.slab + ((size_t)blockIdx.y * G + blockIdx.x) * ((size_t)ntaps * GC_BN * GC_BK);
#pragma unroll
    for (int t = 0; t < GC_MAX_TAPS; ++t) {
        if (t < ntaps) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    slab[((size_t)t * GC_BN + wr * 64 + ct * 16 + gq * 4 + e) * GC_BK + wc * 16 + r] = acc[t][ct][e];
        }
    }
}

// dW[cb*128 + co][ch*64 + ci][tap] (+)= sum over the pair's slabs, fixed order.  One thread owns four consecutive input
// channels (16-byte slab loads, eight slabs requested ahead of the adds that consume them in slab order).
__global__ void gwgrad_reduce_kernel(const float* __restrict__ slab, int nslab, int npairs, int nchunks, int kk, float* __restrict__ dw,
                                     int cin, int accumulate) {
    const size_t per_pair = (size_t)kk * GC_BN * GC_BK;
    const size_t idx = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (idx >= per_pair * npairs) return;
    const int pair = (int)(idx / per_pair);
    const int e = (int)(idx - (size_t)pair * per_pair);
    const int ci = e & (GC_BK - 1), co = (e / GC_BK) & (GC_BN - 1), tap = e / (GC_BK * GC_BN);
    const int cb = pair / nchunks, ch = pair - cb * nchunks;
    const float* p = slab + (size_t)pair * nslab * per_pair + e;
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + 8 <= nslab; i += 8) {
        f32x4_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4_t*>(p + (size_t)(i + k) * per_pair);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; i < nslab; ++i) s += *reinterpret_cast<const f32x4_t*>(p + (size_t)i * per_pair);
    float* q = dw + ((size_t)(cb * GC_BN + co) * cin + ch * GC_BK + ci) * kk + tap;
#pragma unroll
    for (int k = 0; k < 4; ++k) q[(size_t)k * kk] = accumulate ? q[(size_t)k * kk] + s[k] : s[k];
}

// Internal entry (called from mil_wide_wgrad / mil_wide_wgrad_workspace in conv_wide.hip): bf16, cin % 64 == 0, cout % 128 == 0.
// Returns MIL_ERR_UNSUPPORTED when the gather form does not take the shape (the caller then runs its own kernels).
int mil_gwgrad(const void* x, const void* dz, float* dw, void* ws, size_t ws_bytes, int n_img, int H, int W, int cin, int Ho, int Wo,
               int cout, int ks, int stride, int pad, int accumulate, bool query, size_t* need, hipStream_t st) {
    if (cin % GC_BK || cout % GC_BN || !(ks == 1 || ks == 3) || !(stride == 1 || stride == 2) || pad != ks / 2) return MIL_ERR_UNSUPPORTED;
    const size_t xb = (size_t)n_img * H * W * cin * 2, zb = (size_t)n_img * Ho * Wo * cout * 2;
    if (xb >= mil_buffer_limit() || zb >= mil_buffer_limit()) return MIL_ERR_UNSUPPORTED;      // the caller's kernels address with 64 bits
    const int kk = ks * ks, npairs = (cout / GC_BN) * (cin / GC_BK);
    const long Q = (long)n_img * Ho * Wo;
    const int nstages = (int)((Q + GW_P - 1) / GW_P);
    // one workgroup per CU in all (each writes a slab of kk x 32 KB): the pixel stages of a pair are split over gx of them
    int gx = mil_num_cus() / npairs;
    if (gx < 1) gx = 1;
    if (gx > nstages) gx = nstages;
    const size_t bytes = (size_t)npairs * gx * kk * GC_BN * GC_BK * sizeof(float);
    if (query) { *need = bytes; return MIL_OK; }
    if (!ws || ws_bytes < bytes) return MIL_ERR_ARG;
    if (Q == 0) return MIL_OK;
    GWgradArgs a{};
    a.x = (const __bf16*)x; a.dz = (const __bf16*)dz; a.slab = (float*)ws;
    a.n_img = n_img; a.H = H; a.W = W; a.cin = cin; a.Ho = Ho; a.Wo = Wo; a.cout = cout;
    a.stride = stride; a.ntaps = kk; a.nstages = nstages;
    for (int t = 0; t < kk; ++t) a.tap[t] = ((t / ks - pad + 8) << 8) | (t % ks - pad + 8);
    a.x_bytes = (unsigned)xb; a.z_bytes = (unsigned)zb;
    static std::atomic<unsigned long long> attr_set{0};      // per device
    if (mil_device_needs(attr_set)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gwgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * GW_STAGE_BYTES) != hipSuccess)
            return MIL_ERR_LAUNCH;
        mil_device_done(attr_set);
    }
    hipLaunchKernelGGL(gwgrad_kernel, dim3(gx, npairs), dim3(512), 3 * GW_STAGE_BYTES, st, a);
    MIL_CHECK_LAUNCH();
    const size_t total = (size_t)kk * GC_BN * GC_BK * npairs;
    hipLaunchKernelGGL(gwgrad_reduce_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, gx, npairs, cin / GC_BK, kk,
                       dw, cin, accumulate);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
