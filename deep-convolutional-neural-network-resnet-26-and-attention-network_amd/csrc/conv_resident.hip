// 3x3 stride-1 convolutions of the two LAST stages on their small maps, PIXEL-RESIDENT:
//   80 channels on 8x8 maps   (layer 4 at 256x256 tiles): EIGHT whole images per workgroup (8 x 10x10 records = 138 KB)
//   64 channels on 16x16 maps (layer 3 at 256x256 tiles): THREE whole images per workgroup (3 x 18x18 records = 137 KB)
// A workgroup keeps whole images (with their zero border) in LDS and streams the 72-115 KB filter past them — every wave
// reads its filter fragments straight from global memory (L1/L2: the packed layout IS fragment order, 1 KB per wave
// load, two k-steps ahead in registers, the fragment index in the scalar offset of the buffer load) — the opposite of
// conv_igemm_pf_kernel, which keeps the filter resident in LDS and streams 128/256-pixel tiles.  On these maps the
// activations are the small operand (21 / 67 MB per tensor) and the filter-resident form pays for it: on the 80-channel
// maps one 16-pixel row tile per wave means every MFMA reads its own filter fragment from LDS (0.9 MB of LDS reads per
// tile), four tiles per workgroup, two barriers each — 29.5 us per launch for 13 us of traffic.  Here a wave owns four
// (six) row tiles, so a filter fragment feeds four (six) MFMAs, the pixels are loaded in one linear, fully coalesced
// copy per group of images, and one conv needs NO barrier between that copy and its stores.  80 channels: one conv
// 29.5 -> 19-21 us per launch.  (Staging the filter through an LDS double buffer instead costs a barrier per two
// k-steps and measured 22.5-25 us.)
//
// Whole images also mean no halo exchange between workgroups, so a CHAIN of convolutions can run back to back on the
// resident tile: conv k's output (after its epilogue) overwrites the tile's interior and is conv k+1's input; a later
// conv's residual / mask operand may be an earlier conv's output (each lane re-reads exactly the bytes it stored itself).
// Two convs are a whole
// identity-shortcut block forward (nnBlocks.py:175-189: o1 = lrelu(convA(x)+b), y = lrelu(convB(o1)+b+x); the residual
// x is re-read from global memory, L2-hot) and a whole block's data-gradient chain (dmid = lrelu'(o1) * convB^T(dz),
// dx = lrelu'(x) * (convA^T(dmid) + dz)) in ONE launch each instead of two (80 channels: 30 / 34 us against 2 x 29.5);
// five are everything of the last stage behind its stride-2 entry convs (forward), or its whole data-gradient chain.
//
// Epilogue of either conv: out = mask( lrelu?( acc + bias? + res? ) ), mask(v) = v * (act > 0 ? 1 : slope) if act — the
// contract of mil_conv_igemm, same MFMA order as the generic kernels (bit-identical outputs).  bf16 only; every other
// shape takes the generic kernels.
#include "pf_common.cuh"
#include <cstdlib>

struct ResConv {
    const __bf16* w;        // packed fragments [KSTEPS][NT][64][8] (MIL_PACK_FWD or MIL_PACK_DGRAD)
    const float* bias;      // [C] or null
    const __bf16* res;      // [n,S,S,C] or null
    const __bf16* act;      // [n,S,S,C] or null
    __bf16* out;            // [n,S,S,C]
    int lrelu;
};
#define MIL_CHAIN_MAX 6
struct ResArgs {
    const __bf16* x;        // [n,S,S,C]
    ResConv conv[MIL_CHAIN_MAX];      // conv k reads conv k-1's output from the resident tile
    int nconv;
    int n_img, ngroups;
    unsigned bytes;         // n*S*S*C*2
    float slope;
};

#ifndef MIL_RES_WARM
#define MIL_RES_WARM 1          // touch the filters at kernel start (see the kernel)
#endif
#ifndef MIL_RES_BDEPTH
#define MIL_RES_BDEPTH 2        // k-steps of filter fragments in flight per wave
#endif

template <int C, int S, int IMGS>
struct ResGeom {
    static constexpr int CG = C / 8, NT = C / 16;
    static constexpr int PIX = mil_pix_pitch(C, 2);           // LDS record of a pixel (odd 16-byte-slot pitch)
    static constexpr int HS = S + 2;
    static constexpr int IMG = HS * HS * PIX;
    static constexpr int TILE = IMGS * IMG;
    static constexpr int KSTEPS = (9 * CG + 3) / 4;
    // The pixels of the group's images form ONE list (image-major); row tile t = pixels 16t .. 16t+15 of it.  Where S*S is a
    // multiple of 16 (8x8, 16x16 maps) a row tile lies in one image; on the live driver's maps (300x300 tiles: 19x19 and 10x10,
    // round 5) it may straddle two, and the last tiles are partly / wholly beyond the list: such lanes compute on the last real
    // pixel's record (finite) and store nothing.
    static constexpr int NPIX = IMGS * S * S;
    static constexpr int TILES = (NPIX + 15) / 16;
    static constexpr int MW = ((TILES + 7) / 8 + 1) / 2 * 2;  // row tiles per wave (even: the epilogue pairs them)
    static constexpr bool RAGGED = (S * S) % 16 != 0 || TILES != 8 * MW;
    static constexpr int NPIECE = NPIX * CG;                  // 16-byte pieces of a group of images
    static constexpr int NP = (NPIECE + 511) / 512;
    // a wave's row tiles are exactly one image's (80 channels on 8x8 maps: 4 and 4): no wave ever reads another wave's records,
    // so the two convs of a pair need no workgroup barrier between them
    static constexpr bool WAVE_IS_IMAGE = !RAGGED && (S * S / 16 == MW);
    static_assert(C % 16 == 0, "shape");
    static_assert(TILE <= 160 * 1024, "LDS");
};

template <int C, int S, int IMGS>
__global__ __launch_bounds__(512, 2) void conv_resident_kernel(ResArgs a) {
    using G = ResGeom<C, S, IMGS>;
    constexpr int CG = G::CG, NT = G::NT, PIX = G::PIX, HS = G::HS, IMG = G::IMG, KSTEPS = G::KSTEPS, MW = G::MW, NP = G::NP, NPIX = G::NPIX;
    extern __shared__ __attribute__((aligned(16))) char tile[];
    MIL_POISON(tile);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, a.bytes);

#if MIL_RES_WARM
    // ---- pull every conv's filter into L2 now: the k-step loops read it fragment by fragment two k-steps ahead, which
    // hides an L2 hit but not the HBM miss a filter packed 9 ms (30 GB of traffic) ago costs — every k-step's first touch
    // would stall all waves of all workgroups at once.  One dword per 128-byte line and thread, results never used.
    // (the results are consumed at the very end of the kernel: memory returns in order, so the first wait for pixel data
    // covers these loads and nothing waits for them on their own)
    unsigned sink = 0;
    for (int k = 0; k < a.nconv; ++k) {
        const __amdgpu_buffer_rsrc_t rs_wk = mil_rsrc(a.conv[k].w, KSTEPS * NT * 1024);
        for (int line = tid; line < KSTEPS * NT * 8; line += 512)
            sink ^= __builtin_amdgcn_raw_buffer_load_b32(rs_wk, (unsigned)line * 128u, 0, 0);
    }
#endif
    // ---- zero border records, once: the commits and the first conv's epilogue only ever write interiors ----------------
    {
        constexpr int NB = 4 * S + 4, PPR = PIX / 16;
        for (int idx = tid; idx < IMGS * NB * PPR; idx += 512) {
            const int bp = idx / PPR, j = idx - bp * PPR;
            const int im = bp / NB, b = bp - im * NB;
            int hy, hx;
            if (b < HS) { hy = 0; hx = b; }
            else if (b < 2 * HS) { hy = HS - 1; hx = b - HS; }
            else { hy = 1 + ((b - 2 * HS) >> 1); hx = ((b - 2 * HS) & 1) * (HS - 1); }
            *reinterpret_cast<u32x4_t*>(tile + im * IMG + (hy * HS + hx) * PIX + j * 16) = u32x4_t{0u, 0u, 0u, 0u};
        }
    }
    // this wave's row tiles t = wave*MW + m: pixels 16t + r of the group's pixel list; top-left tap record of lane r
    int pixbase[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        int P = (wave * MW + m) * 16 + r;
        if (G::RAGGED && P >= NPIX) P = NPIX - 1;           // beyond the list: any written record
        const int im = P / (S * S), p = P - im * (S * S);
        pixbase[m] = im * IMG + ((p / S) * HS + (p % S)) * PIX;
    }
    // epilogue: after the permlane swap between row tiles 2p and 2p+1 a lane holds channels 16*nt + 8*(gq>>1) .. +7 of
    // pixel r of row tile 2p + (gq&1)
    const int c_off = (gq >> 1) * 16;

    for (int grp = blockIdx.x; grp < a.ngroups; grp += gridDim.x) {
        const int img0 = grp * IMGS;
        if (grp != (int)blockIdx.x) __syncthreads();       // the previous group's last fragment reads are done
        // ---- the group's images: one linear copy of contiguous bytes (NP pieces per thread) ---------------------------
        {
            u32x4_t v[NP];
            const unsigned g0 = (unsigned)img0 * (unsigned)(S * S * C * 2);
#pragma unroll
            for (int i = 0; i < NP; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, g0 + (unsigned)(tid + 512 * i) * 16u, 0, 0);   // beyond n_img: zeros
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int idx = tid + 512 * i, pl = idx / CG, j = idx - pl * CG;
                const int im = pl / (S * S), p = pl - im * (S * S);
                if (!G::RAGGED || idx < G::NPIECE)
                    *reinterpret_cast<u32x4_t*>(tile + im * IMG + (((p / S) + 1) * HS + (p % S) + 1) * PIX + j * 16) = v[i];
            }
        }

        auto run_conv = [&](const ResConv& cv, bool to_lds, bool first) {
            const __amdgpu_buffer_rsrc_t rs_w = mil_rsrc(cv.w, KSTEPS * NT * 1024);
            const __amdgpu_buffer_rsrc_t rs_res = mil_rsrc(cv.res, cv.res ? a.bytes : 0);
            const __amdgpu_buffer_rsrc_t rs_act = mil_rsrc(cv.act, cv.act ? a.bytes : 0);
            const __amdgpu_buffer_rsrc_t rs_out = mil_rsrc(cv.out, a.bytes);
            f32x4_t acc[MW][NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4_t b;
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = cv.bias ? cv.bias[nt * 16 + gq * 4 + i] : 0.f;
#pragma unroll
                for (int m = 0; m < MW; ++m) acc[m][nt] = b;
            }
            constexpr int BD = MIL_RES_BDEPTH;
            Frag8<BF16> bq[BD + 1][NT], aq[2][MW];
            auto fetch_b = [&](int ks) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bq[ks % (BD + 1)][nt].v = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(lane * 16), (ks * NT + nt) * 1024, 0));      // fragment index in the scalar offset
            };
            // k-group q = 4*ks + gq = (tap, 8-channel group) -> byte offset of its 16 bytes from the top-left tap's record,
            // by arithmetic (a table of per-lane offsets would be hoisted over both convs and spill); a q beyond the ninth tap
            // (zero weights) reads the top-left record — always written, always finite (0 x NaN is NaN)
            auto off_ks = [&](int ks) {
                const int q = 4 * ks + gq;
                const int tap = q / CG, cg = q - tap * CG;
                const int ty = (tap * 11) >> 5, tx = tap - ty * 3;            // tap / 3 for tap < 10
                return q < 9 * CG ? (ty * HS + tx) * PIX + cg * 16 : 0;
            };
            auto fetch_a = [&](int ks) {
                const int off = off_ks(ks);
#pragma unroll
                for (int m = 0; m < MW; ++m) aq[ks & 1][m] = lds_frag<BF16>(tile + pixbase[m] + off);
            };
#pragma unroll
            for (int ks = 0; ks < BD; ++ks) fetch_b(ks);
            if (first || !G::WAVE_IS_IMAGE) __syncthreads();         // the pixel tile (or the first conv's output) is visible to every wave
            fetch_a(0);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                if (ks + BD < KSTEPS) fetch_b(ks + BD);
                if (ks + 1 < KSTEPS) fetch_a(ks + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MW; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mma8(bq[ks % (BD + 1)][nt], aq[ks & 1][m], acc[m][nt]);      // D[channel][pixel]
                __builtin_amdgcn_sched_barrier(0);
            }
            if (to_lds && !G::WAVE_IS_IMAGE) __syncthreads();         // every wave is past its last read of the tile this epilogue overwrites
            // ---- epilogue, 8 channels per lane ------------------------------------------------------------------------
#pragma unroll
            for (int p = 0; p < MW / 2; ++p) {
                const int P = (wave * MW + 2 * p + (gq & 1)) * 16 + r;
                const bool in_list = !G::RAGGED || P < NPIX;
                const int Pc = in_list ? P : NPIX - 1, im = Pc / (S * S), px = Pc - im * (S * S);
                const bool img_ok = in_list && img0 + im < a.n_img;
                const unsigned goff = img_ok ? (unsigned)(((img0 + im) * (S * S) + px) * (C * 2) + c_off) : MIL_OOB;
                u32x4_t rr[NT], ra[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (cv.res) rr[nt] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, goff == MIL_OOB ? MIL_OOB : goff + nt * 32, 0, 0);
                    if (cv.act) ra[nt] = __builtin_amdgcn_raw_buffer_load_b128(rs_act, goff == MIL_OOB ? MIL_OOB : goff + nt * 32, 0, 0);
                }
                const int loff = im * IMG + (((px / S) + 1) * HS + (px % S) + 1) * PIX + c_off;
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
                    if (cv.res) {
                        const bf16x8_t tt = __builtin_bit_cast(bf16x8_t, rr[nt]);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] += (float)tt[i];
                    }
                    if (cv.lrelu) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], v[i] * a.slope);      // 0 < slope < 1
                    }
                    if (cv.act) {
                        const bf16x8_t tt = __builtin_bit_cast(bf16x8_t, ra[nt]);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] *= ((float)tt[i] > 0.f ? 1.f : a.slope);
                    }
                    bf16x8_t ov;
#pragma unroll
                    for (int i = 0; i < 8; ++i) ov[i] = (__bf16)v[i];
                    const u32x4_t ou = __builtin_bit_cast(u32x4_t, ov);
                    __builtin_amdgcn_raw_buffer_store_b128(ou, rs_out, goff == MIL_OOB ? MIL_OOB : goff + nt * 32, 0, 0);
                    if (to_lds && in_list) *reinterpret_cast<u32x4_t*>(tile + loff + nt * 32) = img_ok ? ou : u32x4_t{0u, 0u, 0u, 0u};      // the next conv's input
                }
            }
        };
        // the chain: conv k's epilogue overwrites the tile interior with its output, conv k+1 reads it (a run-time loop: one
        // copy of the unrolled k-step body, and no conv's prologue can be hoisted into its predecessor's registers)
#pragma unroll 1
        for (int k = 0; k < a.nconv; ++k) run_conv(a.conv[k], k + 1 < a.nconv, k == 0);
    }
#if MIL_RES_WARM
    asm volatile("" :: "v"(sink));
#endif
}

#include "conv_resident_x3.cuh"

static bool mil_resident_enabled() {
    static const bool v = [] { const char* e = mil_ab_env("MIL_RESIDENT"); return !(e && e[0] == '0'); }();
    return v;
}

template <int C, int S, int IMGS>
static int launch_resident(ResArgs a, hipStream_t st) {
    using G = ResGeom<C, S, IMGS>;
    auto kern = conv_resident_kernel<C, S, IMGS>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::TILE) != hipSuccess)
        return MIL_ERR_LAUNCH;
    a.ngroups = (a.n_img + IMGS - 1) / IMGS;
    // one workgroup per CU is resident (137-138 KB of LDS): groups beyond that are walked by the same workgroups
    int grid = mil_num_cus();
    if (grid > a.ngroups) grid = a.ngroups;
    {   // MIL_RES_GRID_CAP (tests): fewer workgroups than image groups, so that a workgroup walks several groups — the
        // persistent loop (re-copy behind a barrier, partial last group) otherwise needs more than 2048 images to run at all
        const char* e = getenv("MIL_RES_GRID_CAP");
        const int cap = e ? atoi(e) : 0;
        if (cap > 0 && grid > cap) grid = cap;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::TILE, st, a);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// which (channels, map) pairs have a pixel-resident kernel
static int resident_dispatch(ResArgs a, int cp, int H, int W, hipStream_t st) {
    if ((size_t)a.n_img * H * W * cp * 2 >= ((size_t)1 << 31)) return MIL_ERR_UNSUPPORTED;
    a.bytes = (unsigned)((size_t)a.n_img * H * W * cp * 2);
    if (cp == 80 && H == 8 && W == 8) return launch_resident<80, 8, 8>(a, st);
    if (cp == 64 && H == 16 && W == 16) return launch_resident<64, 16, 3>(a, st);
    // the live driver's 300x300 tiles (gbm/classify_combined.py:412): 10x10 and 19x19 maps
    if (cp == 80 && H == 10 && W == 10) return launch_resident<80, 10, 5>(a, st);       // 500 pixels = 32 row tiles (31.25 real), 127 KB
    if (cp == 64 && H == 19 && W == 19) return launch_resident<64, 19, 2>(a, st);       // 722 pixels = 46 of 48 row-tile slots, 127 KB
    return MIL_ERR_UNSUPPORTED;
}

// One conv: the contract of mil_conv_igemm for (C -> C, 3x3, stride 1, pad 1, bf16) on the shapes above.  Returns
// MIL_ERR_UNSUPPORTED for any other shape (or MIL_RESIDENT=0): the caller runs the generic kernels.
int mil_resident_conv(const void* x, const void* wpack, const float* bias_pad, const void* res, const void* act, void* y, int n_img,
                      int H, int W, int cp, int apply_lrelu, float slope, hipStream_t st) {
    if (!mil_resident_enabled() || n_img <= 0 || slope < 0.f || slope >= 1.f) return MIL_ERR_UNSUPPORTED;
    // one conv at a time only where it wins: 80 channels (19-22 us against 29.5); on the 64-channel 16x16 maps the filter-resident
    // kernel is as fast (51-58 us against 53-62: three images per workgroup are 2.7 rounds of copy-then-compute without a
    // prefetch) and only the two-conv form pays (91 us against 2 x 52)
    if (cp != 80) return MIL_ERR_UNSUPPORTED;
    ResArgs a{};
    a.x = (const __bf16*)x; a.n_img = n_img; a.slope = slope; a.nconv = 1;
    a.conv[0].w = (const __bf16*)wpack; a.conv[0].bias = bias_pad; a.conv[0].res = (const __bf16*)res; a.conv[0].act = (const __bf16*)act;
    a.conv[0].out = (__bf16*)y; a.conv[0].lrelu = apply_lrelu;
    return resident_dispatch(a, cp, H, W, st);
}

// The same for MIL_DT_F32S (fp32 tensors, split products): 80 channels on 8x8 maps and 64 channels on 16x16 maps.
int mil_resident_conv_x3(const void* x, const void* wpack, const float* bias_pad, const void* res, const void* act, void* y, int n_img,
                         int H, int W, int cp, int apply_lrelu, float slope, hipStream_t st) {
    if (!mil_resident_enabled() || n_img <= 0 || slope < 0.f || slope >= 1.f) return MIL_ERR_UNSUPPORTED;
    ResArgsX3 a{};
    a.x = (const float*)x; a.n_img = n_img; a.slope = slope; a.nconv = 1;
    a.conv[0].w = (const char*)wpack; a.conv[0].bias = bias_pad; a.conv[0].res = (const float*)res; a.conv[0].act = (const float*)act;
    a.conv[0].out = (float*)y; a.conv[0].lrelu = apply_lrelu;
    return resident_dispatch_x3(a, cp, H, W, st);
}

// A chain of up to MIL_CHAIN_MAX convs on the resident tile (see the header): out_0 = epi_0(conv_0(x)),
// out_k = epi_k(conv_k(out_{k-1})).  A conv's res / act may be the output of an EARLIER conv of the same chain (each lane
// re-reads exactly the bytes it stored itself).
struct MilChainConv {            // mirrors the C-ABI struct of include/mil_hip.h
    const void* wpack; const float* bias; const void* res; const void* act; void* out; int lrelu; int pad_;
};
extern "C" int mil_conv_chain(const void* x, const MilChainConv* convs, int nconv, int n_img, int H, int W, int cp, float slope,
                              int dtype, void* stream) {
    if (!x || !convs || nconv < 1 || nconv > MIL_CHAIN_MAX || n_img < 0) return MIL_ERR_ARG;
    for (int k = 0; k < nconv; ++k) if (!convs[k].wpack || !convs[k].out) return MIL_ERR_ARG;
    if ((dtype != MIL_DT_BF16 && dtype != MIL_DT_F32S) || slope < 0.f || slope >= 1.f || !mil_resident_enabled()) return MIL_ERR_UNSUPPORTED;
    if (n_img == 0) return MIL_OK;
    if (dtype == MIL_DT_F32S) {
        ResArgsX3 b{};
        b.x = (const float*)x; b.n_img = n_img; b.slope = slope; b.nconv = nconv;
        for (int k = 0; k < nconv; ++k) {
            b.conv[k].w = (const char*)convs[k].wpack; b.conv[k].bias = convs[k].bias; b.conv[k].res = (const float*)convs[k].res;
            b.conv[k].act = (const float*)convs[k].act; b.conv[k].out = (float*)convs[k].out; b.conv[k].lrelu = convs[k].lrelu;
        }
        return resident_dispatch_x3(b, cp, H, W, reinterpret_cast<hipStream_t>(stream));
    }
    ResArgs a{};
    a.x = (const __bf16*)x; a.n_img = n_img; a.slope = slope; a.nconv = nconv;
    for (int k = 0; k < nconv; ++k) {
        a.conv[k].w = (const __bf16*)convs[k].wpack; a.conv[k].bias = convs[k].bias; a.conv[k].res = (const __bf16*)convs[k].res;
        a.conv[k].act = (const __bf16*)convs[k].act; a.conv[k].out = (__bf16*)convs[k].out; a.conv[k].lrelu = convs[k].lrelu;
    }
    return resident_dispatch(a, cp, H, W, reinterpret_cast<hipStream_t>(stream));
}

// Two convs (a block forward, or a block's data-gradient chain): mil_conv_chain with nconv = 2.
extern "C" int mil_conv_pair(const void* x, const void* wpackA, const float* biasA, const void* resA, const void* actA,
                             int lreluA, void* outA, const void* wpackB, const float* biasB, const void* resB,
                             const void* actB, int lreluB, void* outB, int n_img, int H, int W, int cp, float slope,
                             int dtype, void* stream) {
    if (!x || !wpackA || !wpackB || !outA || !outB || n_img < 0) return MIL_ERR_ARG;
    const MilChainConv c[2] = {{wpackA, biasA, resA, actA, outA, lreluA, 0}, {wpackB, biasB, resB, actB, outB, lreluB, 0}};
    return mil_conv_chain(x, c, 2, n_img, H, W, cp, slope, dtype, stream);
}
