// 3x3 stride-1 convolutions of the 80-channel stage on its 8x8 maps (layer 4 at 256x256 tiles), PIXEL-RESIDENT:
// a workgroup keeps EIGHT whole images (with their zero border: 8 x 10x10 pixel records = 138 KB) in LDS and streams the
// 115 KB filter past them — every wave reads its filter fragments straight from global memory (L1/L2: the packed layout
// IS fragment order, 1 KB per wave load, two k-steps ahead in registers) — the opposite of conv_igemm_pf_kernel, which
// keeps the filter resident in LDS and streams 128-pixel tiles.  On these maps the activations are the small operand (21 MB per
// tensor, 82 KB per CU) and the filter-resident form pays for it: one 16-pixel row tile per wave means every MFMA reads
// its own filter fragment from LDS (0.9 MB of LDS reads per tile), four tiles per workgroup, two barriers each —
// 29.5 us per launch for 13 us of traffic.  Here a wave owns one image = four row tiles, so a filter fragment feeds four
// MFMAs (4 LDS fragment reads + 5 global fragment loads per 20 MFMAs), the pixels are loaded once in one linear, fully
// coalesced copy, nothing is walked, and after the tile is in LDS there is NO barrier: a wave only ever touches its own
// image.  One conv: 29.5 -> 19-21 us per launch.  (Staging the filter through an LDS double buffer instead — R8_BGLOBAL=0
// — costs a barrier per two k-steps and measured 22.5-25 us.)
//
// Whole images also mean no halo exchange between workgroups, so TWO convolutions can run back to back on the resident
// tile: conv A's output (after its epilogue) overwrites the tile's interior and is conv B's input.  That is a whole
// identity-shortcut block forward (nnBlocks.py:175-189: o1 = lrelu(convA(x)+b), y = lrelu(convB(o1)+b+x); the residual
// x is re-read from global memory, L2-hot) and a whole block's data-gradient chain (dmid = lrelu'(o1) * convB^T(dz),
// dx = lrelu'(x) * (convA^T(dmid) + dz)) in ONE launch each instead of two: 30 / 34 us against 2 x 29.5.
//
// Epilogue of either conv: out = mask( lrelu?( acc + bias? + res? ) ), mask(v) = v * (act > 0 ? 1 : slope) if act — the
// contract of mil_conv_igemm.  bf16, 80 -> 80 channels, 8x8 maps only; everything else takes the generic kernels.
#include "pf_common.cuh"
#include <cstdlib>

struct Res80Conv {
    const __bf16* w;        // packed fragments [23][5][64][8] (MIL_PACK_FWD or MIL_PACK_DGRAD)
    const float* bias;      // [80] or null
    const __bf16* res;      // [n,8,8,80] or null
    const __bf16* act;      // [n,8,8,80] or null
    __bf16* out;            // [n,8,8,80]
    int lrelu;
};
struct Res80Args {
    const __bf16* x;        // [n,8,8,80]
    Res80Conv A, B;         // B unused when the kernel runs one conv
    int n_img;
    unsigned bytes;         // n*64*160
    float slope;
};

constexpr int R8_PIX = 176;                      // LDS record of a pixel: 160 B at an odd 16-byte-slot pitch
constexpr int R8_IMG = 100 * R8_PIX;             // 10x10 records per image
constexpr int R8_TILE = 8 * R8_IMG;              // 140800
constexpr int R8_KSTEPS = 23, R8_NT = 5;
constexpr int R8_KBYTES = R8_NT * 64 * 16;       // one k-step of filter fragments: 5120
constexpr int R8_STAGE = 2 * R8_KBYTES;          // two k-steps per stage
constexpr int R8_NSTAGE = (R8_KSTEPS + 1) / 2;   // 12 (the last stage holds one k-step)


#ifndef R8_BGLOBAL
#define R8_BGLOBAL 1            // 1: filter fragments straight from global memory; 0: staged through an LDS double buffer
#endif
#ifndef R8_BDEPTH
#define R8_BDEPTH 2
#endif
#ifndef R8_STAGE_AHEAD
#define R8_STAGE_AHEAD 1
#endif
constexpr int R8_LDS = R8_TILE + (R8_BGLOBAL ? 0 : 2 * R8_STAGE);   // 140800 (161280 with the staged filter) <= 163840
template <bool TWO>
__global__ __launch_bounds__(512, 2) void conv_res80_kernel(Res80Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tile = smem;
    char* ldsF = smem + R8_TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    const int img0 = blockIdx.x * 8;
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, a.bytes);

    // ---- the eight images: one linear copy of 81920 contiguous bytes (10 pieces per thread); zero border records ----
    {
        u32x4_t v[10];
        const unsigned g0 = (unsigned)img0 * (64 * 160);
#pragma unroll
        for (int i = 0; i < 10; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, g0 + (unsigned)(tid + 512 * i) * 16u, 0, 0);   // beyond n_img: zeros
        // border: 36 records per image, 11 pieces each
        for (int idx = tid; idx < 8 * 36 * 11; idx += 512) {
            const int bp = idx / 11, j = idx - bp * 11;
            const int im = bp / 36, b = bp - im * 36;
            int hy, hx;
            if (b < 10) { hy = 0; hx = b; }
            else if (b < 20) { hy = 9; hx = b - 10; }
            else { hy = 1 + ((b - 20) >> 1); hx = ((b - 20) & 1) * 9; }
            *reinterpret_cast<u32x4_t*>(tile + im * R8_IMG + (hy * 10 + hx) * R8_PIX + j * 16) = u32x4_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int idx = tid + 512 * i, pl = idx / 10, j = idx - pl * 10;
            const int im = pl >> 6, p = pl & 63;
            *reinterpret_cast<u32x4_t*>(tile + im * R8_IMG + (((p >> 3) + 1) * 10 + (p & 7) + 1) * R8_PIX + j * 16) = v[i];
        }
    }
    // this wave's image; row tile m = image rows 2m, 2m+1; lane r = pixel (2m + (r>>3), r&7)
    int pixbase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) pixbase[m] = wave * R8_IMG + ((2 * m + (r >> 3)) * 10 + (r & 7)) * R8_PIX;
    // epilogue: after the permlane swap between row tiles 2p and 2p+1 a lane holds channels 16*nt + 8*(gq>>1) .. +7 of
    // pixel tp = (2p + (gq&1))*16 + r of its image
    const int c_off = (gq >> 1) * 16;
    const bool img_ok = img0 + wave < a.n_img;

    auto run_conv = [&](const Res80Conv& cv, bool to_lds, bool first) {
        const __amdgpu_buffer_rsrc_t rs_w = mil_rsrc(cv.w, R8_KSTEPS * R8_KBYTES);
        const __amdgpu_buffer_rsrc_t rs_res = mil_rsrc(cv.res, cv.res ? a.bytes : 0);
        const __amdgpu_buffer_rsrc_t rs_act = mil_rsrc(cv.act, cv.act ? a.bytes : 0);
        const __amdgpu_buffer_rsrc_t rs_out = mil_rsrc(cv.out, a.bytes);
        f32x4_t acc[4][R8_NT];
#pragma unroll
        for (int nt = 0; nt < R8_NT; ++nt) {
            f32x4_t b;
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = cv.bias ? cv.bias[nt * 16 + gq * 4 + i] : 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m][nt] = b;
        }
#if R8_BGLOBAL
        // Filter fragments straight from global memory (L1/L2: the packed layout IS fragment order, 1 KB per wave load),
        // R8_BDEPTH k-steps ahead in registers: no LDS staging, no barrier in the loop — the eight waves (= eight images)
        // run independently.
        constexpr int BD = R8_BDEPTH;
        Frag8<BF16> bq[BD + 1][R8_NT], aq[2][4];
        auto fetch_b = [&](int ks) {
#pragma unroll
            for (int nt = 0; nt < R8_NT; ++nt)
                bq[ks % (BD + 1)][nt].v = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(lane * 16), (ks * R8_NT + nt) * 1024, 0));      // fragment index in the scalar offset
        };
        // k-group q = 4*ks + gq = (tap, 8-channel group) -> byte offset of its 16 bytes from the top-left tap's record, by
        // arithmetic (a table of 23 per-lane offsets would be hoisted over both convs and spill); q = 90, 91 (a tenth tap
        // that does not exist: zero weights) read the top-left record — always written, always finite (0 x NaN is NaN)
        auto off_ks = [&](int ks) {
            const int q = 4 * ks + gq;
            const int tap = (q * 205) >> 11, cg = q - tap * 10;          // q / 10 for q < 1029
            const int ty = (tap * 11) >> 5, tx = tap - ty * 3;            // tap / 3 for tap < 10
            return q < 90 ? (ty * 10 + tx) * R8_PIX + cg * 16 : 0;
        };
        auto fetch_a = [&](int ks) {
            const int off = off_ks(ks);
#pragma unroll
            for (int m = 0; m < 4; ++m) aq[ks & 1][m] = lds_frag<BF16>(tile + pixbase[m] + off);
        };
#pragma unroll
        for (int ks = 0; ks < BD; ++ks) fetch_b(ks);
        if (first) __syncthreads();             // the pixel tile is visible (the only barrier of the kernel)
        fetch_a(0);
#pragma unroll
        for (int ks = 0; ks < R8_KSTEPS; ++ks) {
            if (ks + BD < R8_KSTEPS) fetch_b(ks + BD);
            if (ks + 1 < R8_KSTEPS) fetch_a(ks + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < R8_NT; ++nt) acc[m][nt] = mma8(bq[ks % (BD + 1)][nt], aq[ks & 1][m], acc[m][nt]);      // D[channel][pixel]
            __builtin_amdgcn_sched_barrier(0);
        }
#else
        // filter stage pieces of this thread: piece tid and (tid < 128) piece 512 + tid of the 640 per stage; the loads run
        // R8_AHEAD stages ahead in registers, the LDS double buffer one stage ahead
        constexpr int R8_AHEAD = R8_STAGE_AHEAD;
        u32x4_t f0[R8_AHEAD], f1[R8_AHEAD];
        auto fetch_stage = [&](int s) {
            const unsigned o = (unsigned)s * R8_STAGE + (unsigned)tid * 16u;
            f0[s % R8_AHEAD] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, o, 0, 0);                    // beyond the 23rd k-step: zeros
            f1[s % R8_AHEAD] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, tid < 128 ? o + 8192u : MIL_OOB, 0, 0);
        };
        auto commit_stage = [&](int s) {
            char* dst = ldsF + (s & 1) * R8_STAGE + tid * 16;
            *reinterpret_cast<u32x4_t*>(dst) = f0[s % R8_AHEAD];
            if (tid < 128) *reinterpret_cast<u32x4_t*>(dst + 8192) = f1[s % R8_AHEAD];
        };
#pragma unroll
        for (int s = 0; s < R8_AHEAD; ++s) fetch_stage(s);
        commit_stage(0);
        __syncthreads();                        // stage 0 (and, for the first conv, the pixel tile) visible
#pragma unroll
        for (int s = 0; s < R8_NSTAGE; ++s) {
            if (s + R8_AHEAD < R8_NSTAGE) fetch_stage(s + R8_AHEAD);
            const char* wb = ldsF + (s & 1) * R8_STAGE;
            constexpr int NK_FULL = 2;
            const int nk = (2 * s + 1 < R8_KSTEPS) ? NK_FULL : 1;
            Frag8<BF16> bf[2][R8_NT], af[2][4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk >= nk) break;
                const int ks = 2 * s + kk;
                // k-group q = 4*ks + gq = (tap, 8-channel group): four compile-time candidates, one per lane group
                auto off_q = [](int q) { const int tap = q / 10, cg = q - tap * 10; return tap < 9 ? ((tap / 3) * 10 + tap % 3) * R8_PIX + cg * 16 : 0; };
                const int o0 = off_q(4 * ks), o1 = off_q(4 * ks + 1), o2 = off_q(4 * ks + 2), o3 = off_q(4 * ks + 3);
                const int off = gq == 0 ? o0 : gq == 1 ? o1 : gq == 2 ? o2 : o3;
#pragma unroll
                for (int nt = 0; nt < R8_NT; ++nt) bf[kk][nt] = lds_frag<BF16>(wb + kk * R8_KBYTES + (nt * 64 + lane) * 16);
#pragma unroll
                for (int m = 0; m < 4; ++m) af[kk][m] = lds_frag<BF16>(tile + pixbase[m] + off);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk >= nk) break;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nt = 0; nt < R8_NT; ++nt) acc[m][nt] = mma8(bf[kk][nt], af[kk][m], acc[m][nt]);      // D[channel][pixel]
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < R8_NSTAGE) commit_stage(s + 1);
            __syncthreads();
        }
#endif
        // ---- epilogue, 8 channels per lane ----------------------------------------------------------------------
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int tp = (2 * p + (gq & 1)) * 16 + r;
            const unsigned goff = img_ok ? (unsigned)(((img0 + wave) * 64 + tp) * 160 + c_off) : MIL_OOB;
            u32x4_t rr[R8_NT], ra[R8_NT];
#pragma unroll
            for (int nt = 0; nt < R8_NT; ++nt) {
                if (cv.res) rr[nt] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, goff + nt * 32, 0, 0);
                if (cv.act) ra[nt] = __builtin_amdgcn_raw_buffer_load_b128(rs_act, goff + nt * 32, 0, 0);
            }
            const int loff = wave * R8_IMG + (((tp >> 3) + 1) * 10 + (tp & 7) + 1) * R8_PIX + c_off;
#pragma unroll
            for (int nt = 0; nt < R8_NT; ++nt) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float lo = acc[2 * p][nt][i], hi = acc[2 * p + 1][nt][i];
                    if (i == 0) mil_swap16<true>(lo, hi); else mil_swap16<false>(lo, hi);
                    v[i] = lo;
                    v[4 + i] = hi;
                }
                if (cv.res) {
                    const bf16x8_t t = __builtin_bit_cast(bf16x8_t, rr[nt]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += (float)t[i];
                }
                if (cv.lrelu) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], v[i] * a.slope);      // 0 < slope < 1
                }
                if (cv.act) {
                    const bf16x8_t t = __builtin_bit_cast(bf16x8_t, ra[nt]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= ((float)t[i] > 0.f ? 1.f : a.slope);
                }
                bf16x8_t ov;
#pragma unroll
                for (int i = 0; i < 8; ++i) ov[i] = (__bf16)v[i];
                const u32x4_t ou = __builtin_bit_cast(u32x4_t, ov);
                __builtin_amdgcn_raw_buffer_store_b128(ou, rs_out, goff + nt * 32, 0, 0);
                // the next conv's input: only this wave reads its image's records, and it is past its last fragment read
                if (to_lds) *reinterpret_cast<u32x4_t*>(tile + loff + nt * 32) = img_ok ? ou : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
    };
    run_conv(a.A, TWO, true);
    // keep the second conv's prologue (its bias values, its first filter fragments) out of the first conv's registers: without
    // the compiler barrier its loads are hoisted to the top of the kernel and spill
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TWO) run_conv(a.B, false, false);
}

static bool mil_res80_enabled() {
    static const bool v = [] { const char* e = getenv("MIL_RES80"); return !(e && e[0] == '0'); }();
    return v;
}

static int launch_res80(const Res80Args& a, bool two, hipStream_t st) {
    auto kern = two ? conv_res80_kernel<true> : conv_res80_kernel<false>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, R8_LDS) != hipSuccess)
        return MIL_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((a.n_img + 7) / 8), dim3(512), R8_LDS, st, a);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// One conv: the contract of mil_conv_igemm for (80 -> 80, 3x3, stride 1, pad 1, 8x8 maps, bf16).  Returns
// MIL_ERR_UNSUPPORTED when the shape is not this one (or MIL_RES80=0): the caller runs the generic kernels.
int mil_res80_conv(const void* x, const void* wpack, const float* bias_pad, const void* res, const void* act, void* y, int n_img,
                   int H, int W, int apply_lrelu, float slope, hipStream_t st) {
    if (!mil_res80_enabled() || H != 8 || W != 8 || n_img <= 0 || slope < 0.f || slope >= 1.f) return MIL_ERR_UNSUPPORTED;
    if ((size_t)n_img * 64 * 160 >= ((size_t)1 << 31)) return MIL_ERR_UNSUPPORTED;
    Res80Args a{};
    a.x = (const __bf16*)x; a.n_img = n_img; a.bytes = (unsigned)((size_t)n_img * 64 * 160); a.slope = slope;
    a.A.w = (const __bf16*)wpack; a.A.bias = bias_pad; a.A.res = (const __bf16*)res; a.A.act = (const __bf16*)act;
    a.A.out = (__bf16*)y; a.A.lrelu = apply_lrelu;
    return launch_res80(a, false, st);
}

// Two convs back to back on the resident tile (see the header): outA = epilogueA(convA(x)), outB = epilogueB(convB(outA)).
extern "C" int mil_conv_pair80(const void* x, const void* wpackA, const float* biasA, const void* resA, const void* actA,
                               int lreluA, void* outA, const void* wpackB, const float* biasB, const void* resB,
                               const void* actB, int lreluB, void* outB, int n_img, int H, int W, int cp, float slope,
                               int dtype, void* stream) {
    if (!x || !wpackA || !wpackB || !outA || !outB || n_img < 0) return MIL_ERR_ARG;
    if (dtype != MIL_DT_BF16 || cp != 80 || H != 8 || W != 8 || slope < 0.f || slope >= 1.f || !mil_res80_enabled()) return MIL_ERR_UNSUPPORTED;
    if (n_img == 0) return MIL_OK;
    if ((size_t)n_img * 64 * 160 >= ((size_t)1 << 31)) return MIL_ERR_UNSUPPORTED;
    Res80Args a{};
    a.x = (const __bf16*)x; a.n_img = n_img; a.bytes = (unsigned)((size_t)n_img * 64 * 160); a.slope = slope;
    a.A.w = (const __bf16*)wpackA; a.A.bias = biasA; a.A.res = (const __bf16*)resA; a.A.act = (const __bf16*)actA;
    a.A.out = (__bf16*)outA; a.A.lrelu = lreluA;
    a.B.w = (const __bf16*)wpackB; a.B.bias = biasB; a.B.res = (const __bf16*)resB; a.B.act = (const __bf16*)actB;
    a.B.out = (__bf16*)outB; a.B.lrelu = lreluB;
    return launch_res80(a, true, reinterpret_cast<hipStream_t>(stream));
}
