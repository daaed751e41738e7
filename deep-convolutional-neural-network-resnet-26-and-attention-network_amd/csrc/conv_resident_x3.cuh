// Pixel-resident 3x3 stride-1 convolutions of the two last stages in SPLIT PRECISION (MIL_DT_F32S: fp32 tensors, bf16x3
// products) — the conv_resident_kernel design (conv_resident.hip) for operands twice as large.  Included by conv_resident.hip.
//
//   80 channels on 8x8 maps:   FOUR whole images per workgroup, hi and lo bf16 planes of 10x10 records each (141 KB of LDS)
//   64 channels on 16x16 maps: ONE whole image per workgroup (18x18 records per plane, 93 KB)
//
// Why this form for these layers: a [hi | lo] filter of 64 / 80 channels is 147 / 235 KB — it cannot stay in LDS next to any
// tile, so the filter-resident kernels fall back to column splits and K-chunking (103-180 us per conv for 22-55 us of matrix
// work, round 3).  Here the IMAGES stay in LDS and every wave streams the packed filter straight from L1/L2 into registers
// (2 KB per wave-load and column tile, the fragment index in the scalar offset of the buffer load); a chain of convolutions
// (a block forward, a stage's data-gradient chain) runs back to back on the resident images, each epilogue writing the next
// conv's input planes.  A lane of the D[channel][pixel] accumulators holds four consecutive channels of a pixel = one 16-byte
// fp32 piece, so the epilogue needs no lane exchange: residual / mask operands are loaded and the output is stored 16 bytes
// per (row tile, column tile) and lane.
#pragma once
#ifndef MIL_RESX3_BD
#define MIL_RESX3_BD 2
#endif

struct ResConvX3 {
    const char* w;          // packed MIL_DT_F32S fragments [KSTEPS][NT][64][32 B] (MIL_PACK_FWD or MIL_PACK_DGRAD)
    const float* bias;      // [C] or null
    const float* res;       // [n,S,S,C] or null
    const float* act;       // [n,S,S,C] or null
    float* out;             // [n,S,S,C]
    int lrelu;
};
struct ResArgsX3 {
    const float* x;         // [n,S,S,C]
    ResConvX3 conv[MIL_CHAIN_MAX];
    int nconv;
    int n_img, ngroups;
    unsigned bytes;         // n*S*S*C*4
    float slope;
};

template <int C, int S, int IMGS>
struct ResGeomX3 {
    static constexpr int CG = C / 8, NT = C / 16;
    static constexpr int PIX = mil_pix_pitch(C, 2);           // record of a pixel in ONE plane (odd 16-byte-slot pitch)
    static constexpr int HS = S + 2;
    static constexpr int IMG = HS * HS * PIX;
    static constexpr int PLANE = IMGS * IMG;                  // hi plane, then lo plane
    static constexpr int TILE = 2 * PLANE;
    static constexpr int KSTEPS = (9 * CG + 3) / 4;
    static constexpr int NPIX = IMGS * S * S;                 // the group's pixels as ONE list (see ResGeom, conv_resident.hip)
    static constexpr int TILES = (NPIX + 15) / 16;
    // The eight waves as WM row groups x WN column groups.  64 channels: 4 x 2 — a wave owns four row tiles and TWO of the four
    // column tiles, so a streamed 2 KB fragment pair feeds four MFMA triples instead of two: per k-step and CU 32 KB from L1
    // (512 cycles of its 64 B/clk) and 64 KB of pixel fragments from LDS (512 cycles) beside 768 cycles of MFMAs per SIMD; the
    // 8 x 1 arrangement asked L1 for 64 KB = 1024 cycles (matrix pipe 0.43 busy).  80 channels (five column tiles): 8 x 1.
    // (more than four row tiles per wave in the 4 x 2 arrangement — the 19x19 maps — would need 96 pixel-fragment registers: 8 x 1)
    static constexpr int WN = (NT % 2 == 0 && (TILES + 3) / 4 <= 4) ? 2 : 1, WM = 8 / WN, NTW = NT / WN;
    static constexpr int MW = (TILES + WM - 1) / WM;          // row tiles per wave
    static constexpr bool RAGGED = (S * S) % 16 != 0 || TILES != WM * MW;
    static constexpr int NPIECE = NPIX * (C / 4);             // 16-byte pieces (four fp32 channels) of a group of images
    static constexpr int NP = (NPIECE + 511) / 512;
    static_assert(C % 16 == 0, "shape");
    static_assert(TILE <= 160 * 1024, "LDS");
};

template <int C, int S, int IMGS>
__global__ __launch_bounds__(512, 2) void conv_resident_x3_kernel(ResArgsX3 a) {
    using G = ResGeomX3<C, S, IMGS>;
    constexpr int CG = G::CG, NT = G::NT, PIX = G::PIX, HS = G::HS, IMG = G::IMG, PLANE = G::PLANE, KSTEPS = G::KSTEPS, MW = G::MW, NP = G::NP, NPIX = G::NPIX;
    constexpr int WN = G::WN, NTW = G::NTW;
    extern __shared__ __attribute__((aligned(16))) char tile[];
    MIL_POISON(tile);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, gq = lane >> 4;
    const int wm = wave / WN, nt0 = (wave % WN) * NTW;       // row group, first column tile of this wave (wave-uniform)
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, a.bytes);

    // pull every conv's filter into L2 now (see conv_resident_kernel): one dword per 128-byte line and thread, never used
    unsigned sink = 0;
    for (int k = 0; k < a.nconv; ++k) {
        const __amdgpu_buffer_rsrc_t rs_wk = mil_rsrc(a.conv[k].w, KSTEPS * NT * 2048);
        for (int line = tid; line < KSTEPS * NT * 16; line += 512)
            sink ^= __builtin_amdgcn_raw_buffer_load_b32(rs_wk, (unsigned)line * 128u, 0, 0);
    }
    // zero border records of both planes, once: commits and epilogues only ever write interiors
    {
        constexpr int NB = 4 * S + 4, PPR = PIX / 16;
        for (int idx = tid; idx < 2 * IMGS * NB * PPR; idx += 512) {
            const int bp = idx / PPR, j = idx - bp * PPR;
            const int im = bp / NB, b = bp - im * NB;                 // im counts images of the hi plane, then of the lo plane
            int hy, hx;
            if (b < HS) { hy = 0; hx = b; }
            else if (b < 2 * HS) { hy = HS - 1; hx = b - HS; }
            else { hy = 1 + ((b - 2 * HS) >> 1); hx = ((b - 2 * HS) & 1) * (HS - 1); }
            *reinterpret_cast<u32x4_t*>(tile + im * IMG + (hy * HS + hx) * PIX + j * 16) = u32x4_t{0u, 0u, 0u, 0u};
        }
    }
    // this wave's row tiles t = wm*MW + m: pixels 16t + r of the group's pixel list
    int pixbase[MW];                 // top-left tap record of lane r's pixel (hi plane)
    int pixrec[MW];                  // the pixel's own record (interior), for the chain's write-back
    int pixglb[MW];                  // (image in group * S*S + pixel) of lane r's pixel; -1: beyond the list (ragged shapes)
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        const int P = (wm * MW + m) * 16 + r;
        const int Pc = (G::RAGGED && P >= NPIX) ? NPIX - 1 : P;
        const int im = Pc / (S * S), p = Pc - im * (S * S);
        pixbase[m] = im * IMG + ((p / S) * HS + (p % S)) * PIX;
        pixrec[m] = im * IMG + (((p / S) + 1) * HS + (p % S) + 1) * PIX;
        pixglb[m] = (G::RAGGED && P >= NPIX) ? -1 : Pc;
    }
    auto split4 = [](const f32x4_t& v, u32x2_t& hi, u32x2_t& lo) {
        bf16x4_t h, l;
        mil_split4(v, h, l);
        hi = __builtin_bit_cast(u32x2_t, h);
        lo = __builtin_bit_cast(u32x2_t, l);
    };

    for (int grp = blockIdx.x; grp < a.ngroups; grp += gridDim.x) {
        const int img0 = grp * IMGS;
        if (grp != (int)blockIdx.x) __syncthreads();       // the previous group's last fragment reads are done
        // ---- the group's images: one linear copy of contiguous bytes, split into the two planes ---------------------------
        {
            u32x4_t v[NP];
            const unsigned g0 = (unsigned)img0 * (unsigned)(S * S * C * 4);
#pragma unroll
            for (int i = 0; i < NP; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, g0 + (unsigned)(tid + 512 * i) * 16u, 0, 0);   // beyond n_img: zeros
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int idx = tid + 512 * i, pl = idx / (C / 4), j = idx - pl * (C / 4);
                const int im = pl / (S * S), p = pl - im * (S * S);
                u32x2_t hi, lo;
                split4(__builtin_bit_cast(f32x4_t, v[i]), hi, lo);
                char* d = tile + im * IMG + (((p / S) + 1) * HS + (p % S) + 1) * PIX + j * 8;
                if (!G::RAGGED || idx < G::NPIECE) {
                    *reinterpret_cast<u32x2_t*>(d) = hi;
                    *reinterpret_cast<u32x2_t*>(d + PLANE) = lo;
                }
            }
        }

        auto run_conv = [&](const ResConvX3& cv, bool to_lds) {
            const __amdgpu_buffer_rsrc_t rs_w = mil_rsrc(cv.w, KSTEPS * NT * 2048);
            const __amdgpu_buffer_rsrc_t rs_res = mil_rsrc(cv.res, cv.res ? a.bytes : 0);
            const __amdgpu_buffer_rsrc_t rs_act = mil_rsrc(cv.act, cv.act ? a.bytes : 0);
            const __amdgpu_buffer_rsrc_t rs_out = mil_rsrc(cv.out, a.bytes);
            f32x4_t acc[MW][NTW];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                f32x4_t b;
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = cv.bias ? cv.bias[(nt0 + nt) * 16 + gq * 4 + i] : 0.f;
#pragma unroll
                for (int m = 0; m < MW; ++m) acc[m][nt] = b;
            }
            constexpr int BD = (WN == 2 && MW <= 4) ? MIL_RESX3_BD : 1;    // k-steps of filter fragments in flight per wave (8 VGPRs per fragment): L2 latency against 24 MFMAs per k-step (three: no change)
            Frag8<F32S> bq[BD + 1][NTW], aq[2][MW];
            auto fetch_b = [&](int ks) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    bq[ks % (BD + 1)][nt].h = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(lane * 32 + nt0 * 2048), (ks * NT + nt) * 2048, 0));
                    bq[ks % (BD + 1)][nt].l = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(lane * 32 + nt0 * 2048 + 16), (ks * NT + nt) * 2048, 0));
                }
            };
            // k-group q = 4*ks + gq = (tap, 8-channel group) -> byte offset of its 16 bytes from the top-left tap's record; a q beyond
            // the ninth tap (zero weights) reads the top-left record — always written, always finite
            auto off_ks = [&](int ks) {
                const int q = 4 * ks + gq;
                const int tap = q / CG, cg = q - tap * CG;
                const int ty = (tap * 11) >> 5, tx = tap - ty * 3;            // tap / 3 for tap < 10
                return q < 9 * CG ? (ty * HS + tx) * PIX + cg * 16 : 0;
            };
            auto fetch_a = [&](int ks) {
                const int off = off_ks(ks);
#pragma unroll
                for (int m = 0; m < MW; ++m) {
                    aq[ks & 1][m].h = *reinterpret_cast<const bf16x8_t*>(tile + pixbase[m] + off);
                    aq[ks & 1][m].l = *reinterpret_cast<const bf16x8_t*>(tile + PLANE + pixbase[m] + off);
                }
            };
#pragma unroll
            for (int ks = 0; ks < BD; ++ks) fetch_b(ks);
            __syncthreads();                       // the pixel planes (or the previous conv's output) are visible to every wave
            fetch_a(0);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                if (ks + BD < KSTEPS) fetch_b(ks + BD);
                if (ks + 1 < KSTEPS) fetch_a(ks + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MW; ++m)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc[m][nt] = mma8(bq[ks % (BD + 1)][nt], aq[ks & 1][m], acc[m][nt]);      // D[channel][pixel]
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue: four consecutive channels of a pixel per lane and column tile; the residual / mask operands of ALL of the
            // wave's pieces are requested before the barrier (one memory round trip per conv, not one per row tile) ---------------
            // (ragged shapes with more than four row tiles per wave: in two halves — the operand registers of all six do not fit)
            constexpr int EH = MW > 4 ? 2 : 1, MH = (MW + EH - 1) / EH;
            unsigned goff[MH];
            u32x4_t rr[MH][NTW], ra[MH][NTW];
#pragma unroll
            for (int eh = 0; eh < EH; ++eh) {
#pragma unroll
                for (int mm = 0; mm < MH; ++mm) {
                    const int m = eh * MH + mm;
                    if (m >= MW) continue;
                    const int im = pixglb[m] / (S * S);
                    goff[mm] = (pixglb[m] >= 0 && img0 + im < a.n_img) ? (unsigned)((img0 * (S * S) + pixglb[m]) * (C * 4) + gq * 16 + nt0 * 64) : MIL_OOB;
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        if (cv.res) rr[mm][nt] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, goff[mm] == MIL_OOB ? MIL_OOB : goff[mm] + nt * 64, 0, 0);
                        if (cv.act) ra[mm][nt] = __builtin_amdgcn_raw_buffer_load_b128(rs_act, goff[mm] == MIL_OOB ? MIL_OOB : goff[mm] + nt * 64, 0, 0);
                    }
                }
                if (to_lds && eh == 0) __syncthreads();           // every wave is past its last read of the planes this epilogue overwrites
#pragma unroll
                for (int mm = 0; mm < MH; ++mm) {
                    const int m = eh * MH + mm;
                    if (m >= MW) continue;
                    const bool img_ok = goff[mm] != MIL_OOB;
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        f32x4_t v = acc[m][nt];
                        if (cv.res) {
                            const f32x4_t tt = __builtin_bit_cast(f32x4_t, rr[mm][nt]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] += tt[i];
                        }
                        if (cv.lrelu) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], v[i] * a.slope);      // 0 < slope < 1
                        }
                        if (cv.act) {
                            const f32x4_t tt = __builtin_bit_cast(f32x4_t, ra[mm][nt]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] *= (tt[i] > 0.f ? 1.f : a.slope);
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs_out, img_ok ? goff[mm] + nt * 64 : MIL_OOB, 0, 0);
                        if (to_lds && (!G::RAGGED || pixglb[m] >= 0)) {      // the next conv's input planes (zeros for images beyond the launch)
                            u32x2_t hi, lo;
                            split4(v, hi, lo);
                            if (!img_ok) { hi = u32x2_t{0u, 0u}; lo = u32x2_t{0u, 0u}; }
                            char* d = tile + pixrec[m] + (nt0 + nt) * 32 + gq * 8;
                            *reinterpret_cast<u32x2_t*>(d) = hi;
                            *reinterpret_cast<u32x2_t*>(d + PLANE) = lo;
                        }
                    }
                }
            }
        };
        // the chain: conv k's epilogue overwrites the interiors with its output, conv k+1 reads them (a run-time loop: one copy
        // of the unrolled k-step body)
#pragma unroll 1
        for (int k = 0; k < a.nconv; ++k) run_conv(a.conv[k], k + 1 < a.nconv);
    }
    asm volatile("" :: "v"(sink));
}

template <int C, int S, int IMGS>
static int launch_resident_x3(ResArgsX3 a, hipStream_t st) {
    using G = ResGeomX3<C, S, IMGS>;
    auto kern = conv_resident_x3_kernel<C, S, IMGS>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::TILE) != hipSuccess)
        return MIL_ERR_LAUNCH;
    a.ngroups = (a.n_img + IMGS - 1) / IMGS;
    int grid = mil_num_cus();                  // one workgroup per CU is resident (93-141 KB of LDS)
    if (grid > a.ngroups) grid = a.ngroups;
    {
        const char* e = getenv("MIL_RES_GRID_CAP");        // tests: fewer workgroups than image groups (see launch_resident)
        const int cap = e ? atoi(e) : 0;
        if (cap > 0 && grid > cap) grid = cap;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::TILE, st, a);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

static int resident_dispatch_x3(ResArgsX3 a, int cp, int H, int W, hipStream_t st) {
    if ((size_t)a.n_img * H * W * cp * 4 >= ((size_t)1 << 31)) return MIL_ERR_UNSUPPORTED;
    a.bytes = (unsigned)((size_t)a.n_img * H * W * cp * 4);
    if (cp == 80 && H == 8 && W == 8) return launch_resident_x3<80, 8, 4>(a, st);
    if (cp == 64 && H == 16 && W == 16) return launch_resident_x3<64, 16, 1>(a, st);
    // the live driver's 300x300 tiles: 10x10 and 19x19 maps (ragged pixel lists, see ResGeomX3)
    if (cp == 80 && H == 10 && W == 10) return launch_resident_x3<80, 10, 2>(a, st);      // 200 pixels = 13 of 16 row-tile slots, 101 KB
    if (cp == 64 && H == 19 && W == 19) return launch_resident_x3<64, 19, 1>(a, st);      // 361 pixels = 23 of 24 row-tile slots, 127 KB
    return MIL_ERR_UNSUPPORTED;
}
