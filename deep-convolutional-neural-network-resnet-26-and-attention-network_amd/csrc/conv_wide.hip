// Channel-blocked implicit-GEMM convolution for WIDE layers (multiples of 64 channels): the second
// backbone configuration named by the north star, `alt_resnet.py` (torchvision-style ResNet with BatchNorm
// removed: bias-free 3x3 / 1x1 convs, ReLU, widths 64/128/256/512 — alt_resnet.py:24-33,35-66,70-145).
//
// The 20–80-channel kernels keep a whole filter and a whole-depth halo tile in LDS; at 128–512 channels
// neither fits, so here the contraction is blocked on both sides:
//   grid.y  = 64-wide OUTPUT-channel block (4 MFMA column tiles),
//   k loop  = 32-wide INPUT-channel chunks; per chunk the halo tile slice [pixels][32] and the filter slice
//             [taps][4][64 lanes][8] are staged in LDS and all taps run from them (one MFMA k-step = one tap
//             of the chunk), accumulators persist across chunks.
// Same operand conventions as conv_igemm.hip (NHWC, fragment-packed weights, fp32 accumulate; T = BF16 or the
// exact-fp32 MFMA path), same fused epilogue (bias?/residual/ReLU-or-LeakyReLU/mask), dgrad = the same kernel
// over dz with transposed+flipped packing, stride-2 dgrad through the zero-insert loader.
// Weight gradient: wide_wgrad_kernel, one (output block, input chunk) pair per grid.y.
#include "pf_common.cuh"

#define WIDE_CK 32      // input channels per chunk
#define WIDE_NB 64      // output channels per block
#define WIDE_NT 4

template <typename T>
struct WideArgs {
    const typename T::elem* x;
    const typename T::elem* w;      // [co_block][ci_chunk][tap][4][64][8]
    const float* bias;              // [cout] or null
    const typename T::elem* res;
    const typename T::elem* act;
    typename T::elem* y;
    ConvGeom g;
    int cin, cout;                  // channel counts of x and y (pixel strides)
    int apply_relu;
    float slope;
};

// halo slice loader: channels [c0, c0+32) of a tensor with `ctot` channels per pixel
template <typename T>
__device__ __forceinline__ void wide_load_halo(char* lds, const typename T::elem* __restrict__ x, const ConvGeom& g,
                                               const TileOrigin& o, int tid, int ctot, int c0) {
    constexpr int ESZ = T::ESZ;
    constexpr int PIXB = mil_pix_pitch(WIDE_CK, ESZ);
    constexpr int N16 = WIDE_CK * ESZ / 16;
    const int s = g.zins ? 1 : g.stride;
    const int iy0 = o.oy0 * s - g.pad, ix0 = o.ox0 * s - g.pad;
    const int npix = (g.hh * g.hw) << g.ti_log2;
    for (int idx = tid; idx < npix * N16; idx += 256) {
        const int hp = idx / N16, j = idx - hp * N16;
        const int ti = hp / (g.hh * g.hw), rem = hp - ti * (g.hh * g.hw);
        const int hy = rem / g.hw, hx = rem - hy * g.hw;
        const int img = o.img0 + ti;
        int iy = iy0 + hy, ix = ix0 + hx;
        bool ok = img < g.n_img && iy >= 0 && ix >= 0;
        if (g.zins) { ok = ok && !((iy | ix) & 1); iy >>= 1; ix >>= 1; }
        ok = ok && iy < g.H && ix < g.W;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (ok) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) +
                        ((((size_t)img * g.H + iy) * g.W + ix) * ctot + c0) * ESZ + j * 16);
        *reinterpret_cast<uint4*>(lds + hp * PIXB + j * 16) = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void wide_conv_kernel(WideArgs<T> a, int lds_w_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int ESZ = T::ESZ;
    constexpr int PIXB = mil_pix_pitch(WIDE_CK, ESZ);
    constexpr int FRAGB = 8 * ESZ;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;              // wave grid: 2 (pixels) x 2 (channels)
    const int r = lane & 15, gq = lane >> 4;
    const int cb = blockIdx.y;
    const TileOrigin o = mil_tile_origin(g, blockIdx.x);
    char* ldsA = smem;
    char* ldsW = smem + lds_w_off;
    const int s_eff = g.zins ? 1 : g.stride;
    const int ntaps = g.ks * g.ks;
    const int nchunks = a.cin / WIDE_CK;

    int pixbase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) pixbase[m] = mil_pix_base<PIXB>(g, (wm * 4 + m) * 16 + r, s_eff);
    f32x4_t acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[m][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const size_t chunk_elems = (size_t)ntaps * WIDE_NT * 64 * 8;
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        wide_load_halo<T>(ldsA, a.x, g, o, tid, a.cin, ch * WIDE_CK);
        {
            mil_stage_filter(ldsW, a.w + ((size_t)cb * nchunks + ch) * chunk_elems, (int)(chunk_elems * ESZ), tid, 256);
        }
        __syncthreads();
        for (int tap = 0; tap < ntaps; ++tap) {
            const int ky = tap / g.ks, kx = tap - ky * g.ks;
            const int toff = (ky * g.hw + kx) * PIXB + gq * FRAGB;
            Frag8<T> bf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = lds_frag<T>(ldsW + ((tap * WIDE_NT + wn * 2 + j) * 64 + lane) * FRAGB);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const Frag8<T> af = lds_frag<T>(ldsA + pixbase[m] + toff);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[m][j] = mma8(af, bf[j], acc[m][j]);
            }
        }
    }
    __syncthreads();
    float* epi = reinterpret_cast<float*>(smem);             // [128 px][64 ch] fp32
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                epi[((wm * 4 + m) * 16 + gq * 4 + i) * WIDE_NB + (wn * 2 + j) * 16 + r] = acc[m][j][i];
    __syncthreads();
    const int tw_mask = (1 << g.tw_log2) - 1, th_mask = (1 << g.th_log2) - 1;
    for (int idx = tid; idx < 128 * (WIDE_NB / 8); idx += 256) {
        const int tp = idx >> 3, c8 = idx & 7;
        const int ox = o.ox0 + (tp & tw_mask);
        const int oy = o.oy0 + ((tp >> g.tw_log2) & th_mask);
        const int img = o.img0 + (tp >> (g.tw_log2 + g.th_log2));
        if (img >= g.n_img || oy >= g.Ho || ox >= g.Wo) continue;
        float v[8];
        {
            const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(epi + tp * WIDE_NB + c8 * 8);
            const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(epi + tp * WIDE_NB + c8 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
        }
        const int c = cb * WIDE_NB + c8 * 8;
        if (a.bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += a.bias[c + j];
        }
        const size_t off = (((size_t)img * g.Ho + oy) * g.Wo + ox) * a.cout + c;
        if (a.res) {
            float rv[8];
            load8<T>(a.res + off, rv);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rv[j];
        }
        if (a.apply_relu) {
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

// Pipelined bf16 form of the kernel above for stride-1 launches without zero insertion (all 3x3 s1 forward convs and
// data gradients of the wide layers: 36 of the 42 wide_conv launches of an alt_resnet [3,3,3,3] step):
//   * the NEXT chunk's halo slice and filter slice are requested into registers before the current chunk's MFMA loop
//     and written to LDS after it (issue-early / write-late), so their L2/HBM latency hides under 72 MFMAs per wave
//     instead of stalling every chunk at a synchronous load -> barrier -> compute -> barrier sequence;
//   * a workgroup owns one tile for all chunks, so every piece's address and predicate is computed ONCE (buffer
//     descriptor, out-of-range offset as the predicate): a chunk adds 64 bytes;
//   * the (tap, row tile) loop is one flattened software pipeline: pixel fragments read two steps ahead through a ring,
//     filter fragments one tap ahead, scheduling fences keep the order (hipcc's own order waits for every fragment
//     right in front of its MFMAs);
//   * KS is a template parameter: tap offsets are loop-invariant registers, the tap loop is unrolled.
template <int KS>
__global__ __launch_bounds__(256, 2) void wide_conv_pf_kernel(WideArgs<BF16> a, int lds_w_off, unsigned x_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    using T = BF16;
    constexpr int PIXB = mil_pix_pitch(WIDE_CK, 2);
    constexpr int NTAP = KS * KS;
    constexpr int NPH = 4;                                   // halo pieces per thread: <= 256 pixels x 4 pieces of 16 B
    constexpr int NPW = NTAP;                                // filter pieces per thread: NTAP * 4 * 64 fragments of 16 B / 256
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;              // wave grid: 2 (pixels) x 2 (channels)
    const int r = lane & 15, gq = lane >> 4;
    const int cb = blockIdx.y;
    const TileOrigin o = mil_tile_origin(g, blockIdx.x);
    char* ldsA = smem;
    char* ldsW = smem + lds_w_off;
    const int nchunks = a.cin / WIDE_CK;
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, x_bytes);

    // halo pieces of this thread: fixed for the whole workgroup (one tile), a chunk only shifts the channel offset
    unsigned h_off[NPH];
    int h_lds[NPH];
    {
        const int npix = (g.hh * g.hw) << g.ti_log2;
        const int iy0 = o.oy0 - g.pad, ix0 = o.ox0 - g.pad;
#pragma unroll
        for (int i = 0; i < NPH; ++i) {
            const int idx = tid + 256 * i;
            const int hp = idx >> 2, j = idx & 3;
            const int ti = hp / (g.hh * g.hw), rem = hp - ti * (g.hh * g.hw);
            const int hy = rem / g.hw, hx = rem - hy * g.hw;
            const int img = o.img0 + ti;
            int iy = iy0 + hy, ix = ix0 + hx;
            const bool used = hp < npix;
            bool ok = used && img < g.n_img && iy >= 0 && ix >= 0;
            if (g.zins) { ok = ok && !((iy | ix) & 1); iy >>= 1; ix >>= 1; }      // transposed stride-2 conv: zeros between the pixels of x
            ok = ok && iy < g.H && ix < g.W;
            h_off[i] = ok ? (unsigned)((((img * g.H + iy) * g.W + ix) * a.cin) * 2 + j * 16) : MIL_OOB;
            h_lds[i] = used ? hp * PIXB + j * 16 : lds_w_off - 16;        // unused slots: 16 spare bytes behind the halo tile
        }
    }
    int pixbase[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) pixbase[m] = mil_pix_base<PIXB>(g, (wm * 4 + m) * 16 + r, 1) + gq * 16;
    int toff[NTAP];
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) toff[tap] = ((tap / KS) * g.hw + (tap % KS)) * PIXB;
    f32x4_t acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[m][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const size_t chunk_bytes = (size_t)NTAP * WIDE_NT * 64 * 16;
    const char* wbase = reinterpret_cast<const char*>(a.w) + (size_t)cb * nchunks * chunk_bytes + (size_t)tid * 16;
    u32x4_t rh[NPH], rw[NPW];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int i = 0; i < NPH; ++i)
            rh[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, h_off[i] == MIL_OOB ? MIL_OOB : h_off[i] + (unsigned)(ch * WIDE_CK * 2), 0, 0);
        const char* src = wbase + (size_t)ch * chunk_bytes;
#pragma unroll
        for (int i = 0; i < NPW; ++i) rw[i] = *reinterpret_cast<const u32x4_t*>(src + i * 256 * 16);
    };
    fetch(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();                       // the previous chunk's fragment reads are done
#pragma unroll
        for (int i = 0; i < NPH; ++i) *reinterpret_cast<u32x4_t*>(ldsA + h_lds[i]) = rh[i];
#pragma unroll
        for (int i = 0; i < NPW; ++i) *reinterpret_cast<u32x4_t*>(ldsW + (tid + 256 * i) * 16) = rw[i];
        __syncthreads();
        if (ch + 1 < nchunks) fetch(ch + 1);   // lands while the loop below runs
        {
            constexpr int TOT = NTAP * 4, LA = TOT > 2 ? 2 : 1, R = LA + 1;
            Frag8<T> ring[R], bq[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bq[0][j] = lds_frag<T>(ldsW + ((wn * 2 + j) * 64 + lane) * 16);
#pragma unroll
            for (int q = 0; q < LA; ++q) ring[q % R] = lds_frag<T>(ldsA + pixbase[q % 4] + toff[q / 4]);
#pragma unroll
            for (int q = 0; q < TOT; ++q) {
                const int tap = q / 4, m = q % 4;
                if (q + LA < TOT) ring[(q + LA) % R] = lds_frag<T>(ldsA + pixbase[(q + LA) % 4] + toff[(q + LA) / 4]);
                if (m == 0 && tap + 1 < NTAP) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) bq[(tap + 1) & 1][j] = lds_frag<T>(ldsW + (((tap + 1) * WIDE_NT + wn * 2 + j) * 64 + lane) * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[m][j] = mma8(ring[q % R], bq[tap & 1][j], acc[m][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();
    float* epi = reinterpret_cast<float*>(smem);             // [128 px][64 ch] fp32
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                epi[((wm * 4 + m) * 16 + gq * 4 + i) * WIDE_NB + (wn * 2 + j) * 16 + r] = acc[m][j][i];
    __syncthreads();
    const int tw_mask = (1 << g.tw_log2) - 1, th_mask = (1 << g.th_log2) - 1;
    for (int idx = tid; idx < 128 * (WIDE_NB / 8); idx += 256) {
        const int tp = idx >> 3, c8 = idx & 7;
        const int ox = o.ox0 + (tp & tw_mask);
        const int oy = o.oy0 + ((tp >> g.tw_log2) & th_mask);
        const int img = o.img0 + (tp >> (g.tw_log2 + g.th_log2));
        if (img >= g.n_img || oy >= g.Ho || ox >= g.Wo) continue;
        float v[8];
        {
            const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(epi + tp * WIDE_NB + c8 * 8);
            const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(epi + tp * WIDE_NB + c8 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
        }
        const int c = cb * WIDE_NB + c8 * 8;
        if (a.bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += a.bias[c + j];
        }
        const size_t off = (((size_t)img * g.Ho + oy) * g.Wo + ox) * a.cout + c;
        if (a.res) {
            float rv[8];
            load8<T>(a.res + off, rv);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rv[j];
        }
        if (a.apply_relu) {
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

// fp32 master [Cout][Cin][k][k] -> [co_block][ci_chunk][tap][4][64][8] (mode 0 forward, 1 dgrad; see conv_igemm)
template <typename T>
__global__ void wide_pack_kernel(const float* __restrict__ w, typename T::elem* __restrict__ out, int cout, int cin, int ks,
                                 int mode, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int kk = ks * ks;
    const int cin_x = mode ? cout : cin, cout_x = mode ? cin : cout;       // channels of the conv as executed
    const int nchunks = cin_x / WIDE_CK;
    const int j = idx & 7, lane = (idx >> 3) & 63, nt = (idx >> 9) & 3;
    size_t t = idx >> 11;
    const int tap = (int)(t % kk); t /= kk;
    const int ch = (int)(t % nchunks);
    const int cb = (int)(t / nchunks);
    const int kin = ch * WIDE_CK + 8 * (lane >> 4) + j;
    const int nout = cb * WIDE_NB + nt * 16 + (lane & 15);
    float val = 0.f;
    if (kin < cin_x && nout < cout_x) {
        if (!mode) val = w[((size_t)nout * cin + kin) * kk + tap];
        else val = w[((size_t)kin * cin + nout) * kk + (kk - 1 - tap)];
    }
    out[idx] = (typename T::elem)val;
}

extern "C" int mil_wide_packed_elems(size_t* elems, int cout, int cin, int ks, int mode) {
    if (!elems || cout % 64 || cin % 32 || (mode && cout % 32) || (mode && cin % 64)) return MIL_ERR_ARG;
    const int cin_x = mode ? cout : cin, cout_x = mode ? cin : cout;
    *elems = (size_t)(cout_x / WIDE_NB) * (cin_x / WIDE_CK) * ks * ks * WIDE_NT * 64 * 8;
    return MIL_OK;
}

extern "C" int mil_wide_pack_weights(const float* w, void* wpack, int cout, int cin, int ks, int mode, int dtype, void* stream) {
    size_t total = 0;
    if (!w || !wpack || mil_wide_packed_elems(&total, cout, cin, ks, mode) != MIL_OK) return MIL_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (dtype == MIL_DT_BF16) hipLaunchKernelGGL(wide_pack_kernel<BF16>, dim3(grid), dim3(256), 0, st, w, (__bf16*)wpack, cout, cin, ks, mode, total);
    else if (dtype == MIL_DT_F32) hipLaunchKernelGGL(wide_pack_kernel<F32>, dim3(grid), dim3(256), 0, st, w, (float*)wpack, cout, cin, ks, mode, total);
    else return MIL_ERR_ARG;
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

#include <cstdlib>
static bool mil_wide_pf_enabled() {          // MIL_WIDE_PF=0: the plain kernel everywhere (A/B runs)
    static const bool v = [] { const char* e = mil_ab_env("MIL_WIDE_PF"); return !(e && e[0] == '0'); }();
    return v;
}

template <typename T>
static int launch_wide(WideArgs<T> a, hipStream_t st) {
    constexpr int ESZ = T::ESZ;
    constexpr int PIXB = mil_pix_pitch(WIDE_CK, ESZ);
    mil_geom_tiles(a.g, 7);
    int a_bytes = ((((a.g.hh * a.g.hw) << a.g.ti_log2) * PIXB) + 15) & ~15;
    const int w_bytes = a.g.ks * a.g.ks * WIDE_NT * 64 * 8 * ESZ;
    if (a_bytes + w_bytes > 160 * 1024 && a.g.ti_log2 > 1) {
        // many tiny maps per tile (fp32, stride 2): the per-image halos dominate; fewer images per tile, the tile is
        // then partly empty (rows beyond Ho/Wo are predicated off) but fits
        mil_geom_set(a.g, 3, 3, 1);
        a_bytes = ((((a.g.hh * a.g.hw) << a.g.ti_log2) * PIXB) + 15) & ~15;
    }
    int lds = a_bytes + w_bytes;
    if (lds < 128 * WIDE_NB * 4) lds = 128 * WIDE_NB * 4;
    if (lds > 160 * 1024) return MIL_ERR_UNSUPPORTED;
    const int tiles = a.g.n_groups * a.g.tiles_y * a.g.tiles_x;
    if (tiles <= 0) return MIL_OK;
    if constexpr (T::DT == MIL_DT_BF16) {
        // stride-1 bf16 launches (zero-insert data gradients of the stride-2 convs included: their halo lives on the full-
        // resolution grid) whose halo fits the register prefetch: the pipelined form
        const int halo_px = (a.g.hh * a.g.hw) << a.g.ti_log2;
        const size_t xb = (size_t)a.g.n_img * a.g.H * a.g.W * a.cin * 2;
        if (mil_wide_pf_enabled() && a.g.stride == 1 && halo_px <= 256 && xb < ((size_t)1 << 31) && (a.g.ks == 3 || a.g.ks == 1)) {
            const int a_pf = a_bytes + 16;                    // + dump slot for the unused halo piece slots
            int lds_pf = a_pf + w_bytes;
            if (lds_pf < 128 * WIDE_NB * 4) lds_pf = 128 * WIDE_NB * 4;
            auto kpf = a.g.ks == 3 ? wide_conv_pf_kernel<3> : wide_conv_pf_kernel<1>;
            if (lds_pf > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kpf), hipFuncAttributeMaxDynamicSharedMemorySize, lds_pf) != hipSuccess)
                return MIL_ERR_LAUNCH;
            hipLaunchKernelGGL(kpf, dim3(tiles, a.cout / WIDE_NB), dim3(256), lds_pf, st, a, a_pf, (unsigned)xb);
            MIL_CHECK_LAUNCH();
            return MIL_OK;
        }
    }
    auto kern = wide_conv_kernel<T>;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return MIL_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(tiles, a.cout / WIDE_NB), dim3(256), lds, st, a, a_bytes);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// y = mask(relu?(conv(x) + bias? + res?)) for channel counts that are multiples of 64 (cin: of 32).
extern "C" int mil_wide_conv(const void* x, const void* wpack, const float* bias, const void* res, const void* act, void* y,
                             int n_img, int H, int W, int cin, int Ho, int Wo, int cout, int ks, int stride, int pad,
                             int zero_insert, int apply_relu, float slope, int dtype, void* stream) {
    if (!x || !wpack || !y || n_img < 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return MIL_ERR_ARG;
    if (cin % WIDE_CK || cout % WIDE_NB || !(ks == 1 || ks == 3) || !(stride == 1 || stride == 2)) return MIL_ERR_UNSUPPORTED;
    ConvGeom g{};
    g.n_img = n_img; g.H = H; g.W = W; g.Ho = Ho; g.Wo = Wo; g.ks = ks; g.stride = zero_insert ? 1 : stride; g.pad = pad;
    g.zins = zero_insert ? 1 : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIL_DT_BF16) {
        WideArgs<BF16> a{};
        a.x = (const __bf16*)x; a.w = (const __bf16*)wpack; a.bias = bias; a.res = (const __bf16*)res; a.act = (const __bf16*)act;
        a.y = (__bf16*)y; a.g = g; a.cin = cin; a.cout = cout; a.apply_relu = apply_relu; a.slope = slope;
        return launch_wide<BF16>(a, st);
    } else if (dtype == MIL_DT_F32) {
        WideArgs<F32> a{};
        a.x = (const float*)x; a.w = (const float*)wpack; a.bias = bias; a.res = (const float*)res; a.act = (const float*)act;
        a.y = (float*)y; a.g = g; a.cin = cin; a.cout = cout; a.apply_relu = apply_relu; a.slope = slope;
        return launch_wide<F32>(a, st);
    }
    return MIL_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------
// Weight gradient for wide layers: grid.y = (output block, input chunk) pair; rows = (tap, 32 input channels),
// cols = 64 output channels, K = pixels (persistent over tiles, one fp32 slab per workgroup, fixed-order reduce).
template <typename T>
struct WideWgradArgs {
    const typename T::elem* x;
    const typename T::elem* dz;
    float* slab;                    // [grid.y][grid.x][(ntaps*2)*16][64]
    ConvGeom g;
    int cin, cout, ntiles, lds_z_off;
};

template <typename T>
__device__ __forceinline__ void wide_load_otile(char* lds, const typename T::elem* __restrict__ z, const ConvGeom& g,
                                                const TileOrigin& o, int tid, int ctot, int c0) {
    constexpr int ESZ = T::ESZ;
    constexpr int PIXZ = mil_pix_pitch(WIDE_NB, ESZ);
    constexpr int N16 = WIDE_NB * ESZ / 16;
    const int tw_mask = (1 << g.tw_log2) - 1, th_mask = (1 << g.th_log2) - 1;
    for (int idx = tid; idx < 128 * N16; idx += 256) {
        const int tp = idx / N16, j = idx - tp * N16;
        const int ox = o.ox0 + (tp & tw_mask);
        const int oy = o.oy0 + ((tp >> g.tw_log2) & th_mask);
        const int img = o.img0 + (tp >> (g.tw_log2 + g.th_log2));
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (img < g.n_img && oy < g.Ho && ox < g.Wo)
            v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(z) +
                    ((((size_t)img * g.Ho + oy) * g.Wo + ox) * ctot + c0) * ESZ + j * 16);
        *reinterpret_cast<uint4*>(lds + tp * PIXZ + j * 16) = v;
    }
}

template <typename T, int KS>
__global__ __launch_bounds__(256) void wide_wgrad_kernel(WideWgradArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int ESZ = T::ESZ;
    constexpr int PIXB = mil_pix_pitch(WIDE_CK, ESZ);
    constexpr int PIXZ = mil_pix_pitch(WIDE_NB, ESZ);
    constexpr int CG = WIDE_CK / 8;                        // 4 row groups per tap
    constexpr int RG = KS * KS * CG, MT = RG / 2, MW = (MT + 3) / 4;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = a.cin / WIDE_CK;
    const int cb = blockIdx.y / nchunks, ch = blockIdx.y - cb * nchunks;
    char* ldsX = smem;
    char* ldsZ = smem + a.lds_z_off;
    int toff[MW];
    bool mvalid[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int mt = wave + 4 * i;
        mvalid[i] = mt < MT;
        int rg, sub;
        if constexpr (T::DT == MIL_DT_BF16) { const int p = lane & 3; rg = 2 * mt + (p >> 1); sub = (p & 1) * 8; }
        else { const int row = lane & 15; rg = 2 * mt + (row >> 3); sub = (row & 7) * 4; }
        if (rg >= RG) rg = 0;
        const int tap = rg / CG, cg = rg - tap * CG;
        toff[i] = ((tap / KS) * g.hw + (tap % KS)) * PIXB + cg * (8 * ESZ) + sub;
    }
    f32x4_t acc[MW][WIDE_NT];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int nt = 0; nt < WIDE_NT; ++nt) acc[i][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const TileOrigin o = mil_tile_origin(g, tile);
        __syncthreads();
        wide_load_halo<T>(ldsX, a.x, g, o, tid, a.cin, ch * WIDE_CK);
        wide_load_otile<T>(ldsZ, a.dz, g, o, tid, a.cout, cb * WIDE_NB);
        __syncthreads();
        if constexpr (T::DT == MIL_DT_BF16) {
            const int q4 = (lane & 15) >> 2, p = lane & 3, gq = lane >> 4;
            for (int k32 = 0; k32 < 128; k32 += 32) {
                const int tp0 = k32 + 8 * gq + q4, tp1 = tp0 + 4;
                const int pb0 = mil_pix_base<PIXB>(g, tp0, g.stride), pb1 = mil_pix_base<PIXB>(g, tp1, g.stride);
                const char* z0 = ldsZ + tp0 * PIXZ + p * 8;
                const char* z1 = ldsZ + tp1 * PIXZ + p * 8;
                bf16x8_t bf[WIDE_NT];
#pragma unroll
                for (int nt = 0; nt < WIDE_NT; ++nt) bf[nt] = mil_tr_pair(z0 + nt * 32, z1 + nt * 32);
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    if (mvalid[i]) {
                        const bf16x8_t af = mil_tr_pair(ldsX + pb0 + toff[i], ldsX + pb1 + toff[i]);
#pragma unroll
                        for (int nt = 0; nt < WIDE_NT; ++nt)
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[nt], acc[i][nt], 0, 0, 0);
                    }
                }
            }
        } else {
            const int gq = lane >> 4, col = lane & 15;
            for (int k4 = 0; k4 < 128; k4 += 4) {
                const int tp = k4 + gq;
                const int pb = mil_pix_base<PIXB>(g, tp, g.stride);
                float bf[WIDE_NT];
#pragma unroll
                for (int nt = 0; nt < WIDE_NT; ++nt) bf[nt] = *reinterpret_cast<const float*>(ldsZ + tp * PIXZ + (nt * 16 + col) * 4);
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    if (mvalid[i]) {
                        const float af = *reinterpret_cast<const float*>(ldsX + pb + toff[i]);
#pragma unroll
                        for (int nt = 0; nt < WIDE_NT; ++nt)
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[nt], acc[i][nt], 0, 0, 0);
                    }
                }
            }
        }
    }
    constexpr size_t SLAB = (size_t)MT * 16 * WIDE_NB;
    float* slab = a.slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SLAB;
    const int gq = lane >> 4, col = lane & 15;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        if (!mvalid[i]) continue;
        const int mt = wave + 4 * i;
#pragma unroll
        for (int nt = 0; nt < WIDE_NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                slab[(size_t)(mt * 16 + gq * 4 + e) * WIDE_NB + nt * 16 + col] = acc[i][nt][e];
    }
}

// Pipelined bf16 form of the kernel above (stride 1, halo <= 256 pixels): the NEXT tile's halo slice and dz tile are
// requested into registers before the current tile's MFMA loop and written to LDS after it; piece -> (halo position, LDS
// offset, relative address) tables are built once per workgroup (buffer descriptors, out-of-range offset as the
// predicate); the four 32-pixel k-steps of a tile run one step ahead on ping-pong operand sets without validity branches
// (a row tile that does not exist reads row group 0 into accumulators that are never stored).
template <int KS>
__global__ __launch_bounds__(256, 2) void wide_wgrad_pf_kernel(WideWgradArgs<BF16> a, unsigned x_bytes, unsigned z_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    constexpr int PIXB = mil_pix_pitch(WIDE_CK, 2);
    constexpr int PIXZ = mil_pix_pitch(WIDE_NB, 2);
    constexpr int CG = WIDE_CK / 8;
    constexpr int RG = KS * KS * CG, MT = RG / 2, MW = (MT + 3) / 4;
    constexpr int NPH = 4, NPZ = 4;                          // 16-byte pieces per thread: halo slice (<= 256 px x 4), dz tile (128 px x 8)
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = a.cin / WIDE_CK;
    const int cb = blockIdx.y / nchunks, ch = blockIdx.y - cb * nchunks;
    char* ldsX = smem;
    char* ldsZ = smem + a.lds_z_off;
    const __amdgpu_buffer_rsrc_t rs_x = mil_rsrc(a.x, x_bytes);
    const __amdgpu_buffer_rsrc_t rs_z = mil_rsrc(a.dz, z_bytes);
    int h_pos[NPH], h_lds[NPH], h_rel[NPH], z_pos[NPZ], z_lds[NPZ], z_rel[NPZ];
    {
        const int npix = (g.hh * g.hw) << g.ti_log2;
        const int tw_mask = (1 << g.tw_log2) - 1, th_mask = (1 << g.th_log2) - 1;
#pragma unroll
        for (int i = 0; i < NPH; ++i) {
            const int idx = tid + 256 * i, hp = idx >> 2, j = idx & 3;
            const int ti = hp / (g.hh * g.hw), rem = hp - ti * (g.hh * g.hw), hy = rem / g.hw, hx = rem - hy * g.hw;
            const bool used = hp < npix;
            h_pos[i] = used ? (ti << 20) | (hy << 10) | hx : -1;
            h_lds[i] = used ? hp * PIXB + j * 16 : a.lds_z_off - 16;          // 16 spare bytes behind the halo tile
            h_rel[i] = ((ti * g.H + hy) * g.W + hx) * (a.cin * 2) + j * 16;
        }
#pragma unroll
        for (int i = 0; i < NPZ; ++i) {
            const int idx = tid + 256 * i, tp = idx >> 3, j = idx & 7;
            const int tx = tp & tw_mask, ty = (tp >> g.tw_log2) & th_mask, ti = tp >> (g.tw_log2 + g.th_log2);
            z_pos[i] = (ti << 20) | (ty << 10) | tx;
            z_lds[i] = tp * PIXZ + j * 16;
            z_rel[i] = ((ti * g.Ho + ty) * g.Wo + tx) * (a.cout * 2) + j * 16;
        }
    }
    u32x4_t rh[NPH], rz[NPZ];
    auto fetch = [&](int tile) {
        const TileOrigin o = mil_tile_origin(g, tile);
        const int iy0 = o.oy0 - g.pad, ix0 = o.ox0 - g.pad, ilim = g.n_img - o.img0;
        const int xbase = ((o.img0 * g.H + iy0) * g.W + ix0) * (a.cin * 2) + ch * (WIDE_CK * 2);      // may be negative; valid lanes are not
        const int zbase = ((o.img0 * g.Ho + o.oy0) * g.Wo + o.ox0) * (a.cout * 2) + cb * (WIDE_NB * 2);
#pragma unroll
        for (int i = 0; i < NPH; ++i) {
            const int p = h_pos[i];
            const bool ok = p >= 0 && (p >> 20) < ilim && (unsigned)(iy0 + ((p >> 10) & 1023)) < (unsigned)g.H &&
                            (unsigned)(ix0 + (p & 1023)) < (unsigned)g.W;
            rh[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(xbase + h_rel[i]) : MIL_OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NPZ; ++i) {
            const int p = z_pos[i];
            const bool ok = (p >> 20) < ilim && o.oy0 + ((p >> 10) & 1023) < g.Ho && o.ox0 + (p & 1023) < g.Wo;
            rz[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, ok ? (unsigned)(zbase + z_rel[i]) : MIL_OOB, 0, 0);
        }
    };
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, gq = lane >> 4;
    int toff[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        int rg = 2 * (wave + 4 * i) + (p4 >> 1);
        if (rg >= RG) rg = 0;
        const int tap = rg / CG, cg = rg - tap * CG;
        toff[i] = ((tap / KS) * g.hw + (tap % KS)) * PIXB + cg * 16 + (p4 & 1) * 8;
    }
    const int wpl0 = mil_pix_base<PIXB>(g, 8 * gq + q4, 1), wpl1 = mil_pix_base<PIXB>(g, 8 * gq + q4 + 4, 1);
    const int zl0 = (8 * gq + q4) * PIXZ + p4 * 8;
    f32x4_t acc[MW][WIDE_NT];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int nt = 0; nt < WIDE_NT; ++nt) acc[i][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if ((int)blockIdx.x < a.ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        __syncthreads();                       // the previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < NPH; ++i) *reinterpret_cast<u32x4_t*>(ldsX + h_lds[i]) = rh[i];
#pragma unroll
        for (int i = 0; i < NPZ; ++i) *reinterpret_cast<u32x4_t*>(ldsZ + z_lds[i]) = rz[i];
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);
        bf16x8_t bq[2][WIDE_NT], aq[2][MW];
        auto load = [&](int k32, bf16x8_t (&bf)[WIDE_NT], bf16x8_t (&af)[MW]) {
            const int kb = mil_pix_base<PIXB>(g, k32, 1);
            const char* z0 = ldsZ + k32 * PIXZ + zl0;
#pragma unroll
            for (int nt = 0; nt < WIDE_NT; ++nt) bf[nt] = mil_tr_pair(z0 + nt * 32, z0 + 4 * PIXZ + nt * 32);
#pragma unroll
            for (int i = 0; i < MW; ++i) af[i] = mil_tr_pair(ldsX + kb + wpl0 + toff[i], ldsX + kb + wpl1 + toff[i]);
        };
        load(0, bq[0], aq[0]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k + 1 < 4) load((k + 1) * 32, bq[(k + 1) & 1], aq[(k + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MW; ++i)
#pragma unroll
                for (int nt = 0; nt < WIDE_NT; ++nt)
                    acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[k & 1][i], bq[k & 1][nt], acc[i][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    constexpr size_t SLAB = (size_t)MT * 16 * WIDE_NB;
    float* slab = a.slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SLAB;
    const int col = lane & 15;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int mt = wave + 4 * i;
        if (mt >= MT) continue;
#pragma unroll
        for (int nt = 0; nt < WIDE_NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                slab[(size_t)(mt * 16 + gq * 4 + e) * WIDE_NB + nt * 16 + col] = acc[i][nt][e];
    }
}

// dW[cb*64+col][ch*32+ci][tap] (+)= sum over the pair's slabs (fixed order); slab row = tap*32 + ci.  One thread owns four
// consecutive output channels (16-byte slab loads, eight slabs requested ahead of the adds that consume them in slab order).
__global__ void wide_wgrad_reduce_kernel(const float* __restrict__ slab, int nslab, int npairs, int nchunks, int ks,
                                         float* __restrict__ dw, int cout, int cin, int accumulate) {
    const int kk = ks * ks;
    const size_t per_pair = (size_t)kk * WIDE_CK * WIDE_NB;
    const size_t idx = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (idx >= per_pair * npairs) return;
    const int pair = (int)(idx / per_pair);
    const int e = (int)(idx - (size_t)pair * per_pair);
    const int row = e / WIDE_NB, col = e - row * WIDE_NB;
    const int tap = row / WIDE_CK, ci = row - tap * WIDE_CK;
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
    float* q = dw + ((size_t)(cb * WIDE_NB + col) * cin + ch * WIDE_CK + ci) * kk + tap;
    const size_t cs = (size_t)cin * kk;                       // next output channel
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k * cs] = accumulate ? q[k * cs] + s[k] : s[k];
}

template <typename T, int KS>
static int run_wide_wgrad(const void* x, const void* dz, float* dw, void* ws, size_t ws_bytes, ConvGeom g, int cin, int cout,
                          int accumulate, bool query, size_t* need, hipStream_t st) {
    constexpr int ESZ = T::ESZ;
    constexpr int PIXB = mil_pix_pitch(WIDE_CK, ESZ), PIXZ = mil_pix_pitch(WIDE_NB, ESZ);
    constexpr int MT = KS * KS * (WIDE_CK / 8) / 2;
    mil_geom_tiles(g, 7);
    const int xb = ((((g.hh * g.hw) << g.ti_log2) * PIXB) + 15) & ~15, zb = 128 * PIXZ;
    if (xb + zb > 160 * 1024) return MIL_ERR_UNSUPPORTED;
    const int ntiles = g.n_groups * g.tiles_y * g.tiles_x;
    const int npairs = (cout / WIDE_NB) * (cin / WIDE_CK);
    // one fp32 slab per workgroup: 512 workgroups (two per CU) keep the slab write + fixed-order reduce at 37 MB per launch —
    // the 2048 of earlier rounds moved 151 MB for a 9 MB gradient (MIL_WIDE_WGRAD_WGS: A/B runs)
    static const int wg_target = [] { const char* e = mil_ab_env("MIL_WIDE_WGRAD_WGS"); return e ? atoi(e) : 512; }();
    int gx = wg_target / npairs;
    if (gx < 4) gx = 4;
    if (gx > 64) gx = 64;
    if (gx > ntiles) gx = ntiles;
    const size_t slab = (size_t)MT * 16 * WIDE_NB;
    const size_t bytes = slab * gx * npairs * sizeof(float);
    if (query) { *need = bytes; return MIL_OK; }
    if (!ws || ws_bytes < bytes) return MIL_ERR_ARG;
    WideWgradArgs<T> a{};
    a.x = (const typename T::elem*)x; a.dz = (const typename T::elem*)dz; a.slab = (float*)ws; a.g = g;
    a.cin = cin; a.cout = cout; a.ntiles = ntiles; a.lds_z_off = xb;
    bool done = false;
    if constexpr (T::DT == MIL_DT_BF16) {
        const size_t xbytes = (size_t)g.n_img * g.H * g.W * cin * 2, zbytes = (size_t)g.n_img * g.Ho * g.Wo * cout * 2;
        const int halo_px = (g.hh * g.hw) << g.ti_log2;
        if (mil_wide_pf_enabled() && g.stride == 1 && halo_px <= 256 && g.hh < 1024 && g.hw < 1024 &&
            xbytes < ((size_t)1 << 31) && zbytes < ((size_t)1 << 31)) {
            a.lds_z_off = xb + 16;                              // + dump slot for the unused halo piece slots
            auto kpf = wide_wgrad_pf_kernel<KS>;
            const int lds = xb + 16 + zb;
            if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kpf), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
                return MIL_ERR_LAUNCH;
            hipLaunchKernelGGL(kpf, dim3(gx, npairs), dim3(256), lds, st, a, (unsigned)xbytes, (unsigned)zbytes);
            MIL_CHECK_LAUNCH();
            done = true;
        }
    }
    if (!done) {
        auto kern = wide_wgrad_kernel<T, KS>;
        const int lds = xb + zb;
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MIL_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(gx, npairs), dim3(256), lds, st, a);
        MIL_CHECK_LAUNCH();
    }
    const size_t total = (size_t)KS * KS * WIDE_CK * WIDE_NB * npairs;
    hipLaunchKernelGGL(wide_wgrad_reduce_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, gx, npairs,
                       cin / WIDE_CK, KS, dw, cout, cin, accumulate);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// conv_gather.hip: the gather form of the weight gradient (bf16; any stride / 1x1)
int mil_gwgrad(const void* x, const void* dz, float* dw, void* ws, size_t ws_bytes, int n_img, int H, int W, int cin, int Ho, int Wo,
               int cout, int ks, int stride, int pad, int accumulate, bool query, size_t* need, hipStream_t st);
static int mil_gwgrad_mode() {           // MIL_GWGRAD: 0 = never, 1 = stride-2 and 1x1 launches (default), 2 = every eligible launch
    static const int v = [] { const char* e = mil_ab_env("MIL_GWGRAD"); return e ? atoi(e) : 1; }();
    return v;
}

static int wide_wgrad_entry(const void* x, const void* dz, float* dw, void* ws, size_t ws_bytes, int n_img, int H, int W, int cin,
                            int Ho, int Wo, int cout, int ks, int stride, int pad, int accumulate, int dtype, bool query,
                            size_t* need, void* stream) {
    if (cin % WIDE_CK || cout % WIDE_NB || !(ks == 1 || ks == 3) || !(stride == 1 || stride == 2)) return MIL_ERR_UNSUPPORTED;
    if (n_img < 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return MIL_ERR_ARG;
    ConvGeom g{};
    g.n_img = n_img; g.H = H; g.W = W; g.Ho = Ho; g.Wo = Wo; g.ks = ks; g.stride = stride; g.pad = pad; g.zins = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIL_DT_BF16 && (mil_gwgrad_mode() == 2 || (mil_gwgrad_mode() == 1 && (stride == 2 || ks == 1)))) {
        // the stride-2 3x3 and the 1x1 gradients have no pipelined form here (wide_wgrad_pf_kernel is stride 1): gather form
        const int rc = mil_gwgrad(x, dz, dw, ws, ws_bytes, n_img, H, W, cin, Ho, Wo, cout, ks, stride, pad, accumulate, query, need, st);
        if (rc != MIL_ERR_UNSUPPORTED) return rc;
    }
    if (dtype == MIL_DT_BF16) return ks == 3 ? run_wide_wgrad<BF16, 3>(x, dz, dw, ws, ws_bytes, g, cin, cout, accumulate, query, need, st)
                                             : run_wide_wgrad<BF16, 1>(x, dz, dw, ws, ws_bytes, g, cin, cout, accumulate, query, need, st);
    if (dtype == MIL_DT_F32) return ks == 3 ? run_wide_wgrad<F32, 3>(x, dz, dw, ws, ws_bytes, g, cin, cout, accumulate, query, need, st)
                                            : run_wide_wgrad<F32, 1>(x, dz, dw, ws, ws_bytes, g, cin, cout, accumulate, query, need, st);
    return MIL_ERR_ARG;
}

extern "C" int mil_wide_wgrad_workspace(size_t* bytes, int n_img, int H, int W, int cin, int Ho, int Wo, int cout, int ks,
                                        int stride, int pad, int dtype) {
    if (!bytes) return MIL_ERR_ARG;
    return wide_wgrad_entry(nullptr, nullptr, nullptr, nullptr, 0, n_img, H, W, cin, Ho, Wo, cout, ks, stride, pad, 0, dtype, true,
                            bytes, nullptr);
}

extern "C" int mil_wide_wgrad(const void* x, const void* dz, float* dw, void* workspace, size_t workspace_bytes, int n_img,
                              int H, int W, int cin, int Ho, int Wo, int cout, int ks, int stride, int pad, int accumulate,
                              int dtype, void* stream) {
    if (!x || !dz || !dw) return MIL_ERR_ARG;
    size_t need = 0;
    return wide_wgrad_entry(x, dz, dw, workspace, workspace_bytes, n_img, H, W, cin, Ho, Wo, cout, ks, stride, pad, accumulate,
                            dtype, false, &need, stream);
}
