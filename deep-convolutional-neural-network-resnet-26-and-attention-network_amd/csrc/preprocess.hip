// Tile pre-processing on the device: the reference finalises every cached uint8 ROI on the host
// with torchvision/Pillow (RoiBuilder.py:193-210: Pad(100) -> RandomCrop(roi) -> Resize(res) -> flips -> ToTensor ->
// Normalize(0.5, 0.5)) and then uploads fp32 tiles.  Here the cached uint8 ROIs stay resident in HBM (2500 x 4.3 MB
// per slide fits easily in 288 GB) and one kernel produces the fp32 NCHW tile stack the encoder reads.
//
// Arithmetic = Pillow's two-pass bilinear resampling, restated (not linked): per output sample a support-scaled
// triangle filter, normalised in float64 and rounded to 22-bit fixed point (host: mil_resize_coeffs), horizontal
// pass to uint8, vertical pass to uint8, each with a half-ulp bias and saturation — bit-exact against Pillow
// (tests/golden/prep_*.npz).  Padding and cropping are index arithmetic: every staged source row sits between two
// zero margins of `pad` pixels in LDS, so taps never need a bounds test; flips are applied at the store.
//
// One workgroup = 8 output rows of one tile: their ~8*scale + 2*support source rows are staged four at a time with
// aligned 4-byte loads, filtered horizontally into an LDS strip [rows][res][3], then the strip is filtered vertically.
#include "common.cuh"
#include <cmath>

#define PREP_BITS 22
#define PREP_BAND 8
#define PREP_ROWS 4

struct PrepArgs {
    const uint8_t* rois;        // [T,S,S,3]
    const int* params;          // [T,4] = top, left, hflip, vflip (train chain) or null (flat chain: no pad/crop/flip)
    const int* bounds;          // [R,2] first source index, count
    const int* kk;              // [R,ksize]
    float* out;                 // [T,3,R,R] fp32 NCHW, or null when xs is given
    __bf16* xs;                 // [T,R/2,R/2,16] bf16 space-to-depth NHWC (channel = c*4 + dy*2 + dx, 12 real), or null
    int T, S, pad, R, ksize;
    int margin, row_pitch;      // bytes: zero margin in front of a staged row (>= 3*pad, 4-aligned); pitch of a staged row
    int lds_h_off, lds_k_off, lds_b_off;
};

__device__ __forceinline__ unsigned prep_clip8(int v) { v >>= PREP_BITS; return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

template <int KS>
__global__ __launch_bounds__(256) void tile_preprocess_kernel(PrepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    MIL_POISON(psm);
    unsigned char* rows = psm;
    unsigned char* hres = psm + a.lds_h_off;
    int* kks = reinterpret_cast<int*>(psm + a.lds_k_off);
    int* bds = reinterpret_cast<int*>(psm + a.lds_b_off);
    const int tid = threadIdx.x, t = blockIdx.y;
    const int S = a.S, R = a.R, ksize = a.ksize;
    for (int i = tid; i < R * ksize; i += 256) kks[i] = a.kk[i];
    for (int i = tid; i < 2 * R; i += 256) bds[i] = a.bounds[i];
    for (int i = tid * 4; i < PREP_ROWS * a.row_pitch; i += 1024) *reinterpret_cast<unsigned*>(rows + i) = 0u;   // margins stay zero
    int top = 0, left = 0, hflip = 0, vflip = 0, pad = 0;
    if (a.params) { top = a.params[4 * t]; left = a.params[4 * t + 1]; hflip = a.params[4 * t + 2]; vflip = a.params[4 * t + 3]; pad = a.pad; }
    __syncthreads();
    const int r0 = blockIdx.x * PREP_BAND, r1 = min(r0 + PREP_BAND, R);
    const int y_lo = bds[2 * r0], y_hi = bds[2 * (r1 - 1)] + bds[2 * (r1 - 1) + 1];
    const int nri = y_hi - y_lo;
    const unsigned char* src_tile = a.rois + (size_t)t * S * S * 3;
    const int row_bytes = S * 3;
    const bool wide = (row_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(src_tile) & 3) == 0;

    // Fast path (16-byte aligned rows, <= 4 KB each): the next four source rows are fetched into registers while the
    // current four are filtered, so the global-load latency is paid once per band instead of once per row group.
    typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
    const int ppr = row_bytes >> 4;                                   // 16-byte pieces per source row
    const bool fast = (row_bytes & 15) == 0 && (a.margin & 15) == 0 && (reinterpret_cast<uintptr_t>(src_tile) & 15) == 0 && ppr <= 256;
    u4_t pre[PREP_ROWS];
    auto fetch_rows = [&](int j0) {
#pragma unroll
        for (int i = 0; i < PREP_ROWS; ++i) {
            const int id = tid + 256 * i, jj = id / ppr, piece = id - jj * ppr;
            const int ys = top + y_lo + j0 + jj - pad;
            const bool ok = jj < PREP_ROWS && j0 + jj < nri && ys >= 0 && ys < S;
            pre[i] = ok ? *reinterpret_cast<const u4_t*>(src_tile + (size_t)ys * row_bytes + piece * 16) : u4_t{0u, 0u, 0u, 0u};
        }
    };
    if (fast) fetch_rows(0);

    for (int j0 = 0; j0 < nri; j0 += PREP_ROWS) {
        // ---- stage up to four source rows (rows of the padded/cropped image; outside the ROI they are zero) ----
        if (fast) {
#pragma unroll
            for (int i = 0; i < PREP_ROWS; ++i) {
                const int id = tid + 256 * i, jj = id / ppr, piece = id - jj * ppr;
                if (jj < PREP_ROWS) *reinterpret_cast<u4_t*>(rows + jj * a.row_pitch + a.margin + piece * 16) = pre[i];
            }
            __syncthreads();
            if (j0 + PREP_ROWS < nri) fetch_rows(j0 + PREP_ROWS);
        } else {
            for (int jj = 0; jj < PREP_ROWS; ++jj) {
                if (j0 + jj >= nri) break;
                const int ys = top + y_lo + j0 + jj - pad;
                const bool inside = ys >= 0 && ys < S;
                unsigned char* dst = rows + jj * a.row_pitch + a.margin;
                const unsigned char* src = src_tile + (size_t)(inside ? ys : 0) * row_bytes;
                if (wide) {
                    for (int b = tid * 4; b < row_bytes; b += 1024)
                        *reinterpret_cast<unsigned*>(dst + b) = inside ? *reinterpret_cast<const unsigned*>(src + b) : 0u;
                } else {
                    for (int b = tid; b < row_bytes; b += 256) dst[b] = inside ? src[b] : (unsigned char)0;
                }
            }
            __syncthreads();
        }
        // ---- horizontal pass of these rows -> strip ------------------------------------------------------------
        // A thread owns output column xr for all staged rows: its KS weights are read once; the 3*KS source bytes of
        // a row are fetched as aligned dwords and funnel-shifted (v_alignbyte) to the window start, so the taps index
        // registers at compile-time positions (weights beyond the window are zero, the bytes behind them are slack).
        for (int xr = tid; xr < R; xr += 256) {
            constexpr int ND = (3 * KS + 3) / 4;
            const int xmin = bds[2 * xr];
            int w[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) w[k] = k < ksize ? kks[xr * ksize + k] : 0;
            const int b0 = a.margin + (left - pad + xmin) * 3;
            const int al = b0 & ~3;
            const unsigned sh = (unsigned)(b0 & 3);
            for (int jj = 0; jj < PREP_ROWS; ++jj) {
                if (j0 + jj >= nri) break;
                const unsigned* q = reinterpret_cast<const unsigned*>(rows + jj * a.row_pitch + al);
                unsigned d[ND + 1], e[ND];
#pragma unroll
                for (int i = 0; i < ND + 1; ++i) d[i] = q[i];
#pragma unroll
                for (int i = 0; i < ND; ++i) e[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
                int s0 = 1 << (PREP_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    s0 += (int)((e[(3 * k) >> 2] >> (((3 * k) & 3) * 8)) & 0xffu) * w[k];
                    s1 += (int)((e[(3 * k + 1) >> 2] >> (((3 * k + 1) & 3) * 8)) & 0xffu) * w[k];
                    s2 += (int)((e[(3 * k + 2) >> 2] >> (((3 * k + 2) & 3) * 8)) & 0xffu) * w[k];
                }
                unsigned char* h = hres + ((j0 + jj) * R + xr) * 3;
                h[0] = (unsigned char)prep_clip8(s0); h[1] = (unsigned char)prep_clip8(s1); h[2] = (unsigned char)prep_clip8(s2);
            }
        }
        __syncthreads();
    }
    // ---- vertical pass of the strip, ToTensor + Normalize, flips at the store ---------------------------------------
    if (a.xs) {
        // space-to-depth bf16 output: what the stem kernels consume (the 7x7/s2 conv as a 4x4/s1 conv over 2x2 pixel blocks).
        // A thread owns one 2x2 block of the band = ONE 32-byte record [c0: 00 01 10 11 | c1 .. | c2 .. | 0 0 0 0]; a flip
        // sends a block to the mirrored block and swaps the pixels inside it.  R is even, bands start on even rows.
        const int R2 = R >> 1;
        for (int it = tid; it < ((r1 - r0) >> 1) * R2; it += 256) {
            const int by = it / R2, bx = it - by * R2;
            __bf16 rec[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) rec[j] = (__bf16)0.0f;
            int oby = 0, obx = 0;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int yr = r0 + 2 * by + dy, xr = 2 * bx + dx;
                    const int ymin = bds[2 * yr], cnt = bds[2 * yr + 1];
                    const unsigned char* p = hres + ((ymin - y_lo) * R + xr) * 3;
                    const int* w = kks + yr * ksize;
                    int s0 = 1 << (PREP_BITS - 1), s1 = s0, s2 = s0;
                    for (int k = 0; k < cnt; ++k) {
                        const int wk = w[k];
                        s0 += (int)p[k * R * 3] * wk; s1 += (int)p[k * R * 3 + 1] * wk; s2 += (int)p[k * R * 3 + 2] * wk;
                    }
                    const int yo = vflip ? R - 1 - yr : yr, xo = hflip ? R - 1 - xr : xr;
                    oby = yo >> 1; obx = xo >> 1;                       // the same block for all four pixels
                    const int q = (yo & 1) * 2 + (xo & 1);
                    const float v0 = ((float)prep_clip8(s0) / 255.0f - 0.5f) / 0.5f, v1 = ((float)prep_clip8(s1) / 255.0f - 0.5f) / 0.5f,
                                v2 = ((float)prep_clip8(s2) / 255.0f - 0.5f) / 0.5f;
                    // (compile-time register indices: select on q)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        if (q == qq) { rec[qq] = (__bf16)v0; rec[4 + qq] = (__bf16)v1; rec[8 + qq] = (__bf16)v2; }
                    }
                }
            }
            bf16x8_t lo8, hi8;
#pragma unroll
            for (int j = 0; j < 8; ++j) { lo8[j] = rec[j]; hi8[j] = rec[8 + j]; }
            __bf16* o = a.xs + (((size_t)t * R2 + oby) * R2 + obx) * 16;
            *reinterpret_cast<bf16x8_t*>(o) = lo8;
            *reinterpret_cast<bf16x8_t*>(o + 8) = hi8;
        }
        return;
    }
    for (int it = tid; it < (r1 - r0) * R; it += 256) {
        const int yy = it / R, xr = it - yy * R, yr = r0 + yy;
        const int ymin = bds[2 * yr], cnt = bds[2 * yr + 1];
        const unsigned char* p = hres + ((ymin - y_lo) * R + xr) * 3;
        const int* w = kks + yr * ksize;
        int s0 = 1 << (PREP_BITS - 1), s1 = s0, s2 = s0;
        for (int k = 0; k < cnt; ++k) {
            const int wk = w[k];
            s0 += (int)p[k * R * 3] * wk; s1 += (int)p[k * R * 3 + 1] * wk; s2 += (int)p[k * R * 3 + 2] * wk;
        }
        const int yo = vflip ? R - 1 - yr : yr, xo = hflip ? R - 1 - xr : xr;
        float* o = a.out + ((size_t)t * 3 * R + yo) * R + xo;
        const size_t plane = (size_t)R * R;
        o[0] = ((float)prep_clip8(s0) / 255.0f - 0.5f) / 0.5f;
        o[plane] = ((float)prep_clip8(s1) / 255.0f - 0.5f) / 0.5f;
        o[2 * plane] = ((float)prep_clip8(s2) / 255.0f - 0.5f) / 0.5f;
    }
}

// Pillow's coefficient construction for a bilinear resize of a whole axis (host, float64 — same operation order).
static int prep_ksize(int in_size, int out_size) {
    double scale = (double)in_size / (double)out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(1.0 * filterscale) * 2 + 1;
}

extern "C" int mil_resize_plan(int in_size, int out_size, int* ksize) {
    if (in_size <= 0 || out_size <= 0 || !ksize) return MIL_ERR_ARG;
    *ksize = prep_ksize(in_size, out_size);
    return MIL_OK;
}

extern "C" int mil_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk) {
    if (in_size <= 0 || out_size <= 0 || !bounds || !kk) return MIL_ERR_ARG;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    double* w = new double[ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double v = (x + xmin - center + 0.5) * ss;
            if (v < 0.0) v = -v;
            w[x] = v < 1.0 ? 1.0 - v : 0.0;
            ww += w[x];
        }
        for (int x = 0; x < ksize; ++x) {
            double c = 0.0;
            if (x < xmax) c = (ww != 0.0) ? w[x] / ww : w[x];
            kk[(size_t)xx * ksize + x] = c < 0 ? (int32_t)(-0.5 + c * (double)(1 << PREP_BITS)) : (int32_t)(0.5 + c * (double)(1 << PREP_BITS));
        }
        bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    }
    delete[] w;
    return MIL_OK;
}

// out [T,3,R,R] fp32 in [-1,1] from uint8 ROIs [T,S,S,3].  params [T,4] (top, left, hflip, vflip with top/left in
// [0, 2*pad]) selects the train chain, null the flat chain.  bounds_host is the host copy of the table (the launcher
// sizes the LDS strip from it); bounds_dev / kk_dev are its device copies (mil_resize_coeffs(S, R)).
static int prep_entry(const uint8_t* rois, const int32_t* params, const int32_t* bounds_host, const int32_t* bounds_dev,
                      const int32_t* kk_dev, float* out, void* xs, int T, int S, int pad, int R, void* stream) {
    if (!rois || !bounds_host || !bounds_dev || !kk_dev || (!out && !xs) || T < 0 || S <= 0 || R <= 0 || pad < 0) return MIL_ERR_ARG;
    if (xs && ((R & 1) || (reinterpret_cast<uintptr_t>(xs) & 15))) return MIL_ERR_UNSUPPORTED;      // 2x2 blocks, 16-byte records
    if (T == 0) return MIL_OK;
    PrepArgs a{};
    a.rois = rois; a.params = params; a.bounds = bounds_dev; a.kk = kk_dev; a.out = out; a.xs = static_cast<__bf16*>(xs);
    a.T = T; a.S = S; a.pad = params ? pad : 0; a.R = R; a.ksize = prep_ksize(S, R);
    int nri_max = 0;
    for (int r0 = 0; r0 < R; r0 += PREP_BAND) {
        const int r1 = r0 + PREP_BAND < R ? r0 + PREP_BAND : R;
        const int n = bounds_host[2 * (r1 - 1)] + bounds_host[2 * (r1 - 1) + 1] - bounds_host[2 * r0];
        if (n > nri_max) nri_max = n;
    }
    a.margin = (3 * a.pad + 15) & ~15;
    a.row_pitch = (a.margin + S * 3 + 3 * a.pad + 3 * 32 + 8 + 15) & ~15;       // + slack read behind the last window
    a.lds_h_off = PREP_ROWS * a.row_pitch;
    a.lds_k_off = (a.lds_h_off + nri_max * R * 3 + 15) & ~15;
    a.lds_b_off = a.lds_k_off + R * a.ksize * 4;
    const int lds = a.lds_b_off + 2 * R * 4;
    if (lds > 160 * 1024 || T > 65535 || a.ksize > 32) return MIL_ERR_UNSUPPORTED;
    auto kern = a.ksize <= 4 ? tile_preprocess_kernel<4> : a.ksize <= 8 ? tile_preprocess_kernel<8> : a.ksize <= 12 ? tile_preprocess_kernel<12>
              : a.ksize <= 16 ? tile_preprocess_kernel<16> : a.ksize <= 24 ? tile_preprocess_kernel<24> : tile_preprocess_kernel<32>;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return MIL_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((R + PREP_BAND - 1) / PREP_BAND, T), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_tile_preprocess(const uint8_t* rois, const int32_t* params, const int32_t* bounds_host, const int32_t* bounds_dev,
                                   const int32_t* kk_dev, float* out, int T, int S, int pad, int R, void* stream) {
    if (!out) return MIL_ERR_ARG;
    return prep_entry(rois, params, bounds_host, bounds_dev, kk_dev, out, nullptr, T, S, pad, R, stream);
}

// The same chain with the output written as the bf16 space-to-depth NHWC tensor xs [T,R/2,R/2,16] the stem kernels read
// (mil_stem_fwd_fused_xs / mil_stem_bwd_fused): a pre-processed bag then never exists as fp32 — the values are the bf16
// roundings of mil_tile_preprocess's, i.e. exactly what the stem's own fp32 -> bf16 conversion produces.  R must be even.
extern "C" int mil_tile_preprocess_s2d(const uint8_t* rois, const int32_t* params, const int32_t* bounds_host, const int32_t* bounds_dev,
                                       const int32_t* kk_dev, void* xs, int T, int S, int pad, int R, void* stream) {
    if (!xs) return MIL_ERR_ARG;
    return prep_entry(rois, params, bounds_host, bounds_dev, kk_dev, nullptr, xs, T, S, pad, R, stream);
}
