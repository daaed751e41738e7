// Bandwidth-bound helpers around the MFMA kernels: input space-to-depth packing, weight packing into
// MFMA fragment order, max-pool (fwd + bwd fused with the stem's LeakyReLU mask), global average pool
// + bias-free linear (fwd + bwd).  Reference ops: gbm/model.py:24-26 (stem conv/LeakyReLU/MaxPool2d),
// gbm/model.py:31-32,58-60 (AdaptiveAvgPool2d + fc).
#include "geom.cuh"
#include "pack.cuh"
#include "pf_common.cuh"

// ---------------------------------------------------------------------------------------------
// fp32 NCHW [n,3,H,W]  ->  NHWC space-to-depth [n, ceil(H/2), ceil(W/2), 16]; channel = c*4 + dy*2 + dx
// (12 real channels, 4 zero).  Turns the 7x7 stride-2 stem into a 4x4 stride-1 conv on 16-B pieces.
template <typename T>
__global__ void stem_s2d_kernel(const float* __restrict__ x, typename T::elem* __restrict__ out, int n, int H, int W,
                                int H2, int W2) {
    const size_t total = (size_t)n * H2 * W2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x2 = (int)(idx % W2);
        const size_t r = idx / W2;
        const int y2 = (int)(r % H2);
        const int img = (int)(r / H2);
        float v0[8], v1[8];
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) {
            float val = 0.f;
            if (ch < 12) {
                const int c = ch >> 2, dy = (ch >> 1) & 1, dx = ch & 1;
                const int iy = 2 * y2 + dy, ix = 2 * x2 + dx;
                if (iy < H && ix < W) val = x[(((size_t)img * 3 + c) * H + iy) * W + ix];
            }
            if (ch < 8) v0[ch] = val; else v1[ch - 8] = val;
        }
        store8<T>(out + idx * 16, v0);
        store8<T>(out + idx * 16 + 8, v1);
    }
}

extern "C" int mil_stem_s2d(const float* x_nchw, void* out, int n, int H, int W, int dtype, void* stream) {
    if (!x_nchw || !out || n < 0 || H <= 0 || W <= 0) return MIL_ERR_ARG;
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const size_t total = (size_t)n * H2 * W2;
    if (total == 0) return MIL_OK;
    int grid = (int)((total + 255) / 256);
    if (grid > 65536) grid = 65536;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIL_DT_BF16) hipLaunchKernelGGL(stem_s2d_kernel<BF16>, dim3(grid), dim3(256), 0, st, x_nchw, (__bf16*)out, n, H, W, H2, W2);
    else if (dtype == MIL_DT_F32) hipLaunchKernelGGL(stem_s2d_kernel<F32>, dim3(grid), dim3(256), 0, st, x_nchw, (float*)out, n, H, W, H2, W2);
    else return MIL_ERR_ARG;
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// ---------------------------------------------------------------------------------------------
// fp32 master weights [Cout][Cin][k][k] -> MFMA fragment order: index maps and job record in pack.cuh.
__global__ void pack_weights_kernel(PackJob j) {
    const int total = j.nsteps * j.NT * 64 * 8;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) mil_pack_job_elem(j, idx);
}

extern "C" int mil_packed_weight_elems(size_t* elems, int cout, int cin, int ks, int mode) {
    if (!elems || cout <= 0 || cin <= 0) return MIL_ERR_ARG;
    PackJob j{};
    j.cout = cout; j.cin = cin; j.ks = ks; j.mode = mode;
    mil_pack_job_dims(&j);
    *elems = (size_t)j.nsteps * j.NT * 64 * 8;
    return MIL_OK;
}

extern "C" int mil_pack_conv_weights(const float* w, const float* bias, void* wpack, float* bias_pad, int cout, int cin,
                                     int ks, int mode, int dtype, void* stream) {
    if (!w || !wpack || cout <= 0 || cin <= 0 || mode < 0 || mode > 3 || (mode == 3 && ks != 3)) return MIL_ERR_ARG;
    if (dtype != MIL_DT_BF16 && dtype != MIL_DT_F32 && dtype != MIL_DT_F32S) return MIL_ERR_ARG;
    PackJob j{};
    j.w = w; j.bias = bias; j.out = wpack; j.bias_pad = bias_pad;
    j.cout = cout; j.cin = cin; j.ks = ks; j.mode = mode; j.dtype = dtype;
    mil_pack_job_dims(&j);
    const int total = j.nsteps * j.NT * 64 * 8;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), j);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, pad 1), NHWC; -inf padding; first maximum in (ky,kx) scan order wins
// (torch semantics).  Records per output element, for the backward pass, the winning tap (bits 0-3) and
// whether the winning value is <= 0 (bit 4): the input is a LeakyReLU output, so that bit is the lrelu'
// mask of the only input element that receives this window's gradient and the backward pass never has to
// re-read the (4x larger) input tensor.
template <typename T>
__global__ void maxpool_fwd_kernel(const typename T::elem* __restrict__ x, typename T::elem* __restrict__ y,
                                   uint8_t* __restrict__ widx, int n, int H, int W, int Ho, int Wo, int CP) {
    const int ng = CP / 8;
    const size_t total = (size_t)n * Ho * Wo * ng;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % ng);
        size_t r = idx / ng;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int img = (int)(r / Ho);
        float best[8];
        uint8_t bi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bi[j] = 0; }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                float v[8];
                load8<T>(x + (((size_t)img * H + iy) * W + ix) * CP + c8 * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (v[j] > best[j]) { best[j] = v[j]; bi[j] = (uint8_t)(ky * 3 + kx); }
            }
        }
        const size_t o = (((size_t)img * Ho + oy) * Wo + ox) * CP + c8 * 8;
        store8<T>(y + o, best);
#pragma unroll
        for (int j = 0; j < 8; ++j) bi[j] |= (best[j] > 0.f) ? 0 : 16;
        uint2 packed;
        packed.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((uint32_t)bi[3] << 24);
        packed.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((uint32_t)bi[7] << 24);
        *reinterpret_cast<uint2*>(widx + o) = packed;
    }
}

// gx[y,x,c] = lrelu'(x[y,x,c]) * sum over the (<=4) pooling windows that contain (y,x) and whose recorded
// winner is (y,x) of gy[window].  Gather form: no atomics, deterministic.  One thread owns the 2x2 block of
// input pixels (2by..2by+1, 2bx..2bx+1) for 8 channels: those four pixels are covered by exactly the four
// windows (by..by+1, bx..bx+1) — the even/even pixel by one of them, the mixed ones by two, the odd/odd one
// by all four — so the four winner records and gradients are loaded once and 9 tap tests replace 16.
// With use_mask the lrelu' factor comes from bit 4 of the winner record.
template <typename T>
__global__ void maxpool_bwd_kernel(const typename T::elem* __restrict__ gy, const uint8_t* __restrict__ widx,
                                   typename T::elem* __restrict__ gx, int n, int H, int W, int Ho, int Wo, int CP,
                                   float slope, int use_mask) {
    const int ng = CP / 8;
    const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;
    const size_t total = (size_t)n * Hb * Wb * ng;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % ng);
        size_t r = idx / ng;
        const int bx = (int)(r % Wb); r /= Wb;
        const int by = (int)(r % Hb);
        const int img = (int)(r / Hb);
        float g[2][2][8];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 8; ++j) g[a][b][j] = 0.f;
#pragma unroll
        for (int wy = 0; wy < 2; ++wy) {
            const int oy = by + wy;
            if (oy >= Ho) continue;
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int ox = bx + wx;
                if (ox >= Wo) continue;
                const size_t o = (((size_t)img * Ho + oy) * Wo + ox) * CP + c8 * 8;
                const uint2 packed = *reinterpret_cast<const uint2*>(widx + o);
                float gv[8];
                load8<T>(gy + o, gv);
                // window (oy,ox) covers rows 2oy-1..2oy+1: block row dy (pixel 2by+dy) is its tap row ky below
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const int ky = 2 * by + dy - (2 * oy - 1);          // wy=0: 1,2   wy=1: -1,0
                    if (ky < 0 || ky > 2) continue;
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int kx = 2 * bx + dx - (2 * ox - 1);
                        if (kx < 0 || kx > 2) continue;
                        const uint32_t me = (uint32_t)(ky * 3 + kx);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const uint32_t w = ((j < 4 ? packed.x : packed.y) >> (8 * (j & 3))) & 0xffu;
                            if ((w & 15u) == me) g[dy][dx][j] += (use_mask && (w & 16u)) ? gv[j] * slope : gv[j];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int y = 2 * by + dy;
            if (y >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int x = 2 * bx + dx;
                if (x >= W) continue;
                store8<T>(gx + (((size_t)img * H + y) * W + x) * CP + c8 * 8, g[dy][dx]);
            }
        }
    }
}

static int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

extern "C" int mil_maxpool_fwd(const void* x, void* y, uint8_t* widx, int n, int H, int W, int cp, int dtype, void* stream) {
    if (!x || !y || !widx || cp % 8) return MIL_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)n * Ho * Wo * (cp / 8);
    if (!total) return MIL_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIL_DT_BF16) hipLaunchKernelGGL(maxpool_fwd_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, st, (const __bf16*)x, (__bf16*)y, widx, n, H, W, Ho, Wo, cp);
    else if (dtype == MIL_DT_F32) hipLaunchKernelGGL(maxpool_fwd_kernel<F32>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, (float*)y, widx, n, H, W, Ho, Wo, cp);
    else return MIL_ERR_ARG;
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_maxpool_bwd(const void* gy, const uint8_t* widx, void* gx, int n, int H, int W, int cp,
                               int apply_lrelu_mask, float slope, int dtype, void* stream) {
    if (!gy || !widx || !gx || cp % 8) return MIL_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)n * ((H + 1) / 2) * ((W + 1) / 2) * (cp / 8);
    if (!total) return MIL_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIL_DT_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, st, (const __bf16*)gy, widx, (__bf16*)gx, n, H, W, Ho, Wo, cp, slope, apply_lrelu_mask);
    else if (dtype == MIL_DT_F32) hipLaunchKernelGGL(maxpool_bwd_kernel<F32>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)gy, widx, (float*)gx, n, H, W, Ho, Wo, cp, slope, apply_lrelu_mask);
    else return MIL_ERR_ARG;
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// ---------------------------------------------------------------------------------------------
// Global average pool over h*w + linear: feats[t][o] = bias[o]? + sum_i mean_p(x[t][p][i]) * Wfc[o][i].
// One workgroup per tile (blockDim = max(C, NF) rounded up to a wave, <= 512); pooled means are kept (fp32)
// for the backward pass.  gbm/model.py:31-32,58-60 (C = 80, no bias); alt_resnet.py:92-93,139-143 (C = 512, bias).
#define POOL_MAXC 512
// The tile's pixels are staged in LDS as floats in chunks (all threads load 16-byte pieces), then one thread per channel
// adds its column IN PIXEL ORDER — the order of the one-thread-per-channel form this replaces (2-byte loads, 0.8 TB/s): same
// bits (the 2-instance bags of the live driver make the head's batch norm ill-conditioned enough for another summation
// order to show in the gradients) — and the C x NF matrix-vector product follows from LDS.
template <typename T>
__global__ void avgpool_fc_fwd_kernel(const typename T::elem* __restrict__ x, const float* __restrict__ wfc,
                                      const float* __restrict__ bias, float* __restrict__ pooled,
                                      float* __restrict__ feats, int hw, int CP, int C, int NF) {
    __shared__ float stage[4096];                            // [pixels of a chunk][CP]
    __shared__ float sp[POOL_MAXC];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int NG = CP >> 3, PCH = 4096 / CP;
    float s0 = 0.f, s1 = 0.f;                                // channels tid and tid + blockDim (CP <= 2 * blockDim)
    for (int i0 = 0; i0 < hw; i0 += PCH) {
        const int npx = hw - i0 < PCH ? hw - i0 : PCH;
        for (int idx = tid; idx < npx * NG; idx += blockDim.x) {
            const int i = idx / NG, cg = idx - i * NG;
            float v[8];
            load8<T>(x + ((size_t)t * hw + i0 + i) * CP + cg * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) stage[i * CP + cg * 8 + j] = v[j];
        }
        __syncthreads();
        if (tid < C) for (int i = 0; i < npx; ++i) s0 += stage[i * CP + tid];
        if (tid + (int)blockDim.x < C) for (int i = 0; i < npx; ++i) s1 += stage[i * CP + tid + blockDim.x];
        __syncthreads();
    }
    for (int c = tid; c < POOL_MAXC; c += blockDim.x) {
        float s = 0.f;
        if (c < C) {
            s = (c == tid ? s0 : s1) / (float)hw;
            pooled[(size_t)t * C + c] = s;
        }
        sp[c] = s;
    }
    __syncthreads();
    for (int c = tid; c < NF; c += blockDim.x) {
        float acc = bias ? bias[c] : 0.f;
        for (int i = 0; i < C; ++i) acc += sp[i] * wfc[(size_t)c * C + i];
        feats[(size_t)t * NF + c] = acc;
    }
}

// dz[t][p][c] = lrelu'(act[t][p][c]) * (sum_o dfeats[t][o] * Wfc[o][c]) / hw      (padded channels -> 0)
// g[c] first (one thread per channel), then the map in 8-channel pieces.
template <typename T>
__global__ void avgpool_fc_bwd_kernel(const float* __restrict__ dfeats, const float* __restrict__ wfc,
                                      const typename T::elem* __restrict__ act, typename T::elem* __restrict__ dz, int hw,
                                      int CP, int C, int NF, float slope) {
    __shared__ float sd[POOL_MAXC], sg[POOL_MAXC];
    const int t = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < POOL_MAXC; c += blockDim.x) sd[c] = (c < NF) ? dfeats[(size_t)t * NF + c] : 0.f;
    __syncthreads();
    for (int c = tid; c < CP; c += blockDim.x) {
        float g = 0.f;
        if (c < C) {
            for (int o = 0; o < NF; ++o) g += sd[o] * wfc[(size_t)o * C + c];
            g /= (float)hw;
        }
        sg[c] = g;
    }
    __syncthreads();
    const int NG = CP >> 3;
    for (int idx = tid; idx < hw * NG; idx += blockDim.x) {
        const int i = idx / NG, cg = idx - i * NG;
        const size_t off = ((size_t)t * hw + i) * CP + cg * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sg[cg * 8 + j];
        if (act) {
            float a[8];
            load8<T>(act + off, a);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= lrelu_grad(a[j], slope);
        }
        store8<T>(dz + off, v);
    }
}

// dWfc[o][i] = sum_t dfeats[t][o] * pooled[t][i]; elements [NF*C, NF*C+NF) are dbias[o] = sum_t dfeats[t][o].
// One wave per output element: lanes stride the tiles, then a fixed shuffle tree (deterministic).
__global__ __launch_bounds__(256) void fc_wgrad_kernel(const float* __restrict__ dfeats, const float* __restrict__ pooled,
                                                       float* __restrict__ dw, float* __restrict__ dbias, int n, int C, int NF,
                                                       int accumulate) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nw = NF * C;
    if (idx >= nw + (dbias ? NF : 0)) return;
    float s = 0.f;
    if (idx < nw) {
        const int o = idx / C, i = idx - o * C;
        for (int t = lane; t < n; t += 64) s += dfeats[(size_t)t * NF + o] * pooled[(size_t)t * C + i];
    } else {
        const int o = idx - nw;
        for (int t = lane; t < n; t += 64) s += dfeats[(size_t)t * NF + o];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane != 0) return;
    float* q = idx < nw ? dw + idx : dbias + (idx - nw);
    *q = accumulate ? *q + s : s;
}

// Two-stage form (see head_wgrad_partial_kernel in mil_head.hip): a workgroup stages 64 rows of [dfeats | pooled | 1] in
// LDS with coalesced loads and accumulates every output element over them in index order; the second stage adds the
// slices in index order (and accumulates into dw/dbias when asked).
#define FCS 64
__global__ __launch_bounds__(1024) void fc_wgrad_partial_kernel(const float* __restrict__ dfeats, const float* __restrict__ pooled,
                                                               float* __restrict__ partial, int n, int C, int NF, int want_bias) {
    extern __shared__ float frows[];                        // [FCS][NF + C + 1 (+pad)]
    MIL_POISON(frows);
    const int stride = (NF + C + 2) & ~1;
    const int tid = threadIdx.x, n0 = blockIdx.x * FCS;
    for (int idx = tid; idx < FCS * NF; idx += 1024) {
        const int nl = idx / NF, o = idx - nl * NF;
        frows[nl * stride + o] = (n0 + nl < n) ? dfeats[(size_t)(n0 + nl) * NF + o] : 0.f;
    }
    for (int idx = tid; idx < FCS * C; idx += 1024) {
        const int nl = idx / C, i = idx - nl * C;
        frows[nl * stride + NF + i] = (n0 + nl < n) ? pooled[(size_t)(n0 + nl) * C + i] : 0.f;
    }
    for (int nl = tid; nl < FCS; nl += 1024) frows[nl * stride + NF + C] = (n0 + nl < n) ? 1.f : 0.f;
    __syncthreads();
    // blockIdx.y = a contiguous share of the output elements (a bag of 256 tiles is four row slices: as ONE block per slice the
    // launch ran on 4 of 256 CUs, 100 us for 0.26 GFLOP); every element keeps its summation order
    const int total = NF * C + (want_bias ? NF : 0);
    const int share = ((total + (int)gridDim.y - 1) / (int)gridDim.y + 1023) & ~1023;
    const int e_end = min(total, ((int)blockIdx.y + 1) * share);
    float* out = partial + (size_t)blockIdx.x * total;
    for (int e = blockIdx.y * share + tid; e < e_end; e += 1024) {
        int ca, cb;
        if (e < NF * C) { ca = e / C; cb = NF + (e - ca * C); } else { ca = e - NF * C; cb = NF + C; }
        float sacc = 0.f;
#pragma unroll 8
        for (int nl = 0; nl < FCS; ++nl) sacc += frows[nl * stride + ca] * frows[nl * stride + cb];
        out[e] = sacc;
    }
}

__global__ __launch_bounds__(256) void fc_wgrad_reduce_kernel(const float* __restrict__ partial, int nslices, int total, int nw,
                                                              float* __restrict__ dw, float* __restrict__ dbias, int accumulate) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float sacc = 0.f;
    for (int k = 0; k < nslices; ++k) sacc += partial[(size_t)k * total + e];
    float* q = e < nw ? dw + e : dbias + (e - nw);
    *q = accumulate ? *q + sacc : sacc;
}

extern "C" int mil_fc_wgrad_workspace(size_t* bytes, int n, int c, int nf) {
    if (!bytes || n < 0 || c <= 0 || nf <= 0) return MIL_ERR_ARG;
    *bytes = (size_t)((n + FCS - 1) / FCS) * ((size_t)nf * c + nf) * sizeof(float);
    return MIL_OK;
}

// dwfc[nf][c] (+)= dfeats^T pooled, dbias[nf] (+)= column sums of dfeats (dbias may be null).
extern "C" int mil_fc_wgrad(const float* dfeats, const float* pooled, float* dwfc, float* dbias, void* workspace,
                            size_t workspace_bytes, int n, int c, int nf, int accumulate, void* stream) {
    if (!dfeats || !pooled || !dwfc || !workspace || n < 0 || c <= 0 || nf <= 0 || c > POOL_MAXC || nf > POOL_MAXC) return MIL_ERR_ARG;
    size_t need = 0;
    mil_fc_wgrad_workspace(&need, n, c, nf);
    if (workspace_bytes < need) return MIL_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nslices = (n + FCS - 1) / FCS;
    const int total = nf * c + (dbias ? nf : 0);
    const int lds = FCS * ((nf + c + 2) & ~1) * 4;
    if (lds > 160 * 1024) return MIL_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(fc_wgrad_partial_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return MIL_ERR_LAUNCH;
    if (nslices > 0) {
        int ny = (2 * mil_num_cus() + nslices - 1) / nslices;                     // about two blocks per CU in all
        const int max_y = (total + 1023) / 1024;
        if (ny > max_y) ny = max_y;
        if (ny < 1) ny = 1;
        hipLaunchKernelGGL(fc_wgrad_partial_kernel, dim3(nslices, ny), dim3(1024), lds, st, dfeats, pooled, (float*)workspace, n, c, nf, dbias ? 1 : 0);
        MIL_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(fc_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)workspace, nslices, total, nf * c, dwfc, dbias, accumulate);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// 256 threads (512 for more than 256 channels): (pixel group, 8-channel group) items; needs cp % 8 == 0
static int pool_block(int c, int nf, int cp) {
    int m = c > nf ? c : nf;
    if (cp > m) m = cp;
    return m > 256 ? 512 : 256;
}

extern "C" int mil_avgpool_fc_fwd(const void* x, const float* wfc, const float* bias, float* pooled, float* feats, int n,
                                  int hw, int cp, int c, int nf, int dtype, void* stream) {
    if (!x || !wfc || !pooled || !feats || c > POOL_MAXC || nf > POOL_MAXC || cp > POOL_MAXC || (cp & 7) || hw <= 0) return MIL_ERR_ARG;
    if (n == 0) return MIL_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int bd = pool_block(c, nf, cp);
    if (dtype == MIL_DT_BF16) hipLaunchKernelGGL(avgpool_fc_fwd_kernel<BF16>, dim3(n), dim3(bd), 0, st, (const __bf16*)x, wfc, bias, pooled, feats, hw, cp, c, nf);
    else if (dtype == MIL_DT_F32) hipLaunchKernelGGL(avgpool_fc_fwd_kernel<F32>, dim3(n), dim3(bd), 0, st, (const float*)x, wfc, bias, pooled, feats, hw, cp, c, nf);
    else return MIL_ERR_ARG;
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_avgpool_fc_bwd(const float* dfeats, const float* wfc, const float* pooled, const void* act, void* dz,
                                  float* dwfc, float* dbias, int n, int hw, int cp, int c, int nf, int accumulate, float slope,
                                  int dtype, void* stream) {
    // dwfc == null: data path only (the parameter gradients then come from mil_fc_wgrad)
    if (!dfeats || !wfc || !pooled || !dz || c > POOL_MAXC || nf > POOL_MAXC || cp > POOL_MAXC || (cp & 7) || hw <= 0) return MIL_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int bd = pool_block(c, nf, cp);
    if (n > 0) {
        if (dtype == MIL_DT_BF16) hipLaunchKernelGGL(avgpool_fc_bwd_kernel<BF16>, dim3(n), dim3(bd), 0, st, dfeats, wfc, (const __bf16*)act, (__bf16*)dz, hw, cp, c, nf, slope);
        else if (dtype == MIL_DT_F32) hipLaunchKernelGGL(avgpool_fc_bwd_kernel<F32>, dim3(n), dim3(bd), 0, st, dfeats, wfc, (const float*)act, (float*)dz, hw, cp, c, nf, slope);
        else return MIL_ERR_ARG;
        MIL_CHECK_LAUNCH();
    }
    if (dwfc) {
        const int total = nf * c + (dbias ? nf : 0);
        hipLaunchKernelGGL(fc_wgrad_kernel, dim3((total + 3) / 4), dim3(256), 0, st, dfeats, pooled, dwfc, dbias, n, c, nf, accumulate);
        MIL_CHECK_LAUNCH();
    }
    return MIL_OK;
}
