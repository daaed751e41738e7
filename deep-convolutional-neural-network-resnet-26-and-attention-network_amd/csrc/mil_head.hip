// Attention-MIL head, forward + backward, fp32, segmented over the bags of a batch.
// Reference arithmetic: gbm/model.py:198-246 (`Attention.forward` after the backbone),
// gbm/model.py:108-111 (`ContextLayer`), nnBlocks.py:71-85,121-134 (label-smoothed soft-target CE).
//
//   per bag b with instances n in [off[b], off[b+1]):
//     Hz  = gamma*(H-mean_n)/sqrt(var_n+eps)+beta          (batch statistics over the bag's instances)
//     Hm  = lrelu(H) * keep/(1-p)                          (keep == null  <=>  eval)
//     A   = W2 tanh(W1 Hz + b1) + b2 ;  Am = sig(-10w)*softplus(A) + sig(10w) ;  A1 = Am / sum_n|Am|
//     B   = Wc lrelu(Wl Hm + bl) + bc ;  M = A1^T B ;  loss = sum_y t_y cw_y (lse(M) - M_y)
//
// All reductions over instances are per bag (never across bags), in fixed order (deterministic).
#include "common.cuh"

#define HL 80
#define HD 40
#define HK 3

struct HeadWeights {
    const float *bn_w, *bn_b, *a_w1, *a_b1, *a_w2, *a_b2, *b_w1, *b_b1, *b_wc, *b_bc, *wmask;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__device__ __forceinline__ float block_sum(float v, float* red) {
    // block reduction (<= 16 waves), wave sums added in wave order, result broadcast to every thread
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = red[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) s += red[k];
    return s;
}

// K1: per-bag column statistics.  stats[b] = { mu[80], rstd[80] }, kld[b] = 0.5*mean(H^2)
// One workgroup per bag; HP row partials per column (thread = (column, partial p), rows p, p+HP, ...), added in p order:
// the tree depends only on the bag's length.  (Three partials on 256 threads took 33 us for 256-instance bags — one serial
// chain of 85 loads per thread — and 0.4 ms for a 4096-instance bag.)
#define HP 12
__global__ __launch_bounds__(1024) void head_colstats_kernel(const float* __restrict__ H, const int* __restrict__ off,
                                                             float* __restrict__ stats, float* __restrict__ kld, float eps) {
    __shared__ float part[HP][HL];
    __shared__ float mu_s[HL];
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n0 = off[b], n1 = off[b + 1], N = n1 - n0;
    const int i = tid % HL, p = tid / HL;
    float s = 0.f, sq = 0.f;
    if (p < HP) for (int n = n0 + p; n < n1; n += HP) { const float v = H[(size_t)n * HL + i]; s += v; sq += v * v; }
    if (p < HP) part[p][i] = s;
    const float sq_tot = block_sum(p < HP ? sq : 0.f, red);
    __syncthreads();
    if (tid < HL) {
        float t = part[0][tid];
#pragma unroll
        for (int k = 1; k < HP; ++k) t += part[k][tid];
        mu_s[tid] = t / (float)N;
    }
    __syncthreads();
    float v2 = 0.f;
    if (p < HP) { const float m = mu_s[i]; for (int n = n0 + p; n < n1; n += HP) { const float d = H[(size_t)n * HL + i] - m; v2 += d * d; } }
    __syncthreads();
    if (p < HP) part[p][i] = v2;
    __syncthreads();
    if (tid < HL) {
        float t = part[0][tid];
#pragma unroll
        for (int k = 1; k < HP; ++k) t += part[k][tid];
        const float var = t / (float)N;
        stats[(size_t)b * 2 * HL + tid] = mu_s[tid];
        stats[(size_t)b * 2 * HL + HL + tid] = 1.f / sqrtf(var + eps);
    }
    if (tid == 0) kld[b] = 0.5f * sq_tot / ((float)N * HL);
}

// K2: per-instance forward of both MLPs.  Saves t=tanh(u) [n,40], v (buffer pre-activation) [n,40],
// A_raw [n,3], B [n].
// One WAVE per slice of HSL instances, lane j < 40 = hidden unit j of both MLPs with its two weight rows (160 values) in
// registers; an instance's normalised / dropped-out feature rows are staged in the wave's LDS strip and read back as
// broadcasts.  Every dot product runs in the same order as the one-thread-per-instance form it replaces (which spent 44 us
// on 16 workgroups: 6.4 k serial FMAs per thread, each with its weight read from LDS): same bits, ~256 waves.
#ifndef HSL
#define HSL 8
#endif
__global__ __launch_bounds__(256) void head_inst_fwd_kernel(const float* __restrict__ H, const int* __restrict__ inst_bag,
                                                            const float* __restrict__ stats, const uint8_t* __restrict__ keep,
                                                            HeadWeights w, float* __restrict__ t_out, float* __restrict__ v_out,
                                                            float* __restrict__ araw, float* __restrict__ bterm, int ntot,
                                                            float slope, float keep_scale) {
    __shared__ __attribute__((aligned(16))) float strip[4][2 * HL + 2 * HD];      // per wave: z[80] m[80] t[40] lv[40]
    __shared__ float w2[HK * HD], wc[HD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < HK * HD; i += 256) w2[i] = w.a_w2[i];
    if (tid < HD) wc[tid] = w.b_wc[tid];
    const int j = lane < HD ? lane : 0;
    float w1r[HL], wlr[HL];
#pragma unroll
    for (int i = 0; i < HL; i += 4) {
        const f32x4_t q1 = *reinterpret_cast<const f32x4_t*>(w.a_w1 + j * HL + i);
        const f32x4_t q2 = *reinterpret_cast<const f32x4_t*>(w.b_w1 + j * HL + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) { w1r[i + e] = q1[e]; wlr[i + e] = q2[e]; }
    }
    const float b1 = w.a_b1[j], bl = w.b_b1[j];
    // feature columns of this lane: i0 = lane, i1 = 64 + lane (lane < 16)
    const int i1 = lane < HL - 64 ? 64 + lane : 0;
    const float g0 = w.bn_w[lane], be0 = w.bn_b[lane], g1 = w.bn_w[i1], be1 = w.bn_b[i1];
    float* z = strip[wave];
    float* m = z + HL;
    float* ts = m + HL;
    float* lv = ts + HD;
    __syncthreads();
    const int base = (blockIdx.x * 4 + wave) * HSL;
    for (int sl = 0; sl < HSL; ++sl) {
        const int n = base + sl;
        const bool live = n < ntot;
        if (live) {
            const float* st = stats + (size_t)inst_bag[n] * 2 * HL;
            const float h0 = H[(size_t)n * HL + lane];
            z[lane] = g0 * ((h0 - st[lane]) * st[HL + lane]) + be0;
            float m0 = lrelu(h0, slope);
            if (keep) m0 = keep[(size_t)n * HL + lane] ? m0 * keep_scale : 0.f;
            m[lane] = m0;
            if (lane < HL - 64) {
                const float h1 = H[(size_t)n * HL + i1];
                z[i1] = g1 * ((h1 - st[i1]) * st[HL + i1]) + be1;
                float m1 = lrelu(h1, slope);
                if (keep) m1 = keep[(size_t)n * HL + i1] ? m1 * keep_scale : 0.f;
                m[i1] = m1;
            }
        }
        __syncthreads();
        if (live && lane < HD) {
            float u = b1, v = bl;
#pragma unroll
            for (int i = 0; i < HL; i += 4) {
                const f32x4_t zq = *reinterpret_cast<const f32x4_t*>(z + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) u += w1r[i + e] * zq[e];
            }
#pragma unroll
            for (int i = 0; i < HL; i += 4) {
                const f32x4_t mq = *reinterpret_cast<const f32x4_t*>(m + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) v += wlr[i + e] * mq[e];
            }
            const float t = tanhf(u);
            t_out[(size_t)n * HD + lane] = t;
            v_out[(size_t)n * HD + lane] = v;
            ts[lane] = t;
            lv[lane] = lrelu(v, slope);
        }
        __syncthreads();
        if (live && lane < HK) {                  // A_raw[k] = b2[k] + sum_j w2[k][j] t[j], in j order
            float a = w.a_b2[lane];
            for (int jj = 0; jj < HD; ++jj) a += w2[lane * HD + jj] * ts[jj];
            araw[(size_t)n * HK + lane] = a;
        } else if (live && lane == HK) {          // B = bc + sum_j wc[j] lrelu(v[j]), in j order
            float bsum = w.b_bc[0];
            for (int jj = 0; jj < HD; ++jj) bsum += wc[jj] * lv[jj];
            bterm[n] = bsum;
        }
    }
}

// Per-bag scalar record written by K3 and read back by the host wrapper / backward kernels.
//   [0..2] Mterm  [3..5] y_pred  [6] loss  [7] error  [8] Aterm_mu  [9] Aterm_var  [10..12] D (L1 norms)
//   [13..15] dloss/dM (for unit upstream grad)  [16] y_hat (as float)  [17] l2 (bag 0 only)
#define HREC 24

__global__ __launch_bounds__(256) void head_bag_fwd_kernel(const float* __restrict__ araw, const float* __restrict__ bterm,
                                                           const int* __restrict__ off, const int64_t* __restrict__ label,
                                                           const float* __restrict__ cw, HeadWeights w, float smoothing,
                                                           float* __restrict__ a1, float* __restrict__ wrois,
                                                           float* __restrict__ rec) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n0 = off[b], n1 = off[b + 1], N = n1 - n0;
    float s0[HK], s1[HK];
#pragma unroll
    for (int k = 0; k < HK; ++k) { s0[k] = sigmoidf_(-10.f * w.wmask[k]); s1[k] = sigmoidf_(10.f * w.wmask[k]); }
    float S[HK] = {0.f, 0.f, 0.f}, sa[HK] = {0.f, 0.f, 0.f}, cr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int n = n0 + tid; n < n1; n += 256) {
        float a[HK];
#pragma unroll
        for (int k = 0; k < HK; ++k) { a[k] = araw[(size_t)n * HK + k]; S[k] += fabsf(s0[k] * softplusf_(a[k]) + s1[k]); sa[k] += a[k]; }
        cr[0] += a[0] * a[0]; cr[1] += a[0] * a[1]; cr[2] += a[0] * a[2];
        cr[3] += a[1] * a[1]; cr[4] += a[1] * a[2]; cr[5] += a[2] * a[2];
    }
#pragma unroll
    for (int k = 0; k < HK; ++k) { S[k] = block_sum(S[k], red); sa[k] = block_sum(sa[k], red); }
#pragma unroll
    for (int k = 0; k < 6; ++k) cr[k] = block_sum(cr[k], red);
    float D[HK], M[HK] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < HK; ++k) D[k] = fmaxf(S[k], 1e-12f);
    for (int n = n0 + tid; n < n1; n += 256) {
        const float bv = bterm[n];
#pragma unroll
        for (int k = 0; k < HK; ++k) {
            const float v = (s0[k] * softplusf_(araw[(size_t)n * HK + k]) + s1[k]) / D[k];
            a1[(size_t)n * HK + k] = v;
            wrois[(size_t)3 * n0 + (size_t)k * N + (n - n0)] = v * bv;
            M[k] += v * bv;
        }
    }
#pragma unroll
    for (int k = 0; k < HK; ++k) M[k] = block_sum(M[k], red);
    float l2 = 0.f;
    if (b == 0) {
        float q1 = 0.f, q2 = 0.f;
        for (int i = tid; i < HD * HL; i += 256) { const float v = w.b_w1[i]; q1 += v * v; }
        if (tid < HD) { const float v = w.b_wc[tid]; q2 = v * v; }
        q1 = block_sum(q1, red); q2 = block_sum(q2, red);
        l2 = 0.5f * (sqrtf(q1) + sqrtf(q2));
    }
    if (tid == 0) {
        float* r = rec + (size_t)b * HREC;
        const float mx = fmaxf(M[0], fmaxf(M[1], M[2]));
        float e[HK], se = 0.f;
#pragma unroll
        for (int k = 0; k < HK; ++k) { e[k] = expf(M[k] - mx); se += e[k]; }
        const float lse = mx + logf(se);
        int yh = 0; float best = e[0];
#pragma unroll
        for (int k = 1; k < HK; ++k) if (e[k] > best) { best = e[k]; yh = k; }
        const int y = (int)label[b];
        float loss = 0.f, tw = 0.f, tk[HK];
#pragma unroll
        for (int k = 0; k < HK; ++k) {
            const float t = (k == y) ? 1.f - smoothing : smoothing / (HK - 1);
            tk[k] = t * (cw ? cw[k] : 1.f);
            tw += tk[k];
            loss += tk[k] * (lse - M[k]);
        }
        const float nrm[HK] = {fmaxf(sqrtf(cr[0]), 1e-12f), fmaxf(sqrtf(cr[3]), 1e-12f), fmaxf(sqrtf(cr[5]), 1e-12f)};
        const float avar = 2.f * (cr[1] / (nrm[0] * nrm[1]) + cr[2] / (nrm[0] * nrm[2]) + cr[4] / (nrm[1] * nrm[2])) / 9.f;
        float amu = 0.f;
#pragma unroll
        for (int k = 0; k < HK; ++k) { const float m = sa[k] / (float)N; amu += m * m; }
#pragma unroll
        for (int k = 0; k < HK; ++k) {
            r[k] = M[k]; r[3 + k] = e[k] / se; r[10 + k] = D[k];
            r[13 + k] = tw * (e[k] / se) - tk[k];
        }
        r[6] = loss; r[7] = (yh == y) ? 0.f : 1.f; r[8] = 0.5f * amu; r[9] = avar; r[16] = (float)yh;
        if (b == 0) r[17] = l2;
    }
}

// K4: per-instance backward.  Writes dA_raw-derived quantities for the weight-gradient pass and the
// buffer-branch part of dH; the BN-branch part is finished by K6.
// One wave per slice of HSL instances, as K2.  Phase A: lane j < 40 = hidden unit (du[j], dv[j]; the few per-instance
// scalars are computed by every lane).  Phase B: lane = feature column i (and 64 + i for i < 16) with that column of both
// weight matrices in registers; du / dv come back from the wave's LDS strip as broadcasts.  Same summation orders as the
// one-thread-per-instance form (62 us on 16 workgroups).
__global__ __launch_bounds__(256) void head_inst_bwd_kernel(const float* __restrict__ H, const int* __restrict__ inst_bag,
                                                            const uint8_t* __restrict__ keep, HeadWeights w,
                                                            const float* __restrict__ t_in, const float* __restrict__ v_in,
                                                            const float* __restrict__ araw, const float* __restrict__ bterm,
                                                            const float* __restrict__ rec, const float* __restrict__ gloss,
                                                            float* __restrict__ du_out, float* __restrict__ dv_out,
                                                            float* __restrict__ da_out, float* __restrict__ dwm_out,
                                                            float* __restrict__ db_out, float* __restrict__ dhz_out,
                                                            float* __restrict__ dH, int ntot, float slope, float keep_scale) {
    __shared__ __attribute__((aligned(16))) float strip[4][2 * HD];      // per wave: du[40] dv[40]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane < HD ? lane : 0;
    const float w2j[HK] = {w.a_w2[j], w.a_w2[HD + j], w.a_w2[2 * HD + j]};
    const float wcj = w.b_wc[j];
    const int i1 = lane < HL - 64 ? 64 + lane : 0;
    float c10[HD], cl0[HD], c11[HD], cl1[HD];        // columns lane and 64+lane of W1 and Wl
#pragma unroll
    for (int jj = 0; jj < HD; ++jj) {
        c10[jj] = w.a_w1[jj * HL + lane]; cl0[jj] = w.b_w1[jj * HL + lane];
        c11[jj] = w.a_w1[jj * HL + i1]; cl1[jj] = w.b_w1[jj * HL + i1];
    }
    float s0k[HK], s1k[HK];
#pragma unroll
    for (int k = 0; k < HK; ++k) { const float wk = w.wmask[k]; s0k[k] = sigmoidf_(-10.f * wk); s1k[k] = sigmoidf_(10.f * wk); }
    float* dus = strip[wave];
    float* dvs = dus + HD;
    const int base = (blockIdx.x * 4 + wave) * HSL;
    for (int sl = 0; sl < HSL; ++sl) {
        const int n = base + sl;
        const bool live = n < ntot;
        if (live) {
            const int b = inst_bag[n];
            const float* r = rec + (size_t)b * HREC;
            const float g = gloss[b];
            const float bv = bterm[n];
            float da[HK], dB = 0.f;
#pragma unroll
            for (int k = 0; k < HK; ++k) {
                const float s0 = s0k[k], s1 = s1k[k];
                const float a = araw[(size_t)n * HK + k];
                const float sp = softplusf_(a);
                const float dM = g * r[13 + k];
                const float D = r[10 + k];
                const float a1 = (s0 * sp + s1) / D;
                dB += dM * a1;
                const float dam = dM * (bv - r[k]) / D;                 // d loss / d A_mask[n,k]
                if (lane == k) dwm_out[(size_t)n * HK + k] = dam * (sp * (-10.f * s0 * (1.f - s0)) + 10.f * s1 * (1.f - s1));
                da[k] = dam * s0 * (a > 20.f ? 1.f : sigmoidf_(a));     // softplus'
                if (lane == k) da_out[(size_t)n * HK + k] = da[k];
            }
            if (lane == HK) db_out[n] = dB;
            if (lane < HD) {
                const float t = t_in[(size_t)n * HD + lane];
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < HK; ++k) s += da[k] * w2j[k];
                const float du = s * (1.f - t * t);
                du_out[(size_t)n * HD + lane] = du;
                const float v = v_in[(size_t)n * HD + lane];
                const float dv = dB * wcj * lrelu_grad(v, slope);
                dv_out[(size_t)n * HD + lane] = dv;
                dus[lane] = du; dvs[lane] = dv;
            }
        }
        __syncthreads();
        if (live) {
            float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
#pragma unroll
            for (int jj = 0; jj < HD; jj += 4) {
                const f32x4_t uq = *reinterpret_cast<const f32x4_t*>(dus + jj);
                const f32x4_t vq = *reinterpret_cast<const f32x4_t*>(dvs + jj);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a0 += uq[e] * c10[jj + e]; b0 += vq[e] * cl0[jj + e];
                    a1 += uq[e] * c11[jj + e]; b1 += vq[e] * cl1[jj + e];
                }
            }
            {
                dhz_out[(size_t)n * HL + lane] = a0;
                const float h = H[(size_t)n * HL + lane];
                float mm = b0 * lrelu_grad(h, slope);
                if (keep) mm = keep[(size_t)n * HL + lane] ? mm * keep_scale : 0.f;
                dH[(size_t)n * HL + lane] = mm;
            }
            if (lane < HL - 64) {
                dhz_out[(size_t)n * HL + i1] = a1;
                const float h = H[(size_t)n * HL + i1];
                float mm = b1 * lrelu_grad(h, slope);
                if (keep) mm = keep[(size_t)n * HL + i1] ? mm * keep_scale : 0.f;
                dH[(size_t)n * HL + i1] = mm;
            }
        }
        __syncthreads();
    }
}

// K5: parameter gradients, one thread per parameter element, summing over every instance of the
// batch in index order.  Layout of `grads`: the 11 head tensors back to back (see mil_hip.h).
#define G_BNW 0
#define G_BNB (G_BNW + HL)
#define G_AW1 (G_BNB + HL)
#define G_AB1 (G_AW1 + HD * HL)
#define G_AW2 (G_AB1 + HD)
#define G_AB2 (G_AW2 + HK * HD)
#define G_BW1 (G_AB2 + HK)
#define G_BB1 (G_BW1 + HD * HL)
#define G_BWC (G_BB1 + HD)
#define G_BBC (G_BWC + HD)
#define G_WM (G_BBC + 1)
#define G_TOTAL (G_WM + HK)

// Two-stage form of the parameter gradients (a one-wave-per-element kernel reads every operand column with a
// 160/320-byte stride from every wave: ~35 M cache-line requests for 2048 instances, 79 us).  Stage 1: one workgroup
// per slice of 64 instances stages each instance's operand row ONCE in LDS (coalesced row loads; derived operands
// z, x-hat, dropout(lrelu(H)), lrelu(v) computed while staging) and every thread accumulates its parameter elements
// over the slice in index order — every element is sum_n row[n][a] * row[n][b] for a column pair (a, b), a constant-one
// column serving the bias sums.  Stage 2 adds the slices in index order.  Deterministic; same values up to fp32
// summation order.
#define WGS 64
#define R_Z 0
#define R_XH (R_Z + HL)
#define R_DH (R_XH + HL)
#define R_M (R_DH + HL)
#define R_DU (R_M + HL)
#define R_DV (R_DU + HD)
#define R_T (R_DV + HD)
#define R_LV (R_T + HD)
#define R_DA (R_LV + HD)
#define R_DWM (R_DA + HK)
#define R_DB (R_DWM + HK)
#define R_ONE (R_DB + 1)
#define R_STRIDE (R_ONE + 1 + ((R_ONE + 1) & 1))      // even

__device__ __forceinline__ void head_elem_cols(int e, int& ca, int& cb) {
    if (e < G_BNB) { ca = R_DH + e; cb = R_XH + e; }
    else if (e < G_AW1) { ca = R_DH + (e - G_BNB); cb = R_ONE; }
    else if (e < G_AB1) { const int q = e - G_AW1; ca = R_DU + q / HL; cb = R_Z + q % HL; }
    else if (e < G_AW2) { ca = R_DU + (e - G_AB1); cb = R_ONE; }
    else if (e < G_AB2) { const int q = e - G_AW2; ca = R_DA + q / HD; cb = R_T + q % HD; }
    else if (e < G_BW1) { ca = R_DA + (e - G_AB2); cb = R_ONE; }
    else if (e < G_BB1) { const int q = e - G_BW1; ca = R_DV + q / HL; cb = R_M + q % HL; }
    else if (e < G_BWC) { ca = R_DV + (e - G_BB1); cb = R_ONE; }
    else if (e < G_BBC) { ca = R_DB; cb = R_LV + (e - G_BWC); }
    else if (e < G_WM) { ca = R_DB; cb = R_ONE; }
    else { ca = R_DWM + (e - G_WM); cb = R_ONE; }
}

__global__ __launch_bounds__(1024) void head_wgrad_partial_kernel(const float* __restrict__ H, const int* __restrict__ inst_bag,
                                                                 const float* __restrict__ stats, const uint8_t* __restrict__ keep,
                                                                 HeadWeights w, const float* __restrict__ t_in,
                                                                 const float* __restrict__ v_in, const float* __restrict__ du,
                                                                 const float* __restrict__ dv, const float* __restrict__ da,
                                                                 const float* __restrict__ dwm, const float* __restrict__ db,
                                                                 const float* __restrict__ dhz, float* __restrict__ partial,
                                                                 int ntot, float slope, float keep_scale) {
    extern __shared__ float rows[];                       // [WGS][R_STRIDE]
    MIL_POISON(rows);
    const int tid = threadIdx.x, n0 = blockIdx.x * WGS;
    for (int idx = tid; idx < WGS * HL; idx += 1024) {      // the 80-wide operands
        const int nl = idx / HL, i = idx - nl * HL, n = n0 + nl;
        float z = 0.f, xh = 0.f, dh = 0.f, m = 0.f;
        if (n < ntot) {
            const float* st = stats + (size_t)inst_bag[n] * 2 * HL;
            const float h = H[(size_t)n * HL + i];
            xh = (h - st[i]) * st[HL + i];
            z = w.bn_w[i] * xh + w.bn_b[i];
            dh = dhz[(size_t)n * HL + i];
            m = lrelu(h, slope);
            if (keep) m = keep[(size_t)n * HL + i] ? m * keep_scale : 0.f;
        }
        float* r = rows + nl * R_STRIDE;
        r[R_Z + i] = z; r[R_XH + i] = xh; r[R_DH + i] = dh; r[R_M + i] = m;
    }
    for (int idx = tid; idx < WGS * HD; idx += 1024) {      // the 40-wide operands
        const int nl = idx / HD, j = idx - nl * HD, n = n0 + nl;
        const bool ok = n < ntot;
        float* r = rows + nl * R_STRIDE;
        r[R_DU + j] = ok ? du[(size_t)n * HD + j] : 0.f;
        r[R_DV + j] = ok ? dv[(size_t)n * HD + j] : 0.f;
        r[R_T + j] = ok ? t_in[(size_t)n * HD + j] : 0.f;
        r[R_LV + j] = ok ? lrelu(v_in[(size_t)n * HD + j], slope) : 0.f;
    }
    for (int idx = tid; idx < WGS * 8; idx += 1024) {       // the narrow operands and the constant-one column
        const int nl = idx >> 3, c = idx & 7, n = n0 + nl;
        const bool ok = n < ntot;
        float* r = rows + nl * R_STRIDE;
        if (c < HK) r[R_DA + c] = ok ? da[(size_t)n * HK + c] : 0.f;
        else if (c < 2 * HK) r[R_DWM + c - HK] = ok ? dwm[(size_t)n * HK + c - HK] : 0.f;
        else if (c == 6) r[R_DB] = ok ? db[n] : 0.f;
        else r[R_ONE] = ok ? 1.f : 0.f;
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * G_TOTAL;
    for (int e = tid; e < G_TOTAL; e += 1024) {
        int ca, cb;
        head_elem_cols(e, ca, cb);
        float s = 0.f;
#pragma unroll 8
        for (int nl = 0; nl < WGS; ++nl) s += rows[nl * R_STRIDE + ca] * rows[nl * R_STRIDE + cb];
        out[e] = s;
    }
}

__global__ __launch_bounds__(256) void head_wgrad_reduce_kernel(const float* __restrict__ partial, int nslices, float* __restrict__ grads) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= G_TOTAL) return;
    float s = 0.f;
    for (int k = 0; k < nslices; ++k) s += partial[(size_t)k * G_TOTAL + e];
    grads[e] = s;
}

// l2 = 0.5*(||Wl||_F + ||Wc||_F) contributes gl2 * 0.5 * W/||W|| to the two buffer weights.
__global__ __launch_bounds__(256) void head_l2_grad_kernel(HeadWeights w, const float* __restrict__ gl2, float* __restrict__ grads) {
    __shared__ float red[16];
    const int tid = threadIdx.x;
    float q1 = 0.f, q2 = 0.f;
    for (int i = tid; i < HD * HL; i += 256) { const float v = w.b_w1[i]; q1 += v * v; }
    if (tid < HD) { const float v = w.b_wc[tid]; q2 = v * v; }
    q1 = block_sum(q1, red); q2 = block_sum(q2, red);
    const float g = gl2[0] * 0.5f;
    const float n1 = sqrtf(q1), n2 = sqrtf(q2);
    for (int i = tid; i < HD * HL; i += 256) grads[G_BW1 + i] += (n1 > 0.f) ? g * w.b_w1[i] / n1 : 0.f;
    if (tid < HD) grads[G_BWC + tid] += (n2 > 0.f) ? g * w.b_wc[tid] / n2 : 0.f;
}

// K6: finish dH with the batch-norm branch:  dH += rstd/N * (N*dx - sum(dx) - xhat*sum(dx*xhat)), dx = dHz*gamma
// (HP row partials per column, as K1)
__global__ __launch_bounds__(1024) void head_bn_bwd_kernel(const float* __restrict__ H, const int* __restrict__ off,
                                                           const float* __restrict__ stats, HeadWeights w,
                                                           const float* __restrict__ dhz, float* __restrict__ dH) {
    __shared__ float p1[HP][HL], p2[HP][HL];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n0 = off[b], n1 = off[b + 1], N = n1 - n0;
    const int i = tid % HL, p = tid / HL;
    const float* st = stats + (size_t)b * 2 * HL;
    float s1 = 0.f, s2 = 0.f;
    if (p < HP) {
        const float gm = w.bn_w[i], mu = st[i], rs = st[HL + i];
        for (int n = n0 + p; n < n1; n += HP) {
            const float dx = dhz[(size_t)n * HL + i] * gm;
            s1 += dx; s2 += dx * ((H[(size_t)n * HL + i] - mu) * rs);
        }
        p1[p][i] = s1; p2[p][i] = s2;
    }
    __syncthreads();
    if (p < HP) {
        const float gm = w.bn_w[i], mu = st[i], rs = st[HL + i];
        float t1 = p1[0][i], t2 = p2[0][i];
#pragma unroll
        for (int k = 1; k < HP; ++k) { t1 += p1[k][i]; t2 += p2[k][i]; }
        const float invn = 1.f / (float)N;
        for (int n = n0 + p; n < n1; n += HP) {
            const float xh = (H[(size_t)n * HL + i] - mu) * rs;
            const float dx = dhz[(size_t)n * HL + i] * gm;
            dH[(size_t)n * HL + i] += rs * (dx - invn * t1 - invn * xh * t2);
        }
    }
}

// ---------------------------------------------------------------------------------------------
static HeadWeights make_weights(const float* const* p) {
    HeadWeights w;
    w.bn_w = p[0]; w.bn_b = p[1]; w.a_w1 = p[2]; w.a_b1 = p[3]; w.a_w2 = p[4]; w.a_b2 = p[5];
    w.b_w1 = p[6]; w.b_b1 = p[7]; w.b_wc = p[8]; w.b_bc = p[9]; w.wmask = p[10];
    return w;
}

extern "C" int mil_head_workspace_floats(size_t* floats, int ntot, int nbags) {
    if (!floats || ntot < 0 || nbags < 0) return MIL_ERR_ARG;
    // stats[nbags*160] t[n*40] v[n*40] araw[n*3] b[n] | du[n*40] dv[n*40] da[n*3] dwm[n*3] db[n] dhz[n*80]
    *floats = (size_t)nbags * 2 * HL + (size_t)ntot * (HD + HD + HK + 1 + HD + HD + HK + HK + 1 + HL)
              + (size_t)((ntot + WGS - 1) / WGS) * G_TOTAL;          // + per-slice partial parameter gradients
    return MIL_OK;
}

struct HeadWs { float *stats, *t, *v, *araw, *b, *du, *dv, *da, *dwm, *db, *dhz, *wpart; };
static HeadWs carve(float* ws, int ntot, int nbags) {
    HeadWs h; float* p = ws;
    h.stats = p; p += (size_t)nbags * 2 * HL;
    h.t = p; p += (size_t)ntot * HD; h.v = p; p += (size_t)ntot * HD; h.araw = p; p += (size_t)ntot * HK; h.b = p; p += ntot;
    h.du = p; p += (size_t)ntot * HD; h.dv = p; p += (size_t)ntot * HD; h.da = p; p += (size_t)ntot * HK;
    h.dwm = p; p += (size_t)ntot * HK; h.db = p; p += ntot; h.dhz = p; p += (size_t)ntot * HL; h.wpart = p;
    return h;
}

// weights: array of 11 device pointers in the order
//   context.bn.weight, context.bn.bias, attention.lin1.weight, attention.lin1.bias, attention.lin2.weight,
//   attention.lin2.bias, buffer.lin1.weight, buffer.lin1.bias, buffer.classifier.weight,
//   buffer.classifier.bias, weight_mask
extern "C" int mil_head_fwd(const float* H, const int* bag_offsets, const int* inst_bag, const int64_t* labels,
                            const uint8_t* keep_mask, const float* class_weights, const float* const* weights,
                            float* workspace, float* a1, float* wrois, float* bterm, float* kld, float* rec, int ntot,
                            int nbags, float slope, float drop_p, float smoothing, float bn_eps, void* stream) {
    if (!H || !bag_offsets || !inst_bag || !labels || !weights || !workspace || !a1 || !wrois || !bterm || !kld || !rec)
        return MIL_ERR_ARG;
    if (ntot <= 0 || nbags <= 0) return MIL_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const HeadWeights w = make_weights(weights);
    HeadWs ws = carve(workspace, ntot, nbags);
    const float ks = 1.f / (1.f - drop_p);
    hipLaunchKernelGGL(head_colstats_kernel, dim3(nbags), dim3(1024), 0, st, H, bag_offsets, ws.stats, kld, bn_eps);
    MIL_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_inst_fwd_kernel, dim3((ntot + 4 * HSL - 1) / (4 * HSL)), dim3(256), 0, st, H, inst_bag, ws.stats, keep_mask, w,
                       ws.t, ws.v, ws.araw, bterm, ntot, slope, ks);
    MIL_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_bag_fwd_kernel, dim3(nbags), dim3(256), 0, st, ws.araw, bterm, bag_offsets, labels, class_weights, w,
                       smoothing, a1, wrois, rec);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// grads: G_TOTAL (=6807) floats, the 11 head tensors' gradients back to back in the `weights` order.
extern "C" int mil_head_bwd(const float* H, const int* bag_offsets, const int* inst_bag, const uint8_t* keep_mask,
                            const float* const* weights, float* workspace, const float* bterm, const float* rec,
                            const float* grad_loss, const float* grad_l2, float* dH, float* grads, int ntot, int nbags,
                            float slope, float drop_p, void* stream) {
    if (!H || !bag_offsets || !inst_bag || !weights || !workspace || !bterm || !rec || !grad_loss || !dH || !grads)
        return MIL_ERR_ARG;
    if (ntot <= 0 || nbags <= 0) return MIL_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const HeadWeights w = make_weights(weights);
    HeadWs ws = carve(workspace, ntot, nbags);
    const float ks = 1.f / (1.f - drop_p);
    hipLaunchKernelGGL(head_inst_bwd_kernel, dim3((ntot + 4 * HSL - 1) / (4 * HSL)), dim3(256), 0, st, H, inst_bag, keep_mask, w, ws.t, ws.v,
                       ws.araw, bterm, rec, grad_loss, ws.du, ws.dv, ws.da, ws.dwm, ws.db, ws.dhz, dH, ntot, slope, ks);
    MIL_CHECK_LAUNCH();
    {
        const int nslices = (ntot + WGS - 1) / WGS;
        const int lds = WGS * R_STRIDE * 4;
        static std::atomic<unsigned long long> attr_set{0};      // per device
        if (mil_device_needs(attr_set)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(head_wgrad_partial_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
                return MIL_ERR_LAUNCH;
            mil_device_done(attr_set);
        }
        hipLaunchKernelGGL(head_wgrad_partial_kernel, dim3(nslices), dim3(1024), lds, st, H, inst_bag, ws.stats, keep_mask, w, ws.t,
                           ws.v, ws.du, ws.dv, ws.da, ws.dwm, ws.db, ws.dhz, ws.wpart, ntot, slope, ks);
        MIL_CHECK_LAUNCH();
        hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3((G_TOTAL + 255) / 256), dim3(256), 0, st, ws.wpart, nslices, grads);
        MIL_CHECK_LAUNCH();
    }
    if (grad_l2) {
        hipLaunchKernelGGL(head_l2_grad_kernel, dim3(1), dim3(256), 0, st, w, grad_l2, grads);
        MIL_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(head_bn_bwd_kernel, dim3(nbags), dim3(1024), 0, st, H, bag_offsets, ws.stats, w, ws.dhz, dH);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

extern "C" int mil_head_grad_floats(void) { return G_TOTAL; }
extern "C" int mil_head_rec_floats(void) { return HREC; }
