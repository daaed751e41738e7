// Fixed-order slab reductions of the weight-gradient kernels, as device functions shared by the per-call reduction
// kernels and by the BATCHED reduction (mil_wgrad_reduce_all): a backward pass has 28 of these ~10 us launches; recorded
// as jobs while the producers run and reduced by ONE launch at the end they cost one launch and their bandwidth.
//
// Recording is the library's only piece of state and it is explicit: between mil_reduce_defer_begin(table, cap) and
// mil_reduce_defer_end() every weight-gradient producer called on THIS thread appends its reduction to the caller's host
// table instead of launching it (its slab workspace must then stay untouched until mil_wgrad_reduce_all has run).
#pragma once
#include "geom.cuh"

struct MilReduceJob {
    const float* slab;             // [nslab][slab_elems]
    float* dw;                     // [cout][cin][ks][ks] fp32 (7x7 for the stem)
    float* db;                     // [cout] or null
    unsigned long long slab_elems;
    int nslab, slab_cols, n_rows, cout, cin, ks;
    int kind;                      // 0: rows (tap*cinp + ci), cols co;   1: rows (tap'*cinp + co) with tap' flipped, cols ci
    int cinp;                      // row pitch per tap (padded channel count: cin side for kind 0, cout side for kind 1)
    int stem_mode, bias_row;       // kind 0: 4x4 space-to-depth taps -> 7x7 filter; slab row that holds the bias sums
    int bias_off, bias_stride;     // kind 1: slab element of db[0] and stride between db[co], db[co+1]
    int accumulate;
    int n_blocks;                  // 32-element blocks of this job (0 = empty slot)
    int block0;                    // first block of this job inside a batched launch
    int pad_;
};

struct MilDeferState { MilReduceJob* jobs; int cap; int n; };
MilDeferState& mil_defer_state();                      // one per thread (defined in conv_wgrad.hip)

#define MIL_RED_VEC 4               // consecutive slab elements per thread (one 16-byte load per slab)
__host__ inline int mil_reduce_blocks(const MilReduceJob& j) {
    const int total = j.kind == 0 ? (j.n_rows + 1) * j.slab_cols : j.n_rows * j.slab_cols + j.cinp;
    return (total + 32 * MIL_RED_VEC - 1) / (32 * MIL_RED_VEC);
}
// true: recorded (the caller must not launch the reduction)
__host__ inline bool mil_try_defer(MilReduceJob j) {
    MilDeferState& s = mil_defer_state();
    if (!s.jobs || s.n >= s.cap) return false;
    j.n_blocks = mil_reduce_blocks(j);
    j.block0 = s.n ? s.jobs[s.n - 1].block0 + s.jobs[s.n - 1].n_blocks : 0;
    s.jobs[s.n++] = j;
    return true;
}

// One 32 x MIL_RED_GROUPS thread block reduces 128 consecutive slab elements of job j (block index blk inside the job), four
// per thread: thread group gq sums slabs gq, gq+G, ... (mil_slab_partial4: one 16-byte load per slab — 512 contiguous bytes
// per slab and block instead of 128, which is what the DRAM pages want: 207 -> ~150 us for the 0.69 GB of a backward pass),
// group 0 adds the G partial sums in order and scatters into the reference weight layout.  Every element sees the same
// summation tree as in the scalar form: it depends only on (nslab, G), bitwise reproducible, batched or not.
__device__ __forceinline__ f32x4_t mil_slab_partial4(const float* __restrict__ slab, size_t slab_elems, size_t src, int gq, int nslab) {
    f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int i = gq;
    for (; i + 7 * MIL_RED_GROUPS < nslab; i += 8 * MIL_RED_GROUPS) {
        f32x4_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4_t*>(slab + (size_t)(i + k * MIL_RED_GROUPS) * slab_elems + src);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; i < nslab; i += MIL_RED_GROUPS) s += *reinterpret_cast<const f32x4_t*>(slab + (size_t)i * slab_elems + src);
    return s;
}

// element e of job j -> its slab element
__device__ __forceinline__ size_t mil_reduce_src(const MilReduceJob& j, int e, int n_w) {
    if (j.kind == 0) {
        const int row = e / j.slab_cols, col = e - row * j.slab_cols;
        return (size_t)(row == j.n_rows ? j.bias_row : row) * j.slab_cols + col;
    }
    return e < n_w ? (size_t)e : (size_t)j.bias_off + (size_t)(e - n_w) * j.bias_stride;
}

// sum v of element e -> dW / db in the reference layout
__device__ __forceinline__ void mil_reduce_scatter(const MilReduceJob& j, int e, int n_w, float v) {
    const int acc = j.accumulate;
    if (j.kind == 0) {
        const int row = e / j.slab_cols, co = e - row * j.slab_cols;
        if (co >= j.cout) return;
        if (row == j.n_rows) { if (j.db) j.db[co] = acc ? j.db[co] + v : v; return; }
        const int tap = row / j.cinp, ch = row - tap * j.cinp;
        if (j.stem_mode) {
            if (ch >= 12) return;
            const int ci = ch >> 2, dy = (ch >> 1) & 1, dx = ch & 1;
            const int ky = 2 * (tap >> 2) + dy - 1, kx = 2 * (tap & 3) + dx - 1;
            if (ky < 0 || ky >= 7 || kx < 0 || kx >= 7) return;
            float* q = j.dw + (((size_t)co * 3 + ci) * 7 + ky) * 7 + kx;
            *q = acc ? *q + v : v;
        } else {
            if (ch >= j.cin) return;
            float* q = j.dw + ((size_t)co * j.cin + ch) * (j.ks * j.ks) + tap;
            *q = acc ? *q + v : v;
        }
    } else {
        if (e >= n_w) {
            const int co = e - n_w;
            if (co < j.cout && j.db) j.db[co] = acc ? j.db[co] + v : v;
            return;
        }
        const int row = e / j.slab_cols, ci = e - row * j.slab_cols;
        const int tapf = row / j.cinp, co = row - tapf * j.cinp;
        if (ci >= j.cin || co >= j.cout) return;
        float* q = j.dw + ((size_t)co * j.cin + ci) * (j.ks * j.ks) + (j.ks * j.ks - 1 - tapf);
        *q = acc ? *q + v : v;
    }
}

__device__ __forceinline__ void mil_reduce_job_block(const MilReduceJob& j, int blk, f32x4_t (*part)[32]) {
    const int c = threadIdx.x & 31, gq = threadIdx.x >> 5;
    const int e0 = (blk * 32 + c) * MIL_RED_VEC;
    const int n_w = j.n_rows * j.slab_cols;
    const int total = j.kind == 0 ? (j.n_rows + 1) * j.slab_cols : n_w + j.cinp;
    // kind 0: the bias row follows the weight rows as one more contiguous row; kind 1: the bias sums are strided
    const int n_vec = j.kind == 0 ? total : n_w;               // elements below this are contiguous in fours (slab_cols % 16 == 0)
    f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (e0 + MIL_RED_VEC <= n_vec) {
        s = mil_slab_partial4(j.slab, (size_t)j.slab_elems, mil_reduce_src(j, e0, n_w), gq, j.nslab);
    } else {
#pragma unroll
        for (int u = 0; u < MIL_RED_VEC; ++u)
            if (e0 + u < total) s[u] = mil_slab_partial(j.slab, (size_t)j.slab_elems, mil_reduce_src(j, e0 + u, n_w), gq, j.nslab);
    }
    part[gq][c] = s;
    __syncthreads();
    if (gq != 0 || e0 >= total) return;
    f32x4_t v = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MIL_RED_GROUPS; ++k) v += part[k][c];
#pragma unroll
    for (int u = 0; u < MIL_RED_VEC; ++u)
        if (e0 + u < total) mil_reduce_scatter(j, e0 + u, n_w, v[u]);
}

// one reduction, launched on its own (kernel defined in conv_wgrad.hip)
__global__ void wgrad_reduce_job_kernel(MilReduceJob j);
__host__ inline void mil_launch_reduce(MilReduceJob j, hipStream_t stream) {
    j.n_blocks = mil_reduce_blocks(j); j.block0 = 0;
    hipLaunchKernelGGL(wgrad_reduce_job_kernel, dim3(j.n_blocks), dim3(32 * MIL_RED_GROUPS), 0, stream, j);
}
// record when a deferral is open on this thread, else launch now
__host__ inline void mil_reduce_or_defer(const MilReduceJob& j, hipStream_t stream, bool may_defer = true) {
    if (may_defer && mil_try_defer(j)) return;
    mil_launch_reduce(j, stream);
}
