// Shared pieces of the persistent, prefetch-pipelined bf16 kernels (conv_igemm_pf_kernel,
// conv_bwd_fused_kernel): buffer-resource addressing, tile walking on the scalar unit, and the
// tile-invariant halo tables.
//
// Global memory goes through raw buffer loads/stores (SGPR descriptor + 32-bit per-lane byte offset):
// no 64-bit per-lane address arithmetic, and predication costs ONE v_cndmask — an invalid lane gets the
// offset MIL_OOB (>= num_records), for which the hardware returns zeros on a load and drops a store, so no
// EXEC-mask save/restore sequences surround the memory instructions.  Tensors handed to these kernels must
// be < 2 GiB (the launchers split larger launches by image).
#pragma once
#include "geom.cuh"

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define MIL_OOB 0x80000000u

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mil_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// Stage `nbytes` (a multiple of 16) of packed filter from global memory into LDS with all `nthr` threads of the workgroup,
// EIGHT 16-byte loads in flight per thread.  The plain copy loop compiles to `global_load; s_waitcnt vmcnt(0); ds_write`
// per iteration: one L2 round trip per 16 bytes and thread — 14 serial round trips (≈10 us) for the 115 KB filter of an
// 80-channel layer on 512 threads, 25 for the 100 KB of the 64->80 stage entry on 256: a third of those launches.
__device__ __forceinline__ void mil_stage_filter(char* lds, const void* src, int nbytes, int tid, int nthr) {
    const __amdgpu_buffer_rsrc_t rs = mil_rsrc(src, (unsigned)nbytes);
    const int step = nthr * 16;
    for (int i0 = tid * 16; i0 < nbytes; i0 += 8 * step) {
        u32x4_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(i0 + k * step), 0, 0);      // beyond nbytes: zeros, not stored
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + k * step < nbytes) *reinterpret_cast<u32x4_t*>(lds + i0 + k * step) = v[k];
    }
}

// Walks tile ids t0, t0+G, t0+2G, ... as (image group, tile row, tile col) with carries instead of
// divisions; every member is wave-uniform (lives in SGPRs).
struct TileWalker {
    int tx, ty, grp;
    int dtx, dty, dgrp;
    int tiles_x, tiles_y;
    __device__ __forceinline__ void init(const ConvGeom& g, int t0, int stride) {
        tiles_x = g.tiles_x; tiles_y = g.tiles_y;
        tx = t0 % tiles_x; int q = t0 / tiles_x; ty = q % tiles_y; grp = q / tiles_y;
        dtx = stride % tiles_x; q = stride / tiles_x; dty = q % tiles_y; dgrp = q / tiles_y;
    }
    __device__ __forceinline__ void advance() {
        tx += dtx; ty += dty; grp += dgrp;
        if (tx >= tiles_x) { tx -= tiles_x; ty += 1; }
        if (ty >= tiles_y) { ty -= tiles_y; grp += 1; }
    }
    __device__ __forceinline__ TileOrigin origin(const ConvGeom& g) const {
        TileOrigin o; o.img0 = grp << g.ti_log2; o.oy0 = ty << g.th_log2; o.ox0 = tx << g.tw_log2; return o;
    }
};

// Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2.  Persistent tile walks start from this
// id instead of blockIdx.x, so that the G/8 workgroups resident on ONE XCD work on consecutive tiles (= neighbouring
// tiles of the same images) at any time and the halo rows/columns neighbouring tiles share are L2 hits there
// instead of a second HBM/MALL fetch from another XCD.
__device__ __forceinline__ int mil_xcd_block_id() {
    const int b = blockIdx.x, G = gridDim.x;
    return (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
}

// Halo pieces (16 B) owned by a thread: flat piece id = tid + 256*i.
//   pos = (ti<<20)|(hy<<10)|hx, or -1 when the slot is unused;  lds = byte offset in the LDS halo tile;
//   rel = byte offset of the piece relative to the halo origin pixel of image img0 (plain loader only)
template <int NP>
struct HaloTables { int pos[NP], lds[NP], rel[NP]; };

// T: element type of the tensor (BF16 default).  For T = F32S (fp32 in HBM, [hi | lo] bf16 planes in LDS) a 16-byte piece is
// four fp32 channels and lands as 8 bytes in the hi plane + 8 bytes in the lo plane (t.lds = the hi address).
template <int CP, int NP, int NTHR = 256, typename T = BF16>
__device__ __forceinline__ void mil_build_halo_tables(HaloTables<NP>& t, const ConvGeom& g, int tid) {
    constexpr int N16 = CP * T::ESZ / 16;
    constexpr int PIXB = mil_pix_pitch(CP, T::ESZ);
    constexpr int JB = T::SPLIT ? 8 : 16;
    const int ppr = g.hw * N16;
    const int total = (g.hh << g.ti_log2) * ppr;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = tid + NTHR * i;
        t.pos[i] = -1; t.lds[i] = 0; t.rel[i] = 0;
        if (idx < total) {
            const int row = idx / ppr, piece = idx - row * ppr;
            const int ti = row / g.hh, hy = row - ti * g.hh;
            const int hx = piece / N16, j = piece - hx * N16;
            t.pos[i] = (ti << 20) | (hy << 10) | hx;
            t.lds[i] = (row * g.hw + hx) * PIXB + j * JB;
            t.rel[i] = g.zins ? j * 16 : ((ti * g.H + hy) * g.W + hx) * (CP * T::ESZ) + j * 16;
        }
    }
}

// Issue the loads of one halo tile into registers (zero for padding / outside the image).
template <int CP, int NP, typename T = BF16>
__device__ __forceinline__ void mil_fetch_halo(u32x4_t (&rx)[NP], __amdgpu_buffer_rsrc_t src, const HaloTables<NP>& t,
                                               const ConvGeom& g, const TileOrigin& o) {
    const int s = g.zins ? 1 : g.stride;
    const int iy0 = o.oy0 * s - g.pad, ix0 = o.ox0 * s - g.pad;
    const int ilim = g.n_img - o.img0;
    if (!g.zins) {
        const int base = ((o.img0 * g.H + iy0) * g.W + ix0) * (CP * T::ESZ);      // may be negative; valid lanes are not
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = t.pos[i];
            const int iy = iy0 + ((p >> 10) & 1023), ix = ix0 + (p & 1023);
            const bool ok = p >= 0 && (p >> 20) < ilim && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(src, ok ? (unsigned)(base + t.rel[i]) : MIL_OOB, 0, 0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = t.pos[i];
            const int iy = iy0 + ((p >> 10) & 1023), ix = ix0 + (p & 1023);
            const bool ok = p >= 0 && (p >> 20) < ilim && iy >= 0 && ix >= 0 && !((iy | ix) & 1) && (iy >> 1) < g.H && (ix >> 1) < g.W;
            const int off = (((o.img0 + (p >> 20)) * g.H + (iy >> 1)) * g.W + (ix >> 1)) * (CP * T::ESZ) + t.rel[i];
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(src, ok ? (unsigned)off : MIL_OOB, 0, 0);
        }
    }
}

// One fetched 16-byte piece -> LDS.  F32S: four fp32 channels -> 8 bytes of the record's hi plane + 8 bytes, LO behind,
// of its lo plane.
template <typename T, int LO>
__device__ __forceinline__ void mil_commit_piece(char* p, const u32x4_t& r) {
    if constexpr (T::SPLIT) {
        const f32x4_t v = __builtin_bit_cast(f32x4_t, r);
        bf16x4_t hi, lo;
        mil_split4(v, hi, lo);
        *reinterpret_cast<bf16x4_t*>(p) = hi;
        *reinterpret_cast<bf16x4_t*>(p + LO) = lo;
    } else {
        *reinterpret_cast<u32x4_t*>(p) = r;
    }
}

template <int NP, typename T = BF16, int CP = 0>
__device__ __forceinline__ void mil_commit_halo(const u32x4_t (&rx)[NP], char* lds, const HaloTables<NP>& t) {
#pragma unroll
    for (int i = 0; i < NP; ++i)
        if (t.pos[i] >= 0) mil_commit_piece<T, CP * 2>(lds + t.lds[i], rx[i]);
}

// Branch-free commit: tables built with mil_halo_tables_use_dump() send the unused slots to a 16-byte dump area, so
// every slot is written unconditionally (a divergent branch per LDS store costs more than the store).
template <int NP>
__device__ __forceinline__ void mil_halo_tables_use_dump(HaloTables<NP>& t, int dump_off) {
#pragma unroll
    for (int i = 0; i < NP; ++i) if (t.pos[i] < 0) t.lds[i] = dump_off;
}
template <int NP, typename T = BF16, int CP = 0>
__device__ __forceinline__ void mil_commit_halo_all(const u32x4_t (&rx)[NP], char* lds, const HaloTables<NP>& t) {
#pragma unroll
    for (int i = 0; i < NP; ++i) mil_commit_piece<T, CP * 2>(lds + t.lds[i], rx[i]);
}

// Output-space tile (no halo) pieces owned by a thread: flat piece id = tid + 256*i -> (tile pixel, 16-B piece).
template <int NP>
struct OtileTables { int pos[NP], lds[NP], rel[NP]; };

template <int CP, int NP, int NTHR = 256, typename T = BF16>
__device__ __forceinline__ void mil_build_otile_tables(OtileTables<NP>& t, const ConvGeom& g, int tid, int tile_px) {
    constexpr int N16 = CP * T::ESZ / 16;
    constexpr int PIXZ = mil_pix_pitch(CP, T::ESZ);
    constexpr int JB = T::SPLIT ? 8 : 16;
    const int tw_mask = (1 << g.tw_log2) - 1, th_mask = (1 << g.th_log2) - 1;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = tid + NTHR * i;
        t.pos[i] = -1; t.lds[i] = 0; t.rel[i] = 0;
        if (idx < tile_px * N16) {
            const int tp = idx / N16, j = idx - tp * N16;
            const int tx = tp & tw_mask, ty = (tp >> g.tw_log2) & th_mask, ti = tp >> (g.tw_log2 + g.th_log2);
            t.pos[i] = (ti << 20) | (ty << 10) | tx;
            t.lds[i] = tp * PIXZ + j * JB;
            t.rel[i] = ((ti * g.Ho + ty) * g.Wo + tx) * (CP * T::ESZ) + j * 16;
        }
    }
}

template <int CP, int NP, typename T = BF16>
__device__ __forceinline__ void mil_fetch_otile(u32x4_t (&rz)[NP], __amdgpu_buffer_rsrc_t src, const OtileTables<NP>& t,
                                                const ConvGeom& g, const TileOrigin& o) {
    const int base = ((o.img0 * g.Ho + o.oy0) * g.Wo + o.ox0) * (CP * T::ESZ);
    const int ylim = g.Ho - o.oy0, xlim = g.Wo - o.ox0, ilim = g.n_img - o.img0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = t.pos[i];
        const bool ok = p >= 0 && (p >> 20) < ilim && ((p >> 10) & 1023) < ylim && (p & 1023) < xlim;
        rz[i] = __builtin_amdgcn_raw_buffer_load_b128(src, ok ? (unsigned)(base + t.rel[i]) : MIL_OOB, 0, 0);
    }
}

template <int NP, typename T = BF16, int CP = 0>
__device__ __forceinline__ void mil_commit_otile(const u32x4_t (&rz)[NP], char* lds, const OtileTables<NP>& t) {
#pragma unroll
    for (int i = 0; i < NP; ++i)
        if (t.pos[i] >= 0) mil_commit_piece<T, CP * 2>(lds + t.lds[i], rz[i]);
}

// v_permlane16_swap: odd 16-lane rows of `a` <-> even rows of `b` (a' = [a.r0 b.r0 a.r2 b.r2], b' = [a.r1 b.r1
// a.r3 b.r3]; checked on hardware).  Inline asm on purpose: with ROCm 7.2's hipcc the builtin
// (__builtin_amdgcn_permlane16_swap) called on several elements of an ext_vector is collapsed into ONE swap whose
// result is reused for all of them (wrong values, no diagnostic).  `pad` inserts the wait states an MFMA result
// needs before a VALU instruction may read it: the compiler does not pad inside asm and may schedule MFMAs of the
// other row tiles right in front of this statement.
template <bool PAD>
__device__ __forceinline__ void mil_swap16(float& a, float& b) {
    if constexpr (PAD) asm volatile("s_nop 7\n\ts_nop 7\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

__device__ __forceinline__ bf16x8_t mil_tr_pair(const char* p0, const char* p1) {
    typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

// The (k-step, row tile) loop of an implicit-GEMM conv as ONE flattened software pipeline (bf16): the pixel fragment of
// step j + LA is read LA steps before the NT MFMAs that consume it (a ring of LA + 1 fragments), the filter fragments of
// k-step sl + 1 are read at the start of k-step sl, and scheduling fences keep that order — hipcc on its own issues each
// fragment read directly in front of its MFMAs behind an lgkmcnt(0), which makes such loops LDS-latency loops.
// acc[m][nt] accumulates D[channel][pixel] = filter fragment (A) x pixel fragment (B); `xaddr(sl, m)` is the LDS address
// of row tile m's fragment for k-step sl.  Everything is unrolled, so ring / buffer indices are compile-time constants.
template <int NT, int MT, int KSTEPS, int LA, class XAddr>
__device__ __forceinline__ void mil_conv_ring(f32x4_t (&acc)[MT][NT], const char* ldsW, int lane, XAddr xaddr) {
    constexpr int TOT = KSTEPS * MT, R = LA + 1;
    Frag8<BF16> ring[R], wq[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wq[0][nt] = lds_frag<BF16>(ldsW + (nt * 64 + lane) * 16);
#pragma unroll
    for (int j = 0; j < LA && j < TOT; ++j) ring[j % R] = lds_frag<BF16>(xaddr(j / MT, j % MT));
#pragma unroll
    for (int j = 0; j < TOT; ++j) {
        const int sl = j / MT, m = j % MT;
        if (j + LA < TOT) ring[(j + LA) % R] = lds_frag<BF16>(xaddr((j + LA) / MT, (j + LA) % MT));
        if (m == 0 && sl + 1 < KSTEPS) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wq[(sl + 1) & 1][nt] = lds_frag<BF16>(ldsW + (((sl + 1) * NT + nt) * 64 + lane) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mma8(wq[sl & 1][nt], ring[j % R], acc[m][nt]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Workgroups of `kern` (256 threads, `lds` bytes of dynamic LDS) that one CU holds at a time, from the runtime's own
// occupancy calculation (registers AND LDS).  Persistent launches size their grid to exactly the resident set: a
// partial second round would leave CUs idle while the stragglers finish.
template <typename K>
__host__ inline int mil_resident_per_cu(K kern, int lds, int cap, int threads = 256) {
    // the answer depends only on (kernel, lds, threads): remember every combination asked for (launch-path cost; the
    // function-pointer TYPE is shared by all instantiations with one signature, so the key holds the pointer itself)
    struct Key { const void* k; int lds, threads; };
    struct Slot { Key key; int n; };
    static thread_local Slot slots[64];
    static thread_local int used = 0;
    const void* kp = reinterpret_cast<const void*>(kern);
    for (int i = 0; i < used; ++i)
        if (slots[i].key.k == kp && slots[i].key.lds == lds && slots[i].key.threads == threads)
            return slots[i].n > cap ? cap : slots[i].n;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, (size_t)lds) != hipSuccess || n < 1) n = 1;
    if (used < 64) slots[used++] = Slot{Key{kp, lds, threads}, n};
    return n > cap ? cap : n;
}

// Compute units of the current device (256 on MI355X), asked once per thread.
__host__ inline int mil_num_cus() {
    static thread_local int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
            n = 256;
        cus = n;
    }
    return cus;
}

// How many images of `bytes_per_img` bytes fit under the 2 GiB buffer limit.
// (MIL_BUFFER_LIMIT_BYTES lowers the limit: the tests use it to drive small launches through the chunked paths)
#include <cstdlib>
__host__ inline size_t mil_buffer_limit() {
    const char* e = getenv("MIL_BUFFER_LIMIT_BYTES");           // read per call (launch path, ~100 ns): a test can change it
    const size_t v = e ? (size_t)atoll(e) : 0;
    return (v >= 65536 && v < ((size_t)1 << 31)) ? v : ((size_t)1 << 31) - 4096;
}
__host__ inline int mil_imgs_under_2g(size_t bytes_per_img) {
    const size_t lim = mil_buffer_limit();
    size_t n = bytes_per_img ? lim / bytes_per_img : 1;
    return (int)(n < 1 ? 1 : (n > (1u << 30) ? (1u << 30) : n));
}
