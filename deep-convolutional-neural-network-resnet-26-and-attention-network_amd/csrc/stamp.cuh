// Diagnostic builds only (`make VARIANT=stamp EXTRA=-DMIL_STAMP`): per-wave s_memtime sums of the phases of a persistent
// kernel's tile loop.  The stamps drain the LDS queue (lgkmcnt(0)) and pin the schedule: read the SHARES, not the run time.
// In the shipped library every macro below is empty and no kernel carries a stamp.
//   kernel:   MIL_STAMP_DECL(N) before the tile loop; MIL_STAMP_BEGIN() at the top of a tile; MIL_STAMP_MARK(i) after phase i;
//             MIL_STAMP_STORE(ptr, nwaves) behind the loop (ptr = [grid][nwaves][N + 2] u64: sums, tiles, 100 MHz ticks)
//   launcher: MilStampBuf (host side) allocates the buffer, and after the launch synchronises, averages over workgroups and
//             prints one line per kernel launch to stderr.
#pragma once
#ifdef MIL_STAMP
#include <cstdio>
#include <vector>
#define MIL_STAMP_DECL(N) unsigned long long st_prev_ = 0, st_sum_[N] = {}; int st_tiles_ = 0; constexpr int ST_N_ = N; \
    const unsigned long long st_rt0_ = __builtin_amdgcn_s_memrealtime();
#define MIL_STAMP_T_(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define MIL_STAMP_BEGIN() { MIL_STAMP_T_(st_prev_) ++st_tiles_; }
#define MIL_STAMP_MARK(i) { unsigned long long t_; MIL_STAMP_T_(t_) st_sum_[i] += t_ - st_prev_; st_prev_ = t_; }
#define MIL_STAMP_STORE(ptr, nwaves) if ((ptr) && (threadIdx.x & 63) == 0) { \
    unsigned long long* d_ = (ptr) + ((size_t)blockIdx.x * (nwaves) + (threadIdx.x >> 6)) * (ST_N_ + 2); \
    for (int i_ = 0; i_ < ST_N_; ++i_) d_[i_] = st_sum_[i_]; \
    d_[ST_N_] = (unsigned long long)st_tiles_; d_[ST_N_ + 1] = __builtin_amdgcn_s_memrealtime() - st_rt0_; }
struct MilStampBuf {
    unsigned long long* dev = nullptr;
    size_t cap = 0;
    unsigned long long* get(size_t n) {
        if (n > cap) { if (dev) (void)hipFree(dev); if (hipMalloc(&dev, n * 8) != hipSuccess) { dev = nullptr; cap = 0; return nullptr; } cap = n; }
        (void)hipMemset(dev, 0, n * 8);
        return dev;
    }
    // phases[i] = name of phase i
    void report(const char* kernel, int grid, int nwaves, int nph, const char* const* phases, hipStream_t st) {
        if (!dev) return;
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h((size_t)grid * nwaves * (nph + 2));
        (void)hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> sum(nph, 0.0);
        double tiles = 0, ticks = 0, cyc = 0;
        for (size_t w = 0; w < (size_t)grid * nwaves; ++w) {
            const unsigned long long* d = h.data() + w * (nph + 2);
            for (int i = 0; i < nph; ++i) { sum[i] += (double)d[i]; cyc += (double)d[i]; }
            tiles += (double)d[nph]; ticks += (double)d[nph + 1];
        }
        if (tiles <= 0) return;
        fprintf(stderr, "[stamp] %s: %.0f cycles per tile and wave, in-kernel clock %.2f GHz |", kernel, cyc / tiles, ticks > 0 ? cyc / ticks * 0.1 : 0.0);
        for (int i = 0; i < nph; ++i) fprintf(stderr, " %s %.0f (%.0f%%)", phases[i], sum[i] / tiles, 100.0 * sum[i] / cyc);
        fprintf(stderr, "\n");
    }
};
#else
#define MIL_STAMP_DECL(N)
#define MIL_STAMP_BEGIN()
#define MIL_STAMP_MARK(i)
#define MIL_STAMP_STORE(ptr, nwaves)
#endif
