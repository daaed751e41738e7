// fp32 master weights [Cout][Cin][k][k] -> MFMA operand-fragment order [kstep][ntile][lane][8], one job per filter.
// Shared by the single-filter entry point (pointwise.hip) and the whole-model table launch (optim.hip).
//   k-group q = 4*kstep + (lane>>4) = tap*CG + cg; element e is K-side channel cg*8+e; column = lane&15.
//   mode 0 (forward):  B[(tap,ci)][co] = W[co][ci][ky][kx]
//   mode 1 (dgrad):    B[(tap,co)][ci] = W[co][ci][k-1-ky][k-1-kx]   (transposed + flipped)
//   mode 2 (stem):     7x7 stride-2 filter re-indexed as a 4x4 filter over the 12 space-to-depth channels
//   mode 3 (dgrad s2): parity-class order, see geom.cuh
// 3x3 filters whose K side has 20 channels carry a SECOND ordering behind the standard k-steps ("K20", geom.cuh):
// 24 k-groups in 6 k-steps instead of 27 in 7, read by the 20-channel kernels that keep the LDS pixel record
// [ch 0-15][ch 16-19][ch 16-19 of the next pixel].
#pragma once
#include "geom.cuh"

#define MIL_PACK_FWD 0
#define MIL_PACK_DGRAD 1
#define MIL_PACK_STEM 2

struct PackJob {
    const float* w;
    const float* bias;
    void* out;
    float* bias_pad;
    int cout, cin, ks, mode;
    int CG, NT, nsteps, dtype;     // nsteps = standard k-steps (+ MIL_K20_STEPS when k20)
    int nsteps_std, k20, pad0_, pad1_;
};

__host__ inline void mil_pack_job_dims(PackJob* j) {
    int cin_exec, cout_exec, ks_exec;
    const int mode = j->mode, cout = j->cout, cin = j->cin, ks = j->ks;
    if (mode == MIL_PACK_STEM) { cin_exec = 16; cout_exec = mil_cpad(cout); ks_exec = 4; }
    else if (mode == MIL_PACK_DGRAD) { cin_exec = mil_cpad(cout); cout_exec = mil_cpad(cin); ks_exec = ks; }
    else { cin_exec = mil_cpad(cin); cout_exec = mil_cpad(cout); ks_exec = ks; }
    j->CG = cin_exec / 8;
    j->NT = (cout_exec + 15) / 16;
    j->nsteps_std = (ks_exec * ks_exec * j->CG + 3) / 4;
    if (mode == MIL_PACK_DGRAD_S2) { j->CG = mil_cpad(cout) / 8; j->NT = (mil_cpad(cin) + 15) / 16; j->nsteps_std = mil_s2_nsteps(j->CG); }
    j->k20 = mil_pack_has_k20(mode, cout, cin, ks) ? 1 : (mil_pack_has_sk6(mode, cout) ? 2 : 0);      // 2: the stem's SK6 order
    j->nsteps = j->nsteps_std + (j->k20 == 1 ? MIL_K20_STEPS : j->k20 == 2 ? MIL_SK6_STEPS : 0);
}

// element idx of job j's packed buffer (also fills bias_pad from the first NT*16 indices)
__device__ __forceinline__ void mil_pack_job_elem(const PackJob& j, int idx) {
    if (idx < j.NT * 16 && j.bias_pad && j.mode != MIL_PACK_DGRAD_S2) {
        const int n_out = (j.mode == MIL_PACK_DGRAD) ? j.cin : j.cout;
        j.bias_pad[idx] = (j.bias && idx < n_out) ? j.bias[idx] : 0.f;
    }
    const int e = idx & 7, lane = (idx >> 3) & 63;
    const int t = idx >> 9;
    const int nt = t % j.NT, s = t / j.NT;
    const int kk = j.ks * j.ks;
    const int nout = nt * 16 + (lane & 15);
    float val = 0.f;
    if (j.mode == MIL_PACK_DGRAD_S2) {      // `bias` carries the projection's weight for this mode
        val = mil_s2_pack_value(j.w, j.bias, s, lane, e, nt, j.cout, j.cin, j.CG);
    } else {
        int tap, kin;
        if (s >= j.nsteps_std) {
            const K20Elem k = j.mode == MIL_PACK_STEM ? mil_sk6_elem(4 * (s - j.nsteps_std) + (lane >> 4), e)
                                                      : mil_k20_elem(4 * (s - j.nsteps_std) + (lane >> 4), e);
            tap = k.tap < 0 ? kk : k.tap; kin = k.ch;
        } else {
            const int q = 4 * s + (lane >> 4);
            tap = q / j.CG; kin = (q - tap * j.CG) * 8 + e;
        }
        if (j.mode == MIL_PACK_FWD) {
            if (tap < kk && kin < j.cin && nout < j.cout) val = j.w[((size_t)nout * j.cin + kin) * kk + tap];
        } else if (j.mode == MIL_PACK_DGRAD) {
            if (tap < kk && kin < j.cout && nout < j.cin) val = j.w[((size_t)kin * j.cin + nout) * kk + (kk - 1 - tap)];
        } else {
            if (tap < 16 && kin < 12 && nout < j.cout) {
                const int c = kin >> 2, dy = (kin >> 1) & 1, dx = kin & 1;
                const int ky = 2 * (tap >> 2) + dy - 1, kx = 2 * (tap & 3) + dx - 1;
                if (ky >= 0 && ky < 7 && kx >= 0 && kx < 7) val = j.w[(((size_t)nout * 3 + c) * 7 + ky) * 7 + kx];
            }
        }
    }
    if (j.dtype == MIL_DT_BF16) reinterpret_cast<__bf16*>(j.out)[idx] = (__bf16)val;
    else if (j.dtype == MIL_DT_F32S) {          // fragment = [hi: 8 bf16][lo: 8 bf16] per lane (32 bytes, as 8 floats)
        const __bf16 h = (__bf16)val;
        __bf16* o = reinterpret_cast<__bf16*>(j.out) + (size_t)(idx >> 3) * 16 + (idx & 7);
        o[0] = h;
        o[8] = (__bf16)(val - (float)h);
    }
    else reinterpret_cast<float*>(j.out)[idx] = val;
}
