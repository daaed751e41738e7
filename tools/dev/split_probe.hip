// Stand-alone probe (GPU box): is lo = bf16(v - bf16(v)) through v_dot2c_f32_bf16 bit-identical to the plain subtraction?
//   hipcc --offload-arch=gfx950 -O3 tools/dev/split_probe.hip -o /tmp/split_probe && /tmp/split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__global__ void k(const float* x, unsigned* plain, unsigned* dot, float* fplain, float* fdot, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float v0 = x[2 * i], v1 = x[2 * i + 1];
    bf16x2_t h; h[0] = (__bf16)v0; h[1] = (__bf16)v1;
    // the constant pairs through an opaque SGPR: written as literals hipcc (ROCm 7.2) folds {-1, 0} into the INLINE constant -1.0,
    // which the instruction reads as 0xBF800000 = {0, -1} (wrong operand, no diagnostic)
    unsigned c0 = 0x0000BF80u, c1 = 0xBF800000u;
    asm("" : "+s"(c0)); asm("" : "+s"(c1));
    const bf16x2_t m0 = __builtin_bit_cast(bf16x2_t, c0), m1 = __builtin_bit_cast(bf16x2_t, c1);
    const float d0 = __builtin_amdgcn_fdot2_f32_bf16(h, m0, v0, false), d1 = __builtin_amdgcn_fdot2_f32_bf16(h, m1, v1, false);
    const float p0 = v0 - (float)h[0], p1 = v1 - (float)h[1];
    bf16x2_t lp, ld;
    lp[0] = (__bf16)p0; lp[1] = (__bf16)p1; ld[0] = (__bf16)d0; ld[1] = (__bf16)d1;
    plain[i] = __builtin_bit_cast(unsigned, lp); dot[i] = __builtin_bit_cast(unsigned, ld);
    fplain[2 * i] = p0; fplain[2 * i + 1] = p1; fdot[2 * i] = d0; fdot[2 * i + 1] = d1;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n);
    srand(3);
    for (int i = 0; i < n; ++i) {
        const int m = i % 4;
        float v = (float)rand() / RAND_MAX * 2.f - 1.f;
        if (m == 1) v *= 1e-3f; if (m == 2) v *= 1e3f; if (m == 3) v *= 1e-20f;
        x[i] = v;
    }
    x[0] = 0.f; x[1] = -0.f; x[2] = 1.f; x[3] = 1e-39f; x[4] = 3.0e38f; x[5] = 1.0039062f;
    float *dx, *fp, *fd; unsigned *dp, *dd;
    hipMalloc(&dx, n * 4); hipMalloc(&fp, n * 4); hipMalloc(&fd, n * 4); hipMalloc(&dp, n * 2); hipMalloc(&dd, n * 2);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dp, dd, fp, fd, n);
    std::vector<unsigned> p(n / 2), d(n / 2); std::vector<float> a(n), b(n);
    hipMemcpy(p.data(), dp, n * 2, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(a.data(), fp, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), fd, n * 4, hipMemcpyDeviceToHost);
    long nbf = 0, nf = 0; int shown = 0;
    for (int i = 0; i < n / 2; ++i) if (p[i] != d[i]) ++nbf;
    for (int i = 0; i < n; ++i) if (memcmp(&a[i], &b[i], 4)) { ++nf; if (shown++ < 12) printf("v=%.9g plain=%.9g (%08x) dot=%.9g (%08x)\n", x[i], a[i], *(unsigned*)&a[i], b[i], *(unsigned*)&b[i]); }
    printf("fp32 differences: %ld of %d; packed bf16 lo differences: %ld of %d pairs\n", nf, n, nbf, n / 2);
    return 0;
}
