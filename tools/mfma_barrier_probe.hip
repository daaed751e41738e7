// Probe (not product code): bare 16x16x32 bf16 MFMA loops, one workgroup per CU, 4 / 8 / 16 waves: what does a workgroup barrier every N MFMAs cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int NPB, bool BAR, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const bf16x8_t* in, float* out, int iters) {
    bf16x8_t a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
    f32x4_t acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < NPB / 16; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NPB, bool BAR, int WAVES>
void run(const char* name, const bf16x8_t* in, float* out) {
    const int total_mfma = 36864;   // per wave
    const int iters = total_mfma / NPB;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NPB, BAR, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * WAVES * total_mfma * 16384.0;
    printf("%-34s %7.1f us  %7.1f TFLOP/s\n", name, ms * 1e3, flops / ms * 1e-9);
}

int main() {
    bf16x8_t* in; float* out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 22);
    std::vector<unsigned short> h(1 << 19);
    unsigned s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((s >> 9) & 0x3ff) + ((s >> 31) << 15)); }   // random bf16 around +-1
    hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
    run<32, false, 8>("8 waves, no barrier", in, out);
    run<16, true, 8>("8 waves, barrier per 16 MFMA", in, out);
    run<32, true, 8>("8 waves, barrier per 32 MFMA", in, out);
    run<64, true, 8>("8 waves, barrier per 64 MFMA", in, out);
    run<128, true, 8>("8 waves, barrier per 128 MFMA", in, out);
    run<32, false, 4>("4 waves, no barrier", in, out);
    run<32, true, 4>("4 waves, barrier per 32 MFMA", in, out);
    run<32, false, 16>("16 waves, no barrier", in, out);
    run<32, true, 16>("16 waves, barrier per 32 MFMA", in, out);
    return 0;
}
