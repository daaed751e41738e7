// ABI version + the streaming-copy calibration kernel.
#include "common.cuh"

extern "C" int mil_abi_version(void) { return 2; }

// What HBM gives a plain streaming kernel on this box: dst[i] = src[i], ONE 16-byte piece per thread, as many workgroups as
// pieces.  bench.py times it as the calibration of its roofline fractions (MI355X_MICROARCH.md: 6.29 TB/s measured for a
// float4 copy against the 8 TB/s specification).  Measured here on 1 GiB: this shape 6.23 TB/s; block-contiguous chunks of
// 4-8 pieces per thread on a persistent grid 5.4-5.9 TB/s; a grid-stride loop (pieces of one thread a whole grid apart)
// 4.4-5.1 TB/s; torch's dst.copy_(src) 4.9 TB/s.
__global__ __launch_bounds__(256) void stream_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

extern "C" int mil_stream_copy(void* dst, const void* src, size_t bytes, void* stream) {
    if (!dst || !src || (bytes & 15) || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15)) return MIL_ERR_ARG;
    if (bytes == 0) return MIL_OK;
    const size_t n16 = bytes / 16, blocks = (n16 + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return MIL_ERR_ARG;
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<const uint4*>(src), static_cast<uint4*>(dst), n16);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

// The hi / lo split of MIL_DT_F32S exactly as every split-precision kernel takes it (mil_split2, common.cuh): hi = bf16(v),
// lo = bf16(v - hi), pairs (2i, 2i+1) through v_dot2c_f32_bf16.  A test entry (tests/test_gpu_kernels.py): the dot-product form
// must equal the plain subtraction bit for bit on finite pairs; a non-finite element turns its PAIR partner's lo half into NaN
// (the dot product multiplies the partner's hi by zero: Inf * 0), which the plain form would not — documented, and harmless on
// this path: a tensor with an Inf / NaN in it already fails every finite check downstream.
__global__ void split_probe_kernel(const float* __restrict__ v, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, int n2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    bf16x2_t h, l;
    mil_split2(v[2 * i], v[2 * i + 1], h, l);
    const unsigned hu = __builtin_bit_cast(unsigned, h), lu = __builtin_bit_cast(unsigned, l);
    hi[2 * i] = (unsigned short)(hu & 0xffffu); hi[2 * i + 1] = (unsigned short)(hu >> 16);
    lo[2 * i] = (unsigned short)(lu & 0xffffu); lo[2 * i + 1] = (unsigned short)(lu >> 16);
}
extern "C" int mil_split_probe(const float* v, uint16_t* hi, uint16_t* lo, int n, void* stream) {
    if (!v || !hi || !lo || n < 0 || (n & 1)) return MIL_ERR_ARG;
    if (n == 0) return MIL_OK;
    hipLaunchKernelGGL(split_probe_kernel, dim3((n / 2 + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), v, hi, lo, n / 2);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}

#ifdef MIL_POISON_LDS
// Diagnostic build only: proves that the poisoning reaches the whole dynamic segment (out[i] = LDS word i after MIL_POISON).
__global__ void poison_probe_kernel(unsigned* out, int words) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MIL_POISON(smem);
    for (int i = threadIdx.x; i < words; i += blockDim.x) out[i] = reinterpret_cast<const unsigned*>(smem)[i];
}
extern "C" int mil_poison_probe(unsigned* out, int lds_bytes, void* stream) {
    if (!out || lds_bytes <= 0 || (lds_bytes & 3) || lds_bytes > 160 * 1024) return MIL_ERR_ARG;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(poison_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        return MIL_ERR_LAUNCH;
    hipLaunchKernelGGL(poison_probe_kernel, dim3(1), dim3(256), lds_bytes, reinterpret_cast<hipStream_t>(stream), out, lds_bytes / 4);
    MIL_CHECK_LAUNCH();
    return MIL_OK;
}
#endif
