// Shared device helpers for the MI355X (gfx950) MIL hot path.  CDNA4 only: 64-lane waves,
// v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32, ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// Split-precision operands: v = hi + lo with hi = bf16(v), lo = bf16(v - hi).  v - hi is exact in fp32, and v_dot2c_f32_bf16
// against the constant pair (-1, 0) / (0, -1) computes it straight from the PACKED hi pair: one conversion, two dot
// instructions and one conversion per two values (the plain form unpacks hi first: 13 vector instructions per four values
// against 8).  Every commit and every split epilogue of the MIL_DT_F32S kernels goes through these two.
#ifndef MIL_SPLIT_DOT2
#define MIL_SPLIT_DOT2 1
#endif
__device__ __forceinline__ void mil_split2(float v0, float v1, bf16x2_t& hi, bf16x2_t& lo) {
    hi[0] = (__bf16)v0; hi[1] = (__bf16)v1;
#if MIL_SPLIT_DOT2
    // The constant pairs go through an opaque SGPR: written as literals, hipcc (ROCm 7.2) folds {-1, 0} into the INLINE constant
    // -1.0, which v_dot2c_f32_bf16 reads as 0xBF800000 = {0, -1} — the wrong operand, no diagnostic (tools/dev/split_probe.hip:
    // every second value wrong as literals, 0 differences in 4 M values incl. zeros, denormals and 3e38 this way).
    unsigned c0 = 0x0000BF80u, c1 = 0xBF800000u;
    asm("" : "+s"(c0));
    asm("" : "+s"(c1));
    const bf16x2_t m0 = __builtin_bit_cast(bf16x2_t, c0), m1 = __builtin_bit_cast(bf16x2_t, c1);
    lo[0] = (__bf16)__builtin_amdgcn_fdot2_f32_bf16(hi, m0, v0, false);
    lo[1] = (__bf16)__builtin_amdgcn_fdot2_f32_bf16(hi, m1, v1, false);
#else
    lo[0] = (__bf16)(v0 - (float)hi[0]); lo[1] = (__bf16)(v1 - (float)hi[1]);
#endif
}
__device__ __forceinline__ void mil_split4(const f32x4_t& v, bf16x4_t& hi, bf16x4_t& lo) {
    bf16x2_t h0, l0, h1, l1;
    mil_split2(v[0], v[1], h0, l0);
    mil_split2(v[2], v[3], h1, l1);
    hi = bf16x4_t{h0[0], h0[1], h1[0], h1[1]};
    lo = bf16x4_t{l0[0], l0[1], l1[0], l1[1]};
}

#define MIL_OK 0
#define MIL_ERR_ARG 1
#define MIL_ERR_UNSUPPORTED 2
#define MIL_ERR_LAUNCH 3

#define MIL_DT_F32 0
#define MIL_DT_BF16 1
// bf16 with the GRADIENT tensors of a 20-channel layer (dz / addend / dx, the pooled-output gradient of the stem) stored
// dense, 20 channels = 40 bytes per pixel, instead of padded to 24; activations keep the padded layout.  Accepted by the
// kernels that form that gradient chain: mil_conv_dgrad_s2 (its output), mil_conv_bwd_fused, mil_stem_bwd_fused(_nchw).
#define MIL_DT_BF16_DGRAD 2
// fp32 tensors in HBM (the layout and every pointwise kernel of MIL_DT_F32), contracted as THREE bf16 MFMAs per k-step:
// each operand is split on its way into LDS into hi = bf16(v) and lo = bf16(v - hi) and the product is taken as
// hi*hi + lo*hi + hi*lo with fp32 accumulation (the lo*lo term, 2^-18 relative, is dropped).  16 significant bits per
// operand at 3/16 of the matrix-pipe time of the exact-f32 MFMA: the path that meets the 1e-3 gate on the logits at speed.
#define MIL_DT_F32S 3
// MIL_DT_F32S with the GRADIENT tensors of a 20-channel layer stored dense: 20 fp32 channels = 80 bytes = five 16-byte pieces
// per pixel instead of 96 (the same chain of kernels as MIL_DT_BF16_DGRAD: mil_conv_dgrad_s2's output, mil_conv_bwd_fused,
// mil_stem_bwd_fused_nchw).
#define MIL_DT_F32S_DGRAD 4

// Channel padding used by every NHWC activation tensor (multiple of 8 elements = one 16-B bf16 piece).
__host__ __device__ constexpr int mil_cpad(int c) { return (c + 7) / 8 * 8; }
// Output-channel tile count (16-wide MFMA columns) <-> padded channel count.
__host__ __device__ constexpr int mil_nt_to_cp(int nt) { return nt == 2 ? 24 : nt == 3 ? 40 : nt == 4 ? 64 : 80; }

// ESZ: bytes per element in HBM and per element of an LDS pixel record; CGB: bytes between two 8-channel groups of one
// LDS pixel record; TR16: the operands in LDS are bf16 (ds_read_b64_tr_b16 applies); SPLIT: hi/lo bf16 planes in LDS.
struct F32 {
    using elem = float;
    static constexpr int ESZ = 4;
    static constexpr int DT = MIL_DT_F32;
    static constexpr int CGB = 32;
    static constexpr bool TR16 = false, SPLIT = false;
};
struct BF16 {
    using elem = __bf16;
    static constexpr int ESZ = 2;
    static constexpr int DT = MIL_DT_BF16;
    static constexpr int CGB = 16;
    static constexpr bool TR16 = true, SPLIT = false;
};
// MIL_DT_F32S: fp32 in HBM; an LDS pixel record of C channels is [hi: C bf16][lo: C bf16] (the same 4*C bytes), a packed
// filter fragment is [hi: 8 bf16][lo: 8 bf16] per lane (the same 32 bytes as 8 floats).
struct F32S {
    using elem = float;
    static constexpr int ESZ = 4;
    static constexpr int DT = MIL_DT_F32S;
    static constexpr int CGB = 16;
    static constexpr bool TR16 = true, SPLIT = true;
};

// One MFMA operand fragment = 8 consecutive-k elements per lane.
template <typename T> struct Frag8;
template <> struct Frag8<BF16> { bf16x8_t v; };
template <> struct Frag8<F32> { f32x4_t lo, hi; };
template <> struct Frag8<F32S> { bf16x8_t h, l; };

template <typename T>
__device__ __forceinline__ Frag8<T> lds_frag(const char* p);
template <>
__device__ __forceinline__ Frag8<BF16> lds_frag<BF16>(const char* p) {
    Frag8<BF16> f; f.v = *reinterpret_cast<const bf16x8_t*>(p); return f;
}
template <>
__device__ __forceinline__ Frag8<F32> lds_frag<F32>(const char* p) {
    Frag8<F32> f;
    f.lo = *reinterpret_cast<const f32x4_t*>(p);
    f.hi = *reinterpret_cast<const f32x4_t*>(p + 16);
    return f;
}

// Fragment of a packed filter (32 bytes per lane: [hi][lo]) / of an LDS pixel record whose lo plane starts LO bytes behind
// its hi plane (LO = 2 * channels of the record).
template <>
__device__ __forceinline__ Frag8<F32S> lds_frag<F32S>(const char* p) {
    Frag8<F32S> f;
    f.h = *reinterpret_cast<const bf16x8_t*>(p);
    f.l = *reinterpret_cast<const bf16x8_t*>(p + 16);
    return f;
}
template <typename T, int LO>
__device__ __forceinline__ Frag8<T> lds_pix_frag(const char* p) {
    if constexpr (T::SPLIT) {
        Frag8<T> f;
        f.h = *reinterpret_cast<const bf16x8_t*>(p);
        f.l = *reinterpret_cast<const bf16x8_t*>(p + LO);
        return f;
    } else {
        return lds_frag<T>(p);
    }
}

// v = hi + lo (+ a remainder below 2^-17 |v|): the two bf16 MFMA operands of a MIL_DT_F32S value.
__device__ __forceinline__ void mil_split8(const float (&v)[8], bf16x8_t& hi, bf16x8_t& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        hi[j] = h;
        lo[j] = (__bf16)(v[j] - (float)h);
    }
}

// acc += A(16 x 32k) * B(32k x 16).  The bf16 form is one v_mfma_f32_16x16x32_bf16; the f32 form is
// eight exact-f32 v_mfma_f32_16x16x4_f32 over the same 32 k (element j of lane-group g is k=(g,j) on
// both operands, so any consistent k order is a valid contraction).
__device__ __forceinline__ f32x4_t mma8(const Frag8<BF16>& a, const Frag8<BF16>& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mma8(const Frag8<F32>& a, const Frag8<F32>& b, f32x4_t c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[j], b.lo[j], c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[j], b.hi[j], c, 0, 0, 0);
    return c;
}

// split form: the small cross terms first, then hi*hi
__device__ __forceinline__ f32x4_t mma8(const Frag8<F32S>& a, const Frag8<F32S>& b, f32x4_t c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
}

// 8 consecutive channels of a tensor <-> 8 floats.
template <typename T> __device__ __forceinline__ void load8(const typename T::elem* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<BF16>(const __bf16* p, float (&v)[8]) {
    bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
}
template <> __device__ __forceinline__ void load8<F32>(const float* p, float (&v)[8]) {
    f32x4_t a = *reinterpret_cast<const f32x4_t*>(p);
    f32x4_t b = *reinterpret_cast<const f32x4_t*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}
template <> __device__ __forceinline__ void load8<F32S>(const float* p, float (&v)[8]) { load8<F32>(p, v); }
template <typename T> __device__ __forceinline__ void store8(typename T::elem* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<BF16>(__bf16* p, const float (&v)[8]) {
    bf16x8_t t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8_t*>(p) = t;
}
template <> __device__ __forceinline__ void store8<F32>(float* p, const float (&v)[8]) {
    f32x4_t a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
    *reinterpret_cast<f32x4_t*>(p) = a;
    *reinterpret_cast<f32x4_t*>(p + 4) = b;
}

template <> __device__ __forceinline__ void store8<F32S>(float* p, const float (&v)[8]) { store8<F32>(p, v); }

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
// derivative of LeakyReLU read off the saved OUTPUT (same sign as the pre-activation for slope>0;
// torch uses grad = x > 0 ? 1 : slope, so 0 maps to slope).
__device__ __forceinline__ float lrelu_grad(float out, float slope) { return out > 0.f ? 1.f : slope; }

// LDS pixel pitch in bytes for a tile of [pixel][C] elements: an odd number of 16-B slots so that 16
// consecutive pixels read by one ds_read_b128 lane group fall on distinct bank slots.
__host__ __device__ constexpr int mil_pix_pitch(int cp, int esz) {
    int n16 = cp * esz / 16;
    return ((n16 & 1) ? n16 : n16 + 1) * 16;
}

// -DMIL_POISON_LDS (diagnostic build, `make POISON=1` -> libmil_hip_poison.so): every kernel fills its whole dynamic LDS
// segment with bf16 NaNs (0x7FC0 0x7FC0 = an fp32 NaN too) before doing anything else, so that a read of LDS bytes the
// kernel never wrote — a zero-weight padding k-step reading behind a tile, a row tile that does not exist — turns into a
// NaN in the output deterministically (0 x NaN) instead of depending on what the previous workgroup left there.
// The segment size is the `hidden_dynamic_lds_size` implicit kernel argument (code object v5: byte 120).
#ifdef MIL_POISON_LDS
__device__ __forceinline__ void mil_poison_lds(void* base) {
    typedef __attribute__((address_space(4))) const unsigned* cu32p;
    const unsigned bytes = ((cu32p)__builtin_amdgcn_implicitarg_ptr())[30];
    unsigned* w = reinterpret_cast<unsigned*>(base);
    for (unsigned i = threadIdx.x; i < bytes / 4; i += blockDim.x) w[i] = 0x7FC07FC0u;
    __syncthreads();
}
#define MIL_POISON(base) mil_poison_lds(base)
#else
#define MIL_POISON(base) ((void)0)
#endif

// A/B switches of the development builds.  The shipped library reads NO ambient environment for kernel selection: the
// switches exist only in `make VARIANT=<name> EXTRA=-DMIL_AB_SWITCHES` builds (tools/README.md).  (Five TEST knobs stay in
// every build, because the -m gpu tests drive small inputs through the large-launch paths with them: MIL_PF_MIN_TILES,
// MIL_BUFFER_LIMIT_BYTES, MIL_RES_GRID_CAP, MIL_BLOCK_STRIP and MIL_STEM_WALK (0 / 1: the tiled / the row-walk form of the
// identity-block forward / of the fused stem forward whatever the launch size); and MIL_LIB_PATH on the Python side selects
// which build is loaded.)
#include <cstdlib>
__host__ inline const char* mil_ab_env(const char* name) {
#ifdef MIL_AB_SWITCHES
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// "Has this been done on the CURRENT device yet?" for per-device one-time set-up (hipFuncSetAttribute is a per-device
// setting: a process-wide flag would leave a second GPU at the 64 KB default).  `done` is one bit per device ordinal;
// two threads racing through the first call both do the (idempotent) set-up.
#include <atomic>
__host__ inline bool mil_device_needs(std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    return !(done.load(std::memory_order_acquire) & (1ull << (dev & 63)));
}
__host__ inline void mil_device_done(std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) done.fetch_or(1ull << (dev & 63), std::memory_order_release);
}

#define MIL_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return MIL_ERR_LAUNCH; } while (0)
