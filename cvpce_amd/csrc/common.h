// Shared device helpers for the cvpce_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define CVPCE_OK 0
#define CVPCE_ERR_ARG 1
#define CVPCE_ERR_LAUNCH 2

#define WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f32_to_bf16(float v) { return (bf16_t)v; }  // v_cvt_pk_bf16_f32: RNE, NaN-preserving

// four floats -> four bf16 with two v_cvt_pk_bf16_f32 (element-wise casts compile to one conversion + a v_perm each)
__device__ __forceinline__ bf16x4 f32x4_to_bf16x4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }

// ---------------------------------------------------------------------------------------------------------------------
// Element type of the 16-bit activations and weights.  The kernels move 16-bit data as opaque 128- / 64-bit words (bf16x8 /
// bf16x4 registers, LDS-DMA pieces, buffer loads, ds_read_b128): the element type matters only where values are MULTIPLIED
// (the MFMA instruction) and where they are CONVERTED (epilogue stores, residual reads).  Every convolution kernel of the
// detector is a template over one of these two policies; both run the matrix pipe at the same rate
// (v_mfma_f32_16x16x32_{bf16,f16}: 8 passes, v_mfma_f32_32x32x16_{bf16,f16}: 16 passes).
//   ElemBF16  the default storage type (BASELINE configs name bf16): 8 exponent / 7 mantissa bits
//   ElemF16   the detector's opt-in accuracy mode (`gln(..., precision='fp16')`): 5 / 10 bits -- 8x finer rounding of
//             weights and activations (head-logit error 1.75 % -> 0.24 % of the logit spread, profiles/r03_numerics_study.json);
//             stores saturate at +-65504 instead of overflowing to infinity.
// `bf16x8` / `bf16x4` / `bf16_t` in a kernel's registers are bit containers under ElemF16.
// ---------------------------------------------------------------------------------------------------------------------
struct ElemBF16 {
    static constexpr bool kF16 = false;
    static __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ bf16x4 pack4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
    static __device__ __forceinline__ bf16_t narrow(float v) { return (bf16_t)v; }
    static __device__ __forceinline__ float widen(bf16_t raw) { return (float)raw; }
};
struct ElemF16 {
    static constexpr bool kF16 = true;
    static __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ bf16x4 pack4(f32x4 v) {   // v_med3_f32 x4 + v_cvt_pk_f16_f32 x2 (RNE)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j], -65504.f, 65504.f);
        return __builtin_bit_cast(bf16x4, __builtin_convertvector(v, f16x4));
    }
    static __device__ __forceinline__ bf16_t narrow(float v) { return __builtin_bit_cast(bf16_t, (f16_t)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f)); }
    static __device__ __forceinline__ float widen(bf16_t raw) { return (float)__builtin_bit_cast(f16_t, raw); }
};

// Bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD
// (round-robin dispatch), so give each XCD label a contiguous chunk of the
// logical tile space -> neighbouring tiles hit the same per-XCD L2.  Speed
// only; correctness never depends on placement.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    if (nwg < 2 * nx) return bid;
    int q = nwg / nx, r = nwg % nx;
    int xcd = bid % nx, idx = bid / nx;
    int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// max over the 2x2 quad formed by 4 adjacent lanes, with DPP quad_perm (one VALU op per step; __shfl_xor would go
// through ds_bpermute = an LDS round trip per value)
__device__ __forceinline__ float quad_max(float x) {
    float y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x = fmaxf(x, y);
    y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));         // quad_perm [2,3,0,1]
    return fmaxf(x, y);
}

// ReLU on the bit pattern: negative floats are negative ints, so max_i32(bits, 0) is relu(x) in ONE v_max_i32 (under
// IEEE mode fmaxf costs a canonicalising v_max x,x in front).  +NaN stays NaN, -NaN becomes 0, -0 becomes +0.
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// max over the 2x2 quad of NON-NEGATIVE floats (i.e. after relu_bits; relu(maxpool(x)) == maxpool(relu(x))): an integer
// max, so each DPP move fuses into its v_max_i32 -- 2 VALU ops per value instead of 6
__device__ __forceinline__ float quad_max_nonneg(float x) {
    int v = __float_as_int(x);
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return __int_as_float(v);
}

// number of workgroups the persistent kernels launch at most (= the CUs their stream may use: 256, or fewer under a CU mask);
// set by cvpce_set_persistent_workgroups (elementwise.hip)
extern int g_cvpce_persistent_wgs;

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instance, device): the attribute belongs to the device that
// is current when it is set, so a process that drives a second GPU must set it there too; the flag is an atomic bit per device
// (two host threads may race to set it -- setting it twice is harmless).
template <auto Kernel>
static inline bool cvpce_smem_attr_done(const void* fn, int bytes) {
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
}

static inline int cvpce_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CVPCE_OK : CVPCE_ERR_LAUNCH;
}
