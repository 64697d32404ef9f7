// Calibration probe, not part of the hot path: a bare bf16 MFMA loop (operands in registers, random data, one wave per
// SIMD, no memory traffic inside the loop) -- the sustained matrix-pipe ceiling of THIS device under its power limit.
// bench.py runs it for >= 2 s and reports the result beside the nominal 2.5 PFLOP/s in `measured_peaks`, so that
// "fraction of the attainable MFMA rate" is checkable from the driver's own bench line (MI355X_MICROARCH.md, DVFS
// give-back items 6-7: no sustained bf16 MFMA stream holds 2.4 GHz on random data, and the clock depends on the MFMA shape).
#include "common.h"
#include "../../include/cvpce_amd.h"

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void mfma_probe_kernel(const bf16_t* __restrict__ operands, float* __restrict__ sink, int iters) {
    extern __shared__ unsigned char pad_lds[];   // 96 KiB requested at launch: one workgroup per CU = one wave per SIMD
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(operands + ((size_t)((blockIdx.x * 4 + wave) % 16) * 8 + i) * 512 + lane * 8);
        b[i] = *reinterpret_cast<const bf16x8*>(operands + ((size_t)((blockIdx.x * 4 + wave) % 16) * 8 + 4 + i) * 512 + lane * 8);
    }
    float total = 0.f;
    if constexpr (SHAPE == 0) {
        f32x16 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)   // in-place accumulate, pinned by asm: the compiler otherwise rotates the loop-carried accumulators through copies
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));   // 256 accumulator registers: AGPRs
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) total += acc[i][j][e];
    } else {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) total += acc[i][j][e];
    }
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = total;
    if (iters < 0) pad_lds[threadIdx.x] = 0;   // keeps the LDS request alive
}

extern "C" int cvpce_probe_mfma_bf16(int shape, int iters, const void* operands, float* sink, int workgroups, void* stream) {
    if (!operands || !sink || iters < 1 || workgroups < 1 || (shape != 0 && shape != 1)) return CVPCE_ERR_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)mfma_probe_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mfma_probe_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
            return CVPCE_ERR_LAUNCH;
        attr_set = true;
    }
    if (shape == 0)
        hipLaunchKernelGGL(mfma_probe_kernel<0>, dim3(workgroups), dim3(256), 96 * 1024, (hipStream_t)stream, (const bf16_t*)operands, sink, iters);
    else
        hipLaunchKernelGGL(mfma_probe_kernel<1>, dim3(workgroups), dim3(256), 96 * 1024, (hipStream_t)stream, (const bf16_t*)operands, sink, iters);
    return cvpce_check_launch();
}
