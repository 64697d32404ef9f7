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

// Second calibration probe: how fast the CUs can stream an L2-resident operand into registers -- the way the halo kernels read
// their weights (buffer_load_dwordx4, every workgroup walks the SAME `bytes`-long buffer `iters` times, 8 loads in flight per
// lane).  bytes <= 4 MiB stays in each XCD's L2; the result bounds every design that re-streams weights per pixel tile (the
// Winograd F(2x2,3x3) study of DESIGN.md needs 4x the weight bytes per MFMA FLOP of the direct kernel).
__global__ __launch_bounds__(512, 2) void l2_stream_probe_kernel(const u32x4* __restrict__ buf, unsigned n16, int iters, float* __restrict__ sink) {
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, n16 * 16u, 0x00020000);
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int it = 0; it < iters; ++it) {
        for (unsigned base = threadIdx.x; base + 7u * 512u < n16; base += 8u * 512u) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(srd, (base + (unsigned)u * 512u) * 16u, 0, 0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
        asm volatile("" : "+v"(acc));     // the next walk is not merged with this one
    }
    sink[(size_t)blockIdx.x * 512 + threadIdx.x] = __uint_as_float(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
}

extern "C" int cvpce_probe_l2_stream(const void* buf, long long bytes, int iters, float* sink, int workgroups, void* stream) {
    if (!buf || !sink || iters < 1 || workgroups < 1 || bytes < 8 * 512 * 16 || bytes % (8 * 512 * 16) != 0 || bytes >= (1LL << 32)) return CVPCE_ERR_ARG;
    hipLaunchKernelGGL(l2_stream_probe_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, (const u32x4*)buf, (unsigned)(bytes / 16), iters, sink);
    return cvpce_check_launch();
}
