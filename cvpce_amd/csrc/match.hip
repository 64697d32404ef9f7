// K11: cosine-distance matrix detections x gallery as ONE MFMA GEMM with a
// fused per-query top-k epilogue -- the (Q,G) matrix is never written to HBM.
// Replaces /root/reference/cvpce/models/classification.py:87-95
// (`distance` = 1 - cosine_similarity, `nearest_neighbors` = argsort[:, :k] of
// a materialised (Q,G,D) x2 gather).
//
//   dist[q][g] = 1 - <Q[q], G[g]> / (max(|Q[q]|,eps) * max(|G[g]|,eps))
//
// Layout: Q [Qn][D], G [Gn][D] row-major (K contiguous for both operands),
// bf16 storage, fp32 accumulate (v_mfma_f32_32x32x16_bf16); or fp32 storage
// on the exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32) for index-exact parity.
// Tile 128 gallery rows x 128 queries, K-step 64, swizzled LDS, register
// staged double buffer (same pipeline as conv_igemm).  Epilogue: the tile's
// distances go to LDS as [g][q] (conflict-free both ways), every query finds
// its k smallest (distance, index) pairs lexicographically (ties -> lowest
// index), partials [Q][tiles_g][k] are merged by a second tiny kernel.
#include "common.h"
#include "../../include/cvpce_amd.h"
#include <math.h>

#define MT_TG 128
#define MT_TQ 128
#define MT_BK 64
#define MATCH_KMAX 16

struct MatchArgs {
    const void* q; const void* g;
    const float* qn; const float* gn;     // L2 norms (already clamped by eps)
    int Qn, Gn, D, k, tiles_g, tiles_q;
    float* part_d; int* part_i;           // [Qn][tiles_g][k]
};

__global__ void row_norm_bf16_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int rows, int D, float eps) {
    const int row = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int i = lane * 8; i < D; i += 64 * 8) {
        uint4 v = *reinterpret_cast<const uint4*>(x + (size_t)row * D + i);
        bf16x8 b = *reinterpret_cast<bf16x8*>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)b[j] * (float)b[j];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[row] = fmaxf(sqrtf(s), eps);
}
__global__ void row_norm_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int D, float eps) {
    const int row = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) { float v = x[(size_t)row * D + i]; s += v * v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[row] = fmaxf(sqrtf(s), eps);
}

extern "C" int cvpce_row_norms(const void* x, float* out, int rows, int D, int is_f32, float eps, void* stream) {
    if (!x || !out || D <= 0 || (!is_f32 && D % 8 != 0)) return CVPCE_ERR_ARG;
    if (rows <= 0) return CVPCE_OK;
    if (is_f32)
        hipLaunchKernelGGL(row_norm_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)x, out, rows, D, eps);
    else
        hipLaunchKernelGGL(row_norm_bf16_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, rows, D, eps);
    return cvpce_check_launch();
}

// lexicographic (d, i) < (e, j)
__device__ __forceinline__ bool lex_lt(float d, int i, float e, int j) { return d < e || (d == e && i < j); }

template <bool F32>
__global__ __launch_bounds__(256, 2) void match_kernel(MatchArgs a) {
    // element size 2 (bf16) or 4 (f32); one LDS row = MT_BK elements
    constexpr int ES = F32 ? 4 : 2;
    constexpr int ROWB = MT_BK * ES;                 // bytes per tile row: 128 / 256
    constexpr int CPR = ROWB / 16;                   // 16-B chunks per row: 8 / 16
    constexpr int RPP = 256 / CPR;                   // 32 / 16
    constexpr int PASS = MT_TG / RPP;                // 4 / 8
    constexpr int RPB = (256 / ROWB) > 0 ? (256 / ROWB) : 1;   // rows per 256-B bank row: 2 / 1
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Gs = smem;                                  // [2][128 rows][ROWB]
    unsigned char* Qs = smem + 2 * MT_TG * ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wc = wid >> 1, wp = wid & 1;
    const int tile_g = blockIdx.x % a.tiles_g, tile_q = blockIdx.x / a.tiles_g;
    const int c = tid % CPR, r0 = tid / CPR;

    const unsigned char* gbase = (const unsigned char*)a.g;
    const unsigned char* qbase = (const unsigned char*)a.q;
    const size_t rowbytes = (size_t)a.D * ES;

    u32x4 greg[PASS], qreg[PASS];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            int gr = tile_g * MT_TG + r0 + i * RPP;
            int qr = tile_q * MT_TQ + r0 + i * RPP;
            u32x4 z = {0u, 0u, 0u, 0u};
            greg[i] = (gr < a.Gn) ? *reinterpret_cast<const u32x4*>(gbase + (size_t)gr * rowbytes + (size_t)kt * ROWB + c * 16) : z;
            qreg[i] = (qr < a.Qn) ? *reinterpret_cast<const u32x4*>(qbase + (size_t)qr * rowbytes + (size_t)kt * ROWB + c * 16) : z;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            int row = r0 + i * RPP;
            int phys = c ^ ((row / RPB) & (CPR - 1));
            *reinterpret_cast<u32x4*>(Gs + (size_t)buf * MT_TG * ROWB + row * ROWB + phys * 16) = greg[i];
            *reinterpret_cast<u32x4*>(Qs + (size_t)buf * MT_TQ * ROWB + row * ROWB + phys * 16) = qreg[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = a.D / MT_BK;
    const int lr = lane & 31, lh = lane >> 5;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);
        const unsigned char* Gb = Gs + (size_t)cur * MT_TG * ROWB;
        const unsigned char* Qb = Qs + (size_t)cur * MT_TQ * ROWB;
        if constexpr (!F32) {
#pragma unroll
            for (int kk = 0; kk < MT_BK / 16; ++kk) {
                const int chunk = kk * 2 + lh;
                bf16x8 af[2], bfr[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    int row = wc * 64 + mt * 32 + lr;
                    af[mt] = *reinterpret_cast<const bf16x8*>(Gb + row * ROWB + ((chunk ^ ((row / RPB) & (CPR - 1))) * 16));
                    int rowq = wp * 64 + mt * 32 + lr;
                    bfr[mt] = *reinterpret_cast<const bf16x8*>(Qb + rowq * ROWB + ((chunk ^ ((rowq / RPB) & (CPR - 1))) * 16));
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
            }
        } else {
            // v_mfma_f32_32x32x2_f32: lane holds A[row lr][k = lh], B[k = lh][col lr]; read 4 k-steps (8 k) per 16-B chunk pair
#pragma unroll
            for (int k4 = 0; k4 < MT_BK / 8; ++k4) {
                // lane half lh reads the 16-B chunk (2*k4 + lh): k = 8*k4 + 4*lh + {0..3}; the 4 MFMAs then pair
                // element e of half 0 with element e of half 1 -- a permutation of k, identical for A and B.
                const int chunk = k4 * 2 + lh;
                f32x4 af[2], bfr[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    int row = wc * 64 + mt * 32 + lr;
                    af[mt] = *reinterpret_cast<const f32x4*>(Gb + row * ROWB + ((chunk ^ ((row / RPB) & (CPR - 1))) * 16));
                    int rowq = wp * 64 + mt * 32 + lr;
                    bfr[mt] = *reinterpret_cast<const f32x4*>(Qb + rowq * ROWB + ((chunk ^ ((rowq / RPB) & (CPR - 1))) * 16));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][e], bfr[nt][e], acc[mt][nt], 0, 0, 0);
            }
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: distances -> LDS [g][q], then per-query top-k -------------
    float* Tl = reinterpret_cast<float*>(smem);                       // 128 x 128 f32 = 64 KiB
    float* cd = reinterpret_cast<float*>(smem + MT_TG * MT_TQ * 4);   // [2][128]
    int* cix = reinterpret_cast<int*>(smem + MT_TG * MT_TQ * 4 + 2 * 128 * 4);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int ql = wp * 64 + nt * 32 + lr;
        const int qg = tile_q * MT_TQ + ql;
        const float qn = (qg < a.Qn) ? a.qn[qg] : 1.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gl = wc * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int gg = tile_g * MT_TG + gl;
                float d = INFINITY;
                if (gg < a.Gn) {
                    d = 1.f - acc[mt][nt][r] / (qn * a.gn[gg]);
                    if (!(d == d)) d = INFINITY;   // a NaN distance (non-finite embedding) sorts last and still yields a valid index
                }
                Tl[gl * MT_TQ + ql] = d;
            }
    }
    __syncthreads();
    const int ql = tid & 127, half = tid >> 7;
    const int qg = tile_q * MT_TQ + ql;
    float pd = -INFINITY;
    int pi = -1;
    for (int r = 0; r < a.k; ++r) {
        float bd = INFINITY;
        int bi = 0x7FFFFFFF;
        for (int g = half * 64; g < half * 64 + 64; ++g) {
            const float d = Tl[g * MT_TQ + ql];
            const int gi = tile_g * MT_TG + g;
            if (lex_lt(pd, pi, d, gi) && lex_lt(d, gi, bd, bi)) { bd = d; bi = gi; }
        }
        cd[half * 128 + ql] = bd;
        cix[half * 128 + ql] = bi;
        __syncthreads();
        const float od = cd[(half ^ 1) * 128 + ql];
        const int oi = cix[(half ^ 1) * 128 + ql];
        if (lex_lt(od, oi, bd, bi)) { bd = od; bi = oi; }
        pd = bd; pi = bi;
        if (half == 0 && qg < a.Qn) {
            const size_t o = ((size_t)qg * a.tiles_g + tile_g) * a.k + r;
            a.part_d[o] = bd;
            a.part_i[o] = bi;
        }
        __syncthreads();
    }
}

__global__ void match_merge_kernel(const float* __restrict__ part_d, const int* __restrict__ part_i, int Qn, int n,
                                   int k, long long* __restrict__ out_idx, float* __restrict__ out_dist) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Qn) return;
    const float* d = part_d + (size_t)q * n;
    const int* ix = part_i + (size_t)q * n;
    float pd = -INFINITY;
    int pi = -1;
    for (int r = 0; r < k; ++r) {
        float bd = INFINITY;
        int bi = 0x7FFFFFFF;
        for (int j = 0; j < n; ++j) {
            const float dj = d[j];
            const int ij = ix[j];
            if (lex_lt(pd, pi, dj, ij) && lex_lt(dj, ij, bd, bi)) { bd = dj; bi = ij; }
        }
        out_idx[(size_t)q * k + r] = (long long)bi;
        if (out_dist) out_dist[(size_t)q * k + r] = bd;
        pd = bd; pi = bi;
    }
}

extern "C" size_t cvpce_match_workspace_bytes(int Qn, int Gn, int k) {
    const size_t tiles_g = (Gn + MT_TG - 1) / MT_TG;
    return (size_t)Qn * tiles_g * k * 8 + 512;
}

extern "C" int cvpce_match_topk(const void* queries, const void* gallery, const float* q_norms, const float* g_norms,
                                int Qn, int Gn, int D, int k, int is_f32, void* workspace, size_t workspace_bytes,
                                long long* out_idx, float* out_dist, void* stream) {
    if (!queries || !gallery || !q_norms || !g_norms || !workspace || !out_idx) return CVPCE_ERR_ARG;
    if (D <= 0 || D % MT_BK != 0 || k < 1 || k > MATCH_KMAX || Gn < k) return CVPCE_ERR_ARG;
    if (Qn <= 0) return CVPCE_OK;
    if (workspace_bytes < cvpce_match_workspace_bytes(Qn, Gn, k)) return CVPCE_ERR_ARG;
    MatchArgs a;
    a.q = queries; a.g = gallery; a.qn = q_norms; a.gn = g_norms; a.Qn = Qn; a.Gn = Gn; a.D = D; a.k = k;
    a.tiles_g = (Gn + MT_TG - 1) / MT_TG;
    a.tiles_q = (Qn + MT_TQ - 1) / MT_TQ;
    a.part_d = (float*)workspace;
    a.part_i = (int*)((char*)workspace + ((size_t)Qn * a.tiles_g * k * 4 + 255) / 256 * 256);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(a.tiles_g * a.tiles_q);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)match_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 128 * 256) != hipSuccess)
            return CVPCE_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)match_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 128 * 4 + 2048) != hipSuccess)
            return CVPCE_ERR_LAUNCH;
        attr_set = true;
    }
    if (is_f32) {
        size_t smem = (size_t)2 * 2 * 128 * 256;    // 128 KiB staging (>= 66 KiB epilogue image)
        hipLaunchKernelGGL(match_kernel<true>, grid, dim3(256), smem, s, a);
    } else {
        size_t smem = (size_t)128 * 128 * 4 + 2048;  // epilogue image dominates (staging needs 64 KiB)
        hipLaunchKernelGGL(match_kernel<false>, grid, dim3(256), smem, s, a);
    }
    hipLaunchKernelGGL(match_merge_kernel, dim3((Qn + 63) / 64), dim3(64), 0, s, a.part_d, a.part_i, Qn, a.tiles_g * k,
                       k, out_idx, out_dist);
    return cvpce_check_launch();
}
