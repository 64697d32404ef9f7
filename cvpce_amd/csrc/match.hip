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
// Tile 128 gallery rows x 128 | 64 queries (chosen per launch by whole rounds of the chip), K-step 64, swizzled LDS, register
// staged double buffer (same pipeline as conv_igemm).  Epilogue: the tile's
// distances go to LDS as [g][q] (conflict-free both ways), every query finds
// its k smallest (distance, index) pairs lexicographically (ties -> lowest
// index), partials [Q][tiles_g][k] are merged by a second tiny kernel.
#include "common.h"
#include "../../include/cvpce_amd.h"
#include <math.h>
#include <cstdlib>
#include <type_traits>

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

// TQ = queries per tile: 128 (waves 2 x 2, 64 gallery rows x 64 queries each) or 64 (64 x 32 each; 48 KiB of LDS, three
// workgroups per CU).  Which one a launch takes is a question of tile COUNT, not of kernel quality: 1 600 x 10 000 gives
// 13 x 79 = 1 027 tiles of 128 x 128 -- on 512 workgroup slots a third round for three tiles.
template <bool F32, int TQ>
__global__ __launch_bounds__(256, TQ == 64 ? 3 : 2) void match_kernel(MatchArgs a) {
    static_assert(TQ == 128 || TQ == 64, "query tile");
    constexpr int NT = TQ / 64;                      // 32-query MFMA blocks per wave
    // element size 2 (bf16) or 4 (f32); one LDS row = MT_BK elements
    constexpr int ES = F32 ? 4 : 2;
    constexpr int ROWB = MT_BK * ES;                 // bytes per tile row: 128 / 256
    constexpr int CPR = ROWB / 16;                   // 16-B chunks per row: 8 / 16
    constexpr int RPP = 256 / CPR;                   // 32 / 16
    constexpr int PASS = MT_TG / RPP;                // 4 / 8
    constexpr int PASSQ = TQ / RPP;                  // query rows per thread
    constexpr int RPB = (256 / ROWB) > 0 ? (256 / ROWB) : 1;   // rows per 256-B bank row: 2 / 1
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Gs = smem;                                  // [2][128 rows][ROWB]
    unsigned char* Qs = smem + 2 * MT_TG * ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wc = wid >> 1, wp = wid & 1;
    // XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin, each with its own 4-MiB L2; with gallery tiles
    // fastest over blockIdx every XCD touches EVERY gallery tile (20 MB at 10 000 x 1 024) and the operands stream from beyond L2
    // (9.5 TB/s measured at 1 600 queries: 43 FLOP per byte of a 128 x 64 tile = the 400 TFLOP/s the kernel was stuck at).
    // Here XCD x gets a contiguous range of the logical order (xcd_remap), and that order is gallery-class major: class c =
    // the gallery tiles c, c + 8, ... with all their query tiles -- an eighth of the gallery (2.5 MB) stays resident in the
    // XCD's L2 while the query tiles sweep past it.
    int tile_g, tile_q;
    {
        int l = xcd_remap((int)blockIdx.x, (int)gridDim.x);
        int cls = 0, ng = (a.tiles_g + 7) >> 3;
        while (cls < 7 && l >= ng * a.tiles_q) {       // at most 7 steps, scalar
            l -= ng * a.tiles_q;
            ++cls;
            ng = (a.tiles_g - cls + 7) >> 3;
        }
        tile_q = l / ng;
        tile_g = cls + 8 * (l - tile_q * ng);
    }
    const int c = tid % CPR, r0 = tid / CPR;

    const unsigned char* gbase = (const unsigned char*)a.g;
    const unsigned char* qbase = (const unsigned char*)a.q;
    const size_t rowbytes = (size_t)a.D * ES;
    const __amdgpu_buffer_rsrc_t srd_g = __builtin_amdgcn_make_buffer_rsrc((void*)a.g, 0, (unsigned)((size_t)a.Gn * rowbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_q = __builtin_amdgcn_make_buffer_rsrc((void*)a.q, 0, (unsigned)((size_t)a.Qn * rowbytes), 0x00020000);

    // register staging, TWO K-steps deep: the loads of step kt + 2 are issued before the MFMAs of step kt, so a load has a
    // whole iteration (and the barrier) to land before it is written to LDS -- one step deep, every iteration waited out most
    // of an L2 / HBM round trip (~2 us per K-step, the whole kernel was that latency times D / 64)
    constexpr int NS = F32 ? 1 : 2;      // staging slots = K-steps a load has to land (the f32 rows are twice as wide: two sets would spill;
                                         // three slots measured the same as two: 67-69 vs 66 us at 1 600 x 10 000 x 1 024)
    u32x4 greg[NS][PASS], qreg[NS][PASSQ];
    auto load_tile = [&](int kt, int slot) {
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            int gr = tile_g * MT_TG + r0 + i * RPP;
            // buffer loads: rows past the end are out of the descriptor's range and read as zeros -- no branch around the load,
            // so the compiler can count the loads in flight (behind a branch it waits with vmcnt(0): the staging collapses)
            greg[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(srd_g, (unsigned)gr * (unsigned)rowbytes + (unsigned)(kt * ROWB + c * 16), 0, 0);
            if (i < PASSQ) {
                int qr = tile_q * TQ + r0 + i * RPP;
                qreg[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(srd_q, (unsigned)qr * (unsigned)rowbytes + (unsigned)(kt * ROWB + c * 16), 0, 0);
            }
        }
    };
    auto store_tile = [&](int buf, int slot) {
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            int row = r0 + i * RPP;
            int phys = c ^ ((row / RPB) & (CPR - 1));
            *reinterpret_cast<u32x4*>(Gs + (size_t)buf * MT_TG * ROWB + row * ROWB + phys * 16) = greg[slot][i];
            if (i < PASSQ) *reinterpret_cast<u32x4*>(Qs + (size_t)buf * TQ * ROWB + row * ROWB + phys * 16) = qreg[slot][i];
        }
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = a.D / MT_BK;
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) load_tile(sl < nk ? sl : nk - 1, sl);
    store_tile(0, 0);
    __syncthreads();
    int cur = 0;
    // (the loop is unrolled by NS so that the staging slot of a step is a compile-time index: registers, not scratch)
    auto k_step = [&](int kt, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value % NS;  // slot holding step kt + 1; the loads of step kt + NS go to the slot step kt came from
        // No branch around the loads or the LDS stores (past the last K-step they re-fetch the last tile into buffers nobody
        // reads): with one straight-line path the compiler's vmcnt bookkeeping is exact -- behind `if (kt + 2 < nk)` it waited
        // for the minimum over both paths, i.e. for the newest loads as well.
        const int last = nk - 1;
        load_tile(kt + NS < nk ? kt + NS : last, (SLOT + NS - 1) % NS);
        __builtin_amdgcn_sched_barrier(0);      // keep the loads ahead of the MFMA section (the scheduler otherwise sinks them to the barrier)
        const unsigned char* Gb = Gs + (size_t)cur * MT_TG * ROWB;
        const unsigned char* Qb = Qs + (size_t)cur * TQ * ROWB;
        if constexpr (!F32) {
#pragma unroll
            for (int kk = 0; kk < MT_BK / 16; ++kk) {
                const int chunk = kk * 2 + lh;
                bf16x8 af[2], bfr[NT];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    int row = wc * 64 + mt * 32 + lr;
                    af[mt] = *reinterpret_cast<const bf16x8*>(Gb + row * ROWB + ((chunk ^ ((row / RPB) & (CPR - 1))) * 16));
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    int rowq = wp * 32 * NT + nt * 32 + lr;
                    bfr[nt] = *reinterpret_cast<const bf16x8*>(Qb + rowq * ROWB + ((chunk ^ ((rowq / RPB) & (CPR - 1))) * 16));
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
            }
        } else {
            // v_mfma_f32_32x32x2_f32: lane holds A[row lr][k = lh], B[k = lh][col lr]; read 4 k-steps (8 k) per 16-B chunk pair
#pragma unroll
            for (int k4 = 0; k4 < MT_BK / 8; ++k4) {
                // lane half lh reads the 16-B chunk (2*k4 + lh): k = 8*k4 + 4*lh + {0..3}; the 4 MFMAs then pair
                // element e of half 0 with element e of half 1 -- a permutation of k, identical for A and B.
                const int chunk = k4 * 2 + lh;
                f32x4 af[2], bfr[NT];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    int row = wc * 64 + mt * 32 + lr;
                    af[mt] = *reinterpret_cast<const f32x4*>(Gb + row * ROWB + ((chunk ^ ((row / RPB) & (CPR - 1))) * 16));
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    int rowq = wp * 32 * NT + nt * 32 + lr;
                    bfr[nt] = *reinterpret_cast<const f32x4*>(Qb + rowq * ROWB + ((chunk ^ ((rowq / RPB) & (CPR - 1))) * 16));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][e], bfr[nt][e], acc[mt][nt], 0, 0, 0);
            }
        }
        store_tile(cur ^ 1, SLOT);
        // raw barrier: `__syncthreads()` also waits for vmcnt(0), i.e. for the loads of step kt + 2 issued a moment ago -- every
        // K-step then costs a whole L2 / HBM round trip and the two-deep staging hides nothing.  What the hand-off needs is that
        // this wave's LDS stores have landed (lgkmcnt) and that every wave is past its reads of the other buffer (the barrier).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur ^= 1;
    };
    int kt = 0;
    for (; kt + NS <= nk; kt += NS) {
        k_step(kt, std::integral_constant<int, 1 % NS>{});
        if constexpr (NS > 1) k_step(kt + 1, std::integral_constant<int, 2 % NS>{});
        if constexpr (NS > 2) k_step(kt + 2, std::integral_constant<int, 3 % NS>{});
    }
    if constexpr (NS > 1) {                                         // the K-steps left over
        if (kt < nk) k_step(kt, std::integral_constant<int, 1 % NS>{});
        if constexpr (NS > 2) { if (kt + 1 < nk) k_step(kt + 1, std::integral_constant<int, 2 % NS>{}); }
    }

    // ---- epilogue: distances -> LDS [g][q], then per-query top-k -------------
    constexpr int NH = 256 / TQ;                                      // threads per query in the scan: 2 | 4 parts of the 128 gallery rows
    float* Tl = reinterpret_cast<float*>(smem);                       // 128 x TQ f32 = 64 | 32 KiB
    float* cd = reinterpret_cast<float*>(smem + MT_TG * TQ * 4);      // [NH][TQ]
    int* cix = reinterpret_cast<int*>(smem + MT_TG * TQ * 4 + 256 * 4);
    float* s_gn = reinterpret_cast<float*>(smem + MT_TG * TQ * 4 + 2 * 256 * 4);      // the tile's 128 gallery norms (one coalesced load
    if (tid < MT_TG) {                                                                 // instead of 64 scattered ones per lane)
        const int gg = tile_g * MT_TG + tid;
        s_gn[tid] = (gg < a.Gn) ? a.gn[gg] : 1.f;
    }
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ql = wp * 32 * NT + nt * 32 + lr;
        const int qg = tile_q * TQ + ql;
        const float qn = (qg < a.Qn) ? a.qn[qg] : 1.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gl = wc * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int gg = tile_g * MT_TG + gl;
                float d = INFINITY;
                if (gg < a.Gn) {
                    d = 1.f - acc[mt][nt][r] / (qn * s_gn[gl]);
                    if (!(d == d)) d = INFINITY;   // a NaN distance (non-finite embedding) sorts last and still yields a valid index
                }
                Tl[gl * TQ + ql] = d;
            }
    }
    __syncthreads();
    const int ql = tid & (TQ - 1), part = tid / TQ;
    const int qg = tile_q * TQ + ql;
    float pd = -INFINITY;
    int pi = -1;
    for (int r = 0; r < a.k; ++r) {
        float bd = INFINITY;
        int bi = 0x7FFFFFFF;
        for (int g0 = part * (MT_TG / NH); g0 < (part + 1) * (MT_TG / NH); g0 += 8) {
            float dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) dv[u] = Tl[(g0 + u) * TQ + ql];             // 8 LDS reads in flight
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int gi = tile_g * MT_TG + g0 + u;
                if (lex_lt(pd, pi, dv[u], gi) && lex_lt(dv[u], gi, bd, bi)) { bd = dv[u]; bi = gi; }
            }
        }
        cd[part * TQ + ql] = bd;
        cix[part * TQ + ql] = bi;
        __syncthreads();
#pragma unroll
        for (int p = 1; p < NH; ++p) {                   // every part learns the tile's minimum: it bounds its next round
            const float od = cd[((part + p) & (NH - 1)) * TQ + ql];
            const int oi = cix[((part + p) & (NH - 1)) * TQ + ql];
            if (lex_lt(od, oi, bd, bi)) { bd = od; bi = oi; }
        }
        pd = bd; pi = bi;
        if (part == 0 && qg < a.Qn) {
            const size_t o = ((size_t)qg * a.tiles_g + tile_g) * a.k + r;
            a.part_d[o] = bd;
            a.part_i[o] = bi;
        }
        __syncthreads();
    }
}

// One WAVE per query: the n partial (distance, index) pairs are spread over the lanes (coalesced loads, all in flight at once --
// one thread walking them paid an L2 round trip per entry), each of the k rounds takes the wave-wide lexicographic minimum of
// the entries greater than the previous pick.
__global__ __launch_bounds__(256) void match_merge_kernel(const float* __restrict__ part_d, const int* __restrict__ part_i, int Qn, int n,
                                                          int k, long long* __restrict__ out_idx, float* __restrict__ out_dist) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= Qn) return;
    const float* d = part_d + (size_t)q * n;
    const int* ix = part_i + (size_t)q * n;
    constexpr int PER = 8;                      // entries per lane held in registers: n <= 512, else the strided loop below
    float dv[PER];
    int iv[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int j = e * 64 + lane;
        dv[e] = (j < n) ? d[j] : INFINITY;
        iv[e] = (j < n) ? ix[j] : 0x7FFFFFFF;
    }
    float pd = -INFINITY;
    int pi = -1;
    for (int r = 0; r < k; ++r) {
        float bd = INFINITY;
        int bi = 0x7FFFFFFF;
#pragma unroll
        for (int e = 0; e < PER; ++e)
            if (lex_lt(pd, pi, dv[e], iv[e]) && lex_lt(dv[e], iv[e], bd, bi)) { bd = dv[e]; bi = iv[e]; }
        for (int j = PER * 64 + lane; j < n; j += 64) {            // (n > 512: the tail straight from memory)
            const float dj = d[j];
            const int ij = ix[j];
            if (lex_lt(pd, pi, dj, ij) && lex_lt(dj, ij, bd, bi)) { bd = dj; bi = ij; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(bd, off);
            const int oi = __shfl_xor(bi, off);
            if (lex_lt(od, oi, bd, bi)) { bd = od; bi = oi; }
        }
        if (lane == 0) {
            out_idx[(size_t)q * k + r] = (long long)bi;
            if (out_dist) out_dist[(size_t)q * k + r] = bd;
        }
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
    {   // 32-bit buffer offsets (rows rounded up to whole tiles stay below 2^32)
        const unsigned long long es = is_f32 ? 4 : 2;
        if (((unsigned long long)Gn + MT_TG) * D * es >= (1ull << 32) || ((unsigned long long)Qn + MT_TQ) * D * es >= (1ull << 32)) return CVPCE_ERR_ARG;
    }
    MatchArgs a;
    a.q = queries; a.g = gallery; a.qn = q_norms; a.gn = g_norms; a.Qn = Qn; a.Gn = Gn; a.D = D; a.k = k;
    a.tiles_g = (Gn + MT_TG - 1) / MT_TG;
    // query tile of the bf16 kernel: 64-query tiles cost ~0.6 of a 128-query tile and run three to a CU instead of two; the
    // cheaper schedule in whole rounds of the chip wins (CVPCE_MATCH_TQ = 64 | 128 forces one: dev A/B)
    int tq = MT_TQ;
    if (!is_f32) {
        static const int forced = []() { const char* e = getenv("CVPCE_MATCH_TQ"); return e ? atoi(e) : 0; }();
        const long long cus = g_cvpce_persistent_wgs;
        const long long t128 = (long long)a.tiles_g * ((Qn + 127) / 128), t64 = (long long)a.tiles_g * ((Qn + 63) / 64);
        const double c128 = (double)((t128 + 2 * cus - 1) / (2 * cus)) * 1.0, c64 = (double)((t64 + 3 * cus - 1) / (3 * cus)) * 0.6;
        tq = forced == 64 || forced == 128 ? forced : (c64 < c128 ? 64 : 128);
    }
    a.tiles_q = (Qn + tq - 1) / tq;
    a.part_d = (float*)workspace;
    a.part_i = (int*)((char*)workspace + ((size_t)Qn * a.tiles_g * k * 4 + 255) / 256 * 256);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(a.tiles_g * a.tiles_q);
    if (!cvpce_smem_attr_done<match_kernel<true, 128>>((const void*)match_kernel<true, 128>, 2 * 2 * 128 * 256) ||
        !cvpce_smem_attr_done<match_kernel<false, 128>>((const void*)match_kernel<false, 128>, 128 * 128 * 4 + 2048 + 512) ||
        !cvpce_smem_attr_done<match_kernel<false, 64>>((const void*)match_kernel<false, 64>, 2 * (128 + 64) * 128))
        return CVPCE_ERR_LAUNCH;
    if (is_f32) {
        size_t smem = (size_t)2 * 2 * 128 * 256;    // 128 KiB staging (>= 66 KiB epilogue image)
        hipLaunchKernelGGL((match_kernel<true, 128>), grid, dim3(256), smem, s, a);
    } else if (tq == 128) {
        size_t smem = (size_t)128 * 128 * 4 + 2048 + 512;  // epilogue image (+ candidates, + the tile's gallery norms) dominates (staging needs 64 KiB)
        hipLaunchKernelGGL((match_kernel<false, 128>), grid, dim3(256), smem, s, a);
    } else {
        size_t smem = (size_t)2 * (128 + 64) * 128;        // 48 KiB of staging dominates (epilogue image 32 KiB + 2.5 KiB)
        hipLaunchKernelGGL((match_kernel<false, 64>), grid, dim3(256), smem, s, a);
    }
    hipLaunchKernelGGL(match_merge_kernel, dim3((Qn + 3) / 4), dim3(256), 0, s, a.part_d, a.part_i, Qn, a.tiles_g * k,
                       k, out_idx, out_dist);
    return cvpce_check_launch();
}
