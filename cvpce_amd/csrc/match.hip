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
    // one-launch form of match_big_kernel (k = 1): per-query 64-bit (distance, row) keys combined by agent-scope atomic minima, the
    // last workgroup to finish (ticket) writes the results and restores the state block (cvpce_match_state_init)
    unsigned long long* keys; int* ticket; long long* out_idx; float* out_dist;
};

// (distance, row) as ONE unsigned 64-bit key whose integer order is the lexicographic order: the float's bits made monotone
// (negative: all bits flipped, non-negative: sign bit set) above the row index
__device__ __forceinline__ unsigned long long match_key(float d, int i) {
    const unsigned b = __float_as_uint(d);
    const unsigned o = b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
    return ((unsigned long long)o << 32) | (unsigned)i;
}
__device__ __forceinline__ float match_key_distance(unsigned long long key) {
    const unsigned o = (unsigned)(key >> 32);
    return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

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

// The distance of one (query, gallery row) pair from its dot product.  bf16 operands: 1 - dot * (1/|q| * 1/|g|), one multiply and one
// fused multiply-add per pair with the two reciprocals formed once per row (a true division per pair is ~10 VALU instructions: in the
// 256 x 320 tile's epilogue that was 160 divisions per lane, a fifth of the kernel).  Every bf16 kernel of this file forms the distance by
// THIS expression from the same reciprocals, so a query's distances do not depend on the kernel that computed them.  f32 operands (the
// reference's arithmetic): the literal 1 - dot / (|q| |g|).  `nq`, `ng`: the norm (F32) or its reciprocal (bf16), see match_norm_term.
template <bool F32>
__device__ __forceinline__ float match_distance(float dot, float nq, float ng) {
    if constexpr (F32) return 1.f - dot / (nq * ng);
    else return __builtin_fmaf(-dot, nq * ng, 1.f);
}
template <bool F32>
__device__ __forceinline__ float match_norm_term(float norm) { return F32 ? norm : 1.f / norm; }

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
        s_gn[tid] = match_norm_term<F32>((gg < a.Gn) ? a.gn[gg] : 1.f);
    }
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ql = wp * 32 * NT + nt * 32 + lr;
        const int qg = tile_q * TQ + ql;
        const float qn = match_norm_term<F32>((qg < a.Qn) ? a.qn[qg] : 1.f);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gl = wc * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int gg = tile_g * MT_TG + gl;
                float d = INFINITY;
                if (gg < a.Gn) {
                    d = match_distance<F32>(acc[mt][nt][r], qn, s_gn[gl]);
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

// ---------------------------------------------------------------------------------------------------------------------
// Large query batches (round 5; BASELINE configs[3] at 1 600 queries): a 256-gallery-row x (64 NQ)-query tile per workgroup of
// EIGHT waves (4 along the gallery x 2 along the queries: a wave owns 64 gallery rows x 32 NQ queries = 2 x NQ accumulators of
// v_mfma_f32_32x32x16_bf16), both operands brought in by LDS-DMA (`buffer_load_dwordx4 ... lds`: 8 neighbouring lanes fetch one
// whole 128-byte line of a row; the XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is applied to the SOURCE
// address, the LDS image of a piece is lane-linear) into two 64-deep K-stages, ONE raw barrier per K-stage, and the hand-off
// hidden behind MFMAs: the last of a stage's four 16-deep steps is multiplied AFTER the next stage's barrier, while that stage's
// first fragments are on their way from LDS.  The 128 x 128 kernel above re-reads every operand byte twice as often from L2
// (2 x (128 + 128) rows per 2 x 128 x 128 products against (256 + 320) rows per 256 x 320) and stages through registers + ds_write:
// at 1 600 x 10 000 x 1 024 its K-steps are bound by that traffic (477 TFLOP/s).
//
// Every product is accumulated by the SAME instruction in the SAME K order as in `match_kernel` (32x32x16, 16-deep steps in
// ascending K, one accumulator per (gallery row, query)), and the distance is formed by the same expression: a query's distances do
// not depend on which of the kernels -- i.e. on how many queries it was batched with -- computed them.
// The tile's k nearest rows per query are selected IN REGISTERS (a lane holds 32 gallery rows of each of its NQ queries, in
// ascending row order), combined across the two lane halves by a lane exchange and across the four gallery waves through 10 KiB
// of LDS; partials [Q][tiles_g][k] as above, merged by match_merge_kernel.
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_m;
typedef __attribute__((address_space(3))) char lds_char_m;
#define MB_ROWB 128                        // bytes of one row of a K-stage (64 bf16)
#ifndef MB_ABL
#define MB_ABL 0
#endif

// MG = gallery fragments (32 rows) per wave: the tile is TG = 128 MG gallery rows (MG = 2: the throughput form; MG = 1: twice the workgroups
// for launches that would otherwise leave most CUs without a tile -- a few hundred queries, where the time is the per-CU operand intake)
template <int NQ, int MG>
__global__ __launch_bounds__(512, 1) void match_big_kernel(MatchArgs a) {
    constexpr int TQ = 64 * NQ;
    constexpr int TG = 128 * MG;
    constexpr int GP = 2 * MG;                          // gallery DMA pieces per wave and stage (TG / 8 rows per piece / 8 waves)
    constexpr int STAGE = (TG + TQ) * MB_ROWB;      // gallery rows, then query rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = wid >> 1, wq = wid & 1;
    const int lr = lane & 31, lh = lane >> 5;
    int tile_g, tile_q;
    {   // XCD-aware order, as in match_kernel: an eighth of the gallery tiles per XCD, all their query tiles
        int l = xcd_remap((int)blockIdx.x, (int)gridDim.x);
        int cls = 0, ng = (a.tiles_g + 7) >> 3;
        while (cls < 7 && l >= ng * a.tiles_q) {
            l -= ng * a.tiles_q;
            ++cls;
            ng = (a.tiles_g - cls + 7) >> 3;
        }
        tile_q = l / ng;
        tile_g = cls + 8 * (l - tile_q * ng);
    }
    const unsigned rowbytes = (unsigned)a.D * 2u;
    const __amdgpu_buffer_rsrc_t srd_g = __builtin_amdgcn_make_buffer_rsrc((void*)a.g, 0, (unsigned)((size_t)a.Gn * rowbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_q = __builtin_amdgcn_make_buffer_rsrc((void*)a.q, 0, (unsigned)((size_t)a.Qn * rowbytes), 0x00020000);

    // DMA piece = 8 rows x 128 B; lane = 8 rr + p fills physical chunk p of row 8 piece + rr with the row's logical chunk
    // p ^ ((row >> 1) & 7).  Wave w takes gallery pieces w, w + 8, w + 16, w + 24 and query pieces w + 8 j (j < NQ); rows beyond the
    // matrix lie beyond the descriptor's range and arrive as zeros.
    const int rr = lane >> 3, pch = lane & 7;
    unsigned vg[GP], vq[NQ];
#pragma unroll
    for (int j = 0; j < GP; ++j) {
        const int row = (wid + 8 * j) * 8 + rr;
        vg[j] = (unsigned)(tile_g * TG + row) * rowbytes + (unsigned)((pch ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int row = (wid + 8 * j) * 8 + rr;
        vq[j] = (unsigned)(tile_q * TQ + row) * rowbytes + (unsigned)((pch ^ ((row >> 1) & 7)) << 4);
    }
    // fragment addresses: A block mg = gallery rows 64 wg + 32 mg + lr, B block nq = query rows 32 NQ wq + 32 nq + lr; the 16-deep step kk
    // of a stage = logical chunks 2 kk + lh, physical chunk ^ ((row >> 1) & 7) = ^ ((lr >> 1) & 7) (the block bases are multiples of 16)
    const unsigned lds0 = (unsigned)(size_t)(lds_char_m*)smem;
    const unsigned sw = (unsigned)((lr >> 1) & 7);
    unsigned ca[4], cb[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const unsigned co = ((unsigned)(2 * kk + lh) ^ sw) << 4;
        ca[kk] = lds0 + (unsigned)((32 * MG * wg + lr) * MB_ROWB) + co;
        cb[kk] = lds0 + (unsigned)(TG * MB_ROWB + (32 * NQ * wq + lr) * MB_ROWB) + co;
    }

    f32x16 acc[MG][NQ];
#pragma unroll
    for (int i = 0; i < MG; ++i)
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 fa[2][MG], fb[2][NQ];            // two fragment sets: step s in set s & 1

#define MB_READ(SET, KK, SOFF)                                                                                              \
    {                                                                                                                       \
        const unsigned a_ = ca[KK] + (SOFF), b_ = cb[KK] + (SOFF);                                                          \
        _Pragma("unroll") for (int i = 0; i < MG; ++i)                                                                       \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[SET][i]) : "v"(a_), "n"(i * 32 * MB_ROWB));              \
        _Pragma("unroll") for (int j = 0; j < NQ; ++j)                                                                      \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET][j]) : "v"(b_), "n"(j * 32 * MB_ROWB));              \
    }
#define MB_MFMA(SET)                                                                                                        \
    {                                                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                                      \
        _Pragma("unroll") for (int i = 0; i < MG; ++i)                                                                       \
            _Pragma("unroll") for (int j = 0; j < NQ; ++j)                                                                  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);            \
        __builtin_amdgcn_s_setprio(0);                                                                                      \
    }
    // the wait, then every fragment register of the set tied to it (the compiler must not move the MFMAs that read them above the wait)
#define MB_WAIT(SET)                                                                                                        \
    {                                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                  \
        _Pragma("unroll") for (int i = 0; i < MG; ++i) asm volatile("" : "+v"(fa[SET][i]));                                 \
        _Pragma("unroll") for (int j = 0; j < NQ; ++j) asm volatile("" : "+v"(fb[SET][j]));                                 \
    }

    // one DMA piece of stage KT: IDX 0..GP-1 = this wave's gallery pieces, GP.. = its query pieces
    constexpr int NP = GP + NQ;
    auto piece = [&](int kt, int buf, int idx) {
        unsigned char* dst = smem + buf * STAGE;
        const int koff = kt * MB_ROWB;
        if (idx < GP) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_g, (lds_void_m*)(dst + (wid + 8 * idx) * 1024), 16, (int)vg[idx], koff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_q, (lds_void_m*)(dst + TG * MB_ROWB + (wid + 8 * (idx - GP)) * 1024), 16, (int)vq[idx - GP], koff, 0, 0);
    };
    // the MFMAs of one step with DMA pieces FIRST .. FIRST + COUNT - 1 of stage KT issued between them, one behind each MFMA: issuing a piece
    // takes the wave ~60-100 clocks (MI355X_MICROARCH.md), which the matrix pipe spends on the MFMA just queued -- all nine pieces in a row
    // right behind the barrier left the pipe idle for as long, in every wave at once
#define MB_MFMA_DMA(SET, KT, BUF, FIRST, COUNT, MORE)                                                                       \
    {                                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < MG; ++i)                                                                       \
            _Pragma("unroll") for (int j = 0; j < NQ; ++j) {                                                                \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);            \
                __builtin_amdgcn_sched_barrier(0);                                                                          \
                if (i * NQ + j < (COUNT) && (MORE)) piece(KT, BUF, (FIRST) + i * NQ + j);   /* (a scalar branch around one instruction) */ \
                __builtin_amdgcn_sched_barrier(0);                                                                          \
            }                                                                                                               \
    }
    constexpr int PA = (NP + 1) / 2;       // pieces issued among the deferred MFMAs; the other NP - PA among step 0's
    static_assert(PA <= MG * NQ && NP - PA <= MG * NQ, "one piece per MFMA");

    const int nk = a.D >> 6;
    // two stages are in flight from the start (both buffers are free)
#pragma unroll
    for (int i = 0; i < NP; ++i) piece(0, 0, i);
    if (nk > 1) {
#pragma unroll
        for (int i = 0; i < NP; ++i) piece(1, 1, i);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");      // stage 0 has landed for this wave; stage 1's pieces stay in flight
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    MB_READ(0, 0, 0u)
    MB_WAIT(0)
    MB_READ(1, 1, 0u)
    __builtin_amdgcn_sched_barrier(0);
    MB_MFMA(0)
    __builtin_amdgcn_sched_barrier(0);
    MB_WAIT(1)
    MB_READ(0, 2, 0u)
    __builtin_amdgcn_sched_barrier(0);
    MB_MFMA(1)
    __builtin_amdgcn_sched_barrier(0);
    MB_WAIT(0)
    MB_READ(1, 3, 0u)
    __builtin_amdgcn_sched_barrier(0);
    MB_MFMA(0)
    __builtin_amdgcn_sched_barrier(0);
    MB_WAIT(1)
    for (int kt = 1; kt < nk; ++kt) {
        const unsigned soff = (unsigned)((kt & 1) * STAGE);
        const int nb = (kt + 1) & 1;
        // stage kt has landed for this wave (its only loads in flight), and -- behind the barrier -- for every wave; every wave has
        // also finished its fragment reads of stage kt - 1 (lgkmcnt(0) above), whose buffer the next DMA overwrites
#if !(MB_ABL & 1)           /* timing-only ablations (tools/dev/build_variant.sh match.hip abl -DMB_ABL=..): 1 no wait for the stage's DMA, 2 no barrier */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if !(MB_ABL & 2)
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        MB_READ(0, 0, soff)
        __builtin_amdgcn_sched_barrier(0);
        // step 3 of stage kt - 1 (its fragments are in registers): covers the barrier skew, the first reads' latency and half of the
        // next stage's DMA issue
        const bool more = kt + 1 < nk;
        MB_MFMA_DMA(1, kt + 1, nb, 0, PA, more)
        __builtin_amdgcn_sched_barrier(0);
        MB_WAIT(0)
        MB_READ(1, 1, soff)
        __builtin_amdgcn_sched_barrier(0);
        MB_MFMA_DMA(0, kt + 1, nb, PA, NP - PA, more)
        __builtin_amdgcn_sched_barrier(0);
        MB_WAIT(1)
        MB_READ(0, 2, soff)
        __builtin_amdgcn_sched_barrier(0);
        MB_MFMA(1)
        __builtin_amdgcn_sched_barrier(0);
        MB_WAIT(0)
        MB_READ(1, 3, soff)
        __builtin_amdgcn_sched_barrier(0);
        MB_MFMA(0)
        __builtin_amdgcn_sched_barrier(0);
        MB_WAIT(1)                                       // step 3's fragments are in registers before the next barrier
    }
    MB_MFMA(1)
#undef MB_MFMA_DMA
#undef MB_READ
#undef MB_MFMA
#undef MB_WAIT

    // ---- epilogue: per-query k smallest (distance, row) of this tile's 256 gallery rows --------------------------------------------
    __builtin_amdgcn_s_barrier();                                   // every wave is past its last fragment read: the staging LDS is free
    float* s_rg = reinterpret_cast<float*>(smem);                   // [256] 1 / |g| (NaN for rows past the gallery: never selected)
    float* cd = reinterpret_cast<float*>(smem + 1024);              // [4][TQ] candidates of the four gallery waves
    int* cix = reinterpret_cast<int*>(smem + 1024 + 4 * TQ * 4);
    float* wd = reinterpret_cast<float*>(smem + 1024 + 8 * TQ * 4); // [TQ] the round's winner (k > 1: bounds the next round)
    int* wix = reinterpret_cast<int*>(smem + 1024 + 9 * TQ * 4);
    if (tid < TG) {
        const int gg = tile_g * TG + tid;
        s_rg[tid] = (gg < a.Gn) ? match_norm_term<false>(a.gn[gg]) : __builtin_nanf("");
    }
    __syncthreads();
    float qnv[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        const int qg = tile_q * TQ + 32 * NQ * wq + 32 * nq + lr;
        qnv[nq] = match_norm_term<false>((qg < a.Qn) ? a.qn[qg] : 1.f);
    }
    const int g0 = tile_g * TG + 32 * MG * wg + 4 * lh;               // gallery row of accumulator element (mg, e): g0 + 32 mg + (e & 3) + 8 (e >> 2)
    float pd[NQ];
    int pi[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) { pd[nq] = -INFINITY; pi[nq] = -1; }
    for (int r = 0; r < a.k; ++r) {
        float bd[NQ];
        int bi[NQ];
        if (a.k == 1) {
            // k = 1 (production.py's classify, the bench): one pass, five VALU instructions per pair (multiply, fused multiply-add,
            // compare, two selects).  The candidate starts as (+inf, the lane's first row): a NaN distance (row past the gallery: NaN
            // norm; non-finite embedding) compares false and is never taken, an all-NaN query keeps (+inf, first row) -- after the
            // lexicographic combines below: the tile's first row, what the general scan and the 128-row kernel return for it
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq) { bd[nq] = INFINITY; bi[nq] = g0; }
#pragma unroll
            for (int mg = 0; mg < MG; ++mg)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 gn4 = *reinterpret_cast<const f32x4*>(s_rg + 32 * MG * wg + 32 * mg + 8 * r4 + 4 * lh);
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int gi = g0 + 32 * mg + e4 + 8 * r4;
#pragma unroll
                        for (int nq = 0; nq < NQ; ++nq) {
                            const float d = match_distance<false>(acc[mg][nq][4 * r4 + e4], qnv[nq], gn4[e4]);
                            const bool take = d < bd[nq];
                            bd[nq] = take ? d : bd[nq];
                            bi[nq] = take ? gi : bi[nq];
                        }
                    }
                }
        } else {
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) { bd[nq] = INFINITY; bi[nq] = 0x7FFFFFFF; }
        // (gallery rows in ascending order: mg, then e -- a lane keeps the lowest row among equal distances by strict comparison)
#pragma unroll
        for (int mg = 0; mg < MG; ++mg)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 gn4 = *reinterpret_cast<const f32x4*>(s_rg + 32 * MG * wg + 32 * mg + 8 * r4 + 4 * lh);   // this lane's four rows' norms
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int gi = g0 + 32 * mg + e4 + 8 * r4;
                    // rows past the gallery (NaN norm) are never candidates; a NaN distance of a real row (non-finite embedding) sorts last
                    const bool real = gn4[e4] == gn4[e4];
#pragma unroll
                    for (int nq = 0; nq < NQ; ++nq) {
                        float d = match_distance<false>(acc[mg][nq][4 * r4 + e4], qnv[nq], gn4[e4]);
                        d = (d == d) ? d : INFINITY;
                        // branch-free (bitwise, not short-circuit): after the previous pick AND before the best so far.  gi ascends along the
                        // scan, so "before the best" is d < best alone
                        const bool after = (d > pd[nq]) | ((d == pd[nq]) & (gi > pi[nq]));
                        // (equal distances keep the lower row = the earlier one of this ascending scan; an all-infinite query -- non-finite
                        // embedding -- must still pick its first admissible row: `bi` untouched means nothing was taken yet)
                        const bool take = real & after & ((d < bd[nq]) | (bi[nq] == 0x7FFFFFFF));
                        bd[nq] = take ? d : bd[nq];
                        bi[nq] = take ? gi : bi[nq];
                    }
                }
            }
        }
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) {
            // the other half of the wave holds the other 32 gallery rows of the same query
            const float od = __shfl_xor(bd[nq], 32);
            const int oi = __shfl_xor(bi[nq], 32);
            if (lex_lt(od, oi, bd[nq], bi[nq])) { bd[nq] = od; bi[nq] = oi; }
            if (lh == 0) {
                cd[wg * TQ + 32 * NQ * wq + 32 * nq + lr] = bd[nq];
                cix[wg * TQ + 32 * NQ * wq + 32 * nq + lr] = bi[nq];
            }
        }
        __syncthreads();
        if (tid < TQ) {
            float bd = cd[tid];
            int bi = cix[tid];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float od = cd[w * TQ + tid];
                const int oi = cix[w * TQ + tid];
                if (lex_lt(od, oi, bd, bi)) { bd = od; bi = oi; }
            }
            wd[tid] = bd;
            wix[tid] = bi;
            const int qg = tile_q * TQ + tid;
            if (qg < a.Qn) {
                if (a.keys) {
                    // (the RETURNED old value makes the atomic's completion visible to this thread's vmcnt: it is performed -- at the
                    // device's coherence point -- before the ticket below is drawn)
                    const unsigned long long old = __hip_atomic_fetch_min(a.keys + qg, match_key(bd, bi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("" ::"v"(old));
                } else {
                    const size_t o = ((size_t)qg * a.tiles_g + tile_g) * a.k + r;
                    a.part_d[o] = bd;
                    a.part_i[o] = bi;
                }
            }
        }
        if (a.keys) {
            // ticket: the workgroup that finishes last has every tile's minima in `keys`; it writes the (Qn, 1) results and puts the
            // state block back (keys all-ones, ticket 0) for the next launch on this state
            int* s_last = reinterpret_cast<int*>(smem + 1024 + 10 * TQ * 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) *s_last = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*s_last == (int)gridDim.x - 1) {
                for (int q = tid; q < a.Qn; q += 512) {
                    const unsigned long long key = __hip_atomic_load(a.keys + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a.out_idx[q] = (long long)(unsigned)(key & 0xFFFFFFFFull);
                    if (a.out_dist) a.out_dist[q] = match_key_distance(key);
                    __hip_atomic_store(a.keys + q, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (tid == 0) __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        if (r + 1 < a.k) {
            __syncthreads();
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq) {
                pd[nq] = wd[32 * NQ * wq + 32 * nq + lr];
                pi[nq] = wix[32 * NQ * wq + 32 * nq + lr];
            }
            __syncthreads();
        }
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

// which bf16 core a launch takes: 0 = by the cost model below, 1 = always the 128-row kernel, 2 = always the 256-row LDS-DMA kernel;
// g_match_nq = 2..5 pins the 256-row kernel's query-tile width (0 = by the model).  Test / measurement switch: results never depend on it.
static int g_match_core = []() { const char* e = getenv("CVPCE_MATCH_CORE"); return !e ? 0 : (e[0] == 's' ? 1 : (e[0] == 'b' ? 2 : 0)); }();
// the one-launch form of cvpce_match_topk_state is OPT-IN: measured (tools/dev/bench_match.py) it loses to GEMM launch + merge launch wherever
// the model picks the fastest tile -- 200 x 10 000 x 512: 10.4 us against 9.6 -- because every workgroup then ends with two dependent
// device-scope atomic round trips (its keys, then the ticket) where the two-launch form ends with plain stores
static int g_match_fused = []() { const char* e = getenv("CVPCE_MATCH_FUSED"); return e ? atoi(e) : 0; }();
static int g_match_mg = []() { const char* e = getenv("CVPCE_MATCH_MG"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2) ? v : 0; }();
static int g_match_nq = []() { const char* e = getenv("CVPCE_MATCH_NQ"); const int v = e ? atoi(e) : 0; return (v >= 2 && v <= 5) ? v : 0; }();
extern "C" int cvpce_match_set_core(int core, int nq, int mg, int one_launch) {
    if (core < 0 || core > 2 || !(nq == 0 || (nq >= 2 && nq <= 5)) || mg < 0 || mg > 2 || (mg == 1 && nq == 5)) return CVPCE_ERR_ARG;
    g_match_core = core;
    g_match_nq = nq;
    g_match_mg = mg;
    g_match_fused = one_launch != 0;
    return CVPCE_OK;
}

// ---- state block of the one-launch top-1 form: `max_queries` 64-bit keys (all-ones = no candidate yet) + the ticket counter -----------
__global__ void match_state_init_kernel(unsigned long long* keys, int n_keys, int* ticket) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_keys) keys[i] = ~0ull;
    if (i == 0) *ticket = 0;
}
extern "C" size_t cvpce_match_state_bytes(int max_queries) { return max_queries <= 0 ? 0 : (size_t)max_queries * 8 + 64; }
extern "C" int cvpce_match_state_init(void* state, size_t state_bytes, void* stream) {
    if (!state || state_bytes < 64 + 8 || ((size_t)state & 7)) return CVPCE_ERR_ARG;
    const int n = (int)((state_bytes - 64) / 8);
    hipLaunchKernelGGL(match_state_init_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)state, n,
                       (int*)((char*)state + (size_t)n * 8));
    return cvpce_check_launch();
}

extern "C" size_t cvpce_match_workspace_bytes(int Qn, int Gn, int k) {
    const size_t tiles_g = (Gn + MT_TG - 1) / MT_TG;
    return (size_t)Qn * tiles_g * k * 8 + 512;
}

static int match_topk_impl(const void* queries, const void* gallery, const float* q_norms, const float* g_norms,
                           int Qn, int Gn, int D, int k, int is_f32, void* workspace, size_t workspace_bytes,
                           void* state, size_t state_bytes, long long* out_idx, float* out_dist, void* stream);

extern "C" int cvpce_match_topk(const void* queries, const void* gallery, const float* q_norms, const float* g_norms,
                                int Qn, int Gn, int D, int k, int is_f32, void* workspace, size_t workspace_bytes,
                                long long* out_idx, float* out_dist, void* stream) {
    return match_topk_impl(queries, gallery, q_norms, g_norms, Qn, Gn, D, k, is_f32, workspace, workspace_bytes, nullptr, 0, out_idx, out_dist, stream);
}
extern "C" int cvpce_match_topk_state(const void* queries, const void* gallery, const float* q_norms, const float* g_norms,
                                      int Qn, int Gn, int D, int k, int is_f32, void* workspace, size_t workspace_bytes,
                                      void* state, size_t state_bytes, long long* out_idx, float* out_dist, void* stream) {
    if (state && (state_bytes < 64 + 8 || ((size_t)state & 7))) return CVPCE_ERR_ARG;
    return match_topk_impl(queries, gallery, q_norms, g_norms, Qn, Gn, D, k, is_f32, workspace, workspace_bytes, state, state_bytes, out_idx, out_dist, stream);
}

static int match_topk_impl(const void* queries, const void* gallery, const float* q_norms, const float* g_norms,
                           int Qn, int Gn, int D, int k, int is_f32, void* workspace, size_t workspace_bytes,
                           void* state, size_t state_bytes, long long* out_idx, float* out_dist, void* stream) {
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
    a.keys = nullptr; a.ticket = nullptr; a.out_idx = out_idx; a.out_dist = out_dist;
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
    // Which bf16 core and tile: the LDS-DMA kernel (match_big_kernel, 128 MG gallery rows x 64 NQ queries per workgroup) or the 128-row
    // register-staged one, by estimated time.  Model (microseconds; least-squares fit to tools/dev/bench_match.py on MI355X, 63 launches of
    // 200 ... 3 200 queries x 1 000 ... 10 000 rows x 512 | 1 024, rms error 8 %, picks the fastest variant or one within 1 % of it on all nine
    // shapes):  launch = fixed + rounds x (D / 512) x round, where a round's cost grows with the share u of the CUs that hold a tile in it
    // (L2 intake and clocks):
    //   LDS-DMA kernel:   fixed 3.9 + 0.48 NQ,  round (0.74 + 1.97 NQ) x (MG = 1 ? 0.575 : 1) x (1 + 0.75 u)
    //   128-row kernel:   fixed 10.4,           round 5.4 x (1 + 0.75 u) (64-query tiles, three per CU) | 9.0 x (1 + 0.75 u) (128-query, two per CU)
    // Every bf16 kernel forms identical distances, so the choice changes no result (cvpce_match_set_core / CVPCE_MATCH_CORE = small | big,
    // CVPCE_MATCH_NQ = 2..5, CVPCE_MATCH_MG = 1 | 2 force one: tests and A/B measurements).
    int big_nq = 0, big_mg = 2;
    if (!is_f32) {
        const int core = g_match_core, forced_nq = g_match_nq;
        const long long cus = g_cvpce_persistent_wgs;
        const double dk = (double)D / 512.0;
        const long long t128 = (long long)a.tiles_g * ((Qn + 127) / 128), t64 = (long long)a.tiles_g * ((Qn + 63) / 64);
        const long long r128 = (t128 + 2 * cus - 1) / (2 * cus), r64 = (t64 + 3 * cus - 1) / (3 * cus);
        const double e128 = 10.4 + (double)r128 * dk * 9.0 * (1.0 + 0.75 * (double)t128 / (double)(r128 * 2 * cus));
        const double e64 = 10.4 + (double)r64 * dk * 5.4 * (1.0 + 0.75 * (double)t64 / (double)(r64 * 3 * cus));
        const double best = e64 < e128 ? e64 : e128;
        double best_big = 1e30;
        for (int mg = 1; mg <= 2; ++mg)
            for (int nq = 2; nq <= 5; ++nq) {
                if ((forced_nq && nq != forced_nq) || (g_match_mg && mg != g_match_mg) || (mg == 1 && nq == 5)) continue;
                const long long t = (long long)((Gn + 128 * mg - 1) / (128 * mg)) * ((Qn + 64 * nq - 1) / (64 * nq)), rounds = (t + cus - 1) / cus;
                const double u = (double)t / (double)(rounds * cus);
                const double e = 3.9 + 0.48 * nq + (double)rounds * dk * (0.74 + 1.97 * nq) * (mg == 1 ? 0.575 : 1.0) * (1.0 + 0.75 * u);
                if (e < best_big) { best_big = e; big_nq = nq; big_mg = mg; }
            }
        // (a handful of queries against a small gallery is launch latency either way: the 128-row kernel keeps those)
        if (core == 1 || (core == 0 && (!(best_big < best) || (long long)Qn * Gn < 64 * 1024))) big_nq = 0;
    }
    a.part_d = (float*)workspace;
    a.part_i = (int*)((char*)workspace + ((size_t)Qn * a.tiles_g * k * 4 + 255) / 256 * 256);    // (laid out for the 128-row tiling: enough for either)
    hipStream_t s = (hipStream_t)stream;
    if (big_nq) {
        const int tg = 128 * big_mg;
        a.tiles_g = (Gn + tg - 1) / tg;
        a.tiles_q = (Qn + 64 * big_nq - 1) / (64 * big_nq);
        if ((unsigned long long)((unsigned long long)Qn + 64 * big_nq) * D * 2 >= (1ull << 32) || ((unsigned long long)Gn + tg) * D * 2 >= (1ull << 32)) return CVPCE_ERR_ARG;
        const dim3 gridb(a.tiles_g * a.tiles_q);
        // k = 1 with a state block that holds a key per query: ONE launch (no partials, no merge kernel)
        const int n_keys = state ? (int)((state_bytes - 64) / 8) : 0;
        const bool fused = state && k == 1 && Qn <= n_keys && g_match_fused;
        if (fused) {
            a.keys = (unsigned long long*)state;
            a.ticket = (int*)((char*)state + (size_t)n_keys * 8);
        }
        const size_t smemb = (size_t)2 * (tg + 64 * big_nq) * MB_ROWB;
#define MB_LAUNCH(NQ_, MG_)                                                                                                    \
        case NQ_ * 4 + MG_:                                                                                                    \
            if (!cvpce_smem_attr_done<match_big_kernel<NQ_, MG_>>((const void*)match_big_kernel<NQ_, MG_>, 2 * (128 * MG_ + 64 * NQ_) * MB_ROWB)) return CVPCE_ERR_LAUNCH; \
            hipLaunchKernelGGL((match_big_kernel<NQ_, MG_>), gridb, dim3(512), smemb, s, a);                                   \
            break;
        switch (big_nq * 4 + big_mg) {
            MB_LAUNCH(2, 2) MB_LAUNCH(3, 2) MB_LAUNCH(4, 2) MB_LAUNCH(5, 2) MB_LAUNCH(2, 1) MB_LAUNCH(3, 1) MB_LAUNCH(4, 1)
            default: return CVPCE_ERR_ARG;
        }
#undef MB_LAUNCH
        if (fused) return cvpce_check_launch();
    } else {
    a.tiles_q = (Qn + tq - 1) / tq;
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
    }
    hipLaunchKernelGGL(match_merge_kernel, dim3((Qn + 3) / 4), dim3(256), 0, s, a.part_d, a.part_i, Qn, a.tiles_g * k,
                       k, out_idx, out_dist);
    return cvpce_check_launch();
}
