// Fused VGG16 stem for the MAC-VGG embedder (torchvision vgg cfg 'D' features[0:5]):
//
//     conv3x3(3->64)+bias+ReLU -> conv3x3(64->64)+bias+ReLU -> MaxPool2d(2,2)
//
// as ONE persistent kernel.  Unfused, conv1_1 writes a 64-channel full-resolution tensor (8.4 MB per 256x256 crop, 13.4 GB
// per 1600-crop batch) that conv1_2 reads straight back -- a quarter of the embedder's time at a few % of its FLOPs.
// Here every workgroup keeps conv1_2's weights resident in LDS (72 KiB), walks 16x16 output tiles, and per tile
//   1. evaluates conv1_1 on the 18x18 halo patch straight from the 3-channel input (MFMA, K = 3 rows of 16 = (kw 0..3)
//      x (c 0..3), slot kw=3 / c=3 carry zero weights) into an LDS image [324 px][64 ch] (zero outside the image =
//      conv1_2's zero padding),
//   2. runs conv1_2's 9 taps x 4 K-steps of MFMAs out of LDS (no global traffic, no barrier inside the tap loop,
//      fragments two K-steps ahead in registers),
//   3. max-pools in registers (pixels are laid out in 2x2-quad order across lanes) and stores only the pooled 8x8x64 tile.
// A workgroup is TWO independent 4-wave teams that share the resident weights and own an LDS image + input staging
// buffer each.  One wave of each team sits on every SIMD, and the teams run out of phase: while one is in its MFMA-bound
// conv1_2 phase the other does the VALU/LDS-bound parts (conv1_1 + its epilogue, pooling, stores, input staging), which
// with one team per workgroup (the first version of this kernel) left the matrix pipe idle for 40 % of every tile
// (profiles/r01d_ablation_stem.md).  The teams never wait for each other: team barriers are LDS counters (ds_add + poll),
// s_barrier is used once, after the weights are resident.
// Biases live in LDS and seed the accumulators (max(x + b) = max(x) + b), which also frees 64 VGPRs.
// HBM traffic per crop: 0.5 MB in (NHWC4 bf16) + 2.1 MB out instead of 0.5 + 8.4 + 8.4 + 2.1 MB.
#include "common.h"
#include "../../include/cvpce_amd.h"

// compile-time timing experiments (never set in the shipped library; tools/ablate.sh): 1 no initial de-phasing,
// 2 raised priority in the conv1_2 phase, 4 NO raised priority outside it (the shipped default raises it: -8 %), 8 skip conv1_1, 16 skip the conv1_2 MFMAs, 32 no weight-fragment LDS reads after the first three K-steps, 64 no pixel-fragment reads, 128 in-kernel stamps (below)
#ifndef CVPCE_DBG
#define CVPCE_DBG 0
#endif

#define S2_T 16                    // output tile edge
#define S2_P1 (S2_T + 2)           // conv1_1 patch edge (halo 1)
#define S2_P0 (S2_T + 4)           // input patch edge (halo 2)
#define S2_NPIX1 (S2_P1 * S2_P1)   // 324
#define S2_NPT 11                  // MFMA pixel tiles of 32 covering 324
#define S2_W2_BYTES (9 * 64 * 128)
#define S2_A1_BYTES (S2_NPIX1 * 128)
#define S2_IN_BYTES (S2_P0 * S2_P0 * 8 + 64)   // + slack for the kw=3 over-read at the patch end
#define S2_BIAS_BYTES 512
#define S2_SMEM (S2_W2_BYTES + 2 * S2_A1_BYTES + 2 * S2_IN_BYTES + S2_BIAS_BYTES + 16)

struct Stem2Args {
    const bf16_t* in;    // [N][H][W][cstride] bf16, channels 0..2 used, channel 3 must be zero (cstride 4 or 8)
    int cstride;
    const bf16_t* w1;    // [64][48]  k = kh*16 + kw*4 + c
    const float* b1;     // [64]
    const bf16_t* w2;    // [9][64][64]  (tap, cout, cin)
    const float* b2;     // [64]
    bf16_t* out;         // [N][H/2][W/2][64]
    int N, H, W;
    int tiles_x, tiles_y, ntiles;
    // LIST launches (the embedder's constant-padding tile skipping, skiplist.hip): the tiles to compute, low word of an entry =
    // (n << 16) | (ty << 8) | tx, *list_count entries; image N - 1 (the constant crop) is read from const_in, not from `in`
    const unsigned long long* list;
    const int* list_count;
    const bf16_t* const_in;
};

// byte offset of 16-B chunk `chunk` of conv1_1-output patch pixel (py, px): the swizzle of conv3x3_halo2.hip, which makes
// the quad-ordered conv1_2 fragment reads (ds_read_b128 in non-contiguous 16-lane groups) conflict-free for every tap
__device__ __forceinline__ int s2_swz0(int u) { return ((u & 3) << 1) | ((u >> 2) & 1); }
__device__ __forceinline__ int s2_a1_off(int py, int px, int chunk) {
    return (py * S2_P1 + px) * 128 + ((chunk ^ s2_swz0(px >> 1) ^ (py & 1)) << 4);
}

typedef __attribute__((address_space(3))) char lds_char;

#if (CVPCE_DBG & 128)
// diagnostic build only (tools/ablate.sh vgg_stem2 128; tools/dev/stem_stamps.py): s_memtime stamps of one wave of each team of
// workgroup 0 at the phase boundaries of its first 16 tiles.  No output value depends on them.
__device__ unsigned long long cvpce_stem_stamps[2][16][6];
extern "C" int cvpce_debug_stem_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cvpce_stem_stamps), sizeof(cvpce_stem_stamps)) == hipSuccess ? 0 : 2;
}
#define S2_STAMP(I) if (blockIdx.x == 0 && wid == 0 && lane == 0 && it_ < 16) cvpce_stem_stamps[team][it_][I] = __builtin_amdgcn_s_memtime();
#else
#define S2_STAMP(I)
#endif

// barrier among the 4 waves of a team: arrive = ds_add on the team's LDS counter (after this wave's LDS traffic has
// drained), wait = poll until all 4 arrivals of this round are in.  `target` counts arrivals expected so far.
__device__ __forceinline__ void team_barrier(unsigned cnt_addr, unsigned& target, int lane) {
    target += 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(cnt_addr), "v"(1u) : "memory");
    for (;;) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cnt_addr) : "memory");
        if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)v) - target) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

template <bool LIST>
__global__ __launch_bounds__(512, 2) void vgg_stem2_kernel(Stem2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* W2 = smem;
    unsigned char* A1all = W2 + S2_W2_BYTES;
    unsigned char* INall = A1all + 2 * S2_A1_BYTES;
    float* BL = reinterpret_cast<float*>(INall + 2 * S2_IN_BYTES);     // [0..63] conv1_1 bias, [64..127] conv1_2 bias
    unsigned* CNT = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(BL) + S2_BIAS_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wid8 >> 2, wid = wid8 & 3, qtid = tid & 255;
    const int lr = lane & 31, lh = lane >> 5;

    // ---- resident weights -> LDS (W2 rows of 128 B, XOR-swizzled like the GEMM tiles), biases, zeroed slack ----
    for (int i = tid; i < 9 * 64 * 8; i += 512) {
        const int row = i >> 3, ch = i & 7;           // row = tap*64 + cout
        const u32x4 v = *reinterpret_cast<const u32x4*>(a.w2 + (size_t)row * 64 + ch * 8);
        *reinterpret_cast<u32x4*>(W2 + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = v;
    }
    if (tid < 64) { BL[tid] = a.b1[tid]; BL[64 + tid] = a.b2[tid]; }
    if (tid < 32) *reinterpret_cast<unsigned*>(INall + (tid >> 4) * S2_IN_BYTES + S2_P0 * S2_P0 * 8 + (tid & 15) * 4) = 0u;
    if (tid < 2) CNT[tid] = 0u;
    __syncthreads();

    unsigned char* A1 = A1all + team * S2_A1_BYTES;
    unsigned char* IN = INall + team * S2_IN_BYTES;
    const unsigned cnt_addr = (unsigned)(size_t)(lds_char*)(CNT + team);
    unsigned bar_target = 0;

    // input patch staging: 400 pixels of 8 B, two per thread of the team (second one only for qtid < 144)
    unsigned long long preg[2];
    // work item -> (image, tile row, tile column): a list entry, or the flat tile index itself
    auto decode = [&](unsigned e, int& n, int& ty, int& tx) {
        if constexpr (LIST) {
            n = (int)(e >> 16);
            ty = (int)((e >> 8) & 0xFF);
            tx = (int)(e & 0xFF);
        } else {
            n = (int)e / (a.tiles_x * a.tiles_y);
            const int r = (int)e - n * (a.tiles_x * a.tiles_y);
            ty = r / a.tiles_x;
            tx = r - ty * a.tiles_x;
        }
    };
    // LIST: `ext` = (ey << 12) | ex, the crop's content extent from the entry's high word -- a pixel with y >= ey or x >= ex is the pad
    // constant and is read from the constant crop (the crop kernel may not have written it: cvpce_crop_resize_content)
    auto load_patch = [&](unsigned item, unsigned ext) {
        int n, ty, tx;
        decode(item, n, ty, tx);
        const bf16_t* img = a.in + (size_t)n * a.H * a.W * a.cstride;
        int ey = a.H, ex = a.W;
        if constexpr (LIST) {
            if (n == a.N - 1) img = a.const_in;
            ey = (int)((ext >> 12) & 0xFFFu);
            ex = (int)(ext & 0xFFFu);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = qtid + k * 256;
            unsigned long long v = 0ull;
            if (p < S2_P0 * S2_P0) {
                const int py = p / S2_P0, px = p - py * S2_P0;
                const int y = ty * S2_T - 2 + py, x = tx * S2_T - 2 + px;
                const bf16_t* src = img;
                if constexpr (LIST) src = (y >= ey || x >= ex) ? a.const_in : img;
                if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W)
                    v = *reinterpret_cast<const unsigned long long*>(src + ((size_t)y * a.W + x) * a.cstride);
            }
            preg[k] = v;
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = qtid + k * 256;
            if (p < S2_P0 * S2_P0) *reinterpret_cast<unsigned long long*>(IN + p * 8) = preg[k];
        }
    };

    const int stride = 2 * (int)gridDim.x;
    int tile = 2 * (int)blockIdx.x + team;
    int ntiles = a.ntiles;
    if constexpr (LIST) ntiles = __builtin_amdgcn_readfirstlane(*a.list_count);
    if (tile >= ntiles) return;                       // (team-uniform; the other team does not wait for this one)
    // the work items of this and the next two iterations (LIST: list entries, fetched two iterations ahead of their use)
    auto item_at = [&](int t) -> unsigned {
        if constexpr (LIST) return t < ntiles ? (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a.list[t]) : 0u;
        return (unsigned)t;
    };
    auto ext_at = [&](int t) -> unsigned {
        if constexpr (LIST) return t < ntiles ? (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a.list[t] >> 32)) : 0u;
        return 0u;
    };
    unsigned it_cur = item_at(tile), it_next = item_at(tile + stride);
    unsigned ex_next = ext_at(tile + stride);
    load_patch(it_cur, ext_at(tile));
    store_patch();
    team_barrier(cnt_addr, bar_target, lane);

    // per-lane constants of the conv1_2 phase: wave w owns output rows 4w..4w+3 (pixel tiles 2w, 2w+1)
    // lane r of a 32-pixel tile: quad q = r>>2 -> columns 2q,2q+1 ; sub = r&3 -> row +(sub>>1), col +(sub&1)
    const int q = lr >> 2, sub = lr & 3;
    // fragment address of (tap (kh,kw), K-step kk) = (rd2[nt][kw] ^ (((2 kk) ^ (kh & 1)) << 4)) + (kh*18 + kw) * 128:
    // the per-lane part of the swizzle (K half lh, row parity, column pair of ox + kw) is folded into rd2 once
    unsigned rd2[2][3];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int oy = 2 * (2 * wid + nt) + (sub >> 1), ox = 2 * q + (sub & 1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            rd2[nt][kw] = (unsigned)((oy * S2_P1 + ox) * 128 + ((lh ^ (oy & 1) ^ s2_swz0(((ox + kw) >> 1) & 7)) << 4));
    }

    // conv1_1's A operand (64 couts x 48 k) is tiny: keep this lane's 6 fragments in registers for the whole kernel
    bf16x8 w1frag[3][2];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
            w1frag[kh][ct] = *reinterpret_cast<const bf16x8*>(a.w1 + (ct * 32 + lr) * 48 + kh * 16 + lh * 8);

    // de-phase the teams once: team 1 starts half a tile late, so that its conv1_1 / epilogue falls into team 0's
    // conv1_2 phase and vice versa (they then keep each other out of phase: whoever shares the matrix pipe slows down)
    if (team == 1 && !(CVPCE_DBG & 1)) {
#pragma unroll 1
        for (int i = 0; i < 6; ++i) __builtin_amdgcn_s_sleep(127);
    }

    int it_ = -1;
    for (; tile < ntiles; tile += stride) {
        ++it_;
        S2_STAMP(0)
        int n, ty, tx;
        decode(it_cur, n, ty, tx);
        const int next = tile + stride;
        if (next < ntiles) load_patch(it_next, ex_next);   // global loads in flight under the conv1_1 phase
        it_cur = it_next;
        it_next = item_at(next + stride);
        ex_next = ext_at(next + stride);
        const bool border = ty == 0 || tx == 0 || ty == a.tiles_y - 1 || tx == a.tiles_x - 1;

        // ================= phase 1: conv1_1 on the 18x18 patch -> A1 =================
        if (!(CVPCE_DBG & 4)) __builtin_amdgcn_s_setprio(2);
        for (int pt = wid; pt < ((CVPCE_DBG & 8) ? 0 : S2_NPT); pt += 4) {
            int pp = pt * 32 + lr;
            const bool real = pp < S2_NPIX1;
            if (!real) pp = S2_NPIX1 - 1;
            const int py = pp / S2_P1, px = pp - py * S2_P1;
            f32x16 acc[2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(BL + c * 32 + 8 * g + 4 * lh);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[c][4 * g + j] = b[j];
                }
            // B fragments: k = kh*16 + 8h + j  <->  pixels (py+kh, px+2h .. px+2h+1), 4 channels each: 16 contiguous bytes.
            union { unsigned long long u[2]; bf16x8 v; } bfrag[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const unsigned char* src = IN + ((py + kh) * S2_P0 + px + 2 * lh) * 8;
                bfrag[kh].u[0] = *reinterpret_cast<const unsigned long long*>(src);
                bfrag[kh].u[1] = *reinterpret_cast<const unsigned long long*>(src + 8);
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1frag[kh][ct], bfrag[kh].v, acc[ct], 0, 0, 0);
            // epilogue: ReLU, zero outside the image (conv1_2 pads conv1_1's OUTPUT with zeros: a mask on the packed
            // pairs, all ones except on the tiles that touch the image border)
            const int y = ty * S2_T - 1 + py, x = tx * S2_T - 1 + px;
            const unsigned keep = ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) ? 0xFFFFFFFFu : 0u;
#ifndef S2_WIDE_STORE
            // eight 8-byte stores per lane and pixel tile (two-way bank conflicts: the 32 pixels of a half wave fall on 16 bank positions).  The
            // conflict-free form below (-DS2_WIDE_STORE) was built and measured in round 5: 1-4 % SLOWER -- the stores are not what this phase
            // waits on, the four extra v_permlane32_swap pairs are (profiles/r05_rejected_experiments.md)
            if (real) {
                const int abase = (py * S2_P1 + px) * 128 + lh * 8;
                const int aswz = (s2_swz0(px >> 1) ^ (py & 1)) << 4;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 r;
#pragma unroll
                        for (int j = 0; j < 4; ++j) r[j] = relu_bits(acc[ct][4 * g + j]);
                        uint2 u = __builtin_bit_cast(uint2, f32x4_to_bf16x4(r));
                        if (border) { u.x &= keep; u.y &= keep; }
                        *reinterpret_cast<uint2*>(A1 + abase + (((ct * 4 + g) << 4) ^ aswz)) = u;
                    }
            }
#else
            // (-DS2_WIDE_STORE, measured and not adopted) A lane holds 4 channels (8 bytes) of each 16-byte chunk, its partner lane (l ^ 32) the
            // other 4: one v_permlane32_swap pair per chunk pair hands lane lh the WHOLE chunk 2m + lh -- 4 stores of 16 bytes, 16 lanes of a
            // quarter wave on 16 different 16-byte bank groups (conflict-free).
            const int abase = (py * S2_P1 + px) * 128;
            const int aswz = (s2_swz0(px >> 1) ^ (py & 1)) << 4;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x4 ra, rb;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ra[j] = relu_bits(acc[ct][8 * m + j]); rb[j] = relu_bits(acc[ct][8 * m + 4 + j]); }
                    uint2 ua = __builtin_bit_cast(uint2, f32x4_to_bf16x4(ra)), ub = __builtin_bit_cast(uint2, f32x4_to_bf16x4(rb));
                    if (border) { ua.x &= keep; ua.y &= keep; ub.x &= keep; ub.y &= keep; }
                    // channels co = ct*32 + 8g + 4lh .. +3 = bytes lh*8 .. +7 of chunk ct*4 + g; g = 2m (ua), 2m + 1 (ub)
                    const auto s0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
                    if (real) *reinterpret_cast<u32x4*>(A1 + abase + (((ct * 4 + 2 * m + lh) << 4) ^ aswz)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
#endif
        }
        S2_STAMP(1)
        team_barrier(cnt_addr, bar_target, lane);
        S2_STAMP(2)
        if (next < ntiles) store_patch();               // IN is free again: stage the next tile's input

        // ================= phase 2: conv1_2 (9 taps x 4 K-steps) out of LDS =================
        f32x16 acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(BL + 64 + mt * 32 + 8 * g + 4 * lh);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[mt][0][4 * g + j] = b[j]; acc[mt][1][4 * g + j] = b[j]; }
            }
#ifndef S2_DEPTH
#define S2_DEPTH 2            // K-steps the fragment reads run ahead of the MFMAs (S2_DEPTH + 1 register slots)
#endif
        bf16x8 af[S2_DEPTH + 1][2], bfr[S2_DEPTH + 1][2];
#define S2_LOAD(S, SLOT)                                                                                       \
        {                                                                                                      \
            const int tap = (S) >> 2, kk = (S) & 3;                                                            \
            const int kh = tap / 3, kw = tap - kh * 3;                                                         \
            const int chunk = kk * 2 + lh;                                                                     \
            if (!(CVPCE_DBG & 32) || (S) < 3) {                                                                \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) {                                                 \
                const int row = tap * 64 + mt * 32 + lr;                                                       \
                af[SLOT][mt] = *reinterpret_cast<const bf16x8*>(W2 + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)); \
            }                                                                                                  \
            }                                                                                                  \
            if (!(CVPCE_DBG & 64) || (S) < 3)                                                                  \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                   \
                bfr[SLOT][nt] = *reinterpret_cast<const bf16x8*>(A1 + ((rd2[nt][kw] ^ (unsigned)(((2 * kk) ^ (kh & 1)) << 4)) + (unsigned)((kh * S2_P1 + kw) * 128))); \
        }
        // fragments run TWO K-steps ahead of the MFMAs (3 register slots); sched_barrier pins that order
        if (!(CVPCE_DBG & 4)) __builtin_amdgcn_s_setprio(0);
        if (CVPCE_DBG & 2) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int s = 0; s < S2_DEPTH; ++s) S2_LOAD(s, s)
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + S2_DEPTH < 36) S2_LOAD(s + S2_DEPTH, (s + S2_DEPTH) % (S2_DEPTH + 1))
            __builtin_amdgcn_sched_barrier(0);
            if (!(CVPCE_DBG & 16))
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s % (S2_DEPTH + 1)][mt], bfr[s % (S2_DEPTH + 1)][nt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef S2_LOAD
        S2_STAMP(3)
        if (CVPCE_DBG & 2) __builtin_amdgcn_s_setprio(0);
        if (!(CVPCE_DBG & 4)) __builtin_amdgcn_s_setprio(2);
        // epilogue: 2x2 max over the quad's 4 lanes (DPP), ReLU.  After pooling the 4 lanes of a quad hold the
        // same 64 values; lane `sub` keeps channel group g = sub, then a v_permlane32_swap pair gives every lane
        // 8 consecutive channels: ONE 16-byte store per lane per pixel tile, all 64 lanes active.
        const int Ho = a.H >> 1, Wo = a.W >> 1;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            unsigned pk[2][2];     // [mt][dword]: this lane's 4 channels (8 sub + 4 lh ..+3) of cout tile mt, bf16x2 packed
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                bf16x4 sel;
#pragma unroll
                for (int j = 0; j < 4; ++j) sel[j] = (bf16_t)0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 r;
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[j] = quad_max_nonneg(relu_bits(acc[mt][nt][4 * g + j]));
                    const bf16x4 o = f32x4_to_bf16x4(r);
                    if (sub == g) sel = o;
                }
                const uint2 u = *reinterpret_cast<const uint2*>(&sel);
                pk[mt][0] = u.x; pk[mt][1] = u.y;
            }
            // lanes 0-31 (lh = 0) end with cout tile 0, channels 8 sub .. 8 sub + 7; lanes 32-63 with cout tile 1
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                auto r = __builtin_amdgcn_permlane32_swap(pk[0][d], pk[1][d], false, false);
                pk[0][d] = r[0]; pk[1][d] = r[1];
            }
            const int oyp = (ty * S2_T) / 2 + (2 * wid + nt), oxp = (tx * S2_T) / 2 + q;
            bf16_t* dst = a.out + ((size_t)(n * Ho + oyp) * Wo + oxp) * 64 + lh * 32 + sub * 8;
            *reinterpret_cast<u32x4*>(dst) = u32x4{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
        }
        S2_STAMP(4)
        team_barrier(cnt_addr, bar_target, lane);        // A1 is free again; the next input patch is visible
        S2_STAMP(5)
    }
}

static int stem_launch(const void* in_nhwc, int in_cstride, const void* const_in, const void* w1, const float* b1, const void* w2,
                       const float* b2, void* out, int N, int H, int W, const unsigned long long* list, const int* count_dev,
                       void* stream) {
    if (N <= 0) return CVPCE_OK;
    if (!in_nhwc || !w1 || !b1 || !w2 || !b2 || !out) return CVPCE_ERR_ARG;
    if (list && (!count_dev || !const_in || N > 65535 || H / S2_T > 255 || W / S2_T > 255)) return CVPCE_ERR_ARG;
    if (in_cstride != 4 && in_cstride != 8) return CVPCE_ERR_ARG;
    if (H % S2_T != 0 || W % S2_T != 0 || H <= 0 || W <= 0) return CVPCE_ERR_ARG;
    if ((long long)N * H * W >= (1LL << 31) / 4) return CVPCE_ERR_ARG;      // in-kernel pixel indices are 32-bit, byte offsets 64-bit
    Stem2Args a;
    a.in = (const bf16_t*)in_nhwc; a.cstride = in_cstride; a.w1 = (const bf16_t*)w1; a.b1 = b1; a.w2 = (const bf16_t*)w2; a.b2 = b2;
    a.out = (bf16_t*)out; a.N = N; a.H = H; a.W = W;
    a.tiles_x = W / S2_T; a.tiles_y = H / S2_T; a.ntiles = N * a.tiles_x * a.tiles_y;
    a.list = list; a.list_count = count_dev; a.const_in = (const bf16_t*)const_in;
    const int pairs = (a.ntiles + 1) / 2;
    const int grid = pairs < g_cvpce_persistent_wgs ? pairs : g_cvpce_persistent_wgs;       // one persistent workgroup (two teams) per CU
    if (list) {
        if (!cvpce_smem_attr_done<vgg_stem2_kernel<true>>((const void*)vgg_stem2_kernel<true>, S2_SMEM)) return CVPCE_ERR_LAUNCH;
        hipLaunchKernelGGL(vgg_stem2_kernel<true>, dim3(grid), dim3(512), S2_SMEM, (hipStream_t)stream, a);
    } else {
        if (!cvpce_smem_attr_done<vgg_stem2_kernel<false>>((const void*)vgg_stem2_kernel<false>, S2_SMEM)) return CVPCE_ERR_LAUNCH;
        hipLaunchKernelGGL(vgg_stem2_kernel<false>, dim3(grid), dim3(512), S2_SMEM, (hipStream_t)stream, a);
    }
    return cvpce_check_launch();
}

extern "C" int cvpce_vgg_stem_fused(const void* in_nhwc, int in_cstride, const void* w1, const float* b1, const void* w2,
                                    const float* b2, void* out, int N, int H, int W, void* stream) {
    return stem_launch(in_nhwc, in_cstride, nullptr, w1, b1, w2, b2, out, N, H, W, nullptr, nullptr, stream);
}

extern "C" int cvpce_vgg_stem_fused_list(const void* in_nhwc, int in_cstride, const void* const_in, const void* w1, const float* b1,
                                         const void* w2, const float* b2, void* out, int N, int H, int W,
                                         const unsigned long long* list, const int* count_dev, void* stream) {
    if (!list || !count_dev || !const_in) return CVPCE_ERR_ARG;
    return stem_launch(in_nhwc, in_cstride, const_in, w1, b1, w2, b2, out, N, H, W, list, count_dev, stream);
}
