// 3x3 / stride 1 / pad 1 convolution, Cin % 64 == 0: halo patch in LDS, WEIGHTS STRAIGHT FROM L2 INTO REGISTERS,
// no workgroup barrier inside a 64-channel chunk (VGG16 conv2_2 ... conv5_3; second generation of conv3x3_halo.hip).
//
// What bounds the LDS-ring kernels (profiles/r01_ablation_conv4_2.md) is not MFMA, LDS bandwidth or latency but the
// per-K-step hand-off: every 32 k all 8 waves meet at a barrier and then issue their DMA pieces in lock step, so both
// waves of a SIMD stall together.  Here the workgroup (8 waves) still owns a 16x16-pixel x TC-cout tile, but
//   * wave w owns 32 couts x (256 / WP) pixels: the weights it needs are needed by no other wave (WP = 1), so it loads
//     its own A fragments with buffer_load_dwordx4 in MFMA layout (lane (m, q) <- 16 B of row m) one 3x3 tap = 64 k
//     = one full 128-B line per row at a time, two taps ahead, into registers -- no LDS ring, no hand-off;
//   * the 18x18x64 input patch of a channel chunk is triple-buffered in LDS (LDS-DMA, zero-filled borders); the only
//     barrier is once per chunk (18 K-steps), so the two waves of a SIMD drift apart and one computes while the
//     other issues loads;
//   * pixel fragments stream through an 8-deep register ring of ds_read_b128 with immediate offsets (all tap shifts
//     are compile-time), conflict-free by the h3 swizzle below.
// Per K-step and wave: 32 MFMA 16x16x32, 16 ds_read_b128 (LDS array 50 % busy), 2 buffer loads.
// K order / weight layout: chunk-major [Cout_pad][K_pad] of include/cvpce_amd.h.  Fused bias / ReLU / MaxPool2d(2,2).
#include "common.h"
#include "../../include/cvpce_amd.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

// compile-time timing experiments (never set in the shipped library; tools/ablate.sh): 1 no s_setprio around the MFMA
// groups, 2 XCD-aware tile order, 4 no patch DMA in the loop, 8 no weight loads in the loop, 16 no output stores,
// 32 no fragment reads in the loop
#ifndef CVPCE_DBG
#define CVPCE_DBG 0
#endif

#define G2_T 16
#define G2_P 18
#define G2_NPIX 324
#define G2_A_BYTES (328 * 128)
#define G2_NPIECE 41

// Patch swizzle (same as conv3x3_halo.hip): 16-byte chunk c of patch pixel (py, px) lives at physical chunk
// c ^ g2_swz(py, px).  ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, ...), i.e. a
// group mixes two K-quarters over complementary halves of the 16 pixel lanes; with the lane -> pixel maps below every
// tap's fragment read is conflict-free at any alignment.
__device__ __forceinline__ int g2_swz0(int u) { return ((u & 3) << 1) | ((u >> 2) & 1); }
__device__ __forceinline__ int g2_swz(int py, int px) { return g2_swz0(px >> 1) ^ (py & 1); }
__device__ __forceinline__ int g2_col(int l16) { return l16 < 4 ? 2 * l16 : (l16 >= 12 ? 2 * (l16 - 8) : 2 * (l16 - 4) + 1); }

struct Halo2Args {
    const bf16_t* in;    // [N][H][W][Cin]
    const bf16_t* wgt;   // [Cout_pad][K_pad], chunk-major K
    const float* bias;
    const unsigned char* mask;   // optional [H][W]: output pixels with mask 0 are stored as zeros (atlas gaps); not with POOL
    bf16_t* out;         // [N][H][W][Cout] or pooled [N][H/2][W/2][Cout]
    int N, H, W, Cin, Cout, K_pad, relu;
    int tiles_x, tiles_y, ptiles, ctiles, ntiles;
    unsigned in_bytes, wgt_bytes;
};

// byte offset (immediate) of pixel block nb of a wave's first block, tap (kh, kw)
template <bool POOL>
__device__ __forceinline__ constexpr int g2_imm(int nb, int kh, int kw) {
    return POOL ? ((2 * (nb >> 1) + kh) * G2_P + 8 * (nb & 1) + kw) * 128 : ((nb + kh) * G2_P + kw) * 128;
}

template <int WC, int WP, bool POOL>
__global__ __launch_bounds__(512, 2) void conv3x3_halo2_kernel(Halo2Args a) {
    constexpr int TC = 32 * WC;
    constexpr int NB = 16 / WP;               // 16-pixel blocks per wave
    constexpr int NG = NB / 4;                // groups of 4 blocks per K-step
    static_assert(WC * WP == 8 && (NB == 16 || NB == 8), "8 waves");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ap = smem;                 // [3][328][64] bf16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wid / WP, wp = wid % WP;
    const int l16 = lane & 15, lq = lane >> 4;

    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_p = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);

    const int nchunks = a.Cin >> 6;
    const int lbid0 = (CVPCE_DBG & 2) ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int my_tiles = (a.ntiles - lbid0 + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;
    const int total_chunks = my_tiles * nchunks;          // < 2^31: checked on the host

    // tile seq -> (image, tile row, tile column, cout tile); cout tile fastest
    const int lbid = (CVPCE_DBG & 2) ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    auto tile_of = [&](int seq, int& n, int& ty, int& tx, int& ct) {
        const int t = lbid + seq * (int)gridDim.x;
        ct = t % a.ctiles;
        const int p = t / a.ctiles;
        n = p / (a.tiles_x * a.tiles_y);
        const int r = p - n * (a.tiles_x * a.tiles_y);
        ty = r / a.tiles_x;
        tx = r - ty * a.tiles_x;
    };

    // ---- patch DMA: piece j fills patch rows 8j .. 8j+7 (row = lane>>3, phys chunk = lane&7); pieces dealt
    //      round-robin to the 8 waves (wave w: pieces w, w+8, ...; 6 for w = 0, else 5) ----
    const int npp = (wid == 0) ? 6 : 5;
    auto issue_patch = [&](int n, int ty, int tx, int c, int buf) {
        const int y0 = ty * G2_T - 1, x0 = tx * G2_T - 1;
        // lane id recomputed here (2 VALU ops, once per patch) instead of living in a VGPR across the K loop; the empty
        // asm also keeps the per-piece constants below from being hoisted out of the chunk loop (18+ VGPRs)
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i < npp) {
                const int j = wid + 8 * i;
                const int pp = j * 8 + (ln >> 3);
                const int py = pp / G2_P, px = pp - py * G2_P;
                const int lchunk = (ln & 7) ^ g2_swz(py, px);
                const int y = y0 + py, x = x0 + px;
                const bool ok = pp < G2_NPIX && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                const unsigned off = (unsigned)((((size_t)(n * a.H + y) * a.W + x) * a.Cin + c * 64) * 2) + (unsigned)(lchunk * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_p, (lds_void*)(Ap + buf * G2_A_BYTES + j * 1024), 16,
                                                         (int)(ok ? off : 0xFFFFFFF0u), 0, 0, 0);
            }
        }
    };
    // patch issue pointer: the next flat chunk to fetch and its tile
    int pi = 0, pi_seq = 0, pi_c = 0, pi_buf = 0, pi_n, pi_ty, pi_tx, pi_ct;
    tile_of(0, pi_n, pi_ty, pi_tx, pi_ct);
    auto issue_next_patch = [&]() {
        if (pi < total_chunks) {
            if (!(CVPCE_DBG & 4) || pi < 2) issue_patch(pi_n, pi_ty, pi_tx, pi_c, pi_buf);
            ++pi;
            if (++pi_buf == 3) pi_buf = 0;
            if (++pi_c == nchunks) {
                pi_c = 0;
                ++pi_seq;
                if (pi_seq < my_tiles) tile_of(pi_seq, pi_n, pi_ty, pi_tx, pi_ct);
            }
        }
    };

    // ---- weights: lane (m = l16, q = lq) loads 16 B = k 8q .. 8q+7 of row m of a 16-cout block; per tap two K-halves ----
    unsigned voff[2];
#pragma unroll
    // MFMA row m = 4q + j of block mt is cout 8q + 4mt + j of the wave's 32: after both blocks accumulator lane group q
    // holds 8 CONSECUTIVE couts (8q .. 8q+7) of its pixel -> one 16-byte store per pixel block instead of two 8-byte ones
    for (int mt = 0; mt < 2; ++mt) voff[mt] = (unsigned)(((wc * 32 + 8 * (l16 >> 2) + 4 * mt + (l16 & 3)) * a.K_pad + lq * 8) * 2);
    // scalar byte offset of (cout tile ct, channel chunk c): ct*TC rows down, c*576 k along
    auto wbase = [&](int ct, int c) { return __builtin_amdgcn_readfirstlane((int)(((unsigned)(ct * TC) * (unsigned)a.K_pad + (unsigned)c * 576u) * 2u)); };

    // ---- pixel fragments: per-lane part of the address ----
    // address of block nb, tap (kh,kw), K-half hf = ((c3[kw] ^ (hf << 6 | (kh & 1) << 4) ^ ((nb & 1) << 4)) + buffer) + g2_imm(nb, kh, kw)
    unsigned c3[3];
    {
        int row, col, par;
        if (POOL) {         // block = 2 rows x 8 columns in 2x2-quad order
            const int q = l16 >> 2, sub = l16 & 3;
            row = wp * (NB / 2) * 2 + (sub >> 1);
            col = 2 * q + (sub & 1);
            par = sub >> 1;
        } else {            // block = one output row of 16 pixels
            row = wp * NB;
            col = g2_col(l16);
            par = 0;
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            c3[kw] = (unsigned)((row * G2_P + col) * 128 + (par << 4)) ^ (unsigned)((lq ^ g2_swz0(((col + kw) >> 1) & 7)) << 4);
    }
    const unsigned lds_a = (unsigned)(size_t)(lds_char*)Ap;

    // accumulators start from the bias of the tile's couts: the epilogue then needs no load (a bias load there sat,
    // with its full L2 latency, between the last MFMA of a tile and its stores)
    auto load_bias = [&](int ct, f32x4* b) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int co = ct * TC + wc * 32 + 8 * lq + 4 * mt;
            b[mt] = (a.bias && co < a.Cout) ? *reinterpret_cast<const f32x4*>(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 acc[2][NB];
    {
        f32x4 b0[2];
        load_bias(pi_ct, b0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = b0[i];
    }

    bf16x8 af[3][2][2];       // [tap % 3][K-half][16-cout block]
    bf16x8 bfr[8];            // pixel-fragment ring, slot = block & 7
    unsigned e0, e1;          // fragment base addresses of the K-step being fetched (even / odd blocks)

#define G2_LOAD_A(SLOT, SBASE, TAP)                                                                            \
    {                                                                                                          \
        _Pragma("unroll") for (int hf_ = 0; hf_ < 2; ++hf_)                                                    \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 2; ++mt_) {                                              \
                const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(srd_w, voff[mt_], (SBASE) + (TAP) * 128 + hf_ * 64, 0); \
                af[SLOT][hf_][mt_] = __builtin_bit_cast(bf16x8, v_);                                           \
            }                                                                                                  \
    }
    // set e0/e1 for K-step R18 (tap = R18 >> 1, K-half = R18 & 1) of patch buffer base BUFB
#define G2_SET_E(R18, BUFB)                                                                                    \
    {                                                                                                          \
        constexpr int tap_ = (R18) >> 1, hf_ = (R18) & 1, kh_ = tap_ / 3, kw_ = tap_ - kh_ * 3;                \
        const unsigned x_ = c3[kw_] ^ (unsigned)((hf_ << 6) | ((kh_ & 1) << 4));                               \
        e0 = x_ + (BUFB);                                                                                      \
        e1 = (x_ ^ 16u) + (BUFB);                                                                              \
    }
    // issue the 4 reads of group GI of K-step R18
#define G2_READS(R18, GI)                                                                                      \
    if constexpr (!(CVPCE_DBG & 32)) {                                                                         \
        constexpr int tap_ = (R18) >> 1, kh_ = tap_ / 3, kw_ = tap_ - kh_ * 3;                                 \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[(4 * (GI) + 0) & 7]) : "v"(e0), "n"(g2_imm<POOL>(4 * (GI) + 0, kh_, kw_))); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[(4 * (GI) + 1) & 7]) : "v"(e1), "n"(g2_imm<POOL>(4 * (GI) + 1, kh_, kw_))); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[(4 * (GI) + 2) & 7]) : "v"(e0), "n"(g2_imm<POOL>(4 * (GI) + 2, kh_, kw_))); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[(4 * (GI) + 3) & 7]) : "v"(e1), "n"(g2_imm<POOL>(4 * (GI) + 3, kh_, kw_))); \
    }
    // the 8 MFMAs of group GI of K-step R18; waits until that group's fragments have landed (the next group's 4 reads,
    // issued just before, may stay in flight).  The "+v" ties keep the compiler from hoisting an MFMA above the wait.
#define G2_MFMAS(R18, GI, NOUT)                                                                                \
    {                                                                                                          \
        constexpr int ts_ = ((R18) >> 1) % 3, hf_ = (R18) & 1;                                                 \
        asm volatile("s_waitcnt lgkmcnt(%4)"                                                                   \
                     : "+v"(bfr[(4 * (GI) + 0) & 7]), "+v"(bfr[(4 * (GI) + 1) & 7]), "+v"(bfr[(4 * (GI) + 2) & 7]), "+v"(bfr[(4 * (GI) + 3) & 7]) \
                     : "n"(NOUT));                                                                             \
        if (!(CVPCE_DBG & 1)) __builtin_amdgcn_s_setprio(1);                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                       \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 2; ++mt_)                                                \
                acc[mt_][4 * (GI) + i_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ts_][hf_][mt_], bfr[(4 * (GI) + i_) & 7], acc[mt_][4 * (GI) + i_], 0, 0, 0); \
        if (!(CVPCE_DBG & 1)) __builtin_amdgcn_s_setprio(0);                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
    // group GI of K-step R18 (not the chunk's last group): prefetch the next group, then compute this one
#define G2_GROUP(R18, GI)                                                                                      \
    if constexpr ((GI) < NG) {                                                                                 \
        if constexpr ((GI) + 1 < NG) {                                                                         \
            G2_READS(R18, (GI) + 1)                                                                            \
            G2_MFMAS(R18, GI, 4)                                                                               \
        } else if constexpr ((R18) + 1 < 18) {                                                                 \
            G2_SET_E((R18) + 1, bufb)                                                                          \
            G2_READS((R18) + 1, 0)                                                                             \
            G2_MFMAS(R18, GI, 4)                                                                               \
        }                                                                                                      \
    }
#define G2_KSTEP(R18) G2_GROUP(R18, 0) G2_GROUP(R18, 1) G2_GROUP(R18, 2) G2_GROUP(R18, 3)
    // one 3x3 tap = two K-steps; first fetch the weights of the tap after next (same chunk, or the next chunk's first two)
#define G2_TAP(T)                                                                                              \
    {                                                                                                          \
        if constexpr (!(CVPCE_DBG & 8)) {                                                                      \
        if constexpr ((T) + 2 < 9) G2_LOAD_A(((T) + 2) % 3, sb_cur, (T) + 2)                                   \
        else G2_LOAD_A(((T) + 2) % 3, sb_next, (T) + 2 - 9)   /* past the last chunk: a harmless reload */     \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        G2_KSTEP(2 * (T))                                                                                      \
        G2_KSTEP(2 * (T) + 1)                                                                                  \
    }

    // ---- prologue ----
    int seq = 0, cchunk = 0, t_n, t_ty, t_tx, t_ct;
    tile_of(0, t_n, t_ty, t_tx, t_ct);
    int n_ct = t_ct;                                    // cout tile of the NEXT chunk's tile
    auto next_ct = [&]() {
        if (cchunk + 1 < nchunks) return t_ct;
        if (seq + 1 < my_tiles) return (lbid + (seq + 1) * (int)gridDim.x) % a.ctiles;
        return t_ct;                                    // no next chunk: any valid address will do
    };
    issue_next_patch();
    issue_next_patch();
    int sb_cur = wbase(t_ct, 0);
    n_ct = next_ct();
    int sb_next = wbase(n_ct, (cchunk + 1 < nchunks) ? cchunk + 1 : 0);
    G2_LOAD_A(0, sb_cur, 0)
    G2_LOAD_A(1, sb_cur, 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned bufb = lds_a;                               // LDS base of the current chunk's patch buffer
    int bufi = 0;
    G2_SET_E(0, bufb)
    G2_READS(0, 0)

    const int lp = lane & 15;
    for (int cc = 0; cc < total_chunks; ++cc) {
        // keep the per-K-step address variants (c3 ^ constant) inside the loop: hoisted they cost 36 VGPRs
        asm volatile("" : "+v"(c3[0]), "+v"(c3[1]), "+v"(c3[2]));
        G2_TAP(0) G2_TAP(1) G2_TAP(2) G2_TAP(3) G2_TAP(4) G2_TAP(5) G2_TAP(6) G2_TAP(7) G2_TAP(8)
        // ---- last group of the chunk: chunk hand-off ----
        // every wave is done with the PREVIOUS chunk's buffer and (vmcnt) its own pieces of the NEXT chunk's patch have
        // landed: at most the 8 weight loads of the next chunk's first two taps are younger than those pieces
        const int nbufi = (bufi == 2) ? 0 : bufi + 1;
        const unsigned nbufb = lds_a + (unsigned)nbufi * G2_A_BYTES;
        // (after the last chunk the barrier, the reads and the weight loads still run -- on valid, unused data --
        // so that the loop body has one shape and the accumulators stay in place)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_next_patch();                              // chunk cc + 2 -> the buffer chunk cc - 1 used
        G2_SET_E(0, nbufb)
        G2_READS(0, 0)
        G2_MFMAS(17, NG - 1, 4)
        bufb = nbufb;
        bufi = nbufi;
        sb_cur = sb_next;

        if (cchunk + 1 == nchunks) {
            // ---- epilogue of this tile (the next tile's patch, weights and first fragments are already in flight) ----
            const int n = t_n, ty = t_ty, tx = t_tx, ct = t_ct;
            f32x4 nbias[2];                      // bias of the NEXT tile's couts: lands while this tile is stored
            load_bias(n_ct, nbias);
#pragma unroll
            for (int nt = 0; nt < NB; ++nt) {
                const int nb = wp * NB + nt;
                size_t opix;
                bool keep = true;                // masked-out pixels (gaps of a level atlas) are stored as zeros
                bool store_lane;                 // ragged right / bottom tiles: pixels outside the image are dropped
                if (POOL) {
                    const int q = lp >> 2;
                    const int oy = (ty * G2_T) / 2 + (nb >> 1), ox = (tx * G2_T) / 2 + 4 * (nb & 1) + q;
                    opix = (size_t)(n * (a.H >> 1) + oy) * (a.W >> 1) + ox;
                    store_lane = (lp & 3) == 0 && oy < (a.H >> 1) && ox < (a.W >> 1);
                } else {
                    const int oy = ty * G2_T + nb, ox = tx * G2_T + g2_col(lp);
                    opix = (size_t)(n * a.H + oy) * a.W + ox;
                    store_lane = oy < a.H && ox < a.W;
                    if (a.mask && store_lane) keep = a.mask[oy * a.W + ox] != 0;
                }
                const int co = ct * TC + wc * 32 + 8 * lq;            // this lane's 8 consecutive couts
                float v[8];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[4 * mt + j] = acc[mt][nt][j];
                if (a.relu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = relu_bits(v[j]);
                    if (POOL) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = quad_max_nonneg(v[j]);
                    }
                } else if (POOL) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = quad_max(v[j]);
                }
                if (store_lane && co < a.Cout && !(CVPCE_DBG & 16)) {
                    const bf16x4 lo = f32x4_to_bf16x4(f32x4{v[0], v[1], v[2], v[3]}), hi = f32x4_to_bf16x4(f32x4{v[4], v[5], v[6], v[7]});
                    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    *reinterpret_cast<u32x4*>(a.out + opix * a.Cout + co) = keep ? u32x4{l2.x, l2.y, h2.x, h2.y} : u32x4{0u, 0u, 0u, 0u};
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NB; ++nt) acc[mt][nt] = nbias[mt];
            cchunk = 0;
            ++seq;
            if (seq < my_tiles) tile_of(seq, t_n, t_ty, t_tx, t_ct);
        } else {
            ++cchunk;
        }
        // weights base of the chunk after the (new) current one
        n_ct = next_ct();
        sb_next = wbase(n_ct, (cchunk + 1 < nchunks) ? cchunk + 1 : 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the trailing prefetch
#undef G2_TAP
#undef G2_KSTEP
#undef G2_GROUP
#undef G2_MFMAS
#undef G2_READS
#undef G2_SET_E
#undef G2_LOAD_A
}

template <int WC, int WP, bool POOL>
static int launch_halo2(Halo2Args a, hipStream_t stream) {
    constexpr int TC = 32 * WC;
    a.ctiles = (a.Cout + TC - 1) / TC;
    a.ntiles = a.ptiles * a.ctiles;
    const int smem = 3 * G2_A_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_halo2_kernel<WC, WP, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return CVPCE_ERR_LAUNCH;
        attr_set = true;
    }
    const int grid = a.ntiles < 256 ? a.ntiles : 256;
    hipLaunchKernelGGL((conv3x3_halo2_kernel<WC, WP, POOL>), dim3(grid), dim3(512), smem, stream, a);
    return cvpce_check_launch();
}

static int halo2_dispatch(const void* in, const void* wgt, const float* bias, const unsigned char* mask, void* out, int N,
                          int H, int W, int Cin, int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, void* stream) {
    if (N <= 0) return CVPCE_OK;
    if (!in || !wgt || !out) return CVPCE_ERR_ARG;
    if (H <= 0 || W <= 0 || Cin % 64 != 0 || Cin <= 0 || Cout % 8 != 0 || Cout <= 0) return CVPCE_ERR_ARG;
    if (fuse_pool2 && ((H & 1) || (W & 1))) return CVPCE_ERR_ARG;
    if (K_pad != 9 * Cin || Cout_pad % 256 != 0 || Cout_pad < Cout) return CVPCE_ERR_ARG;
    if ((long long)N * H * W * Cin * 2 >= (1LL << 32) || (long long)N * H * W * Cout >= (1LL << 31)) return CVPCE_ERR_ARG;
    if ((long long)Cout_pad * K_pad * 2 >= (1LL << 31)) return CVPCE_ERR_ARG;
    Halo2Args a;
    if (mask && fuse_pool2) return CVPCE_ERR_ARG;
    a.in = (const bf16_t*)in; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.mask = mask; a.out = (bf16_t*)out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.K_pad = K_pad; a.relu = relu;
    a.tiles_x = (W + G2_T - 1) / G2_T; a.tiles_y = (H + G2_T - 1) / G2_T; a.ptiles = N * a.tiles_x * a.tiles_y;
    a.in_bytes = (unsigned)((long long)N * H * W * Cin * 2);
    a.wgt_bytes = (unsigned)((long long)Cout_pad * K_pad * 2);
    a.ctiles = a.ntiles = 0;
    if ((long long)a.ptiles * ((Cout + 127) / 128) * (Cin / 64) >= (1LL << 30)) return CVPCE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (Cout > 128) return fuse_pool2 ? launch_halo2<8, 1, true>(a, s) : launch_halo2<8, 1, false>(a, s);
    return fuse_pool2 ? launch_halo2<4, 2, true>(a, s) : launch_halo2<4, 2, false>(a, s);
}

extern "C" int cvpce_conv3x3_halo(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W,
                                  int Cin, int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, void* stream) {
    return halo2_dispatch(in, wgt, bias, nullptr, out, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, fuse_pool2, stream);
}

extern "C" int cvpce_conv3x3_halo_masked(const void* in, const void* wgt, const float* bias, const unsigned char* mask,
                                         void* out, int N, int H, int W, int Cin, int Cout, int K_pad, int Cout_pad,
                                         int relu, void* stream) {
    if (!mask) return CVPCE_ERR_ARG;
    return halo2_dispatch(in, wgt, bias, mask, out, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, 0, stream);
}
