// 3x3 / stride 1 / pad 1 convolution for layers with FEW output channels (Cout <= 128: VGG16 conv2_1, conv2_2), Cin % 64 == 0.
//
// conv3x3_halo2.hip gives every wave 32 couts x 256 pixels, which needs 256 couts per workgroup; at Cout = 128 its waves
// have to split the pixel tile instead, every weight fragment is then fetched by two waves and feeds half as many
// MFMAs -- the weight stream costs 22 % of the layer (profiles/r01d_ablation_halo2.md).  This variant keeps the
// 32-cout x 256-pixel wave tile by making the workgroup's pixel tile 16 rows x 32 columns: waves (wc, wp) = (cout
// group of 32, left / right 16 columns).  To fit the larger patch the channel chunk is 32 (64-byte pixels):
//   * patch 18 x 34 pixels x 32 channels = 38.25 KiB, triple buffered (LDS-DMA, zero-filled borders),
//   * row streaming as in conv3x3_halo2.hip: a step fixes kw, holds the weights of the three taps (kh, kw) of the
//     32-channel sub-chunk (6 loads of 64 B per row, straight from L2 into MFMA layout, one step ahead), reads each of
//     the 18 patch rows once (ds_read_b128, immediate offsets) and issues up to 6 MFMA 16x16x32 on it,
//   * one workgroup barrier per sub-chunk (3 steps); the loop body is one 64-channel chunk = two sub-chunks.
// Pixel lanes are laid out along rows in BOTH modes (lanes {0-3,12-15} = even columns, {4-11} = odd columns), which is
// conflict-free for 64-byte pixels under the chunk ^ ((px >> 2) & 3) swizzle; the fused MaxPool2d(2,2) is a plain max
// of two accumulator rows plus one masked row-rotate per side.
// K order / weight layout: chunk-major [Cout_pad][K_pad] of include/cvpce_amd.h.  Fused bias / ReLU / MaxPool2d(2,2).
#include "common.h"
#include "../../include/cvpce_amd.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

#ifndef CVPCE_DBG
#define CVPCE_DBG 0          // compile-time ablations (tools/ablate.sh): 64 no hand-off wait / barrier, 4 no patch DMA after the first two sub-chunks, 8 no weight
#endif                       // loads in the loop, 16 no stores (runtime-false predicate), 32 no fragment reads; timing only; 2 plain tile order
#define G3_TH 16
#define G3_TW 32
#define G3_PW 34
#define G3_NPIX (18 * 34)
#define G3_NPIECE 39                       // 16 pixels of 64 B per 1-KiB DMA piece
#define G3_A_BYTES (G3_NPIECE * 1024)
// work-list launches (skiplist.hip): the first G3_MAX_SEQ entries of the workgroup's own list in LDS
#define G3_MAX_SEQ 512
#define G3_LLIST_OFF (3 * G3_A_BYTES)
#define G3_SMEM_LIST (G3_LLIST_OFF + G3_MAX_SEQ * 8)

__device__ __forceinline__ int g3_col(int l16) { return l16 < 4 ? 2 * l16 : (l16 >= 12 ? 2 * (l16 - 8) : 2 * (l16 - 4) + 1); }

struct Halo3Args {
    const bf16_t* in;    // [N][H][W][Cin]
    const bf16_t* wgt;   // fragment-major (include/cvpce_amd.h)
    const float* bias;
    bf16_t* out;         // [N][H][W][Cout] or pooled [N][H/2][W/2][Cout]
    int N, H, W, Cin, Cout, K_pad, relu;
    int cgroups;         // Cout_pad / 32
    int tiles_x, tiles_y, ptiles, ctiles, ntiles;
    unsigned in_bytes, wgt_bytes;
    // LIST launches (see conv3x3_halo2.hip): the tiles to compute, *list_count entries; input pixels in the constant region of
    // their crop are read from image N - 1
    const unsigned long long* list;
    const int* list_count;
};

// diagnostic build only (tools/ablate.sh conv3x3_halo3 128; tools/dev/halo3_stamps.py): s_memtime ticks two waves of one workgroup
// spend in [weights wait at the end of a step, hand-off vmcnt, hand-off barrier, epilogue, whole loop]; [5] = timed steps
#if CVPCE_DBG & 128
__device__ unsigned long long cvpce_halo3_stamps[2][6];
extern "C" int cvpce_debug_halo3_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cvpce_halo3_stamps), sizeof(cvpce_halo3_stamps)) == hipSuccess ? 0 : 2;
}
#define G3_TIC() const unsigned long long tic_ = __builtin_amdgcn_s_memtime();
#define G3_TOC(I) st_[I] += __builtin_amdgcn_s_memtime() - tic_;
#define G3_STAMP_STEP() { G3_TIC() asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); G3_TOC(0) st_[5] += 1; }
#else
#define G3_TIC()
#define G3_TOC(I)
#define G3_STAMP_STEP()
#endif

// THIN (round 5): the RetinaNet head's OUTPUT convs (256 -> 9 class logits, 256 -> 36 box regressions; fp32 outputs, no activation): Cout is
// any number <= 128, the output is [N][H][W][Cout] float, and a wave whose 32-cout group holds no real cout only takes part in the patch
// DMA and the barriers (no weight loads, fragment reads or MFMAs).  As register-staged implicit GEMMs these two layers cost 72 + 60 us per
// 4 images for ~10 us of traffic each (36 K-steps of gathers per 128-pixel tile); here the patch of a 16 x 32 tile is fetched once per chunk.
template <typename E, bool POOL, bool LIST, bool THIN = false>
__global__ __launch_bounds__(512, 2) void conv3x3_halo3_kernel(Halo3Args a) {
    static_assert(!THIN || (!POOL && !LIST), "thin-output form: plain launches only");
#if CVPCE_DBG & 128
    unsigned long long st_[6] = {0, 0, 0, 0, 0, 0};
#endif
    constexpr int TC = 128, NB = 16, NG = 4;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ap = smem;                 // [3][612 (+12)][32] bf16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wid >> 1, wp = wid & 1;
    const int l16 = lane & 15, lq = lane >> 4;
    const bool active = THIN ? (wc * 32 < a.Cout) : true;       // (THIN: this wave's cout group holds a real cout; wave-uniform)

    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_p = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);

    const int nchunks = a.Cin >> 6;                       // 64-channel chunks = loop bodies per tile
    // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin; remapped, an XCD walks 32 CONSECUTIVE tiles per round
    // (a whole 128 x 128 crop), so the halo columns / rows neighbouring tiles share are served by its L2 instead of being
    // fetched again from beyond it: conv2_1 +2.3 %, conv2_2 +1.8 % (same call, three alternations, real activations;
    // CVPCE_DBG & 2 = the plain order, dev A/B)
    const int lbid = (CVPCE_DBG & 2) ? (int)blockIdx.x : xcd_remap((int)blockIdx.x, (int)gridDim.x);
    // LIST launches: a contiguous block of the list per workgroup, the cout tile fixed per workgroup (see conv3x3_halo2.hip)
    int my_tiles, l_first = 0;
    if constexpr (LIST) {
        const int entries = __builtin_amdgcn_readfirstlane(*a.list_count);
        const int groups = (int)gridDim.x / a.ctiles;
        const int per = (entries + groups - 1) / groups;
        l_first = (lbid / a.ctiles) * per;
        my_tiles = entries - l_first < per ? entries - l_first : per;
    } else {
        my_tiles = (a.ntiles - lbid + (int)gridDim.x - 1) / (int)gridDim.x;
    }
    if (my_tiles <= 0) return;
    const int total_chunks = my_tiles * nchunks;          // < 2^30: checked on the host
    const int total_sub = 2 * total_chunks;
    unsigned long long* llist = reinterpret_cast<unsigned long long*>(smem + G3_LLIST_OFF);
    if constexpr (LIST) {
        const int staged = my_tiles < G3_MAX_SEQ ? my_tiles : G3_MAX_SEQ;
        for (int idx = tid; idx < staged; idx += 512) llist[idx] = a.list[l_first + idx];
        __syncthreads();
    }

    auto tile_of = [&](int seq, int& n, int& ty, int& tx, int& ct, int& ext) {
        if constexpr (LIST) {
            ct = lbid % a.ctiles;
            const unsigned long long e = seq < G3_MAX_SEQ ? llist[seq] : a.list[l_first + seq];
            const int lo = __builtin_amdgcn_readfirstlane((int)(unsigned)e);
            ext = __builtin_amdgcn_readfirstlane((int)(unsigned)(e >> 32));
            n = lo >> 16;
            ty = (lo >> 8) & 0xFF;
            tx = lo & 0xFF;
            return;
        }
        ext = 0;
        const int t = lbid + seq * (int)gridDim.x;
        ct = t % a.ctiles;
        const int p = t / a.ctiles;
        n = p / (a.tiles_x * a.tiles_y);
        const int r = p - n * (a.tiles_x * a.tiles_y);
        ty = r / a.tiles_x;
        tx = r - ty * a.tiles_x;
    };

    // ---- patch DMA: piece j fills patch pixels 16j .. 16j+15 (pixel = lane>>2, phys chunk = lane&3); pieces dealt
    //      round-robin to the 8 waves (wave w: pieces w, w+8, ...: 5 for w < 7, else 4) ----
    const int npp = (wid < 7) ? 5 : 4;
    auto issue_patch = [&](int n, int ty, int tx, int c32, int buf, int ext) {
        const int y0 = ty * G3_TH - 1, x0 = tx * G3_TW - 1;
        const int ey = (ext >> 12) & 0xFFF, ex = ext & 0xFFF;
        // lane id recomputed here (2 VALU ops, once per patch) instead of living in a VGPR across the K loop; the empty
        // asm also keeps the per-piece constants below from being hoisted out of the chunk loop (18+ VGPRs)
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i < npp) {
                const int j = wid + 8 * i;
                const int pp = j * 16 + (ln >> 2);
                const int py = pp / G3_PW, px = pp - py * G3_PW;
                const int lchunk = (ln & 3) ^ ((px >> 2) & 3);
                const int y = y0 + py, x = x0 + px;
                const bool ok = pp < G3_NPIX && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                int nn = n;
                if constexpr (LIST) nn = (y >= ey || x >= ex) ? a.N - 1 : n;    // constant region of the crop: the constant crop's pixel
                const unsigned off = (unsigned)((((size_t)(nn * a.H + y) * a.W + x) * a.Cin + c32 * 32) * 2) + (unsigned)(lchunk * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_p, (lds_void*)(Ap + buf * G3_A_BYTES + j * 1024), 16,
                                                         (int)(ok ? off : 0xFFFFFFF0u), 0, 0, 0);
            }
        }
    };
    // patch issue pointer: the next flat 32-channel sub-chunk to fetch and its tile
    int pi = 0, pi_seq = 0, pi_c = 0, pi_buf = 0, pi_n, pi_ty, pi_tx, pi_ct, pi_ext;
    tile_of(0, pi_n, pi_ty, pi_tx, pi_ct, pi_ext);
    const int nsub = 2 * nchunks;
    auto issue_next_patch = [&]() {
        if (pi < total_sub) {
            if (!(CVPCE_DBG & 4) || pi < 2) issue_patch(pi_n, pi_ty, pi_tx, pi_c, pi_buf, pi_ext);
            ++pi;
            if (++pi_buf == 3) pi_buf = 0;
            if (++pi_c == nsub) {
                pi_c = 0;
                ++pi_seq;
                if (pi_seq < my_tiles) tile_of(pi_seq, pi_n, pi_ty, pi_tx, pi_ct, pi_ext);
            }
        }
    };

    // ---- weights: lane (m = l16, q = lq) loads 16 B = k 8q .. 8q+7 of row m of a 16-cout block ----
    // MFMA row m = 4q + j of block mt is cout 8q + 4mt + j of the wave's 32: accumulator lane group q then holds 8
    // CONSECUTIVE couts of its pixel -> one 16-byte store per pixel block.  The weights are FRAGMENT-MAJOR (see conv3x3_halo2.hip):
    // the fragment of (chunk c, 32-cout group, kw, K-half, kh, block mt) is one contiguous KiB, lane L's 16 bytes at byte 16 L.
    const unsigned voff1 = (unsigned)(lane * 16);
    auto wbase = [&](int ct, int c) { return __builtin_amdgcn_readfirstlane((int)((unsigned)(c * a.cgroups + ct * (TC / 32) + wc) * 36864u)); };

    // ---- pixel fragments: lane (l16, lq) reads pixel (patch row p, column 16 wp + g3_col(l16) + kw), K-quarter lq:
    //      address = (c3[kw] + buffer) + (p * 34 + kw) * 64 ----
    unsigned c3[3];
    {
        const int col = 16 * wp + g3_col(l16);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) c3[kw] = (unsigned)(col * 64 + ((lq ^ (((col + kw) >> 2) & 3)) << 4));
    }
    const unsigned lds_a = (unsigned)(size_t)(lds_char*)Ap;

    auto load_bias = [&](int ct, f32x4* b) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int co = ct * TC + wc * 32 + 8 * lq + 4 * mt;
            if constexpr (THIN) {                      // Cout is not a multiple of 4: element by element
#pragma unroll
                for (int j = 0; j < 4; ++j) b[mt][j] = (a.bias && co + j < a.Cout) ? a.bias[co + j] : 0.f;
            } else {
                b[mt] = (a.bias && co < a.Cout) ? *reinterpret_cast<const f32x4*>(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
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

    bf16x8 af[2][3][2];       // [step parity][kh][16-cout block]
    bf16x8 bfr[4];            // patch-row ring
    unsigned e0;              // fragment base address of the step being fetched

    // ROW STREAMING (see conv3x3_halo2.hip): step T = sub-chunk * 3 + kw of a body (0..5) holds the weights of the three
    // taps (kh, kw) of its 32 channels, reads each of the 18 patch rows once and issues up to 6 MFMAs on it.
#define G3_LOAD_A(T, SBASE)                                                                                    \
    if constexpr (!(CVPCE_DBG & 8)) if (active) {                                                              \
        _Pragma("unroll") for (int kh_ = 0; kh_ < 3; ++kh_)                                                    \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 2; ++mt_) {                                              \
                const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(srd_w, voff1, (SBASE) + (((((T) % 3) * 2 + (T) / 3) * 3 + kh_) * 2 + mt_) * 1024, 0); \
                af[(T) & 1][kh_][mt_] = __builtin_bit_cast(bf16x8, v_);                                        \
            }                                                                                                  \
    }
#define G3_SET_E(T, BUFB) { e0 = c3[(T) % 3] + (BUFB); }
    // read patch row P of step T into ring slot (P + 2 T) & 3 (a step has 18 rows, 18 = 2 mod 4)
#define G3_READ(T, P)                                                                                          \
    if constexpr (!(CVPCE_DBG & 32)) if (active)                                                               \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[((P) + 2 * (T)) & 3]) : "v"(e0), "n"(((P) * G3_PW + (T) % 3) * 64));
    // the MFMAs of patch row P: output rows P (kh = 0), P-1 (kh = 1), P-2 (kh = 2) where they exist
#define G3_ROW(T, P)                                                                                           \
    if (active) {                                                                                              \
        if constexpr (!(CVPCE_DBG & 32)) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bfr[((P) + 2 * (T)) & 3]));  \
        _Pragma("unroll") for (int kh_ = 0; kh_ < 3; ++kh_)                                                    \
            if ((P) - kh_ >= 0 && (P) - kh_ < NB) {                                                            \
                _Pragma("unroll") for (int mt_ = 0; mt_ < 2; ++mt_)                                            \
                    acc[mt_][(P) - kh_] = E::mfma16(af[(T) & 1][kh_][mt_], bfr[((P) + 2 * (T)) & 3], acc[mt_][(P) - kh_]); \
            }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
#define G3_RP(T, P) G3_READ(T, (P) + 2) G3_ROW(T, P)
    // step inside a sub-chunk (T = 0, 1, 3, 4): fetch the next step's weights, stream the rows; rows 16, 17 prefetch
    // rows 0, 1 of step T + 1 from the same buffer
    // the sub-chunk hand-off: this wave is done with the PREVIOUS sub-chunk's buffer and (vmcnt) its own pieces of the NEXT
    // sub-chunk's patch have landed -- at most the 6 weight loads just issued are younger than those pieces; then the DMA
    // of the sub-chunk after the next one goes into the buffer the previous one used.
    // (Tried: issuing that DMA one step later, BEHIND the next weight loads, so that the in-order return of the loads gives
    // the pieces two steps instead of one before a weight wait can stall on them -- no gain, conv2_1 0.70 -> 0.72 ms.)
#define G3_HANDOFF()                                                                                           \
    {                                                                                                          \
        { G3_TIC() if (!(CVPCE_DBG & 64)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); G3_TOC(1) }         \
        { G3_TIC() if (!(CVPCE_DBG & 64)) __builtin_amdgcn_s_barrier(); G3_TOC(2) }     /* (64: timing-only ablation, no hand-off wait / barrier) */ \
        issue_next_patch();                                                                                    \
    }
    // LIST launches compute only the first `rows_` (4 | 8 | 12 | 16) output rows of a tile: see G2_ROWS_AND_TAIL in conv3x3_halo2.hip
#define G3_ROWS_AND_TAIL(T, TN, BUFB_NEXT, BEFORE)                                                             \
    G3_RP(T, 0) G3_RP(T, 1) G3_RP(T, 2) G3_RP(T, 3) G3_RP(T, 4) G3_RP(T, 5)                                    \
    if (!LIST || rows_ > 4) { G3_RP(T, 6) G3_RP(T, 7) G3_RP(T, 8) G3_RP(T, 9) }                                \
    if (!LIST || rows_ > 8) { G3_RP(T, 10) G3_RP(T, 11) G3_RP(T, 12) G3_RP(T, 13) }                            \
    if (!LIST || rows_ > 12) {                                                                                 \
        G3_RP(T, 14) G3_RP(T, 15)                                                                              \
        BEFORE                                                                                                 \
        G3_SET_E(TN, BUFB_NEXT)                                                                                \
        G3_READ(TN, 0) G3_ROW(T, 16)                                                                           \
        G3_READ(TN, 1) G3_ROW(T, 17)                                                                           \
    } else {                                                                                                   \
        BEFORE                                                                                                 \
        G3_SET_E(TN, BUFB_NEXT)                                                                                \
        G3_READ(TN, 0)                                                                                         \
        G3_READ(TN, 1)                                                                                         \
    }
#define G3_STEP(T)                                                                                             \
    {                                                                                                          \
        G3_LOAD_A((T) + 1, sb_cur)                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        G3_ROWS_AND_TAIL(T, (T) + 1, bufb, )                                                                   \
        G3_STAMP_STEP()                                                                                        \
    }
    // last step of a sub-chunk (T = 2 or 5) with the hand-off.  After the last sub-chunk the barrier, the reads and the
    // weight loads still run, on valid but unused data, so that the loop body has one shape.
#define G3_STEP_HANDOFF(T, TN, SBN)                                                                            \
    {                                                                                                          \
        G3_LOAD_A(TN, SBN)                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        const int nbufi = (bufi == 2) ? 0 : bufi + 1;                                                          \
        const unsigned nbufb = lds_a + (unsigned)nbufi * G3_A_BYTES;                                           \
        G3_ROWS_AND_TAIL(T, TN, nbufb, G3_HANDOFF())                                                           \
        bufb = nbufb;                                                                                          \
        bufi = nbufi;                                                                                          \
    }

    // ---- prologue ----
    int seq = 0, cchunk = 0, t_n, t_ty, t_tx, t_ct, t_ext_;
    tile_of(0, t_n, t_ty, t_tx, t_ct, t_ext_);
    int rows_ = LIST ? ((t_ext_ >> 24) & 0xFF) : NB;     // output rows of the current tile that are computed (LIST: 4 | 8 | 12 | 16)
    int n_ct = t_ct;                                    // cout tile of the NEXT chunk's tile
    auto next_ct = [&]() {
        if (cchunk + 1 < nchunks) return t_ct;
        if constexpr (LIST) return t_ct;                 // (fixed per workgroup)
        if (seq + 1 < my_tiles) return (lbid + (seq + 1) * (int)gridDim.x) % a.ctiles;
        return t_ct;                                    // no next chunk: any valid address will do
    };
    issue_next_patch();
    issue_next_patch();
    int sb_cur = wbase(t_ct, 0);
    n_ct = next_ct();
    int sb_next = wbase(n_ct, (cchunk + 1 < nchunks) ? cchunk + 1 : 0);
    G3_LOAD_A(0, sb_cur)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned bufb = lds_a;                               // LDS base of the current sub-chunk's patch buffer
    int bufi = 0;
    G3_SET_E(0, bufb)
    G3_READ(0, 0)
    G3_READ(0, 1)

    const int lp = lane & 15;
#if CVPCE_DBG & 128
    const unsigned long long loop0_ = __builtin_amdgcn_s_memtime();
#endif
    for (int cc = 0; cc < total_chunks; ++cc) {
        asm volatile("" : "+v"(c3[0]), "+v"(c3[1]), "+v"(c3[2]));
        G3_STEP(0) G3_STEP(1)
        G3_STEP_HANDOFF(2, 3, sb_cur)
        G3_STEP(3) G3_STEP(4)
        G3_STEP_HANDOFF(5, 0, sb_next)
        sb_cur = sb_next;

        if (cchunk + 1 == nchunks) {
            // ---- epilogue of this tile (the next tile's patch, weights and first fragments are already in flight) ----
            const int n = t_n, ty = t_ty, tx = t_tx, ct = t_ct;
            const int rows_t = rows_;            // (LIST) rows of this tile that were computed: the others are not stored
            G3_TIC()
            f32x4 nbias[2];                      // bias of the NEXT tile's couts: lands while this tile is stored
            load_bias(n_ct, nbias);
            const int col = g3_col(lp);
            if constexpr (THIN) {
                // fp32 outputs, [pixel][Cout] with any Cout: this lane's (up to) 8 couts of its pixel, 16 bytes at a time where Cout % 4 == 0
                float* outf = reinterpret_cast<float*>(a.out);
                const int ox = tx * G3_TW + 16 * wp + col;
#pragma unroll
                for (int nt = 0; nt < NB; ++nt) {
                    const int oy = ty * G3_TH + nt;
                    if (active && oy < a.H && ox < a.W) {
                        float* po = outf + ((size_t)(n * a.H + oy) * a.W + ox) * a.Cout;
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            const int co = wc * 32 + 8 * lq + 4 * mt;
                            if ((a.Cout & 3) == 0) {
                                if (co < a.Cout) *reinterpret_cast<f32x4*>(po + co) = acc[mt][nt];
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (co + j < a.Cout) po[co + j] = acc[mt][nt][j];
                            }
                        }
                    }
                }
            } else if (POOL) {
                // rows 2i, 2i+1 are blocks 2i, 2i+1 of the same lane; columns 2k, 2k+1 are lanes A[k], B[k] with
                // A = {0-3,12-15}, B = {4-11}: lane A[k] takes its right neighbour by a row rotate (+4 for lanes 0-3,
                // -4 for lanes 12-15; bank masks 1 and 8)
                const int ox = tx * (G3_TW / 2) + 8 * wp + (col >> 1);
                const bool lane_ok = (lp < 4 || lp >= 12) && ox < (a.W >> 1);
#pragma unroll
                for (int i = 0; i < NB / 2; ++i) {
                    const int oy = ty * (G3_TH / 2) + i;
                    const size_t opix = (size_t)(n * (a.H >> 1) + oy) * (a.W >> 1) + ox;
                    const int co = ct * TC + wc * 32 + 8 * lq;        // this lane's 8 consecutive couts
                    float r[8];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (a.relu) {
                                const unsigned v = max(__float_as_uint(relu_bits(acc[mt][2 * i][j])), __float_as_uint(relu_bits(acc[mt][2 * i + 1][j])));
                                unsigned m = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x12C, 0xF, 0x1, false));   // row_ror:12 -> lane l reads l+4
                                m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0x8, false));            // row_ror:4  -> lane l reads l-4
                                r[4 * mt + j] = __uint_as_float(m);
                            } else {
                                const float v = fmaxf(acc[mt][2 * i][j], acc[mt][2 * i + 1][j]);
                                const float up = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x12C, 0xF, 0x1, false));
                                const float dn = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x124, 0xF, 0x8, false));
                                r[4 * mt + j] = fmaxf(v, fmaxf(up, dn));
                            }
                        }
                    if (lane_ok && oy < (a.H >> 1) && co < a.Cout && (!LIST || 2 * i < rows_t) && (!(CVPCE_DBG & 16) || a.relu == 12345)) {
                        const uint2 l2 = __builtin_bit_cast(uint2, E::pack4(f32x4{r[0], r[1], r[2], r[3]}));
                        const uint2 h2 = __builtin_bit_cast(uint2, E::pack4(f32x4{r[4], r[5], r[6], r[7]}));
                        *reinterpret_cast<u32x4*>(a.out + opix * a.Cout + co) = u32x4{l2.x, l2.y, h2.x, h2.y};
                    }
                }
            } else {
                // A lane holds 8 consecutive couts (16 bytes) of ITS pixel, and neighbouring lanes are neighbouring pixels, Cout * 2 bytes
                // apart: stored as they are, the 64 pieces of a store instruction share no 64-byte block and the texture addresser
                // takes them one lane per clock (see conv3x3_halo2.hip on the weight loads).  The 16 x 4 (pixel, cout group) pieces
                // are transposed across the wave first (4 ds_bpermute per row): lane 4 p + q then holds cout group q of pixel p,
                // and four neighbouring lanes write one contiguous 64-byte run.
                const int tp = lane >> 2, tq = lane & 3;                // after the exchange: pixel column index tp, cout group tq
                const int src4 = (16 * tq + tp) * 4;                    // byte address of the source lane (lq = tq, lp = tp)
                const int ox = tx * G3_TW + 16 * wp + g3_col(tp);
                const int co = ct * TC + wc * 32 + 8 * tq;
#pragma unroll
                for (int nt = 0; nt < NB; ++nt) {
                    const int oy = ty * G3_TH + nt;
                    const size_t opix = (size_t)(n * a.H + oy) * a.W + ox;
                    const bool store_lane = oy < a.H && ox < a.W && (!LIST || nt < rows_t);     // ragged right / bottom tiles; (LIST) computed rows only
                    f32x4 r0 = acc[0][nt], r1 = acc[1][nt];
                    if (a.relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { r0[j] = relu_bits(r0[j]); r1[j] = relu_bits(r1[j]); }
                    }
                    const uint2 l2 = __builtin_bit_cast(uint2, E::pack4(r0)), h2 = __builtin_bit_cast(uint2, E::pack4(r1));
                    u32x4 v;
                    v[0] = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)l2.x);
                    v[1] = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)l2.y);
                    v[2] = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)h2.x);
                    v[3] = (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)h2.y);
                    if (store_lane && co < a.Cout && (!(CVPCE_DBG & 16) || a.relu == 12345))
                        *reinterpret_cast<u32x4*>(a.out + opix * a.Cout + co) = v;
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NB; ++nt) acc[mt][nt] = nbias[mt];
            G3_TOC(3)
            cchunk = 0;
            ++seq;
            if (seq < my_tiles) {
                tile_of(seq, t_n, t_ty, t_tx, t_ct, t_ext_);
                if constexpr (LIST) rows_ = (t_ext_ >> 24) & 0xFF;
            }
        } else {
            ++cchunk;
        }
        n_ct = next_ct();
        sb_next = wbase(n_ct, (cchunk + 1 < nchunks) ? cchunk + 1 : 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the trailing prefetch
#if CVPCE_DBG & 128
    st_[4] = __builtin_amdgcn_s_memtime() - loop0_;
    if (blockIdx.x == 3 && (wid == 0 || wid == 5) && lane == 0)
        for (int i = 0; i < 6; ++i) cvpce_halo3_stamps[wid ? 1 : 0][i] = st_[i];
#endif
#undef G3_STEP_HANDOFF
#undef G3_STEP
#undef G3_HANDOFF
#undef G3_ROWS_AND_TAIL
#undef G3_RP
#undef G3_ROW
#undef G3_READ
#undef G3_SET_E
#undef G3_LOAD_A
}

template <typename E, bool POOL, bool LIST = false>
static int launch_halo3(Halo3Args a, hipStream_t stream) {
    a.ctiles = (a.Cout + 127) / 128;
    a.ntiles = a.ptiles * a.ctiles;
    const int smem = LIST ? G3_SMEM_LIST : 3 * G3_A_BYTES;
    if (!cvpce_smem_attr_done<conv3x3_halo3_kernel<E, POOL, LIST>>((const void*)conv3x3_halo3_kernel<E, POOL, LIST>, smem)) return CVPCE_ERR_LAUNCH;
    int grid = a.ntiles < g_cvpce_persistent_wgs ? a.ntiles : g_cvpce_persistent_wgs;
    if (LIST) grid = grid < a.ctiles ? a.ctiles : grid / a.ctiles * a.ctiles;     // (work-list launches: the same number of workgroups per cout tile)
    hipLaunchKernelGGL((conv3x3_halo3_kernel<E, POOL, LIST>), dim3(grid), dim3(512), smem, stream, a);
    return cvpce_check_launch();
}

template <typename E>
static int halo3_dispatch(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W,
                          int Cin, int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, void* stream,
                          const unsigned long long* list = nullptr, const int* count_dev = nullptr) {
    if (N <= 0) return CVPCE_OK;
    if (!in || !wgt || !out) return CVPCE_ERR_ARG;
    if (H <= 0 || W <= 0 || Cin % 64 != 0 || Cin <= 0 || Cout % 8 != 0 || Cout <= 0) return CVPCE_ERR_ARG;
    if (fuse_pool2 && ((H & 1) || (W & 1))) return CVPCE_ERR_ARG;
    if (K_pad != 9 * Cin || Cout_pad % 256 != 0 || Cout_pad < Cout) return CVPCE_ERR_ARG;
    if ((long long)N * H * W * Cin * 2 >= (1LL << 32) || (long long)N * H * W * Cout >= (1LL << 31)) return CVPCE_ERR_ARG;
    if ((long long)Cout_pad * K_pad * 2 >= (1LL << 31)) return CVPCE_ERR_ARG;
    Halo3Args a;
    a.in = (const bf16_t*)in; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.out = (bf16_t*)out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.K_pad = K_pad; a.relu = relu; a.cgroups = Cout_pad / 32;
    a.tiles_x = (W + G3_TW - 1) / G3_TW; a.tiles_y = (H + G3_TH - 1) / G3_TH; a.ptiles = N * a.tiles_x * a.tiles_y;
    a.in_bytes = (unsigned)((long long)N * H * W * Cin * 2);
    a.wgt_bytes = (unsigned)((long long)Cout_pad * K_pad * 2);
    a.ctiles = a.ntiles = 0;
    a.list = list; a.list_count = count_dev;
    if ((long long)a.ptiles * ((Cout + 127) / 128) * (Cin / 64) >= (1LL << 29)) return CVPCE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (list) {
        if (E::kF16 || !count_dev || N > 65535 || a.tiles_x > 255 || a.tiles_y > 255 || H > 65535 || W > 65535) return CVPCE_ERR_ARG;   // (the list entry exists for the bf16 embedder)
        if constexpr (!E::kF16) return fuse_pool2 ? launch_halo3<E, true, true>(a, s) : launch_halo3<E, false, true>(a, s);
    }
    return fuse_pool2 ? launch_halo3<E, true>(a, s) : launch_halo3<E, false>(a, s);
}

// the thin-output form: fp32 [N][H][W][Cout] outputs, Cout <= 128 and not necessarily a multiple of 8, no activation
template <typename E>
static int halo3_thin_dispatch(const void* in, const void* wgt, const float* bias, float* out, int N, int H, int W, int Cin, int Cout,
                               int K_pad, int Cout_pad, void* stream) {
    if (N <= 0) return CVPCE_OK;
    if (!in || !wgt || !out) return CVPCE_ERR_ARG;
    if (H <= 0 || W <= 0 || Cin % 64 != 0 || Cin <= 0 || Cout <= 0 || Cout > 128) return CVPCE_ERR_ARG;
    if (K_pad != 9 * Cin || Cout_pad % 256 != 0 || Cout_pad < Cout) return CVPCE_ERR_ARG;
    if ((long long)N * H * W * Cin * 2 >= (1LL << 32) || (long long)N * H * W * Cout >= (1LL << 31)) return CVPCE_ERR_ARG;
    if ((long long)Cout_pad * K_pad * 2 >= (1LL << 31)) return CVPCE_ERR_ARG;
    Halo3Args a;
    a.in = (const bf16_t*)in; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.out = reinterpret_cast<bf16_t*>(out);
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.K_pad = K_pad; a.relu = 0; a.cgroups = Cout_pad / 32;
    a.tiles_x = (W + G3_TW - 1) / G3_TW; a.tiles_y = (H + G3_TH - 1) / G3_TH; a.ptiles = N * a.tiles_x * a.tiles_y;
    a.in_bytes = (unsigned)((long long)N * H * W * Cin * 2);
    a.wgt_bytes = (unsigned)((long long)Cout_pad * K_pad * 2);
    a.ctiles = 1; a.ntiles = a.ptiles;
    a.list = nullptr; a.list_count = nullptr;
    if ((long long)a.ptiles * (Cin / 64) >= (1LL << 29)) return CVPCE_ERR_ARG;
    if (!cvpce_smem_attr_done<conv3x3_halo3_kernel<E, false, false, true>>((const void*)conv3x3_halo3_kernel<E, false, false, true>, 3 * G3_A_BYTES)) return CVPCE_ERR_LAUNCH;
    const int grid = a.ntiles < g_cvpce_persistent_wgs ? a.ntiles : g_cvpce_persistent_wgs;
    hipLaunchKernelGGL((conv3x3_halo3_kernel<E, false, false, true>), dim3(grid), dim3(512), 3 * G3_A_BYTES, (hipStream_t)stream, a);
    return cvpce_check_launch();
}
extern "C" int cvpce_conv3x3_halo_thin_out(const void* in, const void* wgt, const float* bias, float* out, int N, int H, int W,
                                           int Cin, int Cout, int K_pad, int Cout_pad, void* stream) {
    return halo3_thin_dispatch<ElemBF16>(in, wgt, bias, out, N, H, W, Cin, Cout, K_pad, Cout_pad, stream);
}
extern "C" int cvpce_conv3x3_halo_thin_out_f16(const void* in, const void* wgt, const float* bias, float* out, int N, int H, int W,
                                               int Cin, int Cout, int K_pad, int Cout_pad, void* stream) {
    return halo3_thin_dispatch<ElemF16>(in, wgt, bias, out, N, H, W, Cin, Cout, K_pad, Cout_pad, stream);
}

extern "C" int cvpce_conv3x3_halo_wide(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W,
                                       int Cin, int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, void* stream) {
    return halo3_dispatch<ElemBF16>(in, wgt, bias, out, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, fuse_pool2, stream);
}
extern "C" int cvpce_conv3x3_halo_wide_f16(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W,
                                           int Cin, int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, void* stream) {
    return halo3_dispatch<ElemF16>(in, wgt, bias, out, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, fuse_pool2, stream);
}

// work-list launch (called by cvpce_conv3x3_halo_list for Cout <= 128; not part of the C ABI of its own)
int cvpce_conv3x3_halo_wide_list(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W, int Cin,
                                 int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, const unsigned long long* list,
                                 const int* count_dev, void* stream) {
    if (!list || !count_dev) return CVPCE_ERR_ARG;
    return halo3_dispatch<ElemBF16>(in, wgt, bias, out, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, fuse_pool2, stream, list, count_dev);
}
