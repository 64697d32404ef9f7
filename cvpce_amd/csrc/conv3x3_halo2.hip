// 3x3 / stride 1 / pad 1 convolution, Cin % 64 == 0, Cout > 128: halo patch in LDS, WEIGHTS STRAIGHT FROM L2 INTO
// REGISTERS, patch rows streamed ONCE per (kw, K-half) and used for all three kh (VGG16 conv3_1 ... conv5_3, RetinaNet
// head towers; third form of the halo kernel).
//
// What bounds the LDS-ring kernels (profiles/r01_ablation_conv4_2.md) is the per-K-step hand-off: every 32 k all 8 waves
// meet at a barrier and issue their DMA pieces in lock step.  Here a workgroup (8 waves) owns a 16x16-pixel x 256-cout
// tile and
//   * wave w owns 32 couts x 256 pixels: its weights are needed by no other wave, so it loads its own A fragments with
//     buffer_load_dwordx4 in MFMA layout (lane (m, q) <- 16 B of row m), one step ahead, into registers -- no LDS ring;
//   * the 18x18x64 input patch of a channel chunk is triple-buffered in LDS (LDS-DMA, zero-filled borders); the only
//     workgroup barrier is the patch hand-off once per chunk, so the two waves of a SIMD drift apart and cover each other;
//   * ROW STREAMING: patch row p shifted by kw is the B operand of output row p (tap kh = 0), p-1 (kh = 1) and p-2
//     (kh = 2).  A step fixes (kw, K-half), holds the 3 x 2 weight fragments of the three kh in registers, reads each of
//     the 18 patch rows ONCE (ds_read_b128, immediate offsets, 4-deep register ring) and issues up to 6 MFMAs on it:
//     108 fragment reads per chunk instead of 288 -- the reads were 21 % of conv4_2 (profiles/r01d_ablation_halo2.md).
// Per chunk and wave: 6 steps x (18 ds_read_b128 + 96 MFMA 16x16x32 + 6 weight loads).
// Weight layout: the fragment-major "halo weight layout" of include/cvpce_amd.h.  Fused bias / ReLU / MaxPool2d(2,2).
#include "common.h"
#include <stdlib.h>
#include "../../include/cvpce_amd.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

// compile-time timing experiments (never set in the shipped library; tools/ablate.sh): 1 no s_setprio around the MFMA
// groups, 2 XCD-aware tile order, 4 no patch DMA in the loop, 8 no weight loads in the loop, 16 no output stores (runtime-false predicate),
// 32 no fragment reads in the loop, 64 no hand-off barrier (races: timing only), 256 (dispatch) 32-aligned maps to the wide kernel,
// 512 in-kernel clock stamps (below; results unchanged)
#ifndef CVPCE_DBG
#define CVPCE_DBG 0
#endif

#ifndef G2_LIST_DEEP_WEIGHTS
#define G2_LIST_DEEP_WEIGHTS 0     // 1: work-list launches prefetch the weight fragments TWO steps ahead (3 slots, 246-256 VGPRs, no spill).
                                   // Measured with the row cut on, same call, three alternations: 27.92 / 27.92 / 27.97 ms (one step ahead)
                                   // vs 27.95 / 27.98 / 28.12 ms per 1 600 crops -- the cut tiles are bound by the weight stream's BYTES
                                   // (conv4_2 without any weight load: 3.99 -> 2.85 ms with the cut; with: 4.81 -> 4.33), not by its latency
#endif
#define G2_T 16
#define G2_P 18
#define G2_NPIX 324
#define G2_A_BYTES (328 * 128)
#define G2_NPIECE 41
// masked launches (level atlas): the workgroup's own tile list and the tiles' pixel masks, built once in LDS
#define G2_MAX_SEQ 512                                   // tiles per workgroup of a masked launch (checked on the host)
#define G2_LTILE_OFF (3 * G2_A_BYTES)                    // [G2_MAX_SEQ] (ty << 16) | tx
#define G2_LROW_OFF (G2_LTILE_OFF + G2_MAX_SEQ * 4)      // [G2_MAX_SEQ][16] one bit per pixel of a tile row
#define G2_SMEM_MASKED (G2_LROW_OFF + G2_MAX_SEQ * 32)
// work-list launches (the embedder's constant-padding tile skipping, skiplist.hip): the first G2_MAX_SEQ entries of the
// workgroup's own list in LDS (later ones, if any, are read from global memory where they are needed)
#define G2_LLIST_OFF (3 * G2_A_BYTES)                    // [G2_MAX_SEQ] 8-byte entries
#define G2_SMEM_LIST (G2_LLIST_OFF + G2_MAX_SEQ * 8)

// Patch swizzle: 16-byte chunk c of patch pixel (py, px) lives at physical chunk
// c ^ g2_swz(py, px).  ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, ...), i.e. a
// group mixes two K-quarters over complementary halves of the 16 pixel lanes; with the lane -> pixel maps below every
// tap's fragment read is conflict-free at any alignment.
__device__ __forceinline__ int g2_swz0(int u) { return ((u & 3) << 1) | ((u >> 2) & 1); }
__device__ __forceinline__ int g2_swz(int py, int px) { return g2_swz0(px >> 1) ^ (py & 1); }
__device__ __forceinline__ int g2_col(int l16) { return l16 < 4 ? 2 * l16 : (l16 >= 12 ? 2 * (l16 - 8) : 2 * (l16 - 4) + 1); }

struct Halo2Args {
    const bf16_t* in;    // [N][H][W][Cin]
    const bf16_t* wgt;   // fragment-major (include/cvpce_amd.h): [chunk][32-cout group][kw][K-half][kh][block][lane][8]
    const float* bias;
    const unsigned char* mask;   // optional [H][W]: output pixels with mask 0 are stored as zeros (atlas gaps); not with POOL
    bf16_t* out;         // [N][H][W][Cout] or pooled [N][H/2][W/2][Cout]; null with gmax: nothing is stored (conv5_3)
    const int* tile_map; // optional: the pixel tiles to compute, (ty << 16) | tx, the same list for every image (level atlas:
                         // tiles that lie wholly in a gap are skipped; their outputs keep the zeros the buffer was created with)
    int tiles_per_image; // entries of tile_map (with a map), else tiles_x * tiles_y
    float* gmax;         // optional MAC descriptor [N][gmax_stride]: gmax[n][gmax_off + co] = max over the map (relu = 1)
    int gmax_stride, gmax_off;
    int N, H, W, Cin, Cout, K_pad, relu;
    int cgroups;         // Cout_pad / 32: 32-cout groups per channel chunk of the fragment-major weights
    int tiles_x, tiles_y, ptiles, ctiles, ntiles;
    unsigned in_bytes, wgt_bytes;
    // LIST launches: the tiles to compute, ((ey_in << 16 | ex_in) << 32) | (n << 16) | (ty << 8) | tx, crop-major; *list_count
    // entries (device-resident); input pixels with y >= ey_in or x >= ex_in are read from image N - 1 (the constant crop)
    const unsigned long long* list;
    const int* list_count;
    // PAIRED launches (the RetinaNet head's two towers as ONE launch, cvpce_conv3x3_halo_masked_paired): cout tile ct is tower ct.
    // in_group_bytes != 0: the input is [towers][N][H][W][Cin] and tile ct reads tower ct's part (0: every tile reads the one input);
    // out_group_elems != 0: the output is [towers][N][H][W][256] (tower ct's 256 couts of the Cout = 256 x towers), not [N][H][W][Cout]
    unsigned in_group_bytes;
    long long out_group_elems;
};

#if CVPCE_DBG & 512
// diagnostic build only (tools/ablate.sh conv3x3_halo2 512; tools/dev/kernel_clock.py): wave 0 of every workgroup stamps the shader-clock
// counter (s_memtime) and the 100-MHz constant counter (s_memrealtime) around its whole K loop; their quotient is the clock the
// kernel actually ran at (MI355X_MICROARCH.md, DVFS give-back item 6).  No output value depends on the stamps.
__device__ unsigned long long cvpce_halo2_clock[1024][2];
extern "C" int cvpce_debug_halo2_clock(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cvpce_halo2_clock), sizeof(cvpce_halo2_clock)) == hipSuccess ? 0 : 2;
}
#endif

// STRIP (work-list launches only): a "tile" is THREE strips -- the first 4 output rows of three listed tiles whose other rows are
// constant (`rows` == 4: a crop's content ends just below a tile boundary).  Run one by one such tiles issue a third of a full
// tile's MFMAs against a full tile's weight stream, which is what bounds them (profiles/r04_rejected_experiments.md); here patch
// rows 6s .. 6s + 5 of the 18-row patch belong to strip s, accumulator rows 4s .. 4s + 3 are its outputs, and the three share
// every weight fragment: 72 MFMAs per step and wave for three tiles' worth of useful rows.
// NW (waves = 32-cout groups per workgroup): 8 -> a 256-cout tile; 4 -> a 128-cout tile for launches with too few pixel tiles to fill
// the chip (the detector's 50 x 50 and 25 x 25 maps: 128 / 32 tiles of 16 x 16 pixels for 256 compute units): twice the workgroups,
// each with half the weight stream and the same patch.
template <typename E, bool POOL, bool GMAX, bool LIST, bool STRIP = false, int NW = 8>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void conv3x3_halo2_kernel(Halo2Args a) {
    static_assert(!STRIP || LIST, "strips come from a work list");
    static_assert(NW == 8 || NW == 4, "cout waves");   // (two waves on a 64-cout tile for the 25 x 25 maps: measured, no further gain)
    constexpr int TC = 32 * NW, NB = 16, NS = STRIP ? 3 : 1;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ap = smem;                 // [3][328][64] bf16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave = 32-cout group
    const int l16 = lane & 15, lq = lane >> 4;

    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_p = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);

    const int nchunks = a.Cin >> 6;
    const int lbid = (CVPCE_DBG & 2) ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    // Which tiles this workgroup computes.  Plain launches: tile lbid, lbid + grid, ... (all tiles cost the same).  LIST launches:
    // a CONTIGUOUS block of the list per workgroup -- listed tiles differ in cost when they are cut at `rows`, and the list is
    // crop-major with a fixed number of tiles per crop, so a stride that is a multiple of it would hand one workgroup nothing but
    // bottom-row (cut) tiles and another nothing but full ones.  The cout tile stays fixed per workgroup (lbid % ctiles: one
    // weight set per workgroup, and with an even grid per XCD), its group's workgroups split the list entries among them.
    int my_tiles, l_first = 0, l_entries = 0;
    if constexpr (LIST) {
        l_entries = __builtin_amdgcn_readfirstlane(*a.list_count);             // <= a.ptiles (checked by the list builder)
        const int units = STRIP ? (l_entries + 2) / 3 : l_entries;             // STRIP: three list entries make one tile
        const int groups = (int)gridDim.x / a.ctiles;                          // workgroups per cout tile (the host launches grid % ctiles == 0)
        const int per = (units + groups - 1) / groups;
        l_first = (lbid / a.ctiles) * per;
        my_tiles = units - l_first < per ? units - l_first : per;
    } else {
        my_tiles = (a.ntiles - lbid + (int)gridDim.x - 1) / (int)gridDim.x;
    }
    if (my_tiles <= 0) return;
    const int total_chunks = my_tiles * nchunks;          // < 2^30: checked on the host
    unsigned long long* llist = reinterpret_cast<unsigned long long*>(smem + G2_LLIST_OFF);
    if constexpr (LIST) {
        const int staged = my_tiles * NS < G2_MAX_SEQ ? my_tiles * NS : G2_MAX_SEQ;
        for (int idx = tid; idx < staged; idx += 64 * NW) {
            const int k = l_first * NS + idx;                                    // (STRIP: past the end of the list the last entry is repeated --
            llist[idx] = a.list[k < l_entries ? k : l_entries - 1];              //  a strip computed twice stores the same values twice)
        }
        __syncthreads();
    }

    // ---- masked launches: tile coordinates and per-row pixel masks of THIS workgroup's tiles go to LDS once.  Read from global
    //      memory where they are used (a tile-map entry per tile change, a mask byte per stored row) every one of those loads
    //      sits in the vmcnt queue behind the epilogue's stores and its wait drains them: 16 round trips per tile ----
    unsigned* ltile = reinterpret_cast<unsigned*>(smem + G2_LTILE_OFF);
    unsigned short* lrow = reinterpret_cast<unsigned short*>(smem + G2_LROW_OFF);
    if (a.mask) {
        for (int idx = tid; idx < my_tiles * 16; idx += 64 * NW) {
            const int sq = idx >> 4, y = idx & 15;
            const int r = ((lbid + sq * (int)gridDim.x) / a.ctiles) % a.tiles_per_image;
            int ty, tx;
            if (a.tile_map) {
                const int packed = a.tile_map[r];
                ty = packed >> 16;
                tx = packed & 0xFFFF;
            } else {
                ty = r / a.tiles_x;
                tx = r - ty * a.tiles_x;
            }
            const int oy = ty * G2_T + y;
            unsigned bits = 0;
            if (oy < a.H) {
#pragma unroll
                for (int c = 0; c < G2_T; ++c) {
                    const int ox = tx * G2_T + c;
                    if (ox < a.W && a.mask[oy * a.W + ox]) bits |= 1u << c;
                }
            }
            lrow[idx] = (unsigned short)bits;
            if (y == 0) ltile[sq] = (unsigned)((ty << 16) | tx);
        }
        __syncthreads();
    }

    // tile seq -> (image, tile row, tile column, cout tile); cout tile fastest.  ext (LIST): the crop's extents on the INPUT tensor
    auto tile_of = [&](int seq, int* n, int* ty, int* tx, int& ct, int* ext) {
        if constexpr (LIST) {
            ct = lbid % a.ctiles;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int k = seq * NS + s, g = l_first * NS + k;
                const unsigned long long e = k < G2_MAX_SEQ ? llist[k] : a.list[g < l_entries ? g : l_entries - 1];
                const int lo = __builtin_amdgcn_readfirstlane((int)(unsigned)e);
                ext[s] = __builtin_amdgcn_readfirstlane((int)(unsigned)(e >> 32));
                n[s] = lo >> 16;
                ty[s] = (lo >> 8) & 0xFF;
                tx[s] = lo & 0xFF;
            }
            return;
        }
        ext[0] = 0;
        const int t = lbid + seq * (int)gridDim.x;
        ct = t % a.ctiles;
        const int p = t / a.ctiles;
        n[0] = p / a.tiles_per_image;
        const int r = p - n[0] * a.tiles_per_image;
        if (a.mask) {
            const int packed = __builtin_amdgcn_readfirstlane((int)ltile[seq]);
            ty[0] = packed >> 16;
            tx[0] = packed & 0xFFFF;
        } else {
            ty[0] = r / a.tiles_x;
            tx[0] = r - ty[0] * a.tiles_x;
        }
    };

    // ---- patch DMA: piece j fills patch rows 8j .. 8j+7 (row = lane>>3, phys chunk = lane&7); pieces dealt
    //      round-robin to the NW waves (wave w: pieces w, w + NW, ...; 41 pieces: 6 for w = 0, else 5 with eight waves) ----
    constexpr int NPP = (41 + NW - 1) / NW;
    const int npp = (41 - wc + NW - 1) / NW;
    auto issue_patch = [&](const int* n, const int* ty, const int* tx, int c, int buf, const int* ext, int ct) {
        // lane id recomputed here (2 VALU ops, once per patch) instead of living in a VGPR across the K loop; the volatile
        // asm also keeps the per-piece constants below from being hoisted out of the chunk loop (18+ VGPRs)
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            if (i < npp) {
                const int j = wc + NW * i;
                const int pp = j * 8 + (ln >> 3);
                const int py = pp / G2_P, px = pp - py * G2_P;
                const int lchunk = (ln & 7) ^ g2_swz(py, px);
                // the tile this patch pixel belongs to: the one tile, or (STRIP) strip py / 6, whose patch rows are 6 s .. 6 s + 5
                int sn = n[0], sty = ty[0], stx = tx[0], sext = ext[0], ry = py;
                if constexpr (STRIP) {
                    const int st = py >= 12 ? 2 : (py >= 6 ? 1 : 0);
                    ry = py - 6 * st;
                    sn = st == 0 ? n[0] : (st == 1 ? n[1] : n[2]);
                    sty = st == 0 ? ty[0] : (st == 1 ? ty[1] : ty[2]);
                    stx = st == 0 ? tx[0] : (st == 1 ? tx[1] : tx[2]);
                    sext = st == 0 ? ext[0] : (st == 1 ? ext[1] : ext[2]);
                }
                const int ey = (sext >> 12) & 0xFFF, ex = sext & 0xFFF;
                const int y = sty * G2_T - 1 + ry, x = stx * G2_T - 1 + px;
                const bool ok = pp < G2_NPIX && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                int nn = sn;
                if constexpr (LIST) nn = (y >= ey || x >= ex) ? a.N - 1 : sn;   // constant region of the crop: the constant crop's pixel
                const unsigned off = (unsigned)((((size_t)(nn * a.H + y) * a.W + x) * a.Cin + c * 64) * 2) + (unsigned)(lchunk * 16) + (unsigned)ct * a.in_group_bytes;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_p, (lds_void*)(Ap + buf * G2_A_BYTES + j * 1024), 16,
                                                         (int)(ok ? off : 0xFFFFFFF0u), 0, 0, 0);
            }
        }
    };
    // patch issue pointer: the next flat chunk to fetch and its tile
    int pi = 0, pi_seq = 0, pi_c = 0, pi_buf = 0, pi_n[NS], pi_ty[NS], pi_tx[NS], pi_ct, pi_ext[NS];
    tile_of(0, pi_n, pi_ty, pi_tx, pi_ct, pi_ext);
    auto issue_next_patch = [&]() {
        if (pi < total_chunks) {
            if (!(CVPCE_DBG & 4) || pi < 2) issue_patch(pi_n, pi_ty, pi_tx, pi_c, pi_buf, pi_ext, pi_ct);
            ++pi;
            if (++pi_buf == 3) pi_buf = 0;
            if (++pi_c == nchunks) {
                pi_c = 0;
                ++pi_seq;
                if (pi_seq < my_tiles) tile_of(pi_seq, pi_n, pi_ty, pi_tx, pi_ct, pi_ext);
            }
        }
    };

    // ---- weights: lane (m = l16, q = lq) loads 16 B = k 8q .. 8q+7 of row m of a 16-cout block.
    // MFMA row m = 4q' + j of block mt is cout 8q' + 4mt + j of the wave's 32: after both blocks accumulator lane group
    // q' holds 8 CONSECUTIVE couts (8q' .. 8q'+7) of its pixel -> one 16-byte store per pixel block ----
    // The weights arrive FRAGMENT-MAJOR (include/cvpce_amd.h, "halo weight layout"): the 16 bytes lane L needs for the fragment
    // (chunk c, 32-cout group, kw, K-half, kh, block mt) sit at byte 16 L of that fragment's 1 KiB block, so a wave's weight load
    // is ONE contiguous KiB.  From the row-major [Cout_pad][K_pad] layout the same load touched 16 rows with no two neighbouring
    // lanes in one 64-byte block, and the texture addresser took one lane per clock for it (~61 clocks per load against 16:
    // measured on the pointwise kernel, csrc/conv1x1.hip) -- the six loads per step of each of the eight waves kept it busy for
    // about as long as the step's MFMAs run.
    const unsigned voff1 = (unsigned)(lane * 16);
    // scalar byte offset of the 36 blocks of (cout tile ct, channel chunk c) of this wave's 32 couts
    auto wbase = [&](int ct, int c) { return __builtin_amdgcn_readfirstlane((int)((unsigned)(c * a.cgroups + ct * (TC / 32) + wc) * 36864u)); };

    // ---- pixel fragments: lane (l16, lq) reads pixel (row p, column g2_col(l16) + kw), K-quarter lq of K-half hf:
    //      address = ((c3[kw] ^ (hf << 6) ^ ((p & 1) << 4)) + buffer) + (p * 18 + kw) * 128 ----
    unsigned c3[3];
    {
        const int col = g2_col(l16);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) c3[kw] = (unsigned)(col * 128) ^ (unsigned)((lq ^ g2_swz0(((col + kw) >> 1) & 7)) << 4);
    }
    const unsigned lds_a = (unsigned)(size_t)(lds_char*)Ap;

    // accumulators start from the bias of the tile's couts: the epilogue then needs no load
    auto load_bias = [&](int ct, f32x4* b) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int co = ct * TC + wc * 32 + 8 * lq + 4 * mt;
            b[mt] = (a.bias && co < a.Cout) ? *reinterpret_cast<const f32x4*>(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 acc[2][NB];         // [16-cout block][output row]
    {
        f32x4 b0[2];
        load_bias(pi_ct, b0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = b0[i];
    }

    // weight slots: the next step's fragments are fetched while this one computes (2 slots); G2_LIST_DEEP_WEIGHTS = 1 makes it two
    // steps ahead for work-list launches (3 slots) -- built, measured, not faster (see the macro)
    constexpr int NA = (LIST && G2_LIST_DEEP_WEIGHTS) ? 3 : 2;
    bf16x8 af[NA][3][2];      // [step % NA][kh][16-cout block]
    bf16x8 bfr[4];            // patch-row ring
    unsigned e0, e1;          // fragment base addresses of the step being fetched (even / odd patch rows)

    // step T = kw * 2 + hf of a chunk (0..5): weights of the three taps (kh, kw), K-half hf
#define G2_LOAD_A(T, SBASE)                                                                                    \
    if constexpr (!(CVPCE_DBG & 8)) {                                                                          \
        _Pragma("unroll") for (int kh_ = 0; kh_ < 3; ++kh_)                                                    \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 2; ++mt_) {                                              \
                const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(srd_w, voff1, (SBASE) + (((T) * 3 + kh_) * 2 + mt_) * 1024, 0); \
                af[(T) % NA][kh_][mt_] = __builtin_bit_cast(bf16x8, v_);                                       \
            }                                                                                                  \
    }
    // at the start of step T: the weights of the step NA - 1 ahead (steps 6, 7 = steps 0, 1 of the next chunk)
#define G2_PREFETCH_A(T)                                                                                       \
    if constexpr (NA == 2) {                                                                                   \
        if constexpr ((T) < 5) { G2_LOAD_A((T) + 1, sb_cur) } else { G2_LOAD_A(0, sb_next) }                   \
    } else {                                                                                                   \
        if constexpr ((T) < 4) { G2_LOAD_A((T) + 2, sb_cur) } else { G2_LOAD_A((T) - 4, sb_next) }             \
    }
#define G2_SET_E(T, BUFB)                                                                                      \
    {                                                                                                          \
        const unsigned x_ = c3[(T) >> 1] ^ (unsigned)(((T) & 1) << 6);                                         \
        e0 = x_ + (BUFB);                                                                                      \
        e1 = (x_ ^ 16u) + (BUFB);                                                                              \
    }
    // read patch row P of step T into ring slot (P + 2 T) & 3 (a step has 18 rows, 18 = 2 mod 4)
#define G2_READ(T, P)                                                                                          \
    if constexpr (!(CVPCE_DBG & 32)) {                                                                         \
        if constexpr (((P) & 1) == 0)                                                                          \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[((P) + 2 * (T)) & 3]) : "v"(e0), "n"(((P) * G2_P + ((T) >> 1)) * 128)); \
        else                                                                                                   \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[((P) + 2 * (T)) & 3]) : "v"(e1), "n"(((P) * G2_P + ((T) >> 1)) * 128)); \
    }
    // patch row P through tap KH feeds output row P - KH -- or (STRIP) row (P % 6) - KH < 4 of strip P / 6 = accumulator row 4 (P / 6) + ...
#define G2_ROW_KH(T, P, KH)                                                                                    \
    {                                                                                                          \
        constexpr int j_ = STRIP ? (P) % 6 - (KH) : (P) - (KH);                                                \
        constexpr int r_ = STRIP ? 4 * ((P) / 6) + j_ : j_;                                                    \
        if constexpr (j_ >= 0 && j_ < (STRIP ? 4 : NB)) {                                                      \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 2; ++mt_)                                                \
                acc[mt_][r_] = E::mfma16(af[(T) % NA][KH][mt_], bfr[((P) + 2 * (T)) & 3], acc[mt_][r_]);       \
        }                                                                                                      \
    }
    // the MFMAs of patch row P: output rows P (kh = 0), P-1 (kh = 1), P-2 (kh = 2) where they exist.  Waits until the
    // row has landed (the two rows prefetched after it may stay in flight); the "+v" tie keeps the compiler from
    // hoisting an MFMA above the wait.
#define G2_ROW(T, P)                                                                                           \
    {                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bfr[((P) + 2 * (T)) & 3]));                                 \
        G2_ROW_KH(T, P, 0) G2_ROW_KH(T, P, 1) G2_ROW_KH(T, P, 2)                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
    // rows 0..15 of step T: prefetch row P + 2, compute row P
#define G2_RP(T, P) G2_READ(T, (P) + 2) G2_ROW(T, P)
    // the chunk hand-off: this wave is done with the PREVIOUS chunk's buffer and (vmcnt) its own pieces of the NEXT chunk's
    // patch have landed -- at most the 6 weight loads just issued are younger than those pieces; then the DMA of the chunk
    // after the next one goes into the buffer the previous chunk used
#define G2_HANDOFF()                                                                                           \
    {                                                                                                          \
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                       \
        if (!(CVPCE_DBG & 64)) __builtin_amdgcn_s_barrier();                                                   \
        issue_next_patch();                                                                                    \
    }
    // LIST launches: a tile computes only its first `rows_` (4 | 8 | 12 | 16) output rows -- the rest lies in the crop's constant
    // region and is never read by anyone.  Output row r needs patch rows r .. r + 2, so a step streams patch rows 0 .. rows_ + 1
    // and stops (three uniform branches per step); the two reads it has prefetched beyond are stale and land, in order, before
    // the next step's rows 0 and 1 that follow them into the same ring slots.  Accumulator rows rows_, rows_ + 1 collect partial
    // sums nobody stores.  BEFORE / AFTER: what precedes the step's hand-over to the next one (the chunk hand-off in the last step).
#define G2_ROWS_AND_TAIL(T, TN, BUFB_NEXT, BEFORE)                                                             \
    G2_RP(T, 0) G2_RP(T, 1) G2_RP(T, 2) G2_RP(T, 3) G2_RP(T, 4) G2_RP(T, 5)                                    \
    if (!LIST || STRIP || rows_ > 4) { G2_RP(T, 6) G2_RP(T, 7) G2_RP(T, 8) G2_RP(T, 9) }                       \
    if (!LIST || STRIP || rows_ > 8) { G2_RP(T, 10) G2_RP(T, 11) G2_RP(T, 12) G2_RP(T, 13) }                   \
    if (!LIST || STRIP || rows_ > 12) {                                                                        \
        G2_RP(T, 14) G2_RP(T, 15)                                                                              \
        BEFORE                                                                                                 \
        G2_SET_E(TN, BUFB_NEXT)                                                                                \
        G2_READ(TN, 0) G2_ROW(T, 16)                                                                           \
        G2_READ(TN, 1) G2_ROW(T, 17)                                                                           \
    } else {                                                                                                   \
        BEFORE                                                                                                 \
        G2_SET_E(TN, BUFB_NEXT)                                                                                \
        G2_READ(TN, 0)                                                                                         \
        G2_READ(TN, 1)                                                                                         \
    }
    // step T < 5: fetch the next step's weights, stream the rows; rows 16, 17 prefetch rows 0, 1 of step T + 1
#define G2_STEP(T)                                                                                             \
    {                                                                                                          \
        G2_PREFETCH_A(T)                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        G2_ROWS_AND_TAIL(T, (T) + 1, bufb, )                                                                   \
    }

    // ---- prologue ----
    int seq = 0, cchunk = 0, t_n[NS], t_ty[NS], t_tx[NS], t_ct, t_ext_[NS];
    tile_of(0, t_n, t_ty, t_tx, t_ct, t_ext_);
    int rows_ = (LIST && !STRIP) ? ((t_ext_[0] >> 24) & 0xFF) : NB;     // output rows of the current tile that are computed (LIST: 4 | 8 | 12 | 16)
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
    {
        constexpr int dbg8_ = CVPCE_DBG & 8;            // (the ablation build still loads the first step's weights)
        _Pragma("unroll") for (int kh_ = 0; kh_ < 3; ++kh_)
            _Pragma("unroll") for (int mt_ = 0; mt_ < 2; ++mt_) {
                const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(srd_w, voff1, sb_cur + (kh_ * 2 + mt_) * 1024, 0);   // step 0
                af[0][kh_][mt_] = __builtin_bit_cast(bf16x8, v_);
                if (dbg8_) af[1][kh_][mt_] = af[0][kh_][mt_];
            }
        if constexpr (NA == 3) { G2_LOAD_A(1, sb_cur) }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned bufb = lds_a;                               // LDS base of the current chunk's patch buffer
    int bufi = 0;
    G2_SET_E(0, bufb)
    G2_READ(0, 0)
    G2_READ(0, 1)

    const int lp = lane & 15;
#if CVPCE_DBG & 512
    const unsigned long long ck0_ = __builtin_amdgcn_s_memtime(), rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
    for (int cc = 0; cc < total_chunks; ++cc) {
        // keep the per-step address variants inside the loop: hoisted they cost VGPRs the accumulators need
        asm volatile("" : "+v"(c3[0]), "+v"(c3[1]), "+v"(c3[2]));
        G2_STEP(0) G2_STEP(1) G2_STEP(2) G2_STEP(3) G2_STEP(4)
        // ---- last step of the chunk, with the chunk hand-off before its rows 16, 17 prefetch from the NEXT buffer ----
        {
            G2_PREFETCH_A(5)             // step 0 (NA = 3: step 1) of the next chunk; past the last chunk a harmless reload
            __builtin_amdgcn_sched_barrier(0);
            // After the last chunk the barrier, the reads and the weight loads still run -- on valid, unused data -- so
            // that the loop body has one shape and the accumulators stay in place.
            const int nbufi = (bufi == 2) ? 0 : bufi + 1;
            const unsigned nbufb = lds_a + (unsigned)nbufi * G2_A_BYTES;
            G2_ROWS_AND_TAIL(5, 0, nbufb, G2_HANDOFF())  // hand-off: chunk cc + 2 -> the buffer chunk cc - 1 used
            bufb = nbufb;
            bufi = nbufi;
        }
        sb_cur = sb_next;

        if (cchunk + 1 == nchunks) {
            // ---- epilogue of this tile (the next tile's patch, weights and first rows are already in flight) ----
            const int ct = t_ct;
            const int rows_t = STRIP ? 4 : rows_;   // (LIST) rows of this tile (STRIP: of each strip) that were computed: the others are not stored
            // accumulator row nt belongs to output row nt of the tile -- or (STRIP) to output row nt % 4 of strip nt / 4
            auto e_n = [&](int nt) { return STRIP ? t_n[(nt >> 2) < NS ? (nt >> 2) : 0] : t_n[0]; };
            auto e_ty = [&](int nt) { return STRIP ? t_ty[(nt >> 2) < NS ? (nt >> 2) : 0] : t_ty[0]; };
            auto e_tx = [&](int nt) { return STRIP ? t_tx[(nt >> 2) < NS ? (nt >> 2) : 0] : t_tx[0]; };
            auto e_row = [&](int nt) { return STRIP ? (nt & 3) : nt; };
            auto e_live = [&](int nt) { return STRIP ? nt < 12 : (!LIST || nt < rows_t); };
            f32x4 nbias[2];                      // bias of the NEXT tile's couts: lands while this tile is stored
            load_bias(n_ct, nbias);
            const int col = g2_col(lp);
            const int co = ct * TC + wc * 32 + 8 * lq;            // this lane's 8 consecutive couts
            if constexpr (GMAX) {
                // MAC descriptor fused (classification.py:46-49 `x.amax(dim=(-2, -1))` of the post-ReLU map): maximum over
                // the tile's 256 pixels per cout -- over the lane's 16 rows in registers, over the 16 pixel lanes of a DPP
                // row by rotates -- then one atomic max per (tile, cout) into the zero-initialised descriptor.  Values are
                // >= 0 after ReLU, so they order like their bit patterns; rounding to bf16 is monotonic, so the maximum
                // of the fp32 values rounded once equals the maximum of the stored bf16 map, bit for bit.
#pragma unroll
                for (int sg = 0; sg < NS; ++sg) {                    // (STRIP: one maximum per strip = per crop)
                    unsigned m[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) m[j] = 0u;
                    const int n = t_n[sg], ty = t_ty[sg], tx = t_tx[sg];
                    if (tx * G2_T + col < a.W) {
#pragma unroll
                        for (int nt = STRIP ? 4 * sg : 0; nt < (STRIP ? 4 * sg + 4 : NB); ++nt)
                            if (ty * G2_T + e_row(nt) < a.H && e_live(nt)) {  // (LIST: the rows this tile computed; the rest is cvpce_mac_init's)
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    m[j] = max(m[j], __float_as_uint(relu_bits(acc[0][nt][j])));
                                    m[4 + j] = max(m[4 + j], __float_as_uint(relu_bits(acc[1][nt][j])));
                                }
                            }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        unsigned v = m[j];
                        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false));   // row_ror:8
                        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false));   // row_ror:4
                        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, false));   // row_ror:2
                        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, false));   // row_ror:1
                        m[j] = v;
                    }
                    if (lp == 0) {
                        unsigned* g = reinterpret_cast<unsigned*>(a.gmax) + (size_t)n * a.gmax_stride + a.gmax_off + co;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (co + j < a.Cout) atomicMax(g + j, __float_as_uint(E::widen(E::narrow(__uint_as_float(m[j])))));
                    }
                }
            }
            if (GMAX && a.out == nullptr) {
                // the map itself has no consumer (conv5_3: only its MAC descriptor is used): nothing to store
            } else if (POOL) {
                // rows 2i, 2i+1 are accumulator rows of the same lane; columns 2k, 2k+1 are lanes A[k], B[k] with
                // A = {0-3,12-15}, B = {4-11}: lane A[k] takes its right neighbour by a row rotate (+4 for lanes 0-3,
                // -4 for lanes 12-15; bank masks 1 and 8)
#pragma unroll
                for (int i = 0; i < NB / 2; ++i) {
                    const int ox = e_tx(2 * i) * (G2_T / 2) + (col >> 1);
                    const bool lane_ok = (lp < 4 || lp >= 12) && ox < (a.W >> 1);
                    const int oy = e_ty(2 * i) * (G2_T / 2) + (e_row(2 * i) >> 1);
                    const size_t opix = (size_t)(e_n(2 * i) * (a.H >> 1) + oy) * (a.W >> 1) + ox;
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
                    if (lane_ok && oy < (a.H >> 1) && co < a.Cout && e_live(2 * i) && (!(CVPCE_DBG & 16) || a.relu == 12345)) {
                        const uint2 l2 = __builtin_bit_cast(uint2, E::pack4(f32x4{r[0], r[1], r[2], r[3]}));
                        const uint2 h2 = __builtin_bit_cast(uint2, E::pack4(f32x4{r[4], r[5], r[6], r[7]}));
                        *reinterpret_cast<u32x4*>(a.out + opix * a.Cout + co) = u32x4{l2.x, l2.y, h2.x, h2.y};
                    }
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NB; ++nt) {
                    const int ox = e_tx(nt) * G2_T + col;
                    const int oy = e_ty(nt) * G2_T + e_row(nt);
                    const size_t opix = (size_t)(e_n(nt) * a.H + oy) * a.W + ox;
                    const bool store_lane = oy < a.H && ox < a.W && e_live(nt);     // ragged right / bottom tiles; (LIST) computed rows only
                    bool keep = true;                // masked-out pixels (gaps of a level atlas) are stored as zeros
                    if (a.mask) keep = ((lrow[seq * 16 + nt] >> col) & 1u) != 0;
                    f32x4 r0 = acc[0][nt], r1 = acc[1][nt];
                    if (a.relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { r0[j] = relu_bits(r0[j]); r1[j] = relu_bits(r1[j]); }
                    }
                    if (store_lane && co < a.Cout && (!(CVPCE_DBG & 16) || a.relu == 12345)) {
                        const uint2 l2 = __builtin_bit_cast(uint2, E::pack4(r0)), h2 = __builtin_bit_cast(uint2, E::pack4(r1));
                        bf16_t* dst = a.out_group_elems ? a.out + (size_t)ct * a.out_group_elems + opix * TC + (co - ct * TC) : a.out + opix * a.Cout + co;
                        *reinterpret_cast<u32x4*>(dst) = keep ? u32x4{l2.x, l2.y, h2.x, h2.y} : u32x4{0u, 0u, 0u, 0u};
                    }
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NB; ++nt) acc[mt][nt] = nbias[mt];
            cchunk = 0;
            ++seq;
            if (seq < my_tiles) {
                tile_of(seq, t_n, t_ty, t_tx, t_ct, t_ext_);
                if constexpr (LIST && !STRIP) rows_ = (t_ext_[0] >> 24) & 0xFF;
            }
        } else {
            ++cchunk;
        }
        // weights base of the chunk after the (new) current one
        n_ct = next_ct();
        sb_next = wbase(n_ct, (cchunk + 1 < nchunks) ? cchunk + 1 : 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the trailing prefetch
#if CVPCE_DBG & 512
    if (wc == 0 && lane == 0 && blockIdx.x < 1024) {
        cvpce_halo2_clock[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - ck0_;
        cvpce_halo2_clock[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - rt0_;
    }
#endif
#undef G2_STEP
#undef G2_HANDOFF
#undef G2_ROWS_AND_TAIL
#undef G2_RP
#undef G2_ROW
#undef G2_ROW_KH
#undef G2_READ
#undef G2_SET_E
#undef G2_PREFETCH_A
#undef G2_LOAD_A
}

template <typename E, bool POOL, bool GMAX, bool LIST = false, bool STRIP = false>
static int launch_halo2(Halo2Args a, hipStream_t stream) {
    a.ctiles = (a.Cout + 255) / 256;
    a.ntiles = a.ptiles * a.ctiles;
    if constexpr (!POOL && !GMAX && !LIST) {
        // few pixel tiles (the detector's small maps): 128-cout tiles on twice as many workgroups
        static const bool narrow_ok = !(getenv("CVPCE_HALO_NARROW") && getenv("CVPCE_HALO_NARROW")[0] == '0');   // dev A/B switch
        if (narrow_ok && !a.mask && a.Cout % 128 == 0 && 2 * a.ntiles <= g_cvpce_persistent_wgs) {
            a.ctiles = a.Cout / 128;
            a.ntiles = a.ptiles * a.ctiles;
            if (!cvpce_smem_attr_done<conv3x3_halo2_kernel<E, false, false, false, false, 4>>((const void*)conv3x3_halo2_kernel<E, false, false, false, false, 4>, 3 * G2_A_BYTES))
                return CVPCE_ERR_LAUNCH;
            hipLaunchKernelGGL((conv3x3_halo2_kernel<E, false, false, false, false, 4>), dim3(a.ntiles), dim3(256), 3 * G2_A_BYTES, stream, a);
            return cvpce_check_launch();
        }
    }
    const int smem = a.mask ? G2_SMEM_MASKED : (LIST ? G2_SMEM_LIST : 3 * G2_A_BYTES);
    if (!cvpce_smem_attr_done<conv3x3_halo2_kernel<E, POOL, GMAX, LIST, STRIP>>((const void*)conv3x3_halo2_kernel<E, POOL, GMAX, LIST, STRIP>, G2_SMEM_MASKED))
        return CVPCE_ERR_LAUNCH;
    int grid = a.ntiles < g_cvpce_persistent_wgs ? a.ntiles : g_cvpce_persistent_wgs;
    if (LIST) grid = grid < a.ctiles ? a.ctiles : grid / a.ctiles * a.ctiles;     // (work-list launches: the same number of workgroups per cout tile)
    if (a.mask && (a.ntiles + grid - 1) / grid > G2_MAX_SEQ) return CVPCE_ERR_ARG;   // the workgroup's tile list lives in LDS
    hipLaunchKernelGGL((conv3x3_halo2_kernel<E, POOL, GMAX, LIST, STRIP>), dim3(grid), dim3(512), smem, stream, a);
    return cvpce_check_launch();
}

template <typename E>
static int halo2_dispatch(const void* in, const void* wgt, const float* bias, const unsigned char* mask, const int* tile_map,
                          int n_map, void* out, float* gmax,
                          int gmax_stride, int gmax_off, int N, int H, int W, int Cin, int Cout, int K_pad, int Cout_pad, int relu,
                          int fuse_pool2, void* stream, int paired = 0) {
    if (N <= 0) return CVPCE_OK;
    if (!in || !wgt || (!out && !gmax)) return CVPCE_ERR_ARG;
    if (gmax && (!relu || mask || Cout <= 128 || gmax_off < 0 || gmax_off + Cout > gmax_stride)) return CVPCE_ERR_ARG;
    if (tile_map && (!mask || n_map <= 0)) return CVPCE_ERR_ARG;
    if (H <= 0 || W <= 0 || Cin % 64 != 0 || Cin <= 0 || Cout % 8 != 0 || Cout <= 0) return CVPCE_ERR_ARG;
    if (fuse_pool2 && ((H & 1) || (W & 1))) return CVPCE_ERR_ARG;
    if (K_pad != 9 * Cin || Cout_pad % 256 != 0 || Cout_pad < Cout) return CVPCE_ERR_ARG;
    if ((long long)N * H * W * Cin * 2 >= (1LL << 32) || (long long)N * H * W * Cout >= (1LL << 31)) return CVPCE_ERR_ARG;
    if ((long long)Cout_pad * K_pad * 2 >= (1LL << 31)) return CVPCE_ERR_ARG;
    if (mask && fuse_pool2) return CVPCE_ERR_ARG;
    // few output channels: the wide-tile kernel keeps every wave's 32-cout x 256-pixel tile (conv3x3_halo3.hip)
    // (CVPCE_DBG & 256, dev A/B only: also route every map that 32-pixel-wide tiles cover exactly to the wide-tile kernel --
    //  measured and not adopted, profiles/r02_ablation_halo2.md)
    if ((Cout <= 128 || ((CVPCE_DBG & 256) && !gmax && W % 32 == 0 && H % 16 == 0)) && !mask)
        return (E::kF16 ? cvpce_conv3x3_halo_wide_f16 : cvpce_conv3x3_halo_wide)(in, wgt, bias, out, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, fuse_pool2, stream);
    Halo2Args a;
    a.in = (const bf16_t*)in; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.mask = mask; a.out = (bf16_t*)out;
    a.gmax = gmax; a.gmax_stride = gmax_stride; a.gmax_off = gmax_off;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.K_pad = K_pad; a.relu = relu; a.cgroups = Cout_pad / 32;
    a.tiles_x = (W + G2_T - 1) / G2_T; a.tiles_y = (H + G2_T - 1) / G2_T;
    a.tile_map = tile_map; a.tiles_per_image = tile_map ? n_map : a.tiles_x * a.tiles_y;
    a.ptiles = N * a.tiles_per_image;
    a.in_bytes = (unsigned)((long long)N * H * W * Cin * 2);
    a.wgt_bytes = (unsigned)((long long)Cout_pad * K_pad * 2);
    a.ctiles = a.ntiles = 0;
    a.list = nullptr; a.list_count = nullptr;
    a.in_group_bytes = 0; a.out_group_elems = 0;
    if (paired) {
        // the two (or more) towers as cout tiles of one launch: Cout = 256 x towers, the output tower-major; the input tower-major too
        // (paired == 2) or one tensor every tower reads (paired == 1: the towers' first layer on the level atlas)
        if (!mask || gmax || fuse_pool2 || Cout % 256 != 0 || Cout < 512) return CVPCE_ERR_ARG;
        const long long towers = Cout / 256;
        if ((long long)N * H * W * Cin * 2 * (paired == 2 ? towers : 1) >= (1LL << 32)) return CVPCE_ERR_ARG;
        a.out_group_elems = (long long)N * H * W * 256;
        if (paired == 2) {
            a.in_group_bytes = (unsigned)((long long)N * H * W * Cin * 2);
            a.in_bytes = (unsigned)((long long)N * H * W * Cin * 2 * towers);
        }
    }
    if ((long long)a.ptiles * ((Cout + 255) / 256) * (Cin / 64) >= (1LL << 30)) return CVPCE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (gmax) return fuse_pool2 ? launch_halo2<E, true, true>(a, s) : launch_halo2<E, false, true>(a, s);
    if (mask) {
        // a masked launch keeps every workgroup's tile list in LDS (G2_MAX_SEQ entries): a batch too large for that is split
        // into launches over whole images (the tile sequence restarts with every image, so the results do not change)
        const long long per_image = (long long)a.tiles_per_image * ((Cout + 255) / 256);
        const long long cap = (long long)G2_MAX_SEQ * g_cvpce_persistent_wgs;
        if (per_image > cap) return CVPCE_ERR_ARG;
        const int n_per = (int)(cap / per_image);
        for (int n0 = 0; n0 < N; n0 += n_per) {
            Halo2Args b = a;
            b.N = (N - n0) < n_per ? (N - n0) : n_per;
            b.in = a.in + (size_t)n0 * H * W * Cin;
            b.out = a.out + (size_t)n0 * H * W * Cout;
            b.ptiles = b.N * a.tiles_per_image;
            b.in_bytes = (unsigned)((long long)b.N * H * W * Cin * 2);
            if (a.out_group_elems) {                                 // tower-major tensors: the towers' parts keep the whole batch's strides
                b.out = a.out + (size_t)n0 * H * W * 256;
                if (a.in_group_bytes) b.in_bytes = a.in_bytes - (unsigned)((long long)n0 * H * W * Cin * 2);
            }
            const int rc = launch_halo2<E, false, false>(b, s);
            if (rc != CVPCE_OK) return rc;
        }
        return CVPCE_OK;
    }
    return fuse_pool2 ? launch_halo2<E, true, false>(a, s) : launch_halo2<E, false, false>(a, s);
}

extern "C" int cvpce_conv3x3_halo(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W,
                                  int Cin, int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, void* stream) {
    return halo2_dispatch<ElemBF16>(in, wgt, bias, nullptr, nullptr, 0, out, nullptr, 0, 0, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, fuse_pool2, stream);
}

extern "C" int cvpce_conv3x3_halo_mac(const void* in, const void* wgt, const float* bias, void* out, float* mac, int mac_stride,
                                      int mac_off, int N, int H, int W, int Cin, int Cout, int K_pad, int Cout_pad,
                                      int fuse_pool2, void* stream) {
    if (!mac) return CVPCE_ERR_ARG;
    return halo2_dispatch<ElemBF16>(in, wgt, bias, nullptr, nullptr, 0, out, mac, mac_stride, mac_off, N, H, W, Cin, Cout, K_pad, Cout_pad, 1, fuse_pool2, stream);
}

extern "C" int cvpce_conv3x3_halo_masked(const void* in, const void* wgt, const float* bias, const unsigned char* mask,
                                         const int* tile_map, int n_tiles, void* out, int N, int H, int W, int Cin, int Cout,
                                         int K_pad, int Cout_pad, int relu, void* stream) {
    if (!mask) return CVPCE_ERR_ARG;
    return halo2_dispatch<ElemBF16>(in, wgt, bias, mask, tile_map, n_tiles, out, nullptr, 0, 0, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, 0, stream);
}

extern "C" int cvpce_conv3x3_halo_masked_paired(const void* in, const void* wgt, const float* bias, const unsigned char* mask,
                                                const int* tile_map, int n_tiles, void* out, int N, int H, int W, int Cin, int Cout,
                                                int K_pad, int Cout_pad, int relu, int in_paired, void* stream) {
    if (!mask) return CVPCE_ERR_ARG;
    return halo2_dispatch<ElemBF16>(in, wgt, bias, mask, tile_map, n_tiles, out, nullptr, 0, 0, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, 0, stream, in_paired ? 2 : 1);
}
extern "C" int cvpce_conv3x3_halo_masked_paired_f16(const void* in, const void* wgt, const float* bias, const unsigned char* mask,
                                                    const int* tile_map, int n_tiles, void* out, int N, int H, int W, int Cin, int Cout,
                                                    int K_pad, int Cout_pad, int relu, int in_paired, void* stream) {
    if (!mask) return CVPCE_ERR_ARG;
    return halo2_dispatch<ElemF16>(in, wgt, bias, mask, tile_map, n_tiles, out, nullptr, 0, 0, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, 0, stream, in_paired ? 2 : 1);
}

// fp16 twins (the detector's accuracy mode; same contracts, element type fp16)
extern "C" int cvpce_conv3x3_halo_f16(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W,
                                      int Cin, int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, void* stream) {
    return halo2_dispatch<ElemF16>(in, wgt, bias, nullptr, nullptr, 0, out, nullptr, 0, 0, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, fuse_pool2, stream);
}

extern "C" int cvpce_conv3x3_halo_masked_f16(const void* in, const void* wgt, const float* bias, const unsigned char* mask,
                                             const int* tile_map, int n_tiles, void* out, int N, int H, int W, int Cin, int Cout,
                                             int K_pad, int Cout_pad, int relu, void* stream) {
    if (!mask) return CVPCE_ERR_ARG;
    return halo2_dispatch<ElemF16>(in, wgt, bias, mask, tile_map, n_tiles, out, nullptr, 0, 0, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, 0, stream);
}

// Work-list launch (bf16 only: the embedder).  Cout <= 128 is forwarded to the wide-tile kernel's list entry.
int cvpce_conv3x3_halo_wide_list(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W, int Cin,
                                 int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, const unsigned long long* list,
                                 const int* count_dev, void* stream);

static int halo_list_launch(const void* in, const void* wgt, const float* bias, void* out, float* mac, int mac_stride,
                            int mac_off, int N, int H, int W, int Cin, int Cout, int K_pad, int Cout_pad, int relu,
                            int fuse_pool2, const unsigned long long* list, const int* count_dev, bool strips, void* stream) {
    if (N <= 0) return CVPCE_OK;
    if (!in || !wgt || (!out && !mac) || !list || !count_dev) return CVPCE_ERR_ARG;
    if (mac && (!relu || Cout <= 128 || mac_off < 0 || mac_off + Cout > mac_stride)) return CVPCE_ERR_ARG;
    if (H <= 0 || W <= 0 || Cin % 64 != 0 || Cin <= 0 || Cout % 8 != 0 || Cout <= 0) return CVPCE_ERR_ARG;
    if (fuse_pool2 && ((H & 1) || (W & 1))) return CVPCE_ERR_ARG;
    if (K_pad != 9 * Cin || Cout_pad % 256 != 0 || Cout_pad < Cout) return CVPCE_ERR_ARG;
    if ((long long)N * H * W * Cin * 2 >= (1LL << 32) || (long long)N * H * W * Cout >= (1LL << 31)) return CVPCE_ERR_ARG;
    if ((long long)Cout_pad * K_pad * 2 >= (1LL << 31) || N > 65535) return CVPCE_ERR_ARG;
    if (Cout <= 128) {
        if (strips) return CVPCE_ERR_ARG;               // (the wide-tile kernel has no strip mode)
        return cvpce_conv3x3_halo_wide_list(in, wgt, bias, out, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, fuse_pool2, list, count_dev, stream);
    }
    Halo2Args a;
    a.in = (const bf16_t*)in; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.mask = nullptr; a.out = (bf16_t*)out;
    a.gmax = mac; a.gmax_stride = mac_stride; a.gmax_off = mac_off;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.K_pad = K_pad; a.relu = relu; a.cgroups = Cout_pad / 32;
    a.tiles_x = (W + G2_T - 1) / G2_T; a.tiles_y = (H + G2_T - 1) / G2_T;
    if (a.tiles_x > 255 || a.tiles_y > 255 || H > 65535 || W > 65535) return CVPCE_ERR_ARG;
    a.tile_map = nullptr; a.tiles_per_image = a.tiles_x * a.tiles_y;
    a.ptiles = N * a.tiles_per_image;
    a.in_bytes = (unsigned)((long long)N * H * W * Cin * 2);
    a.wgt_bytes = (unsigned)((long long)Cout_pad * K_pad * 2);
    a.ctiles = a.ntiles = 0;
    a.list = list; a.list_count = count_dev;
    a.in_group_bytes = 0; a.out_group_elems = 0;
    if ((long long)a.ptiles * ((Cout + 255) / 256) * (Cin / 64) >= (1LL << 30)) return CVPCE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (strips) {
        if (mac) return fuse_pool2 ? launch_halo2<ElemBF16, true, true, true, true>(a, s) : launch_halo2<ElemBF16, false, true, true, true>(a, s);
        return fuse_pool2 ? launch_halo2<ElemBF16, true, false, true, true>(a, s) : launch_halo2<ElemBF16, false, false, true, true>(a, s);
    }
    if (mac) return fuse_pool2 ? launch_halo2<ElemBF16, true, true, true>(a, s) : launch_halo2<ElemBF16, false, true, true>(a, s);
    return fuse_pool2 ? launch_halo2<ElemBF16, true, false, true>(a, s) : launch_halo2<ElemBF16, false, false, true>(a, s);
}

extern "C" int cvpce_conv3x3_halo_list(const void* in, const void* wgt, const float* bias, void* out, float* mac, int mac_stride,
                                       int mac_off, int N, int H, int W, int Cin, int Cout, int K_pad, int Cout_pad, int relu,
                                       int fuse_pool2, const unsigned long long* list, const int* count_dev, void* stream) {
    return halo_list_launch(in, wgt, bias, out, mac, mac_stride, mac_off, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, fuse_pool2, list, count_dev, false, stream);
}

extern "C" int cvpce_conv3x3_halo_strips(const void* in, const void* wgt, const float* bias, void* out, float* mac, int mac_stride,
                                         int mac_off, int N, int H, int W, int Cin, int Cout, int K_pad, int Cout_pad, int relu,
                                         int fuse_pool2, const unsigned long long* strip_list, const int* count_dev, void* stream) {
    return halo_list_launch(in, wgt, bias, out, mac, mac_stride, mac_off, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, fuse_pool2, strip_list, count_dev, true, stream);
}

// [host] row-major chunk-major [Cout_pad][9 Cin] -> the fragment-major halo weight layout (include/cvpce_amd.h)
extern "C" int cvpce_pack_halo_weights(const void* src_rowmajor, void* dst_halo, int Cout_pad, int Cin) {
    if (!src_rowmajor || !dst_halo || Cout_pad <= 0 || Cout_pad % 32 != 0 || Cin <= 0 || Cin % 64 != 0) return CVPCE_ERR_ARG;
    const unsigned short* src = (const unsigned short*)src_rowmajor;
    unsigned short* dst = (unsigned short*)dst_halo;
    const int nch = Cin / 64, groups = Cout_pad / 32;
    const size_t K_pad = (size_t)9 * Cin;
    size_t o = 0;
    for (int c = 0; c < nch; ++c)
        for (int g = 0; g < groups; ++g)
            for (int kw = 0; kw < 3; ++kw)
                for (int half = 0; half < 2; ++half)
                    for (int kh = 0; kh < 3; ++kh)
                        for (int mt = 0; mt < 2; ++mt)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int m = lane & 15, q = lane >> 4;
                                const size_t row = (size_t)32 * g + 8 * (m >> 2) + 4 * mt + (m & 3);
                                const size_t k = (size_t)c * 576 + (kh * 3 + kw) * 64 + half * 32 + q * 8;
                                for (int e = 0; e < 8; ++e) dst[o++] = src[row * K_pad + k + e];
                            }
    return CVPCE_OK;
}
