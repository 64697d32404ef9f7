// 3x3 / stride 1 / pad 1 convolution, Cin % 64 == 0, with the INPUT HALO PATCH resident in LDS (VGG16 conv2_2 ... conv5_3).
//
// In the implicit-GEMM kernels every K-step re-stages a [pixels][32] im2col slice next to the [couts][32] weight
// slice, although the nine taps of one 64-channel chunk read the same 18x18 pixels: LDS-DMA issue is what bounds those
// kernels (profiles/r01_ablation_conv4_2.md).  Here a persistent 8-wave workgroup owns a 16x16-pixel x TC-cout output
// tile at a time and
//   * fetches the 18x18x64 halo patch of a channel chunk ONCE (41 DMA pieces, double buffered: the next chunk's patch --
//     possibly the next tile's -- lands during the 18 K-steps of the current one),
//   * streams only the weights through a 4-slot ring of [TC][32] K-slices (TC/16 pieces per K-step),
//   * reads the pixel fragments of tap (kh,kw) straight out of the patch at a shifted row index.
// DMA pieces per 8.4 MFLOP (TC = 256): 32 + 41/18 = 34.3 instead of 64.  Products run on v_mfma_f32_16x16x32_bf16,
// fragment groups software-pipelined across the ring hand-off exactly like conv_dma16_kernel.
// K order / weight layout: the chunk-major [Cout_pad][K_pad] of include/cvpce_amd.h, so K-step st of a tile is the
// contiguous K range [32 st, 32 st + 32).  Optional fused bias / ReLU / MaxPool2d(2,2).
#include "common.h"
#include "../../include/cvpce_amd.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

#define H3_T 16
#define H3_P 18
#define H3_NPIX 324
#define H3_ROWS 328
#define H3_A_BYTES (H3_ROWS * 128)
#define H3_NPIECE 41
#define H3_NS 4

// Patch swizzle: 16-byte chunk c of patch pixel (py, px) lives at physical chunk c ^ h3_swz(py, px).  ds_read_b128 is
// served in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS), i.e. a group mixes
// two K-quarters (lq, lq^1) over complementary halves of the 16 pixel lanes.  With the pixel lanes laid out as below
// (non-pooled: lanes {0-3,12-15} = even columns, {4-11} = odd columns; pooled: one 2x2 window per lane quad) this
// function makes every tap's fragment read conflict-free at any alignment: bits 2:1 separate 4 consecutive column
// pairs, bit 0 separates the two rows of a pooling window / the other 4 column pairs of a row.
__device__ __forceinline__ int h3_swz0(int u) { return ((u & 3) << 1) | ((u >> 2) & 1); }
__device__ __forceinline__ int h3_swz(int py, int px) { return h3_swz0(px >> 1) ^ (py & 1); }
// non-pooled lane -> column of the 16-pixel row
__device__ __forceinline__ int h3_col(int l16) { return l16 < 4 ? 2 * l16 : (l16 >= 12 ? 2 * (l16 - 8) : 2 * (l16 - 4) + 1); }

struct HaloArgs {
    const bf16_t* in;    // [N][H][W][Cin]
    const bf16_t* wgt;   // [Cout_pad][K_pad], chunk-major K
    const float* bias;
    bf16_t* out;         // [N][H][W][Cout] or pooled [N][H/2][W/2][Cout]
    int N, H, W, Cin, Cout, K_pad, relu, pool;
    int tiles_x, tiles_y, ptiles, ctiles, ntiles;
    unsigned in_bytes, wgt_bytes;
};

// PAIR: the ring hand-off (vmcnt wait + barrier + DMA issue) happens every SECOND K-step and moves two K-slices at a
// time (6 ring slots): with TC = 128 a K-step is only 16 MFMAs per wave, too little to amortise a barrier.
template <int TC, bool PAIR>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_kernel(HaloArgs a) {
    constexpr int BK = 32, NS = PAIR ? 6 : H3_NS;
    constexpr int WC = 2, WP = 4;
    constexpr int MT = TC / WC / 16;          // 16-cout blocks per wave (8 or 4)
    constexpr int MH = MT / 2;
    constexpr int NT = 4;                     // 16-pixel blocks per wave (4 output rows x 16 columns)
    constexpr int WJ = TC / (16 * 8);         // weight DMA pieces per wave per K-step (2 or 1)
    constexpr int SLOT = TC * BK * 2;         // bytes per ring slot

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Wr = smem;                              // [NS][TC][32] bf16
    unsigned char* Ap = smem + NS * SLOT;                  // [2][328][64] bf16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wid / WP, wp = wid % WP;

    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_p = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);

    const int nchunks = a.Cin >> 6;
    const int nst = nchunks * 18;             // K-steps per tile
    // this workgroup's tile sequence: tile = blockIdx.x + i * gridDim.x ; tile -> (cout tile fastest, pixel tile)
    const int my_tiles = (a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;
    const int total_stages = my_tiles * nst;               // < 2^31: checked on the host
    const int total_chunks = my_tiles * nchunks;

    // ---- weight DMA: piece j of wave w fills ring rows j*128 + w*16 .. +15 (row = lane>>2, phys chunk = lane&3) ----
    const int wl_row = wid * 16 + (lane >> 2);
    const int wl_chunk = (lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3);      // T = {0,2,3,1}
    // ---- patch DMA: piece j fills patch rows 8j .. 8j+7 (row = lane>>3, phys chunk = lane&7) ----
    // pieces are dealt round-robin to the 8 waves: wave w issues pieces w, w+8, ... (6 for w = 0, else 5)
    const int npp = (wid == 0) ? 6 : 5;

    auto tile_of = [&](int seq, int& n, int& ty, int& tx, int& ct) {
        const int t = (int)blockIdx.x + seq * (int)gridDim.x;
        ct = t % a.ctiles;
        const int p = t / a.ctiles;
        n = p / (a.tiles_x * a.tiles_y);
        const int r = p - n * (a.tiles_x * a.tiles_y);
        ty = r / a.tiles_x;
        tx = r - ty * a.tiles_x;
    };
    // tile coordinates of the CURRENT and the NEXT tile of this workgroup (index 0 / 1), refreshed once per tile:
    // the per-K-step DMA issue below must not pay integer divisions
    int t_n[2], t_y[2], t_x[2], t_c[2];
    auto load_tiles = [&](int seq) {
        tile_of(seq, t_n[0], t_y[0], t_x[0], t_c[0]);
        if (seq + 1 < my_tiles) tile_of(seq + 1, t_n[1], t_y[1], t_x[1], t_c[1]);
        else { t_n[1] = t_n[0]; t_y[1] = t_y[0]; t_x[1] = t_x[0]; t_c[1] = t_c[0]; }
    };
    load_tiles(0);
    // per-lane constants of the patch DMA pieces (piece j = wid + 8 i): patch row -> (py, px), swizzled chunk
    int pc_off[6];            // byte offset relative to the tile origin pixel (-1,-1) at chunk 0, or -1 if the row is padding
    int pc_py[6], pc_px[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int j = wid + 8 * i;
        const int pp = j * 8 + (lane >> 3);
        pc_py[i] = pp / H3_P;
        pc_px[i] = pp - pc_py[i] * H3_P;
        const int lchunk = (lane & 7) ^ h3_swz(pc_py[i], pc_px[i]);
        pc_off[i] = (j < H3_NPIECE && pp < H3_NPIX) ? lchunk * 16 : -1;
    }
    auto issue_weights = [&](int which, int st, int slot) {   // K-step st of the current (0) / next (1) tile -> ring slot
        const int ct = t_c[which];
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const unsigned off = (unsigned)(((size_t)(ct * TC + j * 128 + wl_row) * a.K_pad + wl_chunk * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void*)(Wr + slot * SLOT + (j * 128 + wid * 16) * (BK * 2)), 16,
                                                     (int)off, st * (BK * 2), 0, 0);
        }
    };
    auto issue_patch = [&](int which, int c, int buf) {       // channel chunk c of the current / next tile -> patch buffer
        const int n = t_n[which], y0 = t_y[which] * H3_T - 1, x0 = t_x[which] * H3_T - 1;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i < npp) {
                const int y = y0 + pc_py[i], x = x0 + pc_px[i];
                const bool ok = pc_off[i] >= 0 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                const unsigned off = (unsigned)((((size_t)(n * a.H + y) * a.W + x) * a.Cin + c * 64) * 2) + (unsigned)pc_off[i];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_p, (lds_void*)(Ap + buf * H3_A_BYTES + (wid + 8 * i) * 1024), 16,
                                                         (int)(ok ? off : 0xFFFFFFF0u), 0, 0, 0);
            }
        }
    };

    // ---- fragment addressing ----
    const unsigned lds_w = (unsigned)(size_t)(lds_char*)Wr, lds_a = (unsigned)(size_t)(lds_char*)Ap;
    const unsigned lrow16 = lane & 15, lq = lane >> 4;
    const unsigned wlane = lds_w + wc * (TC / WC) * (BK * 2) + lrow16 * (BK * 2) + ((lq ^ ((0x78 >> (2 * (lrow16 >> 2))) & 3)) << 4);
    // pixel of this lane in each of the 4 blocks of the wave (patch coordinates of the OUTPUT pixel, tap adds (kh, kw)).
    // fragment address of tap (kh,kw), K-half hf:  ((cbase[nt] ^ s) + off)  with
    //   s   = sw3[kw] ^ (hf << 6) ^ ((kh & 1) << 4)        (per lane, once per K-step)
    //   off = patch buffer + (kh*18 + kw) * 128             (wave-uniform)
    unsigned cbase[NT], sw3[3];
    {
        int col0 = 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            int row, col, par;
            if (a.pool) {   // block = 2 rows x 8 columns in 2x2-quad order
                const int q = lrow16 >> 2, sub = lrow16 & 3;
                row = 4 * wp + 2 * (nt >> 1) + (sub >> 1);
                col = 8 * (nt & 1) + 2 * q + (sub & 1);
                par = (row & 1) ^ (nt & 1);          // +8 columns flips bit 0 of h3_swz0
                if (nt == 0) col0 = col;
            } else {        // block = one output row of 16 pixels
                row = 4 * wp + nt;
                col = h3_col(lrow16);
                par = row & 1;
                col0 = col;
            }
            cbase[nt] = (unsigned)((row * H3_P + col) * 128 + (par << 4));
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) sw3[kw] = (unsigned)(((int)lq ^ h3_swz0(((col0 + kw) >> 1) & 7)) << 4);
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    bf16x8 af[MT], bfr[2][NT];
#define H3_READ_A(SLOTI, M0)                                                                                   \
    {                                                                                                          \
        const unsigned wb_ = wlane + (unsigned)(SLOTI) * SLOT;   /* SLOTI < NS */                                                 \
        _Pragma("unroll") for (int i = 0; i < MH; ++i)                                                         \
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[(M0) + i]) : "v"(wb_ + ((M0) + i) * 16 * (BK * 2)));  \
    }
    // pixel fragments of flat stage G: chunk parity selects the patch buffer, tap shifts the row, half selects 64 B
#define H3_READ_B(R18, BUF, SET)                                                                               \
    {                                                                                                          \
        const int tap = (R18) >> 1, hf = (R18) & 1;                                                            \
        const int kh = (tap * 11) >> 5, kw = tap - kh * 3;      /* tap / 3 for tap < 9 */                      \
        const unsigned off_ = lds_a + (unsigned)(BUF) * H3_A_BYTES + (unsigned)(kh * H3_P + kw) * 128;         \
        const unsigned s_ = (kw == 0 ? sw3[0] : (kw == 1 ? sw3[1] : sw3[2])) ^ (unsigned)((hf << 6) | ((kh & 1) << 4)); \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                      \
            asm volatile("ds_read_b128 %0, %1" : "=v"(bfr[SET][nt]) : "v"((cbase[nt] ^ s_) + off_));           \
    }
#define H3_MFMAS(M0, SET)                                                                                      \
    {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MH; ++i)                                                         \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                  \
                acc[(M0) + i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[(M0) + i], bfr[SET][nt], acc[(M0) + i][nt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                         \
    }

    // ---- prologue: patch of chunk 0, weights of stages 0..2 ----
    issue_patch(0, 0, 0);
    issue_weights(0, 0, 0);                                  // nst >= 18: the first K-steps all belong to tile 0
    issue_weights(0, 1, 1);
    issue_weights(0, 2, 2);
    if (PAIR) issue_weights(0, 3, 3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    H3_READ_A(0, 0)
    H3_READ_B(0, 0, 0)
    // running position (all wave-uniform ints): g = flat K-step, r18 = K-step within the channel chunk, cc = flat chunk
    int g = 0, r18 = 0, cc = 0, cchunk = 0;      // cchunk = channel chunk within the current tile
    int sl = 0;                                   // ring slot of K-step g (g % NS, kept incrementally)
    auto slot_of = [&](int d) { int x = sl + d; return x >= NS ? x - NS : x; };     // slot of K-step g + d, d < NS

    // one K-step; SET = parity of g (pixel-fragment register set), kept a compile-time constant by the 2x unroll below
#define H3_STAGE(SEQ, ST, SET)                                                                                 \
    {                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* group 0 of this stage has landed */         \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        H3_READ_A(sl, MH)                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        H3_MFMAS(0, SET)                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (g + 1 < total_stages) {                                                                            \
            if (!PAIR) {                                                                                       \
                /* hand-off: weights of stage g+1 (and a patch issued >= 1 stage ago) have landed; youngest allowed */ \
                /* outstanding: weights of stage g+2 (WJ pieces) and, right after a patch issue, that patch       */ \
                const bool patch_young = r18 == 1 && cc + 1 < total_chunks;                                    \
                if (g + 2 < total_stages) {                                                                    \
                    if (patch_young) { if (wid == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WJ + 6) : "memory"); \
                                       else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WJ + 5) : "memory"); }    \
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WJ) : "memory");                             \
                } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
                __builtin_amdgcn_s_barrier();                                                                  \
                /* first K-step of a chunk: fetch the NEXT chunk's patch into the other buffer, BEFORE the weights */ \
                if (r18 == 0 && cc + 1 < total_chunks) {                                                       \
                    const int nc_ = cchunk + 1;                                                                \
                    if (nc_ < nchunks) issue_patch(0, nc_, (cc + 1) & 1); else issue_patch(1, 0, (cc + 1) & 1); \
                }                                                                                              \
                if (g + 3 < total_stages) {                                                                    \
                    if ((ST) + 3 < nst) issue_weights(0, (ST) + 3, slot_of(3));                                \
                    else issue_weights(1, (ST) + 3 - nst, slot_of(3));                                         \
                }                                                                                              \
            } else if ((SET) == 1) {                                                                           \
                /* pair hand-off (odd K-step): K-steps g+1, g+2 were issued two steps ago and must have landed; */ \
                /* the only younger DMA can be a patch issued right after them (at r18 == 1)                     */ \
                const bool patch_young = r18 == 3 && cc + 1 < total_chunks;                                    \
                if (patch_young) { if (wid == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");              \
                                   else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }                     \
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
                __builtin_amdgcn_s_barrier();                                                                  \
                _Pragma("unroll") for (int d = 3; d <= 4; ++d)                                                 \
                    if (g + d < total_stages) {                                                                \
                        if ((ST) + d < nst) issue_weights(0, (ST) + d, slot_of(d));                            \
                        else issue_weights(1, (ST) + d - nst, slot_of(d));                                     \
                    }                                                                                          \
                /* second K-step of a chunk: fetch the NEXT chunk's patch, AFTER the weights (it may stay in flight) */ \
                if (r18 == 1 && cc + 1 < total_chunks) {                                                       \
                    const int nc_ = cchunk + 1;                                                                \
                    if (nc_ < nchunks) issue_patch(0, nc_, (cc + 1) & 1); else issue_patch(1, 0, (cc + 1) & 1); \
                }                                                                                              \
            }                                                                                                  \
            const int nr18 = (r18 == 17) ? 0 : r18 + 1;                                                        \
            const int nbuf = (r18 == 17) ? ((cc + 1) & 1) : (cc & 1);                                          \
            H3_READ_A(slot_of(1), 0)                                                                           \
            H3_READ_B(nr18, nbuf, (SET) ^ 1)                                                                   \
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MH + NT) : "memory");                                   \
        } else {                                                                                               \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        H3_MFMAS(MH, SET)                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        ++g;                                                                                                   \
        if (++sl == NS) sl = 0;                                                                                \
        if (++r18 == 18) { r18 = 0; ++cc; ++cchunk; }                                                          \
    }

    const int lp = lane & 15;
    for (int seq = 0; seq < my_tiles; ++seq) {
        if (seq) load_tiles(seq);
        cchunk = 0;
        for (int st = 0; st < nst; st += 2) {                // nst is even: pixel-fragment set = st & 1
            H3_STAGE(seq, st, 0)
            H3_STAGE(seq, st + 1, 1)
        }
        // ---- epilogue of this tile (the next tile's patch / weights are already in flight) ----
        const int n = t_n[0], ty = t_y[0], tx = t_x[0], ct = t_c[0];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            size_t opix;
            bool store_lane;                     // ragged right / bottom tiles: pixels outside the image are dropped
            if (a.pool) {
                const int q = lp >> 2;
                const int oy = (ty * H3_T) / 2 + 2 * wp + (nt >> 1), ox = (tx * H3_T) / 2 + 4 * (nt & 1) + q;
                opix = (size_t)(n * (a.H >> 1) + oy) * (a.W >> 1) + ox;
                store_lane = (lp & 3) == 0 && oy < (a.H >> 1) && ox < (a.W >> 1);
            } else {
                const int oy = ty * H3_T + 4 * wp + nt, ox = tx * H3_T + h3_col(lp);
                opix = (size_t)(n * a.H + oy) * a.W + ox;
                store_lane = oy < a.H && ox < a.W;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int co = ct * TC + wc * (TC / WC) + mt * 16 + 4 * (int)lq;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][j];
                if (a.bias && co < a.Cout) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += b[j];
                }
                if (a.pool) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = quad_max(v[j]);
                }
                if (a.relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                if (store_lane && co < a.Cout) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f32_to_bf16(v[j]);
                    *reinterpret_cast<bf16x4*>(a.out + opix * a.Cout + co) = o;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mt][nt][e] = 0.f;
            }
        }
    }
#undef H3_STAGE
#undef H3_READ_A
#undef H3_READ_B
#undef H3_MFMAS
}

template <int TC, bool PAIR>
static int launch_halo(HaloArgs a, hipStream_t stream) {
    a.ctiles = (a.Cout + TC - 1) / TC;
    a.ntiles = a.ptiles * a.ctiles;
    const int smem = (PAIR ? 6 : H3_NS) * TC * 64 + 2 * H3_A_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_halo_kernel<TC, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return CVPCE_ERR_LAUNCH;
        attr_set = true;
    }
    const int grid = a.ntiles < 256 ? a.ntiles : 256;
    hipLaunchKernelGGL((conv3x3_halo_kernel<TC, PAIR>), dim3(grid), dim3(512), smem, stream, a);
    return cvpce_check_launch();
}

extern "C" int cvpce_conv3x3_halo_ring(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W,
                                  int Cin, int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, void* stream) {
    if (N <= 0) return CVPCE_OK;
    if (!in || !wgt || !out) return CVPCE_ERR_ARG;
    if (H <= 0 || W <= 0 || Cin % 64 != 0 || Cin <= 0 || Cout % 4 != 0 || Cout <= 0) return CVPCE_ERR_ARG;
    if (fuse_pool2 && ((H & 1) || (W & 1))) return CVPCE_ERR_ARG;
    if (K_pad != 9 * Cin || Cout_pad % 256 != 0 || Cout_pad < Cout) return CVPCE_ERR_ARG;
    if ((long long)N * H * W * Cin * 2 >= (1LL << 32) || (long long)N * H * W * Cout >= (1LL << 31)) return CVPCE_ERR_ARG;
    HaloArgs a;
    a.in = (const bf16_t*)in; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.out = (bf16_t*)out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.K_pad = K_pad; a.relu = relu; a.pool = fuse_pool2 ? 1 : 0;
    a.tiles_x = (W + H3_T - 1) / H3_T; a.tiles_y = (H + H3_T - 1) / H3_T; a.ptiles = N * a.tiles_x * a.tiles_y;
    a.in_bytes = (unsigned)((long long)N * H * W * Cin * 2);
    a.wgt_bytes = (unsigned)((long long)Cout_pad * K_pad * 2);
    a.ctiles = a.ntiles = 0;
    if (Cout > 128) return launch_halo<256, false>(a, (hipStream_t)stream);
    return launch_halo<128, true>(a, (hipStream_t)stream);
}
