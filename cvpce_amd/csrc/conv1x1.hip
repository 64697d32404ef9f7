// 1x1 convolution (any stride) = a plain GEMM out[M][Cout] = in[M'][Cin] . W[Cout][Cin]^T with fused bias / residual /
// nearest-upsampled residual / ReLU -- the 40 pointwise convolutions of the ResNet-50 + FPN detector (and MACResNet).
//
// These GEMMs are small and mostly HBM-bound (K = 64 ... 2048, a 64-channel -> 256-channel layer moves 370 MB for 10 GFLOP),
// and the LDS-ring kernels spend their time in prologue / epilogue: 60-330 TFLOP/s, 8-byte scattered stores.  Here there
// is NO LDS and NO barrier: every wave owns a 64-pixel x 64-cout tile and loads both MFMA operands straight from global
// memory in fragment layout (lane (m, q) <- 16 B of row m: pixels rows are Cin*2 contiguous bytes in NHWC, weight rows
// K_pad*2), double-buffered in registers; the 4 waves of a workgroup take neighbouring tiles (cout tile fastest) so the
// shared operand hits L1.  Cout rows are permuted inside the wave tile (MFMA row 4q+j of block mt = cout 16q + 4mt + j)
// so that a lane ends with 16 consecutive couts: the 4 lanes of a pixel write one full 128-byte line.
#include "common.h"
#include <stdlib.h>
#include "../../include/cvpce_amd.h"

struct C1Args {
    const bf16_t* in;    // [N][H][W][Cin]
    const bf16_t* wgt;   // [Cout_pad][K_pad], k = ci
    const float* bias;   // [Cout] or null
    const bf16_t* res;   // [N][Hr][Wr][Cout] or null
    bf16_t* out;         // [N][Ho][Wo][Cout]
    int N, H, W, Cin, Cout, stride, Ho, Wo, K_pad, M, relu, res_mode, Hr, Wr;
    int ntile_n, ntiles;
    unsigned in_bytes, wgt_bytes;
};

// ---- epilogue: lane (pixel l16 of block nt, q = lq) holds couts ct*64 + 16q .. +15 ----
template <typename E>
__device__ __forceinline__ void c1_epilogue(const C1Args& a, const f32x4 (&acc)[4][4], int ct, int pt, int l16, int lq, int hw) {
    const int co = ct * 64 + 16 * lq;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int m = pt * 64 + nt * 16 + l16;
        if (m >= a.M) continue;
        float v[16];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * mt + j] = acc[mt][nt][j];
        if (a.res_mode) {
            size_t rp = (size_t)m;
            if (a.res_mode == 2) {
                const int n = m / hw, r = m - n * hw, oy = r / a.Wo, ox = r - oy * a.Wo;
                rp = (size_t)(n * a.Hr + (oy * a.Hr) / a.Ho) * a.Wr + (ox * a.Wr) / a.Wo;
            }
#ifdef C1_ABL_ST       /* timing-only ablation: residual pieces and stores as lane-contiguous runs (wrong addresses inside the tile's rows) */
            const size_t ab_ = ((size_t)(pt * 64 + nt * 16) * a.Cout + (size_t)ct * 64) + (size_t)(threadIdx.x & 63) * 16;
            const bf16x8 r0 = *reinterpret_cast<const bf16x8*>(a.res + (a.res_mode == 2 ? rp * a.Cout + co : ab_));
            const bf16x8 r1 = *reinterpret_cast<const bf16x8*>(a.res + (a.res_mode == 2 ? rp * a.Cout + co : ab_) + 8);
#else
            const bf16x8 r0 = *reinterpret_cast<const bf16x8*>(a.res + rp * a.Cout + co);
            const bf16x8 r1 = *reinterpret_cast<const bf16x8*>(a.res + rp * a.Cout + co + 8);
#endif
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] += E::widen(r0[j]); v[8 + j] += E::widen(r1[j]); }
        }
        if (a.relu) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = relu_bits(v[j]);
        }
        unsigned pk[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2 u = __builtin_bit_cast(uint2, E::pack4(f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]}));
            pk[2 * g] = u.x; pk[2 * g + 1] = u.y;
        }
#ifdef C1_ABL_ST
        bf16_t* dst = a.out + ((size_t)(pt * 64 + nt * 16) * a.Cout + (size_t)ct * 64) + (size_t)(threadIdx.x & 63) * 16;
#else
        bf16_t* dst = a.out + (size_t)m * a.Cout + co;
#endif
        *reinterpret_cast<u32x4*>(dst) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(dst + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
    }
}

template <typename E>
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(C1Args a) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroups are dealt to the 8 XCDs round-robin: with the remap the (up to 8) workgroups that share a pixel tile -- one per
    // four cout tiles, cout fastest -- run on ONE XCD and its L2 serves their re-reads; in blockIdx order a 2048-cout layer
    // fetched every input row eight times from beyond L2
    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x) * 4 + wid;
    if (tile >= a.ntiles) return;
    const int ct = tile % a.ntile_n, pt = tile / a.ntile_n;
    const int l16 = lane & 15, lq = lane >> 4;

    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_i = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);

    unsigned aoff[4], boff[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        aoff[mt] = (unsigned)(((ct * 64 + 16 * (l16 >> 2) + 4 * mt + (l16 & 3)) * a.K_pad + lq * 8) * 2);
    const int hw = a.Ho * a.Wo;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        int m = pt * 64 + nt * 16 + l16;
        m = m < a.M ? m : a.M - 1;                                    // ragged last tile: loads clamped, stores masked
        const int n = m / hw, r = m - n * hw, oy = r / a.Wo, ox = r - oy * a.Wo;
        boff[nt] = (unsigned)((((size_t)(n * a.H + oy * a.stride) * a.W + ox * a.stride) * a.Cin + lq * 8) * 2);
    }

    f32x4 acc[4][4];          // [16-cout block][16-pixel block]
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 b = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + ct * 64 + 16 * lq + 4 * mt) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = b;
    }

    bf16x8 A[2][4], B[2][4];
#define C1_LOAD(BUF, KS)                                                                                       \
    {                                                                                                          \
        const int so_ = (KS) * 64;                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                     \
            A[BUF][i_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_w, aoff[i_], so_, 0)); \
            B[BUF][i_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_i, boff[i_], so_, 0)); \
        }                                                                                                      \
    }
#define C1_MFMA(BUF)                                                                                           \
    _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_)                                                        \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_)                                                    \
            acc[mt_][nt_] = E::mfma16(A[BUF][mt_], B[BUF][nt_], acc[mt_][nt_]);

    const int ks = a.K_pad >> 5;          // K-steps of 32 (even: K_pad % 64 == 0)
    C1_LOAD(0, 0)
    for (int k = 0; k < ks; k += 2) {
        C1_LOAD(1, k + 1)
        C1_MFMA(0)
        if (k + 2 < ks) C1_LOAD(0, k + 2)
        C1_MFMA(1)
    }
#undef C1_LOAD
#undef C1_MFMA

    c1_epilogue<E>(a, acc, ct, pt, l16, lq, hw);
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride-1 form (round 4).  The kernel above loads its MFMA operands in fragment layout: lane (row l16, K-chunk lq) fetches 16
// bytes of ITS row, so the 64 lanes of one load touch 16 rows and no two neighbouring lanes share a 64-byte block -- the
// texture addresser then handles one lane per clock (measured: ~61 clocks per `buffer_load_dwordx4`, 16 B per CU per clock,
// whatever the ring depth or the order of the K-steps; tools/dev/bench_1x1.py with the C1_ABL ablations of that build), and the
// 64 operand loads of a wave tile were 3/5 of its time.  Here every wave streams its operands through a PRIVATE slice of LDS with
// `buffer_load_dwordx4 ... lds`: 8 neighbouring lanes fetch one whole 128-byte line of a row (a 64-deep K-stage: 64 cout rows +
// 64 pixel rows = 16 KiB, 16 loads), the fragments are read back with ds_read_b128 (chunks XOR-swizzled by row bits so that the 16
// lanes of a quarter wave hit 16 different 16-byte bank groups), two stages per wave.  No barrier anywhere -- a wave waits for
// its own loads with a counted vmcnt -- and the waves are persistent: the first stage of a wave's NEXT tile is in flight while
// it runs its epilogue.  Pixel rows are addressed as m * Cin (stride 1: output pixel m IS input pixel m); rows past M lie
// beyond the buffer's num_records and read as zeros.
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
#define C1S_STAGE 16384                 // bytes of one K-stage of one wave: A 64 x 128 B, then B 64 x 128 B
#define C1S_WAVE (2 * C1S_STAGE)

template <typename E>
__global__ __launch_bounds__(256, 1) void conv1x1_stream_kernel(C1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int gw = (int)blockIdx.x * 4 + wid, GW = (int)gridDim.x * 4;
    if (gw >= a.ntiles) return;
    unsigned char* wbase = smem + wid * C1S_WAVE;
    const int hw = a.Ho * a.Wo;
    const int nstage = a.K_pad >> 6;
    const unsigned rowbytes = (unsigned)a.K_pad * 2u;

    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_i = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);

    // DMA piece i of a stage (i = 0..7): LDS rows 8 i .. 8 i + 7 of the A tile / of the B tile; lane = 8 rr + p fills physical chunk p
    // of row 8 i + rr with the row's logical chunk p ^ f(row).  f_A(r) = bit 1 of r | bits 4, 5 of r << 1 (= l16 >> 1 of the lane
    // that reads row r under the cout permutation below), f_B(r) = (r >> 1) & 7.
    const int rr = lane >> 3, p = lane & 7;
    auto issue = [&](int ct, int pt, int st, int slot) {
        unsigned char* dst = wbase + slot * C1S_STAGE;
        const int koff = st * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int fa = ((rr >> 1) & 1) | (i & 6), fb = (4 * i + (rr >> 1)) & 7;
            const unsigned va = (unsigned)(ct * 64 + 8 * i + rr) * rowbytes + (unsigned)((p ^ fa) << 4);
            const unsigned vb = (unsigned)(pt * 64 + 8 * i + rr) * rowbytes + (unsigned)((p ^ fb) << 4);   // (< 2^32: checked on the host)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void*)(dst + i * 1024), 16, (int)va, koff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_i, (lds_void*)(dst + 8192 + i * 1024), 16, (int)vb, koff, 0, 0);
        }
    };
    // fragment addresses: A block mt = LDS row 16 (l16 >> 2) + 4 mt + (l16 & 3) (MFMA row 4 q + j of block mt = cout 16 q + 4 mt + j: a lane
    // ends with 16 consecutive couts), B block nt = row 16 nt + l16; K-half h = logical chunk 4 h + lq, physical chunk ^ (l16 >> 1)
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)wbase;
    const unsigned ch0 = (unsigned)((lq ^ (l16 >> 1)) << 4), ch1 = (unsigned)(((4 + lq) ^ (l16 >> 1)) << 4);
    const unsigned ra = lds0 + (unsigned)((16 * (l16 >> 2) + (l16 & 3)) * 128), rb = lds0 + 8192u + (unsigned)(l16 * 128);

    int t = gw, ct = t % a.ntile_n, pt = t / a.ntile_n, slot = 0;
    issue(ct, pt, 0, 0);
    for (;;) {
        f32x4 acc[4][4];          // [16-cout block][16-pixel block]
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const f32x4 b = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + ct * 64 + 16 * lq + 4 * mt) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = b;
        }
        const int tn = t + GW;
        const int ctn = tn % a.ntile_n, ptn = tn / a.ntile_n;
        for (int st = 0; st < nstage; ++st) {
            // the stage after this one -- of this tile, or the first of the wave's next tile -- goes into the other slot (whose
            // fragments were read, and waited for, a stage ago)
            const bool more = st + 1 < nstage || tn < a.ntiles;
            if (st + 1 < nstage) issue(ct, pt, st + 1, slot ^ 1);
            else if (tn < a.ntiles) issue(ctn, ptn, 0, slot ^ 1);
            // this stage has landed: only the 16 loads just issued may still be in flight (vmcnt counts this wave's own loads and
            // stores in order: the epilogue's stores, older than those 16, are waited for as well)
            if (more) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bf16x8 af[2][4], bf[2][4];
            const unsigned so = (unsigned)(slot * C1S_STAGE);
            const unsigned a0 = ra + so + ch0, a1 = ra + so + ch1, b0 = rb + so + ch0, b1 = rb + so + ch1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[0][i]) : "v"(a0), "n"(i * 512));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[0][i]) : "v"(b0), "n"(i * 2048));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[1][i]) : "v"(a1), "n"(i * 512));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[1][i]) : "v"(b1), "n"(i * 2048));
            }
            asm volatile("s_waitcnt lgkmcnt(8)"
                         : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(bf[0][2]), "+v"(bf[0][3]));
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = E::mfma16(af[0][mt], bf[0][nt], acc[mt][nt]);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]), "+v"(bf[1][0]), "+v"(bf[1][1]), "+v"(bf[1][2]), "+v"(bf[1][3]));
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = E::mfma16(af[1][mt], bf[1][nt], acc[mt][nt]);
            slot ^= 1;
        }
        c1_epilogue<E>(a, acc, ct, pt, l16, lq, hw);
        if (tn >= a.ntiles) break;
        t = tn; ct = ctn; pt = ptn;
    }
}

// K <= 256 and a power-of-two number of cout tiles: the wave keeps ONE cout tile for its whole life and its 64 x K weight tile in
// REGISTERS (K / 32 x 4 fragments = at most 128 VGPRs, fetched once, in fragment layout); only pixel stages go through LDS: 8 KiB
// per 64-deep stage, a ring of four -- three stages in flight per wave instead of one, half the loads per stage and none of
// the weight re-reads (a wave tile of the form above fetches 4 x as many weight bytes from L2 as pixel bytes when Cout = 4 K).
template <typename E, int KS64>
__global__ __launch_bounds__(256, 1) void conv1x1_stream_wreg_kernel(C1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int gw = (int)blockIdx.x * 4 + wid, GW = (int)gridDim.x * 4;      // GW % ntile_n == 0 (host)
    const int ct = gw % a.ntile_n;
    const int pstep = GW / a.ntile_n, ptiles = a.ntiles / a.ntile_n;
    int pt = gw / a.ntile_n;
    if (pt >= ptiles) return;
    unsigned char* wbase = smem + wid * C1S_WAVE;
    const int hw = a.Ho * a.Wo;
    const unsigned rowbytes = (unsigned)a.K_pad * 2u;
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_i = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);

    bf16x8 af[2 * KS64][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#ifdef C1_ABL_WFM      /* timing-only ablation: the weight fragments as contiguous KiB (wrong values) -- what a fragment-major copy would cost */
        const unsigned ao = (unsigned)(((ct * 4 + mt) * 2 * KS64) * 1024 + lane * 16);
#pragma unroll
        for (int k = 0; k < 2 * KS64; ++k) af[k][mt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_w, ao, k * 1024, 0));
#else
        const unsigned ao = (unsigned)(((ct * 64 + 16 * (l16 >> 2) + 4 * mt + (l16 & 3)) * a.K_pad + lq * 8) * 2);
#pragma unroll
        for (int k = 0; k < 2 * KS64; ++k) af[k][mt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_w, ao, k * 64, 0));
#endif
    }
    f32x4 bias4[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        bias4[mt] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + ct * 64 + 16 * lq + 4 * mt) : f32x4{0.f, 0.f, 0.f, 0.f};

    const int rr = lane >> 3, p = lane & 7;
    auto issue = [&](int ptile, int st, int slot) {           // pixel stage: 64 rows x 128 B, rows 8 i .. 8 i + 7 per piece
        unsigned char* dst = wbase + slot * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int fb = (4 * i + (rr >> 1)) & 7;
            const unsigned vb = (unsigned)(ptile * 64 + 8 * i + rr) * rowbytes + (unsigned)((p ^ fb) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_i, (lds_void*)(dst + i * 1024), 16, (int)vb, st * 128, 0, 0);
        }
    };
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)wbase;
    const unsigned rb0 = lds0 + (unsigned)(l16 * 128) + (unsigned)((lq ^ (l16 >> 1)) << 4);
    const unsigned rb1 = lds0 + (unsigned)(l16 * 128) + (unsigned)(((4 + lq) ^ (l16 >> 1)) << 4);

    // flat stage index g = tile j * KS64 + st over this wave's tiles; stages g + 1 .. g + 3 are in flight while g is multiplied
    const int mytiles = (ptiles - pt + pstep - 1) / pstep;
    const int total = mytiles * KS64;
    int ipt = pt, ist = 0, issued = 0;                      // issue pointer
    auto issue_next = [&]() {
        if (issued < total) {
            issue(ipt, ist, issued & 3);
            ++issued;
            if (++ist == KS64) { ist = 0; ipt += pstep; }
        }
    };
    issue_next(); issue_next(); issue_next();
    int g = 0;
    for (int j = 0; j < mytiles; ++j, pt += pstep) {
        f32x4 acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = bias4[mt];
#pragma unroll
        for (int st = 0; st < KS64; ++st, ++g) {
            issue_next();
            // stage g has landed once at most the loads of the (up to three) younger stages are outstanding; stores of an epilogue
            // in between are older than some of them, so the count can only over-wait
            const int younger = issued - g - 1;
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bf16x8 bf[2][4];
            const unsigned so = (unsigned)((g & 3) * 8192);
            const unsigned b0 = rb0 + so, b1 = rb1 + so;
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[0][i]) : "v"(b0), "n"(i * 2048));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[1][i]) : "v"(b1), "n"(i * 2048));
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(bf[0][2]), "+v"(bf[0][3]));
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = E::mfma16(af[2 * st][mt], bf[0][nt], acc[mt][nt]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[1][0]), "+v"(bf[1][1]), "+v"(bf[1][2]), "+v"(bf[1][3]));
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = E::mfma16(af[2 * st + 1][mt], bf[1][nt], acc[mt][nt]);
        }
        c1_epilogue<E>(a, acc, ct, pt, l16, lq, hw);
    }
}

template <typename E>
static int conv1x1_dispatch(const void* in, const void* wgt, const float* bias, const void* res, void* out, int N,
                            int H, int W, int Cin, int Cout, int stride, int Ho, int Wo, int K_pad, int Cout_pad,
                            int relu, int res_mode, int Hr, int Wr, void* stream) {
    if (N <= 0) return CVPCE_OK;
    if (!in || !wgt || !out) return CVPCE_ERR_ARG;
    if (H <= 0 || W <= 0 || stride < 1 || Cin <= 0 || Cin % 64 != 0 || K_pad != Cin || Cout <= 0 || Cout % 64 != 0) return CVPCE_ERR_ARG;
    if (Cout_pad % 256 != 0 || Cout_pad < Cout) return CVPCE_ERR_ARG;
    if (Ho != (H - 1) / stride + 1 || Wo != (W - 1) / stride + 1) return CVPCE_ERR_ARG;
    if (res_mode < 0 || res_mode > 2 || (res_mode && !res) || (res_mode == 1 && (Hr != Ho || Wr != Wo))) return CVPCE_ERR_ARG;
    if (res_mode == 2 && (Hr <= 0 || Wr <= 0)) return CVPCE_ERR_ARG;
    if ((long long)N * H * W * Cin * 2 >= (1LL << 32) || (long long)Cout_pad * K_pad * 2 >= (1LL << 31)) return CVPCE_ERR_ARG;
    if ((long long)N * Ho * Wo >= (1LL << 31) - 64) return CVPCE_ERR_ARG;
    C1Args a;
    a.in = (const bf16_t*)in; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.res = (const bf16_t*)res; a.out = (bf16_t*)out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.stride = stride; a.Ho = Ho; a.Wo = Wo; a.K_pad = K_pad;
    a.M = N * Ho * Wo; a.relu = relu; a.res_mode = res_mode; a.Hr = Hr; a.Wr = Wr;
    a.ntile_n = Cout / 64;
    const long long nt = (long long)((a.M + 63) / 64) * a.ntile_n;
    if (nt >= (1LL << 31)) return CVPCE_ERR_ARG;
    a.ntiles = (int)nt;
    a.in_bytes = (unsigned)((long long)N * H * W * Cin * 2);
    a.wgt_bytes = (unsigned)((long long)Cout_pad * K_pad * 2);
    static const bool use_stream = !(getenv("CVPCE_C1_STREAM") && getenv("CVPCE_C1_STREAM")[0] == '0');   // dev A/B switch
    if (stride == 1 && use_stream) {
        if (!cvpce_smem_attr_done<conv1x1_stream_kernel<E>>((const void*)conv1x1_stream_kernel<E>, 4 * C1S_WAVE)) return CVPCE_ERR_LAUNCH;
        const int want = (a.ntiles + 3) / 4;
        int grid = want < g_cvpce_persistent_wgs ? want : g_cvpce_persistent_wgs;
        static const bool use_wreg = !(getenv("CVPCE_C1_WREG") && getenv("CVPCE_C1_WREG")[0] == '0');   // dev A/B switch
        const int ks64 = K_pad >> 6;
        if (use_wreg && (ks64 == 1 || ks64 == 2 || ks64 == 4) && (a.ntile_n & (a.ntile_n - 1)) == 0 && a.ntile_n <= 4 * grid &&
            (4 * grid) % a.ntile_n == 0) {
            if (!cvpce_smem_attr_done<conv1x1_stream_wreg_kernel<E, 4>>((const void*)conv1x1_stream_wreg_kernel<E, 4>, 4 * C1S_WAVE) ||
                !cvpce_smem_attr_done<conv1x1_stream_wreg_kernel<E, 2>>((const void*)conv1x1_stream_wreg_kernel<E, 2>, 4 * C1S_WAVE) ||
                !cvpce_smem_attr_done<conv1x1_stream_wreg_kernel<E, 1>>((const void*)conv1x1_stream_wreg_kernel<E, 1>, 4 * C1S_WAVE))
                return CVPCE_ERR_LAUNCH;
            if (ks64 == 4) hipLaunchKernelGGL((conv1x1_stream_wreg_kernel<E, 4>), dim3(grid), dim3(256), 4 * C1S_WAVE, (hipStream_t)stream, a);
            else if (ks64 == 2) hipLaunchKernelGGL((conv1x1_stream_wreg_kernel<E, 2>), dim3(grid), dim3(256), 4 * C1S_WAVE, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((conv1x1_stream_wreg_kernel<E, 1>), dim3(grid), dim3(256), 4 * C1S_WAVE, (hipStream_t)stream, a);
            return cvpce_check_launch();
        }
        hipLaunchKernelGGL(conv1x1_stream_kernel<E>, dim3(grid), dim3(256), 4 * C1S_WAVE, (hipStream_t)stream, a);
        return cvpce_check_launch();
    }
    hipLaunchKernelGGL(conv1x1_kernel<E>, dim3((a.ntiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    return cvpce_check_launch();
}

extern "C" int cvpce_conv1x1_nhwc_bf16(const void* in, const void* wgt, const float* bias, const void* res, void* out, int N,
                                       int H, int W, int Cin, int Cout, int stride, int Ho, int Wo, int K_pad, int Cout_pad,
                                       int relu, int res_mode, int Hr, int Wr, void* stream) {
    return conv1x1_dispatch<ElemBF16>(in, wgt, bias, res, out, N, H, W, Cin, Cout, stride, Ho, Wo, K_pad, Cout_pad, relu, res_mode, Hr, Wr, stream);
}
extern "C" int cvpce_conv1x1_nhwc_f16(const void* in, const void* wgt, const float* bias, const void* res, void* out, int N,
                                      int H, int W, int Cin, int Cout, int stride, int Ho, int Wo, int K_pad, int Cout_pad,
                                      int relu, int res_mode, int Hr, int Wr, void* stream) {
    return conv1x1_dispatch<ElemF16>(in, wgt, bias, res, out, N, H, W, Cin, Cout, stride, Ho, Wo, K_pad, Cout_pad, relu, res_mode, Hr, Wr, stream);
}
