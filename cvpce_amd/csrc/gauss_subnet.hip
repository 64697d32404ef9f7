// The detector's whole GaussianSubnet in ONE launch (/root/reference/cvpce/models/proposals.py:81-107):
//   up2(x) -> conv3x3 64->32 + ReLU -> conv3x3 32->32 + ReLU -> conv3x3 32->16 + ReLU -> conv1x1 16->16 + ReLU -> conv1x1 16->1 + ReLU | Tanh
// x is GaussianLayer's 64-channel output at half the output resolution (its `self.up`, proposals.py:79, is never materialised).
//
// As four launches (three of csrc/thin3x3.hip + the pointwise tail of csrc/elementwise.hip) the chain wrote and re-read two 32-channel
// and one 16-channel map of the full 400 x 400 output size (33 MB per image) for 10.4 GFLOP, and its 16->16->1 tail was a VALU kernel at
// 0.02 of the HBM rate.  Here NOTHING but x and the one-channel fp32 result touches memory:
//   * a WAVE owns a strip of 28 output columns and walks down it, one output row per iteration; all five layers' weights (65 KB) live in
//     its registers for its whole life (281 VGPRs: one wave per SIMD, four independent waves per workgroup, no barrier anywhere);
//   * the three 3x3 layers form a software pipeline over rows, skewed by two rows per layer so that an iteration's fragment reads depend
//     only on rows written in EARLIER iterations: iteration t computes layer-1 row a = y0 - 2 + t into a private 4-row ring in LDS, layer-2 row
//     a - 2 from that ring into a second ring, layer-3 row a - 4 from the second ring into registers, and finishes that row through the two
//     pointwise layers as MFMA 16x16x16 straight on the accumulator layout (no cross-lane traffic); lanes of row 0 store 16 floats;
//   * column halo: every layer computes two groups of 16 columns; layer 1 covers columns X - 2 .. X + 29, layer 2 X - 1 .. X + 28 (+2 unused),
//     layer 3 X .. X + 27 (+4 unused): 32 / 28 = 1.14 x the useful MFMAs instead of a 3x3 tile's (T + 4)^2 / T^2;
//   * x rows arrive by `buffer_load ... lds` into a 4-row ring of STORED rows (18 stored pixels x 128 B under the 34 logical ones), two
//     stored rows ahead; rows / columns outside the image are outside the buffer's range and arrive as zeros (layer 1's zero padding);
//     layer-1 / layer-2 values at positions outside the image are stored as ZEROS (they are the next layer's zero padding, not conv outputs).
// Numerics: every layer exactly as the unfused kernels -- fp32 accumulation over the same K order (tap-major), one rounding of each layer's
// output to the storage type E where the unfused form stored it; the pointwise tail accumulates inside MFMA 16x16x16 instead of an fmaf chain.
#include "common.h"
#include "../../include/cvpce_amd.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

#ifndef GS_ABL
#define GS_ABL 0       // compile-time timing experiments (never set in the shipped library): 1 no 3x3 MFMAs, 2 no fragment reads, 4 no x DMA / waits, 8 no ring stores
#endif
// diagnostic build only (-DGS_STAMP, tools/dev/build_variant.sh): wave 0 of workgroup 0 sums the shader clocks (s_memtime) it spends in the
// iteration's parts [top (DMA issue + wait), region 1, 2, 3, 4, 5, iterations]; no output value depends on the stamps
#ifdef GS_STAMP
__device__ unsigned long long cvpce_gauss_subnet_stamps[8];
extern "C" int cvpce_debug_gauss_subnet_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cvpce_gauss_subnet_stamps), sizeof(cvpce_gauss_subnet_stamps)) == hipSuccess ? 0 : 2;
}
#define GS_T(I) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_[I] += now_ - tic_; tic_ = now_; }
#else
#define GS_T(I)
#endif
#define GS_STRIP 28
#define GS_IN_SLOT 3072        // a stored x row: 18 pixels x 128 B in three DMA pieces of 1 KiB (the third: 2 pixels + zeros)
#define GS_MID_SLOT 2304       // a layer-1 / layer-2 row: 36 pixel slots x 64 B (32 written, reads reach slot 33)
#define GS_IN_OFF 0
#define GS_L1_OFF (4 * GS_IN_SLOT)
#define GS_L2_OFF (GS_L1_OFF + 4 * GS_MID_SLOT)
#define GS_DUMMY_OFF (GS_L2_OFF + 4 * GS_MID_SLOT)  // 1 KiB nobody reads: where the lanes that need no zero fix-up send theirs (branch-free)
#define GS_BIAS_OFF (GS_DUMMY_OFF + 1024)           // b1 [32] b2 [32] b3 [16] b4 [16] f32: the accumulators' start values, read where a chain starts
#define GS_WAVE (GS_BIAS_OFF + 512)                 // 32 256 B per wave, 126 KiB per workgroup

struct GaussSubnetArgs {
    const bf16_t* x;                       // [N][H/2][W/2][64]
    const bf16_t *w1, *w2, *w3, *w4, *w5;  // row-major [Cout_pad][K_pad]: K = 576 / 288 / 288 (tap-major, channel fastest) / k4_pad / k5_pad
    const float *b1, *b2, *b3, *b4, *b5;   // [32] [32] [16] [16] [1] or null
    float* out;                            // [N][H][W]
    int N, H, W, act;
    int k4_pad, k5_pad;
    int strips, chunks, rows_per_task, ntasks;
    unsigned x_bytes, out_bytes;
};

// Ring swizzles.  ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...): a group mixes two
// K-quarters (lq) over complementary halves of the 16 pixel lanes.  With the swizzle first taken over from csrc/thin3x3.hip every fragment read
// was a two-way bank conflict (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE) and the kernel was bound by LDS cycles (4 waves x 80 KiB of
// fragments per iteration).  These two are conflict-free for every tap column and both column groups (exhaustive check over the lane groups):
//   mid rings (64-byte pixels, slot p):            chunk c of pixel p at physical chunk c ^ ((p >> 1) & 3)
//   x ring (128-byte stored pixels, patch pixel sp): chunk c at c ^ h(sp), h = (sp & 6) ^ ((sp >> 1) & 4); h(sp + 8) = h(sp) ^ 4, which is what
//   makes column group 1's K-step ks sit exactly 1 024 bytes behind group 0's K-step 1 - ks.
__device__ __forceinline__ int gs_swz_m(int p) { return (p >> 1) & 3; }
__device__ __forceinline__ int gs_swz_x(int sp) { return (sp & 6) ^ ((sp >> 1) & 4); }

template <typename E> struct GsTail;
template <> struct GsTail<ElemBF16> {
    static __device__ __forceinline__ f32x4 mfma(bf16x4 a, bf16x4 b, f32x4 c) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    }
};
template <> struct GsTail<ElemF16> {
    static __device__ __forceinline__ f32x4 mfma(bf16x4 a, bf16x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
    }
};

// ReLU + the rounding to the storage type of four accumulator values.  fp16: the clamp of the store (+-65504) and the ReLU are ONE
// v_med3_f32(x, 0, 65504) per value; bf16: the ReLU on the bit pattern, no clamp.
template <typename E>
__device__ __forceinline__ bf16x4 gs_relu_pack(f32x4 v) {
    if constexpr (E::kF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j], 0.f, 65504.f);
        return __builtin_bit_cast(bf16x4, __builtin_convertvector(v, f16x4));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = relu_bits(v[j]);
        return __builtin_convertvector(v, bf16x4);
    }
}

// One wave per SIMD: whatever is not an MFMA has to be issued INSIDE the MFMA chains (an MFMA 16x16x32 holds the vector issue for 8 of
// its 16 clocks: two other instructions per MFMA are free, the rest add their full 4 clocks each -- as first built, with the chains back
// to back, an iteration took 4 970 clocks for 2 080 of MFMA).  The groups below tell hipcc's scheduler to deal the region's other
// instructions between the MFMAs: per MFMA pair, R ds_reads and V VALU instructions.
#define GS_INTERLEAVE(PAIRS, R, V)                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < (PAIRS); ++i_) {         \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);           \
        __builtin_amdgcn_sched_group_barrier(0x100, (R), 0);         \
        __builtin_amdgcn_sched_group_barrier(0x002, (V), 0);         \
    }

template <typename E, int ACT>
__global__ __launch_bounds__(256, 1) void gauss_subnet_kernel(GaussSubnetArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int gw = (int)blockIdx.x * 4 + wid, GW = (int)gridDim.x * 4;
    if (gw >= a.ntasks) return;
    unsigned char* wbase = smem + wid * GS_WAVE;
    const int Hs = a.H >> 1, Ws = a.W >> 1;

    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);

    // ---- weights, register-resident.  3x3 layers: lane (m = l16, q = lq) holds k = 8 q .. 8 q + 7 of MFMA row m; with two 16-cout blocks
    //      row m = 4 q' + j of block mt is cout 8 q' + 4 mt + j, so that an accumulator lane ends with 8 CONSECUTIVE couts of its pixel
    //      (one 16-byte LDS store); layer 3 (one block): row m is cout m.  Pointwise layers (MFMA 16x16x16): lane (m, q) holds k = 4 q .. 4 q + 3.
    //      Layers 1 and 2 (216 registers) are PINNED in the accumulator file ("+a"): their MFMAs read them there (A operands may be AGPRs);
    //      left to itself hipcc parks them there too but copies each back in front of its MFMA (311 v_accvgpr_read per iteration).
    const int row2 = 8 * (l16 >> 2) + (l16 & 3);
    bf16x8 w1f[18][2], w2f[9][2], w3f[9];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const bf16_t* r1 = a.w1 + (size_t)(row2 + 4 * mt) * 576 + lq * 8;
        const bf16_t* r2 = a.w2 + (size_t)(row2 + 4 * mt) * 288 + lq * 8;
#pragma unroll
        for (int t = 0; t < 18; ++t) w1f[t][mt] = *reinterpret_cast<const bf16x8*>(r1 + t * 32);
#pragma unroll
        for (int t = 0; t < 9; ++t) w2f[t][mt] = *reinterpret_cast<const bf16x8*>(r2 + t * 32);
    }
    {
        const bf16_t* r3 = a.w3 + (size_t)l16 * 288 + lq * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t) w3f[t] = *reinterpret_cast<const bf16x8*>(r3 + t * 32);
    }
#pragma unroll
    for (int t = 0; t < 18; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) asm volatile("" : "+a"(w1f[t][mt]));
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) asm volatile("" : "+a"(w2f[t][mt]));
#pragma unroll
    for (int t = 0; t < 9; ++t) asm volatile("" : "+a"(w3f[t]));
    const bf16x4 w4f = *reinterpret_cast<const bf16x4*>(a.w4 + (size_t)l16 * a.k4_pad + lq * 4);
    // the 16 -> 1 layer as MFMA row 0 for column group 0 and as row 4 for group 1: both groups accumulate into ONE register -- lanes
    // 0 .. 15 end with group 0's 16 pixels, lanes 16 .. 31 with group 1's: one 128-byte store per output row
    const bf16x4 w5raw = *reinterpret_cast<const bf16x4*>(a.w5 + lq * 4), zero16 = __builtin_bit_cast(bf16x4, uint2{0u, 0u});
    const bf16x4 w5f0 = l16 == 0 ? w5raw : zero16, w5f1 = l16 == 4 ? w5raw : zero16;
    // the biases live in LDS (the vector file is full: with them in 24 registers hipcc had to move the accumulators from register to
    // register along a chain, and an MFMA whose C is not its D waits out the whole MFMA before it: 27 clocks per MFMA instead of 16)
    {
        float* bl = reinterpret_cast<float*>(wbase + GS_BIAS_OFF);
        if (lane < 32) { bl[lane] = a.b1 ? a.b1[lane] : 0.f; bl[32 + lane] = a.b2 ? a.b2[lane] : 0.f; }
        if (lane < 16) { bl[64 + lane] = a.b3 ? a.b3[lane] : 0.f; bl[80 + lane] = a.b4 ? a.b4[lane] : 0.f; }
    }
    const float bias5 = a.b5 ? a.b5[0] : 0.f;

    // ---- fragment addresses (LDS byte addresses of this lane inside ring slot 0; a slot base and, for column group 1, 1 024 are added).
    // x ring: L1 column group g, tap column kw reads the stored pixel under logical column X - 3 + 16 g + l16 + kw: patch pixel
    //   sp = (16 g + l16 + kw + 1) >> 1 of the 18 (patch pixel 0 is stored column X / 2 - 2); its 8 chunks are swizzled by gs_swz_x(sp)
    //   and K-step ks of a tap is chunk 4 ks + lq.  Group 1 is patch pixel sp + 8, whose swizzle differs by 4: its K-step ks sits 1 024
    //   bytes behind group 0's K-step 1 - ks.
    // mid rings: group g, tap column kw reads pixel slot p = 16 g + l16 + kw; its 4 chunks are swizzled by gs_swz_m(p) (group 1: + 1 024).
    const unsigned lds_w = (unsigned)(size_t)(lds_char*)wbase;
    unsigned lx[3][2], lm[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int sp = (l16 + kw + 1) >> 1, p = l16 + kw;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) lx[kw][ks] = lds_w + GS_IN_OFF + (unsigned)(sp * 128 + (((4 * ks + lq) ^ gs_swz_x(sp)) << 4));
        lm[kw] = lds_w + (unsigned)(p * 64 + ((lq ^ gs_swz_m(p)) << 4));
    }
    const unsigned ldummy = lds_w + GS_DUMMY_OFF + (unsigned)(lane * 16);
    const unsigned lb = lds_w + GS_BIAS_OFF + (unsigned)(lq * 32);      // b1 / b2: couts 8 lq + 4 mt ..; b3 / b4 (+ 256 / 320): couts 4 lq .. at lb - 16 lq
    auto ldsf4 = [](unsigned addr) { return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((size_t)addr); };
    const unsigned ls = lds_w + (unsigned)(l16 * 64 + ((lq ^ gs_swz_m(l16)) << 4));     // where this lane's 8 couts of pixel slot l16 go (group 1: + 1 024)
    const bf16x8 abl_frag = w3f[0];
    auto lds128 = [&](unsigned addr) {
        if constexpr (GS_ABL & 2) { bf16x8 v = abl_frag; asm volatile("" : "+v"(v) : "v"(addr)); return v; }
        else return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((size_t)addr);
    };
    auto sts128 = [](unsigned addr, u32x4 v) {
        if constexpr (GS_ABL & 8) asm volatile("" :: "v"(addr), "v"(v));
        else *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((size_t)addr) = v;
    };
    auto mfma3 = [](bf16x8 w, bf16x8 f, f32x4 c) {
        if constexpr (GS_ABL & 1) { asm volatile("" : "+v"(c) : "v"(f)); return c; }
        else return E::mfma16(w, f, c);
    };
    // DMA lane constants: piece j (0, 1, 2) lane L -> patch pixel 8 j + (L >> 3), physical chunk L & 7 (piece 2: lanes 0 .. 15 only)
    unsigned dchunk[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int sp = 8 * j + (lane >> 3);
        dchunk[j] = (unsigned)(((lane & 7) ^ gs_swz_x(sp)) << 4);
    }

    for (int task = gw; task < a.ntasks; task += GW) {
        const int chunk = task % a.chunks, strip_id = task / a.chunks;
        const int n = strip_id / a.strips, X = (strip_id - n * a.strips) * GS_STRIP;
        const int y0 = chunk * a.rows_per_task;
        const int y1 = y0 + a.rows_per_task < a.H ? y0 + a.rows_per_task : a.H;
        const int rows = y1 - y0;
        if (rows <= 0) continue;
        const int xs0 = (X >> 1) - 2;                                        // stored column of patch pixel 0
        // lanes whose layer-1 / layer-2 column lies outside the image (their values are the next layer's ZERO padding), and whether this
        // strip has any; the output columns this lane stores (lanes 0 .. 15: group 0, lanes 16 .. 31: group 1, the strip's 28 columns)
        bool z1[2], z2[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            z1[g] = (unsigned)(X - 2 + 16 * g + l16) >= (unsigned)a.W;
            z2[g] = (unsigned)(X - 1 + 16 * g + l16) >= (unsigned)a.W;
        }
        const int ocol = 16 * lq + l16;
        const bool ostore = lq < 2 && ocol < GS_STRIP && X + ocol < a.W;
        const unsigned obase = (unsigned)((n * a.H) * a.W + X + ocol) * 4u;
        bool dok[3];
        unsigned dcol[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int sx = xs0 + 8 * j + (lane >> 3);
            dok[j] = (unsigned)sx < (unsigned)Ws && (j < 2 || lane < 16);
            dcol[j] = (unsigned)sx * 128u + dchunk[j];
        }
        auto issue_row = [&](int r) {                                        // stored row r -> ring slot r & 3
            unsigned char* dst = wbase + GS_IN_OFF + (r & 3) * GS_IN_SLOT;
            const bool rok = (unsigned)r < (unsigned)Hs;
            const unsigned rbase = (unsigned)((n * Hs + r) * Ws) * 128u;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_x, (lds_void*)(dst + j * 1024), 16, (int)((rok && dok[j]) ? rbase + dcol[j] : 0xFFFFFFF0u), 0, 0, 0);
        };
        // the wave's previous task is done with the rings: all its fragment reads were consumed by MFMAs issued before this point
        const int r_first = (y0 - 3) >> 1, r_last = (y1 + 2) >> 1;          // stored rows under logical rows y0 - 3 .. y1 + 2 (arithmetic shifts)
        int next = r_first;
        for (; next <= r_last && next <= r_first + 2; ++next) issue_row(next);

        // One straight-line body for every iteration (no branch but the DMA's and the edge fix-ups): an iteration runs layer 3 on row rc,
        // layer 2 on row rb, layer 1 on row ra whether or not those rows belong to this task -- rows before a stage's first (the pipeline
        // filling) and after its last are computed from whatever the rings hold and are never read by a row that counts (a valid row reads
        // only rows written for it, see the header); their stores go to an out-of-range buffer offset.  The reads of a stage are issued
        // beside the MFMAs of the stage before it, layer 3's one iteration ahead; hipcc counts every wait.
        bf16x8 f3[18];                                                       // layer-3 fragments of the NEXT iteration's row (both column groups)
        auto read3 = [&](int row) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const unsigned sb = GS_L2_OFF + ((row + kh - 1) & 3) * GS_MID_SLOT;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const unsigned ad = lm[kw] + sb;
                    f3[kh * 3 + kw] = lds128(ad);
                    f3[9 + kh * 3 + kw] = lds128(ad + 1024);
                }
            }
        };
        read3(y0 - 6);                                                       // (iteration 0's layer-3 row: nothing of it is kept)
        const int iters = rows + 6;
#ifdef GS_STAMP
        unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tic_ = __builtin_amdgcn_s_memtime();
#endif
        for (int t = 0; t < iters; ++t) {
            const int ra = y0 - 2 + t, rb = ra - 2, rc = ra - 4;             // layer-1 / layer-2 / layer-3 (= output) row of this iteration
            if (!(GS_ABL & 4) && t <= rows + 3) {                            // layer 1 still has rows to compute: x rows in, wait for ra's
                const int r_need = (ra + 1) >> 1;
                if (next <= r_last && next <= r_need + 2) { issue_row(next); ++next; }
                const int younger = next - 1 - r_need;                       // stored rows issued after the last one this row reads
                if (younger >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const bool ok1 = (unsigned)ra < (unsigned)a.H, ok2 = (unsigned)rb < (unsigned)a.H;
            __builtin_amdgcn_sched_barrier(0);
            GS_T(0)

            // ---- region 1: layer 3 row rc on f3 (two interleaved chains) || layer-2 fragment reads ----
            bf16x8 f2[18];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const unsigned sb = GS_L1_OFF + ((rb + kh - 1) & 3) * GS_MID_SLOT;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const unsigned ad = lm[kw] + sb;
                    f2[kh * 3 + kw] = lds128(ad);
                    f2[9 + kh * 3 + kw] = lds128(ad + 1024);
                }
            }
            const f32x4 bias3 = ldsf4(lb - 16 * lq + 256);
            f32x4 a3[2] = {bias3, bias3};
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int g = 0; g < 2; ++g) a3[g] = mfma3(w3f[tp], f3[g * 9 + tp], a3[g]);
            GS_INTERLEAVE(9, 2, 1)
            __builtin_amdgcn_sched_barrier(0);
            GS_T(1)

            // ---- region 2: layer 2 row rb on f2 (four chains) || layer-1 group-0 fragment reads; the pointwise tail of row rc ----
            bf16x8 f1a[18], f1b[18];
            unsigned xa[3][3][2];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const unsigned sb = (unsigned)((((ra + kh - 1) >> 1) & 3) * GS_IN_SLOT);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        xa[kh][kw][ks] = lx[kw][ks] + sb;
                        f1a[(kh * 3 + kw) * 2 + ks] = lds128(xa[kh][kw][ks]);
                    }
            }
            f32x4 a2[2][2] = {{ldsf4(lb + 128), ldsf4(lb + 144)}, {ldsf4(lb + 128), ldsf4(lb + 144)}};
            const f32x4 bias4 = ldsf4(lb - 16 * lq + 320);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) a2[g][mt] = mfma3(w2f[tp][mt], f2[g * 9 + tp], a2[g][mt]);
            f32x4 z4 = {bias5, 0.f, 0.f, 0.f};
            {
                f32x4 h0 = GsTail<E>::mfma(w4f, gs_relu_pack<E>(a3[0]), bias4), h1 = GsTail<E>::mfma(w4f, gs_relu_pack<E>(a3[1]), bias4);
                z4 = GsTail<E>::mfma(w5f0, gs_relu_pack<E>(h0), z4);
                z4 = GsTail<E>::mfma(w5f1, gs_relu_pack<E>(h1), z4);
            }
            GS_INTERLEAVE(18, 1, 2)
            __builtin_amdgcn_sched_barrier(0);
            GS_T(2)

            // ---- region 3: layer 1 row ra, column group 0 on f1a || group-1 fragment reads; store row rc; layer-2 row rb into its ring ----
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) f1b[(kh * 3 + kw) * 2 + ks] = lds128(xa[kh][kw][1 - ks] + 1024);
            f32x4 a1[2] = {ldsf4(lb), ldsf4(lb + 16)};
#pragma unroll
            for (int tp = 0; tp < 18; ++tp)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) a1[mt] = mfma3(w1f[tp][mt], f1a[tp], a1[mt]);
            {
                // tanh(z) = 1 - 2 / (1 + e^(2 z)): exact limits at +-inf, absolute error ~1e-7 (v_exp_f32 + v_rcp_f32)
                const float z = z4[0];
                const float th = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(z * 2.885390081777927f));
                const float o = ACT == 2 ? th : (ACT == 1 ? fmaxf(z, 0.f) : z);
                const bool st_ok = ostore && t >= 6;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), srd_o, st_ok ? obase + (unsigned)(rc * a.W) * 4u : 0xFFFFFFF0u, 0, 0);
            }
            const unsigned d2 = ls + GS_L2_OFF + (unsigned)((rb & 3) * GS_MID_SLOT);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint2 lo = __builtin_bit_cast(uint2, gs_relu_pack<E>(a2[g][0])), hi = __builtin_bit_cast(uint2, gs_relu_pack<E>(a2[g][1]));
                sts128(d2 + 1024 * g, u32x4{lo.x, lo.y, hi.x, hi.y});
            }
            GS_INTERLEAVE(18, 1, 2)
            __builtin_amdgcn_sched_barrier(0);
            // positions outside the image are layer 3's ZERO padding, not conv outputs: a second store of zeros behind the first (LDS is in order);
            // lanes that need none send theirs to a dummy KiB -- no branch, the iteration stays one scheduling block
#pragma unroll
            for (int g = 0; g < 2; ++g) sts128((z2[g] || !ok2) ? d2 + 1024 * g : ldummy, u32x4{0u, 0u, 0u, 0u});
            __builtin_amdgcn_sched_barrier(0);
            GS_T(3)

            // ---- region 4: layer 1 row ra, column group 1 on f1b || the NEXT iteration's layer-3 fragment reads (its rows rc .. rc + 2 of the
            //      layer-2 ring: rc + 2 = rb was stored in region 3); group 0 into the layer-1 ring ----
            read3(rc + 1);
            f32x4 a1b[2] = {ldsf4(lb), ldsf4(lb + 16)};
#pragma unroll
            for (int tp = 0; tp < 18; ++tp)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) a1b[mt] = mfma3(w1f[tp][mt], f1b[tp], a1b[mt]);
            const unsigned d1 = ls + GS_L1_OFF + (unsigned)((ra & 3) * GS_MID_SLOT);
            {
                const uint2 lo = __builtin_bit_cast(uint2, gs_relu_pack<E>(a1[0])), hi = __builtin_bit_cast(uint2, gs_relu_pack<E>(a1[1]));
                sts128(d1, u32x4{lo.x, lo.y, hi.x, hi.y});
            }
            GS_INTERLEAVE(18, 1, 1)
            __builtin_amdgcn_sched_barrier(0);
            GS_T(4)
            // ---- group 1 into the layer-1 ring (the next iteration's layer-2 reads need this row) ----
            {
                const uint2 lo = __builtin_bit_cast(uint2, gs_relu_pack<E>(a1b[0])), hi = __builtin_bit_cast(uint2, gs_relu_pack<E>(a1b[1]));
                sts128(d1 + 1024, u32x4{lo.x, lo.y, hi.x, hi.y});
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) sts128((z1[g] || !ok1) ? d1 + 1024 * g : ldummy, u32x4{0u, 0u, 0u, 0u});
#ifdef GS_STAMP
            __builtin_amdgcn_sched_barrier(0);
            GS_T(5)
            st_[6] += 1;
#endif
        }
#ifdef GS_STAMP
        if (gw == 0) for (int i = 0; i < 8; ++i) cvpce_gauss_subnet_stamps[i] = st_[i];
#endif
        // the next task's first DMA must not overwrite x rows whose fragment reads are still in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <typename E>
static int gauss_subnet_dispatch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                                 const void* w4, const float* b4, int k4_pad, const void* w5, const float* b5, int k5_pad, float* out, int N, int H, int W,
                                 int act, void* stream) {
    if (N <= 0) return CVPCE_OK;
    if (!x || !w1 || !w2 || !w3 || !w4 || !w5 || !out || H <= 0 || W <= 0 || (H & 1) || (W & 1) || act < 0 || act > 2) return CVPCE_ERR_ARG;
    if (k4_pad < 16 || (k4_pad & 3) || k5_pad < 16) return CVPCE_ERR_ARG;
    if ((long long)N * (H / 2) * (W / 2) * 128 >= (1LL << 32) - 65536 || (long long)N * H * W * 4 >= (1LL << 32) - 65536) return CVPCE_ERR_ARG;
    GaussSubnetArgs a;
    a.x = (const bf16_t*)x;
    a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.w3 = (const bf16_t*)w3; a.w4 = (const bf16_t*)w4; a.w5 = (const bf16_t*)w5;
    a.b1 = b1; a.b2 = b2; a.b3 = b3; a.b4 = b4; a.b5 = b5;
    a.out = out; a.N = N; a.H = H; a.W = W; a.act = act; a.k4_pad = k4_pad; a.k5_pad = k5_pad;
    a.strips = (W + GS_STRIP - 1) / GS_STRIP;
    a.x_bytes = (unsigned)((long long)N * (H / 2) * (W / 2) * 128);
    a.out_bytes = (unsigned)((long long)N * H * W * 4);
    // Row chunks per strip: the split that minimises rounds x iterations per task over the wave slots (4 per compute unit; a task of r rows
    // runs r + 6 iterations).  A function of (N, H, W) only -- every output pixel's arithmetic is the same whatever the split.
    const long long slots = 4LL * g_cvpce_persistent_wgs;
    const long long col_tasks = (long long)N * a.strips;
    int best_chunks = 1;
    long long best_cost = -1;
    for (int c = 1; c <= (H + 7) / 8; ++c) {
        const int rpt = (H + c - 1) / c;
        const long long tasks = col_tasks * ((H + rpt - 1) / rpt);
        const long long cost = ((tasks + slots - 1) / slots) * (rpt + 6);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_chunks = c; }
    }
    a.rows_per_task = (H + best_chunks - 1) / best_chunks;
    a.chunks = (H + a.rows_per_task - 1) / a.rows_per_task;
    const long long nt = col_tasks * a.chunks;
    if (nt >= (1LL << 31)) return CVPCE_ERR_ARG;
    a.ntasks = (int)nt;
    const int want = (a.ntasks + 3) / 4;
    const dim3 grid(want < g_cvpce_persistent_wgs ? want : g_cvpce_persistent_wgs);      // one workgroup (120 KiB of LDS, 4 waves of ~400 VGPRs) per compute unit
    hipStream_t s = (hipStream_t)stream;
    if (act == 2) {
        if (!cvpce_smem_attr_done<gauss_subnet_kernel<E, 2>>((const void*)gauss_subnet_kernel<E, 2>, 4 * GS_WAVE)) return CVPCE_ERR_LAUNCH;
        hipLaunchKernelGGL((gauss_subnet_kernel<E, 2>), grid, dim3(256), 4 * GS_WAVE, s, a);
    } else if (act == 1) {
        if (!cvpce_smem_attr_done<gauss_subnet_kernel<E, 1>>((const void*)gauss_subnet_kernel<E, 1>, 4 * GS_WAVE)) return CVPCE_ERR_LAUNCH;
        hipLaunchKernelGGL((gauss_subnet_kernel<E, 1>), grid, dim3(256), 4 * GS_WAVE, s, a);
    } else {
        if (!cvpce_smem_attr_done<gauss_subnet_kernel<E, 0>>((const void*)gauss_subnet_kernel<E, 0>, 4 * GS_WAVE)) return CVPCE_ERR_LAUNCH;
        hipLaunchKernelGGL((gauss_subnet_kernel<E, 0>), grid, dim3(256), 4 * GS_WAVE, s, a);
    }
    return cvpce_check_launch();
}

extern "C" int cvpce_gauss_subnet_bf16(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                                       const void* w4, const float* b4, int k4_pad, const void* w5, const float* b5, int k5_pad, float* out, int N, int H,
                                       int W, int act, void* stream) {
    return gauss_subnet_dispatch<ElemBF16>(x, w1, b1, w2, b2, w3, b3, w4, b4, k4_pad, w5, b5, k5_pad, out, N, H, W, act, stream);
}
extern "C" int cvpce_gauss_subnet_f16(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                                      const void* w4, const float* b4, int k4_pad, const void* w5, const float* b5, int k5_pad, float* out, int N, int H,
                                      int W, int act, void* stream) {
    return gauss_subnet_dispatch<ElemF16>(x, w1, b1, w2, b2, w3, b3, w4, b4, k4_pad, w5, b5, k5_pad, out, N, H, W, act, stream);
}
