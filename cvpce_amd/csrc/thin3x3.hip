// 3x3 / stride 1 / pad 1 convolutions with 32 input channels and 16 or 32 output channels: the thin layers of the detector's
// Gaussian subnet (/root/reference/cvpce/models/proposals.py:81-107: 3x3 32 -> 32 and 32 -> 16 over the 400 x 400 map, ReLU).
//
// As implicit GEMMs these layers ran at 140-240 TFLOP/s: 164 / 123 MB of traffic took 99 / 86 us (8 images).  Their weights are tiny
// (9 x 32 x 32 values = 18 MFMA fragments), so a wave keeps them in REGISTERS for its whole life and the kernel is a stream:
//   * a wave owns a column strip of 16 output pixels and walks DOWN it: every new output row needs ONE new input row (18 pixels x
//     64 bytes), fetched into a private ring of 8 row slots in LDS by `buffer_load ... lds` (4 neighbouring lanes = one pixel's 64
//     contiguous bytes: the texture addresser coalesces them), 4 rows ahead; rows / pixels outside the image lie
//     outside the buffer's range and arrive as zeros (the padding);
//   * the 9 pixel fragments of an output row are 9 ds_read_b128 (pixel p's chunks XOR-swizzled by (p >> 2) & 3: conflict-free),
//     then 9 (18) MFMAs 16x16x32 with M = couts, N = the 16 pixels; bias + ReLU; every lane stores its pixel's 8 (4) consecutive couts;
//   * no barrier: a wave waits for its own loads with a counted vmcnt.  Persistent waves over (image, strip, 50-row chunk) tasks.
// UP form (the subnet's first layer, 3x3 64 -> 32 over the nearest-2x upsample of the 200 x 200 x 64 GaussianLayer output, which is never
// materialised): the ring holds STORED rows (logical row r = stored row r >> 1: one new row every two output rows), a row slot is the 10
// stored pixels x 128 bytes under the 18 logical ones (8 neighbouring lanes = one pixel's 128-byte line), two K-steps per tap.
// Numerics as cvpce_conv2d_nhwc_bf16 (fp32 accumulation over the same K order tap-major, one rounding of the output).
#include "common.h"
#include "../../include/cvpce_amd.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

#define T3_RING 8
#define T3_SLOT 2048          // bytes of a row slot: 18 pixels x 64 B in two DMA pieces of 1 KiB
#define T3_AHEAD 4
#define T3_WAVE (T3_RING * T3_SLOT)

struct Thin3Args {
    const bf16_t* in;    // [N][H][W][32]
    const bf16_t* wgt;   // row-major [Cout_pad][K_pad = 288], k = (kh * 3 + kw) * 32 + ci
    const float* bias;   // [Cout] or null
    bf16_t* out;         // [N][H][W][Cout]
    int N, H, W, K_pad, relu;
    int segs, rows_per_task, tasks_per_strip, ntasks;
    unsigned in_bytes, wgt_bytes;
};

template <typename E, int MB, bool UP = false>      // MB 16-cout blocks: Cout = 16 MB; UP: Cin = 64 read through a nearest-2x upsample
__global__ __launch_bounds__(256, 2) void thin3x3_kernel(Thin3Args a) {
    constexpr int KS = UP ? 2 : 1;                  // 32-channel K-steps per tap
    constexpr int PIXB = UP ? 128 : 64;             // bytes per input pixel
    const int Hs = UP ? a.H >> 1 : a.H, Ws = UP ? a.W >> 1 : a.W;      // stored input size (H, W: the output = logical input size)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int gw = (int)blockIdx.x * 4 + wid, GW = (int)gridDim.x * 4;
    if (gw >= a.ntasks) return;
    unsigned char* wbase = smem + wid * T3_WAVE;
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)wbase;
    constexpr int COUT = 16 * MB;

    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_i = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);

    // weights: lane (m = l16, q = lq) holds k = 8 q .. 8 q + 7 of MFMA row m.  MB = 2: row m = 4 q' + j of block mt is cout 8 q' + 4 mt + j,
    // so that an accumulator lane ends with 8 consecutive couts of its pixel; MB = 1: row m is cout m (4 consecutive couts per lane).
    bf16x8 wf[9 * KS][MB];       // [tap * KS + K-step]
#pragma unroll
    for (int mt = 0; mt < MB; ++mt) {
        const int row = MB == 2 ? 8 * (l16 >> 2) + 4 * mt + (l16 & 3) : l16;
        const unsigned wo = (unsigned)((row * a.K_pad + lq * 8) * 2);
#pragma unroll
        for (int t = 0; t < 9 * KS; ++t) wf[t][mt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_w, wo, t * 64, 0));
    }
    f32x4 bias4[MB];
#pragma unroll
    for (int mt = 0; mt < MB; ++mt) {
        const int co = MB == 2 ? 8 * lq + 4 * mt : 4 * lq;
        bias4[mt] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // fragment addresses inside a row slot: tap column kw reads pixel p = l16 + kw (patch pixel 0 is image column x0 - 1)
    // (UP: logical pixel x0 - 1 + l16 + kw is stored patch pixel (l16 + kw + 1) >> 1 of the 10 -- patch pixel 0 is stored column
    // x0 / 2 - 1 --, its 8 chunks swizzled by (p >> 1) & 7; K-step s of a tap reads chunk 4 s + lq)
    unsigned foff[3][KS];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int p = UP ? (l16 + kw + 1) >> 1 : l16 + kw;
            foff[kw][ks] = UP ? (unsigned)(p * 128 + (((4 * ks + lq) ^ ((p >> 1) & 7)) << 4)) : (unsigned)(p * 64 + ((lq ^ ((p >> 2) & 3)) << 4));
        }
    // DMA lane constants: piece A lane L -> patch pixel L >> 2, physical chunk L & 3; piece B lanes 0 .. 7 -> patch pixels 16, 17
    // (UP: patch pixel L >> 3, physical chunk L & 7; piece B lanes 0 .. 15 -> patch pixels 8, 9)
    const int pA = UP ? lane >> 3 : lane >> 2, pB = UP ? 8 + (lane >> 3) : 16 + (lane >> 2), cph = UP ? lane & 7 : lane & 3;
    const unsigned chA = UP ? (unsigned)((cph ^ ((pA >> 1) & 7)) << 4) : (unsigned)((cph ^ ((pA >> 2) & 3)) << 4);
    const unsigned chB = UP ? (unsigned)((cph ^ ((pB >> 1) & 7)) << 4) : (unsigned)((cph ^ ((pB >> 2) & 3)) << 4);
    constexpr int NB_LANES = UP ? 16 : 8;

    for (int task = gw; task < a.ntasks; task += GW) {
        const int strip = task / a.tasks_per_strip, yc = task - strip * a.tasks_per_strip;
        const int n = strip / a.segs, x0 = (strip - n * a.segs) * 16;
        const int y0 = yc * a.rows_per_task;
        const int y1 = y0 + a.rows_per_task < a.H ? y0 + a.rows_per_task : a.H;       // output rows y0 .. y1 - 1, input rows y0 - 1 .. y1
        // (stored) image columns of this lane's patch pixels; r below is a STORED row
        const int xs0 = UP ? (x0 >> 1) - 1 : x0 - 1;
        const int xA = xs0 + pA, xB = xs0 + pB;
        const bool okA = (unsigned)xA < (unsigned)Ws, okB = lane < NB_LANES && (unsigned)xB < (unsigned)Ws;
        auto issue_row = [&](int r) {
            unsigned char* dst = wbase + ((r + 1) & (T3_RING - 1)) * T3_SLOT;
            const bool rok = (unsigned)r < (unsigned)Hs;
            const unsigned rbase = (unsigned)((n * Hs + r) * Ws) * (unsigned)PIXB;
            const unsigned va = (rok && okA) ? rbase + (unsigned)xA * (unsigned)PIXB + chA : 0xFFFFFFF0u;
            const unsigned vb = (rok && okB) ? rbase + (unsigned)xB * (unsigned)PIXB + chB : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_i, (lds_void*)dst, 16, (int)va, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_i, (lds_void*)(dst + 1024), 16, (int)vb, 0, 0, 0);
        };
        // a new task re-uses the ring: every fragment read of the previous task was waited for (lgkmcnt(0)) before its last MFMAs
        // stored rows needed: first = row of logical row y0 - 1, last = row of logical row y1 (UP: an arithmetic shift, -1 >> 1 = -1)
        const int r_first = UP ? (y0 - 1) >> 1 : y0 - 1, r_last = UP ? y1 >> 1 : y1;
        int next = r_first;
        for (; next <= r_last && next < r_first + 1 + T3_AHEAD; ++next) issue_row(next);
        const int xo = x0 + l16;
        for (int y = y0; y < y1; ++y) {
            const int r_need = UP ? (y + 1) >> 1 : y + 1;           // the last stored row output row y reads
            if (next <= r_last && next <= r_need + T3_AHEAD - 1) { issue_row(next); ++next; }
            // the rows up to r_need have landed once at most the pieces of the rows issued after it are outstanding (stores of
            // earlier output rows, older than those pieces, can only make the wait longer)
            const int younger = next - 1 - r_need;
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bf16x8 fr[9 * KS];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int r = UP ? (y + kh - 1) >> 1 : y + kh - 1;                                  // stored row of logical row y + kh - 1
                const unsigned sb = lds0 + (unsigned)(((r + 1) & (T3_RING - 1)) * T3_SLOT);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        asm volatile("ds_read_b128 %0, %1" : "=v"(fr[(kh * 3 + kw) * KS + ks]) : "v"(sb + foff[kw][ks]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]), "+v"(fr[4]), "+v"(fr[5]), "+v"(fr[6]), "+v"(fr[7]), "+v"(fr[8]));
            if constexpr (UP)
                asm volatile("" : "+v"(fr[9]), "+v"(fr[10]), "+v"(fr[11]), "+v"(fr[12]), "+v"(fr[13]), "+v"(fr[14]), "+v"(fr[15]), "+v"(fr[16]), "+v"(fr[17]));
            f32x4 acc[MB];
#pragma unroll
            for (int mt = 0; mt < MB; ++mt) acc[mt] = bias4[mt];
#pragma unroll
            for (int t = 0; t < 9 * KS; ++t)
#pragma unroll
                for (int mt = 0; mt < MB; ++mt) acc[mt] = E::mfma16(wf[t][mt], fr[t], acc[mt]);
            if (a.relu) {
#pragma unroll
                for (int mt = 0; mt < MB; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mt][j] = relu_bits(acc[mt][j]);
            }
            if (xo < a.W) {
                bf16_t* dst = a.out + ((size_t)(n * a.H + y) * a.W + xo) * COUT;
                if constexpr (MB == 2) {
                    const uint2 l2 = __builtin_bit_cast(uint2, E::pack4(acc[0])), h2 = __builtin_bit_cast(uint2, E::pack4(acc[1]));
                    *reinterpret_cast<u32x4*>(dst + 8 * lq) = u32x4{l2.x, l2.y, h2.x, h2.y};
                } else {
                    *reinterpret_cast<uint2*>(dst + 4 * lq) = __builtin_bit_cast(uint2, E::pack4(acc[0]));
                }
            }
        }
    }
}

template <typename E>
static int thin3x3_dispatch(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W, int Cin, int Cout,
                            int K_pad, int Cout_pad, int relu, int in_up_shift, void* stream) {
    // H, W: the OUTPUT size (= the input's, or twice the stored input's with in_up_shift = 1)
    if (N <= 0) return CVPCE_OK;
    if (!in || !wgt || !out || H <= 0 || W <= 0) return CVPCE_ERR_ARG;
    if (in_up_shift == 0) {
        if (Cin != 32 || (Cout != 16 && Cout != 32) || K_pad != 288) return CVPCE_ERR_ARG;
    } else if (in_up_shift == 1) {
        if (Cin != 64 || Cout != 32 || K_pad != 576 || (H & 1) || (W & 1)) return CVPCE_ERR_ARG;
    } else {
        return CVPCE_ERR_ARG;
    }
    if (Cout_pad < Cout || relu < 0 || relu > 1) return CVPCE_ERR_ARG;
    if ((long long)N * H * W * 64 >= (1LL << 32) - 65536) return CVPCE_ERR_ARG;
    Thin3Args a;
    a.in = (const bf16_t*)in; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.out = (bf16_t*)out;
    a.N = N; a.H = H; a.W = W; a.K_pad = K_pad; a.relu = relu;
    a.segs = (W + 15) / 16;
    a.rows_per_task = 50;
    a.tasks_per_strip = (H + a.rows_per_task - 1) / a.rows_per_task;
    const long long nt = (long long)N * a.segs * a.tasks_per_strip;
    if (nt >= (1LL << 31)) return CVPCE_ERR_ARG;
    a.ntasks = (int)nt;
    a.in_bytes = in_up_shift ? (unsigned)((long long)N * (H / 2) * (W / 2) * 128) : (unsigned)((long long)N * H * W * 64);
    a.wgt_bytes = (unsigned)((long long)Cout_pad * K_pad * 2);
    const int want = (a.ntasks + 3) / 4;
    const int cap = 2 * g_cvpce_persistent_wgs;                     // two workgroups (64 KiB of LDS each) per compute unit
    const dim3 grid(want < cap ? want : cap);
    hipStream_t s = (hipStream_t)stream;
    if (in_up_shift) {
        if (!cvpce_smem_attr_done<thin3x3_kernel<E, 2, true>>((const void*)thin3x3_kernel<E, 2, true>, 4 * T3_WAVE)) return CVPCE_ERR_LAUNCH;
        hipLaunchKernelGGL((thin3x3_kernel<E, 2, true>), grid, dim3(256), 4 * T3_WAVE, s, a);
    } else if (Cout == 32) {
        if (!cvpce_smem_attr_done<thin3x3_kernel<E, 2>>((const void*)thin3x3_kernel<E, 2>, 4 * T3_WAVE)) return CVPCE_ERR_LAUNCH;
        hipLaunchKernelGGL((thin3x3_kernel<E, 2>), grid, dim3(256), 4 * T3_WAVE, s, a);
    } else {
        if (!cvpce_smem_attr_done<thin3x3_kernel<E, 1>>((const void*)thin3x3_kernel<E, 1>, 4 * T3_WAVE)) return CVPCE_ERR_LAUNCH;
        hipLaunchKernelGGL((thin3x3_kernel<E, 1>), grid, dim3(256), 4 * T3_WAVE, s, a);
    }
    return cvpce_check_launch();
}

extern "C" int cvpce_conv3x3_thin_bf16(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W, int Cin,
                                       int Cout, int K_pad, int Cout_pad, int relu, int in_up_shift, void* stream) {
    return thin3x3_dispatch<ElemBF16>(in, wgt, bias, out, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, in_up_shift, stream);
}
extern "C" int cvpce_conv3x3_thin_f16(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W, int Cin,
                                      int Cout, int K_pad, int Cout_pad, int relu, int in_up_shift, void* stream) {
    return thin3x3_dispatch<ElemF16>(in, wgt, bias, out, N, H, W, Cin, Cout, K_pad, Cout_pad, relu, in_up_shift, stream);
}
