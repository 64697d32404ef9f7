// Implicit-GEMM convolution for gfx950 (CDNA4): NHWC bf16 activations, bf16
// weights packed [Cout_pad][K_pad] (K order: see include/cvpce_amd.h), fp32
// accumulation on v_mfma_f32_32x32x16_bf16, fused bias / residual(+nearest
// upsample) / ReLU / 2x2-max-pool epilogue, bf16 or fp32 NHWC output.
//
// GEMM view (transposed so that the accumulator's register axis runs along
// Cout -> each lane owns 4 consecutive output channels of ONE pixel and the
// NHWC store is 8 B/lane, 64 B contiguous per pixel):
//     D[cout][pixel] = sum_k Wgt[cout][k] * Im2col[pixel][k]
// The im2col matrix is never materialised: every 16-byte K-chunk of a pixel
// row is gathered straight from the NHWC tensor (one tap, 8 channels) into an
// XOR-swizzled LDS tile (conflict-free ds_read_b128 fragments).
//
// Two kernels share the tiling, the fragment reads and the epilogue:
//   conv_dma_kernel    the workhorse (Cin % 64 == 0, big M): 8 waves, 256x256
//                      (or 4 waves, 128/64 x 256) output tile, K-step 64, operands
//                      staged by LDS-DMA (`global_load_lds_dwordx4`, no VGPR round
//                      trip, no ds_write); the XOR swizzle is applied to the per-lane
//                      SOURCE address (the DMA writes LDS lane-linearly) and padding
//                      halo / ragged rows read a 16-byte zero page.  Double buffered:
//                      the DMA of K-step t+1 is in flight under the MFMAs of step t.
//   conv_igemm_kernel  the generic fallback (any Cin % 8 == 0, small M, thin
//                      channels): 4 waves, 128/64 x 128 tile, register-staged.
//
// fuse_pool2: the M axis enumerates output pixels in 2x2-quad order so the four
// pixels of a pooling window sit in four adjacent lanes; the epilogue takes the
// max with two cross-lane shuffles and stores only the pooled tensor
// (VGG16 conv1_2 / conv2_2 / conv3_3 / conv4_3 + MaxPool2d(2,2)).
//
// Covers every convolution of the hot path.
// Reference semantics: torch.nn.Conv2d as used at
//   /root/reference/cvpce/models/proposals.py:54,68,84 and torchvision 0.9
//   resnet/vgg/fpn/retinanet.
#include "common.h"
#include "../../include/cvpce_amd.h"

// compile-time timing experiments (never set in the shipped library): 1 no DMA in the K loop, 2 no barrier,
// 4 no pixel DMA, 8 no MFMA.  Built by tools/ablate.sh into side libraries selected with CVPCE_LIB.
#ifndef CVPCE_DBG
#define CVPCE_DBG 0
#endif

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef __attribute__((address_space(3))) char lds_char;

__device__ __attribute__((aligned(16))) unsigned int cvpce_zero_page[4] = {0u, 0u, 0u, 0u};

struct ConvArgs {
    const bf16_t* in;
    const bf16_t* wgt;
    const float* bias;
    const bf16_t* res;
    void* out;
    int N, H, W, Cin;
    int Cout, KH, KW, stride, pad, Ho, Wo;
    int K_pad;
    int M;
    int relu, out_f32;
    int in_up_shift;
    int res_mode, Hr, Wr;
    int pool;
    int tiles_p, tiles_c;
    unsigned in_bytes, wgt_bytes;   // buffer-descriptor extents of `in` / `wgt` (LDS-DMA bounds check = zero fill)
    // split-K launches of the register-staged kernel (small maps with a long K: the launch is a chain of K-steps on a few workgroups)
    int ksplit;                     // workgroups per output tile, each with nk / ksplit K-steps
    float* ws;                      // [ksplit][tiles][16 f32x4 per thread][256 threads] fp32 partial tiles, in accumulator-register order
};

// M index -> (image, oy, ox).  Row-major, or 2x2-quad order when the pool is fused.
__device__ __forceinline__ void decode_m(const ConvArgs& a, int m, int& img, int& oy, int& ox) {
    if (a.pool) {
        const int q = m >> 2, sub = m & 3;
        const int qw = a.Wo >> 1, qhw = (a.Ho >> 1) * qw;
        img = q / qhw;
        const int rem = q - img * qhw;
        const int qy = rem / qw, qx = rem - qy * qw;
        oy = 2 * qy + (sub >> 1);
        ox = 2 * qx + (sub & 1);
    } else {
        const int hw = a.Ho * a.Wo;
        img = m / hw;
        const int rem = m - img * hw;
        oy = rem / a.Wo;
        ox = rem - oy * a.Wo;
    }
}

// Epilogue shared by both kernels: lane owns pixel (lane&31) of each 32-wide
// pixel tile and channels 8g + 4h + {0..3} of each 32-channel tile.
template <typename E, int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MT][NT], int c_base, int p_base, int lane) {
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int m = p_base + nt * 32 + lr;
        const bool m_ok = m < a.M;
        size_t res_pix = 0;
        if (a.res_mode == 1) {
            res_pix = (size_t)m;
        } else if (a.res_mode == 2 && m_ok) {
            int img, oy, ox;
            decode_m(a, m, img, oy, ox);
            const int ry = (oy * a.Hr) / a.Ho, rx = (ox * a.Wr) / a.Wo;
            res_pix = (size_t)(img * a.Hr + ry) * a.Wr + rx;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = c_base + mt * 32 + 8 * g + 4 * lh;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][4 * g + j];
                if (a.out_f32) {
                    if (!m_ok || co >= a.Cout) continue;
                    float* o = reinterpret_cast<float*>(a.out) + (size_t)m * a.Cout;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (co + j < a.Cout) {
                            float x = v[j];
                            if (a.bias) x += a.bias[co + j];
                            if (a.res_mode) x += E::widen(a.res[res_pix * a.Cout + co + j]);
                            if (a.relu == 1) x = fmaxf(x, 0.f);
                            else if (a.relu == 2) x = tanhf(x);
                            o[co + j] = x;
                        }
                    }
                } else if (a.pool) {
                    // all 4 lanes of a quad share m_ok / co: shuffles stay convergent
                    if (a.bias && co < a.Cout) {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += b[j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = quad_max(v[j]);
                        if (a.relu == 1) x = fmaxf(x, 0.f);
                        v[j] = x;
                    }
                    if (m_ok && co < a.Cout && (lane & 3) == 0) {
                        bf16x4 o;
#pragma unroll
                        for (int j = 0; j < 1; ++j) o = E::pack4(f32x4{v[0], v[1], v[2], v[3]});
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)(m >> 2) * a.Cout + co) = o;
                    }
                } else {
                    if (!m_ok || co >= a.Cout) continue;
                    // Cout % 4 == 0 is required for bf16 outputs (checked on the host)
                    if (a.bias) {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += b[j];
                    }
                    if (a.res_mode) {
                        const bf16x4 r = *reinterpret_cast<const bf16x4*>(a.res + res_pix * a.Cout + co);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += E::widen(r[j]);
                    }
                    if (a.relu == 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = relu_bits(v[j]);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 1; ++j) o = E::pack4(f32x4{v[0], v[1], v[2], v[3]});
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.Cout + co) = o;
                }
            }
        }
    }
}

// Epilogue for the 16x16x32 accumulator layout: lane owns pixel (lane&15) of each 16-pixel block and channels
// 4*(lane>>4) + {0..3} of each 16-channel block.
template <typename E, int MT, int NT>
__device__ __forceinline__ void conv_epilogue16(const ConvArgs& a, f32x4 (&acc)[MT][NT], int c_base, int p_base, int lane) {
    const int lp = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int m = p_base + nt * 16 + lp;
        const bool m_ok = m < a.M;
        size_t res_pix = 0;
        if (a.res_mode == 1) {
            res_pix = (size_t)m;
        } else if (a.res_mode == 2 && m_ok) {
            int img, oy, ox;
            decode_m(a, m, img, oy, ox);
            const int ry = (oy * a.Hr) / a.Ho, rx = (ox * a.Wr) / a.Wo;
            res_pix = (size_t)(img * a.Hr + ry) * a.Wr + rx;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int co = c_base + mt * 16 + 4 * lq;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][j];
            if (a.out_f32) {
                if (!m_ok || co >= a.Cout) continue;
                float* o = reinterpret_cast<float*>(a.out) + (size_t)m * a.Cout;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (co + j < a.Cout) {
                        float x = v[j];
                        if (a.bias) x += a.bias[co + j];
                        if (a.res_mode) x += E::widen(a.res[res_pix * a.Cout + co + j]);
                        if (a.relu == 1) x = fmaxf(x, 0.f);
                        else if (a.relu == 2) x = tanhf(x);
                        o[co + j] = x;
                    }
                }
            } else if (a.pool) {
                if (a.bias && co < a.Cout) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += b[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = quad_max(v[j]);
                    if (a.relu == 1) x = fmaxf(x, 0.f);
                    v[j] = x;
                }
                if (m_ok && co < a.Cout && (lane & 3) == 0) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 1; ++j) o = E::pack4(f32x4{v[0], v[1], v[2], v[3]});
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)(m >> 2) * a.Cout + co) = o;
                }
            } else {
                if (!m_ok || co >= a.Cout) continue;
                if (a.bias) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += b[j];
                }
                if (a.res_mode) {
                    const bf16x4 r = *reinterpret_cast<const bf16x4*>(a.res + res_pix * a.Cout + co);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += E::widen(r[j]);
                }
                if (a.relu == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = relu_bits(v[j]);
                }
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 1; ++j) o = E::pack4(f32x4{v[0], v[1], v[2], v[3]});
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.Cout + co) = o;
            }
        }
    }
}

// One K-step (BK) of MFMAs from a swizzled [rows][BK] LDS image pair.
template <typename E, int MT, int NT, int BK>
__device__ __forceinline__ void mfma_kstep(const bf16_t* Wb, const bf16_t* Pb, int wrow0, int prow0, int lane,
                                           f32x16 (&acc)[MT][NT]) {
    constexpr int CPR = BK / 8, RPB = 256 / (BK * 2);
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
        const int chunk = kk * 2 + lh;
        bf16x8 af[MT], bfr[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = wrow0 + mt * 32 + lr;
            af[mt] = *reinterpret_cast<const bf16x8*>(Wb + row * BK + ((chunk ^ ((row / RPB) & (CPR - 1))) * 8));
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int row = prow0 + nt * 32 + lr;
            bfr[nt] = *reinterpret_cast<const bf16x8*>(Pb + row * BK + ((chunk ^ ((row / RPB) & (CPR - 1))) * 8));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = E::mfma32(af[mt], bfr[nt], acc[mt][nt]);
    }
}

// ===========================================================================
// LDS-DMA kernel
// ===========================================================================
template <typename E, int TC, int TP, int WC, int WP, int MINW>
__global__ __launch_bounds__(WC * WP * 64, MINW) void conv_dma_kernel(ConvArgs a) {
    constexpr int BK = 64;
    constexpr int NW = WC * WP;
    constexpr int WJ = TC / (8 * NW);      // weight DMA instructions per wave per K-step (8 rows each)
    constexpr int PJ = TP / (8 * NW);      // pixel  DMA instructions per wave per K-step
    constexpr int MT = TC / WC / 32;
    constexpr int NT = TP / WP / 32;
    static_assert(WJ >= 1 && PJ >= 1 && (8 * NW) % 16 == 0, "tile/wave geometry");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ws = reinterpret_cast<bf16_t*>(smem);          // [2][TC][BK]
    bf16_t* Ps = Ws + 2 * TC * BK;                         // [2][TP][BK]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wid / WP, wp = wid % WP;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_c = swz % a.tiles_c, tile_p = swz / a.tiles_c;

    // DMA geometry: instruction j of wave w fills tile rows j*8*NW + w*8 .. +7 (1 KiB, lane-linear:
    // row = lane>>3, physical chunk = lane&7).  The swizzle f(row) = (row>>1)&7 does not depend on j
    // (8*NW is a multiple of 16), so each lane stages ONE logical K-chunk for all its rows.
    const int lrow = wid * 8 + (lane >> 3);
    const int lchunk = (lane & 7) ^ ((lrow >> 1) & 7);

    const int Hl = a.H << a.in_up_shift, Wl = a.W << a.in_up_shift;
    int pbase[PJ], piy[PJ], pix[PJ];
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int m = tile_p * TP + j * 8 * NW + lrow;
        if (m < a.M) {
            int img, oy, ox;
            decode_m(a, m, img, oy, ox);
            piy[j] = oy * a.stride - a.pad;
            pix[j] = ox * a.stride - a.pad;
            pbase[j] = img * a.H * a.W;
        } else {
            piy[j] = -(1 << 28);
            pix[j] = 0;
            pbase[j] = 0;
        }
    }
    int ci = lchunk * 8, kh = 0, kw = 0;      // (channel, tap) of this lane's K-chunk; Cin % 64 == 0 here
    const bf16_t* wsrc = a.wgt + (size_t)(tile_c * TC + lrow) * a.K_pad + lchunk * 8;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(cvpce_zero_page);

    // A K-step's DMA is split in two: CVPCE_DMA_ADDR computes the per-lane source pointers of the NEXT
    // K-step (VALU only), CVPCE_DMA_ISSUE(i) fires piece i.  The pieces are spread between the four
    // MFMA groups of the CURRENT K-step, so DMA issue slots hide under matrix-pipe time instead of
    // forming an MFMA-free bubble at the head of every K-step (both waves of a SIMD run in lockstep
    // after the barrier, so the partner wave cannot fill that bubble).
    constexpr int NP = WJ + PJ;              // DMA pieces per wave per K-step
    constexpr int NPK = (NP + 3) / 4;        // pieces issued per MFMA group
    const bf16_t* dsrc[NP];
#define CVPCE_DMA_ADDR(KT)                                                                                     \
    {                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < WJ; ++j)                                                         \
            dsrc[j] = wsrc + (size_t)j * 8 * NW * a.K_pad + (KT) * BK;                                         \
        const bool tap_ok = ci < a.Cin;                                                                        \
        _Pragma("unroll") for (int j = 0; j < PJ; ++j) {                                                       \
            const int iy = piy[j] + kh, ix = pix[j] + kw;                                                      \
            const bool ok = tap_ok && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;              \
            const size_t off =                                                                                 \
                (size_t)(pbase[j] + (iy >> a.in_up_shift) * a.W + (ix >> a.in_up_shift)) * a.Cin + ci;         \
            dsrc[WJ + j] = ok ? a.in + off : zero;                                                             \
        }                                                                                                      \
        /* K order for Cin % 64 == 0: (64-channel chunk, kh, kw, channel) -- the 9 taps of one chunk are */    \
        /* consecutive K-steps, so their overlapping pixel reads hit the XCD's L2 instead of the fabric   */    \
        if (++kw == a.KW) {                                                                                    \
            kw = 0;                                                                                            \
            if (++kh == a.KH) { kh = 0; ci += BK; }                                                            \
        }                                                                                                      \
    }
#define CVPCE_DMA_ISSUE(I, BUF)                                                                                \
    {                                                                                                          \
        if ((I) < WJ)                                                                                          \
            __builtin_amdgcn_global_load_lds((gbl_void*)dsrc[(I)],                                             \
                (lds_void*)(Ws + (BUF) * TC * BK + ((I) * 8 * NW + wid * 8) * BK), 16, 0, 0);                  \
        else                                                                                                   \
            __builtin_amdgcn_global_load_lds((gbl_void*)dsrc[(I)],                                             \
                (lds_void*)(Ps + (BUF) * TP * BK + (((I) - WJ) * 8 * NW + wid * 8) * BK), 16, 0, 0);           \
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = a.K_pad / BK;
    CVPCE_DMA_ADDR(0)
#pragma unroll
    for (int i = 0; i < NP; ++i) CVPCE_DMA_ISSUE(i, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) CVPCE_DMA_ADDR(kt + 1)
        const bf16_t* Wb = Ws + cur * TC * BK;
        const bf16_t* Pb = Ps + cur * TP * BK;
        // fragments are double buffered in registers: the ds_reads of group kk+1 are in flight under the
        // MFMAs of group kk, so only the first group of a K-step exposes LDS latency
        bf16x8 af[2][MT], bfr[2][NT];
#define CVPCE_LOAD_FRAGS(KK, SLOT)                                                                             \
        {                                                                                                      \
            const int chunk = (KK) * 2 + lh;                                                                   \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                \
                const int row = wc * (TC / WC) + mt * 32 + lr;                                                 \
                af[SLOT][mt] = *reinterpret_cast<const bf16x8*>(Wb + row * BK + ((chunk ^ ((row >> 1) & 7)) * 8)); \
            }                                                                                                  \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                \
                const int row = wp * (TP / WP) + nt * 32 + lr;                                                 \
                bfr[SLOT][nt] = *reinterpret_cast<const bf16x8*>(Pb + row * BK + ((chunk ^ ((row >> 1) & 7)) * 8)); \
            }                                                                                                  \
        }
        CVPCE_LOAD_FRAGS(0, 0)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) CVPCE_LOAD_FRAGS(kk + 1, (kk + 1) & 1)
            if (more) {
#pragma unroll
                for (int i = kk * NPK; i < (kk + 1) * NPK && i < NP; ++i) CVPCE_DMA_ISSUE(i, cur ^ 1)
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ds_reads / DMA issue AHEAD of this MFMA group
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = E::mfma32(af[kk & 1][mt], bfr[kk & 1][nt], acc[mt][nt]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef CVPCE_LOAD_FRAGS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
#undef CVPCE_DMA_ADDR
#undef CVPCE_DMA_ISSUE
    conv_epilogue<E, MT, NT>(a, acc, tile_c * TC + wc * (TC / WC), tile_p * TP + wp * (TP / WP), lane);
}

// ===========================================================================
// LDS-DMA kernel, 4-stage ring: K-step 32, three K-steps of DMA in flight
// ===========================================================================
// Issued -> landed latency of an LDS-DMA piece under load is ~1 us (about one 64-deep K-step of
// MFMA time), so a 2-buffer scheme exposes it every step.  Here the ring holds four 32-deep
// K-steps (4 x 32 KiB at 256x256): stage t+3 is issued while stage t computes, the wave waits with
// a COUNTED vmcnt (never 0 in steady state) and a raw s_barrier (a __syncthreads() would drain
// vmcnt to 0).  K order: (64-channel chunk, kh, kw, 32-channel half, channel).
template <typename E, int TC, int TP, int WC, int WP, int MINW, int NS>
__global__ __launch_bounds__(WC * WP * 64, MINW) void conv_dma4_kernel(ConvArgs a) {
    constexpr int BK = 32;
    static_assert(NS == 3 || NS == 4, "ring depth");
    constexpr int NW = WC * WP;
    constexpr int WJ = TC / (16 * NW);     // weight DMA pieces per wave per K-step (16 rows of 64 B each)
    constexpr int PJ = TP / (16 * NW);
    constexpr int NP = WJ + PJ;
    constexpr int MT = TC / WC / 32;
    constexpr int NT = TP / WP / 32;
    static_assert(WJ >= 1 && PJ >= 1, "tile/wave geometry");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ws = reinterpret_cast<bf16_t*>(smem);          // [NS][TC][BK]
    bf16_t* Ps = Ws + NS * TC * BK;                        // [NS][TP][BK]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wid / WP, wp = wid % WP;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_c = swz % a.tiles_c, tile_p = swz / a.tiles_c;

    // piece j of wave w fills rows j*16*NW + w*16 .. +15 (row = lane>>2, physical chunk = lane&3);
    // swizzle f(row) = (row>>2)&3 = (lane>>4)&3 is the same for every piece of a lane
    const int lrow = wid * 16 + (lane >> 2);
    const int lchunk = (lane & 3) ^ ((lane >> 4) & 3);

    // Per-lane, K-invariant part of every pixel-piece address, computed ONCE: byte offset of (image, oy*stride,
    // ox*stride, lane's 8-channel slice) and a bit mask of the taps that fall inside the image.  The tap / channel
    // walk of the K loop is wave-uniform (SALU); each DMA piece then costs one v_add + one v_cndmask of address math.
    // (Ablation, VGG conv4_2: with the per-piece index recomputation the DMA stage cost 0.31 ms of a 1.37 ms launch.)
    unsigned poff[PJ], pmask[PJ];
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int m = tile_p * TP + j * 16 * NW + lrow;
        poff[j] = 0u;
        pmask[j] = 0u;
        if (m < a.M) {
            int img, oy, ox;
            decode_m(a, m, img, oy, ox);
            const int y0 = oy * a.stride, x0 = ox * a.stride;
            poff[j] = (unsigned)((((size_t)(img * a.H + y0) * a.W + x0) * a.Cin + lchunk * 8) * 2);
            for (int t = 0; t < a.KH * a.KW; ++t) {
                const int iy = y0 - a.pad + t / a.KW, ix = x0 - a.pad + t % a.KW;
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) pmask[j] |= 1u << t;
            }
        }
    }
    // wave-uniform K walk: (64-channel chunk, kh, kw, 32-channel half)
    int kh = 0, kw = 0, half = 0, cbase = 0, tapi = 0;
    int tap_off = ((0 - a.pad) * a.W + (0 - a.pad)) * a.Cin * 2;     // byte shift of tap (kh,kw), may be negative

    // LDS-DMA through BUFFER instructions (`buffer_load_dwordx4 ... offen lds`), not `global_load_lds`: the latter is
    // FLAT-encoded, and hipcc then treats every later LDS wait as lgkmcnt(0) (flat ops may return out of order), which
    // serialises the fragment prefetch.  A buffer descriptor also gives the zero fill for free: an out-of-range
    // offset (padding halo, ragged rows, K padding) returns 0 -- no zero page, no select on a 64-bit pointer.
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_p = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    unsigned woff[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j)
        woff[j] = (unsigned)(((size_t)(tile_c * TC + lrow + j * 16 * NW) * a.K_pad + lchunk * 8) * 2);
#define CVPCE_DMA4_STAGE(KT, BUF)                                                                              \
    {                                                                                                          \
        const int kbyte = (KT) * (BK * 2);                                                                     \
        _Pragma("unroll") for (int j = 0; j < WJ; ++j)                                                         \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w,                                                    \
                (lds_void*)(Ws + (BUF) * TC * BK + (j * 16 * NW + wid * 16) * BK), 16, (int)woff[j], kbyte, 0, 0); \
        const int uni = tap_off + (cbase + 32 * half) * 2;                                                     \
        const bool ch_ok = cbase < a.Cin;                                                                      \
        _Pragma("unroll") for (int j = 0; j < PJ; ++j) {                                                       \
            const bool ok = ch_ok && ((pmask[j] >> tapi) & 1u);                                                \
            if (!(CVPCE_DBG & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_p,                              \
                (lds_void*)(Ps + (BUF) * TP * BK + (j * 16 * NW + wid * 16) * BK), 16,                         \
                (int)(ok ? poff[j] + (unsigned)uni : 0xFFFFFFF0u), 0, 0, 0);                                   \
        }                                                                                                      \
        half ^= 1;                                                                                             \
        if (half == 0) {                                                                                       \
            ++tapi;                                                                                            \
            tap_off += a.Cin * 2;                                                                              \
            if (++kw == a.KW) {                                                                                \
                kw = 0;                                                                                        \
                tap_off += (a.W - a.KW) * a.Cin * 2;                                                           \
                if (++kh == a.KH) {                                                                            \
                    kh = 0; tapi = 0; cbase += 64;                                                             \
                    tap_off = ((0 - a.pad) * a.W + (0 - a.pad)) * a.Cin * 2;                                   \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = a.K_pad / BK;
    // Software pipeline over 16-deep fragment groups (two per stage).  The ds_reads of group g+1 are issued
    // BEFORE the MFMAs of group g and the stage hand-off (counted vmcnt + barrier + next DMA issue) sits between the
    // two MFMA groups of a stage, so the matrix pipe keeps draining 8 queued MFMAs while the wave waits, syncs and
    // issues DMA.  (Ablation on VGG conv4_2: fragment reads + barrier with neither DMA nor MFMA took 0.72 ms of a
    // 1.43 ms launch when they were serialised with the MFMAs.)
    //   stage s lives in ring slot s & 3; when MFMA(kt, group 1) is issued, stage kt+1 must be visible (its group-0
    //   fragments are being fetched) and stages kt+2, kt+3 are in flight.
    CVPCE_DMA4_STAGE(0, 0)
    if (nk > 1) CVPCE_DMA4_STAGE(1, 1)
    if (NS == 4 && nk > 2) CVPCE_DMA4_STAGE(2, 2)
    const int lr = lane & 31, lh = lane >> 5;
    int wrow[MT], prow[NT];       // byte-free row bases and swizzle terms of this lane's fragments
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wrow[mt] = wc * (TC / WC) + mt * 32 + lr;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) prow[nt] = wp * (TP / WP) + nt * 32 + lr;
    bf16x8 af[2][MT], bfr[2][NT];
    // Fragment reads are hand-issued (`ds_read_b128` in inline asm) and hand-counted: hipcc's waitcnt pass falls back to
    // lgkmcnt(0) for LDS reads that are pending across the loop back-edge or sit next to LDS-DMA, which would make
    // every MFMA group wait for the group that was only just prefetched.  Inline asm hides the reads from that pass;
    // `s_waitcnt lgkmcnt(N)` + sched_barrier below are the only LDS waits in the loop (cdna_hip_programming.md 5.4 r18).
    const unsigned lds_w0 = (unsigned)(size_t)(lds_char*)Ws, lds_p0 = (unsigned)(size_t)(lds_char*)Ps;
    unsigned woffb[MT], poffb[NT], wsw[MT], psw[NT];      // row byte offsets and swizzle terms of this lane's fragments
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { woffb[mt] = wrow[mt] * (BK * 2); wsw[mt] = (wrow[mt] >> 2) & 3; }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { poffb[nt] = prow[nt] * (BK * 2); psw[nt] = (prow[nt] >> 2) & 3; }
#define CVPCE_FRAGS(SLOT, STAGE, KK)                                                                           \
    {                                                                                                          \
        const unsigned wb_ = lds_w0 + ((STAGE) % NS) * (TC * BK * 2), pb_ = lds_p0 + ((STAGE) % NS) * (TP * BK * 2); \
        const unsigned chunk = (KK) * 2 + lh;                                                                  \
        if (!(CVPCE_DBG & 16) || (STAGE) == 0) {                                                               \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                      \
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[SLOT][mt]) : "v"(wb_ + woffb[mt] + ((chunk ^ wsw[mt]) << 4))); \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                      \
            asm volatile("ds_read_b128 %0, %1" : "=v"(bfr[SLOT][nt]) : "v"(pb_ + poffb[nt] + ((chunk ^ psw[nt]) << 4))); \
        }                                                                                                      \
    }
#define CVPCE_MFMAS(SLOT)                                                                                      \
    {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        if ((CVPCE_DBG & 32)) {                                                                                \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                      \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                \
                f32x4* q4 = reinterpret_cast<f32x4*>(&acc[mt][nt]);                                            \
                q4[2 * (SLOT)] = E::mfma16(af[SLOT][mt], bfr[SLOT][nt], q4[2 * (SLOT)]); \
                q4[2 * (SLOT) + 1] = E::mfma16(af[SLOT][mt], bfr[SLOT][nt], q4[2 * (SLOT) + 1]); \
            }                                                                                                  \
        } else if (!(CVPCE_DBG & 8)) {                                                                         \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                      \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                  \
                acc[mt][nt] = E::mfma32(af[SLOT][mt], bfr[SLOT][nt], acc[mt][nt]); \
        }                                                                                                      \
        __builtin_amdgcn_s_setprio(0);                                                                         \
    }
    // stage 0 becomes visible; fetch its first fragment group
    if (NS == 4 && nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    CVPCE_FRAGS(0, 0, 0)
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // group 0 (issued one MFMA group ago) has landed
        __builtin_amdgcn_sched_barrier(0);
        CVPCE_FRAGS(1, kt, 1)                       // group 1 of this stage, in flight under MFMA group 0
        __builtin_amdgcn_sched_barrier(0);
        CVPCE_MFMAS(0)
        __builtin_amdgcn_sched_barrier(0);
        // hand-off: stage kt+1 must have landed for every wave; slot (kt+NS-1)%NS == (kt-1)%NS is free for the next DMA.
        // Stages kt+2 .. kt+NS-2 (NS-3 of them) may stay in flight across the wait.
        if (kt + 1 < nk) {
            const int ahead = nk - 2 - kt;
            if (NS == 4 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(CVPCE_DBG & 2)) __builtin_amdgcn_s_barrier();
            if (kt + NS - 1 < nk && !(CVPCE_DBG & 1)) CVPCE_DMA4_STAGE(kt + NS - 1, (kt + NS - 1) % NS)
            CVPCE_FRAGS(0, kt + 1, 0)               // group 0 of the next stage, in flight under MFMA group 1
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory");   // group 1 landed; the MT+NT newer reads stay in flight
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        CVPCE_MFMAS(1)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef CVPCE_FRAGS
#undef CVPCE_MFMAS
#undef CVPCE_DMA4_STAGE
    conv_epilogue<E, MT, NT>(a, acc, tile_c * TC + wc * (TC / WC), tile_p * TP + wp * (TP / WP), lane);
}

// ===========================================================================
// LDS-DMA kernel, 4-stage ring, v_mfma_f32_16x16x32_bf16 variant
// ===========================================================================
// Same staging / ring / hand-off as conv_dma4_kernel, but the products run on the 16x16x32 MFMA shape: at equal
// cycles per FLOP the chip holds a higher clock on it under load (measured here: MFMA-only stream of VGG conv4_2
// 0.99 -> 0.89 ms, whole kernel +7 %; cf. MI355X_MICROARCH.md DVFS item 7).  A lane's fragment is row (lane&15),
// 16-byte K-chunk (lane>>4) of a 16-row block, so one ds_read_b128 covers a whole 16x32 operand block; the XOR term
// T[(row>>2)&3], T = {0,2,3,1}, keeps those reads conflict-free.  Accumulators: col = lane&15 (pixel),
// row = 4*(lane>>4)+reg (cout).
// Issued -> landed latency of an LDS-DMA piece under load is ~1 us (about one 64-deep K-step of
// MFMA time), so a 2-buffer scheme exposes it every step.  Here the ring holds four 32-deep
// K-steps (4 x 32 KiB at 256x256): stage t+3 is issued while stage t computes, the wave waits with
// a COUNTED vmcnt (never 0 in steady state) and a raw s_barrier (a __syncthreads() would drain
// vmcnt to 0).  K order: (64-channel chunk, kh, kw, 32-channel half, channel).
template <typename E, int TC, int TP, int WC, int WP, int MINW, int NS>
__global__ __launch_bounds__(WC * WP * 64, MINW) void conv_dma16_kernel(ConvArgs a) {
    constexpr int BK = 32;
    static_assert(NS == 3 || NS == 4, "ring depth");
    constexpr int NW = WC * WP;
    constexpr int WJ = TC / (16 * NW);     // weight DMA pieces per wave per K-step (16 rows of 64 B each)
    constexpr int PJ = TP / (16 * NW);
    constexpr int NP = WJ + PJ;
    constexpr int MT = TC / WC / 16;      // 16-row cout blocks per wave
    constexpr int NT = TP / WP / 16;      // 16-pixel blocks per wave
    static_assert(MT % 2 == 0, "cout blocks split in two pipeline groups");
    static_assert(WJ >= 1 && PJ >= 1, "tile/wave geometry");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ws = reinterpret_cast<bf16_t*>(smem);          // [NS][TC][BK]
    bf16_t* Ps = Ws + NS * TC * BK;                        // [NS][TP][BK]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wid / WP, wp = wid % WP;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_c = swz % a.tiles_c, tile_p = swz / a.tiles_c;

    // piece j of wave w fills rows j*16*NW + w*16 .. +15 (row = lane>>2, physical chunk = lane&3);
    // swizzle T[(row>>2)&3] = T[(lane>>4)&3] is the same for every piece of a lane
    const int lrow = wid * 16 + (lane >> 2);
    const int lchunk = (lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3);   // T = {0,2,3,1} packed in 0x78

    // Per-lane, K-invariant part of every pixel-piece address, computed ONCE: byte offset of (image, oy*stride,
    // ox*stride, lane's 8-channel slice) and a bit mask of the taps that fall inside the image.  The tap / channel
    // walk of the K loop is wave-uniform (SALU); each DMA piece then costs one v_add + one v_cndmask of address math.
    // (Ablation, VGG conv4_2: with the per-piece index recomputation the DMA stage cost 0.31 ms of a 1.37 ms launch.)
    unsigned poff[PJ], pmask[PJ];
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int m = tile_p * TP + j * 16 * NW + lrow;
        poff[j] = 0u;
        pmask[j] = 0u;
        if (m < a.M) {
            int img, oy, ox;
            decode_m(a, m, img, oy, ox);
            const int y0 = oy * a.stride, x0 = ox * a.stride;
            poff[j] = (unsigned)((((size_t)(img * a.H + y0) * a.W + x0) * a.Cin + lchunk * 8) * 2);
            for (int t = 0; t < a.KH * a.KW; ++t) {
                const int iy = y0 - a.pad + t / a.KW, ix = x0 - a.pad + t % a.KW;
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) pmask[j] |= 1u << t;
            }
        }
    }
    // wave-uniform K walk: (64-channel chunk, kh, kw, 32-channel half)
    int kh = 0, kw = 0, half = 0, cbase = 0, tapi = 0;
    int tap_off = ((0 - a.pad) * a.W + (0 - a.pad)) * a.Cin * 2;     // byte shift of tap (kh,kw), may be negative

    // LDS-DMA through BUFFER instructions (`buffer_load_dwordx4 ... offen lds`), not `global_load_lds`: the latter is
    // FLAT-encoded, and hipcc then treats every later LDS wait as lgkmcnt(0) (flat ops may return out of order), which
    // serialises the fragment prefetch.  A buffer descriptor also gives the zero fill for free: an out-of-range
    // offset (padding halo, ragged rows, K padding) returns 0 -- no zero page, no select on a 64-bit pointer.
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_p = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    unsigned woff[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j)
        woff[j] = (unsigned)(((size_t)(tile_c * TC + lrow + j * 16 * NW) * a.K_pad + lchunk * 8) * 2);
#define CVPCE_DMA4_STAGE(KT, BUF)                                                                              \
    {                                                                                                          \
        const int kbyte = (KT) * (BK * 2);                                                                     \
        _Pragma("unroll") for (int j = 0; j < WJ; ++j)                                                         \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w,                                                    \
                (lds_void*)(Ws + (BUF) * TC * BK + (j * 16 * NW + wid * 16) * BK), 16, (int)woff[j], kbyte, 0, 0); \
        const int uni = tap_off + (cbase + 32 * half) * 2;                                                     \
        const bool ch_ok = cbase < a.Cin;                                                                      \
        _Pragma("unroll") for (int j = 0; j < PJ; ++j) {                                                       \
            const bool ok = ch_ok && ((pmask[j] >> tapi) & 1u);                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_p,                              \
                (lds_void*)(Ps + (BUF) * TP * BK + (j * 16 * NW + wid * 16) * BK), 16,                         \
                (int)(ok ? poff[j] + (unsigned)uni : 0xFFFFFFF0u), 0, 0, 0);                                   \
        }                                                                                                      \
        half ^= 1;                                                                                             \
        if (half == 0) {                                                                                       \
            ++tapi;                                                                                            \
            tap_off += a.Cin * 2;                                                                              \
            if (++kw == a.KW) {                                                                                \
                kw = 0;                                                                                        \
                tap_off += (a.W - a.KW) * a.Cin * 2;                                                           \
                if (++kh == a.KH) {                                                                            \
                    kh = 0; tapi = 0; cbase += 64;                                                             \
                    tap_off = ((0 - a.pad) * a.W + (0 - a.pad)) * a.Cin * 2;                                   \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    const int nk = a.K_pad / BK;
    // Software pipeline over 16-deep fragment groups (two per stage).  The ds_reads of group g+1 are issued
    // BEFORE the MFMAs of group g and the stage hand-off (counted vmcnt + barrier + next DMA issue) sits between the
    // two MFMA groups of a stage, so the matrix pipe keeps draining 8 queued MFMAs while the wave waits, syncs and
    // issues DMA.  (Ablation on VGG conv4_2: fragment reads + barrier with neither DMA nor MFMA took 0.72 ms of a
    // 1.43 ms launch when they were serialised with the MFMAs.)
    //   stage s lives in ring slot s & 3; when MFMA(kt, group 1) is issued, stage kt+1 must be visible (its group-0
    //   fragments are being fetched) and stages kt+2, kt+3 are in flight.
    CVPCE_DMA4_STAGE(0, 0)
    if (nk > 1) CVPCE_DMA4_STAGE(1, 1)
    if (NS == 4 && nk > 2) CVPCE_DMA4_STAGE(2, 2)
    // Fragment pipeline.  Group 0 = cout blocks 0..MT/2-1 + ALL pixel blocks (MT/2 + NT reads), group 1 = cout blocks
    // MT/2..MT-1 (MT/2 reads).  Pixel fragments are double buffered across stages (group 1 of stage t still multiplies
    // stage t's pixels while group 0 of stage t+1 is being fetched).
    constexpr int MH = MT / 2;
    const unsigned lds_w0 = (unsigned)(size_t)(lds_char*)Ws, lds_p0 = (unsigned)(size_t)(lds_char*)Ps;
    const unsigned lrow16 = lane & 15;
    const unsigned lane_off = lrow16 * (BK * 2) + (((lane >> 4) ^ ((0x78 >> (2 * (lrow16 >> 2))) & 3)) << 4);
    const unsigned wlane = lds_w0 + wc * (TC / WC) * (BK * 2) + lane_off;
    const unsigned plane = lds_p0 + wp * (TP / WP) * (BK * 2) + lane_off;
    bf16x8 af[MT], bfr[2][NT];
#define CVPCE_READ_A(STAGE, M0)                                                                                \
    {                                                                                                          \
        const unsigned wb_ = wlane + ((STAGE) % NS) * (TC * BK * 2);                                           \
        _Pragma("unroll") for (int i = 0; i < MH; ++i)                                                         \
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[(M0) + i]) : "v"(wb_ + ((M0) + i) * 16 * (BK * 2)));  \
    }
#define CVPCE_READ_B(STAGE, SET)                                                                               \
    {                                                                                                          \
        const unsigned pb_ = plane + ((STAGE) % NS) * (TP * BK * 2);                                           \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                      \
            asm volatile("ds_read_b128 %0, %1" : "=v"(bfr[SET][nt]) : "v"(pb_ + nt * 16 * (BK * 2)));          \
    }
#define CVPCE_MFMAS(M0, SET)                                                                                   \
    {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i = 0; i < MH; ++i)                                                         \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                  \
                acc[(M0) + i][nt] = E::mfma16(af[(M0) + i], bfr[SET][nt], acc[(M0) + i][nt]); \
        __builtin_amdgcn_s_setprio(0);                                                                         \
    }
    // stage 0 becomes visible; fetch its group 0
    if (NS == 4 && nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    CVPCE_READ_A(0, 0)
    CVPCE_READ_B(0, 0)
    // the loop body is written for even/odd stages explicitly so that the pixel-fragment set index is a constant
#define CVPCE_STAGE_BODY(KT, SET)                                                                              \
    {                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* group 0 of stage KT has landed */           \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        CVPCE_READ_A(KT, MH)                                    /* group 1, in flight under MFMA group 0 */    \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        CVPCE_MFMAS(0, SET)                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if ((KT) + 1 < nk) {                                                                                   \
            const int ahead = nk - 2 - (KT);                                                                   \
            if (NS == 4 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");               \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                              \
            __builtin_amdgcn_s_barrier();                                                                      \
            if ((KT) + NS - 1 < nk) CVPCE_DMA4_STAGE((KT) + NS - 1, ((KT) + NS - 1) % NS)                      \
            CVPCE_READ_A((KT) + 1, 0)                           /* group 0 of the next stage */                \
            CVPCE_READ_B((KT) + 1, (SET) ^ 1)                                                                  \
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MH + NT) : "memory");   /* group 1 landed */            \
        } else {                                                                                               \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        CVPCE_MFMAS(MH, SET)                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        CVPCE_STAGE_BODY(kt, 0)
        CVPCE_STAGE_BODY(kt + 1, 1)
    }
    if (kt < nk) CVPCE_STAGE_BODY(kt, 0)
#undef CVPCE_STAGE_BODY
#undef CVPCE_READ_A
#undef CVPCE_READ_B
#undef CVPCE_MFMAS
#undef CVPCE_DMA4_STAGE
    conv_epilogue16<E, MT, NT>(a, acc, tile_c * TC + wc * (TC / WC), tile_p * TP + wp * (TP / WP), lane);
}

// ===========================================================================
// generic register-staged kernel
// ===========================================================================
// SPLIT: a.ksplit workgroups share one output tile, each over its own range of K-steps; every one of them writes its fp32 partial tile
// to a.ws (in accumulator-register order: 1 KiB per wave and store) and conv_splitk_finish_kernel, the next launch on the stream, adds
// the partials in split order 0, 1, ... and runs the epilogue.  (One launch with the tile's last-arriving workgroup finishing it was
// built first: the agent-scope fences it needs around the hand-over -- buffer_wbl2 / buffer_inv of the whole L2, per workgroup -- made
// the detector 0.24 ms SLOWER per 4 images, profiles/r05_rejected_experiments.md.)
template <typename E, int TC, int TP, int BK, int WC, int WP, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
    static_assert(!SPLIT || BK == 64, "split launches use the chunk-major K order");
    constexpr int CPR = BK / 8;             // 16-byte chunks per tile row
    constexpr int RPP = 256 / CPR;          // tile rows covered per staging pass
    constexpr int WPASS = TC / RPP;
    constexpr int PPASS = TP / RPP;
    constexpr int MT = TC / WC / 32;
    constexpr int NT = TP / WP / 32;
    constexpr int RPB = 256 / (BK * 2);     // tile rows per 256-byte LDS bank row
    static_assert(WC * WP == 4, "4 waves");
    static_assert(WPASS >= 1 && PPASS >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ws = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Ps = Ws + 2 * TC * BK;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wc = wid / WP, wp = wid % WP;
    int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = a.tiles_c * a.tiles_p;
    int split = 0;
    if constexpr (SPLIT) {
        split = swz / ntile;
        swz -= split * ntile;
    }
    const int tile_c = swz % a.tiles_c, tile_p = swz / a.tiles_c;

    const int c = tid % CPR;
    const int r0 = tid / CPR;

    const int Hl = a.H << a.in_up_shift, Wl = a.W << a.in_up_shift;

    int pbase[PPASS], piy[PPASS], pix[PPASS];
#pragma unroll
    for (int i = 0; i < PPASS; ++i) {
        const int m = tile_p * TP + r0 + i * RPP;
        if (m < a.M) {
            int img, oy, ox;
            decode_m(a, m, img, oy, ox);
            piy[i] = oy * a.stride - a.pad;
            pix[i] = ox * a.stride - a.pad;
            pbase[i] = img * a.H * a.W;
        } else {
            piy[i] = -(1 << 28);
            pix[i] = 0;
            pbase[i] = 0;
        }
    }
    // K-chunk state of this thread: (kh, kw, ci) of element k = kt*BK + c*8
    int tap = (c * 8) / a.Cin;
    int ci = c * 8 - tap * a.Cin;
    int kh = tap / a.KW;
    int kw = tap - kh * a.KW;

    const int nk = a.K_pad / BK;
    int k_begin = 0, k_end = nk;
    if constexpr (SPLIT) {
        // this workgroup's K-steps; K-step kt of the chunk-major order is (channel chunk kt / taps, tap kt % taps)
        const int per = (nk + a.ksplit - 1) / a.ksplit;
        k_begin = split * per;
        k_end = k_begin + per < nk ? k_begin + per : nk;
        const int taps = a.KH * a.KW;
        const int chunk = k_begin / taps, r = k_begin - chunk * taps;
        kh = r / a.KW;
        kw = r - kh * a.KW;
        ci = chunk * BK + c * 8;
    }

    const bf16_t* wrow = a.wgt + (size_t)(tile_c * TC + r0) * a.K_pad + c * 8;

    // Register staging, TWO K-steps deep, and no branch anywhere around a load: the gather goes through a buffer descriptor
    // over the input tensor, a tap outside the image (zero padding), a ragged row or a K-step past the end gets an offset
    // outside the descriptor's range and reads as zeros.  Behind `if (ok)` branches the compiler cannot count the loads
    // in flight and waits with vmcnt(0) before the next ones are issued: every K-step then costs a full L2 round trip
    // (the same repair as in csrc/match.hip; these small-M layers are pure latency).
    const __amdgpu_buffer_rsrc_t srd_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    u32x4 wreg[2][WPASS], preg[2][PPASS];

#define CVPCE_LOAD_TILE(KT, SLOT)                                                                              \
    {                                                                                                          \
        const int ktw_ = (KT) < nk ? (KT) : nk - 1;      /* past the end: re-read the last weight K-step */      \
        _Pragma("unroll") for (int i = 0; i < WPASS; ++i) wreg[SLOT][i] =                                      \
            *reinterpret_cast<const u32x4*>(wrow + (size_t)i * RPP * a.K_pad + ktw_ * BK);                     \
        const bool tap_ok = (BK == 64) ? (ci < a.Cin) : (kh < a.KH);                                           \
        _Pragma("unroll") for (int i = 0; i < PPASS; ++i) {                                                    \
            const int iy = piy[i] + kh, ix = pix[i] + kw;                                                      \
            const bool ok = tap_ok && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;              \
            const unsigned off =                                                                               \
                ((unsigned)(pbase[i] + (iy >> a.in_up_shift) * a.W + (ix >> a.in_up_shift)) * (unsigned)a.Cin + (unsigned)ci) * 2u; \
            preg[SLOT][i] = __builtin_amdgcn_raw_buffer_load_b128(srd_in, ok ? off : 0xFFFFFFF0u, 0, 0);      \
        }                                                                                                      \
        if (BK == 64) { /* chunk-major K order (Cin % 64 == 0), see conv_dma_kernel */                         \
            if (++kw == a.KW) {                                                                                \
                kw = 0;                                                                                        \
                if (++kh == a.KH) { kh = 0; ci += BK; }                                                        \
            }                                                                                                  \
        } else {        /* tap-major K order: (kh, kw, channel) */                                             \
            ci += BK;                                                                                          \
            while (ci >= a.Cin) {                                                                              \
                ci -= a.Cin;                                                                                   \
                if (++kw == a.KW) { kw = 0; ++kh; }                                                            \
            }                                                                                                  \
        }                                                                                                      \
    }
#define CVPCE_STORE_TILE(BUF, SLOT)                                                                            \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < WPASS; ++i) {                                                    \
            const int row = r0 + i * RPP;                                                                      \
            const int phys = c ^ ((row / RPB) & (CPR - 1));                                                    \
            *reinterpret_cast<u32x4*>(Ws + (BUF) * TC * BK + row * BK + phys * 8) = wreg[SLOT][i];             \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < PPASS; ++i) {                                                    \
            const int row = r0 + i * RPP;                                                                      \
            const int phys = c ^ ((row / RPB) & (CPR - 1));                                                    \
            *reinterpret_cast<u32x4*>(Ps + (BUF) * TP * BK + row * BK + phys * 8) = preg[SLOT][i];             \
        }                                                                                                      \
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    CVPCE_LOAD_TILE(k_begin, 0)
    CVPCE_LOAD_TILE(k_begin + 1, 1)
    CVPCE_STORE_TILE(0, 0)
    __syncthreads();
    int cur = 0;
    // One straight-line step: issue the loads of step kt + 2 into the slot step kt's data just left, run the MFMAs of step kt
    // from LDS, write step kt + 1 (loaded one step ago) to the other LDS buffer.  Unrolled by two so that the register
    // slots are compile-time indices; past the last K-step the loads fetch zeros / the last weight step into buffers
    // nobody reads.
    // (timing-only ablations of the register-staged kernel, tools/ablate.sh conv_igemm: CVPCE_DBG 1024 no global loads in the loop,
    //  2048 no MFMAs, 4096 no barrier, 8192 no LDS stores)
#define CVPCE_KSTEP(KT, SLOT)                                                                                  \
    {                                                                                                          \
        if constexpr (!(CVPCE_DBG & 1024)) { CVPCE_LOAD_TILE((KT) + 2, (SLOT) ^ 1) }                           \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if constexpr (!(CVPCE_DBG & 2048))                                                                     \
            mfma_kstep<E, MT, NT, BK>(Ws + cur * TC * BK, Ps + cur * TP * BK, wc * (TC / WC), wp * (TP / WP), lane, acc); \
        if constexpr (!(CVPCE_DBG & 8192)) { CVPCE_STORE_TILE(cur ^ 1, SLOT) }                                 \
        if constexpr (!(CVPCE_DBG & 4096)) __syncthreads();                                                    \
        cur ^= 1;                                                                                              \
    }
    int kt = k_begin;
    for (; kt + 1 < k_end; kt += 2) {
        CVPCE_KSTEP(kt, 1)
        CVPCE_KSTEP(kt + 1, 0)
    }
    if (kt < k_end) CVPCE_KSTEP(kt, 1)
#undef CVPCE_KSTEP
#undef CVPCE_LOAD_TILE
#undef CVPCE_STORE_TILE
    if constexpr (SPLIT) {
        // (a split whose K range is empty -- ksplit does not divide nk -- still hands in its zeros and takes a ticket)
        constexpr int Q = MT * NT * 4;                                   // f32x4 per thread
        f32x4* part = reinterpret_cast<f32x4*>(a.ws) + ((size_t)(split * ntile + swz) * Q) * 256 + tid;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    part[((mt * NT + nt) * 4 + g) * 256] =
                        f32x4{acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
        return;                                                          // conv_splitk_finish_kernel adds the partial tiles and stores
    }
    conv_epilogue<E, MT, NT>(a, acc, tile_c * TC + wc * (TC / WC), tile_p * TP + wp * (TP / WP), lane);
}

template <typename E, int TC, int TP, int BK, int WC, int WP>
static int launch_conv(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_p = (a.M + TP - 1) / TP;
    a.tiles_c = (a.Cout + TC - 1) / TC;
    size_t smem = (size_t)2 * (TC + TP) * BK * sizeof(bf16_t);
    dim3 grid(a.tiles_p * a.tiles_c);
    hipLaunchKernelGGL((conv_igemm_kernel<E, TC, TP, BK, WC, WP>), grid, dim3(256), smem, stream, a);
    return cvpce_check_launch();
}

// second launch of a split-K conv: workgroup = one 64 x 64 quarter of an output tile (the (mt, nt) pair of a wave's accumulators),
// thread = the thread of conv_igemm_kernel<128, 128, 64, 2, 2> that held them; every partial of the thread is in flight before the
// first add (one round trip to memory for the launch; as a loop over the splits it was four), the adds run in split order, the
// epilogue is the unsplit kernel's
template <typename E, int S>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(ConvArgs a) {
    constexpr int MT = 2, NT = 2, Q = MT * NT * 4, WP = 2;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wc = wid / WP, wp = wid % WP;
    const int pair = blockIdx.x & 3, tile = blockIdx.x >> 2, ntile = a.tiles_c * a.tiles_p;
    const int mt = pair >> 1, nt = pair & 1;
    const int tile_c = tile % a.tiles_c, tile_p = tile / a.tiles_c;
    const f32x4* all = reinterpret_cast<const f32x4*>(a.ws) + ((size_t)tile * Q + pair * 4) * 256 + tid;
    const int ns = S ? S : a.ksplit;
    f32x16 acc[1][1];
    if constexpr (S != 0) {
        f32x4 v[S][4];
#pragma unroll
        for (int sp = 0; sp < S; ++sp)
#pragma unroll
            for (int g = 0; g < 4; ++g) v[sp][g] = all[(size_t)sp * ntile * Q * 256 + g * 256];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = v[0][g][j];
#pragma unroll
                for (int sp = 1; sp < S; ++sp) x += v[sp][g][j];
                acc[0][0][4 * g + j] = x;
            }
    } else {
        for (int sp = 0; sp < ns; ++sp)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = all[(size_t)sp * ntile * Q * 256 + g * 256];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0][0][4 * g + j] = sp == 0 ? v[j] : acc[0][0][4 * g + j] + v[j];
            }
    }
    conv_epilogue<E, 1, 1>(a, acc, tile_c * 128 + wc * 64 + mt * 32, tile_p * 128 + wp * 64 + nt * 32, lane);
}

// split-K launch pair of the 128 x 128 register-staged kernel
template <typename E>
static int launch_conv_split(const ConvArgs& a0, int ksplit, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_p = (a.M + 127) / 128;
    a.tiles_c = (a.Cout + 127) / 128;
    const long long ntile = (long long)a.tiles_p * a.tiles_c;
    if (ntile * ksplit >= (1LL << 24)) return CVPCE_ERR_ARG;
    const size_t need = (size_t)ksplit * ntile * 128 * 128 * sizeof(float);
    if (!workspace || workspace_bytes < need || ((size_t)workspace & 15)) return CVPCE_ERR_ARG;
    a.ksplit = ksplit;
    a.ws = (float*)workspace;
    const size_t smem = (size_t)2 * (128 + 128) * 64 * sizeof(bf16_t);
    hipLaunchKernelGGL((conv_igemm_kernel<E, 128, 128, 64, 2, 2, true>), dim3((unsigned)(ntile * ksplit)), dim3(256), smem, stream, a);
    const int rc = cvpce_check_launch();
    if (rc != CVPCE_OK) return rc;
    if (ksplit == 4) hipLaunchKernelGGL((conv_splitk_finish_kernel<E, 4>), dim3((unsigned)ntile * 4), dim3(256), 0, stream, a);
    else if (ksplit == 2) hipLaunchKernelGGL((conv_splitk_finish_kernel<E, 2>), dim3((unsigned)ntile * 4), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv_splitk_finish_kernel<E, 0>), dim3((unsigned)ntile * 4), dim3(256), 0, stream, a);
    return cvpce_check_launch();
}

template <typename E, int TC, int TP, int WC, int WP, int MINW>
static int launch_conv_dma(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_p = (a.M + TP - 1) / TP;
    a.tiles_c = (a.Cout + TC - 1) / TC;
    const size_t smem = (size_t)2 * (TC + TP) * 64 * sizeof(bf16_t);
    if (!cvpce_smem_attr_done<conv_dma_kernel<E, TC, TP, WC, WP, MINW>>((const void*)conv_dma_kernel<E, TC, TP, WC, WP, MINW>, (int)smem))
        return CVPCE_ERR_LAUNCH;     // (one flag per template instance and device)
    dim3 grid(a.tiles_p * a.tiles_c);
    hipLaunchKernelGGL((conv_dma_kernel<E, TC, TP, WC, WP, MINW>), grid, dim3(WC * WP * 64), smem, stream, a);
    return cvpce_check_launch();
}

template <typename E, int TC, int TP, int WC, int WP, int MINW, int NS>
static int launch_conv_dma4(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_p = (a.M + TP - 1) / TP;
    a.tiles_c = (a.Cout + TC - 1) / TC;
    const size_t smem = (size_t)NS * (TC + TP) * 32 * sizeof(bf16_t);
    if (!cvpce_smem_attr_done<conv_dma4_kernel<E, TC, TP, WC, WP, MINW, NS>>((const void*)conv_dma4_kernel<E, TC, TP, WC, WP, MINW, NS>, (int)smem))
        return CVPCE_ERR_LAUNCH;
    dim3 grid(a.tiles_p * a.tiles_c);
    hipLaunchKernelGGL((conv_dma4_kernel<E, TC, TP, WC, WP, MINW, NS>), grid, dim3(WC * WP * 64), smem, stream, a);
    return cvpce_check_launch();
}

template <typename E, int TC, int TP, int WC, int WP, int MINW, int NS>
static int launch_conv_dma16(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_p = (a.M + TP - 1) / TP;
    a.tiles_c = (a.Cout + TC - 1) / TC;
    const size_t smem = (size_t)NS * (TC + TP) * 32 * sizeof(bf16_t);
    if (!cvpce_smem_attr_done<conv_dma16_kernel<E, TC, TP, WC, WP, MINW, NS>>((const void*)conv_dma16_kernel<E, TC, TP, WC, WP, MINW, NS>, (int)smem))
        return CVPCE_ERR_LAUNCH;
    dim3 grid(a.tiles_p * a.tiles_c);
    hipLaunchKernelGGL((conv_dma16_kernel<E, TC, TP, WC, WP, MINW, NS>), grid, dim3(WC * WP * 64), smem, stream, a);
    return cvpce_check_launch();
}

#include <stdlib.h>
// dev A/B switch (environment, read once): CVPCE_NO_TC32=1 sends thin outputs through the 64-cout tile as before round 3
static bool use_tc32() {
    static const bool on = !(getenv("CVPCE_NO_TC32") && getenv("CVPCE_NO_TC32")[0] == '1');
    return on;
}

template <typename E>
static int conv2d_dispatch(const void* in, const void* wgt, const float* bias, const void* res,
                           void* out, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                           int stride, int pad, int Ho, int Wo, int K_pad, int Cout_pad,
                           int act, int out_f32, int in_up_shift, int res_mode, int Hr, int Wr,
                           int fuse_pool2, int force_generic, void* stream, int ksplit = 0, void* workspace = nullptr,
                           size_t workspace_bytes = 0) {
    if (N <= 0) return CVPCE_OK;
    if (!in || !wgt || !out) return CVPCE_ERR_ARG;
    if (Cin % 8 != 0 || K_pad % 32 != 0 || (Cin % 64 == 0 && K_pad % 64 != 0) || Cout_pad % 256 != 0 || Cout_pad < Cout)
        return CVPCE_ERR_ARG;
    if (K_pad < KH * KW * Cin) return CVPCE_ERR_ARG;
    if (!out_f32 && (Cout % 4 != 0)) return CVPCE_ERR_ARG;
    if (!out_f32 && act == 2) return CVPCE_ERR_ARG;
    if (in_up_shift < 0 || in_up_shift > 1) return CVPCE_ERR_ARG;
    if (res_mode && !res) return CVPCE_ERR_ARG;
    const long long Hl = (long long)H << in_up_shift, Wl = (long long)W << in_up_shift;
    if (Ho != (Hl + 2 * pad - KH) / stride + 1 || Wo != (Wl + 2 * pad - KW) / stride + 1) return CVPCE_ERR_ARG;
    if ((long long)N * H * W * Cin >= (1LL << 31) || (long long)N * Ho * Wo >= (1LL << 31) / 4) return CVPCE_ERR_ARG;
    if (res_mode == 1 && (Hr != Ho || Wr != Wo)) return CVPCE_ERR_ARG;
    if (fuse_pool2 && (out_f32 || res_mode || (Ho & 1) || (Wo & 1))) return CVPCE_ERR_ARG;
    ConvArgs a;
    a.in = (const bf16_t*)in; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.res = (const bf16_t*)res; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.Ho = Ho; a.Wo = Wo; a.K_pad = K_pad; a.M = N * Ho * Wo; a.relu = act; a.out_f32 = out_f32;
    a.in_up_shift = in_up_shift; a.res_mode = res_mode; a.Hr = Hr; a.Wr = Wr; a.pool = fuse_pool2 ? 1 : 0;
    a.tiles_p = a.tiles_c = 0;
    a.ksplit = 0; a.ws = nullptr;
    a.in_bytes = (unsigned)((long long)N * H * W * Cin * 2);
    a.wgt_bytes = (unsigned)((long long)Cout_pad * K_pad * 2);
    hipStream_t s = (hipStream_t)stream;
    const bool bk64 = (Cin % 64 == 0);
    if (ksplit) {
        // the caller asks for the split-K form (cvpce_conv2d_splitk_*): the 128-cout register-staged kernel's shapes only
        if (ksplit < 2 || ksplit > 16 || !bk64 || Cout <= 64 || fuse_pool2 || ksplit > K_pad / 64) return CVPCE_ERR_ARG;
        return launch_conv_split<E>(a, ksplit, workspace, workspace_bytes, s);
    }
    // LDS-DMA workhorse: K-step 64 within one tap, and enough pixel tiles to fill 256 CUs.
    //   Cout >= 192: 8 waves, 256x256 tile, 128 KiB LDS, 1 workgroup / CU (2 waves / SIMD)
    //   Cout <= 128: 4 waves, 128x128 or 64x128 tile, 64 / 48 KiB LDS, 2 workgroups / CU
    const long long tiles256 = ((long long)a.M + 255) / 256;
    if (bk64 && force_generic != 1) {
        if (Cout >= 192 && tiles256 * ((Cout + 255) / 256) >= 128)
            return (force_generic == 2 || in_up_shift || KH * KW > 32) ? launch_conv_dma<E, 256, 256, 2, 4, 2>(a, s)
                   : (force_generic == 3)                                ? launch_conv_dma4<E, 256, 256, 2, 4, 2, 4>(a, s)
                                                                        : launch_conv_dma16<E, 256, 256, 2, 4, 2, 4>(a, s);
        if (Cout > 64 && Cout <= 128 && tiles256 >= 128) return launch_conv_dma<E, 128, 128, 2, 2, 2>(a, s);
        if (Cout > 32 && Cout <= 64 && tiles256 >= 128) return launch_conv_dma<E, 64, 128, 2, 2, 2>(a, s);
    }
    if (Cout > 64) {
        return bk64 ? launch_conv<E, 128, 128, 64, 2, 2>(a, s) : launch_conv<E, 128, 128, 32, 2, 2>(a, s);
    } else if (Cout <= 32 && bk64 && use_tc32()) {
        // thin outputs on wide inputs (cls_logits 256 -> 9, the Gaussian subnet's 64 -> 32): a 32-cout tile halves the MFMA work spent
        // on cout padding (with K-step 32 a 32-row weight tile is less than one staging pass of the 256 threads: not instantiated)
        return launch_conv<E, 32, 128, 64, 1, 4>(a, s);
    } else {
        return bk64 ? launch_conv<E, 64, 128, 64, 2, 2>(a, s) : launch_conv<E, 64, 128, 32, 2, 2>(a, s);
    }
}

#define CVPCE_CONV2D_ARGS in, wgt, bias, res, out, N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, K_pad, Cout_pad, act, out_f32, \
                          in_up_shift, res_mode, Hr, Wr, fuse_pool2, force_generic, stream
extern "C" int cvpce_conv2d_nhwc_bf16(const void* in, const void* wgt, const float* bias, const void* res,
                                      void* out, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                      int stride, int pad, int Ho, int Wo, int K_pad, int Cout_pad,
                                      int act, int out_f32, int in_up_shift, int res_mode, int Hr, int Wr,
                                      int fuse_pool2, int force_generic, void* stream) {
    return conv2d_dispatch<ElemBF16>(CVPCE_CONV2D_ARGS);
}
extern "C" int cvpce_conv2d_nhwc_f16(const void* in, const void* wgt, const float* bias, const void* res,
                                     void* out, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                     int stride, int pad, int Ho, int Wo, int K_pad, int Cout_pad,
                                     int act, int out_f32, int in_up_shift, int res_mode, int Hr, int Wr,
                                     int fuse_pool2, int force_generic, void* stream) {
    return conv2d_dispatch<ElemF16>(CVPCE_CONV2D_ARGS);
}
#undef CVPCE_CONV2D_ARGS

// Split-K form of the register-staged kernel for small maps with a long K (ResNet-50 layer4's 3x3 convs on 25 x 25 maps, the FPN's
// P5 / P6 / P7 convs): `ksplit` workgroups per 128 x 128 output tile, fp32 partial tiles through `workspace`, summed in split order by
// a second launch (results do not vary from run to run; they differ from the unsplit kernel's in the last bits of the fp32 sum).
// workspace: cvpce_conv2d_splitk_workspace_bytes(M = N Ho Wo, Cout, ksplit) bytes, 16-byte aligned, used by one conv at a time.
extern "C" size_t cvpce_conv2d_splitk_workspace_bytes(long long M, int Cout, int ksplit) {
    if (M <= 0 || Cout <= 0 || ksplit < 2) return 0;
    const long long ntile = ((M + 127) / 128) * ((Cout + 127) / 128);
    return (size_t)ksplit * ntile * 128 * 128 * sizeof(float);
}
extern "C" int cvpce_conv2d_splitk_bf16(const void* in, const void* wgt, const float* bias, const void* res,
                                        void* out, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                        int stride, int pad, int Ho, int Wo, int K_pad, int Cout_pad,
                                        int act, int out_f32, int in_up_shift, int res_mode, int Hr, int Wr,
                                        int ksplit, void* workspace, size_t workspace_bytes, void* stream) {
    if (ksplit < 2) return CVPCE_ERR_ARG;
    return conv2d_dispatch<ElemBF16>(in, wgt, bias, res, out, N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, K_pad, Cout_pad, act, out_f32,
                                     in_up_shift, res_mode, Hr, Wr, 0, 0, stream, ksplit, workspace, workspace_bytes);
}
extern "C" int cvpce_conv2d_splitk_f16(const void* in, const void* wgt, const float* bias, const void* res,
                                       void* out, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                       int stride, int pad, int Ho, int Wo, int K_pad, int Cout_pad,
                                       int act, int out_f32, int in_up_shift, int res_mode, int Hr, int Wr,
                                       int ksplit, void* workspace, size_t workspace_bytes, void* stream) {
    if (ksplit < 2) return CVPCE_ERR_ARG;
    return conv2d_dispatch<ElemF16>(in, wgt, bias, res, out, N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, K_pad, Cout_pad, act, out_f32,
                                    in_up_shift, res_mode, Hr, Wr, 0, 0, stream, ksplit, workspace, workspace_bytes);
}
