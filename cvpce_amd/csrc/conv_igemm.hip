// Implicit-GEMM convolution for gfx950 (CDNA4): NHWC bf16 activations, bf16
// weights packed [Cout_pad][K_pad] with K = (kh*KW + kw)*Cin + ci, fp32
// accumulation on v_mfma_f32_32x32x16_bf16, fused bias / residual(+nearest
// upsample) / ReLU epilogue, bf16 or fp32 NHWC output.
//
// GEMM view (transposed so that the accumulator's register axis runs along
// Cout -> each lane owns 4 consecutive output channels of ONE pixel and the
// NHWC store is 8 B/lane, 64 B contiguous per pixel):
//     D[cout][pixel] = sum_k Wgt[cout][k] * Im2col[pixel][k]
// The im2col matrix is never materialised: every 16-byte K-chunk of a pixel
// row is gathered straight from the NHWC tensor (one tap, 8 channels), with
// zero fill for the padding halo, staged through registers into an
// XOR-swizzled LDS tile (conflict-free ds_read_b128 fragments), double
// buffered with the global loads of K-step t+1 in flight under the MFMAs of
// step t.
//
// Covers every convolution of the hot path:
//   ResNet-50 7x7 s2 / 1x1 / 3x3 s1,s2; FPN 1x1, 3x3, 3x3 s2; RetinaNet head
//   3x3; Gaussian branch 1x1/3x3 (incl. reading a nearest-2x-upsampled input
//   without materialising it); VGG16 3x3.
// Reference semantics: torch.nn.Conv2d as used at
//   /root/reference/cvpce/models/proposals.py:54,68,84 and torchvision 0.9
//   resnet/vgg/fpn/retinanet.
#include "common.h"
#include "../../include/cvpce_amd.h"

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
    int tiles_p, tiles_c;
};

template <int TC, int TP, int BK, int WC, int WP>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
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
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_c = swz % a.tiles_c, tile_p = swz / a.tiles_c;

    const int c = tid % CPR;
    const int r0 = tid / CPR;

    const int Hl = a.H << a.in_up_shift, Wl = a.W << a.in_up_shift;
    const int HoWo = a.Ho * a.Wo;

    int pbase[PPASS], piy[PPASS], pix[PPASS];
#pragma unroll
    for (int i = 0; i < PPASS; ++i) {
        int m = tile_p * TP + r0 + i * RPP;
        if (m < a.M) {
            int img = m / HoWo;
            int rem = m - img * HoWo;
            int oy = rem / a.Wo;
            int ox = rem - oy * a.Wo;
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

    const bf16_t* wrow = a.wgt + (size_t)(tile_c * TC + r0) * a.K_pad + c * 8;

    u32x4 wreg[WPASS], preg[PPASS];

#define CVPCE_LOAD_TILE(KT)                                                                                    \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < WPASS; ++i) wreg[i] =                                            \
            *reinterpret_cast<const u32x4*>(wrow + (size_t)i * RPP * a.K_pad + (KT) * BK);                     \
        const bool tap_ok = kh < a.KH;                                                                         \
        _Pragma("unroll") for (int i = 0; i < PPASS; ++i) {                                                    \
            const int iy = piy[i] + kh, ix = pix[i] + kw;                                                      \
            const bool ok = tap_ok && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;              \
            const size_t off =                                                                                 \
                (size_t)(pbase[i] + (iy >> a.in_up_shift) * a.W + (ix >> a.in_up_shift)) * a.Cin + ci;         \
            u32x4 v = {0u, 0u, 0u, 0u};                                                                          \
            if (ok) v = *reinterpret_cast<const u32x4*>(a.in + off);                                           \
            preg[i] = v;                                                                                       \
        }                                                                                                      \
        ci += BK;                                                                                              \
        while (ci >= a.Cin) {                                                                                  \
            ci -= a.Cin;                                                                                       \
            if (++kw == a.KW) { kw = 0; ++kh; }                                                                \
        }                                                                                                      \
    }
#define CVPCE_STORE_TILE(BUF)                                                                                  \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < WPASS; ++i) {                                                    \
            const int row = r0 + i * RPP;                                                                      \
            const int phys = c ^ ((row / RPB) & (CPR - 1));                                                    \
            *reinterpret_cast<u32x4*>(Ws + (BUF) * TC * BK + row * BK + phys * 8) = wreg[i];                   \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < PPASS; ++i) {                                                    \
            const int row = r0 + i * RPP;                                                                      \
            const int phys = c ^ ((row / RPB) & (CPR - 1));                                                    \
            *reinterpret_cast<u32x4*>(Ps + (BUF) * TP * BK + row * BK + phys * 8) = preg[i];                   \
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
    CVPCE_LOAD_TILE(0)
    CVPCE_STORE_TILE(0)
    __syncthreads();
    int cur = 0;
    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) CVPCE_LOAD_TILE(kt + 1)
        const bf16_t* Wb = Ws + cur * TC * BK;
        const bf16_t* Pb = Ps + cur * TP * BK;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int chunk = kk * 2 + lh;
            bf16x8 af[MT], bfr[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int row = wc * (TC / WC) + mt * 32 + lr;
                af[mt] = *reinterpret_cast<const bf16x8*>(Wb + row * BK + ((chunk ^ ((row / RPB) & (CPR - 1))) * 8));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                int row = wp * (TP / WP) + nt * 32 + lr;
                bfr[nt] = *reinterpret_cast<const bf16x8*>(Pb + row * BK + ((chunk ^ ((row / RPB) & (CPR - 1))) * 8));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
        if (more) CVPCE_STORE_TILE(cur ^ 1)
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane owns pixel (lane&31) of each 32-wide pixel tile and
    // channels 8g + 4h + {0..3} of each 32-channel tile.
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int m = tile_p * TP + wp * (TP / WP) + nt * 32 + lr;
        if (m >= a.M) continue;
        size_t res_pix = 0;
        if (a.res_mode == 1) {
            res_pix = (size_t)m;
        } else if (a.res_mode == 2) {
            int img = m / HoWo;
            int rem = m - img * HoWo;
            int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            int ry = (oy * a.Hr) / a.Ho, rx = (ox * a.Wr) / a.Wo;
            res_pix = (size_t)(img * a.Hr + ry) * a.Wr + rx;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = tile_c * TC + wc * (TC / WC) + mt * 32 + 8 * g + 4 * lh;
                if (co >= a.Cout) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][4 * g + j];
                if (a.out_f32) {
                    float* o = reinterpret_cast<float*>(a.out) + (size_t)m * a.Cout;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (co + j < a.Cout) {
                            float x = v[j];
                            if (a.bias) x += a.bias[co + j];
                            if (a.res_mode) x += bf16_to_f32(a.res[res_pix * a.Cout + co + j]);
                            if (a.relu == 1) x = fmaxf(x, 0.f);
                            else if (a.relu == 2) x = tanhf(x);
                            o[co + j] = x;
                        }
                    }
                } else {
                    // Cout % 4 == 0 is required for bf16 outputs (checked on the host)
                    if (a.bias) {
                        f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += b[j];
                    }
                    if (a.res_mode) {
                        bf16x4 r = *reinterpret_cast<const bf16x4*>(a.res + res_pix * a.Cout + co);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += bf16_to_f32(r[j]);
                    }
                    if (a.relu == 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f32_to_bf16(v[j]);
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(a.out) + (size_t)m * a.Cout + co) = o;
                }
            }
        }
    }
}

template <int TC, int TP, int BK, int WC, int WP>
static int launch_conv(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_p = (a.M + TP - 1) / TP;
    a.tiles_c = (a.Cout + TC - 1) / TC;
    size_t smem = (size_t)2 * (TC + TP) * BK * sizeof(bf16_t);
    dim3 grid(a.tiles_p * a.tiles_c);
    hipLaunchKernelGGL((conv_igemm_kernel<TC, TP, BK, WC, WP>), grid, dim3(256), smem, stream, a);
    return cvpce_check_launch();
}

extern "C" int cvpce_conv2d_nhwc_bf16(const void* in, const void* wgt, const float* bias, const void* res,
                                      void* out, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                      int stride, int pad, int Ho, int Wo, int K_pad, int Cout_pad,
                                      int act, int out_f32, int in_up_shift, int res_mode, int Hr, int Wr,
                                      void* stream) {
    if (N <= 0) return CVPCE_OK;
    if (!in || !wgt || !out) return CVPCE_ERR_ARG;
    if (Cin % 8 != 0 || K_pad % 64 != 0 || Cout_pad % 128 != 0 || Cout_pad < Cout) return CVPCE_ERR_ARG;
    if (K_pad < KH * KW * Cin) return CVPCE_ERR_ARG;
    if (!out_f32 && (Cout % 4 != 0)) return CVPCE_ERR_ARG;
    if (!out_f32 && act == 2) return CVPCE_ERR_ARG;
    if (in_up_shift < 0 || in_up_shift > 1) return CVPCE_ERR_ARG;
    if (res_mode && !res) return CVPCE_ERR_ARG;
    const long long Hl = (long long)H << in_up_shift, Wl = (long long)W << in_up_shift;
    if (Ho != (Hl + 2 * pad - KH) / stride + 1 || Wo != (Wl + 2 * pad - KW) / stride + 1) return CVPCE_ERR_ARG;
    if ((long long)N * H * W * Cin >= (1LL << 31) || (long long)N * Ho * Wo >= (1LL << 31) / 4) return CVPCE_ERR_ARG;
    if (res_mode == 1 && (Hr != Ho || Wr != Wo)) return CVPCE_ERR_ARG;
    ConvArgs a;
    a.in = (const bf16_t*)in; a.wgt = (const bf16_t*)wgt; a.bias = bias; a.res = (const bf16_t*)res; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.Ho = Ho; a.Wo = Wo; a.K_pad = K_pad; a.M = N * Ho * Wo; a.relu = act; a.out_f32 = out_f32;
    a.in_up_shift = in_up_shift; a.res_mode = res_mode; a.Hr = Hr; a.Wr = Wr; a.tiles_p = a.tiles_c = 0;
    hipStream_t s = (hipStream_t)stream;
    const bool bk64 = (Cin % 64 == 0);
    if (Cout > 64) {
        return bk64 ? launch_conv<128, 128, 64, 2, 2>(a, s) : launch_conv<128, 128, 32, 2, 2>(a, s);
    } else {
        return bk64 ? launch_conv<64, 128, 64, 2, 2>(a, s) : launch_conv<64, 128, 32, 2, 2>(a, s);
    }
}
