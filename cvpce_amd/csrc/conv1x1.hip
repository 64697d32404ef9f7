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

    // ---- epilogue: lane (pixel l16 of block nt, q = lq) holds couts ct*64 + 16q .. +15 ----
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
            const bf16x8 r0 = *reinterpret_cast<const bf16x8*>(a.res + rp * a.Cout + co);
            const bf16x8 r1 = *reinterpret_cast<const bf16x8*>(a.res + rp * a.Cout + co + 8);
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
        bf16_t* dst = a.out + (size_t)m * a.Cout + co;
        *reinterpret_cast<u32x4*>(dst) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(dst + 8) = u32x4{pk[4], pk[5], pk[6], pk[7]};
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
