// The detector's stem in one launch: conv 7x7 stride 2 pad 3 (3 -> 64, FrozenBatchNorm folded into weights and bias)
// + ReLU + MaxPool2d(3, stride 2, pad 1)  --  torchvision resnet50 `conv1 / bn1 / relu / maxpool`, reached from
// cvpce/models/proposals.py:202-216 (`resnet_fpn_backbone`) -> `GaussianLayerNetwork.forward` (proposals.py:166-168).
//
// The generic implicit-GEMM kernel pads K = 7*7*3 = 147 to 7*7*8 = 392 (it wants 8-channel pixels), writes the
// 64-channel 1/2-resolution map (164 MB for 8 images of 800^2) and a second launch pools it.  Here one workgroup owns
// an 8x8 tile of POOLED pixels:
//   * the 39x39 input pixels under it are staged once into LDS as 4-channel pixels (8 bytes; channel 3 = 0);
//   * K is laid out as 7 kernel rows x 8 kw slots x 4 channels = 224 (slot 7 and channel 3 carry zero weights), so a
//     K-step of 16 is four horizontally adjacent input pixels = ONE 16-byte LDS read per lane, 14 K-steps per pixel tile;
//   * the weights (64 x 224 bf16) sit in registers in MFMA layout: a wave holds the 14 fragments of its 32 output
//     channels for the whole kernel;
//   * the 17x17 convolution pixels of the tile go through LDS as bf16 (bias + ReLU applied), zero where they fall outside
//     the convolution map (max-pool padding never wins against a ReLU output), then 3x3/2 pooled and stored.
// v_mfma_f32_32x32x16_bf16: A = weights (32 couts x 16 k), B = pixels (16 k x 32 px).
#include "common.h"

namespace {

constexpr int GS_PT = 8;                    // pooled tile edge
constexpr int GS_CT = 2 * GS_PT + 1;        // 17 conv pixels per edge
constexpr int GS_IT = 2 * GS_CT + 5;        // 39 input pixels per edge
constexpr int GS_IP = 40;                   // LDS input row pitch in pixels (16-byte aligned rows, kw slot 7 stays in the row)
constexpr int GS_NPIX = GS_CT * GS_CT;      // 289
constexpr int GS_PXT = (GS_NPIX + 31) / 32; // 10 pixel tiles of 32
constexpr int GS_CP = 136;                  // bytes per conv pixel in LDS (64 bf16 + 8 pad: 8-byte writes spread over banks)
constexpr int GS_IN_BYTES = GS_IT * GS_IP * 8;          // 12480
constexpr int GS_CONV_BYTES = GS_NPIX * GS_CP;          // 39304
constexpr int GS_THREADS = 256;

struct GlnStemArgs {
    const unsigned char* in;   // N x H x W x 8 bf16 (channels 0..2 used)
    const bf16x8* w;           // [2 ct][14 k-steps][64 lanes] fragments
    const float* bias;         // 64
    bf16_t* out;               // N x Hp x Wp x 64
    int N, H, W, Hc, Wc, Hp, Wp, tiles_x, tiles_y;
};

template <typename E>
__global__ __launch_bounds__(GS_THREADS) void gln_stem_kernel(GlnStemArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char IN[GS_IN_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char CV[GS_CONV_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int per_image = a.tiles_x * a.tiles_y;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n = bid / per_image, t = bid - n * per_image;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int ct = wave & 1, half = wave >> 1;

    // weights: 14 fragments of this wave's 32 output channels
    bf16x8 wf[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) wf[s] = a.w[(ct * 14 + s) * 64 + lane];

    // input patch: rows 4*py0 - 5 .. + 38
    const int iy0 = 4 * ty * GS_PT - 5, ix0 = 4 * tx * GS_PT - 5;
    const unsigned char* img = a.in + (size_t)n * a.H * a.W * 16;
    for (int i = tid; i < GS_IT * GS_IP; i += GS_THREADS) {
        const int r = i / GS_IP, c = i - r * GS_IP;
        const int y = iy0 + r, x = ix0 + c;
        uint2 v = make_uint2(0u, 0u);
        if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W)
            v = *reinterpret_cast<const uint2*>(img + ((size_t)y * a.W + x) * 16);
        v.y &= 0xFFFFu;                                   // channel 3 (NHWC8 padding is zero already; keep it certain)
        *reinterpret_cast<uint2*>(IN + i * 8) = v;
    }
    __syncthreads();

    float bias[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[4 * g + j] = a.bias[ct * 32 + 8 * g + 4 * lh + j];

    const int cy0 = 2 * ty * GS_PT - 1, cx0 = 2 * tx * GS_PT - 1;
#pragma unroll 1
    for (int pt = half; pt < GS_PXT; pt += 2) {
        int p = pt * 32 + lr;
        const bool real = p < GS_NPIX;
        if (!real) p = GS_NPIX - 1;
        const int cy = p / GS_CT, cx = p - cy * GS_CT;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bias[i];
        const unsigned char* src = IN + ((2 * cy) * GS_IP + 2 * cx + 2 * lh) * 8;
        bf16x8 bfr[14];
#pragma unroll
        for (int s = 0; s < 14; ++s)                       // s = kh*2 + h: pixels 2cx + 4h + 2lh, +1 of row 2cy + kh
            bfr[s] = *reinterpret_cast<const bf16x8*>(src + ((s >> 1) * GS_IP + 4 * (s & 1)) * 8);
#pragma unroll
        for (int s = 0; s < 14; ++s) acc = E::mfma32(wf[s], bfr[s], acc);
        const int y = cy0 + cy, x = cx0 + cx;
        const unsigned keep = ((unsigned)y < (unsigned)a.Hc && (unsigned)x < (unsigned)a.Wc) ? 0xFFFFFFFFu : 0u;
        if (real) {
            unsigned char* dst = CV + p * GS_CP + (ct * 32 + 4 * lh) * 2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 r;
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = relu_bits(acc[4 * g + j]);
                uint2 u = __builtin_bit_cast(uint2, E::pack4(r));
                u.x &= keep;
                u.y &= keep;
                *reinterpret_cast<uint2*>(dst + g * 16) = u;
            }
        }
    }
    __syncthreads();

    // 3x3 stride-2 max-pool of the 17x17 tile: thread <-> (pooled pixel, 4-channel chunk); non-negative bf16 / fp16 values compare as int16
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    for (int i = tid; i < GS_PT * GS_PT * 16; i += GS_THREADS) {
        const int ch = i & 15, pp = i >> 4;
        const int ppy = pp / GS_PT, ppx = pp - ppy * GS_PT;
        const int py = ty * GS_PT + ppy, px = tx * GS_PT + ppx;
        if (py >= a.Hp || px >= a.Wp) continue;
        s16x4 m = {0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const s16x4 v = *reinterpret_cast<const s16x4*>(CV + ((2 * ppy + dy) * GS_CT + 2 * ppx + dx) * GS_CP + ch * 8);
                m = __builtin_elementwise_max(m, v);
            }
        *reinterpret_cast<s16x4*>(reinterpret_cast<unsigned char*>(a.out) + (((size_t)n * a.Hp + py) * a.Wp + px) * 128 + ch * 8) = m;
    }
}

}  // namespace

// in: N x H x W x 8 bf16 (what cvpce_gln_transform writes); w_frag: the 64 x (7 x 8 x 4) weights in MFMA fragment order
// [ct][kh*2+h][lane][8] with lane = lh*32 + cout%32 holding kw = 4h + 2lh + j/4, channel j%4 (zeros for kw = 7 and
// channel 3); bias: 64 floats; out: N x Hp x Wp x 64 bf16 with Hc = (H - 1)/2 + 1, Hp = (Hc - 1)/2 + 1.
template <typename E>
static int gln_stem_dispatch(const void* in, const void* w_frag, const float* bias, void* out, int N, int H, int W, void* stream) {
    if (!in || !w_frag || !bias || !out || N <= 0 || H <= 0 || W <= 0) return CVPCE_ERR_ARG;
    GlnStemArgs a;
    a.in = (const unsigned char*)in;
    a.w = (const bf16x8*)w_frag;
    a.bias = bias;
    a.out = (bf16_t*)out;
    a.N = N; a.H = H; a.W = W;
    a.Hc = (H - 1) / 2 + 1; a.Wc = (W - 1) / 2 + 1;
    a.Hp = (a.Hc - 1) / 2 + 1; a.Wp = (a.Wc - 1) / 2 + 1;
    a.tiles_y = (a.Hp + GS_PT - 1) / GS_PT; a.tiles_x = (a.Wp + GS_PT - 1) / GS_PT;
    const long long blocks = (long long)N * a.tiles_x * a.tiles_y;
    if (blocks > 0x7FFFFFFFLL || (long long)N * H * W * 16 > 0xFFFFFFFFFFLL) return CVPCE_ERR_ARG;
    hipLaunchKernelGGL(gln_stem_kernel<E>, dim3((unsigned)blocks), dim3(GS_THREADS), 0, (hipStream_t)stream, a);
    return cvpce_check_launch();
}

extern "C" int cvpce_gln_stem_fused(const void* in, const void* w_frag, const float* bias, void* out, int N, int H, int W,
                                    void* stream) {
    return gln_stem_dispatch<ElemBF16>(in, w_frag, bias, out, N, H, W, stream);
}
extern "C" int cvpce_gln_stem_fused_f16(const void* in, const void* w_frag, const float* bias, void* out, int N, int H, int W,
                                        void* stream) {
    return gln_stem_dispatch<ElemF16>(in, w_frag, bias, out, N, H, W, stream);
}
