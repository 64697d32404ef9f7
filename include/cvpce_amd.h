/* cvpce_amd -- C ABI of the MI355X (gfx950) hot path of laitalaj/cvpce.
 *
 * The reference has no FFI of its own (it is pure Python over PyTorch /
 * torchvision); the drop-in boundary is its Python API.  This
 * header is the layer directly under that API: plain pointers + sizes, no
 * torch types.  Every pointer is a DEVICE pointer unless marked [host]; every
 * call is asynchronous on `stream` (a hipStream_t passed as void*), performs no
 * allocation and no synchronisation, and returns CVPCE_OK (0) or an error code
 * (1 = bad argument, 2 = launch failure).  Citations are to /root/reference.
 *
 * Tensor layouts
 *   activations : NHWC bf16, C % 8 == 0 (3-channel images are carried as C = 8,
 *                 channels 3..7 zero)
 *   conv weights: bf16 [Cout_pad][K_pad], zero padded, Cout_pad % 256 == 0;
 *                 Cin % 64 == 0: K index = (((ci/64)*KH + kh)*KW + kw)*64 + ci%64, K_pad % 64 == 0
 *                                (chunk-major: the KH*KW taps of a 64-channel chunk are adjacent)
 *                 otherwise    : K index = (kh*KW + kw)*Cin + ci, K_pad % 32 == 0
 *                 (FrozenBN / BatchNorm-eval already folded in by the host)
 *   halo weight layout (the `wgt` of every cvpce_conv3x3_halo* entry point; 3x3, Cin % 64 == 0): the same Cout_pad x K_pad
 *                 values FRAGMENT-MAJOR, one contiguous KiB per MFMA weight fragment --
 *                 [ci/64][cout/32][kw][(ci%64)/32][kh][mt][lane][8] with lane = 16 q + m, m = 0..15, q = 0..3:
 *                 lane (m, q) of block mt holds cout 32 (cout/32) + 8 (m >> 2) + 4 mt + (m & 3), channels
 *                 64 (ci/64) + 32 ((ci%64)/32) + 8 q .. + 7 of tap (kh, kw).  cvpce_pack_halo_weights converts.
 */
#ifndef CVPCE_AMD_H
#define CVPCE_AMD_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CVPCE_ACT_NONE 0
#define CVPCE_ACT_RELU 1
#define CVPCE_ACT_TANH 2 /* fp32 outputs only (GaussianSubnet tanh head, proposals.py:85) */

/* Implicit-GEMM convolution + fused epilogue  out = act(conv(in) + bias + residual).
 * Replaces every nn.Conv2d on the path: torchvision ResNet-50/FPN/RetinaNetHead/VGG16
 *  and cvpce/models/proposals.py:54,68,84 (Gaussian branch).
 *   in_up_shift 1 : the logical input is the nearest-2x upsample of `in`
 *                   (GaussianLayer.forward `self.up(x)`, proposals.py:79) -- not materialised
 *   res_mode 0/1/2: none / same-size residual (Bottleneck add) / nearest-resampled
 *                   residual [N][Hr][Wr][Cout] (FPN top-down add; proposals.py:76)
 *   out_f32       : NHWC fp32 output (head logits / box regressions / gaussians)
 *   fuse_pool2    : the following nn.MaxPool2d(2,2) is taken in the epilogue; `out` is
 *                   [N][Ho/2][Wo/2][Cout] (VGG16 `features` conv+ReLU+pool triples)
 *   force_generic : 1 = always use the register-staged fallback kernel (A/B tests) */
int cvpce_conv2d_nhwc_bf16(const void* in, const void* wgt, const float* bias, const void* res, void* out,
                           int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                           int Ho, int Wo, int K_pad, int Cout_pad, int act, int out_f32, int in_up_shift,
                           int res_mode, int Hr, int Wr, int fuse_pool2, int force_generic, void* stream);

/* Split-K form of cvpce_conv2d_nhwc_bf16 for small maps with a long K (torchvision ResNet-50 layer4's 3x3 convs on the 25 x 25 maps of
 * an 800-pixel image, the stride-2 3x3 convs that open layer3 / layer4, the FPN's P5 output conv and P6 / P7 -- proposals.py:109-139
 * builds them): as the unsplit launch these are a chain of 36-72 K-steps on a few dozen workgroups.  Same operands, layouts and
 * epilogue (Cin % 64 == 0, Cout > 64, no fused pool); `ksplit` (2..16) workgroups per 128 x 128 output tile, each over nk / ksplit
 * K-steps, fp32 partial tiles through `workspace`, added in split order by a second launch on the same stream -- the result does not
 * vary from run to run; it differs from the unsplit kernel's in the last bits of the fp32 sum.  workspace: at least
 * cvpce_conv2d_splitk_workspace_bytes(N Ho Wo, Cout, ksplit) bytes, 16-byte aligned, one conv at a time. */
size_t cvpce_conv2d_splitk_workspace_bytes(long long M, int Cout, int ksplit);
int cvpce_conv2d_splitk_bf16(const void* in, const void* wgt, const float* bias, const void* res, void* out,
                             int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                             int Ho, int Wo, int K_pad, int Cout_pad, int act, int out_f32, int in_up_shift,
                             int res_mode, int Hr, int Wr, int ksplit, void* workspace, size_t workspace_bytes, void* stream);

/* 1x1 convolution (nn.Conv2d(k=1, stride s, pad 0) of the ResNet-50 bottlenecks / downsample paths, the FPN lateral convs and the
 * Gaussian layer's lateral, proposals.py:68) as a pointwise GEMM: Cin % 64 == 0 (= K_pad), Cout % 64 == 0; same operands, weight
 * layout, residual modes (0 none, 1 same-size add, 2 nearest-upsampled add from [N][Hr][Wr][Cout]) and numerics as
 * cvpce_conv2d_nhwc_bf16; relu = 0/1; out: [N][Ho][Wo][Cout] bf16 with Ho = (H-1)/stride + 1.  stride 1: streaming kernels (both
 * operands through per-wave LDS by coalesced LDS-DMA, no barrier, persistent waves; Cin <= 256 and a power-of-two Cout / 64: the
 * wave's weights stay in registers); other strides: operands loaded in fragment layout. */
int cvpce_conv1x1_nhwc_bf16(const void* in, const void* wgt, const float* bias, const void* res, void* out, int N, int H,
                            int W, int Cin, int Cout, int stride, int Ho, int Wo, int K_pad, int Cout_pad, int relu,
                            int res_mode, int Hr, int Wr, void* stream);

/* Fused VGG16 stem: features[0:5] = conv3x3(3->64)+ReLU, conv3x3(64->64)+ReLU, MaxPool2d(2,2)
 * (torchvision vgg cfg 'D'; reached from cvpce/models/classification.py:27,36) as one persistent kernel with
 * both layers' weights resident in LDS.  in: NHWC bf16 with `in_cstride` (4 or 8) channels per pixel,
 * channels 0..2 used, channel 3 zero; H, W multiples of 16.  w1: bf16 [64][48], k = kh*16 + kw*4 + c
 * (slots kw = 3 and c = 3 zero); w2: bf16 [9][64][64] = (tap, cout, cin); out: [N][H/2][W/2][64] bf16. */
int cvpce_vgg_stem_fused(const void* in_nhwc, int in_cstride, const void* w1, const float* b1, const void* w2,
                         const float* b2, void* out, int N, int H, int W, void* stream);

/* Fused detector stem: torchvision resnet50 conv1 (7x7, stride 2, pad 3, 3 -> 64) + FrozenBatchNorm2d (folded into
 * w/bias) + ReLU + MaxPool2d(3, 2, 1), reached from cvpce/models/proposals.py:202-216 (resnet_fpn_backbone) via
 * GaussianLayerNetwork.forward (proposals.py:166-168).  in: [N][H][W][8] bf16 as cvpce_gln_transform writes it (channels
 * 0..2 used).  w_frag: bf16 [2][14][64][8], the 64 x (7 kh x 8 kw slots x 4 channels) weights in MFMA fragment order:
 * entry [ct][kh*2 + h][lh*32 + r][j] = w[ct*32 + r][c = j%4][kh][kw = 4h + 2lh + j/4], zero for kw = 7 and c = 3.
 * bias: 64 floats.  out: [N][Hp][Wp][64] bf16 with Hc = (H-1)/2 + 1, Hp = (Hc-1)/2 + 1 (same for W).  Numerics: the
 * convolution output is rounded to bf16 before pooling, exactly as conv2d_nhwc_bf16 followed by maxpool2d_nhwc_bf16. */
int cvpce_gln_stem_fused(const void* in_nhwc8, const void* w_frag, const float* bias, void* out, int N, int H, int W,
                         void* stream);

/* 3x3 / stride 1 / pad 1 convolution of a THIN layer (GaussianSubnet, cvpce/models/proposals.py:81-107; row-major weights as for
 * cvpce_conv2d_nhwc_bf16): in_up_shift 0: Cin = 32 (K_pad 288), Cout = 16 | 32 -- the 3x3 32 -> 32 and 32 -> 16 over the 400 x 400 map;
 * in_up_shift 1: Cin = 64 (K_pad 576), Cout = 32, `in` is [N][H/2][W/2][64] and is read through its nearest-2x upsample
 * (GaussianLayer's `self.up`, proposals.py:79, never materialised) -- the subnet's first layer.  H, W: the OUTPUT size.  A wave
 * keeps the layer's weights in registers and walks down a 16-pixel column strip with a ring of input rows in LDS; relu = 0/1; out bf16
 * [N][H][W][Cout]. */
int cvpce_conv3x3_thin_bf16(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W, int Cin, int Cout,
                            int K_pad, int Cout_pad, int relu, int in_up_shift, void* stream);

/* [host] The row-major conv weights of a 3x3 layer with Cin % 64 == 0 ([Cout_pad][K_pad = 9 Cin] 16-bit values, chunk-major K)
 * into the fragment-major halo weight layout (same size; `src` and `dst` are HOST buffers and must not overlap). */
int cvpce_pack_halo_weights(const void* src_rowmajor, void* dst_halo, int Cout_pad, int Cin);

/* 3x3 / stride 1 / pad 1 convolution with Cin % 64 == 0 and the input halo patch resident in LDS (VGG16 conv2_2 ..
 * conv5_3, RetinaNet head / FPN 3x3s): same operands and numerics as cvpce_conv2d_nhwc_bf16 with the weights in the HALO
 * WEIGHT LAYOUT (a wave's weight fragment is one contiguous KiB: row-major, its 16 row pieces cost the texture addresser
 * one lane per clock); any H, W
 * (ragged 16x16 tiles are masked; H, W even when pooling), Cout % 8 == 0;
 * relu = 0/1; fuse_pool2 = 1 stores MaxPool2d(2,2) of the result ([N][H/2][W/2][Cout]). */
int cvpce_conv3x3_halo(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W, int Cin,
                       int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, void* stream);
/* Same contract for layers with FEW output channels (Cout <= 128 per cout tile: VGG16 conv2_1 / conv2_2): 16x32-pixel
 * tiles and 32-channel sub-chunks keep the 32-cout x 256-pixel wave tile of cvpce_conv3x3_halo, so every weight
 * fragment is fetched once and feeds 16 MFMAs. */
int cvpce_conv3x3_halo_wide(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W, int Cin,
                            int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, void* stream);
/* cvpce_conv3x3_halo (ReLU on, Cout > 128) with the MAC descriptor of classification.py:46-49 fused into the epilogue:
 * mac[n * mac_stride + mac_off + c] = max over the post-ReLU output map of image n, channel c -- exactly
 * `x.amax(dim=(-2, -1))` of the bf16 map the convolution would store (rounding is monotonic).  `mac` must be zero-filled
 * by the caller (values are >= 0; the kernel takes an atomic max per tile).  `out` may be NULL: the map itself is then
 * never written (VGG16 conv5_3, whose only consumer is the descriptor); with fuse_pool2 = 1 `out` receives
 * MaxPool2d(2,2) of the map (conv4_3 feeding pool4), the descriptor is still taken over the unpooled map. */
int cvpce_conv3x3_halo_mac(const void* in, const void* wgt, const float* bias, void* out, float* mac, int mac_stride,
                           int mac_off, int N, int H, int W, int Cin, int Cout, int K_pad, int Cout_pad, int fuse_pool2,
                           void* stream);
/* cvpce_conv3x3_halo on a LEVEL ATLAS: several feature maps that share the conv weights (the 5 FPN levels under the
 * RetinaNet head, torchvision RetinaNetHead reached from cvpce/models/proposals.py:166) are packed side by side into one
 * [N][H][W][Cin] canvas with >= 1 zero pixel between them; mask is [H][W] bytes, 1 on level pixels, 0 on the gaps.
 * Outputs on mask-0 pixels are stored as zeros, so the result is again a valid atlas (the gaps ARE the next layer's
 * zero padding) and one launch replaces one launch per level.  tile_map (optional, device): the n_tiles 16x16-pixel tiles
 * to compute, (ty << 16) | tx, the same list for every image -- tiles that lie wholly in a gap are left out and `out` must
 * then already hold zeros there (the caller keeps zero-initialised atlas buffers).  At most 512 x (workgroups of the persistent
 * grid) tiles per launch: every workgroup keeps its own tiles' coordinates and pixel masks in LDS. */
int cvpce_conv3x3_halo_masked(const void* in, const void* wgt, const float* bias, const unsigned char* mask,
                              const int* tile_map, int n_tiles, void* out, int N, int H, int W, int Cin, int Cout, int K_pad,
                              int Cout_pad, int relu, void* stream);

/* cvpce_conv3x3_halo_masked for the head's TWO towers in one launch (torchvision RetinaNetHead: classification_head.conv and
 * regression_head.conv are two independent chains of four 3x3 256 -> 256 convs + ReLU over the same features; reached from
 * cvpce/models/proposals.py:166).  Layer i of both towers = one conv with Cout = 256 x towers whose cout tile t is tower t:
 * wgt / bias = the towers' layer-i weights concatenated along Cout (fragment-major halo layout of the [Cout][9 Cin] matrix);
 * in_paired = 0: `in` is ONE atlas [N][H][W][Cin] every tower reads (the first layer); in_paired = 1: `in` is tower-major
 * [towers][N][H][W][Cin], tile t reads part t (the later layers).  `out` is ALWAYS tower-major [towers][N][H][W][256]: part t is
 * tower t's layer-i output, a valid atlas for the next paired launch and contiguous for the tower's output conv.  One launch has
 * towers x the tiles of a per-tower launch on the same persistent grid: 568 tiles in three rounds instead of 2 x (284 in two). */
int cvpce_conv3x3_halo_masked_paired(const void* in, const void* wgt, const float* bias, const unsigned char* mask,
                                     const int* tile_map, int n_tiles, void* out, int N, int H, int W, int Cin, int Cout, int K_pad,
                                     int Cout_pad, int relu, int in_paired, void* stream);

/* One ResNet-50 bottleneck block (torchvision `Bottleneck`, v1.5, STRIDE 1, FrozenBatchNorm2d folded into weights and biases;
 * reached from cvpce/models/proposals.py:202-216) in one launch, the two P-channel intermediates held in LDS:
 *   out = relu( conv1x1_w3( relu( conv3x3_w2( relu( conv1x1_w1(x) + b1 ) ) + b2 ) ) + b3 + res )
 * x: [N][H][W][Cin] (Cin % 64 == 0), res / out: [N][H][W][4P] (res = x for the identity blocks, else the projection
 * shortcut's output), P in {64, 128, 256} (layer1 / layer2 / layer3).  Weights in the layouts of cvpce_conv2d_nhwc_bf16:
 * w1 [c1_pad][k1_pad] (k = ci), w2 [c2_pad][k2_pad = 9P] chunk-major, w3 [c3_pad][k3_pad] (k = ci); biases fp32, required.
 * Same rounding points as the three separate launches (both intermediates are rounded to the storage type). */
int cvpce_bottleneck_fused(const void* x, const void* res, const void* w1, const float* b1, const void* w2, const float* b2,
                           const void* w3, const float* b3, void* out, int N, int H, int W, int Cin, int P, int k1_pad, int k2_pad,
                           int k3_pad, int c1_pad, int c2_pad, int c3_pad, void* stream);

/* The same block with FRAGMENT-MAJOR weights (round 5): every MFMA weight fragment the kernel loads is one contiguous KiB (lane
 * L = 16 lq + l16 at byte 16 L, 8 consecutive k per lane) -- read from the row-major tensors a fragment is 16 rows x 64 bytes, which the
 * texture addresser serves one lane per clock.  No padding rows or columns:
 *   w1 [Cin/32 K-steps ks][P/16 cout blocks b][lane][8]:  cout = 32 (b >> 1) + 8 (l16 >> 2) + (l16 & 3) + 4 (b & 1),  k = 32 ks + 8 lq + e
 *   w2 [P/CW groups cg][6 P/64 steps s = 6 c64 + 2 kw + half][kh][CW/16 blocks h][lane][8]:  CW = 16 at P = 64, else 32;
 *        cout = CW cg + (CW == 32 ? 8 (l16 >> 2) + (l16 & 3) + 4 h : l16),  k = ((3 c64 + kh) 3 + kw) 64 + 32 half + 8 lq + e (chunk-major K)
 *   w3 [4P/32 groups g][P/32 K-steps ks][2 blocks h][lane][8]:  cout = 32 g + 8 (l16 >> 2) + (l16 & 3) + 4 h,  k = 32 ks + 8 lq + e
 * (writer: cvpce_amd/ops.py pack_bottleneck_weights).  Results are bit-identical to cvpce_bottleneck_fused. */
int cvpce_bottleneck_fused_fm(const void* x, const void* res, const void* w1, const float* b1, const void* w2, const float* b2,
                              const void* w3, const float* b3, void* out, int N, int H, int W, int Cin, int P, void* stream);

/* Upper bound on the workgroups the persistent convolution kernels launch (default 256 = one per CU).  A host that
 * runs them on a stream restricted to fewer CUs (hipExtStreamCreateWithCUMask) sets the bound to that CU count.
 * Process-wide; 1 <= n <= 256. */
int cvpce_set_persistent_workgroups(int n);

/* nn.MaxPool2d (VGG 2x2 s2; ResNet stem 3x3 s2 p1), NHWC bf16 */
int cvpce_maxpool2d_nhwc_bf16(const void* in, void* out, int N, int H, int W, int C, int k, int stride, int pad,
                              int Ho, int Wo, void* stream);
/* F.relu on a bf16 buffer (LastLevelP6P7: p7(F.relu(p6))) */
int cvpce_relu_bf16(const void* in, void* out, long long n, void* stream);
/* The last two layers of GaussianSubnet in one pass (cvpce/models/proposals.py:96-107: conv1x1 16 -> 16 + ReLU, conv1x1 16 -> 1 +
 * ReLU (act 1) | Tanh (act 2)): x [npix][16] bf16, w2 [>= 16 rows][k2_pad] bf16 (k < 16 used), w3 [>= 1 row][..] bf16 (k < 16 of row 0),
 * b2 [16] / b3 [1] f32 or NULL -> out [npix] f32.  The hidden layer is rounded to bf16 as the two-launch form stores it. */
int cvpce_gauss_tail_bf16(const void* x, const void* w2, const float* b2, const void* w3, const float* b3, float* out, long long npix,
                          int k2_pad, int act, void* stream);
/* The WHOLE GaussianSubnet in one launch (cvpce/models/proposals.py:81-107, called from GaussianLayerNetwork at :139 on the output of
 * GaussianLayer.forward, :73-79): up2(x) -> conv3x3 64->32 + ReLU -> conv3x3 32->32 + ReLU -> conv3x3 32->16 + ReLU -> conv1x1 16->16 + ReLU
 * -> conv1x1 16->1 + ReLU (act 1) | Tanh (act 2) | nothing (act 0).  x: [N][H/2][W/2][64] bf16, read through its nearest-2x upsample
 * (`self.up`, :79, never materialised); H, W: the OUTPUT size (even).  w1 [>= 32 rows][576], w2 [>= 32][288], w3 [>= 16][288]: row-major
 * 3x3 weights as for cvpce_conv2d_nhwc_bf16 (k = (kh * 3 + kw) * Cin + ci); w4 [>= 16 rows][k4_pad] (k < 16 used), w5 [>= 1 row][k5_pad]
 * (k < 16 of row 0); b1 [32] b2 [32] b3 [16] b4 [16] b5 [1] f32 or NULL -> out [N][H][W] f32.  Every intermediate layer is rounded to bf16
 * exactly where the one-launch-per-layer form stores it; none of them reaches memory.  Replaces three cvpce_conv3x3_thin_bf16 launches and
 * cvpce_gauss_tail_bf16. */
int cvpce_gauss_subnet_bf16(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                            const void* w4, const float* b4, int k4_pad, const void* w5, const float* b5, int k5_pad, float* out, int N, int H,
                            int W, int act, void* stream);
/* x.amax(dim=(-2,-1)) -> out[n*out_stride + out_off + c]  (classification.py:46-49) */
int cvpce_global_max_nhwc_bf16(const void* in, float* out, int N, int HW, int C, int out_stride, int out_off,
                               void* stream);
/* Level atlas <-> per-level tensors in one launch: the RetinaNet head (torchvision RetinaNetHead loops over the pyramid levels
 * with shared weights; reached from cvpce/models/proposals.py:166) runs on ONE zero-separated atlas of the L <= 8 levels.
 * levels[l]: [N][h[l]][w[l]][bytes_per_pixel]; atlas: [N][hc][wc][bytes_per_pixel] with level l at rows oy[l].., columns
 * ox[l]..; to_atlas = 1 copies the levels in, 0 copies them out.  bytes_per_pixel % 4 == 0.  Host arrays are read at the call. */
int cvpce_atlas_copy(void* const* levels, const int* h, const int* w, const int* oy, const int* ox, int L, int N,
                     void* atlas, int hc, int wc, int bytes_per_pixel, int to_atlas, void* stream);

/* desc / norm(desc).clamp(min=eps) (classification.py:51); out_bf16 optional */
int cvpce_l2_normalize_f32(const float* in, float* out, void* out_bf16, int B, int D, float eps, void* stream);

/* GeneralizedRCNNTransform for one image (normalise, bilinear resize to (h,w), zero pad to
 * (Hp,Wp)) -> NHWC8 bf16 slot of the batch.  img: f32 CHW [3][H0][W0].  mean3/std3 [host]. */
int cvpce_gln_transform(const float* img, void* out_nhwc8, int H0, int W0, int h, int w, int Hp, int Wp,
                        const float* mean3, const float* std3, void* stream);

/* The same for a whole batch in one launch: imgs / H0 / W0 / h / w are [host] arrays of n entries (read at the call), image i
 * goes to slot i of out_nhwc8 [n][Hp][Wp][8].  Bit-identical to n calls of cvpce_gln_transform. */
int cvpce_gln_transform_batch(const float* const* imgs, const int* H0, const int* W0, const int* h, const int* w, int n,
                              void* out_nhwc8, int Hp, int Wp, const float* mean3, const float* std3, void* stream);

/* production.py:20 + datautils.py:234-239: crop boxes (xyxy f32, truncated like .to(long)) from the
 * original image, pad to square with 0.5, bilinear resize to SxS.  Boxes p >= *count_dev are skipped
 * (count_dev may be NULL).  mode 0: f32 NCHW in [0,1]; mode 1: NHWC8 bf16 with scale_to_tanh
 * (utils.py:280) and the MACVGG normalisation (classification.py:41-44) fused; mode 2: the same as NHWC4 bf16 (8-byte
 * pixels, channel 3 zero: the input layout of cvpce_vgg_stem_fused with in_cstride = 4).  mean3/std3 [host]. */
int cvpce_crop_resize(const float* img, const float* boxes, const int* count_dev, int max_boxes, void* out,
                      int H0, int W0, int S, int mode, const float* mean3, const float* std3, void* stream);
/* (B,3,S,S) f32 NCHW -> NHWC8 bf16: optional x*2-1, then (x-mean)/std (Classifier.classify / build_index input) */
int cvpce_pack_embed_input(const float* in, void* out_nhwc8, int B, int S, int to_tanh, const float* mean3,
                           const float* std3, void* stream);

/* ---- constant-padding tile skipping of the embedder ----------------------------------------------------------------------
 * datautils.py:232-239 (`resize_for_classification`) pads every crop to a square with the constant 0.5, top-left anchored,
 * before the resize to SxS: the part of a crop below / right of the box content is the same constant in every crop, and so is
 * every conv output whose receptive field lies inside it (classification.py:38-51 is a stack of 3x3 convs, ReLUs and 2x2
 * pools).  A pass of the embedder therefore carries one extra image -- the all-padding CONSTANT CROP, always the LAST image
 * (index N - 1) of every tensor -- and the *_list entry points below (i) compute only the tiles of a work list and (ii) read an
 * input pixel that lies in the constant region of its crop from the constant crop instead (it may never have been written).
 * Results are bit-identical to the plain entry points: a skipped pixel would have been computed by the same kernel at the
 * same tile position from the same operand bits as the constant crop's pixel.
 *
 * cvpce_crop_extents: ext_out[p] = (rows, cols) int32 pair per box, the content extent of crop p at the crop resolution S:
 * cvpce_crop_resize writes EXACTLY the pad constant to every pixel with oy >= rows or ox >= cols.  p >= *count_dev: (S, S).
 * boxes_per_image > 0: `boxes` holds the slots of several images of ONE size (boxes_per_image each) and count_dev one count per
 * image -- the whole batch in one launch. */
int cvpce_crop_extents(const float* boxes, const int* count_dev, int max_boxes, int boxes_per_image, int H0, int W0, int S,
                       int* ext_out, void* stream);
/* cvpce_crop_resize (modes 1 / 2, even S) writing the crops' CONTENT only: pixels with oy >= ext[p].rows or ox >= ext[p].cols
 * (`ext` = cvpce_crop_extents of the same boxes, launched before) are left UNWRITTEN.  Input for the *_list entry points only,
 * which read those pixels from the constant crop (ii above); the written pixels are bit-identical to cvpce_crop_resize's. */
int cvpce_crop_resize_content(const float* img, const float* boxes, const int* count_dev, int max_boxes, void* out,
                              int H0, int W0, int S, int mode, const float* mean3, const float* std3, const int* ext, void* stream);
/* The same extents read off crops that already exist as (B,3,S,S) f32 tensors (the input of Classifier.classify,
 * production.py:57-74): every pixel with y >= rows or x >= cols equals `pad` (0.5) in all three channels.  Data-driven: a tensor
 * without constant borders gets (S, S). */
int cvpce_pad_extents(const float* in, int B, int S, float pad, int* ext_out, void* stream);
/* One layer of the pass for cvpce_embed_worklists.  The pass is a chain of ops on the crop -- 3x3 convs (a crop's content extent
 * grows by 1) and 2x2 pools (it halves, upwards) -- given as `pool_mask` (bit i set: op i is a pool); a tensor is named by the
 * number of ops before it, and is constant on rows >= e_y / columns >= e_x, e = min(size, the crop's extent through those ops). */
typedef struct {
    int H, W;                  /* the layer's OUTPUT tensor (after a fused pool) */
    int tile_h, tile_w;        /* the kernel's tile in output-tensor pixels (halo2 16x16, pooled 8x8; halo3 16x32, pooled 8x16; stem 8x8) */
    int out_ops;               /* ops of the pass up to and including this layer (= the name of its output tensor) */
    int in_H, in_W;            /* the layer's INPUT tensor */
    int in_ops;                /* ... and its name */
    int skip;                  /* 0: list every tile; 1: leave out the tiles that are wholly constant; 2: ... and cut the listed tiles at their
                                  last non-constant row (`rows`); 3: ... and move the tiles with rows == 4 to the layer's STRIP list */
} cvpce_skip_layer;
/* ext0 [n_images - 1][2] (cvpce_crop_extents; the constant crop, image n_images - 1, is implied), layers [host] ->
 * lists[l * list_stride + i] for i < counts[l] (device): the tiles layer l computes, crop-major,
 *   entry = ((rows << 24 | ey_in << 12 | ex_in) << 32) | (n << 16) | (ty << 8) | tx
 * (ey_in / ex_in < 4096: the INPUT tensor's extents of crop n; rows in {4, 8, 12, 16}: the tile's conv-output rows that are not
 * wholly constant, rounded up to 4 -- the halo kernels compute only those, the rest of the tile is neither computed nor stored)
 * list_stride >= n_images * tiles of the largest layer.  No host synchronisation: the kernels read counts[l] themselves.
 * `lists` has 2 * n_layers rows: row n_layers + l is layer l's STRIP list (skip == 3: its tiles with rows == 4, same entry format,
 * for cvpce_conv3x3_halo_strips; empty otherwise).
 * counts has 3 * n_layers entries: counts[n_layers + l] = the MFMA work layer l performs, in sixteenths of a full tile (a tile cut
 * at `rows` < 16 streams rows + 1 of its 16 row groups, a strip counts 4) -- what `roofline` counts as executed FLOPs;
 * counts[2 * n_layers + l] = entries of layer l's strip list.
 * computed (optional, [n_layers][n_images][2]): per layer and crop, (conv-output rows computed, tile columns listed) -- the region
 * cvpce_mac_init has to cover from the constant crop. */
int cvpce_embed_worklists(const int* ext0, int n_images, int S, unsigned pool_mask, const cvpce_skip_layer* layers, int n_layers,
                          unsigned long long* lists, long long list_stride, int* counts, int* computed, void* stream);
/* Start values of the MAC descriptor (classification.py:46-49 `amax` over the map) for a layer that runs over a work list WITH tiles /
 * rows left out (`cvpce_skip_layer.skip` != 0 on a layer launched with mac != NULL): desc[n][desc_off + c] = max over the part of
 * crop n's map that the list leaves to the constant crop, from the row / column suffix maxima of the constant crop's own map
 * (f32 [Hc + 1][C] / [Wc + 1][C], entry r = max over rows >= r, last entry 0), indexed by `computed` of that layer. */
int cvpce_mac_init(float* desc, int n_images, int desc_stride, int desc_off, int C, const float* row_suffix_max,
                   const float* col_suffix_max, int Hc, int Wc, const int* computed, void* stream);
/* cvpce_vgg_stem_fused over a work list (tile = 8x8 output pixels).  `in` holds images 0 .. N-2, `const_in` (one image in the
 * same layout) is read as image N - 1. */
int cvpce_vgg_stem_fused_list(const void* in_nhwc, int in_cstride, const void* const_in, const void* w1, const float* b1,
                              const void* w2, const float* b2, void* out, int N, int H, int W,
                              const unsigned long long* list, const int* count_dev, void* stream);
/* cvpce_conv3x3_halo / cvpce_conv3x3_halo_mac (mac != NULL: relu, MAC descriptor in the epilogue; out may then be NULL) over a
 * work list; input pixels in the constant region of their crop are read from image N - 1.  Cout <= 128 runs the wide-tile
 * kernel, mac needs Cout > 128.  With mac the maximum is taken over the listed tiles' computed rows only: a list that leaves
 * tiles / rows out needs `mac` started by cvpce_mac_init (a list of every tile: zeros, as for cvpce_conv3x3_halo_mac). */
int cvpce_conv3x3_halo_list(const void* in, const void* wgt, const float* bias, void* out, float* mac, int mac_stride,
                            int mac_off, int N, int H, int W, int Cin, int Cout, int K_pad, int Cout_pad, int relu,
                            int fuse_pool2, const unsigned long long* list, const int* count_dev, void* stream);

/* RetinaNet.postprocess_detections + batched_nms + transform.postprocess (torchvision 0.9) and the
 * confidence-prefix count of production.py:15.  logits/regs/gh/gw/stride_* are [host] arrays of L
 * entries; logits[l] -> f32 [N][gh*gw*A*K], regs[l] -> f32 [N][gh*gw*A][4] (NHWC conv outputs).
 * base_anchors f32 [L][A][4]; image_hw int [N][2] resized sizes; ratios f32 [N][2] (orig/resized h,w).
 * Outputs: boxes [N][dpi][4] (original pixels), scores [N][dpi], labels i64 [N][dpi], count [N],
 * conf_count [N] = #scores > conf_thresh.  The workspace (cvpce_detect_workspace_bytes: candidates, sorted candidates,
 * the IoU bit matrix, N * L * 12 chunk runs of 8 KiB for the chunked top-k) is scratch: nothing is kept across calls.
 * Ordering: by fp32 logit (a monotone refinement of the score order), then the lower index (csrc/detect.hip). */
size_t cvpce_detect_workspace_bytes(int N, int L, int topk);
int cvpce_detect_postprocess(const float* const* logits, const float* const* regs, const int* gh, const int* gw,
                             const int* stride_h, const int* stride_w, const float* base_anchors,
                             const int* image_hw, const float* ratios, int L, int N, int A, int K, int topk,
                             float score_thresh, float nms_thresh, float xform_clip, int detections_per_img,
                             float conf_thresh, void* workspace, size_t workspace_bytes, float* out_boxes,
                             float* out_scores, long long* out_labels, int* out_count, int* out_conf_count,
                             void* stream);

/* classification.py:87-95: nearest_neighbors(anchors=gallery, queries, k) -> (Q,k) int64, ascending
 * cosine distance, ties to the lower index.  Rows bf16 (is_f32 = 0) or f32 (is_f32 = 1), D % 64 == 0;
 * norms from cvpce_row_norms (already clamped at eps). */
int cvpce_row_norms(const void* x, float* out, int rows, int D, int is_f32, float eps, void* stream);
size_t cvpce_match_workspace_bytes(int Qn, int Gn, int k);
int cvpce_match_topk(const void* queries, const void* gallery, const float* q_norms, const float* g_norms,
                     int Qn, int Gn, int D, int k, int is_f32, void* workspace, size_t workspace_bytes,
                     long long* out_idx, float* out_dist, void* stream);
/* The same with a caller-owned STATE block (cvpce_match_state_bytes(max_queries) bytes, 8-byte aligned, initialised ONCE by
 * cvpce_match_state_init and restored by every call that uses it): with k = 1, bf16 rows and Qn <= max_queries the whole search is ONE
 * launch -- per-query 64-bit (distance, row) keys combined by device-scope atomic minima, results written by the last workgroup to
 * finish -- instead of a GEMM launch + a merge launch.  One state block per launch that may be in flight at a time (two streams: two
 * blocks).  state = NULL, k > 1, f32 rows or a query count beyond the block: exactly cvpce_match_topk.  Identical results either way.
 * OPT-IN (cvpce_match_set_core(..., one_launch = 1) or CVPCE_MATCH_FUSED=1): measured slower than the two launches at every size
 * (200 x 10 000 x 512: 10.4 us against 9.6 us), so by default this entry point IS cvpce_match_topk. */
size_t cvpce_match_state_bytes(int max_queries);
int cvpce_match_state_init(void* state, size_t state_bytes, void* stream);
int cvpce_match_topk_state(const void* queries, const void* gallery, const float* q_norms, const float* g_norms,
                           int Qn, int Gn, int D, int k, int is_f32, void* workspace, size_t workspace_bytes,
                           void* state, size_t state_bytes, long long* out_idx, float* out_dist, void* stream);
/* Test / measurement switch of the bf16 path (process-wide; results never depend on it: every bf16 kernel forms identical
 * distances).  core 0 = choose per launch by estimated time (default), 1 = the 128-row register-staged kernel, 2 = the 256-row
 * LDS-DMA kernel; nq = 0 | 2..5 pins that kernel's query tile to 64 nq rows, mg = 0 | 1 | 2 its gallery tile to 128 mg rows
 * (mg = 1 exists for nq <= 4); one_launch = 0 keeps cvpce_match_topk_state on the two-launch form. */
int cvpce_match_set_core(int core, int nq, int mg, int one_launch);

/* 3x3 / stride 1 / pad 1 convolution with FEW, arbitrarily many output channels and fp32 output, no activation: torchvision
 * RetinaNetHead's `cls_logits` (256 -> A * K = 9) and `bbox_reg` (256 -> 4 A = 36) output convs, reached from
 * cvpce/models/proposals.py:162-168.  in [N][H][W][Cin] (Cin % 64 == 0), wgt in the halo weight layout (Cout_pad % 256 == 0, rows
 * beyond Cout zero), out [N][H][W][Cout] float, 1 <= Cout <= 128.  The 16 x 32-pixel halo-patch kernel of cvpce_conv3x3_halo_wide with
 * only the waves that hold a real cout computing. */
int cvpce_conv3x3_halo_thin_out(const void* in, const void* wgt, const float* bias, float* out, int N, int H, int W,
                                int Cin, int Cout, int K_pad, int Cout_pad, void* stream);

/* cvpce_conv3x3_halo_list's companion for a layer's STRIP list (Cout > 128 only): three listed tiles of which only the first 4
 * output rows are not constant are computed as ONE tile -- patch rows 6 s .. 6 s + 5 and accumulator rows 4 s .. 4 s + 3 belong to
 * strip s, all three share every weight fragment.  Same operands as cvpce_conv3x3_halo_list; writes the strips' rows into the same
 * `out` (and `mac`) the list launch of the layer writes the other tiles into. */
int cvpce_conv3x3_halo_strips(const void* in, const void* wgt, const float* bias, void* out, float* mac, int mac_stride,
                              int mac_off, int N, int H, int W, int Cin, int Cout, int K_pad, int Cout_pad, int relu,
                              int fuse_pool2, const unsigned long long* strip_list, const int* count_dev, void* stream);

/* ---- fp16 twins: the detector's opt-in accuracy mode ------------------------------------------------------------------
 * `gln(..., precision='fp16')` / `GaussianLayerNetwork.set_precision('fp16')` stores the detector's weights and inter-layer
 * activations as IEEE fp16 instead of bf16 (10 instead of 7 mantissa bits; v_mfma_f32_*_f16 runs at the bf16 rate).  Each
 * function below has exactly the contract of the function it is named after (same layouts, same epilogues, fp32
 * accumulation and fp32 head / gaussian outputs), with every "bf16" operand read as fp16; stores saturate at +-65504.
 * Reference semantics are unchanged: nn.Conv2d / FrozenBatchNorm2d / MaxPool2d / F.relu of torchvision's RetinaNet as built by
 * cvpce/models/proposals.py:109-139,162-168. */
int cvpce_conv2d_nhwc_f16(const void* in, const void* wgt, const float* bias, const void* res, void* out,
                          int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                          int Ho, int Wo, int K_pad, int Cout_pad, int act, int out_f32, int in_up_shift,
                          int res_mode, int Hr, int Wr, int fuse_pool2, int force_generic, void* stream);
int cvpce_conv2d_splitk_f16(const void* in, const void* wgt, const float* bias, const void* res, void* out,
                            int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                            int Ho, int Wo, int K_pad, int Cout_pad, int act, int out_f32, int in_up_shift,
                            int res_mode, int Hr, int Wr, int ksplit, void* workspace, size_t workspace_bytes, void* stream);
int cvpce_conv3x3_halo_thin_out_f16(const void* in, const void* wgt, const float* bias, float* out, int N, int H, int W,
                                    int Cin, int Cout, int K_pad, int Cout_pad, void* stream);
int cvpce_bottleneck_fused_fm_f16(const void* x, const void* res, const void* w1, const float* b1, const void* w2, const float* b2,
                                  const void* w3, const float* b3, void* out, int N, int H, int W, int Cin, int P, void* stream);
int cvpce_conv1x1_nhwc_f16(const void* in, const void* wgt, const float* bias, const void* res, void* out, int N, int H,
                           int W, int Cin, int Cout, int stride, int Ho, int Wo, int K_pad, int Cout_pad, int relu,
                           int res_mode, int Hr, int Wr, void* stream);
int cvpce_gln_stem_fused_f16(const void* in_nhwc8, const void* w_frag, const float* bias, void* out, int N, int H, int W,
                             void* stream);
int cvpce_conv3x3_halo_f16(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W, int Cin,
                           int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, void* stream);
int cvpce_conv3x3_halo_wide_f16(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W, int Cin,
                                int Cout, int K_pad, int Cout_pad, int relu, int fuse_pool2, void* stream);
int cvpce_conv3x3_halo_masked_paired_f16(const void* in, const void* wgt, const float* bias, const unsigned char* mask,
                                         const int* tile_map, int n_tiles, void* out, int N, int H, int W, int Cin, int Cout, int K_pad,
                                         int Cout_pad, int relu, int in_paired, void* stream);
int cvpce_conv3x3_halo_masked_f16(const void* in, const void* wgt, const float* bias, const unsigned char* mask,
                                  const int* tile_map, int n_tiles, void* out, int N, int H, int W, int Cin, int Cout, int K_pad,
                                  int Cout_pad, int relu, void* stream);
int cvpce_bottleneck_fused_f16(const void* x, const void* res, const void* w1, const float* b1, const void* w2, const float* b2,
                               const void* w3, const float* b3, void* out, int N, int H, int W, int Cin, int P, int k1_pad, int k2_pad,
                               int k3_pad, int c1_pad, int c2_pad, int c3_pad, void* stream);
int cvpce_maxpool2d_nhwc_f16(const void* in, void* out, int N, int H, int W, int C, int k, int stride, int pad,
                             int Ho, int Wo, void* stream);
int cvpce_relu_f16(const void* in, void* out, long long n, void* stream);
int cvpce_conv3x3_thin_f16(const void* in, const void* wgt, const float* bias, void* out, int N, int H, int W, int Cin, int Cout,
                           int K_pad, int Cout_pad, int relu, int in_up_shift, void* stream);
int cvpce_gauss_subnet_f16(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                           const void* w4, const float* b4, int k4_pad, const void* w5, const float* b5, int k5_pad, float* out, int N, int H,
                           int W, int act, void* stream);
int cvpce_gauss_tail_f16(const void* x, const void* w2, const float* b2, const void* w3, const float* b3, float* out, long long npix,
                         int k2_pad, int act, void* stream);
int cvpce_gln_transform_f16(const float* img, void* out_nhwc8, int H0, int W0, int h, int w, int Hp, int Wp,
                            const float* mean3, const float* std3, void* stream);
int cvpce_gln_transform_batch_f16(const float* const* imgs, const int* H0, const int* W0, const int* h, const int* w, int n,
                                  void* out_nhwc8, int Hp, int Wp, const float* mean3, const float* std3, void* stream);

/* Calibration probe (not on the hot path; bench.py `measured_peaks`): a bare bf16 MFMA loop on register operands --
 * shape 0 = v_mfma_f32_32x32x16_bf16, 1 = v_mfma_f32_16x16x32_bf16; `workgroups` x 4 waves (one per SIMD) each issue
 * iters x 16 MFMAs on 4 x 4 independent accumulators.  operands: >= 128 KiB of random bf16; sink: workgroups * 256 floats.
 * FLOPs = workgroups * 4 * iters * 16 * F with F = 2*32*32*16 = 32768 per MFMA for shape 0 and 2*16*16*32 = 16384 for shape 1. */
int cvpce_probe_mfma_bf16(int shape, int iters, const void* operands, float* sink, int workgroups, void* stream);

/* Calibration probe (bench.py `measured_peaks`): `workgroups` x 512 threads each walk the same L2-resident buffer (bytes: a
 * multiple of 64 KiB, < 4 GiB) `iters` times with 16-byte buffer loads, 8 in flight per lane -- the rate at which the halo
 * kernels can stream weights from L2 into registers.  Bytes moved = workgroups * iters * bytes.  sink: workgroups * 512 floats. */
int cvpce_probe_l2_stream(const void* buf, long long bytes, int iters, float* sink, int workgroups, void* stream);

#ifdef __cplusplus
}
#endif
#endif
