// One ResNet bottleneck block in ONE launch:  out = relu( W3 . relu( conv3x3( relu(W1 . x + b1) ) + b2 ) + b3 + residual )
// -- torchvision resnet50 `Bottleneck` (v1.5, stride 1) with FrozenBatchNorm2d folded into the weights, the block the
// detector's body is made of (reached from cvpce/models/proposals.py:202-216 `resnet_fpn_backbone`).  Instantiated for
// P = 64 / 128 / 256 (layer1 / layer2 / layer3; layer4's 512-wide intermediates do not fit the LDS) and verified for all
// three; the detector USES it for layer1 and (since the fragment-major weights of round 5) layer2 (cvpce_amd/ops.py
// FUSED_BOTTLENECK_MAX_PLANES): at P = 256 a tile is a ~110 us serial chain and a 50x50 map has too few tiles to hide it.
//
// Why: as three launches the block moves its two P-channel intermediates through HBM (and pays three launch latencies on
// the detector's critical chain: at 1-8 images per batch these launches are 15-100 us each and HBM- or latency-bound,
// profiles/r03_detector_timeline.md; layer1 per 8-image block: 454 MB of HBM traffic instead of 656 MB, 173 vs 202-206 us).  Here a workgroup (8 waves) owns a 14x14-pixel output tile:
//   stage A  mid1[16x16 halo pixels][P] = relu(W1 . x + b1), zero outside the image (it is the 3x3's zero padding);
//            both MFMA operands straight from global memory in fragment layout (weights: L2 hits; x: the tile's pixels
//            once), result as bf16/fp16 into LDS
//   stage B  mid2[14 rows x 16][P] = relu(conv3x3(mid1) + b2): weights from L2 into registers, mid1 rows streamed from LDS
//            ONCE per (kw, K-half) and used for the three kh taps (the row streaming of conv3x3_halo2.hip);
//            accumulators stay in registers until every wave has finished reading mid1, then mid2 overwrites it in LDS
//   stage C  out[14x14][4P] = relu(W3 . mid2 + b3 + residual): weights from L2, mid2 from LDS, 16-byte stores.
// Rounding points are those of the three-launch schedule (mid1 and mid2 are rounded to the storage type), so the result
// differs from it only by fp32 summation order.  A pixel block of 16 MFMA columns is one halo / output ROW (16 wide: 14
// valid output columns + 2 that are computed and dropped).
#include "common.h"
#include "../../include/cvpce_amd.h"

namespace {

constexpr int BT = 14;            // output tile edge
constexpr int BH = 16;            // halo tile edge (= pixels per MFMA column block)
#ifndef BN_X_DEPTH
#define BN_X_DEPTH(P) ((P) == 64 ? 4 : 2)   // ring slots of the x fragments in stage A (tools/dev A/B: -DBN_X_DEPTH(P)=2)
#endif
constexpr int BN_M1_PIX = 272;    // mid1 pixels in LDS: 16 x 16 + the wrap-around of the kw-shifted reads of the last row

struct BneckArgs {
    const bf16_t* x;      // [N][H][W][Cin]
    const bf16_t* res;    // [N][H][W][4P]  (x itself when Cin == 4P and the block has no projection shortcut)
    const bf16_t* w1;     // [..][k1_pad]  rows = P couts, k = ci
    const bf16_t* w2;     // [..][k2_pad]  rows = P couts, chunk-major k = ((ci/64*3 + kh)*3 + kw)*64 + ci%64
    const bf16_t* w3;     // [..][k3_pad]  rows = 4P couts, k = ci
    const float *b1, *b2, *b3;
    bf16_t* out;          // [N][H][W][4P]
    int N, H, W, Cin, k1_pad, k2_pad, k3_pad, tiles_x, tiles_y;
    unsigned x_bytes, w1_bytes, w2_bytes, w3_bytes, res_bytes;
};

// 16-byte chunk swizzle of an LDS pixel row of RB bytes: 16 consecutive pixels reading the same logical chunk hit 16
// different 16-byte bank groups (128-byte rows: two pixels share a 256-byte bank row, so the term advances every 2nd pixel)
template <int RB>
__device__ __forceinline__ unsigned lds_swz(unsigned px) { return RB == 128 ? ((px >> 1) & 7u) : (px & 15u); }

// FM (round 5): the three weight tensors are FRAGMENT-MAJOR -- every MFMA weight fragment this kernel loads is one contiguous KiB, lane L's 16
// bytes at byte 16 L (layouts: include/cvpce_amd.h cvpce_bottleneck_fused_fm; writer: cvpce_amd/ops.py pack_bottleneck_weights).  From the
// row-major [Cout][K] tensors a fragment is 16 rows x 64 bytes, no two neighbouring lanes share a 64-byte block, and the texture addresser
// takes such a load one lane per clock (~61 instead of 16 clocks): 94 of the 138 loads a wave issues per tile are weight fragments, and with
// eight waves on one addresser that was most of a tile's 24 us.
template <typename E, int P, bool FM>
__global__ __launch_bounds__(512, 1) void bneck_kernel(BneckArgs a) {
    constexpr int RB = 2 * P;                      // bytes per LDS pixel row (mid1 and mid2 alike)
    constexpr int CB1 = P / 16;                    // 16-cout blocks of mid1
    constexpr int SG1 = CB1 / 4;                   // stage A: sub-groups of 4 cout blocks (64 couts) per K-step
    constexpr int CW = P == 64 ? 16 : 32;          // stage B: couts per wave
    constexpr int NCB = CW / 16;                   //          16-cout blocks per wave
    constexpr int NCG = P / CW;                    //          cout groups
    constexpr int RW = BT / (8 / NCG);             //          output rows per wave (7 or 14)
    constexpr int NG3 = (4 * P / 32) / 8;          // stage C: passes of 32 couts per wave
    static_assert(P == 64 || P == 128 || P == 256, "bottleneck width");
    static_assert(RW * (8 / NCG) == BT, "rows split evenly");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // mid1 [272][P], later mid2 [224][P]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int per_image = a.tiles_x * a.tiles_y;
    const int n = bid / per_image, t = bid - n * per_image;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int y0 = ty * BT, x0 = tx * BT;

    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1, 0, a.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, a.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w3, 0, a.w3_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, a.res_bytes, 0x00020000);

    // row of a 32-cout pair of MFMA blocks that lane m = l16 feeds: MFMA row 4q'+j' of block h is cout 8q' + 4h + j', so that
    // accumulator lane group q ends up with the 8 CONSECUTIVE couts 8q .. 8q+7 of its pixel (one 16-byte chunk)
    const int pair_row0 = 8 * (l16 >> 2) + (l16 & 3);     // + 4h

    // =====================================================================================================================
    // stage A: mid1 = relu(W1 . x + b1) on the 16 x 16 halo tile; wave w owns halo rows 2w, 2w+1 and all P couts
    // =====================================================================================================================
    {
        // x fragments: an MFMA lane (pixel l16, K-quarter lq) needs 16 bytes of ITS pixel -- loaded that way neighbouring lanes are
        // neighbouring pixels 2 Cin bytes apart (one lane per clock in the texture addresser).  Lane 4 p + q loads quarter q of pixel p
        // instead (64-byte runs) and the pieces are exchanged across the wave when a K-step is consumed (4 ds_bpermute per fragment).
        unsigned xoff[2];
        bool xin[2];                  // (of the ACCUMULATOR lane's pixel: the epilogue's zero mask)
        bool xin_t[2];                // (of the pixel this lane LOADS)
        const int xtp = lane >> 2, xtq = lane & 3;
        const int x_to_frag = (4 * l16 + lq) * 4;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int iy = y0 - 1 + 2 * wid + nt, ix = x0 - 1 + l16, ixt = x0 - 1 + xtp;
            xin[nt] = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            xin_t[nt] = (unsigned)iy < (unsigned)a.H && (unsigned)ixt < (unsigned)a.W;
            xoff[nt] = xin_t[nt] ? (unsigned)((((size_t)(n * a.H + iy) * a.W + ixt) * a.Cin + xtq * 8) * 2) : 0xFFFFFFF0u;
        }
        unsigned woff[CB1];
#pragma unroll
        for (int b = 0; b < CB1; ++b) woff[b] = (unsigned)(((32 * (b >> 1) + pair_row0 + 4 * (b & 1)) * a.k1_pad + lq * 8) * 2);

        f32x4 acc[CB1][2];
#pragma unroll
        for (int b = 0; b < CB1; ++b) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b1 + 32 * (b >> 1) + 8 * lq + 4 * (b & 1));
            acc[b][0] = bias;
            acc[b][1] = bias;
        }
        const int nk = a.Cin >> 5;                 // K-steps of 32
        // x comes from HBM (2-3 us under load), the weights from L2: the x fragments run BD - 1 K-steps ahead of the MFMAs (at P = 64
        // the accumulators are small and the registers are there), the weight fragments one work item ahead
        constexpr int BD = BN_X_DEPTH(P);
        bf16x8 A[2][4], B[BD][2];
        // work items (K-step ks, sub-group sg) in order; the weight loads of item i+1 are issued before the MFMAs of item i
#define BN_LOAD_A(SLOT, KS, SG)                                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                       \
            A[SLOT][i_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_w1, FM ? (unsigned)((((KS) * CB1 + 4 * (SG) + i_) * 64 + lane) * 16) : woff[4 * (SG) + i_] + (unsigned)((KS) * 64), 0, 0));
#define BN_LOAD_B(SLOT, KS)                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                       \
            B[SLOT][i_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_x, xin_t[i_] ? xoff[i_] + (unsigned)((KS) * 64) : 0xFFFFFFF0u, 0, 0));
#pragma unroll
        for (int h = 0; h < BD - 1; ++h)
            if (h < nk) BN_LOAD_B(h, h)
        BN_LOAD_A(0, 0, 0)
        for (int ks = 0; ks < nk; ks += BD) {
#pragma unroll
            for (int h = 0; h < BD; ++h) {                     // BD K-steps per trip: the B slot is a constant
                const int k = ks + h;
                if (k < nk) {
                    if (k + BD - 1 < nk) BN_LOAD_B(((h + BD - 1) % BD), k + BD - 1)
                    bf16x8 Bf[2];                                  // this K-step's x fragments in MFMA layout
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const u32x4 raw = __builtin_bit_cast(u32x4, B[h][nt]);
                        u32x4 v;
#pragma unroll
                        for (int d = 0; d < 4; ++d) v[d] = (unsigned)__builtin_amdgcn_ds_bpermute(x_to_frag, (int)raw[d]);
                        Bf[nt] = __builtin_bit_cast(bf16x8, v);
                    }
#pragma unroll
                    for (int sg = 0; sg < SG1; ++sg) {
                        const int slot = (h * SG1 + sg) & 1;
                        if (sg + 1 < SG1) { BN_LOAD_A((slot ^ 1), k, sg + 1) }
                        else if (k + 1 < nk) { BN_LOAD_A((slot ^ 1), k + 1, 0) }
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                acc[4 * sg + i][nt] = E::mfma16(A[slot][i], Bf[nt], acc[4 * sg + i][nt]);
                    }
                }
            }
        }
#undef BN_LOAD_A
#undef BN_LOAD_B
        // epilogue: relu, zero outside the image, 8 consecutive couts per lane and block pair -> one 16-byte LDS chunk
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const unsigned px = (unsigned)((2 * wid + nt) * BH + l16);
            unsigned char* row = smem + px * RB;
#pragma unroll
            for (int g = 0; g < CB1 / 2; ++g) {
                f32x4 lo = acc[2 * g][nt], hi = acc[2 * g + 1][nt];
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo[j] = relu_bits(lo[j]); hi[j] = relu_bits(hi[j]); }
                uint2 l2 = __builtin_bit_cast(uint2, E::pack4(lo)), h2 = __builtin_bit_cast(uint2, E::pack4(hi));
                const unsigned keep = xin[nt] ? 0xFFFFFFFFu : 0u;
                const unsigned chunk = (unsigned)(4 * g + lq);
                *reinterpret_cast<u32x4*>(row + ((chunk ^ lds_swz<RB>(px)) << 4)) = u32x4{l2.x & keep, l2.y & keep, h2.x & keep, h2.y & keep};
            }
        }
        // the wrap-around pixels behind the tile (read by the kw-shifted fragments of output columns 14, 15 only): keep them finite
        if (tid < (BN_M1_PIX - BH * BH) * (RB / 16)) *reinterpret_cast<u32x4*>(smem + BH * BH * RB + tid * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();

    // =====================================================================================================================
    // stage B: mid2 = relu(conv3x3(mid1) + b2); wave (cg, rg) owns couts cg*CW .. +CW-1 and output rows rg*RW .. +RW-1
    // =====================================================================================================================
    const int cg = wid % NCG, rg = wid / NCG;
    const int row0 = rg * RW;
    f32x4 acc2[NCB][RW];
    {
        unsigned w2off[NCB];
#pragma unroll
        for (int h = 0; h < NCB; ++h) {
            const int cout = cg * CW + (NCB == 2 ? pair_row0 + 4 * h : l16);
            w2off[h] = (unsigned)((cout * a.k2_pad + lq * 8) * 2);
            const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b2 + cg * CW + (NCB == 2 ? 8 * lq + 4 * h : 4 * lq));
#pragma unroll
            for (int r = 0; r < RW; ++r) acc2[h][r] = bias;
        }
        bf16x8 A2[2][3][NCB];
        // step s = (c64, kw, half): weights of the three kh taps; K offset of tap (kh, kw), chunk c64, half: ((c64*3 + kh)*3 + kw)*64 + half*32
#define BN_LOAD_A2(SLOT, S)                                                                                    \
        {                                                                                                      \
            const int c64_ = (S) / 6, r6_ = (S) - 6 * c64_, kw_ = r6_ >> 1, hf_ = r6_ & 1;                      \
            _Pragma("unroll") for (int kh_ = 0; kh_ < 3; ++kh_)                                                \
                _Pragma("unroll") for (int h_ = 0; h_ < NCB; ++h_)                                             \
                    A2[SLOT][kh_][h_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(       \
                        srd_w2, FM ? (unsigned)((((((cg * ((P / 64) * 6) + (S)) * 3 + kh_) * NCB + h_) * 64) + lane) * 16)   \
                                   : w2off[h_] + (unsigned)((((c64_ * 3 + kh_) * 3 + kw_) * 64 + hf_ * 32) * 2), 0, 0)); \
        }
        constexpr int NS = (P / 64) * 6;
        BN_LOAD_A2(0, 0)
        for (int s = 0; s < NS; s += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int st = s + par;
                if (st + 1 < NS) BN_LOAD_A2((par ^ 1), st + 1)
                const int c64 = st / 6, r6 = st - 6 * c64, kw = r6 >> 1, hf = r6 & 1;
                const unsigned chunk = (unsigned)(c64 * 8 + hf * 4 + lq);
#pragma unroll
                for (int rr = 0; rr < RW + 2; ++rr) {
                    const unsigned px = (unsigned)((row0 + rr) * BH + l16 + kw);
                    const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(smem + px * RB + ((chunk ^ lds_swz<RB>(px)) << 4));
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
                        if (rr - kh >= 0 && rr - kh < RW) {
#pragma unroll
                            for (int h = 0; h < NCB; ++h) acc2[h][rr - kh] = E::mfma16(A2[par][kh][h], bfr, acc2[h][rr - kh]);
                        }
                    if ((rr & 1) == 1) __builtin_amdgcn_sched_barrier(0);     // at most two row fragments in flight: hoisting all 16 reads spills
                }
            }
        }
#undef BN_LOAD_A2
    }
    __syncthreads();          // every wave is done reading mid1: mid2 takes its place
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const unsigned px = (unsigned)((row0 + r) * BH + l16);
        unsigned char* row = smem + px * RB;
        if (NCB == 2) {
            f32x4 lo = acc2[0][r], hi = acc2[NCB - 1][r];
#pragma unroll
            for (int j = 0; j < 4; ++j) { lo[j] = relu_bits(lo[j]); hi[j] = relu_bits(hi[j]); }
            const uint2 l2 = __builtin_bit_cast(uint2, E::pack4(lo)), h2 = __builtin_bit_cast(uint2, E::pack4(hi));
            const unsigned chunk = (unsigned)(cg * (CW / 8) + lq);
            *reinterpret_cast<u32x4*>(row + ((chunk ^ lds_swz<RB>(px)) << 4)) = u32x4{l2.x, l2.y, h2.x, h2.y};
        } else {
            f32x4 v = acc2[0][r];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = relu_bits(v[j]);
            const unsigned chunk = (unsigned)(cg * (CW / 8) + (lq >> 1));          // couts cg*16 + 4 lq .. +3: half a chunk
            *reinterpret_cast<uint2*>(row + ((chunk ^ lds_swz<RB>(px)) << 4) + 8 * (lq & 1)) = __builtin_bit_cast(uint2, E::pack4(v));
        }
    }
    __syncthreads();

    // =====================================================================================================================
    // stage C: out = relu(W3 . mid2 + b3 + residual); wave w owns the 32-cout groups w, w + 8, ...; all 14 rows
    // =====================================================================================================================
    const int ox = x0 + l16;
    const bool col_ok = l16 < BT && ox < a.W;
    constexpr int HR = BT / 2;                                   // rows per half pass: 2 x 7 accumulator rows + their residuals fit the registers
#pragma unroll 1
    for (int hp = 0; hp < 2 * NG3; ++hp) {
        const int pass = hp >> 1, rbase = (hp & 1) * HR;
        const int c0 = (pass * 8 + wid) * 32;                    // first cout of this pass
        unsigned w3off[2];
        f32x4 acc3[2][HR];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            w3off[h] = (unsigned)(((c0 + pair_row0 + 4 * h) * a.k3_pad + lq * 8) * 2);
            const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b3 + c0 + 8 * lq + 4 * h);
#pragma unroll
            for (int r = 0; r < HR; ++r) acc3[h][r] = bias;
        }
        // residual of this pass's 32 couts at its 7 rows of 16 pixels: in flight under the MFMAs.  An accumulator lane (pixel l16, cout
        // group lq) needs 16 bytes of ITS pixel, and neighbouring lanes are neighbouring pixels 8 P bytes apart -- loaded that way no two
        // neighbouring lanes share a 64-byte block and the texture addresser takes the load one lane per clock (see FM above).  Instead
        // lane 4 p + q loads cout group q of pixel p (four neighbouring lanes = one 64-byte run) and the pieces are exchanged across the
        // wave just before they are added (4 ds_bpermute per row); the stores go the same way round.
        const int tp = lane >> 2, tq = lane & 3;                      // transposed role: pixel tp, cout group tq
        const int tox = x0 + tp;
        const bool tcol_ok = tp < BT && tox < a.W;
        u32x4 resv[HR];
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            const int oy = y0 + rbase + r;
            const bool ok = tcol_ok && oy < a.H;
            const unsigned off = ok ? (unsigned)((((size_t)(n * a.H + oy) * a.W + tox) * (4 * P) + c0 + 8 * tq) * 2) : 0xFFFFFFF0u;
            resv[r] = __builtin_amdgcn_raw_buffer_load_b128(srd_r, off, 0, 0);
        }
        bf16x8 A3[2][2];
#define BN_LOAD_A3(SLOT, KS)                                                                                   \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                       \
            A3[SLOT][h_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_w3, FM ? (unsigned)(((((pass * 8 + wid) * (P / 32) + (KS)) * 2 + h_) * 64 + lane) * 16) : w3off[h_] + (unsigned)((KS) * 64), 0, 0));
        constexpr int NK3 = P / 32;
        BN_LOAD_A3(0, 0)
#pragma unroll
        for (int ks = 0; ks < NK3; ++ks) {
            if (ks + 1 < NK3) BN_LOAD_A3(((ks + 1) & 1), ks + 1)
            const unsigned chunk = (unsigned)(4 * ks + lq);
#pragma unroll
            for (int r = 0; r < HR; ++r) {
                const unsigned px = (unsigned)((rbase + r) * BH + l16);
                const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(smem + px * RB + ((chunk ^ lds_swz<RB>(px)) << 4));
#pragma unroll
                for (int h = 0; h < 2; ++h) acc3[h][r] = E::mfma16(A3[ks & 1][h], bfr, acc3[h][r]);
            }
        }
#undef BN_LOAD_A3
        const int to_acc = (4 * l16 + lq) * 4;                        // byte address of the lane that loaded THIS lane's (pixel l16, group lq) piece
        const int to_run = (16 * tq + tp) * 4;                        // ... of the accumulator lane that holds the piece THIS lane stores
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            const int oy = y0 + rbase + r;
            u32x4 rraw;
#pragma unroll
            for (int d = 0; d < 4; ++d) rraw[d] = (unsigned)__builtin_amdgcn_ds_bpermute(to_acc, (int)resv[r][d]);
            const bf16x8 rv = __builtin_bit_cast(bf16x8, rraw);
            f32x4 lo = acc3[0][r], hi = acc3[1][r];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = relu_bits(lo[j] + E::widen(rv[j]));
                hi[j] = relu_bits(hi[j] + E::widen(rv[4 + j]));
            }
            const uint2 l2 = __builtin_bit_cast(uint2, E::pack4(lo)), h2 = __builtin_bit_cast(uint2, E::pack4(hi));
            u32x4 v;
            v[0] = (unsigned)__builtin_amdgcn_ds_bpermute(to_run, (int)l2.x);
            v[1] = (unsigned)__builtin_amdgcn_ds_bpermute(to_run, (int)l2.y);
            v[2] = (unsigned)__builtin_amdgcn_ds_bpermute(to_run, (int)h2.x);
            v[3] = (unsigned)__builtin_amdgcn_ds_bpermute(to_run, (int)h2.y);
            if (tcol_ok && oy < a.H)
                *reinterpret_cast<u32x4*>(a.out + ((size_t)(n * a.H + oy) * a.W + tox) * (4 * P) + c0 + 8 * tq) = v;
        }
    }
}

template <typename E, int P, bool FM>
int launch_bneck(const BneckArgs& a, hipStream_t stream) {
    const int smem = BN_M1_PIX * 2 * P;
    if (!cvpce_smem_attr_done<bneck_kernel<E, P, FM>>((const void*)bneck_kernel<E, P, FM>, smem)) return CVPCE_ERR_LAUNCH;
    hipLaunchKernelGGL((bneck_kernel<E, P, FM>), dim3((unsigned)(a.N * a.tiles_x * a.tiles_y)), dim3(512), smem, stream, a);
    return cvpce_check_launch();
}

template <typename E, bool FM = false>
int bneck_dispatch(const void* x, const void* res, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3,
                   const float* b3, void* out, int N, int H, int W, int Cin, int P, int k1_pad, int k2_pad, int k3_pad, int c1_pad,
                   int c2_pad, int c3_pad, void* stream) {
    if (N <= 0) return CVPCE_OK;
    if (!x || !res || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !out) return CVPCE_ERR_ARG;
    if (P != 64 && P != 128 && P != 256) return CVPCE_ERR_ARG;
    if (H <= 0 || W <= 0 || Cin <= 0 || Cin % 64 != 0 || k1_pad < Cin || k2_pad != 9 * P || k3_pad < P) return CVPCE_ERR_ARG;
    if (c1_pad < P || c2_pad < P || c3_pad < 4 * P) return CVPCE_ERR_ARG;
    if ((long long)N * H * W * Cin * 2 >= (1LL << 32) || (long long)N * H * W * 4 * P * 2 >= (1LL << 32)) return CVPCE_ERR_ARG;
    BneckArgs a;
    a.x = (const bf16_t*)x; a.res = (const bf16_t*)res; a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.w3 = (const bf16_t*)w3;
    a.b1 = b1; a.b2 = b2; a.b3 = b3; a.out = (bf16_t*)out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.k1_pad = k1_pad; a.k2_pad = k2_pad; a.k3_pad = k3_pad;
    a.tiles_x = (W + BT - 1) / BT; a.tiles_y = (H + BT - 1) / BT;
    if ((long long)N * a.tiles_x * a.tiles_y >= (1LL << 31)) return CVPCE_ERR_ARG;
    a.x_bytes = (unsigned)((long long)N * H * W * Cin * 2);
    a.res_bytes = (unsigned)((long long)N * H * W * 4 * P * 2);
    if (FM) {         // fragment-major tensors hold exactly the P x Cin, P x 9P and 4P x P weights (no padding rows / columns)
        a.w1_bytes = (unsigned)((long long)P * Cin * 2);
        a.w2_bytes = (unsigned)((long long)P * 9 * P * 2);
        a.w3_bytes = (unsigned)((long long)4 * P * P * 2);
    } else {
        a.w1_bytes = (unsigned)((long long)c1_pad * k1_pad * 2);
        a.w2_bytes = (unsigned)((long long)c2_pad * k2_pad * 2);
        a.w3_bytes = (unsigned)((long long)c3_pad * k3_pad * 2);
    }
    hipStream_t s = (hipStream_t)stream;
    switch (P) {
        case 64: return launch_bneck<E, 64, FM>(a, s);
        case 128: return launch_bneck<E, 128, FM>(a, s);
        default: return launch_bneck<E, 256, FM>(a, s);
    }
}

}  // namespace

extern "C" int cvpce_bottleneck_fused(const void* x, const void* res, const void* w1, const float* b1, const void* w2, const float* b2,
                                      const void* w3, const float* b3, void* out, int N, int H, int W, int Cin, int P, int k1_pad,
                                      int k2_pad, int k3_pad, int c1_pad, int c2_pad, int c3_pad, void* stream) {
    return bneck_dispatch<ElemBF16>(x, res, w1, b1, w2, b2, w3, b3, out, N, H, W, Cin, P, k1_pad, k2_pad, k3_pad, c1_pad, c2_pad, c3_pad, stream);
}
extern "C" int cvpce_bottleneck_fused_fm(const void* x, const void* res, const void* w1, const float* b1, const void* w2, const float* b2,
                                         const void* w3, const float* b3, void* out, int N, int H, int W, int Cin, int P, void* stream) {
    return bneck_dispatch<ElemBF16, true>(x, res, w1, b1, w2, b2, w3, b3, out, N, H, W, Cin, P, Cin, 9 * P, P, P, P, 4 * P, stream);
}
extern "C" int cvpce_bottleneck_fused_fm_f16(const void* x, const void* res, const void* w1, const float* b1, const void* w2, const float* b2,
                                             const void* w3, const float* b3, void* out, int N, int H, int W, int Cin, int P, void* stream) {
    return bneck_dispatch<ElemF16, true>(x, res, w1, b1, w2, b2, w3, b3, out, N, H, W, Cin, P, Cin, 9 * P, P, P, P, 4 * P, stream);
}
extern "C" int cvpce_bottleneck_fused_f16(const void* x, const void* res, const void* w1, const float* b1, const void* w2, const float* b2,
                                          const void* w3, const float* b3, void* out, int N, int H, int W, int Cin, int P, int k1_pad,
                                          int k2_pad, int k3_pad, int c1_pad, int c2_pad, int c3_pad, void* stream) {
    return bneck_dispatch<ElemF16>(x, res, w1, b1, w2, b2, w3, b3, out, N, H, W, Cin, P, k1_pad, k2_pad, k3_pad, c1_pad, c2_pad, c3_pad, stream);
}
