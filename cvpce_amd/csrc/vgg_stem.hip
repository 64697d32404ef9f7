// Fused VGG16 stem for the MAC-VGG embedder (torchvision vgg cfg 'D' features[0:5]):
//
//     conv3x3(3->64)+bias+ReLU -> conv3x3(64->64)+bias+ReLU -> MaxPool2d(2,2)
//
// as ONE persistent kernel.  Unfused, conv1_1 writes a 64-channel full-resolution tensor
// (8.4 MB per 256x256 crop, 13.4 GB per 1600-crop batch) that conv1_2 reads straight back --
// together a quarter of the embedder's time at a few % of its FLOPs.  Here every workgroup keeps
// ALL weights of both layers resident in LDS (72 KiB + 6 KiB of the CU's 160 KiB), walks 16x16
// output tiles, and per tile
//   1. evaluates conv1_1 on the 18x18 halo patch straight from the 3-channel input (MFMA, K = 3 rows
//      of 16 = (kw 0..3) x (c 0..3), slot kw=3 / c=3 carry zero weights) into an LDS image
//      [324 px][64 ch] (zero outside the image = conv1_2's zero padding),
//   2. runs conv1_2's 9 taps x 4 K-steps of MFMAs directly out of LDS (no global traffic, no
//      barrier inside the tap loop, fragments double buffered in registers),
//   3. max-pools in registers (pixels are laid out in 2x2-quad order across lanes) and stores only
//      the pooled 8x8x64 tile.
// HBM traffic per crop: 0.5 MB in (NHWC4 bf16) + 2.1 MB out instead of 0.5 + 8.4 + 8.4 + 2.1 MB.
#include "common.h"
#include "../../include/cvpce_amd.h"

// compile-time timing experiments (never set in the shipped library; tools/ablate.sh): 1 skip phase 1 (conv1_1),
// 2 skip phase 2 MFMAs, 4 skip phase 2 fragment reads, 8 skip the output stores, 16 skip the input patch loads
#ifndef CVPCE_DBG
#define CVPCE_DBG 0
#endif

#define ST_T 16                    // output tile edge
#define ST_P1 (ST_T + 2)           // conv1_1 patch edge (halo 1)
#define ST_P0 (ST_T + 4)           // input patch edge (halo 2)
#define ST_NPIX1 (ST_P1 * ST_P1)   // 324
#define ST_ROWS1 352               // 11 MFMA pixel tiles of 32
#define ST_W2_BYTES (9 * 64 * 128)
#define ST_W1_BYTES (64 * 96)
#define ST_A1_BYTES (ST_ROWS1 * 128)
#define ST_IN_BYTES (ST_P0 * ST_P0 * 8 + 64)   // + slack for the kw=3 over-read at the patch end
#define ST_SMEM (ST_W2_BYTES + ST_W1_BYTES + ST_A1_BYTES + 2 * ST_IN_BYTES)

struct StemArgs {
    const bf16_t* in;    // [N][H][W][cstride] bf16, channels 0..2 used, channel 3 must be zero (cstride 4 or 8)
    int cstride;
    const bf16_t* w1;    // [64][48]  k = kh*16 + kw*4 + c
    const float* b1;     // [64]
    const bf16_t* w2;    // [9][64][64]  (tap, cout, cin)
    const float* b2;     // [64]
    bf16_t* out;         // [N][H/2][W/2][64]
    int N, H, W;
    int tiles_x, tiles_y, ntiles;
};

// byte offset of 16-B chunk `chunk` of conv1_1-output patch pixel (py, px).  The XOR makes the conv1_2 fragment reads
// (ds_read_b128, served in the non-contiguous 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}: four 2x2 quads whose
// column pairs are distinct mod 4) conflict-free for every tap: bits 2:1 = column pair mod 4, bit 0 = row parity
// (^ column pair bit 2).  See conv3x3_halo.hip h3_swz.
__device__ __forceinline__ int a1_swz0(int u) { return ((u & 3) << 1) | ((u >> 2) & 1); }
__device__ __forceinline__ int a1_off(int py, int px, int chunk) {
    return (py * ST_P1 + px) * 128 + ((chunk ^ a1_swz0(px >> 1) ^ (py & 1)) << 4);
}

__global__ __launch_bounds__(256, 1) void vgg_stem_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* W2 = smem;
    unsigned char* W1 = W2 + ST_W2_BYTES;
    unsigned char* A1 = W1 + ST_W1_BYTES;
    unsigned char* IN = A1 + ST_A1_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;

    // ---- resident weights -> LDS (W2 rows of 128 B, XOR-swizzled like the GEMM tiles) ----
    for (int i = tid; i < 9 * 64 * 8; i += 256) {
        const int row = i >> 3, ch = i & 7;           // row = tap*64 + cout
        const u32x4 v = *reinterpret_cast<const u32x4*>(a.w2 + (size_t)row * 64 + ch * 8);
        *reinterpret_cast<u32x4*>(W2 + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = v;
    }
    for (int i = tid; i < 64 * 6; i += 256)
        *reinterpret_cast<u32x4*>(W1 + i * 16) = *reinterpret_cast<const u32x4*>(a.w1 + (size_t)i * 8);
    // zero the slack behind both input buffers once (read by the kw=3 slot, multiplied by zero weights)
    if (tid < 16) {
        *reinterpret_cast<unsigned*>(IN + ST_P0 * ST_P0 * 8 + (tid & 15) * 4) = 0u;
        *reinterpret_cast<unsigned*>(IN + ST_IN_BYTES + ST_P0 * ST_P0 * 8 + (tid & 15) * 4) = 0u;
    }

    // input patch staging: 400 pixels of 8 B, two per thread (second one only for tid < 144)
    unsigned long long preg[2];
    auto load_patch = [&](int tile) {
        const int n = tile / (a.tiles_x * a.tiles_y);
        const int r = tile - n * (a.tiles_x * a.tiles_y);
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = tid + k * 256;
            unsigned long long v = 0ull;
            if (p < ST_P0 * ST_P0) {
                const int py = p / ST_P0, px = p - py * ST_P0;
                const int y = ty * ST_T - 2 + py, x = tx * ST_T - 2 + px;
                if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W)
                    v = *reinterpret_cast<const unsigned long long*>(a.in + ((size_t)(n * a.H + y) * a.W + x) * a.cstride);
            }
            preg[k] = v;
        }
    };
    auto store_patch = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = tid + k * 256;
            if (p < ST_P0 * ST_P0) *reinterpret_cast<unsigned long long*>(IN + buf * ST_IN_BYTES + p * 8) = preg[k];
        }
    };

    int tile = blockIdx.x;
    int buf = 0;
    if (tile < a.ntiles) { load_patch(tile); store_patch(0); }
    __syncthreads();

    // per-lane constants of the conv1_2 phase: wave w owns output rows 4w..4w+3 (pixel tiles 2w, 2w+1)
    // lane r of a 32-pixel tile: quad q = r>>2 -> columns 2q,2q+1 ; sub = r&3 -> row +(sub>>1), col +(sub&1)
    const int q = lr >> 2, sub = lr & 3;
    // fragment address of (tap (kh,kw), K-step kk) = (rd2[nt][kw] ^ (((2 kk) ^ (kh & 1)) << 4)) + (kh*18 + kw) * 128:
    // the per-lane part of the swizzle (K half lh, row parity, column pair of ox + kw) is folded into rd2 once
    unsigned rd2[2][3];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int oy = 2 * (2 * wid + nt) + (sub >> 1), ox = 2 * q + (sub & 1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            rd2[nt][kw] = (unsigned)((oy * ST_P1 + ox) * 128 + ((lh ^ (oy & 1) ^ a1_swz0(((ox + kw) >> 1) & 7)) << 4));
    }

    // biases live in registers for the whole persistent loop (a global load per tile would put an
    // L2 round trip on the critical path of both epilogues)
    f32x4 bias1[2][4], bias2[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bias1[ct][g] = *reinterpret_cast<const f32x4*>(a.b1 + ct * 32 + 8 * g + 4 * lh);
            bias2[ct][g] = *reinterpret_cast<const f32x4*>(a.b2 + ct * 32 + 8 * g + 4 * lh);
        }

    // conv1_1's A operand (64 couts x 48 k) is tiny: keep this lane's 6 fragments in registers for the whole kernel
    bf16x8 w1frag[3][2];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
            w1frag[kh][ct] = *reinterpret_cast<const bf16x8*>(W1 + (ct * 32 + lr) * 96 + kh * 32 + lh * 16);

    for (; tile < a.ntiles; tile += gridDim.x) {
        const int n = tile / (a.tiles_x * a.tiles_y);
        const int rem = tile - n * (a.tiles_x * a.tiles_y);
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int next = tile + gridDim.x;
        if (next < a.ntiles && !(CVPCE_DBG & 16)) load_patch(next);          // global loads in flight under both MFMA phases

        // ================= phase 1: conv1_1 on the 18x18 patch -> A1 =================
        const unsigned char* INb = IN + buf * ST_IN_BYTES;
        for (int pt = wid; pt < ((CVPCE_DBG & 1) ? 0 : ST_ROWS1 / 32); pt += 4) {
            int pp = pt * 32 + lr;
            const bool real = pp < ST_NPIX1;
            if (!real) pp = ST_NPIX1 - 1;
            const int py = pp / ST_P1, px = pp - py * ST_P1;
            f32x16 acc[2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
            // B fragments: k = kh*16 + 8h + j  <->  pixels (py+kh, px+2h .. px+2h+1), 4 channels each: 16 contiguous bytes.
            // All three are fetched before the MFMAs; the A fragments (conv1_1 weights) never leave registers.
            union { unsigned long long u[2]; bf16x8 v; } bfrag[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const unsigned char* src = INb + ((py + kh) * ST_P0 + px + 2 * lh) * 8;
                bfrag[kh].u[0] = *reinterpret_cast<const unsigned long long*>(src);
                bfrag[kh].u[1] = *reinterpret_cast<const unsigned long long*>(src + 8);
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1frag[kh][ct], bfrag[kh].v, acc[ct], 0, 0, 0);
            // epilogue: bias + ReLU, zero outside the image (conv1_2 pads conv1_1's OUTPUT with zeros)
            const int y = ty * ST_T - 1 + py, x = tx * ST_T - 1 + px;
            const bool inside = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            if (real) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = ct * 32 + 8 * g + 4 * lh;
                        const f32x4 b = bias1[ct][g];
                        bf16x4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = f32_to_bf16(inside ? fmaxf(acc[ct][4 * g + j] + b[j], 0.f) : 0.f);
                        *reinterpret_cast<bf16x4*>(A1 + a1_off(py, px, co >> 3) + (co & 7) * 2) = o;
                    }
            }
        }
        __syncthreads();

        // ================= phase 2: conv1_2 (9 taps x 4 K-steps) out of LDS =================
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        bf16x8 af[3][2], bfr[3][2];
#define ST_LOAD(S, SLOT)                                                                                       \
        {                                                                                                      \
            const int tap = (S) >> 2, kk = (S) & 3;                                                            \
            const int kh = tap / 3, kw = tap - kh * 3;                                                         \
            const int chunk = kk * 2 + lh;                                                                     \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) {                                                 \
                const int row = tap * 64 + mt * 32 + lr;                                                       \
                af[SLOT][mt] = *reinterpret_cast<const bf16x8*>(W2 + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)); \
            }                                                                                                  \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                   \
                bfr[SLOT][nt] = *reinterpret_cast<const bf16x8*>(A1 + ((rd2[nt][kw] ^ (unsigned)(((2 * kk) ^ (kh & 1)) << 4)) + (unsigned)((kh * ST_P1 + kw) * 128))); \
        }
        // fragments run TWO K-steps ahead of the MFMAs (3 register slots); sched_barrier pins that order --
        // left alone, the scheduler sinks each ds_read group to just before its consumer and exposes the
        // full LDS latency on every K-step
        ST_LOAD(0, 0)
        ST_LOAD(1, 1)
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + 2 < 36 && !(CVPCE_DBG & 4)) ST_LOAD(s + 2, (s + 2) % 3)
            __builtin_amdgcn_sched_barrier(0);
            if (!(CVPCE_DBG & 2)) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s % 3][mt], bfr[s % 3][nt], acc[mt][nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef ST_LOAD
        // epilogue: bias, 2x2 max over the quad's 4 lanes (DPP), ReLU.  After pooling the 4 lanes of a quad hold the
        // same 64 values; lane `sub` keeps channel group g = sub, then a v_permlane32_swap pair gives every lane
        // 8 consecutive channels: ONE 16-byte store per lane per pixel tile, all 64 lanes active.
        const int Ho = a.H >> 1, Wo = a.W >> 1;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            unsigned pk[2][2];     // [mt][dword]: this lane's 4 channels (8 sub + 4 lh ..+3) of cout tile mt, bf16x2 packed
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                bf16x4 sel;
#pragma unroll
                for (int j = 0; j < 4; ++j) sel[j] = (bf16_t)0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = bias2[mt][g];
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f32_to_bf16(fmaxf(quad_max(acc[mt][nt][4 * g + j] + b[j]), 0.f));
                    if (sub == g) sel = o;
                }
                const uint2 u = *reinterpret_cast<const uint2*>(&sel);
                pk[mt][0] = u.x; pk[mt][1] = u.y;
            }
            // lanes 0-31 (lh = 0) end with cout tile 0, channels 8 sub .. 8 sub + 7; lanes 32-63 with cout tile 1
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                auto r = __builtin_amdgcn_permlane32_swap(pk[0][d], pk[1][d], false, false);
                pk[0][d] = r[0]; pk[1][d] = r[1];
            }
            const int oyp = (ty * ST_T) / 2 + (2 * wid + nt), oxp = (tx * ST_T) / 2 + q;
            bf16_t* dst = a.out + ((size_t)(n * Ho + oyp) * Wo + oxp) * 64 + lh * 32 + sub * 8;
            if (!(CVPCE_DBG & 8)) *reinterpret_cast<u32x4*>(dst) = u32x4{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
        }
        if (next < a.ntiles) store_patch(buf ^ 1);
        __syncthreads();      // A1 is free again; the next input patch is visible
        buf ^= 1;
    }
}

extern "C" int cvpce_vgg_stem_fused_1q(const void* in_nhwc, int in_cstride, const void* w1, const float* b1, const void* w2,
                                    const float* b2, void* out, int N, int H, int W, void* stream) {
    if (N <= 0) return CVPCE_OK;
    if (!in_nhwc || !w1 || !b1 || !w2 || !b2 || !out) return CVPCE_ERR_ARG;
    if (in_cstride != 4 && in_cstride != 8) return CVPCE_ERR_ARG;
    if (H % ST_T != 0 || W % ST_T != 0 || H <= 0 || W <= 0) return CVPCE_ERR_ARG;
    if ((long long)N * H * W >= (1LL << 31) / 64) return CVPCE_ERR_ARG;
    StemArgs a;
    a.in = (const bf16_t*)in_nhwc; a.cstride = in_cstride; a.w1 = (const bf16_t*)w1; a.b1 = b1; a.w2 = (const bf16_t*)w2; a.b2 = b2;
    a.out = (bf16_t*)out; a.N = N; a.H = H; a.W = W;
    a.tiles_x = W / ST_T; a.tiles_y = H / ST_T; a.ntiles = N * a.tiles_x * a.tiles_y;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)vgg_stem_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ST_SMEM) != hipSuccess)
            return CVPCE_ERR_LAUNCH;
        attr_set = true;
    }
    int grid = a.ntiles < 256 ? a.ntiles : 256;      // one persistent workgroup per CU
    hipLaunchKernelGGL(vgg_stem_kernel, dim3(grid), dim3(256), ST_SMEM, (hipStream_t)stream, a);
    return cvpce_check_launch();
}
