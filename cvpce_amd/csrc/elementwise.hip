// HBM-bound helper kernels of the hot path (NHWC bf16, 16 B per lane):
//   max-pool (VGG 2x2 s2, ResNet stem 3x3 s2 p1), ReLU, MAC global max over
//   H,W (classification.py:46-49 `amax`), L2 normalisation with clamp
//   (classification.py:51).
#include "common.h"
#include "../../include/cvpce_amd.h"
#include <math.h>

template <typename E>
__device__ __forceinline__ uint4 max_elem8(uint4 a, uint4 b) {
    bf16x8 x = *reinterpret_cast<bf16x8*>(&a), y = *reinterpret_cast<bf16x8*>(&b), r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (E::widen(x[i]) > E::widen(y[i])) ? x[i] : y[i];
    return *reinterpret_cast<uint4*>(&r);
}

template <typename E>
__global__ void maxpool_nhwc_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int N, int H, int W,
                                    int C, int k, int stride, int pad, int Ho, int Wo) {
    const int C8 = C / 8;
    const long long total = (long long)N * Ho * Wo * C8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int c8 = (int)(i % C8);
        long long p = i / C8;
        int ox = (int)(p % Wo);
        p /= Wo;
        int oy = (int)(p % Ho);
        int n = (int)(p / Ho);
        bf16x8 neg;
#pragma unroll
        for (int j = 0; j < 8; ++j) neg[j] = E::kF16 ? __builtin_bit_cast(bf16_t, (f16_t)(-INFINITY)) : (bf16_t)(-INFINITY);
        uint4 acc = *reinterpret_cast<uint4*>(&neg);
        for (int dy = 0; dy < k; ++dy) {
            int iy = oy * stride - pad + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int dx = 0; dx < k; ++dx) {
                int ix = ox * stride - pad + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                uint4 v = *reinterpret_cast<const uint4*>(in + ((size_t)(n * H + iy) * W + ix) * C + c8 * 8);
                acc = max_elem8<E>(acc, v);
            }
        }
        *reinterpret_cast<uint4*>(out + ((size_t)(n * Ho + oy) * Wo + ox) * C + c8 * 8) = acc;
    }
}

template <typename E>
static int maxpool_dispatch(const void* in, void* out, int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, void* stream) {
    if (!in || !out || C % 8 != 0 || k < 1 || stride < 1) return CVPCE_ERR_ARG;
    if (Ho != (H + 2 * pad - k) / stride + 1 || Wo != (W + 2 * pad - k) / stride + 1) return CVPCE_ERR_ARG;
    long long total = (long long)N * Ho * Wo * (C / 8);
    if (total <= 0) return CVPCE_OK;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool_nhwc_kernel<E>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in,
                       (bf16_t*)out, N, H, W, C, k, stride, pad, Ho, Wo);
    return cvpce_check_launch();
}

extern "C" int cvpce_maxpool2d_nhwc_bf16(const void* in, void* out, int N, int H, int W, int C, int k, int stride,
                                         int pad, int Ho, int Wo, void* stream) {
    return maxpool_dispatch<ElemBF16>(in, out, N, H, W, C, k, stride, pad, Ho, Wo, stream);
}
extern "C" int cvpce_maxpool2d_nhwc_f16(const void* in, void* out, int N, int H, int W, int C, int k, int stride,
                                        int pad, int Ho, int Wo, void* stream) {
    return maxpool_dispatch<ElemF16>(in, out, N, H, W, C, k, stride, pad, Ho, Wo, stream);
}

// ReLU on 16-bit sign-magnitude floats (bf16 and fp16 alike): positive values pass, everything else -- negatives, -0, and
// like `(float)x > 0 ? x : 0` NaNs -- becomes +0.
__global__ void relu_bf16_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, long long n8, int is_f16) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        uint4 v = in[i];
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 x = *reinterpret_cast<s16x8*>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int mag = x[j] & 0x7FFF, inf = is_f16 ? 0x7C00 : 0x7F80;      // positive and not NaN
            x[j] = (x[j] > 0 && mag <= inf) ? x[j] : (short)0;
        }
        out[i] = *reinterpret_cast<uint4*>(&x);
    }
}

static int relu_dispatch(const void* in, void* out, long long n, int is_f16, void* stream) {
    if (!in || !out || n % 8 != 0) return CVPCE_ERR_ARG;
    long long n8 = n / 8;
    if (n8 == 0) return CVPCE_OK;
    int blocks = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
    hipLaunchKernelGGL(relu_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, (uint4*)out, n8, is_f16);
    return cvpce_check_launch();
}
extern "C" int cvpce_relu_bf16(const void* in, void* out, long long n, void* stream) { return relu_dispatch(in, out, n, 0, stream); }
extern "C" int cvpce_relu_f16(const void* in, void* out, long long n, void* stream) { return relu_dispatch(in, out, n, 1, stream); }

// The last two layers of the detector's Gaussian subnet in one pass (proposals.py:96-107: conv1x1 16 -> 16 + ReLU, conv1x1 16 -> 1 +
// ReLU | Tanh): one thread per pixel reads its 16 channels (32 B), keeps the 16 x 16 + 16 weights in LDS (broadcast reads), rounds
// the hidden layer to the storage type exactly where the two-launch form stored it, writes one float.  As two implicit-GEMM launches
// the pair took 80 us on 8 x 400 x 400 pixels (a 16-channel tensor written and read back for 272 MACs per pixel).
template <typename E>
__global__ __launch_bounds__(256) void gauss_tail_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w2, const float* __restrict__ b2,
                                                         const bf16_t* __restrict__ w3, const float* __restrict__ b3, float* __restrict__ out,
                                                         long long npix, int k2_pad, int act) {
    __shared__ float sw2[16 * 16], sb2[16], sw3[16], sb3;
    const int tid = threadIdx.x;
    sw2[tid] = E::widen(w2[(tid >> 4) * k2_pad + (tid & 15)]);
    if (tid < 16) { sb2[tid] = b2 ? b2[tid] : 0.f; sw3[tid] = E::widen(w3[tid]); }
    if (tid == 0) sb3 = b3 ? b3[0] : 0.f;
    __syncthreads();
    for (long long p = blockIdx.x * 256ll + tid; p < npix; p += (long long)gridDim.x * 256) {
        const bf16x8 lo = *reinterpret_cast<const bf16x8*>(x + p * 16), hi = *reinterpret_cast<const bf16x8*>(x + p * 16 + 8);
        float v[16];
#pragma unroll
        for (int c = 0; c < 8; ++c) { v[c] = E::widen(lo[c]); v[8 + c] = E::widen(hi[c]); }
        float z = sb3;
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            float h = sb2[o];
#pragma unroll
            for (int c = 0; c < 16; ++c) h = fmaf(sw2[o * 16 + c], v[c], h);
            h = E::widen(E::narrow(h > 0.f ? h : 0.f));
            z = fmaf(sw3[o], h, z);
        }
        out[p] = act == 2 ? tanhf(z) : (act == 1 ? (z > 0.f ? z : 0.f) : z);
    }
}

template <typename E>
static int gauss_tail_dispatch(const void* x, const void* w2, const float* b2, const void* w3, const float* b3, float* out, long long npix,
                               int k2_pad, int act, void* stream) {
    if (!x || !w2 || !w3 || !out || npix < 0 || k2_pad < 16 || act < 0 || act > 2) return CVPCE_ERR_ARG;
    if (npix == 0) return CVPCE_OK;
    const long long want = (npix + 255) / 256;
    hipLaunchKernelGGL(gauss_tail_kernel<E>, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)w2, b2, (const bf16_t*)w3, b3, out, npix, k2_pad, act);
    return cvpce_check_launch();
}
extern "C" int cvpce_gauss_tail_bf16(const void* x, const void* w2, const float* b2, const void* w3, const float* b3, float* out, long long npix,
                                     int k2_pad, int act, void* stream) {
    return gauss_tail_dispatch<ElemBF16>(x, w2, b2, w3, b3, out, npix, k2_pad, act, stream);
}
extern "C" int cvpce_gauss_tail_f16(const void* x, const void* w2, const float* b2, const void* w3, const float* b3, float* out, long long npix,
                                    int k2_pad, int act, void* stream) {
    return gauss_tail_dispatch<ElemF16>(x, w2, b2, w3, b3, out, npix, k2_pad, act, stream);
}

// Global max over H*W for each (image, channel): block = (64-channel slab, image);
// 256 threads = 8 channel-octets x 32 pixel lanes, 16 B loads, LDS tree over the pixel lanes.
__global__ void global_max_nhwc_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int HW, int C,
                                       int out_stride, int out_off) {
    __shared__ float red[32][65];
    const int n = blockIdx.y, cs = blockIdx.x * 64;
    const int o = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    const bf16_t* base = in + (size_t)n * HW * C + cs + o * 8;
    if (cs + o * 8 < C) {
        for (int p = pl; p < HW; p += 32) {
            uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)p * C);
            bf16x8 x = *reinterpret_cast<bf16x8*>(&v);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)x[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[pl][o * 8 + j] = m[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = -INFINITY;
        for (int p = 0; p < 32; ++p) v = fmaxf(v, red[p][threadIdx.x]);
        if (cs + threadIdx.x < C) out[(size_t)n * out_stride + out_off + cs + threadIdx.x] = v;
    }
}

extern "C" int cvpce_global_max_nhwc_bf16(const void* in, float* out, int N, int HW, int C, int out_stride,
                                          int out_off, void* stream) {
    if (!in || !out || C % 8 != 0 || HW <= 0) return CVPCE_ERR_ARG;
    if (N <= 0) return CVPCE_OK;
    hipLaunchKernelGGL(global_max_nhwc_kernel, dim3((C + 63) / 64, N), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, out, HW, C, out_stride, out_off);
    return cvpce_check_launch();
}

// desc / ||desc||_2.clamp(min=eps); one wave per row.
__global__ void l2_normalize_kernel(const float* __restrict__ in, float* __restrict__ out, bf16_t* __restrict__ out_bf16,
                                    int B, int D, float eps) {
    const int row = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* x = in + (size_t)row * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += x[i] * x[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    float nrm = fmaxf(sqrtf(s), eps);
    for (int i = lane; i < D; i += 64) {
        float v = x[i] / nrm;
        out[(size_t)row * D + i] = v;
        if (out_bf16) out_bf16[(size_t)row * D + i] = f32_to_bf16(v);
    }
}

extern "C" int cvpce_l2_normalize_f32(const float* in, float* out, void* out_bf16, int B, int D, float eps, void* stream) {
    if (!in || !out || D <= 0) return CVPCE_ERR_ARG;
    if (B <= 0) return CVPCE_OK;
    hipLaunchKernelGGL(l2_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, in, out,
                       (bf16_t*)out_bf16, B, D, eps);
    return cvpce_check_launch();
}

int g_cvpce_persistent_wgs = 256;

extern "C" int cvpce_set_persistent_workgroups(int n) {
    if (n < 1 || n > 256) return CVPCE_ERR_ARG;
    g_cvpce_persistent_wgs = n;
    return CVPCE_OK;
}


// ---------------------------------------------------------------------------
// Level atlas <-> per-level tensors in ONE launch (the detector's head towers run on an atlas of the five pyramid levels;
// five slice copies in, ten out were fifteen launches on the critical chain).  Unit = one 4- or 16-byte word of a pixel.
struct AtlasCopyArgs {
    void* level[8];                 // [N][h][w][bytes_per_pixel]
    int h[8], w[8], oy[8], ox[8], first[9];   // first[l] = pixels of levels < l (per image)
    void* atlas;                    // [N][hc][wc][bytes_per_pixel]
    int L, N, hc, wc, bpp, to_atlas;
};

template <typename V>
__global__ __launch_bounds__(256) void atlas_copy_kernel(AtlasCopyArgs a) {
    const int upp = a.bpp / (int)sizeof(V);                       // units per pixel
    const long long total = (long long)a.N * a.first[a.L] * upp;
    for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long long)gridDim.x * 256) {
        const long long p = u / upp;
        const int within = (int)(u - p * upp);
        const int n = (int)(p / a.first[a.L]);
        const int q = (int)(p - (long long)n * a.first[a.L]);
        int l = 0;
#pragma unroll
        for (int i = 1; i < 8; ++i) l += (i < a.L && q >= a.first[i]) ? 1 : 0;
        const int r = q - a.first[l], y = r / a.w[l], x = r - y * a.w[l];
        V* lv = reinterpret_cast<V*>(a.level[l]) + ((size_t)(n * a.h[l] + y) * a.w[l] + x) * upp + within;
        V* at = reinterpret_cast<V*>(a.atlas) + ((size_t)(n * a.hc + a.oy[l] + y) * a.wc + a.ox[l] + x) * upp + within;
        if (a.to_atlas) *at = *lv; else *lv = *at;
    }
}

extern "C" int cvpce_atlas_copy(void* const* levels, const int* h, const int* w, const int* oy, const int* ox, int L, int N,
                                void* atlas, int hc, int wc, int bytes_per_pixel, int to_atlas, void* stream) {
    if (N <= 0 || L <= 0) return CVPCE_OK;
    if (!levels || !h || !w || !oy || !ox || !atlas || L > 8 || hc <= 0 || wc <= 0 || bytes_per_pixel <= 0 || bytes_per_pixel % 4) return CVPCE_ERR_ARG;
    AtlasCopyArgs a;
    a.first[0] = 0;
    for (int l = 0; l < 8; ++l) {
        const bool on = l < L;
        a.level[l] = on ? levels[l] : nullptr;
        a.h[l] = on ? h[l] : 0; a.w[l] = on ? w[l] : 1; a.oy[l] = on ? oy[l] : 0; a.ox[l] = on ? ox[l] : 0;
        if (on && (!levels[l] || h[l] <= 0 || w[l] <= 0 || oy[l] < 0 || ox[l] < 0 || oy[l] + h[l] > hc || ox[l] + w[l] > wc)) return CVPCE_ERR_ARG;
        a.first[l + 1] = a.first[l] + (on ? h[l] * w[l] : 0);
    }
    a.atlas = atlas; a.L = L; a.N = N; a.hc = hc; a.wc = wc; a.bpp = bytes_per_pixel; a.to_atlas = to_atlas;
    const bool wide = bytes_per_pixel % 16 == 0;
    const long long total = (long long)N * a.first[L] * (bytes_per_pixel / (wide ? 16 : 4));
    if (total <= 0) return CVPCE_OK;
    const long long want = (total + 255) / 256;
    const unsigned grid = (unsigned)(want < 8192 ? want : 8192);
    if (wide) hipLaunchKernelGGL(atlas_copy_kernel<uint4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(atlas_copy_kernel<unsigned>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return cvpce_check_launch();
}
