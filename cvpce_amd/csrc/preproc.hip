// Input-side gather kernels (HBM-bound):
//   K0  GeneralizedRCNNTransform (torchvision 0.9):
//       per-channel normalise, bilinear resize (align_corners=False, scale =
//       in/out), zero pad to the batch shape -> NHWC8 bf16 (3 real channels +
//       5 zero channels so the stem conv can gather 16-byte K-chunks).
//   K9  RoI crop (production.py:20: boxes.to(long) truncation, crop from the
//       ORIGINAL image) + pad to square with 0.5, top-left anchored
//       (datautils.py:234-238) + bilinear resize to 256x256 (ttf.resize on a
//       tensor in torchvision 0.9 = F.interpolate bilinear, no antialias),
//       optionally fused with scale_to_tanh (utils.py:280-281) and the MACVGG
//       input normalisation (classification.py:41-44) writing NHWC8 bf16.
#include "common.h"
#include "../../include/cvpce_amd.h"
#pragma clang fp contract(off)

// PyTorch area_pixel_compute_source_index (align_corners=False, non-cubic)
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

template <typename E>
__global__ void gln_transform_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int H0, int W0, int h,
                                     int w, int Hp, int Wp, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= Wp) return;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.f;     // (+0 has the same bits in bf16 and fp16)
    if (y < h && x < w) {
        const float sy = (float)H0 / (float)h, sx = (float)W0 / (float)w;
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        src_index(sy, y, H0, y0, y1, ly0, ly1);
        src_index(sx, x, W0, x0, x1, lx0, lx1);
        const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* p = img + (size_t)c * H0 * W0;
            float v00 = (p[(size_t)y0 * W0 + x0] - mean[c]) / stdv[c];
            float v01 = (p[(size_t)y0 * W0 + x1] - mean[c]) / stdv[c];
            float v10 = (p[(size_t)y1 * W0 + x0] - mean[c]) / stdv[c];
            float v11 = (p[(size_t)y1 * W0 + x1] - mean[c]) / stdv[c];
            float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
            o[c] = E::narrow(v);
        }
    }
    *reinterpret_cast<bf16x8*>(out + ((size_t)y * Wp + x) * 8) = o;
}

template <typename E>
static int gln_transform_dispatch(const float* img, void* out_nhwc8, int H0, int W0, int h, int w, int Hp, int Wp,
                                  const float* mean3, const float* std3, void* stream) {
    if (!img || !out_nhwc8 || !mean3 || !std3) return CVPCE_ERR_ARG;
    if (h > Hp || w > Wp || h <= 0 || w <= 0 || H0 <= 0 || W0 <= 0) return CVPCE_ERR_ARG;
    dim3 grid((Wp + 127) / 128, Hp);
    hipLaunchKernelGGL(gln_transform_kernel<E>, grid, dim3(128), 0, (hipStream_t)stream, img, (bf16_t*)out_nhwc8, H0, W0,
                       h, w, Hp, Wp, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    return cvpce_check_launch();
}

// The whole batch in ONE launch (grid.z = image): on the detector's critical chain eight 12-us launches in a row cost 0.1 ms.
struct TransformBatchArgs {
    const float* img[32];
    int H0[32], W0[32], h[32], w[32];
};

template <typename E>
__global__ void gln_transform_batch_kernel(TransformBatchArgs a, bf16_t* __restrict__ out, int Hp, int Wp, float m0, float m1, float m2,
                                           float s0, float s1, float s2) {
    const int i = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= Wp) return;
    const float* __restrict__ img = a.img[i];
    const int H0 = a.H0[i], W0 = a.W0[i], h = a.h[i], w = a.w[i];
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.f;
    if (y < h && x < w) {                                   // (the arithmetic of gln_transform_kernel, operation for operation)
        const float sy = (float)H0 / (float)h, sx = (float)W0 / (float)w;
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        src_index(sy, y, H0, y0, y1, ly0, ly1);
        src_index(sx, x, W0, x0, x1, lx0, lx1);
        const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* p = img + (size_t)c * H0 * W0;
            float v00 = (p[(size_t)y0 * W0 + x0] - mean[c]) / stdv[c];
            float v01 = (p[(size_t)y0 * W0 + x1] - mean[c]) / stdv[c];
            float v10 = (p[(size_t)y1 * W0 + x0] - mean[c]) / stdv[c];
            float v11 = (p[(size_t)y1 * W0 + x1] - mean[c]) / stdv[c];
            float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
            o[c] = E::narrow(v);
        }
    }
    *reinterpret_cast<bf16x8*>(out + (((size_t)i * Hp + y) * Wp + x) * 8) = o;
}

template <typename E>
static int gln_transform_batch_dispatch(const float* const* imgs, const int* H0, const int* W0, const int* h, const int* w, int n,
                                        void* out_nhwc8, int Hp, int Wp, const float* mean3, const float* std3, void* stream) {
    if (n <= 0) return CVPCE_OK;
    if (!imgs || !H0 || !W0 || !h || !w || !out_nhwc8 || !mean3 || !std3 || Hp <= 0 || Wp <= 0) return CVPCE_ERR_ARG;
    for (int base = 0; base < n; base += 32) {
        const int m = n - base < 32 ? n - base : 32;
        TransformBatchArgs a;
        for (int i = 0; i < 32; ++i) {
            const int j = base + (i < m ? i : 0);
            if (!imgs[j] || h[j] > Hp || w[j] > Wp || h[j] <= 0 || w[j] <= 0 || H0[j] <= 0 || W0[j] <= 0) return CVPCE_ERR_ARG;
            a.img[i] = imgs[j]; a.H0[i] = H0[j]; a.W0[i] = W0[j]; a.h[i] = h[j]; a.w[i] = w[j];
        }
        dim3 grid((Wp + 127) / 128, Hp, m);
        hipLaunchKernelGGL(gln_transform_batch_kernel<E>, grid, dim3(128), 0, (hipStream_t)stream, a,
                           (bf16_t*)out_nhwc8 + (size_t)base * Hp * Wp * 8, Hp, Wp, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
        const int rc = cvpce_check_launch();
        if (rc != CVPCE_OK) return rc;
    }
    return CVPCE_OK;
}

extern "C" int cvpce_gln_transform_batch(const float* const* imgs, const int* H0, const int* W0, const int* h, const int* w, int n,
                                         void* out_nhwc8, int Hp, int Wp, const float* mean3, const float* std3, void* stream) {
    return gln_transform_batch_dispatch<ElemBF16>(imgs, H0, W0, h, w, n, out_nhwc8, Hp, Wp, mean3, std3, stream);
}
extern "C" int cvpce_gln_transform_batch_f16(const float* const* imgs, const int* H0, const int* W0, const int* h, const int* w, int n,
                                             void* out_nhwc8, int Hp, int Wp, const float* mean3, const float* std3, void* stream) {
    return gln_transform_batch_dispatch<ElemF16>(imgs, H0, W0, h, w, n, out_nhwc8, Hp, Wp, mean3, std3, stream);
}

extern "C" int cvpce_gln_transform(const float* img, void* out_nhwc8, int H0, int W0, int h, int w, int Hp, int Wp,
                                   const float* mean3, const float* std3, void* stream) {
    return gln_transform_dispatch<ElemBF16>(img, out_nhwc8, H0, W0, h, w, Hp, Wp, mean3, std3, stream);
}
extern "C" int cvpce_gln_transform_f16(const float* img, void* out_nhwc8, int H0, int W0, int h, int w, int Hp, int Wp,
                                       const float* mean3, const float* std3, void* stream) {
    return gln_transform_dispatch<ElemF16>(img, out_nhwc8, H0, W0, h, w, Hp, Wp, mean3, std3, stream);
}

// One block row per (crop, output row); boxes are read on the device (no host sync).
// mode 0: f32 NCHW in [0,1] (the reference's `resize_for_classification` output)
// mode 1: bf16 NHWC8, scale_to_tanh + MACVGG normalisation fused
// mode 2: the same values as bf16 NHWC4 (8 bytes per pixel: half the bytes written here and read back by the fused VGG
//         stem, which takes 4- or 8-channel pixels)
__global__ void crop_resize_kernel(const float* __restrict__ img, const float* __restrict__ boxes,
                                   const int* __restrict__ count, void* __restrict__ out, int H0, int W0, int S,
                                   int mode, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int p = blockIdx.z;
    if (count && p >= *count) return;
    const int oy = blockIdx.y;
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    if (ox >= S) return;
    const float* b = boxes + (size_t)p * 4;
    // .to(torch.long): truncation toward zero; Python slice clamping to the image
    long long x1 = (long long)b[0], y1 = (long long)b[1], x2 = (long long)b[2], y2 = (long long)b[3];
    x1 = x1 < 0 ? 0 : (x1 > W0 ? W0 : x1);
    x2 = x2 < 0 ? 0 : (x2 > W0 ? W0 : x2);
    y1 = y1 < 0 ? 0 : (y1 > H0 ? H0 : y1);
    y2 = y2 < 0 ? 0 : (y2 > H0 ? H0 : y2);
    int cw = (int)(x2 - x1), ch = (int)(y2 - y1);
    if (cw < 0) cw = 0;
    if (ch < 0) ch = 0;
    const int larger = cw > ch ? cw : ch;
    float v[3] = {0.5f, 0.5f, 0.5f};
    if (larger > 0) {
        const float sc = (float)larger / (float)S;
        int yy0, yy1, xx0, xx1;
        float ly0, ly1, lx0, lx1;
        src_index(sc, oy, larger, yy0, yy1, ly0, ly1);
        src_index(sc, ox, larger, xx0, xx1, lx0, lx1);
        // a pixel all four of whose taps are padding (yy0 >= ch or xx0 >= cw: the taps only move further out) is EXACTLY the
        // pad constant -- the interpolation weights sum to 1 only up to an ulp; cvpce_crop_extents reports the same predicate
        if (yy0 < ch && xx0 < cw) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* pl = img + (size_t)c * H0 * W0;
                auto at = [&](int yy, int xx) -> float {
                    return (yy < ch && xx < cw) ? pl[(size_t)(y1 + yy) * W0 + (x1 + xx)] : 0.5f;
                };
                v[c] = ly0 * (lx0 * at(yy0, xx0) + lx1 * at(yy0, xx1)) + ly1 * (lx0 * at(yy1, xx0) + lx1 * at(yy1, xx1));
            }
        }
    }
    if (mode == 0) {
        float* o = reinterpret_cast<float*>(out) + (size_t)p * 3 * S * S;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(size_t)c * S * S + (size_t)oy * S + ox] = v[c];
    } else {
        const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = f32_to_bf16((v[c] * 2.f - 1.f - mean[c]) / stdv[c]);
        if (mode == 1) {
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(out) + (((size_t)p * S + oy) * S + ox) * 8) = o;
        } else {
            const bf16x4 o4 = {o[0], o[1], o[2], o[3]};
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(out) + (((size_t)p * S + oy) * S + ox) * 4) = o4;
        }
    }
}

// Modes 1 / 2 for the embedder, TWO output pixels per thread: 24 independent gathers in flight per thread instead of 12 (the
// one-pixel kernel is bound by load latency, not bandwidth) and one 16-byte store per thread in mode 2.  Same arithmetic,
// expression for expression, as crop_resize_kernel.
// ext (optional, cvpce_crop_resize_content): the crops' content extents -- rows / columns beyond them hold the pad constant and are
// NOT written (their workgroups / threads leave at once): the work-list embedder reads them from the constant crop.
// A workgroup is CROP2_ROWS output rows of one crop, one row per 128 threads (blockDim = (128, CROP2_ROWS)).  One row per workgroup
// is the fastest form measured (per 200-crop launch of the bench's wide boxes, under the profiler: whole crops 64 us, content only
// 48 us -- 60 % of its 51 200 workgroups leave at once); four rows per 512-thread workgroup 78 / 52 us; eight rows per THREAD, one
// after the other, 104 / 92 us (a row's 24 gathers are the latency: they want many independent waves).  What bounds the content-only
// form is VALU issue: ~700 instructions per wave (index arithmetic, six IEEE divisions, the box decode) x 40 000 live waves x 4 clocks
// on 1 024 SIMDs = 52 us; fetching count / extents / box in one round trip, branch-free gathers and two rows per workgroup each
// changed nothing (44.0 / 44.1 / 43.3 us).
#define CROP2_ROWS 1
template <int MODE>
__global__ void crop_resize2_kernel(const float* __restrict__ img, const float* __restrict__ boxes, const int* __restrict__ count,
                                    bf16_t* __restrict__ out, int H0, int W0, int S, float m0, float m1, float m2, float s0,
                                    float s1, float s2, const int* __restrict__ ext, int nbox) {
    // XCD-aware order (one workgroup per row: gridDim.x == 1): workgroups are dealt to the 8 XCDs round-robin in launch order, and
    // neighbouring output rows read the same source rows -- all rows of a crop go to ONE XCD's L2 (crop = 8 * (idx / rows) + xcd);
    // in launch order every source row was fetched into several L2s (PMC: 1.58 x the algorithmic bytes)
    int p = blockIdx.z, by = blockIdx.y;
    if (gridDim.x == 1) {
        const int L = (int)(blockIdx.y + gridDim.y * blockIdx.z), xcd = L & 7, idx = L >> 3;
        p = (idx / (int)gridDim.y) * 8 + xcd;
        by = idx % (int)gridDim.y;
    }
    if (p >= nbox) return;                                 // (the grid's z extent is rounded up to a multiple of 8)
    // the count, the crop's extents and its box are fetched TOGETHER, before the first of them is looked at: as `if (p >= *count)
    // return; if (oy >= ext[..]) return; b = boxes[..]` they were three dependent round trips at the head of every wave's life
    const float* b = boxes + (size_t)p * 4;
    const int* cptr = count ? count : reinterpret_cast<const int*>(b);
    const int* eptr = ext ? ext + 2 * p : reinterpret_cast<const int*>(b);
    const int n_valid = *cptr, e_rows = eptr[0], e_cols = eptr[1];
    const float bx1 = b[0], by1 = b[1], bx2 = b[2], by2 = b[3];
    if (count && p >= n_valid) return;
    const int oy_first = by * CROP2_ROWS + threadIdx.y;
    const int ox0 = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (ox0 >= S) return;
    int oy_end = oy_first + 1 < S ? oy_first + 1 : S;
    if (ext) {
        if (ox0 >= e_cols) return;
        oy_end = oy_end < e_rows ? oy_end : e_rows;
    }
    if (oy_first >= oy_end) return;
    long long x1 = (long long)bx1, y1 = (long long)by1, x2 = (long long)bx2, y2 = (long long)by2;
    x1 = x1 < 0 ? 0 : (x1 > W0 ? W0 : x1);
    x2 = x2 < 0 ? 0 : (x2 > W0 ? W0 : x2);
    y1 = y1 < 0 ? 0 : (y1 > H0 ? H0 : y1);
    y2 = y2 < 0 ? 0 : (y2 > H0 ? H0 : y2);
    int cw = (int)(x2 - x1), ch = (int)(y2 - y1);
    if (cw < 0) cw = 0;
    if (ch < 0) ch = 0;
    const int larger = cw > ch ? cw : ch;
    const float sc = (float)larger / (float)S;
    int xx0[2] = {0, 0}, xx1[2] = {0, 0};
    float lx0[2] = {0.f, 0.f}, lx1[2] = {0.f, 0.f};
    if (larger > 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) src_index(sc, ox0 + u, larger, xx0[u], xx1[u], lx0[u], lx1[u]);
    }
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    for (int oy = oy_first; oy < oy_end; ++oy) {
        float v[2][3] = {{0.5f, 0.5f, 0.5f}, {0.5f, 0.5f, 0.5f}};
        if (larger > 0) {
            int yy0, yy1;
            float ly0, ly1;
            src_index(sc, oy, larger, yy0, yy1, ly0, ly1);
            // the 24 taps: every load is issued (at an address clamped into the plane), padding taps are replaced afterwards -- behind
            // `tap inside ? load : 0.5` branches the loads went out one exec-mask region at a time
            float t[2][3][4];
            const int plane = H0 * W0;                       // (< 2^31: checked on the host)
            const int yo[2] = {yy0, yy1};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int xo[2] = {xx0[u], xx1[u]};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int yy = yo[k >> 1], xx = xo[k & 1];
                    const bool in = yy < ch && xx < cw;
                    int off = ((int)y1 + yy) * W0 + ((int)x1 + xx);
                    off = in ? off : 0;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float raw = img[(size_t)c * plane + off];
                        t[u][c][k] = in ? raw : 0.5f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (yy0 < ch && xx0[u] < cw) {               // else: all four taps are padding -> exactly 0.5 (see crop_resize_kernel)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        v[u][c] = ly0 * (lx0[u] * t[u][c][0] + lx1[u] * t[u][c][1]) + ly1 * (lx0[u] * t[u][c][2] + lx1[u] * t[u][c][3]);
                }
        }
        bf16_t o[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[u][c] = f32_to_bf16((v[u][c] * 2.f - 1.f - mean[c]) / stdv[c]);
            o[u][3] = (bf16_t)0.f;
        }
        const size_t pix = ((size_t)p * S + oy) * S + ox0;
        if (MODE == 2) {
            const bf16x8 w = {o[0][0], o[0][1], o[0][2], o[0][3], o[1][0], o[1][1], o[1][2], o[1][3]};
            if (ox0 + 1 < S) *reinterpret_cast<bf16x8*>(out + pix * 4) = w;
            else *reinterpret_cast<bf16x4*>(out + pix * 4) = bf16x4{o[0][0], o[0][1], o[0][2], o[0][3]};
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (ox0 + u < S) {
                    const bf16x8 w = {o[u][0], o[u][1], o[u][2], (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                    *reinterpret_cast<bf16x8*>(out + (pix + u) * 8) = w;
                }
        }
    }
}

static int crop_resize_launch(const float* img, const float* boxes, const int* count_dev, int max_boxes, void* out,
                             int H0, int W0, int S, int mode, const float* mean3, const float* std3, const int* ext, void* stream) {
    if (!img || !boxes || !out || S <= 0 || H0 <= 0 || W0 <= 0) return CVPCE_ERR_ARG;
    if (mode < 0 || mode > 2 || (mode != 0 && (!mean3 || !std3))) return CVPCE_ERR_ARG;
    if (ext && (mode == 0 || S % 2 != 0)) return CVPCE_ERR_ARG;
    if ((long long)H0 * W0 >= (1LL << 31)) return CVPCE_ERR_ARG;        // (32-bit pixel offsets within a plane)
    if (max_boxes <= 0) return CVPCE_OK;
    if (max_boxes > 65528) return CVPCE_ERR_ARG;          // (grid z, rounded up to a multiple of 8)
    dim3 grid((S + 127) / 128, S, max_boxes);
    float m[3] = {0, 0, 0}, s[3] = {1, 1, 1};
    if (mode != 0) for (int i = 0; i < 3; ++i) { m[i] = mean3[i]; s[i] = std3[i]; }
    if (mode != 0 && S % 2 == 0) {
        dim3 grid2((S / 2 + 127) / 128, (S + CROP2_ROWS - 1) / CROP2_ROWS, (max_boxes + 7) / 8 * 8);
        if (mode == 2)
            hipLaunchKernelGGL(crop_resize2_kernel<2>, grid2, dim3(128, CROP2_ROWS), 0, (hipStream_t)stream, img, boxes, count_dev, (bf16_t*)out,
                               H0, W0, S, m[0], m[1], m[2], s[0], s[1], s[2], ext, max_boxes);
        else
            hipLaunchKernelGGL(crop_resize2_kernel<1>, grid2, dim3(128, CROP2_ROWS), 0, (hipStream_t)stream, img, boxes, count_dev, (bf16_t*)out,
                               H0, W0, S, m[0], m[1], m[2], s[0], s[1], s[2], ext, max_boxes);
        return cvpce_check_launch();
    }
    hipLaunchKernelGGL(crop_resize_kernel, grid, dim3(128), 0, (hipStream_t)stream, img, boxes, count_dev, out, H0, W0,
                       S, mode, m[0], m[1], m[2], s[0], s[1], s[2]);
    return cvpce_check_launch();
}

extern "C" int cvpce_crop_resize(const float* img, const float* boxes, const int* count_dev, int max_boxes, void* out,
                                 int H0, int W0, int S, int mode, const float* mean3, const float* std3, void* stream) {
    return crop_resize_launch(img, boxes, count_dev, max_boxes, out, H0, W0, S, mode, mean3, std3, nullptr, stream);
}

// The crops' CONTENT only (modes 1 / 2, even S): pixels with oy >= ext[p].rows or ox >= ext[p].cols (cvpce_crop_extents of the same
// boxes, launched before this) are left unwritten -- valid input only for the work-list embedder, which never reads them.
extern "C" int cvpce_crop_resize_content(const float* img, const float* boxes, const int* count_dev, int max_boxes, void* out,
                                         int H0, int W0, int S, int mode, const float* mean3, const float* std3, const int* ext,
                                         void* stream) {
    if (!ext) return CVPCE_ERR_ARG;
    return crop_resize_launch(img, boxes, count_dev, max_boxes, out, H0, W0, S, mode, mean3, std3, ext, stream);
}

// Content extent of every crop at the crop resolution S (the embedder's constant-padding tile skipping, skiplist.hip):
// ext[p] = (rows, cols) such that every output pixel with oy >= rows or ox >= cols is exactly the pad constant -- the predicate
// under which crop_resize*_kernel writes the constant, evaluated with the same src_index.  Boxes beyond *count: (S, S).
__global__ void crop_extents_kernel(const float* __restrict__ boxes, const int* __restrict__ count, int max_boxes, int per_image, int H0,
                                    int W0, int S, int* __restrict__ ext) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= max_boxes) return;
    int ey = S, ex = S;
    // per_image > 0: the boxes of several images of one size, `per_image` slots each, one count per image
    const bool live = !count || (per_image > 0 ? (p % per_image) < count[p / per_image] : p < *count);
    if (live) {
        const float* b = boxes + (size_t)p * 4;
        long long x1 = (long long)b[0], y1 = (long long)b[1], x2 = (long long)b[2], y2 = (long long)b[3];
        x1 = x1 < 0 ? 0 : (x1 > W0 ? W0 : x1);
        x2 = x2 < 0 ? 0 : (x2 > W0 ? W0 : x2);
        y1 = y1 < 0 ? 0 : (y1 > H0 ? H0 : y1);
        y2 = y2 < 0 ? 0 : (y2 > H0 ? H0 : y2);
        int cw = (int)(x2 - x1), ch = (int)(y2 - y1);
        if (cw < 0) cw = 0;
        if (ch < 0) ch = 0;
        const int larger = cw > ch ? cw : ch;
        ey = ex = 0;                                         // (degenerate box: the whole crop is the constant)
        if (larger > 0) {
            const float sc = (float)larger / (float)S;
            // the first source index of output row / column o is monotone in o (fp32 rounding is monotone): the number of o with
            // index < limit is the first o whose index reaches it -- a bisection instead of a scan of all S positions
            auto first_at_least = [&](int limit) {
                int lo = 0, hi = S;                          // answer in [lo, hi]
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    int i0, i1;
                    float l0, l1;
                    src_index(sc, mid, larger, i0, i1, l0, l1);
                    if (i0 < limit) lo = mid + 1; else hi = mid;
                }
                return lo;
            };
            ey = first_at_least(ch);
            ex = first_at_least(cw);
        }
    }
    ext[2 * p] = ey;
    ext[2 * p + 1] = ex;
}

extern "C" int cvpce_crop_extents(const float* boxes, const int* count_dev, int max_boxes, int boxes_per_image, int H0, int W0, int S,
                                  int* ext_out, void* stream) {
    if (!boxes || !ext_out || S <= 0 || H0 <= 0 || W0 <= 0 || boxes_per_image < 0) return CVPCE_ERR_ARG;
    if (max_boxes <= 0) return CVPCE_OK;
    if (boxes_per_image > 0 && max_boxes % boxes_per_image != 0) return CVPCE_ERR_ARG;
    hipLaunchKernelGGL(crop_extents_kernel, dim3((max_boxes + 63) / 64), dim3(64), 0, (hipStream_t)stream, boxes, count_dev, max_boxes,
                       boxes_per_image, H0, W0, S, ext_out);
    return cvpce_check_launch();
}

// Content extent of crops that are already materialised as (B,3,S,S) f32 tensors (the reference-shaped `Classifier.classify`
// input, production.py:57-74): ext[b] = (rows, cols) such that every pixel with y >= rows or x >= cols equals `pad` in all three
// channels -- found by looking at the data, so it holds for any tensor (crops made by cvpce_crop_resize carry exact 0.5 padding;
// a tensor without constant borders simply gets (S, S)).  One workgroup per (row, crop); ext_out is zeroed by the entry point.
__global__ void pad_extents_kernel(const float* __restrict__ in, int S, float pad, int* __restrict__ ext) {
    const int y = blockIdx.x, b = blockIdx.y;
    const float* p = in + (size_t)b * 3 * S * S + (size_t)y * S;
    int mx = 0;
    for (int x = threadIdx.x; x < S; x += blockDim.x)
        if (p[x] != pad || p[(size_t)S * S + x] != pad || p[2 * (size_t)S * S + x] != pad) mx = x + 1;
    __shared__ int smax;
    if (threadIdx.x == 0) smax = 0;
    __syncthreads();
    if (mx) atomicMax(&smax, mx);
    __syncthreads();
    if (threadIdx.x == 0 && smax) {
        atomicMax(&ext[2 * b], y + 1);
        atomicMax(&ext[2 * b + 1], smax);
    }
}

extern "C" int cvpce_pad_extents(const float* in, int B, int S, float pad, int* ext_out, void* stream) {
    if (!in || !ext_out || S <= 0) return CVPCE_ERR_ARG;
    if (B <= 0) return CVPCE_OK;
    if (B > 65535) return CVPCE_ERR_ARG;
    if (hipMemsetAsync(ext_out, 0, sizeof(int) * 2 * (size_t)B, (hipStream_t)stream) != hipSuccess) return CVPCE_ERR_LAUNCH;
    hipLaunchKernelGGL(pad_extents_kernel, dim3(S, B), dim3(S < 256 ? 64 : 256), 0, (hipStream_t)stream, in, S, pad, ext_out);
    return cvpce_check_launch();
}

// (B,3,S,S) f32 NCHW -> NHWC8 bf16 with optional scale_to_tanh then (x - mean) / std.
__global__ void pack_embed_input_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long long npix, int SS,
                                        int to_tanh, float m0, float m1, float m2, float s0, float s1, float s2) {
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        long long b = i / SS;
        int p = (int)(i - b * SS);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = in[((size_t)b * 3 + c) * SS + p];
            if (to_tanh) v = v * 2.f - 1.f;
            o[c] = f32_to_bf16((v - mean[c]) / stdv[c]);
        }
        *reinterpret_cast<bf16x8*>(out + (size_t)i * 8) = o;
    }
}

extern "C" int cvpce_pack_embed_input(const float* in, void* out_nhwc8, int B, int S, int to_tanh, const float* mean3,
                                      const float* std3, void* stream) {
    if (!in || !out_nhwc8 || !mean3 || !std3 || S <= 0) return CVPCE_ERR_ARG;
    long long npix = (long long)B * S * S;
    if (npix <= 0) return CVPCE_OK;
    int blocks = (int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_embed_input_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out_nhwc8,
                       npix, S * S, to_tanh, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    return cvpce_check_launch();
}
