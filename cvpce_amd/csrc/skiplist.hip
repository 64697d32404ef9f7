// Work lists for the embedder's constant-padding tile skipping.
//
// `resize_for_classification` (/root/reference/cvpce/datautils.py:232-239) pads every crop to a square with the constant 0.5
// (top-left anchored) before the bilinear resize to 256 x 256: a 2.6 : 1 box leaves 61 % of the embedder's input constant.  A
// conv output whose receptive field lies wholly in that region does not depend on the crop: it equals the same pixel of the
// all-padding crop (same kernel, same tile position, same K order => the same bits, image-border effects included).  So every
// pass of the embedder carries ONE extra crop -- the all-padding "constant crop", the last image of every tensor -- and
//   * a tile all of whose OUTPUT pixels lie in the constant region of its crop is not computed and not stored,
//   * a kernel that reads an input pixel in the constant region of its crop reads the constant crop's pixel instead
//     (conv3x3_halo2.hip / conv3x3_halo3.hip: the image index of the patch DMA's per-lane offset).
// Constant region of a tensor of the pass: rows >= e_y or columns >= e_x, where e is the crop's content extent e0
// (cvpce_crop_extents) taken through the ops between the crop and that tensor -- a 3x3 conv grows it by 1, a 2x2 pool halves it
// upwards -- and clamped to the tensor's size.  The op chain of the whole pass is one bit string (cvpce_embed_worklists `pool_mask`);
// a tensor is named by how many ops precede it, so its producer and its consumers compute the same, exact, extent.
//
// This kernel turns the extents into one compacted, crop-major tile list per layer (device-resident counts: no host sync):
//   entry = ((rows << 24 | ey_in << 12 | ex_in) << 32) | (n << 16) | (ty << 8) | tx
// rows = how many of the tile's 16 conv-output rows (before a fused pool) are NOT wholly constant, rounded up to 4: the halo
// kernels stop streaming patch rows behind them (rows of a tile below the crop's content extent are never read by anyone).
// Layers with skip == 3 get a second list, the STRIP list: their tiles with rows == 4, which conv3x3_halo2.hip computes three at a time.
#include "common.h"
#include "../../include/cvpce_amd.h"

#define SKL_MAX_LAYERS 16

struct WorklistArgs {
    const int* ext0;                 // [n - 1][2] content rows / columns at the crop resolution S; image n - 1 is the constant crop
    int n, S, nl;
    unsigned pool_mask;              // bit i: op i of the pass is a 2x2 pool (else a 3x3 conv)
    cvpce_skip_layer L[SKL_MAX_LAYERS];
    unsigned long long* lists;       // [2 * nl][stride]: the layers' lists, then their strip lists
    long long stride;
    int* counts;                     // [3 * nl]: tiles listed per layer, sixteenths of a tile's MFMA work actually performed per layer, strip-list entries
    int* computed;                   // optional [nl][n][2]: per crop, the conv-output rows a layer computes (rows below are left to the
                                     // constant crop) and its listed tile columns -- what cvpce_mac_init needs
};

__device__ __forceinline__ int skl_extent(int e0, int S, unsigned pool_mask, int nops, int size) {
    if (e0 >= S) return size;
    int e = e0;
    for (int i = 0; i < nops; ++i) e = ((pool_mask >> i) & 1u) ? (e + 1) >> 1 : e + 1;
    return e < size ? e : size;
}

__global__ __launch_bounds__(1024) void embed_worklists_kernel(WorklistArgs a) {
    __shared__ int s_cnt[1024];                        // scan scratch
    __shared__ int s_end[1024], s_send[1024];          // per crop of the current chunk: END offset of its entries in the chunk's list / strip list
    __shared__ unsigned s_geo[1024], s_ext[1024];      // (ny << 16 | nx << 8 | rows of the last tile row), (ey_in << 12 | ex_in)
    __shared__ int s_base, s_sbase, s_units;
    const cvpce_skip_layer L = a.L[blockIdx.x];
    const int tid = threadIdx.x;
    const int tiles_y = (L.H + L.tile_h - 1) / L.tile_h, tiles_x = (L.W + L.tile_w - 1) / L.tile_w;
    unsigned long long* list = a.lists + (long long)blockIdx.x * a.stride;
    unsigned long long* slist = a.lists + (long long)(a.nl + blockIdx.x) * a.stride;     // the layer's strip list (skip == 3)
    if (tid == 0) s_base = s_sbase = s_units = 0;
    __syncthreads();
    // inclusive scan of one value per thread over the 1024 slots (Hillis-Steele), left in s_cnt
    auto scan = [&](int v) {
        s_cnt[tid] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int t = tid >= d ? s_cnt[tid - d] : 0;
            __syncthreads();
            s_cnt[tid] += t;
            __syncthreads();
        }
    };
    for (int n0 = 0; n0 < a.n; n0 += 1024) {
        const int n = n0 + tid;
        int ny = 0, nx = 0, eiy = 0, eix = 0, last_rows = 16;
        if (n < a.n) {
            const bool is_const = n == a.n - 1;
            const int ey0 = is_const ? a.S : a.ext0[2 * n], ex0 = is_const ? a.S : a.ext0[2 * n + 1];
            ny = tiles_y; nx = tiles_x;
            if (L.skip) {
                const int eoy = skl_extent(ey0, a.S, a.pool_mask, L.out_ops, L.H), eox = skl_extent(ex0, a.S, a.pool_mask, L.out_ops, L.W);
                const int cy = (eoy + L.tile_h - 1) / L.tile_h, cx = (eox + L.tile_w - 1) / L.tile_w;   // tile row ty is computed iff ty * tile_h < eoy
                ny = cy < tiles_y ? cy : tiles_y;
                nx = cx < tiles_x ? cx : tiles_x;
                if (L.skip >= 2 && ny > 0) {
                    // the last listed tile row: its conv-output rows that are not constant, in sixteenths of the tile, rounded up to 4
                    const int left = eoy - (ny - 1) * L.tile_h;
                    const int act = left < L.tile_h ? left : L.tile_h;
                    last_rows = ((act * 16 + L.tile_h - 1) / L.tile_h + 3) & ~3;
                    if (last_rows > 16) last_rows = 16;
                }
            }
            eiy = skl_extent(ey0, a.S, a.pool_mask, L.in_ops, L.in_H);
            eix = skl_extent(ex0, a.S, a.pool_mask, L.in_ops, L.in_W);
        }
        if (nx == 0) ny = 0;
        // a last tile row with 4 useful rows goes to the strip list (three such tiles are computed as one, conv3x3_halo2.hip STRIP)
        const bool strips = L.skip >= 3 && ny > 0 && last_rows == 4;
        const int scnt = strips ? nx : 0, cnt = ny * nx - scnt;
        if (a.computed && n < a.n && blockIdx.y == 0) {
            // rows [0, rc) of the conv output (before a fused pool) are computed for this crop: whole tile rows, the last one cut
            a.computed[((long long)blockIdx.x * a.n + n) * 2] = ny > 0 ? (ny - 1) * 16 + last_rows : 0;
            a.computed[((long long)blockIdx.x * a.n + n) * 2 + 1] = nx;
        }
        // MFMA work of this crop's tiles in sixteenths of a tile: full tiles 16, a tile cut at `rows` rows + 1 (it streams patch rows
        // 0 .. rows + 1), a strip 4 (three strips share one pass of 72 MFMAs per step against a full tile's 96)
        if (ny > 0) atomicAdd(&s_units, nx * (16 * (ny - 1) + (strips ? 4 : (last_rows < 16 ? last_rows + 1 : 16))));
        s_geo[tid] = ((unsigned)ny << 16) | ((unsigned)nx << 8) | (unsigned)last_rows;
        s_ext[tid] = ((unsigned)eiy << 12) | (unsigned)eix;
        const int base0 = s_base, sbase0 = s_sbase;          // (read before the scans' barriers; updated behind them)
        scan(cnt);
        s_end[tid] = s_cnt[tid];
        const int total = s_cnt[1023];
        __syncthreads();
        scan(scnt);
        s_send[tid] = s_cnt[tid];
        const int stotal = s_cnt[1023];
        __syncthreads();
        // the chunk's entries, one per thread and trip, coalesced; blockIdx.y splits them among the layer's workgroups (every
        // workgroup runs the same scans).  entry e belongs to the first crop whose END offset exceeds e.
        auto crop_of = [&](const int* ends, int e) {
            int lo = 0, hi = 1023;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (ends[mid] > e) hi = mid; else lo = mid + 1;
            }
            return lo;
        };
        for (int e = (int)blockIdx.y * 1024 + tid; e < total; e += (int)gridDim.y * 1024) {
            const int c = crop_of(s_end, e);
            const unsigned geo = s_geo[c];
            const int cny = (int)(geo >> 16), cnx = (int)((geo >> 8) & 0xFF), j = e - (c ? s_end[c - 1] : 0);
            const int ty = j / cnx, tx = j - ty * cnx;
            const unsigned rows = (ty == cny - 1) ? (geo & 0xFF) : 16u;
            list[base0 + e] = ((unsigned long long)((rows << 24) | s_ext[c]) << 32) |
                              (unsigned long long)(((unsigned)(n0 + c) << 16) | ((unsigned)ty << 8) | (unsigned)tx);
        }
        for (int e = (int)blockIdx.y * 1024 + tid; e < stotal; e += (int)gridDim.y * 1024) {
            const int c = crop_of(s_send, e);
            const unsigned geo = s_geo[c];
            const int cny = (int)(geo >> 16), tx = e - (c ? s_send[c - 1] : 0);
            slist[sbase0 + e] = ((unsigned long long)((4u << 24) | s_ext[c]) << 32) |
                                (unsigned long long)(((unsigned)(n0 + c) << 16) | ((unsigned)(cny - 1) << 8) | (unsigned)tx);
        }
        __syncthreads();
        if (tid == 0) { s_base = base0 + total; s_sbase = sbase0 + stotal; }
        __syncthreads();
    }
    if (tid == 0 && blockIdx.y == 0) {
        a.counts[blockIdx.x] = s_base;
        a.counts[a.nl + blockIdx.x] = s_units;
        a.counts[2 * a.nl + blockIdx.x] = s_sbase;
    }
}

extern "C" int cvpce_embed_worklists(const int* ext0, int n_images, int S, unsigned pool_mask, const cvpce_skip_layer* layers, int n_layers,
                                     unsigned long long* lists, long long list_stride, int* counts, int* computed, void* stream) {
    if (n_layers <= 0) return CVPCE_OK;
    if (!ext0 && n_images > 1) return CVPCE_ERR_ARG;
    if (!layers || !lists || !counts || n_layers > SKL_MAX_LAYERS || n_images <= 0 || n_images > 65535 || S <= 0 || S > 32767) return CVPCE_ERR_ARG;
    WorklistArgs a;
    a.ext0 = ext0; a.n = n_images; a.S = S; a.nl = n_layers; a.pool_mask = pool_mask; a.lists = lists; a.stride = list_stride; a.counts = counts; a.computed = computed;
    for (int i = 0; i < n_layers; ++i) {
        const cvpce_skip_layer& l = layers[i];
        if (l.H <= 0 || l.W <= 0 || l.tile_h <= 0 || l.tile_w <= 0 || l.in_H <= 0 || l.in_W <= 0 || l.in_H > 4095 || l.in_W > 4095) return CVPCE_ERR_ARG;
        if (l.in_ops < 0 || l.out_ops < l.in_ops || l.out_ops > 32) return CVPCE_ERR_ARG;
        const long long ty = (l.H + l.tile_h - 1) / l.tile_h, tx = (l.W + l.tile_w - 1) / l.tile_w;
        if (ty > 255 || tx > 255 || ty * tx * n_images > list_stride) return CVPCE_ERR_ARG;
        a.L[i] = l;
    }
    const int split = n_images >= 64 ? 8 : 1;
    hipLaunchKernelGGL(embed_worklists_kernel, dim3(n_layers, split), dim3(1024), 0, (hipStream_t)stream, a);
    return cvpce_check_launch();
}


// MAC descriptor start values for the layers whose maximum (classification.py:46-49) ranges over tiles / rows the work lists leave
// out: what is left out is constant, i.e. the constant crop's own map there, whose row / column suffix maxima the host tabulated once
// (rtab[r][c] = max over rows >= r, ctab[x][c] = max over columns >= x; both end with a zero row).  desc[n][off + c] starts at
// max(rtab[rows computed for crop n][c], ctab[16 * tile columns listed for crop n][c]); the conv kernel's atomic maxima do the rest.
__global__ void mac_init_kernel(float* __restrict__ desc, int stride, int off, int C, const float* __restrict__ rtab, const float* __restrict__ ctab,
                                int Hc, int Wc, const int* __restrict__ computed) {
    const int n = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    int rc = computed[2 * n], cc = computed[2 * n + 1] * 16;
    rc = rc < Hc ? rc : Hc;
    cc = cc < Wc ? cc : Wc;
    desc[(size_t)n * stride + off + c] = fmaxf(rtab[(size_t)rc * C + c], ctab[(size_t)cc * C + c]);
}

extern "C" int cvpce_mac_init(float* desc, int n_images, int desc_stride, int desc_off, int C, const float* row_suffix_max,
                              const float* col_suffix_max, int Hc, int Wc, const int* computed, void* stream) {
    if (n_images <= 0) return CVPCE_OK;
    if (!desc || !row_suffix_max || !col_suffix_max || !computed || C <= 0 || Hc <= 0 || Wc <= 0 || desc_off < 0 || desc_off + C > desc_stride || n_images > 65535)
        return CVPCE_ERR_ARG;
    hipLaunchKernelGGL(mac_init_kernel, dim3((C + 255) / 256, n_images), dim3(256), 0, (hipStream_t)stream, desc, desc_stride, desc_off, C,
                       row_suffix_max, col_suffix_max, Hc, Wc, computed);
    return cvpce_check_launch();
}
