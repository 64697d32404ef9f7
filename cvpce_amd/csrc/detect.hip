// RetinaNet post-processing on the GPU, torchvision 0.9
// semantics (Appendix A: postprocess_detections, BoxCoder.decode_single,
// clip_boxes_to_image, batched_nms/nms, transform.postprocess), reached from
// /root/reference/cvpce/models/proposals.py:176-181 via RetinaNet.forward and
// consumed at /root/reference/cvpce/production.py:13-15.
//
//   decode_select  one 1024-thread workgroup per 8192-logit CHUNK of a (level, image): score threshold (exact, see
//                score_passes), exact top-k of the chunk by radix select on (logit key, ~index) held in registers; the
//                chunk's survivors (<= topk, unsorted) go to a scratch run.  P3 of an 800 x 800 image is 11 chunks on 11 CUs
//                where one workgroup used to walk all 90 000 logits.
//   decode_merge   one workgroup per (level, image): the same selection over the level's runs (a chunk's top-k holds every
//                member of the level's top-k that lies in the chunk), LDS bitonic sort of the <= 1024 survivors, anchor
//                synthesis + box decode + clip.
//   decode_topk  the one-kernel form of the two above (one workgroup per (level, image), logits beyond 56 per thread
//                streamed per pass): kept for levels of more than 32 chunks and for workspaces without room for the runs.
//   nms_sort     one workgroup per image: the levels' sorted candidate lists are merged by RANK (every candidate finds, by
//                binary search in LDS, how many candidates of the other levels precede it) -> sorted by (logit desc,
//                level asc, rank asc), i.e. the order of the concatenated candidate list under a stable sort.
//   nms_mask     64x64 IoU bit-matrix tiles (upper triangle), one wave per tile.
//   nms_scan     one workgroup per image: 64-box chunks; in-chunk dependencies
//                by one wave (v_readlane on the diagonal words, jumping from
//                survivor to survivor), cross-chunk by all 16 waves OR-ing the
//                kept rows (8 independent row loads per thread); early exit at
//                detections_per_img; rescale to the original image, count the
//                score > confidence prefix.
//
// Ordering rule: the reference sorts on sigmoid scores with an unstable sort,
// leaving ties unspecified.  Here every ordering uses the fp32 *logit* (a
// monotone refinement of the score order) and then the lower index.
#include "common.h"
#include "../../include/cvpce_amd.h"
#include <math.h>
#include <stdlib.h>
#pragma clang fp contract(off)

#define MAX_LEVELS 8
#define SORT_CAP 8192

struct DecodeArgs {
    const float* logits[MAX_LEVELS];   // [N][gh*gw*A*K]
    const float* regs[MAX_LEVELS];     // [N][gh*gw*A][4]
    int gh[MAX_LEVELS], gw[MAX_LEVELS], sh[MAX_LEVELS], sw[MAX_LEVELS];
    const float* base_anchors;         // [L][A][4]
    const int* image_hw;               // [N][2] resized (unpadded) sizes
    int L, N, A, K, topk;
    float score_thresh, xform_clip;
    float* cand_boxes;                 // [N][L*topk][4]
    float* cand_scores;                // [N][L*topk]
    float* cand_logits;                // [N][L*topk]
    int* cand_labels;                  // [N][L*topk]
    int* cand_count;                   // [N][L]
    // chunked form (decode_select_kernel + decode_merge_kernel)
    int chunk_first[MAX_LEVELS + 1];   // first chunk of each level in an image's chunk list; [L] = chunks per image
    unsigned long long* runs;          // [N][chunks per image][1024] (logit key << 32 | ~index in level), unsorted, 0 = empty
    float t_lo, t_hi;                  // score_passes: logits outside [t_lo, t_hi] are decided without evaluating the sigmoid
};

__device__ __forceinline__ unsigned ordered_key(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.f / (1.f + expf(-x)); }

// descending bitonic sort of n (power of two) 64-bit keys in LDS by `nthreads` threads
__device__ void bitonic_sort_desc(unsigned long long* s, int n, int tid, int nthreads) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n; i += nthreads) {
                int p = i ^ j;
                if (p > i) {
                    unsigned long long a = s[i], b = s[p];
                    bool desc = ((i & k) == 0);
                    if (desc ? (a < b) : (a > b)) { s[i] = b; s[p] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// the same sort for exactly 1024 keys by 1024 threads, one key per thread IN A REGISTER: the 45 network stages whose partner
// lies in the same wave (distance < 64) are two 32-bit lane exchanges, only the 10 stages at distance >= 64 go through LDS
// and a barrier (the LDS form above takes 55 barriers)
__device__ __forceinline__ void bitonic_sort1024_desc(unsigned long long* s, int tid) {
    unsigned long long v = s[tid];
    for (int k = 2; k <= 1024; k <<= 1) {
        const bool desc = (tid & k) == 0;
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned long long o;
            if (j >= 64) {
                __syncthreads();
                s[tid] = v;
                __syncthreads();
                o = s[tid ^ j];
            } else {
                o = __shfl_xor(v, j);
            }
            const bool keep_max = ((tid & j) == 0) == desc;
            v = keep_max ? (v > o ? v : o) : (v < o ? v : o);
        }
    }
    __syncthreads();
    s[tid] = v;
    __syncthreads();
}

__global__ __launch_bounds__(1024) void decode_topk_kernel(DecodeArgs a) {
    __shared__ unsigned long long sel[1024];
    __shared__ int hist[256];
    __shared__ int wtot[16];
    __shared__ int s_cnt, s_prefix, s_need, s_nsel, s_eqbase, s_eqtotal;

    const int level = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const int gh = a.gh[level], gw = a.gw[level];
    const int n = gh * gw * a.A * a.K;
    const float* lg = a.logits[level] + (size_t)img * n;
    const float* rg = a.regs[level] + (size_t)img * (size_t)(gh * gw * a.A) * 4;

    if (tid == 0) { s_cnt = 0; s_nsel = 0; s_eqbase = 0; s_eqtotal = 0; }
    sel[tid] = 0ull;
    __syncthreads();
    // The first R * 1024 logits of a level are read from memory ONCE: every thread keeps R keys in registers (0 = not a
    // candidate) and the count, the four radix passes and the compaction run on them; what lies beyond (P3 of an 800 x 800
    // image has 90 000 logits, 128 VGPRs at 1024 threads hold 56 keys per thread) is STREAMED: every pass re-reads it with
    // UNROLL independent loads in flight per thread.  Six full passes over memory cost the P3 workgroup 6 x 11 round trips
    // of ~2.5 us; now 11 + 5 x 4.  Both parts evaluate the score test exactly as the oracle does.
    constexpr int UNROLL = 8, R = 56;
    const int lane = tid & 63;
    const int nreg = n < R * 1024 ? n : R * 1024;       // logits [0, nreg) live in registers, [nreg, n) are streamed
    unsigned keys[R];
    // wave-aggregated histogram update: the lanes that share the first active lane's bin are counted by ONE LDS atomic (on the
    // leading digits nearly every key of a level falls into the same two or three bins, and 64 lanes hitting one bin
    // serialise), the others add individually
    auto hist_add = [&](bool pred, unsigned bin) {
        const unsigned long long act = __ballot(pred);
        if (act) {
            const int leader = __ffsll((long long)act) - 1;
            const unsigned lb = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
            const bool same = pred && bin == lb;
            const unsigned long long sm = __ballot(same);
            if (lane == leader) atomicAdd(&hist[lb], __popcll(sm));
            if (pred && !same) atomicAdd(&hist[bin], 1);
        }
    };
    auto take = [&](unsigned key, int i) {
        const int pos = atomicAdd(&s_nsel, 1);
        if (pos < 1024) sel[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    };
    // count candidates above the score threshold
    int local = 0;
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int i = u * 1024 + tid;
        const float l = lg[i < nreg ? i : nreg - 1];            // (clamped, not predicated: the loads stay independent)
        const bool ok = i < nreg && sigmoidf_ref(l) > a.score_thresh;
        keys[u] = ok ? ordered_key(l) : 0u;                     // ordered_key is 0 only for one NaN pattern, never a candidate
        local += ok ? 1 : 0;
    }
    for (int base = nreg; base < n; base += UNROLL * 1024) {
        float v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const int i = base + u * 1024 + tid; v[u] = i < n ? lg[i] : -INFINITY; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) local += (sigmoidf_ref(v[u]) > a.score_thresh) ? 1 : 0;   // sigmoid(-inf) = 0
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
    if (lane == 0) atomicAdd(&s_cnt, local);
    __syncthreads();
    const int cnt = s_cnt;
    const int k = cnt < a.topk ? cnt : a.topk;

    unsigned T = 0;       // key of the k-th largest candidate
    int need_eq = 0;      // how many candidates with key == T are taken (lowest index first)
    if (cnt > k) {
        unsigned prefix = 0, mask = 0;
        int need = k;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const unsigned key = keys[u];
                hist_add(key != 0u && (key & mask) == prefix, (key >> shift) & 255u);
            }
            for (int base = nreg; base < n; base += UNROLL * 1024) {
                float v[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) { const int i = base + u * 1024 + tid; v[u] = i < n ? lg[i] : -INFINITY; }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const unsigned key = ordered_key(v[u]);
                    hist_add((sigmoidf_ref(v[u]) > a.score_thresh) && ((key & mask) == prefix), (key >> shift) & 255u);
                }
            }
            __syncthreads();
            if (tid == 0) {
                int cum = 0, b = 255;
                for (; b > 0; --b) {
                    if (cum + hist[b] >= need) break;
                    cum += hist[b];
                }
                s_prefix = (int)(prefix | ((unsigned)b << shift));
                s_need = need - cum;
                s_eqtotal = hist[b];          // after the last pass: how many candidates carry exactly the key T
            }
            __syncthreads();
            prefix = (unsigned)s_prefix;
            need = s_need;
            mask |= 0xFFu << shift;
            __syncthreads();
        }
        T = prefix;
        need_eq = need;
    }
    if (cnt <= k || need_eq == s_eqtotal) {
        // every candidate with key >= T is taken (all ties at T fit): the order of arrival is irrelevant, the bitonic sort
        // below orders by (key, index)
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const unsigned key = keys[u];
            if (key != 0u && (cnt <= k || key >= T)) take(key, u * 1024 + tid);
        }
        for (int base = nreg; base < n; base += UNROLL * 1024) {
            float v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { const int i = base + u * 1024 + tid; v[u] = i < n ? lg[i] : -INFINITY; }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const unsigned key = ordered_key(v[u]);
                if ((sigmoidf_ref(v[u]) > a.score_thresh) && (cnt <= k || key >= T)) take(key, base + u * 1024 + tid);
            }
        }
    } else {
        // more candidates tie at the key T than fit: take the first need_eq of them in index order (ordered compaction)
        for (int base = 0; base < n; base += 1024) {
            const int i = base + tid;
            bool valid = false, gt = false, eq = false;
            unsigned key = 0;
            if (i < n) {
                float l = lg[i];
                valid = sigmoidf_ref(l) > a.score_thresh;
                key = ordered_key(l);
                gt = valid && key > T; eq = valid && key == T;
            }
            bool tk = gt;
            unsigned long long bal = __ballot(eq);
            int w = tid >> 6;
            int pre = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wtot[w] = __popcll(bal);
            __syncthreads();
            int before = s_eqbase;
            for (int ww = 0; ww < w; ++ww) before += wtot[ww];
            if (eq && before + pre < need_eq) tk = true;
            __syncthreads();
            if (tid == 0) {
                int t = 0;
                for (int ww = 0; ww < 16; ++ww) t += wtot[ww];
                s_eqbase += t;
            }
            __syncthreads();
            if (tk) take(key, i);
        }
    }
    __syncthreads();
    bitonic_sort_desc(sel, 1024, tid, 1024);

    const int slot = a.L * a.topk;
    if (tid == 0) a.cand_count[img * a.L + level] = k;
    if (tid < k) {
        const unsigned long long comp = sel[tid];
        const int idx = (int)(0xFFFFFFFFu - (unsigned)(comp & 0xFFFFFFFFull));
        const float l = lg[idx];
        const int aidx = idx / a.K, label = idx - aidx * a.K;
        const int cell = aidx / a.A, an = aidx - cell * a.A;
        const int y = cell / gw, x = cell - y * gw;
        const float shx = (float)x * (float)a.sw[level], shy = (float)y * (float)a.sh[level];
        const float* ba = a.base_anchors + ((size_t)level * a.A + an) * 4;
        const float ax1 = shx + ba[0], ay1 = shy + ba[1], ax2 = shx + ba[2], ay2 = shy + ba[3];
        const float w = ax2 - ax1, h = ay2 - ay1;
        const float cx = ax1 + 0.5f * w, cy = ay1 + 0.5f * h;
        const float4 r = *reinterpret_cast<const float4*>(rg + (size_t)aidx * 4);
        const float dw = fminf(r.z, a.xform_clip), dh = fminf(r.w, a.xform_clip);
        const float pcx = r.x * w + cx, pcy = r.y * h + cy;
        const float pw = expf(dw) * w, ph = expf(dh) * h;
        float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
        const float ih = (float)a.image_hw[img * 2 + 0], iw = (float)a.image_hw[img * 2 + 1];
        x1 = fminf(fmaxf(x1, 0.f), iw); x2 = fminf(fmaxf(x2, 0.f), iw);
        y1 = fminf(fmaxf(y1, 0.f), ih); y2 = fminf(fmaxf(y2, 0.f), ih);
        const size_t o = (size_t)img * slot + (size_t)level * a.topk + tid;
        *reinterpret_cast<float4*>(a.cand_boxes + o * 4) = make_float4(x1, y1, x2, y2);
        a.cand_scores[o] = sigmoidf_ref(l);
        a.cand_logits[o] = l;
        a.cand_labels[o] = label;
    }
}

// ---------------------------------------------------------------------------
// Chunked form.  sigmoid(l) > thresh is what the oracle evaluates (fp32 expf, add, divide); that result can only depend on
// rounding for logits within a narrow band around logit(thresh), so outside [t_lo, t_hi] (host: +-1e-3 (1 + |logit(thresh)|),
// i.e. a relative change of the score of >= 1e-5 for thresholds in [1e-6, 0.99] against an evaluation error of a few 1e-7)
// the comparison of the logit decides, and the ~40 instructions of the sigmoid run for the band only.
__device__ __forceinline__ bool score_passes(float l, float thresh, float t_lo, float t_hi) {
    return l > t_hi || (l >= t_lo && sigmoidf_ref(l) > thresh);      // NaN: false, as sigmoid(NaN) > thresh is
}

#define DECODE_CHUNK 8192          // logits per decode_select workgroup (8 per thread, in registers)
#define DECODE_MAX_RUNS 32         // runs one decode_merge workgroup holds in registers (one entry per thread and run)
#define DECODE_RUNS_PER_LEVEL 12   // workspace room: N * L * 12 runs (800^2: 17 per image; 1333^2: 43)

struct SelectShared {
    unsigned long long sel[1024];
    unsigned long long prefix;
    int hist[256];
    int cnt, nsel, need, eqtotal;
};

// Exact top-k of the composites comp[u] != 0 (u < nslots; unique 64-bit values) held in the registers of a 1024-thread
// workgroup: k = min(count, topk) of them end up in sh.sel[0 .. k) (unsorted, the rest of sel is zero); returns k.
// Radix select from the top byte down; it stops at the first byte where every composite sharing the prefix fits (on random
// logits after 3 of 8 bytes; ties of the logit are resolved by the index bytes, lowest index first, with no special case).
template <int R>
__device__ __forceinline__ int select_topk_regs(const unsigned long long (&comp)[R], int nslots, int topk, SelectShared& sh, int tid) {
    const int lane = tid & 63;
    if (tid == 0) { sh.cnt = 0; sh.nsel = 0; }
    sh.sel[tid] = 0ull;
    __syncthreads();
    int local = 0;
#pragma unroll
    for (int u = 0; u < R; ++u)
        if (u < nslots) local += comp[u] != 0ull ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
    if (lane == 0 && local) atomicAdd(&sh.cnt, local);
    __syncthreads();
    const int cnt = sh.cnt;
    const int k = cnt < topk ? cnt : topk;
    unsigned long long T = 1ull;                                     // (valid composites are >= 2^32)
    if (cnt > k) {
        unsigned long long prefix = 0ull, mask = 0ull;
        int need = k;
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (tid < 256) sh.hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < R; ++u) {
                if (u < nslots) {
                    const unsigned long long c = comp[u];
                    const bool pred = c != 0ull && (c & mask) == prefix;
                    const unsigned bin = (unsigned)(c >> shift) & 255u;
                    // wave-aggregated: the lanes sharing the first active lane's bin cost ONE LDS atomic (on the leading byte
                    // nearly all keys of a level fall into two or three bins), the others add individually
                    const unsigned long long act = __ballot(pred);
                    if (act) {
                        const int leader = __ffsll((long long)act) - 1;
                        const unsigned lb = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
                        const bool same = pred && bin == lb;
                        const unsigned long long sm = __ballot(same);
                        if (lane == leader) atomicAdd(&sh.hist[lb], __popcll(sm));
                        if (pred && !same) atomicAdd(&sh.hist[bin], 1);
                    }
                }
            }
            __syncthreads();
            if (tid < 64) {
                // the bin where the count from the top reaches `need`: lane l owns bins 255 - 4 l .. 252 - 4 l, wave scan
                const int b0 = 255 - 4 * lane;
                const int h0 = sh.hist[b0], h1 = sh.hist[b0 - 1], h2 = sh.hist[b0 - 2], h3 = sh.hist[b0 - 3];
                const int own = h0 + h1 + h2 + h3;
                int incl = own;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int t = __shfl_up(incl, off);
                    if (lane >= off) incl += t;
                }
                const unsigned long long hit = __ballot(incl >= need);
                if (hit && lane == __ffsll((long long)hit) - 1) {
                    int cum = incl - own, b = b0, h = h0;
                    if (cum + h0 < need) {
                        cum += h0; b = b0 - 1; h = h1;
                        if (cum + h1 < need) {
                            cum += h1; b = b0 - 2; h = h2;
                            if (cum + h2 < need) { cum += h2; b = b0 - 3; h = h3; }
                        }
                    }
                    sh.prefix = prefix | ((unsigned long long)b << shift);
                    sh.need = need - cum;
                    sh.eqtotal = h;
                }
            }
            __syncthreads();
            prefix = sh.prefix;
            need = sh.need;
            mask |= 0xFFull << shift;
            if (need == sh.eqtotal) break;                           // every composite with this prefix is taken
        }
        T = prefix;
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        if (u < nslots) {
            const unsigned long long c = comp[u];
            const bool pred = c != 0ull && c >= T;
            const unsigned long long bal = __ballot(pred);
            if (bal) {
                const int leader = __ffsll((long long)bal) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&sh.nsel, __popcll(bal));
                base = __builtin_amdgcn_readlane(base, leader);
                const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (pred && pos < 1024) sh.sel[pos] = c;
            }
        }
    }
    __syncthreads();
    return k;
}

__global__ __launch_bounds__(1024) void decode_select_kernel(DecodeArgs a) {
    __shared__ SelectShared sh;
    const int chunk = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    int level = 0;
    while (level + 1 < a.L && chunk >= a.chunk_first[level + 1]) ++level;
    const int n = a.gh[level] * a.gw[level] * a.A * a.K;
    const int c0 = (chunk - a.chunk_first[level]) * DECODE_CHUNK;
    const int c1 = c0 + DECODE_CHUNK < n ? c0 + DECODE_CHUNK : n;
    const float* lg = a.logits[level] + (size_t)img * n;
    constexpr int R = DECODE_CHUNK / 1024;
    unsigned long long comp[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int i = c0 + u * 1024 + tid;
        const float l = lg[i < c1 ? i : c1 - 1];                     // (clamped, not predicated: the loads stay independent)
        const bool ok = i < c1 && score_passes(l, a.score_thresh, a.t_lo, a.t_hi);
        // ordered_key is 0 only for one NaN pattern, never a candidate
        comp[u] = ok ? (((unsigned long long)ordered_key(l) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i)) : 0ull;
    }
    const int k = select_topk_regs<R>(comp, R, a.topk, sh, tid);
    (void)k;
    // the whole 1024-entry run is written (zeros past the chunk's survivors): the merge kernel loads its runs without waiting
    // for their counts
    a.runs[((size_t)img * a.chunk_first[a.L] + chunk) * 1024 + tid] = sh.sel[tid];
}

__global__ __launch_bounds__(1024) void decode_merge_kernel(DecodeArgs a) {
    __shared__ SelectShared sh;
    const int level = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const int gh = a.gh[level], gw = a.gw[level];
    const int n = gh * gw * a.A * a.K;
    const float* lg = a.logits[level] + (size_t)img * n;
    const float* rg = a.regs[level] + (size_t)img * (size_t)(gh * gw * a.A) * 4;
    const int nch = a.chunk_first[level + 1] - a.chunk_first[level];         // <= DECODE_MAX_RUNS (host)
    const size_t run0 = (size_t)img * a.chunk_first[a.L] + a.chunk_first[level];
    unsigned long long comp[DECODE_MAX_RUNS];
#pragma unroll
    for (int u = 0; u < DECODE_MAX_RUNS; ++u) {
        comp[u] = 0ull;
        if (u < nch) comp[u] = a.runs[(run0 + u) * 1024 + tid];    // (independent loads, all in flight together)
    }
    const int k = select_topk_regs<DECODE_MAX_RUNS>(comp, nch, a.topk, sh, tid);
    bitonic_sort1024_desc(sh.sel, tid);

    const int slot = a.L * a.topk;
    if (tid == 0) a.cand_count[img * a.L + level] = k;
    if (tid < k) {
        const unsigned long long c = sh.sel[tid];
        const int idx = (int)(0xFFFFFFFFu - (unsigned)(c & 0xFFFFFFFFull));
        const float l = lg[idx];
        const int aidx = idx / a.K, label = idx - aidx * a.K;
        const int cell = aidx / a.A, an = aidx - cell * a.A;
        const int y = cell / gw, x = cell - y * gw;
        const float shx = (float)x * (float)a.sw[level], shy = (float)y * (float)a.sh[level];
        const float* ba = a.base_anchors + ((size_t)level * a.A + an) * 4;
        const float ax1 = shx + ba[0], ay1 = shy + ba[1], ax2 = shx + ba[2], ay2 = shy + ba[3];
        const float w = ax2 - ax1, h = ay2 - ay1;
        const float cx = ax1 + 0.5f * w, cy = ay1 + 0.5f * h;
        const float4 r = *reinterpret_cast<const float4*>(rg + (size_t)aidx * 4);
        const float dw = fminf(r.z, a.xform_clip), dh = fminf(r.w, a.xform_clip);
        const float pcx = r.x * w + cx, pcy = r.y * h + cy;
        const float pw = expf(dw) * w, ph = expf(dh) * h;
        float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
        const float ih = (float)a.image_hw[img * 2 + 0], iw = (float)a.image_hw[img * 2 + 1];
        x1 = fminf(fmaxf(x1, 0.f), iw); x2 = fminf(fmaxf(x2, 0.f), iw);
        y1 = fminf(fmaxf(y1, 0.f), ih); y2 = fminf(fmaxf(y2, 0.f), ih);
        const size_t o = (size_t)img * slot + (size_t)level * a.topk + tid;
        *reinterpret_cast<float4*>(a.cand_boxes + o * 4) = make_float4(x1, y1, x2, y2);
        a.cand_scores[o] = sigmoidf_ref(l);
        a.cand_logits[o] = l;
        a.cand_labels[o] = label;
    }
}

// ---------------------------------------------------------------------------
struct NmsArgs {
    const float* cand_boxes; const float* cand_scores; const float* cand_logits; const int* cand_labels;
    const int* cand_count;
    int L, N, topk, Tmax, words;       // Tmax = roundup(L*topk, 64); words = Tmax/64
    float nms_thresh, conf_thresh;
    int max_keep;                      // detections_per_img
    const float* ratios;               // [N][2] (ratio_h, ratio_w) = orig / resized
    float* s_boxes; float* s_scores; int* s_labels; float* s_off; int* s_total;   // sorted scratch
    unsigned long long* mask;          // [N][Tmax][words]
    float* out_boxes; float* out_scores; long long* out_labels; int* out_count; int* out_conf_count;
    // Two-phase NMS: greedy NMS never looks past the candidate that completes detections_per_img, so phase 1 runs mask + scan
    // on the first `limit` candidates only; an image whose kept boxes reach max_keep there (or that has no more candidates) is
    // final (done[img] = 1), and phase 2 -- the same two kernels over all candidates -- returns at once for it.
    int limit;                         // candidates considered (>= Tmax: all)
    int phase;                         // 0: the only phase; 1: first of two (sets done); 2: second (skips done images)
    int* done;                         // [N]
};

__global__ __launch_bounds__(1024) void nms_sort_kernel(NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned keys[];   // L segments of 1024 logit keys (<= 32 KiB)
    __shared__ int offs[MAX_LEVELS + 1];
    __shared__ float s_max[16];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int slot = a.L * a.topk;
    if (tid == 0) {
        int t = 0;
        for (int l = 0; l < a.L; ++l) { offs[l] = t; t += a.cand_count[img * a.L + l]; }
        offs[a.L] = t;
    }
    __syncthreads();
    const int T = offs[a.L];
    // Each level's candidates arrive sorted (decode: logit descending, then lower anchor index): the keys of level l are laid
    // into the 1024-entry segment l of LDS and every candidate computes its RANK in the merged order -- its rank in its own
    // level plus, by binary search, the number of candidates of every other level that precede it -- and is copied straight to
    // that position (a bitonic merge of the segments took 36 network stages over 8192 64-bit keys).  Equal logits: the lower
    // level first, then the lower rank, which is the order of the concatenated candidate list -- so a candidate of level l
    // counts the keys >= its own in the levels below l and the keys > its own in the levels above.
    float mx = -INFINITY;
    for (int i = tid; i < a.L * 1024; i += 1024) {
        unsigned kv = 0u;
        const int l = i >> 10, r = i & 1023;
        if (r < offs[l + 1] - offs[l]) {
            const int src = l * a.topk + r;
            kv = ordered_key(a.cand_logits[(size_t)img * slot + src]);
            const float4 b = *reinterpret_cast<const float4*>(a.cand_boxes + ((size_t)img * slot + src) * 4);
            mx = fmaxf(mx, fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
        }
        keys[i] = kv;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((tid & 63) == 0) s_max[tid >> 6] = mx;
    __syncthreads();
    mx = s_max[0];
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, s_max[w]);
    if (tid == 0) a.s_total[img] = T;
    for (int l = 0; l < a.L; ++l) {                                  // (i = l * 1024 + tid: the level is uniform per iteration)
        const int r = tid;
        if (r >= offs[l + 1] - offs[l]) continue;
        const unsigned kv = keys[l * 1024 + r];
        int rank = r;
        for (int m = 0; m < a.L; ++m) {
            if (m == l) continue;
            const unsigned* seg = keys + m * 1024;
            const unsigned bound = m < l ? kv - 1u : kv;             // count keys > bound (kv >= 1: ordered_key of a candidate)
            int lo = 0, hi = offs[m + 1] - offs[m];
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (seg[mid] > bound) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        const size_t src = (size_t)img * slot + l * a.topk + r;
        const size_t dst = (size_t)img * a.Tmax + rank;
        *reinterpret_cast<float4*>(a.s_boxes + dst * 4) = *reinterpret_cast<const float4*>(a.cand_boxes + src * 4);
        a.s_scores[dst] = a.cand_scores[src];
        const int lab = a.cand_labels[src];
        a.s_labels[dst] = lab;
        a.s_off[dst] = (float)lab * (mx + 1.f);   // batched_nms class offset (0 for num_classes == 1)
    }
}

__global__ __launch_bounds__(64) void nms_mask_kernel(NmsArgs a) {
    const int cb = blockIdx.x, rb = blockIdx.y, img = blockIdx.z;
    if (cb < rb || (a.phase == 2 && a.done[img])) return;
    const int T = a.s_total[img] < a.limit ? a.s_total[img] : a.limit;
    if (rb * 64 >= T || cb * 64 >= T) return;
    __shared__ float4 cbox[64];
    __shared__ float carea[64];
    const int t = threadIdx.x;
    {
        const int j = cb * 64 + t;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < T) {
            b = *reinterpret_cast<const float4*>(a.s_boxes + ((size_t)img * a.Tmax + j) * 4);
            const float o = a.s_off[(size_t)img * a.Tmax + j];
            b.x += o; b.y += o; b.z += o; b.w += o;
        }
        cbox[t] = b;
        carea[t] = (b.z - b.x) * (b.w - b.y);
    }
    __syncthreads();
    const int i = rb * 64 + t;
    if (i >= T) return;
    float4 bi = *reinterpret_cast<const float4*>(a.s_boxes + ((size_t)img * a.Tmax + i) * 4);
    {
        const float o = a.s_off[(size_t)img * a.Tmax + i];
        bi.x += o; bi.y += o; bi.z += o; bi.w += o;
    }
    const float area_i = (bi.z - bi.x) * (bi.w - bi.y);
    const float thr = a.nms_thresh;
    // iou = inter / uni > thr, as the oracle evaluates it -- but the division runs only where rounding could decide: p = fl(thr *
    // uni) is within 2^-24 of thr * uni, so an `inter` beyond p (1 +- 4e-6) fixes the side of thr the rounded quotient falls on
    // (0 <= thr <= 1; other thresholds always divide).  The wave divides only if one of its lanes is inside the band.
    const bool shortcut = thr >= 0.f && thr <= 1.f;
    unsigned long long bits = 0ull;
#pragma unroll 8
    for (int jj = 0; jj < 64; ++jj) {
        const int j = cb * 64 + jj;
        const float4 bj = cbox[jj];
        const float area_j = carea[jj];
        const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
        const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
        const float iw = fmaxf(xx2 - xx1, 0.f), ih = fmaxf(yy2 - yy1, 0.f);
        const float inter = iw * ih;
        const float uni = area_i + area_j - inter;
        const float p = thr * uni;
        const bool yes = inter > p * 1.000004f, no = inter < p * 0.999996f;
        bool over = yes;
        if (__ballot(!(shortcut && (yes || no)))) over = inter / uni > thr;
        if (over && j > i && j < T) bits |= (1ull << jj);
    }
    a.mask[((size_t)img * a.Tmax + i) * a.words + cb] = bits;
}

// Greedy scan over the IoU bit matrix, one 1024-thread workgroup per image.  The chunk-to-chunk dependency is serial by
// nature (a box survives iff no KEPT higher-ranked box suppresses it), so the work per 64-box chunk is kept short:
//   * wave 0 resolves the chunk: instead of visiting all 64 boxes it jumps from one surviving box to the next
//     (`avail` = not yet suppressed), OR-ing that box's diagonal word into the suppressed set -- as many steps as boxes are
//     kept in the chunk (a handful), the diagonal words fetched with v_readlane at a scalar index;
//   * all 16 waves then OR the kept rows into the removed-bits of the later words: thread (g, w) owns word w and the kept
//     boxes 8g .. 8g+7 of the chunk, issues its 8 row loads together (independent, coalesced over w) and merges with one
//     LDS atomic OR -- the one-wave version fetched the kept rows one after the other, ~0.5 us of L2 latency each.
__global__ __launch_bounds__(1024) void nms_scan_kernel(NmsArgs a) {
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.phase == 2 && a.done[img]) return;
    const int T = a.s_total[img] < a.limit ? a.s_total[img] : a.limit;
    const int nw = (T + 63) / 64;
    __shared__ int keep[SORT_CAP];
    __shared__ unsigned long long rem[SORT_CAP / 64];      // removed bits per 64-box word
    __shared__ unsigned long long s_kept;
    __shared__ int s_nkept;
    if (tid < SORT_CAP / 64) rem[tid] = 0ull;
    if (tid == 0) s_nkept = 0;
    __syncthreads();
    const unsigned long long* M = a.mask + (size_t)img * a.Tmax * a.words;
    const int w = tid & 127, g = tid >> 7;                // word owned in the propagation step, group of 8 chunk rows
    // (the diagonal word of chunk b + 1 is fetched while chunk b is resolved and propagated: its L2 latency leaves the chain)
    unsigned long long diag_next = (wave == 0 && lane < T) ? M[(size_t)lane * a.words] : 0ull;
    // The rows a chunk may have to propagate -- thread (g, w): word w of its rows 8 g .. 8 g + 7 -- are fetched THREE chunks ahead
    // into a register ring, whether the boxes turn out kept or not (40 KiB of L2 reads per chunk): fetched after the chunk was
    // resolved, their ~1 us round trip was the length of a chunk step (105 us for the 79 chunks of dpi 1000).
    unsigned long long pv[4][8];
    auto fetch_rows = [&](int bb, unsigned long long (&dst)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int r = bb * 64 + 8 * g + j;
            r = r < T ? r : T - 1;                          // (rows past the end are never kept: the value is masked out)
            dst[j] = (bb < nw && w > bb && w < nw) ? M[(size_t)r * a.words + w] : 0ull;
        }
    };
    fetch_rows(0, pv[0]);
    fetch_rows(1, pv[1]);
    fetch_rows(2, pv[2]);
    bool full = false;
    for (int b4 = 0; b4 < nw && !full; b4 += 4) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int b = b4 + s4;
            if (b >= nw) break;
            fetch_rows(b + 3, pv[(s4 + 3) & 3]);
            if (wave == 0) {
                const int row = b * 64 + lane;
                const unsigned long long diag = diag_next;
                {
                    int rn = row + 64;
                    rn = rn < T ? rn : T - 1;
                    const unsigned long long nx = M[(size_t)rn * a.words + (b + 1 < nw ? b + 1 : b)];
                    diag_next = (row + 64 < T) ? nx : 0ull;
                }
                const unsigned dlo = (unsigned)(diag & 0xFFFFFFFFull), dhi = (unsigned)(diag >> 32);
                const int rows_here = (T - b * 64) < 64 ? (T - b * 64) : 64;
                const unsigned long long valid = rows_here == 64 ? ~0ull : ((1ull << rows_here) - 1ull);
                unsigned long long cur = rem[b];                // uniform
                unsigned long long kept = 0ull;
                unsigned long long avail = ~cur & valid;
                while (avail) {
                    const int kk = __ffsll((long long)avail) - 1;
                    kept |= 1ull << kk;
                    const unsigned lo = __builtin_amdgcn_readlane(dlo, kk), hi = __builtin_amdgcn_readlane(dhi, kk);
                    cur |= ((unsigned long long)hi << 32) | lo;   // the diagonal word holds only bits above kk
                    avail = ~cur & valid & ~((2ull << kk) - 1ull);
                }
                const int nkept = s_nkept;
                if ((kept >> lane) & 1ull) {
                    const int pos = nkept + __popcll(kept & ((1ull << lane) - 1ull));
                    if (pos < SORT_CAP) keep[pos] = row;
                }
                if (lane == 0) { s_kept = kept; s_nkept = nkept + __popcll(kept); }
            }
            __syncthreads();
            const unsigned long long kept = s_kept;
            if (s_nkept >= a.max_keep) { full = true; break; }
            // propagate the kept rows of this chunk to the later words
            if (w > b && w < nw) {
                const unsigned sub = (unsigned)(kept >> (8 * g)) & 0xFFu;
                unsigned long long acc = 0ull;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc |= ((sub >> j) & 1u) ? pv[s4][j] : 0ull;
                if (acc) atomicOr(&rem[w], acc);
            }
            __syncthreads();
        }
    }
    __syncthreads();
    const int nkept = s_nkept;
    if (a.phase == 1) {
        // final iff detections_per_img boxes are kept (what follows cannot change them) or there are no further candidates
        const bool final = nkept >= a.max_keep || a.s_total[img] <= a.limit;
        if (tid == 0) a.done[img] = final ? 1 : 0;
        if (!final) return;
    }
    const int nout = nkept < a.max_keep ? nkept : a.max_keep;
    const float rh = a.ratios[img * 2 + 0], rw = a.ratios[img * 2 + 1];
    __shared__ int s_conf;
    if (tid == 0) s_conf = 0;
    __syncthreads();
    int nconf = 0;
    for (int r = tid; r < nout; r += 1024) {
        const size_t src = (size_t)img * a.Tmax + keep[r];
        const float4 bx = *reinterpret_cast<const float4*>(a.s_boxes + src * 4);
        const size_t dst = (size_t)img * a.max_keep + r;
        *reinterpret_cast<float4*>(a.out_boxes + dst * 4) = make_float4(bx.x * rw, bx.y * rh, bx.z * rw, bx.w * rh);
        const float sc = a.s_scores[src];
        a.out_scores[dst] = sc;
        a.out_labels[dst] = (long long)a.s_labels[src];
        nconf += (sc > a.conf_thresh) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nconf += __shfl_xor(nconf, off);
    if (lane == 0 && nconf) atomicAdd(&s_conf, nconf);
    __syncthreads();
    if (tid == 0) { a.out_count[img] = nout; a.out_conf_count[img] = s_conf; }
}

extern "C" size_t cvpce_detect_workspace_bytes(int N, int L, int topk) {
    const size_t slot = (size_t)L * topk;
    const size_t Tmax = (slot + 63) / 64 * 64, words = Tmax / 64;
    size_t b = 0;
    b += (size_t)N * slot * (4 + 1 + 1 + 1) * 4;      // cand boxes/scores/logits/labels
    b += (size_t)N * L * 4;                            // cand_count
    b += (size_t)N * Tmax * (4 + 1 + 1 + 1) * 4;       // sorted boxes/scores/labels/off
    b += (size_t)N * 4;                                // s_total
    b += (size_t)N * Tmax * words * 8;                 // mask
    b += (size_t)N * L * DECODE_RUNS_PER_LEVEL * 1024 * 8;   // chunk runs
    b += (size_t)N * 4;                                // done flags of the two-phase NMS
    return b + 16 * 256;   // every sub-allocation is rounded up to 256 B
}

extern "C" int cvpce_detect_postprocess(const float* const* logits, const float* const* regs, const int* gh,
                                        const int* gw, const int* stride_h, const int* stride_w,
                                        const float* base_anchors, const int* image_hw, const float* ratios, int L,
                                        int N, int A, int K, int topk, float score_thresh, float nms_thresh,
                                        float xform_clip, int detections_per_img, float conf_thresh, void* workspace,
                                        size_t workspace_bytes, float* out_boxes, float* out_scores,
                                        long long* out_labels, int* out_count, int* out_conf_count, void* stream) {
    if (!logits || !regs || !gh || !gw || !stride_h || !stride_w || !base_anchors || !image_hw || !ratios ||
        !workspace || !out_boxes || !out_scores || !out_labels || !out_count || !out_conf_count)
        return CVPCE_ERR_ARG;
    if (L < 1 || L > MAX_LEVELS || A < 1 || K < 1 || topk < 1 || topk > 1024 || detections_per_img < 1)
        return CVPCE_ERR_ARG;
    if ((size_t)L * topk > SORT_CAP || detections_per_img > SORT_CAP) return CVPCE_ERR_ARG;
    if (N <= 0) return CVPCE_OK;
    if (workspace_bytes < cvpce_detect_workspace_bytes(N, L, topk)) return CVPCE_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t slot = (size_t)L * topk;
    const size_t Tmax = (slot + 63) / 64 * 64, words = Tmax / 64;
    char* p = (char*)workspace;
    auto take = [&](size_t bytes) { char* q = p; p += (bytes + 255) / 256 * 256; return (void*)q; };
    // mask first: keeps 8-byte alignment trivially
    unsigned long long* mask = (unsigned long long*)take((size_t)N * Tmax * words * 8);
    float* cand_boxes = (float*)take((size_t)N * slot * 16);
    float* cand_scores = (float*)take((size_t)N * slot * 4);
    float* cand_logits = (float*)take((size_t)N * slot * 4);
    int* cand_labels = (int*)take((size_t)N * slot * 4);
    int* cand_count = (int*)take((size_t)N * L * 4);
    float* s_boxes = (float*)take((size_t)N * Tmax * 16);
    float* s_scores = (float*)take((size_t)N * Tmax * 4);
    int* s_labels = (int*)take((size_t)N * Tmax * 4);
    float* s_off = (float*)take((size_t)N * Tmax * 4);
    int* s_total = (int*)take((size_t)N * 4);
    const size_t max_runs = (size_t)L * DECODE_RUNS_PER_LEVEL;          // per image
    unsigned long long* runs = (unsigned long long*)take((size_t)N * max_runs * 1024 * 8);
    int* done = (int*)take((size_t)N * 4);
    if ((size_t)(p - (char*)workspace) > workspace_bytes) return CVPCE_ERR_ARG;

    DecodeArgs d;
    for (int l = 0; l < L; ++l) {
        if (!logits[l] || !regs[l] || gh[l] < 1 || gw[l] < 1) return CVPCE_ERR_ARG;
        d.logits[l] = logits[l]; d.regs[l] = regs[l];
        d.gh[l] = gh[l]; d.gw[l] = gw[l]; d.sh[l] = stride_h[l]; d.sw[l] = stride_w[l];
    }
    d.base_anchors = base_anchors; d.image_hw = image_hw; d.L = L; d.N = N; d.A = A; d.K = K; d.topk = topk;
    d.score_thresh = score_thresh; d.xform_clip = xform_clip;
    d.cand_boxes = cand_boxes; d.cand_scores = cand_scores; d.cand_logits = cand_logits; d.cand_labels = cand_labels;
    d.cand_count = cand_count;
    // chunked form unless a level is too large for one merge workgroup or the image's chunks exceed the workspace's runs
    bool chunked = true;
    int chunks = 0;
    for (int l = 0; l < L; ++l) {
        const long long n = (long long)gh[l] * gw[l] * A * K;
        const long long nch = (n + DECODE_CHUNK - 1) / DECODE_CHUNK;
        if (n > 0x7FFFFFFFll) return CVPCE_ERR_ARG;
        if (nch > DECODE_MAX_RUNS) chunked = false;
        d.chunk_first[l] = chunks;
        chunks += (int)(nch < DECODE_MAX_RUNS ? nch : DECODE_MAX_RUNS);
    }
    d.chunk_first[L] = chunks;
    if ((size_t)chunks > max_runs) chunked = false;
    d.runs = runs;
    d.t_lo = -INFINITY; d.t_hi = INFINITY;                             // (outside this band of thresholds: always the sigmoid)
    if (score_thresh >= 1e-6f && score_thresh <= 0.99f) {
        const double t = log((double)score_thresh / (1.0 - (double)score_thresh)), m = 1e-3 * (1.0 + fabs(t));
        d.t_lo = (float)(t - m); d.t_hi = (float)(t + m);
    }
    if (chunked && !getenv("CVPCE_DECODE_ONE_KERNEL")) {
        hipLaunchKernelGGL(decode_select_kernel, dim3(chunks, N), dim3(1024), 0, s, d);
        hipLaunchKernelGGL(decode_merge_kernel, dim3(L, N), dim3(1024), 0, s, d);
    } else {
        hipLaunchKernelGGL(decode_topk_kernel, dim3(L, N), dim3(1024), 0, s, d);
    }

    NmsArgs n;
    n.cand_boxes = cand_boxes; n.cand_scores = cand_scores; n.cand_logits = cand_logits; n.cand_labels = cand_labels;
    n.cand_count = cand_count; n.L = L; n.N = N; n.topk = topk; n.Tmax = (int)Tmax; n.words = (int)words;
    n.nms_thresh = nms_thresh; n.conf_thresh = conf_thresh; n.max_keep = detections_per_img; n.ratios = ratios;
    n.s_boxes = s_boxes; n.s_scores = s_scores; n.s_labels = s_labels; n.s_off = s_off; n.s_total = s_total;
    n.mask = mask; n.out_boxes = out_boxes; n.out_scores = out_scores; n.out_labels = out_labels;
    n.out_count = out_count; n.out_conf_count = out_conf_count;
    hipLaunchKernelGGL(nms_sort_kernel, dim3(N), dim3(1024), (size_t)L * 1024 * 4, s, n);
    // phase 1 on the first 8 x detections_per_img candidates (>= 1024), phase 2 on all of them for the images that need it
    size_t limit = (size_t)detections_per_img * 8;
    limit = ((limit < 1024 ? 1024 : limit) + 63) / 64 * 64;
    n.done = done;
    if (limit >= Tmax || getenv("CVPCE_NMS_ONE_PHASE")) {
        n.limit = (int)Tmax; n.phase = 0;
        hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)words, (unsigned)words, N), dim3(64), 0, s, n);
        hipLaunchKernelGGL(nms_scan_kernel, dim3(N), dim3(1024), 0, s, n);
    } else {
        const unsigned w1 = (unsigned)(limit / 64);
        n.limit = (int)limit; n.phase = 1;
        hipLaunchKernelGGL(nms_mask_kernel, dim3(w1, w1, N), dim3(64), 0, s, n);
        hipLaunchKernelGGL(nms_scan_kernel, dim3(N), dim3(1024), 0, s, n);
        n.limit = (int)Tmax; n.phase = 2;
        hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)words, (unsigned)words, N), dim3(64), 0, s, n);
        hipLaunchKernelGGL(nms_scan_kernel, dim3(N), dim3(1024), 0, s, n);
    }
    return cvpce_check_launch();
}
