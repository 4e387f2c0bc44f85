// NMS family for the detect -> crop hand-off.
//
//   yv_custom_nms        README.md:62-84          class-agnostic greedy NMS, bit-exact keep list
//   yv_efficient_nms     tech.md:41-47            EfficientNMS_TRT output contract (KAT-2)
//   yv_postprocess_dets  解读.md:82-99 + README.md:41 + utils/trainClass.py:70-93
//   yv_compact_crops     device-side batch assembly (no host sync between stages)
//
// One workgroup owns one image / one box set: the greedy scan is sequential by
// definition, so the parallelism is (sets) x (candidates of a set).  Scores are
// ordered with a 64-bit key {~orderable(score), index}: every key is distinct,
// so the order (score desc, index asc) is total and the result deterministic.
// All IoU arithmetic is f32 with one rounding per operation (explicit _rn
// intrinsics: no FMA contraction, IEEE divide) in torchvision's operation
// order, which is what makes the keep list bit-exact against the reference.
#include "yv_common.h"
#include <atomic>

namespace {

__device__ __forceinline__ uint32_t desc_key(float s) {
    uint32_t u = __float_as_uint(s);
    if (u == 0x80000000u) u = 0u;                                   // -0 == +0 (tie -> index order)
    uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // order-preserving map
    return ~asc;                                                    // smaller key = larger score
}

__device__ __forceinline__ float box_area(float4 b) { return __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y)); }

__device__ __forceinline__ float iou_f32(float4 a, float area_a, float4 b, float area_b) {
    float ltx = fmaxf(a.x, b.x), lty = fmaxf(a.y, b.y);
    float rbx = fminf(a.z, b.z), rby = fminf(a.w, b.w);
    float w = fmaxf(__fsub_rn(rbx, ltx), 0.0f), h = fmaxf(__fsub_rn(rby, lty), 0.0f);
    float inter = __fmul_rn(w, h);
    float uni = __fsub_rn(__fadd_rn(area_a, area_b), inter);
    return __fdiv_rn(inter, uni);
}

__device__ __forceinline__ int next_pow2(int v, int lo) {
    int p = lo;
    while (p < v) p <<= 1;
    return p;
}

// ascending bitonic sort of np (power of two) u64 keys held in LDS
__device__ void bitonic_sort_u64(uint64_t* k, int np) {
    for (int size = 2; size <= np; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (np >> 1); t += blockDim.x) {
                int lo = 2 * t - (t & (stride - 1));
                int hi = lo + stride;
                bool asc = ((lo & size) == 0);
                uint64_t a = k[lo], b = k[hi];
                if ((a > b) == asc) { k[lo] = b; k[hi] = a; }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void alive_init(uint64_t* alive, int n, int nwords) {
    for (int w = threadIdx.x; w < nwords; w += blockDim.x) {
        int lo = w * 64;
        uint64_t m = 0;
        if (n >= lo + 64) m = ~0ull;
        else if (n > lo) m = (~0ull) >> (64 - (n - lo));
        alive[w] = m;
    }
}

__device__ __forceinline__ int next_alive(const uint64_t* alive, int cur, int n, int nwords) {
    for (int w = cur >> 6; w < nwords; ++w) {
        uint64_t m = alive[w];
        if (w == (cur >> 6)) m &= (~0ull) << (cur & 63);
        if (m) {
            int i = w * 64 + __builtin_ctzll(m);
            return i < n ? i : n;
        }
    }
    return n;
}

// greedy scan over candidates already in (score desc) order.
//   MODE 0 (custom_nms):   a later box survives a kept box iff iou <  thr (NaN does not survive)
//   MODE 1 (efficient):    a later box of the SAME class is dropped iff iou > thr
// on_keep(rank, sorted_pos) is called by thread 0 for every kept box. Returns #kept (uniform).
template <int MODE, typename OnKeep>
__device__ int greedy_scan(const float4* sb, const int16_t* cls, int n, float thr, int max_keep, uint64_t* alive,
                           int nwords, OnKeep on_keep) {
    int nk = 0, cur = 0;
    while (nk < max_keep) {
        int i = next_alive(alive, cur, n, nwords);
        if (i >= n) break;
        if (threadIdx.x == 0) on_keep(nk, i);
        ++nk;
        float4 bi = sb[i];
        float ai = box_area(bi);
        int ci = MODE == 1 ? (int)cls[i] : 0;
        for (int j = i + 1 + threadIdx.x; j < n; j += blockDim.x) {
            if ((alive[j >> 6] >> (j & 63)) & 1ull) {
                float4 bj = sb[j];
                bool drop;
                if (MODE == 0) {
                    drop = !(iou_f32(bi, ai, bj, box_area(bj)) < thr);
                } else {
                    drop = ((int)cls[j] == ci) && (iou_f32(bi, ai, bj, box_area(bj)) > thr);
                }
                if (drop) atomicAnd((unsigned long long*)&alive[j >> 6], ~(1ull << (j & 63)));
            }
        }
        __syncthreads();
        cur = i + 1;
    }
    return nk;
}

// ---------------------------------------------------------------------------------------------
// custom_nms: one workgroup per set.  LDS: keys[np] | alive[np/64] | boxes[np] (when they fit)
// ---------------------------------------------------------------------------------------------
constexpr int CN_LDS_BOX_MAX = 4096;

__global__ void custom_nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                  const int32_t* __restrict__ counts, int n_max, int np_max, float thr,
                                  int32_t* __restrict__ keep, int32_t* __restrict__ num_keep, float4* ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x;
    int n = counts ? counts[s] : n_max;
    n = n < 0 ? 0 : (n > n_max ? n_max : n);
    uint64_t* keys = (uint64_t*)smem;
    uint64_t* alive = keys + np_max;
    float4* sb = (np_max <= CN_LDS_BOX_MAX) ? (float4*)(alive + (((np_max >> 6) + 2) & ~1)) : (ws + (size_t)s * n_max);
    const float4* B = (const float4*)boxes + (size_t)s * n_max;
    const float* S = scores + (size_t)s * n_max;
    int32_t* K = keep + (size_t)s * n_max;

    const int np = next_pow2(n, 64);
    for (int i = threadIdx.x; i < np; i += blockDim.x)
        keys[i] = i < n ? (((uint64_t)desc_key(S[i]) << 32) | (uint32_t)i) : ~0ull;
    for (int i = threadIdx.x; i < n_max; i += blockDim.x) K[i] = -1;
    bitonic_sort_u64(keys, np);
    for (int i = threadIdx.x; i < n; i += blockDim.x) sb[i] = B[(uint32_t)keys[i]];
    const int nwords = (np + 63) >> 6;
    alive_init(alive, n, nwords);
    __threadfence_block();
    __syncthreads();
    int nk = greedy_scan<0>(sb, nullptr, n, thr, n, alive, nwords,
                            [&](int rank, int pos) { K[rank] = (int32_t)(uint32_t)keys[pos]; });
    if (threadIdx.x == 0) num_keep[s] = nk;
}

// ---------------------------------------------------------------------------------------------
// EfficientNMS contract: one workgroup (1024 threads) per image.
// ---------------------------------------------------------------------------------------------
constexpr int EN_MAXK = 4096;
constexpr int EN_THREADS = 1024;

struct EnShared {
    uint32_t hist[2048];
    uint32_t wave_cnt[16];
    uint32_t sel_bin, sel_before, n_sel, n_eq_taken, total_cnt;
};

__device__ __forceinline__ uint32_t block_count(bool flag, uint32_t* wave_cnt) {
    uint64_t m = __ballot(flag);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) wave_cnt[wave] = (uint32_t)__builtin_popcountll(m);
    __syncthreads();
    uint32_t t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wave_cnt[w];
    return t;
}

// exclusive rank of `flag` among the block's threads (thread order) + block total
__device__ __forceinline__ uint32_t block_rank(bool flag, uint32_t* wave_cnt, uint32_t* total) {
    uint64_t m = __ballot(flag);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) wave_cnt[wave] = (uint32_t)__builtin_popcountll(m);
    __syncthreads();
    uint32_t before = 0, t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        uint32_t c = wave_cnt[w];
        if (w < wave) before += c;
        t += c;
    }
    *total = t;
    return before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
}

__global__ __launch_bounds__(EN_THREADS) void efficient_nms_kernel(
    const float* __restrict__ boxes, const float* __restrict__ scores, int A, int nc, float score_thr, float iou_thr,
    int max_out, int pre_topk, int32_t* __restrict__ num_dets, float* __restrict__ out_boxes,
    float* __restrict__ out_scores, int32_t* __restrict__ out_labels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = (uint64_t*)smem;                          // EN_MAXK
    float4* sb = (float4*)(keys + EN_MAXK);                    // EN_MAXK
    int16_t* cls = (int16_t*)(sb + EN_MAXK);                   // EN_MAXK
    uint64_t* alive = (uint64_t*)(cls + EN_MAXK);              // EN_MAXK/64
    EnShared* sh = (EnShared*)(alive + EN_MAXK / 64);

    const int b = blockIdx.x, tid = threadIdx.x;
    const int total = A * nc;
    const float* S = scores + (size_t)b * total;
    const float4* Bx = (const float4*)boxes + (size_t)b * A;
    const int chunks = (total + EN_THREADS - 1) / EN_THREADS;

    for (int i = tid; i < max_out; i += EN_THREADS) {
        ((float4*)out_boxes)[(size_t)b * max_out + i] = make_float4(0.f, 0.f, 0.f, 0.f);
        out_scores[(size_t)b * max_out + i] = 0.f;
        out_labels[(size_t)b * max_out + i] = 0;
    }

    // every thread caches the keys of its candidates in registers (A*nc <= 48K: YOLOv8 at 640 with nc <= 5);
    // larger problems re-read the scores from L2 each pass
    constexpr int EN_CACHE = 48;
    const bool cached = chunks <= EN_CACHE;
    uint32_t kreg[EN_CACHE];
    uint32_t mine = 0;
#pragma unroll
    for (int c = 0; c < EN_CACHE; ++c) {
        kreg[c] = 0xFFFFFFFFu;                                    // sentinel: not a candidate
        const int i = c * EN_THREADS + tid;
        if (c < chunks && i < total) {
            const float v = S[i];
            if (v > score_thr) { kreg[c] = desc_key(v); ++mine; }
        }
    }
    if (!cached) {
        mine = 0;
        for (int c = 0; c < chunks; ++c) {
            const int i = c * EN_THREADS + tid;
            if (i < total && S[i] > score_thr) ++mine;
        }
    }
    if (tid == 0) sh->total_cnt = 0;
    __syncthreads();
    if (mine) atomicAdd(&sh->total_cnt, mine);
    __syncthreads();
    const uint32_t cnt = sh->total_cnt;
    const uint32_t K = (uint32_t)pre_topk;

    // binary radix select of the K-th best 32-bit score key: 32 counting passes over register-resident keys
    // (no histogram: random-weight detectors put every score into a handful of bins and LDS-atomic
    // histograms serialise on them).  Result: key_star and how many of the keys == key_star to take.
    uint32_t key_star = 0xFFFFFFFFu, need_eq = 0xFFFFFFFFu;      // take everything by default
    const bool select = cnt > K;
    if (select) {
        uint32_t prefix = 0, need = K;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t hi_mask = bit == 31 ? 0u : (0xFFFFFFFFu << (bit + 1));
            uint32_t c0 = 0;
            if (cached) {
#pragma unroll
                for (int c = 0; c < EN_CACHE; ++c) {
                    const uint32_t k = kreg[c];
                    const bool cand = k != 0xFFFFFFFFu;        // a score > threshold is never NaN, so no real key is ~0
                    c0 += (cand && ((k & hi_mask) == prefix) && !((k >> bit) & 1u)) ? 1u : 0u;
                }
            } else {
                for (int c = 0; c < chunks; ++c) {
                    const int i = c * EN_THREADS + tid;
                    if (i < total) {
                        const float v = S[i];
                        if (v > score_thr) {
                            const uint32_t k = desc_key(v);
                            c0 += (((k & hi_mask) == prefix) && !((k >> bit) & 1u)) ? 1u : 0u;
                        }
                    }
                }
            }
            // block sum (uniform result)
            uint32_t w = c0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
            __syncthreads();
            if ((tid & 63) == 0) sh->wave_cnt[tid >> 6] = w;
            __syncthreads();
            uint32_t t0 = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t0 += sh->wave_cnt[q];
            if (need > t0) { need -= t0; prefix |= (1u << bit); }
        }
        key_star = prefix;
        need_eq = need;              // how many of the candidates with key == key_star to take (index order)
    }

    // compaction of the selected set into LDS keys (set is exact; order fixed by the sort below)
    if (tid == 0) { sh->n_sel = 0; sh->n_eq_taken = 0; }
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        int i = c * EN_THREADS + tid;
        bool pass = false, eq = false;
        uint32_t k = 0;
        if (i < total) {
            float v = S[i];
            if (v > score_thr) {
                k = desc_key(v);
                pass = !select || k < key_star;
                eq = select && k == key_star;
            }
        }
        bool take = pass;
        if (select) {                                           // ties on the cut score: lowest flat index first
            uint32_t tot;
            uint32_t r = block_rank(eq, sh->wave_cnt, &tot);
            uint32_t base = sh->n_eq_taken;
            if (eq && base + r < need_eq) take = true;
            __syncthreads();
            if (tid == 0) sh->n_eq_taken = base + tot;
        }
        if (take) {
            uint32_t pos = atomicAdd(&sh->n_sel, 1u);
            if (pos < (uint32_t)EN_MAXK) keys[pos] = ((uint64_t)k << 32) | (uint32_t)i;
        }
        if (select) __syncthreads();
    }
    __syncthreads();
    int n = (int)min(sh->n_sel, (uint32_t)EN_MAXK);
    const int np = next_pow2(n, 64);
    for (int i = n + tid; i < np; i += EN_THREADS) keys[i] = ~0ull;
    bitonic_sort_u64(keys, np);
    for (int i = tid; i < n; i += EN_THREADS) {
        uint32_t flat = (uint32_t)keys[i];
        sb[i] = Bx[flat / (uint32_t)nc];
        cls[i] = (int16_t)(flat % (uint32_t)nc);
    }
    const int nwords = (np + 63) >> 6;
    alive_init(alive, n, nwords);
    __syncthreads();
    int nk = greedy_scan<1>(sb, cls, n, iou_thr, max_out, alive, nwords, [&](int rank, int pos) {
        uint32_t flat = (uint32_t)keys[pos];
        ((float4*)out_boxes)[(size_t)b * max_out + rank] = sb[pos];
        out_scores[(size_t)b * max_out + rank] = S[flat];
        out_labels[(size_t)b * max_out + rank] = (int32_t)cls[pos];
    });
    if (tid == 0) num_dets[b] = nk;
}

// ---------------------------------------------------------------------------------------------
// EfficientNMS contract, multi-workgroup form (round 2).  The single-workgroup kernel above spends ~340 us per image on
// 32 of 256 CUs; here the work of one image is spread out:
//   en2_filter   grid (chunks, B)   streams the scores once at HBM rate, compacts the candidates (score > thr) of an
//                                    image into its list of u64 keys {~orderable(score), flat index}, counts per class
//   en2_select   grid (B)           ONLY for images with more than pre_topk candidates: exact top-k by (score desc, flat
//                                    index asc) with the register-resident binary radix select of the kernel above
//   en2_class    grid (nc, B)       per-class greedy NMS is independent of the other classes: sort the class's candidates,
//                                    then walk them in tiles of 64 - suppression by earlier kept boxes and the 64 x 64
//                                    in-tile suppression matrix are computed by all waves in parallel (ballots), the
//                                    sequential part is a bit-scan over one 64-bit word per tile; stops at max_out kept
//   en2_merge    grid (B)           merges the per-class kept lists: top max_out by (score desc, flat asc) = exactly
//                                    what the sequential scan over all classes keeps first, because whether a candidate
//                                    is kept depends only on higher-ranked candidates of ITS class
// Same IoU arithmetic, same total order, same outputs bit for bit (tests/test_gpu_boxes.py runs both forms).
// ---------------------------------------------------------------------------------------------
constexpr int EN2_FT = 256;                 // filter: threads per workgroup
constexpr int EN2_FPT = 16;                 // filter: scores per thread  -> 4096 scores per workgroup
constexpr int EN2_SEGS = 11;                // chunks per image in segmented mode (44 K scores)
constexpr int EN2_CC_LDS = 2048;            // per-class counters staged in LDS up to this many classes

struct En2Ws {                              // device pointers into the caller's workspace (see yv_efficient_nms_ws_bytes)
    uint64_t* cand;                         // (B, lcap)  lcap = list capacity >= pre_topk (see en2_lcap)
    int lcap;
    uint64_t* kept;                         // (B, nc, max_out)
    uint32_t* count;                        // (B)      candidates of the image (may exceed pre_topk)
    uint32_t* ccount;                       // (B, nc)  candidates per class (of the selected set)
    uint32_t* nkept;                        // (B, nc)
    uint32_t* done;                         // (B)      1: the head produced the image's final outputs
    uint32_t* seg;                          // (B, EN2_SEGS) segmented mode: candidates of each 4096-score chunk of the image
    int nseg;                               // > 0: segmented mode with this many chunks per image (see en2_filter_kernel)
};

__device__ __forceinline__ float key_score(uint32_t key) {          // inverse of desc_key
    const uint32_t asc = ~key;
    const uint32_t u = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
    return __uint_as_float(u);
}

__global__ __launch_bounds__(EN2_FT) void en2_filter_kernel(const float* __restrict__ scores, int total, int nc, float thr,
                                                            int K, En2Ws ws) {
    __shared__ uint32_t wave_cnt[EN2_FT / 64];
    __shared__ uint32_t base_sh;
    __shared__ uint32_t cc[EN2_CC_LDS];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* S = scores + (size_t)b * total;
    const int i0 = blockIdx.x * (EN2_FT * EN2_FPT);
    const bool cc_lds = nc <= EN2_CC_LDS;
    if (cc_lds) for (int c = tid; c < nc; c += EN2_FT) cc[c] = 0;
    // element j of this thread is score number idx(j): 16-byte loads (a wave reads 1 KB per instruction) where the image's score
    // block allows them (16-byte aligned, the chunk inside `total`), 4-byte loads otherwise; the list is a SET (every consumer
    // orders it by key), so the order in which a chunk's candidates are written does not matter
    const bool vec = (((uintptr_t)S & 15) == 0) && i0 + EN2_FT * EN2_FPT <= total;
    auto idx = [&](int j) __attribute__((always_inline)) { return vec ? i0 + ((j >> 2) * EN2_FT + tid) * 4 + (j & 3) : i0 + j * EN2_FT + tid; };
    float v[EN2_FPT];
    if (vec) {
#pragma unroll
        for (int j4 = 0; j4 < EN2_FPT / 4; ++j4) {           // all loads in flight
            const float4 q4 = *(const float4*)(S + i0 + (j4 * EN2_FT + tid) * 4);
            v[4 * j4] = q4.x; v[4 * j4 + 1] = q4.y; v[4 * j4 + 2] = q4.z; v[4 * j4 + 3] = q4.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < EN2_FPT; ++j) {                  // all loads in flight; lane-contiguous 256-byte wave accesses
            const int i = i0 + j * EN2_FT + tid;
            v[j] = i < total ? S[i] : 0.0f;
        }
    }
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < EN2_FPT; ++j) {
        const int i = idx(j);
        mine += (i < total && v[j] > thr) ? 1u : 0u;
    }
    // exclusive rank of this thread's candidates inside the workgroup
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < EN2_FT / 64; ++w) { const uint32_t c = wave_cnt[w]; if (w < wave) before += c; tot += c; }
    // SEGMENTED mode (images of up to 44 K scores, i.e. any detector-sized problem): the chunk's candidates go to the chunk's own
    // 4096-entry region of the list and its count to seg[b][chunk] - plain stores, nothing to zero beforehand, so the call needs
    // no memset launch in front (a dependent launch costs ~4 us of a ~32 us call); per-class counts are left to the general
    // path.  Otherwise: one list per image through an atomic counter that the caller's memset zeroed.
    const bool segm = ws.nseg > 0;
    if (tid == 0) {
        if (segm) { ws.seg[(size_t)b * EN2_SEGS + blockIdx.x] = tot; base_sh = blockIdx.x * (EN2_FT * EN2_FPT); }
        else base_sh = tot ? atomicAdd(&ws.count[b], tot) : 0u;
    }
    __syncthreads();
    if (tot == 0) return;
    uint32_t pos = base_sh + before + (incl - mine);
    uint64_t* list = ws.cand + (size_t)b * ws.lcap;
#pragma unroll
    for (int j = 0; j < EN2_FPT; ++j) {
        const int i = idx(j);
        if (i < total && v[j] > thr) {
            if (pos < (uint32_t)ws.lcap) list[pos] = ((uint64_t)desc_key(v[j]) << 32) | (uint32_t)i;
            ++pos;
            if (!segm) {
                const uint32_t c = (uint32_t)i % (uint32_t)nc;
                if (cc_lds) atomicAdd(&cc[c], 1u); else atomicAdd(&ws.ccount[(size_t)b * nc + c], 1u);
            }
        }
    }
    if (cc_lds && !segm) {
        __syncthreads();
        for (int c = tid; c < nc; c += EN2_FT) { const uint32_t n = cc[c]; if (n) atomicAdd(&ws.ccount[(size_t)b * nc + c], n); }
    }
}

// segmented list: candidate number g of the image (chunks in order) sits at list[g + shift of its chunk]; pre[s] = candidates in
// the chunks before s, dlt[s] = s * 4096 - pre[s] (uniform values; chunk 0 has shift 0)
struct SegMap { uint32_t pre[EN2_SEGS], dlt[EN2_SEGS]; int nseg; };
__device__ __forceinline__ uint32_t seg_index(const SegMap& m, uint32_t g) {
    uint32_t d = 0;
#pragma unroll
    for (int sgi = 1; sgi < EN2_SEGS; ++sgi)
        if (sgi < m.nseg && g >= m.pre[sgi]) d = m.dlt[sgi];
    return g + d;
}

// images with more than K candidates: exact selection of the K best (score desc, flat asc), rewritten into the list.
// One workgroup per image keeps the image's candidate keys in registers (A*nc <= 48K; larger problems take the bit-serial
// fallback below) and runs a radix select with 8-bit digits starting at the HIGHEST BIT IN WHICH THE KEYS DIFFER (scores of
// one detector share sign and most of the exponent: starting at bit 31 would put every key into one or two bins and
// serialise the LDS atomics).  Histograms are private per wave; ties on the cut key are taken in flat-index order.
constexpr int ES_CACHE = 44;
struct SelLds {
    uint32_t hist[16][256];
    uint32_t tot[256];
    uint32_t wave_cnt[16];
    uint32_t sel[4];
};

// K-th smallest of the candidate keys held in registers (sentinel ~0 = no candidate) by the 1024 threads of a workgroup: 8-bit
// radix digits from the highest differing bit down.  Returns (uniformly) the cut key and how many keys EQUAL to it belong to
// the K smallest.  Requires at least K candidates.
template <int NR>
__device__ __forceinline__ void radix_select_regs(const uint32_t (&kreg)[NR], uint32_t K, SelLds& L, uint32_t& key_star,
                                                  uint32_t& need_eq) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t ka = 0xFFFFFFFFu, ko = 0u;
#pragma unroll
    for (int c = 0; c < NR; ++c)
        if (kreg[c] != 0xFFFFFFFFu) { ka &= kreg[c]; ko |= kreg[c]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ka &= __shfl_xor(ka, o, 64); ko |= __shfl_xor(ko, o, 64); }
    __syncthreads();
    if (lane == 0) { L.hist[0][wave] = ka; L.hist[1][wave] = ko; }
    __syncthreads();
    ka = 0xFFFFFFFFu; ko = 0u;
#pragma unroll
    for (int w = 0; w < 16; ++w) { ka &= L.hist[0][w]; ko |= L.hist[1][w]; }
    __syncthreads();
    const uint32_t diff = ka ^ ko;
    uint32_t prefix = ka, need = K;
    if (diff) {
        int pos = 31 - __builtin_clz(diff);
        prefix = ka & ~((pos == 31) ? 0xFFFFFFFFu : ((2u << pos) - 1u));       // the bits above `pos` are common
        while (pos >= 0) {
            const int lo = pos >= 7 ? pos - 7 : 0;
            const uint32_t nbins = 1u << (pos - lo + 1);
            const uint32_t hi_mask = pos == 31 ? 0u : ~((2u << pos) - 1u);
            for (int i = tid; i < 16 * 256; i += EN_THREADS) (&L.hist[0][0])[i] = 0;
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                const uint32_t k = kreg[c];
                if (k != 0xFFFFFFFFu && (k & hi_mask) == prefix) atomicAdd(&L.hist[wave][(k >> lo) & (nbins - 1u)], 1u);
            }
            __syncthreads();
            if (tid < 256) {
                uint32_t t = 0;
#pragma unroll
                for (int w = 0; w < 16; ++w) t += L.hist[w][tid];
                L.tot[tid] = t;
            }
            __syncthreads();
            if (wave == 0) {                                  // 64 lanes x 4 consecutive bins: find the bin holding the need-th key
                const uint32_t t0 = L.tot[lane * 4], t1 = L.tot[lane * 4 + 1], t2 = L.tot[lane * 4 + 2], t3 = L.tot[lane * 4 + 3];
                const uint32_t mine = t0 + t1 + t2 + t3;
                uint32_t incl = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
                const uint32_t before = incl - mine;
                if (before < need && need <= incl) {          // exactly one lane
                    uint32_t bsel = lane * 4, bb = before;
                    if (need > bb + t0) { bb += t0; ++bsel; if (need > bb + t1) { bb += t1; ++bsel; if (need > bb + t2) { bb += t2; ++bsel; } } }
                    L.sel[0] = bsel; L.sel[1] = bb;
                }
            }
            __syncthreads();
            prefix |= L.sel[0] << lo;
            need -= L.sel[1];
            pos = lo - 1;
        }
    }
    key_star = prefix; need_eq = need;
}

// block-wide exclusive position of this thread's `mine` items (thread order) and the total; uses L.wave_cnt
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t mine, SelLds& L, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    __syncthreads();
    if (lane == 63) L.wave_cnt[wave] = incl;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const uint32_t v = L.wave_cnt[w]; if (w < wave) before += v; tot += v; }
    total = tot;
    return before + incl - mine;
}

__device__ __forceinline__ void en2_select_body(const float* __restrict__ scores, int A, int nc, float score_thr, int K,
                                                const En2Ws& ws, int b, SelLds& L, uint32_t* cc) {
    uint32_t (&wave_cnt)[16] = L.wave_cnt;
    uint32_t (&sel)[4] = L.sel;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = A * nc;
    const float* S = scores + (size_t)b * total;
    const int chunks = (total + EN_THREADS - 1) / EN_THREADS;
    uint64_t* list = ws.cand + (size_t)b * ws.lcap;
    const bool cc_lds = nc <= EN2_CC_LDS;
    uint32_t key_star, need_eq;
    if (chunks <= ES_CACHE) {
        uint32_t kreg[ES_CACHE];
#pragma unroll
        for (int c = 0; c < ES_CACHE; ++c) {
            kreg[c] = 0xFFFFFFFFu;                                // not a candidate (no real key is ~0: that would be a NaN score)
            const int i = c * EN_THREADS + tid;
            if (c < chunks && i < total) { const float v = S[i]; if (v > score_thr) kreg[c] = desc_key(v); }
        }
        radix_select_regs(kreg, (uint32_t)K, L, key_star, need_eq);
        // ---- compaction.  Keys below the cut: any order (the consumer sorts).  Keys EQUAL to the cut: when all of them are
        //      wanted (always, unless scores tie exactly on the cut) they are appended the same way; otherwise the first need_eq
        //      in flat-index order are chosen by the ordered pass at the end of this kernel (shared with the fallback)
        uint32_t n_le = 0, n_eq = 0;
#pragma unroll
        for (int c = 0; c < ES_CACHE; ++c) {
            const uint32_t k = kreg[c];
            n_le += (k <= key_star && k != 0xFFFFFFFFu) ? 1u : 0u;
            n_eq += (k == key_star && k != 0xFFFFFFFFu) ? 1u : 0u;
        }
        uint32_t le_total, eq_total;
        (void)block_excl_scan(n_eq, L, eq_total);
        const uint32_t first = block_excl_scan(n_le, L, le_total);
        __syncthreads();
        if (eq_total == need_eq) {                                // uniform: every key up to and including the cut is selected
            if (cc_lds) for (int c = tid; c < nc; c += EN_THREADS) cc[c] = 0;
            for (int c = tid; c < nc; c += EN_THREADS) ws.ccount[(size_t)b * nc + c] = 0;     // recounted for the selected set
            __syncthreads();
            uint32_t pos = first;
#pragma unroll
            for (int c = 0; c < ES_CACHE; ++c) {
                const uint32_t k = kreg[c];
                if (k <= key_star && k != 0xFFFFFFFFu) {
                    const uint32_t flat = (uint32_t)(c * EN_THREADS + tid);
                    if (pos < (uint32_t)K) list[pos] = ((uint64_t)k << 32) | flat;
                    ++pos;
                    const uint32_t cl = flat % (uint32_t)nc;
                    if (cc_lds) atomicAdd(&cc[cl], 1u); else atomicAdd(&ws.ccount[(size_t)b * nc + cl], 1u);
                }
            }
            __syncthreads();
            if (cc_lds) for (int c = tid; c < nc; c += EN_THREADS) { const uint32_t n = cc[c]; if (n) ws.ccount[(size_t)b * nc + c] = n; }
            return;
        }
    } else {
        // ---- more than 48K scores per image: bit-serial select over the scores in L2 (the round-1 algorithm) ------------
        uint32_t prefix = 0, need = (uint32_t)K;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t hi_mask = bit == 31 ? 0u : (0xFFFFFFFFu << (bit + 1));
            uint32_t c0 = 0;
            for (int c = 0; c < chunks; ++c) {
                const int i = c * EN_THREADS + tid;
                if (i < total) {
                    const float v = S[i];
                    if (v > score_thr) { const uint32_t k = desc_key(v); c0 += (((k & hi_mask) == prefix) && !((k >> bit) & 1u)) ? 1u : 0u; }
                }
            }
            uint32_t w = c0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
            __syncthreads();
            if (lane == 0) wave_cnt[wave] = w;
            __syncthreads();
            uint32_t t0 = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t0 += wave_cnt[q];
            if (need > t0) { need -= t0; prefix |= (1u << bit); }
        }
        key_star = prefix; need_eq = need;
    }
    // ---- ordered compaction (scores re-read from L2, chunk by chunk): ties on the cut key in flat-index order ---------------
    if (cc_lds) for (int c = tid; c < nc; c += EN_THREADS) cc[c] = 0;
    for (int c = tid; c < nc; c += EN_THREADS) ws.ccount[(size_t)b * nc + c] = 0;
    if (tid == 0) { sel[0] = 0; sel[1] = 0; }                     // [0] selected so far, [1] cut-key ties seen so far
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        const int i = c * EN_THREADS + tid;
        bool pass = false, eq = false;
        uint32_t k = 0;
        if (i < total) {
            const float v = S[i];
            if (v > score_thr) { k = desc_key(v); pass = k < key_star; eq = k == key_star; }
        }
        uint32_t tt;
        const uint32_t r = block_rank(eq, wave_cnt, &tt);
        const uint32_t base = sel[1];
        const bool take = pass || (eq && base + r < need_eq);
        __syncthreads();
        if (tid == 0) sel[1] = base + tt;
        if (take) {
            const uint32_t at = atomicAdd(&sel[0], 1u);
            if (at < (uint32_t)K) list[at] = ((uint64_t)k << 32) | (uint32_t)i;
            const uint32_t cl = (uint32_t)i % (uint32_t)nc;
            if (cc_lds) atomicAdd(&cc[cl], 1u); else atomicAdd(&ws.ccount[(size_t)b * nc + cl], 1u);
        }
        __syncthreads();
    }
    if (cc_lds) for (int c = tid; c < nc; c += EN_THREADS) { const uint32_t n = cc[c]; if (n) ws.ccount[(size_t)b * nc + c] = n; }
}

// Fast path, one workgroup per image: the sequential scan only ever looks at the candidates ranked before the max_out-th
// kept box, which for ordinary detections is a short prefix of the ranking.  So: take the EN2_HEAD best candidates (radix
// select over register-resident keys), order them (rank sort: 512 x 512 comparisons spread over 1024 threads, two barriers -
// a bitonic network would need 45), walk them in 64-wide tiles exactly like en2_one_class but with the class test inside
// the suppression predicate, and stop at max_out kept.  If max_out boxes were kept - or every candidate of the image was in
// the head - the result IS the sequential result and the image is marked done; otherwise (heavy suppression, ties on the
// head's cut key, more than 45K scores) the general kernels below redo the image from scratch.
// Greedy resolution of one 64-candidate tile in parallel rounds (every wave runs it redundantly, uniform control flow).
// alive: candidates no box kept in an earlier tile suppresses (uniform); col: the members i < lane of the tile that would
// suppress this lane's candidate; room: how many boxes may still be kept.  Candidate j is kept iff it is alive and no KEPT
// i < j suppresses it.  Per round every undecided candidate with a kept suppressor dies and every one whose suppressors are
// all decided is kept - the lowest undecided one always can - so the loop takes (longest suppression chain) rounds, 2-4 on
// detector output, instead of one scalar step per kept box.  The scan stops at max_out: only the first `room` kept count.
__device__ __forceinline__ uint64_t tile_resolve(uint64_t alive, uint64_t col, int room) {
    const int lane = threadIdx.x & 63;
    const uint64_t me = 1ull << lane;
    uint64_t und = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(alive >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)alive);
    uint64_t kept = 0;
    while (und) {
        const bool mine = (und & me) != 0;
        const bool die = mine && (col & kept) != 0;
        const bool keep = mine && !die && (col & und) == 0;
        const uint64_t k = __ballot(keep), d = __ballot(die);
        kept |= k;
        und &= ~(k | d);
    }
    if (__builtin_popcountll(kept) > room)
        kept = __ballot((kept & me) != 0 && __builtin_popcountll(kept & (me - 1ull)) < room);
    return kept;
}

constexpr int EN2_HEAD = 512;                // capacity of the head: images with this many candidates or fewer are taken whole
constexpr int EN2_WIN = 320;                 // otherwise a prefix of EN2_WIN/2 .. EN2_WIN members (rank sort cost grows with its square)
constexpr int EN2_HR = 16;                  // candidate keys per thread, images with up to 16 K candidates ...
constexpr int EN2_HRW = ES_CACHE;           // ... and up to 44 K (keys only: the flat index of a selected key is re-read from the list)
struct HeadLds {                            // (one instance in the kernel: the two instantiations of the head share it)
    uint64_t keys[EN2_HEAD];                // unordered, then sorted
    uint64_t tmpk[EN2_HEAD];
    uint32_t rank[EN2_HEAD];
    float4 sb[EN2_HEAD];
    uint16_t scl[EN2_HEAD];
    uint64_t Mcol[2][64];                   // in-tile suppressor masks, double-buffered over tiles
    uint64_t dead[16];
};
template <int NR>
__device__ __forceinline__ bool en2_head_body(const float* __restrict__ boxes, const float* __restrict__ scores, int A, int nc,
                                              float iou_thr, int max_out, int K, const En2Ws& ws, int b, uint32_t cnt,
                                              const SegMap& sm, SelLds& L, HeadLds& H, unsigned char* dyn, int32_t* __restrict__ num_dets,
                                              float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                              int32_t* __restrict__ out_labels) {
    uint64_t (&keys)[EN2_HEAD] = H.keys;
    uint64_t (&tmpk)[EN2_HEAD] = H.tmpk;
    uint32_t (&rank)[EN2_HEAD] = H.rank;
    float4 (&sb)[EN2_HEAD] = H.sb;
    uint16_t (&scl)[EN2_HEAD] = H.scl;
    uint64_t (&Mcol)[2][64] = H.Mcol;
    uint64_t (&dead)[16] = H.dead;
    float4* kbox = (float4*)dyn;                               // kept boxes (max_out float4), keys, classes
    uint64_t* kkey = (uint64_t*)(kbox + max_out);
    uint16_t* kcl = (uint16_t*)(kkey + max_out);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = A * nc;
    const float* S = scores + (size_t)b * total;
    // the image's candidates were compacted into its list by en2_filter_kernel (the launch before): up to NR per thread
    const uint64_t* list = ws.cand + (size_t)b * ws.lcap;
    uint32_t kreg[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) {
        const uint32_t i = (uint32_t)(c * EN_THREADS + tid);
        kreg[c] = i < cnt ? ((const uint32_t*)list)[2 * seg_index(sm, i) + 1] : 0xFFFFFFFFu;     // (the key is the high word of the entry)
    }
    for (int i = tid; i < 16 * 256; i += EN_THREADS) (&L.hist[0][0])[i] = 0;     // (for the first histogram pass, under the loads)
    const uint32_t head = (uint32_t)(K < EN2_HEAD ? K : EN2_HEAD);
    const uint32_t win = head < (uint32_t)EN2_WIN ? head : (uint32_t)EN2_WIN;     // prefix size aimed at when a cut is needed
    uint32_t key_star = 0xFFFFFFFEu;                            // take every candidate
    if (cnt > head) {
        // ANY prefix of the ranking with between head/2 and head members will do, so the cut is put on a radix-digit boundary:
        // usually ONE histogram pass (8 bits below the highest bit in which the keys differ), a second one only when the
        // boundary bin straddles the window; a digit boundary never splits equal keys, so ties need no care here
        uint32_t ka = 0xFFFFFFFFu, ko = 0u;
#pragma unroll
        for (int c = 0; c < NR; ++c)
            if (kreg[c] != 0xFFFFFFFFu) { ka &= kreg[c]; ko |= kreg[c]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ka &= __shfl_xor(ka, o, 64); ko |= __shfl_xor(ko, o, 64); }
        if (lane == 0) { L.tot[wave] = ka; L.tot[16 + wave] = ko; }      // (not L.hist: zeroed above for the first pass)
        __syncthreads();
        ka = 0xFFFFFFFFu; ko = 0u;
#pragma unroll
        for (int w = 0; w < 16; ++w) { ka &= L.tot[w]; ko |= L.tot[16 + w]; }
        const uint32_t diff = ka ^ ko;
        if (!diff) return false;                                     // more than `head` identical scores: general path
        int pos = 31 - __builtin_clz(diff);
        uint32_t prefix = ka & ~((pos == 31) ? 0xFFFFFFFFu : ((2u << pos) - 1u));
        uint32_t before = 0;
        bool ok = false, first = true;
        while (pos >= 0) {
            const int lo = pos >= 7 ? pos - 7 : 0;
            const uint32_t nbins = 1u << (pos - lo + 1);
            const uint32_t hi_mask = pos == 31 ? 0u : ~((2u << pos) - 1u);
            if (!first) {                                       // (zeroed at kernel entry for the first pass)
                for (int i = tid; i < 16 * 256; i += EN_THREADS) (&L.hist[0][0])[i] = 0;
                __syncthreads();
            }
            first = false;
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                const uint32_t k = kreg[c];
                if (k != 0xFFFFFFFFu && (k & hi_mask) == prefix) atomicAdd(&L.hist[wave][(k >> lo) & (nbins - 1u)], 1u);
            }
            __syncthreads();
            if (tid < 256) {
                uint32_t t = 0;
#pragma unroll
                for (int w = 0; w < 16; ++w) t += L.hist[w][tid];
                L.tot[tid] = t;
            }
            __syncthreads();
            if (wave == 0) {                                    // number of leading bins that fit into the window, and their total
                const uint32_t t0 = L.tot[lane * 4], t1 = L.tot[lane * 4 + 1], t2 = L.tot[lane * 4 + 2], t3 = L.tot[lane * 4 + 3];
                const uint32_t mine4 = t0 + t1 + t2 + t3;
                uint32_t incl = mine4;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
                const uint32_t b0 = before + incl - mine4;      // candidates ranked before this lane's first bin
                const uint32_t c0 = b0 + t0, c1 = c0 + t1, c2 = c1 + t2, c3 = c2 + t3;
                const uint32_t fit = (c0 <= win ? 1u : 0u) + (c1 <= win ? 1u : 0u) + (c2 <= win ? 1u : 0u) + (c3 <= win ? 1u : 0u);
                uint32_t nb = fit;                               // cumulative counts are monotone: fitting bins form a prefix
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) nb += __shfl_xor(nb, o, 64);
                if ((uint32_t)lane == (nb ? (nb - 1) >> 2 : 0u)) {
                    const uint32_t within = nb ? ((nb - 1) & 3u) : 0u;
                    const uint32_t upto = nb == 0 ? before : (within == 0 ? c0 : within == 1 ? c1 : within == 2 ? c2 : c3);
                    L.sel[0] = nb; L.sel[1] = upto;
                }
            }
            __syncthreads();
            const uint32_t nb = L.sel[0], upto = L.sel[1];      // bins [0, nb) fit; `upto` candidates are ranked below bin nb
            if (upto >= win / 2 || nb >= nbins) {
                key_star = prefix + (nb << lo) - 1u;            // every key below bin nb of this prefix range
                ok = upto > 0 && nb < nbins;                    // (nb == nbins cannot happen while more than `head` keys are in range)
                break;
            }
            before = upto;                                      // bin nb straddles the window: refine inside it
            prefix |= nb << lo;
            pos = lo - 1;
            __syncthreads();
        }
        if (!ok) return false;                                       // one score value shared by hundreds of candidates: general path
    }
    uint32_t n_le = 0, n;
#pragma unroll
    for (int c = 0; c < NR; ++c) n_le += (kreg[c] <= key_star) ? 1u : 0u;        // the sentinel ~0 is above every cut
    uint32_t pos = block_excl_scan(n_le, L, n);
#pragma unroll
    for (int c = 0; c < NR; ++c)
        if (kreg[c] <= key_star) { if (pos < (uint32_t)EN2_HEAD) tmpk[pos] = list[seg_index(sm, (uint32_t)(c * EN_THREADS + tid))]; ++pos; }
    if (tid < EN2_HEAD) rank[tid] = 0;
    n = n < (uint32_t)EN2_HEAD ? n : (uint32_t)EN2_HEAD;        // (n == head when cnt > head, else cnt)
    if (tid >= (int)n && tid < EN2_HEAD) tmpk[tid] = ~0ull;     // sentinels: never "before" a key, so the count loop needs no bound
    __syncthreads();
    {   // rank sort: thread (i, part) counts the keys of its part that precede key i; keys are distinct -> a permutation.
        // Trip count a multiple of 16 (sentinel padding) and unrolled: the broadcast LDS reads of a batch are in flight
        // together (one read per iteration with a data-dependent bound cost a full LDS round trip each: 12.6 us of the
        // kernel's 40); the loop is bound by its 64-bit compares, so all 1024 threads share the n x n comparisons.
        const uint32_t sh = n <= 128u ? 7u : n <= 256u ? 8u : 9u;                 // keys padded to 128 / 256 / 512 ...
        const uint32_t i = tid & ((1u << sh) - 1u), part = tid >> sh, parts = (uint32_t)EN_THREADS >> sh;   // ... 8 / 4 / 2 parts
        const uint32_t plen = (((n + parts - 1u) / parts) + 15u) & ~15u;         // parts * plen <= EN2_HEAD
        if (i < n) {
            const uint64_t ki = tmpk[i];
            const uint64_t* p = tmpk + part * plen;
            uint32_t r = 0;
            for (uint32_t j0 = 0; j0 < plen; j0 += 16) {
#pragma unroll
                for (int j = 0; j < 16; ++j) r += p[j0 + j] < ki ? 1u : 0u;
            }
            atomicAdd(&rank[i], r);
        }
    }
    __syncthreads();
    const float4* Bx = (const float4*)boxes + (size_t)b * A;
    if (tid < (int)n) {
        const uint64_t k = tmpk[tid];
        const uint32_t r = rank[tid], flat = (uint32_t)k;
        keys[r] = k;
        sb[r] = Bx[flat / (uint32_t)nc];
        scl[r] = (uint16_t)(flat % (uint32_t)nc);
    }
    __syncthreads();
    if (tid < 128) (&Mcol[0][0])[tid] = 0;
    __syncthreads();
    int nk = 0;
    for (uint32_t t0 = 0; t0 < n && nk < max_out; t0 += 64) {
        uint64_t* mc = Mcol[(t0 >> 6) & 1];
        if (wave == 1) Mcol[((t0 >> 6) & 1) ^ 1][lane] = 0;      // the next tile's masks (last read before the previous tile's closing barrier)
        const uint32_t j = t0 + lane;
        const bool valid = j < n;
        const float4 bj = valid ? sb[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        const uint32_t cj = valid ? scl[j] : 0xFFFFu;
        const float aj = box_area(bj);
        uint64_t d = 0;
        for (int k = wave; k < nk; k += 16) {
            const float4 bk = kbox[k];
            d |= __ballot(valid && (uint32_t)kcl[k] == cj && iou_f32(bk, box_area(bk), bj, aj) > iou_thr);
        }
        if (lane == 0) dead[wave] = d;
        uint64_t col = 0;                                       // in-tile suppressors of candidate j among this wave's rows
        for (int i = wave; i < 64; i += 16) {
            const uint32_t ji = t0 + i;
            if (ji < n) {
                const float4 bi = sb[ji];
                if (valid && lane > i && (uint32_t)scl[ji] == cj && iou_f32(bi, box_area(bi), bj, aj) > iou_thr) col |= 1ull << i;
            }
        }
        if (col) atomicOr((unsigned long long*)&mc[lane], (unsigned long long)col);
        __syncthreads();
        uint64_t alive = __ballot(valid);
#pragma unroll
        for (int w = 0; w < 16; ++w) alive &= ~dead[w];
        const uint64_t keptmask = tile_resolve(alive, mc[lane], max_out - nk);
        const int nk2 = nk + __builtin_popcountll(keptmask);
        if (wave == 0 && ((keptmask >> lane) & 1ull)) {
            const int r = nk + __builtin_popcountll(keptmask & ((1ull << lane) - 1ull));
            kbox[r] = bj; kcl[r] = (uint16_t)cj; kkey[r] = keys[j];
        }
        nk = nk2;
        __syncthreads();
    }
    if (nk < max_out && cnt > n) return false;                       // the head did not suffice: general path (done stays 0)
    for (int i = tid; i < max_out; i += EN_THREADS) {
        const size_t o = (size_t)b * max_out + i;
        if (i < nk) {
            ((float4*)out_boxes)[o] = kbox[i];
            out_scores[o] = S[(uint32_t)kkey[i]];
            out_labels[o] = (int32_t)kcl[i];
        } else {
            ((float4*)out_boxes)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
            out_scores[o] = 0.f;
            out_labels[o] = 0;
        }
    }
    if (tid == 0) { num_dets[b] = nk; ws.done[b] = 1u; }
    return true;
}


// per-class greedy NMS over one image's candidate list.  TIER 0: classes with <= 1024 candidates (256 threads, 26 KB of LDS,
// several workgroups per CU); TIER 1: up to 4096 (1024 threads).  Both are launched over the same (nc, B) grid and a
// workgroup leaves at once when the class belongs to the other tier (the count is only known on the device).
template <int CAP, int THREADS>
__device__ __forceinline__ void en2_one_class(unsigned char* smem, const float* __restrict__ boxes, int A, int nc, int c, int b,
                                              float iou_thr, int max_out, int K, const En2Ws& ws) {
    constexpr int NW = THREADS / 64;
    uint64_t* keys = (uint64_t*)smem;                          // CAP
    float4* sb = (float4*)(keys + CAP);                        // CAP   boxes in sorted order
    float4* kbox = sb + CAP;                                   // max_out kept boxes
    uint64_t* Mcol = (uint64_t*)(kbox + max_out);              // 2 x 64 in-tile suppressor masks (double-buffered over tiles)
    uint64_t* dead = Mcol + 128;                               // NW partial masks
    uint32_t* misc = (uint32_t*)(dead + NW);                   // [0] n_c
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t n_c;
    uint32_t n = __atomic_load_n(&ws.count[b], __ATOMIC_RELAXED);     // (coherent loads: inside en2_front_kernel the list and
    n = n < (uint32_t)K ? n : (uint32_t)K;                           //  the counts come from other workgroups of the same launch)
    const uint64_t* list = ws.cand + (size_t)b * ws.lcap;
    if (tid == 0) misc[0] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += THREADS) {
        const uint64_t k = __atomic_load_n(list + i, __ATOMIC_RELAXED);
        if ((uint32_t)k % (uint32_t)nc == (uint32_t)c) {
            const uint32_t pos = atomicAdd(&misc[0], 1u);
            if (pos < (uint32_t)CAP) keys[pos] = k;
        }
    }
    __syncthreads();
    n_c = misc[0] < (uint32_t)CAP ? misc[0] : (uint32_t)CAP;
    const int np = next_pow2((int)n_c, 64);
    for (int i = (int)n_c + tid; i < np; i += THREADS) keys[i] = ~0ull;
    bitonic_sort_u64(keys, np);
    const float4* Bx = (const float4*)boxes + (size_t)b * A;
    for (uint32_t i = tid; i < n_c; i += THREADS) sb[i] = Bx[(uint32_t)keys[i] / (uint32_t)nc];
    __syncthreads();

    uint64_t* out = ws.kept + ((size_t)b * nc + c) * max_out;
    if (tid < 128) Mcol[tid] = 0;
    __syncthreads();
    int nk = 0;
    for (uint32_t t0 = 0; t0 < n_c && nk < max_out; t0 += 64) {
        uint64_t* mc = Mcol + ((t0 >> 6) & 1) * 64;
        if (wave == 1) Mcol[(((t0 >> 6) & 1) ^ 1) * 64 + lane] = 0;   // the next tile's masks (last read before the previous tile's closing barrier)
        const uint32_t j = t0 + lane;
        const bool valid = j < n_c;
        const float4 bj = valid ? sb[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float aj = box_area(bj);
        // (1) suppression by the boxes kept in earlier tiles: the kept list is dealt round-robin to the waves
        uint64_t d = 0;
        for (int k = wave; k < nk; k += NW) {
            const float4 bk = kbox[k];
            d |= __ballot(valid && iou_f32(bk, box_area(bk), bj, aj) > iou_thr);
        }
        if (lane == 0) dead[wave] = d;
        // (2) in-tile suppression: which earlier members of the tile would suppress candidate j (this wave's share of the rows)
        uint64_t col = 0;
        for (int i = wave; i < 64; i += NW) {
            const uint32_t ji = t0 + i;
            if (ji < n_c) {                                     // uniform
                const float4 bi = sb[ji];
                if (valid && lane > i && iou_f32(bi, box_area(bi), bj, aj) > iou_thr) col |= 1ull << i;
            }
        }
        if (col) atomicOr((unsigned long long*)&mc[lane], (unsigned long long)col);
        __syncthreads();
        // (3) the sequential part: every wave resolves the tile redundantly (tile_resolve: parallel rounds)
        uint64_t alive = __ballot(valid);
#pragma unroll
        for (int w = 0; w < NW; ++w) alive &= ~dead[w];
        const uint64_t keptmask = tile_resolve(alive, mc[lane], max_out - nk);
        const int nk2 = nk + __builtin_popcountll(keptmask);
        if (wave == 0 && ((keptmask >> lane) & 1ull)) {
            const int r = nk + __builtin_popcountll(keptmask & ((1ull << lane) - 1ull));
            kbox[r] = bj;
            out[r] = keys[j];
        }
        nk = nk2;
        __syncthreads();
    }
    if (tid == 0) ws.nkept[(size_t)b * nc + c] = (uint32_t)nk;
    __syncthreads();
}

// final step, once per image: the first max_out of the union of the per-class kept lists (by score, flat index) = what the
// sequential scan keeps first; `smem`: EN_MAXK keys.
__device__ __forceinline__ void en2_merge_body(unsigned char* smem, const float* __restrict__ boxes, const float* __restrict__ scores,
                                               int A, int nc, int max_out, const En2Ws& ws, int b, int32_t* __restrict__ num_dets,
                                               float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                               int32_t* __restrict__ out_labels) {
    uint64_t* keys = (uint64_t*)smem;                          // EN_MAXK
    __shared__ uint32_t tot_sh;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (tid == 0) tot_sh = 0;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int c = wave; c < nc; c += nthr >> 6) {                       // a wave per class list (order is irrelevant: sorted below)
        const uint32_t nkc = __atomic_load_n(&ws.nkept[(size_t)b * nc + c], __ATOMIC_RELAXED);
        if (!nkc) continue;
        uint32_t off = lane == 0 ? atomicAdd(&tot_sh, nkc) : 0u;
        off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
        const uint64_t* src = ws.kept + ((size_t)b * nc + c) * max_out;
        for (uint32_t k = lane; k < nkc; k += 64) if (off + k < (uint32_t)EN_MAXK) keys[off + k] = __atomic_load_n(src + k, __ATOMIC_RELAXED);
    }
    __syncthreads();
    const int n = (int)(tot_sh < (uint32_t)EN_MAXK ? tot_sh : (uint32_t)EN_MAXK);
    const int np = next_pow2(n, 64);
    for (int i = n + tid; i < np; i += nthr) keys[i] = ~0ull;
    bitonic_sort_u64(keys, np);
    const int nout = n < max_out ? n : max_out;
    const float4* Bx = (const float4*)boxes + (size_t)b * A;
    for (int i = tid; i < max_out; i += nthr) {
        const size_t o = (size_t)b * max_out + i;
        if (i < nout) {
            const uint32_t flat = (uint32_t)keys[i];
            ((float4*)out_boxes)[o] = Bx[flat / (uint32_t)nc];
            out_scores[o] = scores[(size_t)b * A * nc + flat];
            out_labels[o] = (int32_t)(flat % (uint32_t)nc);
        } else {
            ((float4*)out_boxes)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
            out_scores[o] = 0.f;
            out_labels[o] = 0;
        }
    }
    if (tid == 0) num_dets[b] = nout;
}

// en2_front_kernel, one workgroup per image: head -> [exact top-K select] -> [the image's heavy classes].  One launch where there
// were three (head, select, class_large): a dependent launch costs the stream ~4 us even when its workgroups exit at once, and
// at batch 32 the whole NMS is 30-40 us.  Measured and rejected: (i) this workgroup streaming the image's scores itself instead
// of the en2_filter_kernel grid (three latency-bound batches on one CU: 42 -> 52 us per call at batch 32, 60 -> 129 us at batch
// 256); (ii) filter chunks and head in one launch with a "last workgroup of the image goes on" ticket, and the same for the
// merge: the release / acquire fences the hand-over needs write back and invalidate the XCD's L2 (the chunks of an image run
// on different XCDs): 70 us at batch 32, 540 us at batch 256.
__global__ __launch_bounds__(EN_THREADS) void en2_front_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                               int A, int nc, float score_thr, float iou_thr, int max_out, int K,
                                                               En2Ws ws, int32_t* __restrict__ num_dets,
                                                               float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                               int32_t* __restrict__ out_labels) {
    __shared__ SelLds L;
    __shared__ HeadLds H;
    __shared__ uint32_t cc[EN2_CC_LDS];
    __shared__ uint32_t heavy[16];
    __shared__ uint32_t n_heavy;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];     // en2_one_class<EN_MAXK> (covers the head's kept boxes)
    const int b = blockIdx.x, tid = threadIdx.x;
    SegMap sm;
    sm.nseg = ws.nseg;
    uint32_t cnt = 0;
    if (ws.nseg > 0) {                                         // segmented list: per-chunk counts -> prefix sums (uniform)
#pragma unroll
        for (int sgi = 0; sgi < EN2_SEGS; ++sgi) {
            const uint32_t n = sgi < ws.nseg ? ws.seg[(size_t)b * EN2_SEGS + sgi] : 0u;
            sm.pre[sgi] = cnt;
            sm.dlt[sgi] = (uint32_t)(sgi * (EN2_FT * EN2_FPT)) - cnt;
            cnt += n;
        }
    } else {
#pragma unroll
        for (int sgi = 0; sgi < EN2_SEGS; ++sgi) { sm.pre[sgi] = 0; sm.dlt[sgi] = 0; }
        cnt = ws.count[b];
    }
    bool done = false;                                         // (cnt beyond the list capacity: general path)
    if (cnt <= (uint32_t)(EN2_HR * EN_THREADS))
        done = en2_head_body<EN2_HR>(boxes, scores, A, nc, iou_thr, max_out, K, ws, b, cnt, sm, L, H, dyn, num_dets, out_boxes,
                                     out_scores, out_labels);
    else if (cnt <= (uint32_t)ws.lcap)
        done = en2_head_body<EN2_HRW>(boxes, scores, A, nc, iou_thr, max_out, K, ws, b, cnt, sm, L, H, dyn, num_dets, out_boxes,
                                      out_scores, out_labels);
    if (done) return;
    __syncthreads();
    if (ws.nseg > 0) {
        // nothing was zeroed for this call: leave what the memset + atomic filter would have left for the general path
        if (tid == 0) { ws.count[b] = cnt; ws.done[b] = 0u; }
        if (cnt <= (uint32_t)K) {
            // the list becomes contiguous in place: chunk 0's candidates stay, the others move down behind them - into chunk 0's
            // own region (cnt <= K <= 4096 entries), which no other chunk's candidates occupy
            uint64_t* list = ws.cand + (size_t)b * ws.lcap;
            const bool cc_lds = nc <= EN2_CC_LDS;
            if (cc_lds) for (int c = tid; c < nc; c += EN_THREADS) cc[c] = 0;
            else for (int c = tid; c < nc; c += EN_THREADS) ws.ccount[(size_t)b * nc + c] = 0;
            __syncthreads();
            for (uint32_t gi = tid; gi < cnt; gi += EN_THREADS) {
                const uint64_t k = list[seg_index(sm, gi)];
                if (gi >= sm.pre[1]) list[gi] = k;             // (ws.nseg == 1: pre[1] = cnt, nothing moves)
                const uint32_t cl = (uint32_t)k % (uint32_t)nc;
                if (cc_lds) atomicAdd(&cc[cl], 1u); else atomicAdd(&ws.ccount[(size_t)b * nc + cl], 1u);
            }
            __syncthreads();
            if (cc_lds) for (int c = tid; c < nc; c += EN_THREADS) ws.ccount[(size_t)b * nc + c] = cc[c];
            __threadfence_block();                             // list / counts are read back below (coherent loads)
        }
    }
    __syncthreads();
    if (cnt > (uint32_t)K) {                                   // else the list already is the selected set
        en2_select_body(scores, A, nc, score_thr, K, ws, b, L, cc);
        __threadfence_block();                                 // the rewritten list / counts are read back below (coherent loads)
        __syncthreads();
    }
    // heavy classes (more than 1024 selected candidates: at most K / 1024 of them), one after the other in this workgroup
    if (tid == 0) n_heavy = 0;
    __syncthreads();
    for (int c = tid; c < nc; c += EN_THREADS)
        if (__atomic_load_n(&ws.ccount[(size_t)b * nc + c], __ATOMIC_RELAXED) > 1024u) {
            const uint32_t p = atomicAdd(&n_heavy, 1u);
            if (p < 16u) heavy[p] = (uint32_t)c;
        }
    __syncthreads();
    const int nh = (int)(n_heavy < 16u ? n_heavy : 16u);
    for (int h = 0; h < nh; ++h) en2_one_class<EN_MAXK, 1024>(dyn, boxes, A, nc, (int)heavy[h], b, iou_thr, max_out, K, ws);
}

// en2_classes_kernel, grid (nc, B) x 256 threads, only for images the head could not finish: per-class greedy NMS of the classes
// with 1..1024 selected candidates (26 KB of LDS, several workgroups per CU).
__global__ __launch_bounds__(256) void en2_classes_kernel(const float* __restrict__ boxes, int A, int nc, float iou_thr,
                                                          int max_out, int K, En2Ws ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int c = blockIdx.x, b = blockIdx.y;
    if (ws.done[b]) return;
    const uint32_t n_c = ws.ccount[(size_t)b * nc + c];
    if (n_c == 0 && threadIdx.x == 0) ws.nkept[(size_t)b * nc + c] = 0;        // (segmented mode: no memset zeroed it)
    if (n_c == 0 || n_c > 1024u) return;
    en2_one_class<1024, 256>(smem, boxes, A, nc, c, b, iou_thr, max_out, K, ws);
}

// Few classes (nc <= EN2_TAIL_NC, the reference's detector has 5): ONE launch for what en2_classes_kernel + en2_merge_kernel do -
// a workgroup per image takes the image's small classes one after the other and merges.  For images the head finished (all of
// them on ordinary detector output) the saving is one idle dependent launch, ~4 us of a 35 us call.
constexpr int EN2_TAIL_NC = 8;
__global__ __launch_bounds__(1024) void en2_tail_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, int A,
                                                        int nc, float iou_thr, int max_out, int K, En2Ws ws,
                                                        int32_t* __restrict__ num_dets, float* __restrict__ out_boxes,
                                                        float* __restrict__ out_scores, int32_t* __restrict__ out_labels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    if (ws.done[b]) return;
    for (int c = 0; c < nc; ++c) {
        const uint32_t n_c = ws.ccount[(size_t)b * nc + c];    // (uniform; written by earlier launches)
        if (n_c == 0 && threadIdx.x == 0) ws.nkept[(size_t)b * nc + c] = 0;     // (segmented mode: no memset zeroed it)
        if (n_c == 0 || n_c > 1024u) continue;                 // heavy classes were done by en2_front_kernel
        en2_one_class<1024, 1024>(smem, boxes, A, nc, c, b, iou_thr, max_out, K, ws);
    }
    __threadfence_block();                                     // this workgroup's kept lists: read back coherently by the merge
    __syncthreads();
    en2_merge_body(smem, boxes, scores, A, nc, max_out, ws, b, num_dets, out_boxes, out_scores, out_labels);
}

__global__ __launch_bounds__(256) void en2_merge_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                        int A, int nc, int max_out, En2Ws ws, int32_t* __restrict__ num_dets,
                                                        float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                        int32_t* __restrict__ out_labels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    if (ws.done[b]) return;
    en2_merge_body(smem, boxes, scores, A, nc, max_out, ws, b, num_dets, out_boxes, out_scores, out_labels);
}

// ---------------------------------------------------------------------------------------------
// postprocess: restore coords, score filter, dedupe (custom_nms), int cast, inflate.
// ---------------------------------------------------------------------------------------------
constexpr int PP_MAX_SLOTS = 1024;
constexpr int PP_THREADS = 128;

__device__ __forceinline__ int floordiv_pos(int a, int b) {      // Python a // b for b > 0
    int q = a / b;
    if ((a % b != 0) && (a < 0)) --q;
    return q;
}
__device__ __forceinline__ int f2i(float v, int mode) {
    if (!(v == v)) return 0;
    v = fminf(fmaxf(v, -1.0e9f), 1.0e9f);
    if (mode == 1) v = rintf(v);                                 // half to even, like torch.round
    return (int)v;                                               // toward zero, like Python int()
}

__global__ __launch_bounds__(PP_THREADS) void postprocess_kernel(
    const int32_t* __restrict__ num_dets, const float* __restrict__ bboxes, const float* __restrict__ scores,
    const int32_t* __restrict__ labels, int slots, int np, const float* __restrict__ ratio,
    const float* __restrict__ dwdh, const int32_t* __restrict__ img_wh, float conf, float dedupe_iou, int coord_mode,
    int max_crops, int32_t* __restrict__ det_count, int32_t* __restrict__ det_box, float* __restrict__ det_score,
    int32_t* __restrict__ det_label, int32_t* __restrict__ crop_rect, int32_t* __restrict__ crop_ok) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = (uint64_t*)smem;                 // np
    float4* sb = (float4*)(keys + np);                // np
    uint64_t* alive = (uint64_t*)(sb + np);           // np/64 + 1
    int32_t* kept_pos = (int32_t*)(alive + (np >> 6) + 1);   // np
    int& n_pass_sh = *(int*)(kept_pos + np);                 // all LDS in the dynamic region (16-B aligned base)

    const int b = blockIdx.x, tid = threadIdx.x;
    int n = num_dets[b];
    n = n < 0 ? 0 : (n > slots ? slots : n);
    const float r = ratio[b], dw = dwdh[2 * b], dh = dwdh[2 * b + 1];
    const float4* Bx = (const float4*)bboxes + (size_t)b * slots;
    const float* S = scores + (size_t)b * slots;

    if (tid == 0) n_pass_sh = 0;
    __syncthreads();
    int mine = 0;
    for (int j = tid; j < np; j += PP_THREADS) {
        uint64_t k = ~0ull;
        if (j < n) {
            float sc = S[j];
            if (!(sc < conf)) {                       // `if score < 0.35: continue`
                k = ((uint64_t)desc_key(sc) << 32) | (uint32_t)j;
                ++mine;
            }
        }
        keys[j] = k;
    }
    if (mine) atomicAdd(&n_pass_sh, mine);
    for (int j = tid; j < slots; j += PP_THREADS) {
        size_t o = (size_t)b * slots + j;
        ((int4*)det_box)[o] = make_int4(0, 0, 0, 0);
        ((int4*)crop_rect)[o] = make_int4(0, 0, 0, 0);
        det_score[o] = 0.f;
        det_label[o] = 0;
        crop_ok[o] = 0;
    }
    bitonic_sort_u64(keys, np);
    const int m = n_pass_sh;
    for (int i = tid; i < m; i += PP_THREADS) {
        float4 v = Bx[(uint32_t)keys[i]];
        // bboxes -= dwdh ; bboxes /= ratio   (f32, two roundings per coordinate)
        v.x = __fdiv_rn(__fsub_rn(v.x, dw), r);
        v.y = __fdiv_rn(__fsub_rn(v.y, dh), r);
        v.z = __fdiv_rn(__fsub_rn(v.z, dw), r);
        v.w = __fdiv_rn(__fsub_rn(v.w, dh), r);
        sb[i] = v;
    }
    const int nwords = (np + 63) >> 6;
    alive_init(alive, m, nwords);
    __syncthreads();
    const int cap = max_crops > 0 ? (max_crops < m ? max_crops : m) : m;
    int nk;
    if (dedupe_iou > 0.0f) {
        nk = greedy_scan<0>(sb, nullptr, m, dedupe_iou, cap, alive, nwords,
                            [&](int rank, int pos) { kept_pos[rank] = pos; });
    } else {                                         // dedupe disabled: every filtered detection is reported
        nk = cap;
        for (int i = tid; i < nk; i += PP_THREADS) kept_pos[i] = i;
    }
    __syncthreads();
    const int W = img_wh[2 * b], H = img_wh[2 * b + 1];
    for (int k = tid; k < nk; k += PP_THREADS) {
        const int pos = kept_pos[k];
        const uint32_t slot = (uint32_t)keys[pos];
        const float4 v = sb[pos];
        int x_min = f2i(v.x, coord_mode), y_min = f2i(v.y, coord_mode);
        int x_max = f2i(v.z, coord_mode), y_max = f2i(v.w, coord_mode);
        size_t o = (size_t)b * slots + k;
        ((int4*)det_box)[o] = make_int4(x_min, y_min, x_max, y_max);
        det_score[o] = S[slot];
        det_label[o] = labels[(size_t)b * slots + slot];
        // crop_image, eval branch (utils/trainClass.py:76-77,85-91)
        int dis_x = floordiv_pos(x_max - x_min, 10), dis_y = floordiv_pos(y_max - y_min, 10);
        int hx = floordiv_pos(dis_x, 2), hy = floordiv_pos(dis_y, 2);
        int cx1 = min(W, x_max + hx), cx0 = max(0, x_min - hx);
        int cy1 = min(H, y_max + hy), cy0 = max(0, y_min - hy);
        ((int4*)crop_rect)[o] = make_int4(cx0, cy0, cx1, cy1);
        crop_ok[o] = (cx1 > cx0 && cy1 > cy0) ? 1 : 0;
    }
    if (tid == 0) det_count[b] = nk;
}

__global__ void compact_crops_kernel(const int32_t* __restrict__ det_count, const int32_t* __restrict__ crop_rect,
                                     const int32_t* __restrict__ crop_ok, int B, int slots, int cap, int parts,
                                     int32_t* __restrict__ crop_list, int32_t* __restrict__ crop_total) {
    // single workgroup: images are few (<= a few thousand); order = image, then detection rank
    __shared__ int32_t base_sh;
    __shared__ uint32_t wave_cnt[16];
    const int tid = threadIdx.x;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    const int total_items = B * slots;
    for (int c0 = 0; c0 < total_items; c0 += blockDim.x) {
        int i = c0 + tid;
        bool ok = false;
        int b = 0, k = 0;
        if (i < total_items) {
            b = i / slots;
            k = i - b * slots;
            ok = k < det_count[b] && crop_ok[i] != 0;
        }
        uint32_t tot;
        uint32_t rnk = block_rank(ok, wave_cnt, &tot);
        int base = base_sh;
        if (ok) {
            int pos = base + (int)rnk;
            if (pos < cap) {
                int4 rc = ((const int4*)crop_rect)[i];
                int32_t* o = crop_list + (size_t)pos * 6;
                o[0] = b; o[1] = rc.x; o[2] = rc.y; o[3] = rc.z; o[4] = rc.w; o[5] = k;
            }
        }
        __syncthreads();
        if (tid == 0) base_sh = base + (int)tot;
        __syncthreads();
    }
    int total = base_sh < cap ? base_sh : cap;
    for (int p = total + tid; p < cap; p += blockDim.x) {
        int32_t* o = crop_list + (size_t)p * 6;
        o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0;
    }
    if (tid == 0) crop_total[0] = total;
    // counts of `parts` equal slices of the list (ceil(cap / parts) entries each; the classifier runs the slices on concurrent streams)
    if (tid < parts) {
        const int base = (cap + parts - 1) / parts, lo = tid * base;
        const int n = cap - lo < base ? (cap - lo > 0 ? cap - lo : 0) : base;
        const int c = total - lo;
        crop_total[1 + tid] = c < 0 ? 0 : (c > n ? n : c);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------- C ABI
extern "C" size_t yv_custom_nms_ws_bytes(int n_sets, int n_max) {
    if (n_sets <= 0 || n_max <= 0) return 0;
    int np = 64;
    while (np < n_max) np <<= 1;
    return np <= CN_LDS_BOX_MAX ? 0 : (size_t)n_sets * (size_t)n_max * sizeof(float4);
}

extern "C" int yv_custom_nms(const float* boxes, const float* scores, const int32_t* counts, int n_sets, int n_max,
                             float iou_threshold, int32_t* keep, int32_t* num_keep, void* ws, size_t ws_bytes,
                             void* stream) {
    if (n_sets < 0 || n_max < 0 || !keep || !num_keep) return YV_ERR_ARG;
    if (n_sets == 0) return YV_OK;
    if (n_max == 0) {
        hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int32_t) * (size_t)n_sets, (hipStream_t)stream);
        return e == hipSuccess ? YV_OK : YV_ERR_LAUNCH;
    }
    if (!boxes || !scores) return YV_ERR_ARG;
    if (n_max > 16384) return YV_ERR_LIMIT;
    int np = 64;
    while (np < n_max) np <<= 1;
    size_t need = yv_custom_nms_ws_bytes(n_sets, n_max);
    if (need && (!ws || ws_bytes < need)) return YV_ERR_WORKSPACE;
    size_t lds = (size_t)np * 8 + (size_t)(((np >> 6) + 2) & ~1) * 8 + (np <= CN_LDS_BOX_MAX ? (size_t)np * 16 : 0);
    int threads = np < 1024 ? np : 1024;
    if (hipFuncSetAttribute((const void*)custom_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
        return YV_ERR_LAUNCH;
    hipLaunchKernelGGL(custom_nms_kernel, dim3(n_sets), dim3(threads), lds, (hipStream_t)stream, boxes, scores, counts,
                       n_max, np, iou_threshold, keep, num_keep, (float4*)ws);
    return yv_launch_status();
}

extern "C" int yv_efficient_nms(const float* boxes, const float* scores, int B, int A, int nc, float score_threshold,
                                float iou_threshold, int max_out, int pre_topk, int32_t* num_dets, float* out_boxes,
                                float* out_scores, int32_t* out_labels, void* stream) {
    if (B < 0 || A <= 0 || nc <= 0 || max_out <= 0 || !boxes || !scores || !num_dets || !out_boxes || !out_scores ||
        !out_labels)
        return YV_ERR_ARG;
    if (B == 0) return YV_OK;
    if (pre_topk <= 0 || pre_topk > EN_MAXK || nc > 32767 || (long long)A * nc > 0x7fffffffLL) return YV_ERR_LIMIT;
    size_t lds = (size_t)EN_MAXK * (8 + 16 + 2) + (EN_MAXK / 64) * 8 + sizeof(EnShared) + 16;
    if (hipFuncSetAttribute((const void*)efficient_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return YV_ERR_LAUNCH;
    hipLaunchKernelGGL(efficient_nms_kernel, dim3(B), dim3(EN_THREADS), lds, (hipStream_t)stream, boxes, scores, A, nc,
                       score_threshold, iou_threshold, max_out, pre_topk, num_dets, out_boxes, out_scores, out_labels);
    return yv_launch_status();
}

// capacity of an image's candidate list: every candidate of ordinary score sets fits (the fast path then never re-reads the
// scores), never less than pre_topk, never more than there are scores
static int en2_lcap(int A, int nc, int K) {
    const long long total = (long long)A * nc;
    // up to 44 K scores per image: one 4096-entry region per filter chunk (segmented mode); beyond: the 44 K best-effort list
    long long c = total <= (long long)EN2_HRW * EN_THREADS ? (total + 4095) / 4096 * 4096 : (long long)EN2_HRW * EN_THREADS;
    if (c < K) c = K;
    return (int)((c + 63) & ~63LL);
}

static size_t en2_layout(int B, int A, int nc, int max_out, int K, En2Ws* w, unsigned char* base, size_t* zero_bytes) {
    // [count (B) | ccount (B*nc) | nkept (B*nc) | done (B)] zeroed per call, then cand (B*K u64), kept (B*nc*max_out u64)
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_count = take((size_t)B * 4), o_cc = take((size_t)B * nc * 4), o_nk = take((size_t)B * nc * 4);
    const size_t o_done = take((size_t)B * 4);
    if (zero_bytes) *zero_bytes = off;
    const size_t o_seg = take((size_t)B * EN2_SEGS * 4);
    const int lcap = en2_lcap(A, nc, K);
    const size_t o_cand = take((size_t)B * lcap * 8), o_kept = take((size_t)B * nc * max_out * 8);
    if (w && base) {
        w->count = (uint32_t*)(base + o_count); w->ccount = (uint32_t*)(base + o_cc); w->nkept = (uint32_t*)(base + o_nk);
        w->done = (uint32_t*)(base + o_done); w->seg = (uint32_t*)(base + o_seg); w->nseg = 0;
        w->cand = (uint64_t*)(base + o_cand); w->kept = (uint64_t*)(base + o_kept); w->lcap = lcap;
    }
    return off;
}

extern "C" size_t yv_efficient_nms_ws_bytes(int B, int A, int nc, int max_out, int pre_topk) {
    if (B <= 0 || A <= 0 || nc <= 0 || max_out <= 0 || pre_topk <= 0 || pre_topk > EN_MAXK) return 0;
    return en2_layout(B, A, nc, max_out, pre_topk, nullptr, nullptr, nullptr);
}

extern "C" int yv_efficient_nms_ws(const float* boxes, const float* scores, int B, int A, int nc, float score_threshold,
                                   float iou_threshold, int max_out, int pre_topk, int32_t* num_dets, float* out_boxes,
                                   float* out_scores, int32_t* out_labels, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || A <= 0 || nc <= 0 || max_out <= 0 || !boxes || !scores || !num_dets || !out_boxes || !out_scores ||
        !out_labels)
        return YV_ERR_ARG;
    if (B == 0) return YV_OK;
    if (pre_topk <= 0 || pre_topk > EN_MAXK || nc > 32767 || (long long)A * nc > 0x7fffffffLL || max_out > EN_MAXK)
        return YV_ERR_LIMIT;
    if (B > 65535 || nc > 65535) return YV_ERR_LIMIT;            // grid.y / grid.x
    En2Ws w;
    size_t zero_bytes = 0;
    const size_t need = en2_layout(B, A, nc, max_out, pre_topk, &w, (unsigned char*)ws, &zero_bytes);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) return YV_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int total = A * nc;
    const int chunks = (total + EN2_FT * EN2_FPT - 1) / (EN2_FT * EN2_FPT);
    // segmented candidate lists (no counter to zero, no memset launch) whenever an image is at most EN2_SEGS chunks
    w.nseg = chunks <= EN2_SEGS && (long long)chunks * (EN2_FT * EN2_FPT) <= w.lcap ? chunks : 0;
    if (!w.nseg && hipMemsetAsync(ws, 0, zero_bytes, st) != hipSuccess) return YV_ERR_LAUNCH;
    hipLaunchKernelGGL(en2_filter_kernel, dim3(chunks, B), dim3(EN2_FT), 0, st, scores, total, nc, score_threshold, pre_topk, w);
    auto lds_of = [&](int cap, int nw) { return (size_t)cap * (8 + 16) + (size_t)max_out * 16 + 128 * 8 + (size_t)nw * 8 + 16; };
    const size_t head_dyn = (size_t)max_out * (16 + 8 + 2) + 16;
    const size_t lds1 = lds_of(EN_MAXK, 16) > head_dyn ? lds_of(EN_MAXK, 16) : head_dyn, lds0 = lds_of(1024, 4);
    // dynamic-LDS limits already granted, PER DEVICE (the attribute belongs to the device's copy of the kernel; they only ever grow;
    // atomics: request threads of different streams may race here - the worst case is a repeated, idempotent grant)
    static std::atomic<size_t> attr_front[64], attr_classes[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return YV_ERR_LAUNCH;
    if (lds1 > attr_front[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)en2_front_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) != hipSuccess)
            return YV_ERR_LAUNCH;
        attr_front[dev].store(lds1, std::memory_order_release);
    }
    if (lds0 > 32768 && lds0 > attr_classes[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)en2_classes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0) != hipSuccess)
            return YV_ERR_LAUNCH;
        attr_classes[dev].store(lds0, std::memory_order_release);
    }
    hipLaunchKernelGGL(en2_front_kernel, dim3(B), dim3(EN_THREADS), lds1, st, boxes, scores, A, nc, score_threshold, iou_threshold,
                       max_out, pre_topk, w, num_dets, out_boxes, out_scores, out_labels);
    if (nc <= EN2_TAIL_NC) {
        const size_t ldst = lds_of(1024, 16) > (size_t)EN_MAXK * 8 ? lds_of(1024, 16) : (size_t)EN_MAXK * 8;
        hipLaunchKernelGGL(en2_tail_kernel, dim3(B), dim3(1024), ldst, st, boxes, scores, A, nc, iou_threshold, max_out, pre_topk, w,
                           num_dets, out_boxes, out_scores, out_labels);
        return yv_launch_status();
    }
    hipLaunchKernelGGL(en2_classes_kernel, dim3(nc, B), dim3(256), lds0, st, boxes, A, nc, iou_threshold, max_out, pre_topk, w);
    hipLaunchKernelGGL(en2_merge_kernel, dim3(B), dim3(256), (size_t)EN_MAXK * 8, st, boxes, scores, A, nc, max_out, w, num_dets,
                       out_boxes, out_scores, out_labels);
    return yv_launch_status();
}

extern "C" int yv_postprocess_dets(const int32_t* num_dets, const float* bboxes, const float* scores,
                                   const int32_t* labels, int B, int slots, const float* ratio, const float* dwdh,
                                   const int32_t* img_wh, float conf_threshold, float dedupe_iou, int coord_mode,
                                   int max_crops, int32_t* det_count, int32_t* det_box, float* det_score,
                                   int32_t* det_label, int32_t* crop_rect, int32_t* crop_ok, void* stream) {
    if (B < 0 || slots <= 0 || !num_dets || !bboxes || !scores || !labels || !ratio || !dwdh || !img_wh ||
        !det_count || !det_box || !det_score || !det_label || !crop_rect || !crop_ok)
        return YV_ERR_ARG;
    if (coord_mode != 0 && coord_mode != 1) return YV_ERR_ARG;
    if (B == 0) return YV_OK;
    if (slots > PP_MAX_SLOTS) return YV_ERR_LIMIT;
    int np = 64;
    while (np < slots) np <<= 1;
    size_t lds = (size_t)np * (8 + 16 + 4) + ((size_t)(np >> 6) + 1) * 8 + 16;
    hipLaunchKernelGGL(postprocess_kernel, dim3(B), dim3(PP_THREADS), lds, (hipStream_t)stream, num_dets, bboxes,
                       scores, labels, slots, np, ratio, dwdh, img_wh, conf_threshold, dedupe_iou, coord_mode,
                       max_crops, det_count, det_box, det_score, det_label, crop_rect, crop_ok);
    return yv_launch_status();
}

extern "C" int yv_compact_crops(const int32_t* det_count, const int32_t* crop_rect, const int32_t* crop_ok, int B,
                                int slots, int cap, int32_t* crop_list, int32_t* crop_total, void* stream) {
    if (B < 0 || slots <= 0 || cap < 0 || !det_count || !crop_rect || !crop_ok || !crop_list || !crop_total)
        return YV_ERR_ARG;
    hipLaunchKernelGGL(compact_crops_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, det_count, crop_rect,
                       crop_ok, B, slots, cap, 0, crop_list, crop_total);
    return yv_launch_status();
}

extern "C" int yv_compact_crops_split(const int32_t* det_count, const int32_t* crop_rect, const int32_t* crop_ok, int B,
                                      int slots, int cap, int parts, int32_t* crop_list, int32_t* crop_total, void* stream) {
    if (B < 0 || slots <= 0 || cap < 0 || parts < 0 || parts > 64 || !det_count || !crop_rect || !crop_ok || !crop_list || !crop_total)
        return YV_ERR_ARG;
    hipLaunchKernelGGL(compact_crops_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, det_count, crop_rect,
                       crop_ok, B, slots, cap, parts, crop_list, crop_total);
    return yv_launch_status();
}

