// crop + nearest resize + normalize (+ patch-major layout) in one gather pass.
//
// Replaces, per detected object (SURVEY.md section 8 rows B1 tail, B2):
//   PIL crop            utils/trainClass.py:92   (right/bottom exclusive)
//   A.Resize(224,224, INTER_NEAREST) + A.Normalize(.5,.5)   utils/trainClass.py:218-221 == app.py:39-42
//   HWC -> CHW          utils/trainClass.py:265-266
// and additionally writes the ViT patch-embed A operand directly (layout 2), so
// the conv k=s=P of timm's PatchEmbed becomes a plain GEMM with no im2col pass.
//
// HBM-bound integer/byte work: 224*224*3 source bytes gathered (crop windows are
// small and L2-resident) and one coalesced 16-byte store per 8 output values.
// The source-index tables follow OpenCV's rule  s = min(floor(d * (1/(dst/src))), src-1)
// evaluated in f64 with one rounding per operation (== the host's doubles),
// so the gather is exact by construction for any crop size.
#include "yv_common.h"

namespace {

constexpr int CR_THREADS = 256;
constexpr int CR_MAX_S = 512;
constexpr int CR_ROWS = 16;                  // source rows staged per block (= output rows per block)
constexpr int CR_BUF = 16 * 1032;             // bytes of the staging buffer (see the kernel)

__device__ __forceinline__ int nearest_src(int d, int dst, int src) {
    double fx = __ddiv_rn((double)dst, (double)src);
    double ifx = __ddiv_rn(1.0, fx);
    int s = (int)floor(__dmul_rn((double)d, ifx));
    return s < src - 1 ? s : src - 1;
}

// (x - 127.5) * fl32(1/127.5): albumentations' Normalize in f32, two roundings
__device__ __forceinline__ float norm_u8(uint8_t v, float rcp) { return __fmul_rn(__fsub_rn((float)v, 127.5f), rcp); }

// grid: (S / ROWS_PER_BLOCK, cap).  Each block produces ROWS output rows of one crop.
template <int LAYOUT>
__global__ __launch_bounds__(CR_THREADS) void crop_kernel(const uint8_t* __restrict__ images, int H, int W,
                                                          size_t img_stride, const int32_t* __restrict__ crop_list,
                                                          const int32_t* __restrict__ crop_total, int S, int P,
                                                          int rows_per_block, float rcp, void* __restrict__ out, int n_images,
                                                          int groups_per_block) {
    __shared__ int tx[CR_MAX_S];
    __shared__ int ty_sh[64];
    __shared__ int rshift[CR_ROWS];
    __shared__ __attribute__((aligned(16))) unsigned char rowbuf[LAYOUT == 2 ? CR_BUF : 16];
    const int r = blockIdx.y;
    if (crop_total && r >= crop_total[0]) return;
    const int32_t* rec = crop_list + (size_t)r * 6;
    const int img = rec[0], x0 = rec[1], y0 = rec[2], x1 = rec[3], y1 = rec[4];
    const int cw = x1 - x0, ch = y1 - y0;
    if (cw <= 0 || ch <= 0) return;      // degenerate rect: compaction never emits one; guard anyway
    // a record that does not lie inside an image of the batch is never read from (defence against a corrupted list:
    // an out-of-range gather would be a GPU memory fault)
    if (img < 0 || img >= n_images || x0 < 0 || y0 < 0 || x1 > W || y1 > H) return;
    for (int d = threadIdx.x; d < S; d += CR_THREADS) tx[d] = (x0 + nearest_src(d, S, cw)) * 3;
    const uint8_t* src = images + (size_t)img * img_stride;
    // A block walks groups_per_block consecutive row groups of its crop: the chain in front of the first useful byte - crop count,
    // crop record, the f64 column table, a barrier - is ~3 us of dependent latency, and at 1,024 crops x 14 groups the kernel was
    // 14,336 blocks of 21 KB each, waves parked 75 % of their cycles (profiles/r03_pmc_crop.json).  Few crops: one group per block
    // (the grid must still fill the chip).
    for (int gi = 0; gi < groups_per_block; ++gi) {
    const int row0 = (blockIdx.x * groups_per_block + gi) * rows_per_block;
    if (row0 >= S) break;
    if (gi) __syncthreads();             // the previous group's gathers are done with ty_sh / the staging buffer
    for (int d = threadIdx.x; d < rows_per_block; d += CR_THREADS) ty_sh[d] = y0 + nearest_src(row0 + d, S, ch);
    __syncthreads();
    const int groups = S >> 3;                       // 8 output pixels per item
    if (LAYOUT == 2) {
        // Source rows through LDS.  A per-lane BYTE gather from global memory costs the texture path about a lane per clock whatever
        // the cache does (measured on the detector's stem: 32 such loads per 64 pixels = 80-100 us per 32 images, 44 us once
        // staged), and this kernel did 24 of them per 8 output pixels.  The source-row segments [3 x0, 3 x1) of a PASS of rows are
        // copied with 8-byte loads of ALIGNED words (the word holding a valid byte lies in that byte's page, so the up to 7 bytes
        // read around a segment can never fault), then gathered byte by byte from LDS.  The buffer is 16.5 KB - eight blocks per
        // CU; at 33 KB / four blocks the kernel moved 3.8 TB/s, waves parked 75 % of their cycles, now 4.9 - and holds as many
        // rows per pass as fit: 16 for crops up to 336 pixels wide, 8 up to 680, ... 1 up to 5,496.
        const int words = (cw * 3 + 7 + 7) >> 3;             // 8-byte words that cover a segment at any alignment
        const int pitch = (words + 1 + (words & 1)) * 8;     // an ODD number of 8-byte words: the 16 rows of a column sit in 16 different banks
        int rpp = CR_BUF / pitch;
        rpp = rpp >= 16 ? 16 : rpp >= 8 ? 8 : rpp >= 4 ? 4 : rpp >= 2 ? 2 : rpp;
        if (rpp > rows_per_block) rpp = rows_per_block;
        const bool staged = rpp >= 1 && rows_per_block <= CR_ROWS;
        const int nrows = staged ? rpp : rows_per_block;
        const int lg_rows = 31 - __builtin_clz(nrows);      // (a power of two on the by-patch path)
        // patch-major bf16 (the classifier's operand): one item = 8 consecutive output pixels of a row, ALL THREE channels - the
        // 8 column-table reads and the row / patch arithmetic are shared by the channels, and the 24 source bytes are 8 runs
        // of 3 adjacent bytes (the first version gave each channel its own item: three times the LDS reads and index math for
        // the same bytes)
        const int items = nrows * groups;
        const int gp = S / P;                        // patches per side
        for (int r_begin = 0; r_begin < rows_per_block; r_begin += nrows) {
            if (staged) {
                if (r_begin) __syncthreads();                    // the previous pass's gathers are done with the buffer
                const int nst = nrows * words;
                // four words per thread and trip, ALL loaded before the first is stored: one memory round trip per trip (a loop of
                // load -> LDS store pairs paid the full HBM latency per iteration, 3-4 of them for a 150-pixel crop)
                for (int base = threadIdx.x; base < nst; base += 4 * CR_THREADS) {
                    uint2 wv[4];
                    unsigned char* dst[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int it = base + k * CR_THREADS;
                        dst[k] = nullptr;
                        if (it < nst) {
                            const int yp = it / words, u = it - yp * words;
                            const int yl = r_begin + yp;
                            const uintptr_t first = (uintptr_t)(src + (size_t)ty_sh[yl] * (size_t)W * 3 + (size_t)x0 * 3);
                            const int lead = (int)(first & 7);
                            if (u == 0) rshift[yl] = yp * pitch + lead - x0 * 3;   // LDS byte of source byte b of the row = rshift + b
                            if (u * 8 < lead + cw * 3) {         // only words that hold a byte of the segment
                                wv[k] = *(const uint2*)((first & ~(uintptr_t)7) + (size_t)u * 8);
                                dst[k] = rowbuf + yp * pitch + u * 8;
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (dst[k]) *(uint2*)dst[k] = wv[k];
                }
                __syncthreads();
            }
            auto run = [&](auto line_of) __attribute__((always_inline)) {
                for (int it = threadIdx.x; it < items; it += CR_THREADS) {
                    // one patch row per block (P = 16 = rows_per_block): consecutive lanes take the two 8-pixel halves of consecutive
                    // rows of ONE patch, whose (channel, 16 x 16) block is 512 contiguous bytes of the operand - a wave's store is
                    // two full blocks (16-row passes) instead of 32-byte pieces of 28 of them
                    const bool by_patch = P == 16 && rows_per_block == 16;
                    const int g = by_patch ? ((it >> (1 + lg_rows)) << 1) | (it & 1) : it % groups;
                    const int yl = r_begin + (by_patch ? (it >> 1) & (nrows - 1) : it / groups);
                    const int y = row0 + yl, x = g * 8;
                    const auto line = line_of(yl);
                    float v[3][8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const auto px = line + tx[x + q];
#pragma unroll
                        for (int c = 0; c < 3; ++c) v[c][q] = norm_u8(px[c], rcp);
                    }
                    const size_t row = (size_t)r * gp * gp + (size_t)(y / P) * gp + (x / P);
                    uint16_t* o = (uint16_t*)out + row * (size_t)(3 * P * P) + (y % P) * P + (x % P);
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        *(uint4*)(o + (size_t)c * P * P) = make_uint4(pack_bf16x2(v[c][0], v[c][1]), pack_bf16x2(v[c][2], v[c][3]),
                                                                      pack_bf16x2(v[c][4], v[c][5]), pack_bf16x2(v[c][6], v[c][7]));
                }
            };
            // (two instantiations so that the staged one reads LDS with ds_read_u8 rather than through a generic pointer)
            if (staged) run([&](int yl) __attribute__((always_inline)) { return (const uint8_t*)rowbuf + rshift[yl]; });
            else run([&](int yl) __attribute__((always_inline)) { return src + (size_t)ty_sh[yl] * (size_t)W * 3; });
        }
        continue;
    }
    const int items = rows_per_block * 3 * groups;
    for (int it = threadIdx.x; it < items; it += CR_THREADS) {
        int g = it % groups;
        int t = it / groups;
        int c = t % 3;
        int yl = t / 3;
        int y = row0 + yl;
        const uint8_t* line = src + (size_t)ty_sh[yl] * (size_t)W * 3 + c;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = norm_u8(line[tx[g * 8 + q]], rcp);
        if (LAYOUT == 0) {
            float* o = (float*)out + (((size_t)r * 3 + c) * S + y) * S + g * 8;
            ((float4*)o)[0] = make_float4(v[0], v[1], v[2], v[3]);
            ((float4*)o)[1] = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            uint16_t* o = (uint16_t*)out + (((size_t)r * 3 + c) * S + y) * S + g * 8;
            *(uint4*)o = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                    pack_bf16x2(v[6], v[7]));
        }
    }
    }   // row groups of this block
}


// Letterbox (YOLOTensorRT_yolodet_py_解读.md:67-69): aspect-preserving bilinear resize into the top-left
// anchored window (left, top, nw, nh) of an S x S canvas filled with 114.  Geometry (ratio, padding) is
// computed per image on the host in f64 exactly as the published upstream helper does; the kernel only
// resamples.  Sampling rule: src = (dst + 0.5) * (src_size / dst_size) - 0.5, edge-clamped, f32 blend,
// round half up (OpenCV's fixed-point INTER_LINEAR is not reproducible without the library: unpinned).
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ src, int Hc, int Wc,
                                                        const int32_t* __restrict__ geom, int S,
                                                        uint8_t* __restrict__ out) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= S || y >= S) return;
    const int32_t* gm = geom + b * 6;                       // w, h, nw, nh, left, top
    const int w = gm[0], h = gm[1], nw = gm[2], nh = gm[3], left = gm[4], top = gm[5];
    uint8_t* o = out + (((size_t)b * S + y) * S + x) * 3;
    const int dx = x - left, dy = y - top;
    if (dx < 0 || dx >= nw || dy < 0 || dy >= nh) { o[0] = o[1] = o[2] = 114; return; }
    const uint8_t* im = src + (size_t)b * Hc * Wc * 3;
    if (nw == w && nh == h) {
        const uint8_t* p = im + ((size_t)dy * Wc + dx) * 3;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        return;
    }
    float sx = ((float)dx + 0.5f) * ((float)w / (float)nw) - 0.5f;
    float sy = ((float)dy + 0.5f) * ((float)h / (float)nh) - 0.5f;
    sx = fminf(fmaxf(sx, 0.f), (float)(w - 1));
    sy = fminf(fmaxf(sy, 0.f), (float)(h - 1));
    const int x0 = (int)sx, y0 = (int)sy;
    const int x1 = x0 + 1 < w ? x0 + 1 : w - 1, y1 = y0 + 1 < h ? y0 + 1 : h - 1;
    const float fx = sx - (float)x0, fy = sy - (float)y0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = im[((size_t)y0 * Wc + x0) * 3 + c], bq = im[((size_t)y0 * Wc + x1) * 3 + c];
        const float cq = im[((size_t)y1 * Wc + x0) * 3 + c], d = im[((size_t)y1 * Wc + x1) * 3 + c];
        const float v = (a + (bq - a) * fx) * (1.f - fy) + (cq + (d - cq) * fx) * fy;
        o[c] = (uint8_t)fminf(fmaxf(floorf(v + 0.5f), 0.f), 255.f);
    }
}

int g_crop_gpb = 0;                 // > 0: forced row groups per block (yv_crop_debug)

}  // namespace

extern "C" int yv_crop_debug(int groups_per_block) { g_crop_gpb = groups_per_block; return YV_OK; }

extern "C" int yv_crop_resize_norm(const uint8_t* images, int B, int H, int W, size_t img_stride,
                                   const int32_t* crop_list, const int32_t* crop_total, int cap, int out_size,
                                   int patch, int layout, void* out, void* stream) {
    if (!images || !crop_list || !out || B <= 0 || H <= 0 || W <= 0 || cap < 0) return YV_ERR_ARG;
    if (layout < 0 || layout > 2) return YV_ERR_ARG;
    if (out_size <= 0 || out_size > CR_MAX_S || (out_size & 7)) return YV_ERR_LIMIT;
    if (layout == 2 && (patch < 8 || (patch & 7) || out_size % patch)) return YV_ERR_ARG;
    if (cap == 0) return YV_OK;
    // one block per 16 output rows (= one patch row at P=16)
    int rows = 16;
    if (out_size % rows) rows = 8;
    const float rcp = 1.0f / 127.5f;                 // fl32(1/127.5), computed once on the host
    // row groups per block: two where the list is long enough to keep every CU's eight block slots busy regardless
    const int ngroups = out_size / rows;
    // (measured at 1,024 crops x 14 groups: 1 -> 96 us, 2 -> 94, 4 -> 108, 14 -> 131; at 128 crops 1 -> 14.1, 2 -> 16.0)
    int gpb = g_crop_gpb > 0 ? g_crop_gpb : ((long long)cap * ngroups >= 8192 ? 2 : 1);
    gpb = gpb < 1 ? 1 : (gpb > ngroups ? ngroups : gpb);
    dim3 grid((ngroups + gpb - 1) / gpb, cap), block(CR_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (layout == 0)
        hipLaunchKernelGGL(crop_kernel<0>, grid, block, 0, st, images, H, W, img_stride, crop_list, crop_total,
                           out_size, patch, rows, rcp, out, B, gpb);
    else if (layout == 1)
        hipLaunchKernelGGL(crop_kernel<1>, grid, block, 0, st, images, H, W, img_stride, crop_list, crop_total,
                           out_size, patch, rows, rcp, out, B, gpb);
    else
        hipLaunchKernelGGL(crop_kernel<2>, grid, block, 0, st, images, H, W, img_stride, crop_list, crop_total,
                           out_size, patch, rows, rcp, out, B, gpb);
    return yv_launch_status();
}

extern "C" int yv_letterbox(const uint8_t* src, int B, int Hc, int Wc, const int32_t* geom, int S, uint8_t* out,
                            void* stream) {
    if (!src || !geom || !out || B <= 0 || Hc <= 0 || Wc <= 0 || S <= 0) return YV_ERR_ARG;
    dim3 grid((S + 15) / 16, (S + 15) / 16, B);
    hipLaunchKernelGGL(letterbox_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, Hc, Wc, geom, S, out);
    return yv_launch_status();
}
