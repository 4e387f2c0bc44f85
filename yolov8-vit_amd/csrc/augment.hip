// Training-time augmentation of classifier crops, fused with the patch-embed operand builder.
//
// Replaces, per training crop, the stochastic part of
//   data_transforms['train']            utils/trainClass.py:199-216
// that follows Resize + Normalize: HorizontalFlip, [RandomCrop(200) + PadIfNeeded], ShiftScaleRotate,
// ChannelShuffle, OneOf[GridDistortion | ElasticTransform], CoarseDropout.  The host draws the random
// numbers (yvhip/augment.py: one record per sample); this kernel applies a record in ONE gather pass
// over the normalised (B,3,S,S) f32 batch and writes the bf16 patch-major rows the patch-embed GEMM
// reads - the augmented image never exists in HBM in any other form.
//
// A record describes the composed inverse map  output pixel -> source pixel:
//   (x, y) --lut--> (cx, cy)        per-axis piecewise-linear tables (GridDistortion; identity otherwise)
//          --aff--> (u, v)          2x3 affine = inverse(ShiftScaleRotate) . inverse(Elastic affine)
//          --bilinear, REFLECT_101--> four integer taps in the flipped / crop-padded frame
//          --imap--> source column / row   integer tables (flip, crop offset + reflected padding)
//          --perm--> source channel
// followed by the dropout holes (value 0 = mid-grey after Normalize).
//
// HBM-bound gather: 4 taps per output value out of an L2-resident 600 KB sample, one 16-byte store per
// 8 values.  All arithmetic is f32 with one rounding per operation (-ffp-contract=off), in the order the
// oracle (oracle/augment.py) states, so the result is bit-exact against it.
#include "yv_common.h"

namespace {

constexpr int AUG_THREADS = 128;

// cv2.BORDER_REFLECT_101 index for |i| <= 4n + 1 (the coordinates are clamped to that range): folds instead of a division
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    const int per = 2 * n - 2;
    int m = i < 0 ? -i : i;                       // the pattern is even
#pragma unroll
    for (int k = 0; k < 3; ++k) m = m >= per ? m - per : m;        // |i| <= 4n + 1 < 3 * per for n >= 4; smaller n: loop below
    while (m >= per) m -= per;
    return m < n ? m : per - m;
}

// grid: B * (S/P)^2 workgroups, one per output row (patch)
__global__ __launch_bounds__(AUG_THREADS) void augment_kernel(const float* __restrict__ x, int S, int P,
                                                              const float* __restrict__ geo, const int32_t* __restrict__ idx,
                                                              uint16_t* __restrict__ out) {
    const int g = S / P;
    const int row = blockIdx.x;
    const int b = row / (g * g);
    const int gy = (row / g) % g, gx = row % g;
    const float* ge = geo + (size_t)b * (6 + 2 * S);
    const int32_t* id = idx + (size_t)b * (36 + 2 * S);
    const float a0 = ge[0], a1 = ge[1], a2 = ge[2], a3 = ge[3], a4 = ge[4], a5 = ge[5];
    const float* lutx = ge + 6;
    const float* luty = lutx + S;
    int nh = id[3];
    nh = nh < 0 ? 0 : (nh > 8 ? 8 : nh);
    const int32_t* holes = id + 4;
    const int32_t* mapx = id + 36;
    const int32_t* mapy = mapx + S;
    const float* src = x + (size_t)b * 3 * S * S;
    const float lim = (float)(4 * S);
    const int per_c = P * (P >> 3);
    const int items = 3 * per_c;
    for (int it = threadIdx.x; it < items; it += AUG_THREADS) {
        const int c = it / per_c;
        const int rem = it - c * per_c;
        const int py = rem / (P >> 3), p8 = rem - py * (P >> 3);
        const int oy = gy * P + py, ox0 = gx * P + p8 * 8;
        int sc = id[c];
        sc = sc < 0 ? 0 : (sc > 2 ? 2 : sc);
        const float* plane = src + (size_t)sc * S * S;
        const float cy = luty[oy];
        float vals[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ox = ox0 + j;
            const float cx = lutx[ox];
            float u = __fadd_rn(__fadd_rn(__fmul_rn(a0, cx), __fmul_rn(a1, cy)), a2);
            float v = __fadd_rn(__fadd_rn(__fmul_rn(a3, cx), __fmul_rn(a4, cy)), a5);
            u = fminf(fmaxf(u, -lim), lim);                      // also maps NaN to -lim: never an out-of-range index
            v = fminf(fmaxf(v, -lim), lim);
            const float uf = floorf(u), vf = floorf(v);
            const float fx = __fsub_rn(u, uf), fy = __fsub_rn(v, vf);
            const int ix = (int)uf, iy = (int)vf;
            int x0 = mapx[reflect101(ix, S)], x1 = mapx[reflect101(ix + 1, S)];
            int y0 = mapy[reflect101(iy, S)], y1 = mapy[reflect101(iy + 1, S)];
            x0 = min(max(x0, 0), S - 1); x1 = min(max(x1, 0), S - 1);        // tables come from the host: never trust them
            y0 = min(max(y0, 0), S - 1); y1 = min(max(y1, 0), S - 1);
            const float v00 = plane[(size_t)y0 * S + x0], v01 = plane[(size_t)y0 * S + x1];
            const float v10 = plane[(size_t)y1 * S + x0], v11 = plane[(size_t)y1 * S + x1];
            const float gx1 = __fsub_rn(1.0f, fx), gy1 = __fsub_rn(1.0f, fy);
            const float top = __fadd_rn(__fmul_rn(v00, gx1), __fmul_rn(v01, fx));
            const float bot = __fadd_rn(__fmul_rn(v10, gx1), __fmul_rn(v11, fx));
            float r = __fadd_rn(__fmul_rn(top, gy1), __fmul_rn(bot, fy));
            for (int h = 0; h < nh; ++h) {
                const int32_t* q = holes + h * 4;                                 // x1, y1, x2, y2 (right / bottom exclusive)
                if (ox >= q[0] && ox < q[2] && oy >= q[1] && oy < q[3]) r = 0.0f;
            }
            vals[j] = r;
        }
        uint4 pk;
        pk.x = pack_bf16x2(vals[0], vals[1]);
        pk.y = pack_bf16x2(vals[2], vals[3]);
        pk.z = pack_bf16x2(vals[4], vals[5]);
        pk.w = pack_bf16x2(vals[6], vals[7]);
        *reinterpret_cast<uint4*>(out + (size_t)row * (3 * P * P) + (size_t)c * P * P + py * P + p8 * 8) = pk;
    }
}

}  // namespace

extern "C" int yv_augment_patchify(const float* x, int B, int S, int P, const float* geo, const int32_t* idx, void* out,
                                   void* stream) {
    if (!x || !geo || !idx || !out) return YV_ERR_ARG;
    if (B < 0 || S <= 0 || P < 8 || (P & 7) || S % P) return YV_ERR_ARG;
    if (B == 0) return YV_OK;
    const int g = S / P;
    augment_kernel<<<dim3((unsigned)(B * g * g)), dim3(AUG_THREADS), 0, (hipStream_t)stream>>>(
        x, S, P, geo, idx, (uint16_t*)out);
    return yv_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Detector training augmentation (what `model.train()` of utils/trainYolo.py:28 applies by default):
// Mosaic(4) -> RandomPerspective(scale, translate) -> HSV gains -> horizontal flip, as ONE gather pass per output image.
// Sources are "tiles": images already resized to long side S in the top-left corner of an S x S slot (yv_letterbox).
// The 2S x 2S mosaic canvas is never built: a canvas pixel is looked up in the (at most four) placement rectangles.
//   rec_f (B,6)  f32: inverse affine, output pixel -> canvas coordinate
//   rec_i (B,34) i32: {n_tiles, flip} then per tile {tile id, x1a, y1a, x2a, y2a, x1b, y1b, 0}: canvas rectangle
//                     [x1a,x2a) x [y1a,y2a) shows the tile from (x1b, y1b)
//   lut   (B,3,256) u8: hue / saturation / value tables applied in 8-bit HSV (H in [0,180))
// Bilinear taps outside every rectangle read the fill value 114.  f32 arithmetic, one rounding per operation, in the
// order oracle/yolo_augment.py states.
namespace {

constexpr int MOS_THREADS = 256;

// one canvas pixel, all three channels: the first placement rectangle that contains (i, j) decides
__device__ __forceinline__ void mos_tap(const uint8_t* __restrict__ tiles, const int32_t* __restrict__ ri, int nt, int n_tiles_total,
                                        int S, int i, int j, float (&px)[3]) {
    px[0] = px[1] = px[2] = 114.0f;
    for (int t = 0; t < nt; ++t) {
        const int32_t* q = ri + 2 + t * 8;
        if (i >= q[1] && i < q[3] && j >= q[2] && j < q[4]) {
            const int id = q[0], sx = i - q[1] + q[5], sy = j - q[2] + q[6];
            if (id < 0 || id >= n_tiles_total || sx < 0 || sx >= S || sy < 0 || sy >= S) return;
            const uint8_t* p = tiles + (((size_t)id * S + sy) * S + sx) * 3;
            px[0] = (float)p[0]; px[1] = (float)p[1]; px[2] = (float)p[2];
            return;
        }
    }
}

__device__ __forceinline__ float round_half_up(float v) { return floorf(__fadd_rn(v, 0.5f)); }

__global__ __launch_bounds__(MOS_THREADS) void mosaic_kernel(const uint8_t* __restrict__ tiles, int n_tiles_total, int S,
                                                             const float* __restrict__ rec_f, const int32_t* __restrict__ rec_i,
                                                             const uint8_t* __restrict__ lut, uint8_t* __restrict__ out) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * MOS_THREADS + threadIdx.x;
    if (p >= S * S) return;
    const int y = p / S, x = p - y * S;
    const float* a = rec_f + (size_t)b * 6;
    const int32_t* ri = rec_i + (size_t)b * 34;
    int nt = ri[0];
    nt = nt < 0 ? 0 : (nt > 4 ? 4 : nt);
    const float xs = (float)(ri[1] ? S - 1 - x : x), ys = (float)y;
    float u = __fadd_rn(__fadd_rn(__fmul_rn(a[0], xs), __fmul_rn(a[1], ys)), a[2]);
    float v = __fadd_rn(__fadd_rn(__fmul_rn(a[3], xs), __fmul_rn(a[4], ys)), a[5]);
    const float lim = (float)(8 * S);
    u = fminf(fmaxf(u, -lim), lim);
    v = fminf(fmaxf(v, -lim), lim);
    const float uf = floorf(u), vf = floorf(v);
    const float fx = __fsub_rn(u, uf), fy = __fsub_rn(v, vf);
    const float gx = __fsub_rn(1.0f, fx), gy = __fsub_rn(1.0f, fy);
    const int i0 = (int)uf, j0 = (int)vf;
    float rgb[3], t00[3], t01[3], t10[3], t11[3];
    mos_tap(tiles, ri, nt, n_tiles_total, S, i0, j0, t00);
    mos_tap(tiles, ri, nt, n_tiles_total, S, i0 + 1, j0, t01);
    mos_tap(tiles, ri, nt, n_tiles_total, S, i0, j0 + 1, t10);
    mos_tap(tiles, ri, nt, n_tiles_total, S, i0 + 1, j0 + 1, t11);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = __fadd_rn(__fmul_rn(t00[c], gx), __fmul_rn(t01[c], fx));
        const float bot = __fadd_rn(__fmul_rn(t10[c], gx), __fmul_rn(t11[c], fx));
        rgb[c] = fminf(fmaxf(round_half_up(__fadd_rn(__fmul_rn(top, gy), __fmul_rn(bot, fy))), 0.0f), 255.0f);
    }
    // 8-bit HSV: V = max, S = 255 * (V - min) / V, H = half degrees in [0,180)
    const float R = rgb[0], G = rgb[1], Bc = rgb[2];
    const float vmax = fmaxf(R, fmaxf(G, Bc)), vmin = fminf(R, fminf(G, Bc));
    const float diff = __fsub_rn(vmax, vmin);
    float h = 0.0f, s = 0.0f;
    if (vmax > 0.0f) s = round_half_up(__fdiv_rn(__fmul_rn(255.0f, diff), vmax));
    if (diff > 0.0f) {
        if (vmax == R) h = __fdiv_rn(__fmul_rn(60.0f, __fsub_rn(G, Bc)), diff);
        else if (vmax == G) h = __fadd_rn(120.0f, __fdiv_rn(__fmul_rn(60.0f, __fsub_rn(Bc, R)), diff));
        else h = __fadd_rn(240.0f, __fdiv_rn(__fmul_rn(60.0f, __fsub_rn(R, G)), diff));
        if (h < 0.0f) h = __fadd_rn(h, 360.0f);
    }
    int h8 = (int)round_half_up(__fmul_rn(h, 0.5f));
    if (h8 >= 180) h8 -= 180;
    const uint8_t* l = lut + (size_t)b * 768;
    const float H2 = (float)l[h8], S2 = (float)l[256 + (int)s], V2 = (float)l[512 + (int)vmax];
    // back: sector = H / 30 (half degrees), f = fractional part
    const float hs = __fdiv_rn(H2, 30.0f);
    const float sec = floorf(hs);
    const float f = __fsub_rn(hs, sec);
    const float sn = __fdiv_rn(S2, 255.0f);
    const float pp = __fmul_rn(V2, __fsub_rn(1.0f, sn));
    const float qq = __fmul_rn(V2, __fsub_rn(1.0f, __fmul_rn(sn, f)));
    const float tt = __fmul_rn(V2, __fsub_rn(1.0f, __fmul_rn(sn, __fsub_rn(1.0f, f))));
    const int si = ((int)sec) % 6;
    float r2, g2, b2;
    switch (si) {
        case 0: r2 = V2; g2 = tt; b2 = pp; break;
        case 1: r2 = qq; g2 = V2; b2 = pp; break;
        case 2: r2 = pp; g2 = V2; b2 = tt; break;
        case 3: r2 = pp; g2 = qq; b2 = V2; break;
        case 4: r2 = tt; g2 = pp; b2 = V2; break;
        default: r2 = V2; g2 = pp; b2 = qq; break;
    }
    uint8_t* o = out + (((size_t)b * S + y) * S + x) * 3;
    o[0] = (uint8_t)fminf(fmaxf(round_half_up(r2), 0.0f), 255.0f);
    o[1] = (uint8_t)fminf(fmaxf(round_half_up(g2), 0.0f), 255.0f);
    o[2] = (uint8_t)fminf(fmaxf(round_half_up(b2), 0.0f), 255.0f);
}

}  // namespace

extern "C" int yv_mosaic_augment(const uint8_t* tiles, int n_tiles, int B, int S, const float* rec_f, const int32_t* rec_i,
                                 const uint8_t* lut, uint8_t* out, void* stream) {
    if (!tiles || !rec_f || !rec_i || !lut || !out) return YV_ERR_ARG;
    if (n_tiles <= 0 || B < 0 || S <= 0) return YV_ERR_ARG;
    if (B == 0) return YV_OK;
    mosaic_kernel<<<dim3((unsigned)((S * S + MOS_THREADS - 1) / MOS_THREADS), (unsigned)B), dim3(MOS_THREADS), 0,
                    (hipStream_t)stream>>>(tiles, n_tiles, S, rec_f, rec_i, lut, out);
    return yv_launch_status();
}
