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

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    const int per = 2 * n - 2;
    int m = i % per;
    if (m < 0) m += per;
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
