// Detect-head decode:
//   make_anchors (cell centre + 0.5)                    docs/YOLO_TensorRT_Technical.md:14-30
//   DFL: softmax over 16 bins . arange(16) -> l,t,r,b    docs/YOLO_TensorRT_Technical.md:72-77
//   x1y1 = anchor - lt ; x2y2 = anchor + rb ; * stride  -> xyxy in input pixels
//   scores = sigmoid(cls)
// HBM-bound: 276 B read + 36 B written per anchor; one thread per (anchor, side)
// reads its 16 logits as four 16-byte loads; a 4-lane group shares one anchor so
// its box leaves as one float4.
#include "yv_common.h"

namespace {

struct DecodeArgs {
    const float* box[3];
    const float* cls[3];
    int hw[3];        // side length of each scale
    int a0[3];        // first anchor index of each scale
};

__global__ __launch_bounds__(256) void decode_kernel(DecodeArgs a, int cls_ld, int B, int A, int nc,
                                                     float* __restrict__ boxes, float* __restrict__ scores) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long anchor_g = gid >> 2;                 // (b, anchor)
    const int side = (int)(gid & 3);
    if (anchor_g >= (long long)B * A) return;
    const int b = (int)(anchor_g / A);
    const int an = (int)(anchor_g - (long long)b * A);
    const int s = an >= a.a0[2] ? 2 : (an >= a.a0[1] ? 1 : 0);
    const int w = a.hw[s];
    const int local = an - a.a0[s];
    const int y = local / w, x = local - y * w;
    const size_t pix = ((size_t)b * w + y) * w + x;
    const float4* p = (const float4*)(a.box[s] + pix * 64 + side * 16);
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 t = p[q];
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    float m = v[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) m = fmaxf(m, v[q]);
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        float e = expf(v[q] - m);
        den += e;
        num += e * (float)q;
    }
    const float dist = num / den;
    const float stride = (float)(8 << s);
    const float ac = ((side & 1) ? (float)y : (float)x) + 0.5f;
    const float coord = ((side < 2) ? (ac - dist) : (ac + dist)) * stride;
    // gather the 4 sides of this anchor (lanes 4k..4k+3) and store one float4
    const int lane = threadIdx.x & 63, base = lane & ~3;
    float c0 = __shfl(coord, base, 64), c1 = __shfl(coord, base + 1, 64);
    float c2 = __shfl(coord, base + 2, 64), c3 = __shfl(coord, base + 3, 64);
    if (side == 0) ((float4*)boxes)[anchor_g] = make_float4(c0, c1, c2, c3);
    // class scores: side-th lane of the group handles classes side, side+4, ...
    const float* cp = a.cls[s] + pix * cls_ld;
    for (int c = side; c < nc; c += 4) scores[anchor_g * nc + c] = 1.0f / (1.0f + expf(-cp[c]));
}

// ---------------------------------------------------------------------------------------------
// Fused Detect tail (round 3): the last 1 x 1 convolutions of both branches (cv2.i.2: 64 -> 4 x 16 box logits, cv3.i.2: c3 -> nc
// class logits) + the decode above, for the three scales in ONE launch.  The box / class logits never exist in HBM (2.3 MB per
// image written by six latency-bound launches and read back by the decode) and seven launches become one.
//   * input: per scale the NHWC bf16 buffer [pixels][c2 + c3] the two 3 x 3 convolutions cv2.i.1 / cv3.i.1 write (box features
//     in channels 0..63, class features behind them);
//   * a wave owns 16 pixels at a time: the MFMA takes the WEIGHT fragment as its A operand (rows = output channels) and the pixel
//     fragment as B, read straight from global memory in fragment shape (a pixel's 64 channels are used once: no LDS staging); all
//     weight fragments of both convolutions stay in registers for the wave's whole pixel range;
//   * accumulation order = the convolution kernel's (K step 64 as two 32-deep MFMAs, bias added last), and the DFL / sigmoid
//     arithmetic is the decode kernel's, statement for statement, on a 16-pixel x 64-channel f32 tile passed through LDS: results
//     are bit-identical to the unfused path (tests/test_gpu_boxes.py).
// c2 = 64 (YOLOv8 n / s / m), nc <= 16, c3 = 32 KC3 in {64, 128, 192}; anything else takes the unfused path.
// ---------------------------------------------------------------------------------------------
struct TailArgs {
    const uint16_t* feat[3];       // (B * hw * hw, ld) bf16
    const uint16_t* w2[3];         // (64, 64) bf16 row-major (out channel, in channel)
    const float* b2[3];            // (64)
    const uint16_t* w3[3];         // (16, c3) bf16, rows >= nc zero
    const float* b3[3];            // (16)
    int hw[3], a0[3], blk0[3];     // side length, first anchor, first workgroup of each scale
};

template <int KC3>
__global__ __launch_bounds__(256) void detect_tail_kernel(TailArgs a, int ld, int B, int A, int nc, int px_per_wg,
                                                          float* __restrict__ boxes, float* __restrict__ scores) {
    __shared__ __attribute__((aligned(16))) float tile[4][16][64 + 4];            // per wave: 16 pixels x 64 box logits
    __shared__ __attribute__((aligned(16))) float ctile[4][16][16 + 1];           // per wave: 16 pixels x 16 class logits
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = (int)blockIdx.x >= a.blk0[2] ? 2 : ((int)blockIdx.x >= a.blk0[1] ? 1 : 0);
    const int w = a.hw[s];
    const long long npx = (long long)B * w * w;
    const long long p0 = (long long)((int)blockIdx.x - a.blk0[s]) * px_per_wg;
    const int fr = lane & 15, fq = lane >> 4;
    // weight fragments (A operand): lane (row fr, k group fq) holds W[i * 16 + fr][ks * 32 + fq * 8 .. + 7]
    bf16x8 fw2[4][2], fw3[KC3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fw2[i][ks] = *(const bf16x8*)(a.w2[s] + (i * 16 + fr) * 64 + ks * 32 + fq * 8);
#pragma unroll
    for (int ks = 0; ks < KC3; ++ks) fw3[ks] = *(const bf16x8*)(a.w3[s] + fr * (KC3 * 32) + ks * 32 + fq * 8);
    float4 bb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bb[i] = *(const float4*)(a.b2[s] + i * 16 + fq * 4);
    const float4 bc = *(const float4*)(a.b3[s] + fq * 4);
    const float stride = (float)(8 << s);
    const int iters = px_per_wg / 64;                             // 16 pixels per wave and iteration
    // the pixel fragments of iteration it+1 are fetched before iteration it is computed (a wave's iterations were one dependent
    // HBM round trip each: 46 us for 74 MB)
    bf16x8 nx2[2], nx3[KC3];
    auto fetch = [&](int it) __attribute__((always_inline)) {
        const long long pb = p0 + (long long)(it * 4 + wave) * 16;
        long long pix = pb + fr < npx ? pb + fr : npx - 1;
        pix = pix < 0 ? 0 : pix;
        const uint16_t* fp = a.feat[s] + pix * ld;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) nx2[ks] = *(const bf16x8*)(fp + ks * 32 + fq * 8);
#pragma unroll
        for (int ks = 0; ks < KC3; ++ks) nx3[ks] = *(const bf16x8*)(fp + 64 + ks * 32 + fq * 8);
    };
    fetch(0);
    for (int it = 0; it < iters; ++it) {
        const long long pbase = p0 + (long long)(it * 4 + wave) * 16;
        if (pbase >= npx) break;                                  // (wave-uniform)
        bf16x8 fx2[2], fx3[KC3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fx2[ks] = nx2[ks];
#pragma unroll
        for (int ks = 0; ks < KC3; ++ks) fx3[ks] = nx3[ks];
        if (it + 1 < iters) fetch(it + 1);
        f32x4 ab[4], ac = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ab[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) ab[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw2[i][ks], fx2[ks], ab[i], 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < KC3; ++ks) ac = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw3[ks], fx3[ks], ac, 0, 0, 0);
        // accumulator: lane (pixel fr, fq) holds output channels i * 16 + fq * 4 + e
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(float4*)&tile[wave][fr][i * 16 + fq * 4] = make_float4(ab[i][0] + bb[i].x, ab[i][1] + bb[i].y, ab[i][2] + bb[i].z, ab[i][3] + bb[i].w);
        ctile[wave][fr][fq * 4 + 0] = ac[0] + bc.x; ctile[wave][fr][fq * 4 + 1] = ac[1] + bc.y;
        ctile[wave][fr][fq * 4 + 2] = ac[2] + bc.z; ctile[wave][fr][fq * 4 + 3] = ac[3] + bc.w;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- the decode kernel's arithmetic: lane = (pixel lane >> 2, side lane & 3) ----------------------------------
        const int pl = lane >> 2, side = lane & 3;
        const long long pg = pbase + pl;                          // pixel index inside the scale
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = *(const float4*)&tile[wave][pl][side * 16 + q * 4];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        float m = v[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) m = fmaxf(m, v[q]);
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float e = expf(v[q] - m);
            den += e;
            num += e * (float)q;
        }
        const float dist = num / den;
        const long long hw2 = (long long)w * w;
        const int b = (int)((pg < npx ? pg : npx - 1) / hw2);
        const int local = (int)((pg < npx ? pg : npx - 1) - (long long)b * hw2);
        const int y = local / w, x = local - y * w;
        const float acn = ((side & 1) ? (float)y : (float)x) + 0.5f;
        const float coord = ((side < 2) ? (acn - dist) : (acn + dist)) * stride;
        const int base = lane & ~3;
        const float c0 = __shfl(coord, base, 64), c1 = __shfl(coord, base + 1, 64);
        const float c2 = __shfl(coord, base + 2, 64), c3 = __shfl(coord, base + 3, 64);
        const long long anchor_g = (long long)b * A + a.a0[s] + local;
        if (pg < npx) {
            if (side == 0) ((float4*)boxes)[anchor_g] = make_float4(c0, c1, c2, c3);
            for (int c = side; c < nc; c += 4) scores[anchor_g * nc + c] = 1.0f / (1.0f + expf(-ctile[wave][pl][c]));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();                          // the tile is wave-private: reads retired before the next pass writes
    }
}

}  // namespace

extern "C" int yv_detect_decode(const float* box0, const float* box1, const float* box2, const float* cls0,
                                const float* cls1, const float* cls2, int cls_ld, int B, int size, int nc,
                                float* boxes, float* scores, void* stream) {
    if (!box0 || !box1 || !box2 || !cls0 || !cls1 || !cls2 || !boxes || !scores) return YV_ERR_ARG;
    if (B <= 0 || size <= 0 || (size % 32) || nc <= 0 || cls_ld < nc) return YV_ERR_ARG;
    DecodeArgs a;
    a.box[0] = box0; a.box[1] = box1; a.box[2] = box2;
    a.cls[0] = cls0; a.cls[1] = cls1; a.cls[2] = cls2;
    int A = 0;
    for (int s = 0; s < 3; ++s) {
        a.hw[s] = size / (8 << s);
        a.a0[s] = A;
        A += a.hw[s] * a.hw[s];
    }
    long long threads = (long long)B * A * 4;
    int blocks = (int)((threads + 255) / 256);
    hipLaunchKernelGGL(decode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, cls_ld, B, A, nc, boxes,
                       scores);
    return yv_launch_status();
}

extern "C" int yv_detect_tail(const void* feat0, const void* feat1, const void* feat2, int ld, int c3,
                              const void* const* w2, const float* const* b2, const void* const* w3, const float* const* b3,
                              int B, int size, int nc, float* boxes, float* scores, void* stream) {
    if (!feat0 || !feat1 || !feat2 || !w2 || !b2 || !w3 || !b3 || !boxes || !scores) return YV_ERR_ARG;
    if (B <= 0 || size <= 0 || (size % 32) || nc <= 0 || nc > 16 || ld < 64 + c3 || (ld & 7)) return YV_ERR_ARG;
    if (c3 != 64 && c3 != 128 && c3 != 192) return YV_ERR_LIMIT;
    TailArgs a;
    a.feat[0] = (const uint16_t*)feat0; a.feat[1] = (const uint16_t*)feat1; a.feat[2] = (const uint16_t*)feat2;
    const int px_per_wg = 256;                                    // four 16-pixel groups per wave: ~1,050 workgroups at batch 32
    int A = 0, blk = 0;
    for (int s = 0; s < 3; ++s) {
        if (!w2[s] || !b2[s] || !w3[s] || !b3[s]) return YV_ERR_ARG;
        a.w2[s] = (const uint16_t*)w2[s]; a.b2[s] = b2[s]; a.w3[s] = (const uint16_t*)w3[s]; a.b3[s] = b3[s];
        a.hw[s] = size / (8 << s);
        a.a0[s] = A;
        A += a.hw[s] * a.hw[s];
        a.blk0[s] = blk;
        const long long px = (long long)B * a.hw[s] * a.hw[s];
        if (px > 0x7fffffffLL) return YV_ERR_LIMIT;
        blk += (int)((px + px_per_wg - 1) / px_per_wg);
    }
    hipStream_t st = (hipStream_t)stream;
    if (c3 == 64) hipLaunchKernelGGL(detect_tail_kernel<2>, dim3(blk), dim3(256), 0, st, a, ld, B, A, nc, px_per_wg, boxes, scores);
    else if (c3 == 128) hipLaunchKernelGGL(detect_tail_kernel<4>, dim3(blk), dim3(256), 0, st, a, ld, B, A, nc, px_per_wg, boxes, scores);
    else hipLaunchKernelGGL(detect_tail_kernel<6>, dim3(blk), dim3(256), 0, st, a, ld, B, A, nc, px_per_wg, boxes, scores);
    return yv_launch_status();
}
