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
