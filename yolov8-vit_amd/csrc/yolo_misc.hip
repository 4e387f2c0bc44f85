// Non-GEMM pieces of the detector:
//   yv_stem_conv  blob (RGB u8 / 255, YOLOTensorRT_yolodet_py_解读.md:70-74) fused with model.0
//                 (Conv 3->C 3x3 s2 + bias + SiLU, test.ipynb:25): K = 27 is too thin for MFMA
//                 and the layer is HBM-bound (3 B read, 2C B written per output pixel), so it is
//                 a direct f32 VALU convolution that never materialises the f32 blob.
//   yv_sppf_pool  model.9's three chained MaxPool2d(5,1,2) == clipped 5/9/13 windows, one pass,
//                 written into the channel slices the following 1x1 conv concatenates.
#include "yv_common.h"

namespace {

constexpr int STEM_MAX_C = 64;

__global__ __launch_bounds__(256) void stem_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                   const float* __restrict__ wgt, const float* __restrict__ bias,
                                                   int Cout, uint16_t* __restrict__ out, int out_ld, long long npix) {
    __shared__ float ws[27 * STEM_MAX_C];       // [tap*3+c][cout]
    __shared__ float bs[STEM_MAX_C];
    for (int i = threadIdx.x; i < 27 * Cout; i += 256) {
        const int k = i / Cout, co = i - k * Cout;
        ws[i] = wgt[co * 27 + k];
    }
    for (int i = threadIdx.x; i < Cout; i += 256) bs[i] = bias[i];
    __syncthreads();
    const int Ho = H >> 1, Wo = W >> 1;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const int b = (int)(p / ((long long)Ho * Wo));
    const int rem = (int)(p - (long long)b * Ho * Wo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    float x[27];
    const uint8_t* src = img + (size_t)b * H * W * 3;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const uint8_t* px = src + ((size_t)(ok ? iy : 0) * W + (ok ? ix : 0)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) x[(ky * 3 + kx) * 3 + c] = ok ? (float)px[c] / 255.0f : 0.f;
        }
    uint16_t* o = out + p * out_ld;
    for (int cg = 0; cg < Cout; cg += 8) {
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = bs[cg + q];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const float4 w0 = *(const float4*)(ws + k * Cout + cg), w1 = *(const float4*)(ws + k * Cout + cg + 4);
            acc[0] = fmaf(x[k], w0.x, acc[0]); acc[1] = fmaf(x[k], w0.y, acc[1]);
            acc[2] = fmaf(x[k], w0.z, acc[2]); acc[3] = fmaf(x[k], w0.w, acc[3]);
            acc[4] = fmaf(x[k], w1.x, acc[4]); acc[5] = fmaf(x[k], w1.y, acc[5]);
            acc[6] = fmaf(x[k], w1.z, acc[6]); acc[7] = fmaf(x[k], w1.w, acc[7]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = acc[q] / (1.0f + __expf(-acc[q]));      // SiLU
        *(uint4*)(o + cg) = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]),
                                       pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
    }
}

__device__ __forceinline__ void max8(float* m, uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        m[2 * e] = fmaxf(m[2 * e], bf16_to_f32((uint16_t)(w[e] & 0xffff)));
        m[2 * e + 1] = fmaxf(m[2 * e + 1], bf16_to_f32((uint16_t)(w[e] >> 16)));
    }
}
__device__ __forceinline__ uint4 pack8(const float* m) {
    return make_uint4(pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]),
                      pack_bf16x2(m[6], m[7]));
}

__global__ __launch_bounds__(256) void sppf_kernel(uint16_t* __restrict__ buf, int H, int W, int ld, int c,
                                                   long long items) {
    const long long it = (long long)blockIdx.x * 256 + threadIdx.x;
    if (it >= items) return;
    const int cgs = c >> 3;
    const int cg = (int)(it % cgs);
    const long long pix = it / cgs;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const long long b = pix / ((long long)W * H);
    float m5[8], m9[8], m13[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) m5[q] = m9[q] = m13[q] = -INFINITY;
    for (int dy = -6; dy <= 6; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -6; dx <= 6; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const uint4 v = *(const uint4*)(buf + ((b * H + yy) * W + xx) * ld + cg * 8);
            const int r = max(abs(dy), abs(dx));
            max8(m13, v);
            if (r <= 4) max8(m9, v);
            if (r <= 2) max8(m5, v);
        }
    }
    uint16_t* o = buf + ((b * H + y) * W + x) * ld + cg * 8;
    *(uint4*)(o + c) = pack8(m5);
    *(uint4*)(o + 2 * c) = pack8(m9);
    *(uint4*)(o + 3 * c) = pack8(m13);
}

}  // namespace

extern "C" int yv_stem_conv(const uint8_t* images, int B, int H, int W, const float* weight, const float* bias,
                            int Cout, void* out, int out_ld, void* stream) {
    if (!images || !weight || !bias || !out || B <= 0 || H <= 0 || W <= 0) return YV_ERR_ARG;
    if ((H & 1) || (W & 1) || (Cout & 7) || Cout <= 0 || (out_ld & 7) || out_ld < Cout) return YV_ERR_ARG;
    if (Cout > STEM_MAX_C) return YV_ERR_LIMIT;
    const long long npix = (long long)B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, images, H,
                       W, weight, bias, Cout, (uint16_t*)out, out_ld, npix);
    return yv_launch_status();
}

extern "C" int yv_sppf_pool(void* buf, int B, int H, int W, int ld, int c, void* stream) {
    if (!buf || B <= 0 || H <= 0 || W <= 0 || c <= 0) return YV_ERR_ARG;
    if ((c & 7) || (ld & 7) || ld < 4 * c) return YV_ERR_ARG;
    const long long items = (long long)B * H * W * (c / 8);
    hipLaunchKernelGGL(sppf_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (uint16_t*)buf, H, W, ld, c, items);
    return yv_launch_status();
}
