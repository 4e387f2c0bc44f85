// Non-GEMM pieces of the detector:
//   yv_stem_conv  blob (RGB u8 / 255, YOLOTensorRT_yolodet_py_解读.md:70-74) fused with model.0
//                 (Conv 3->C 3x3 s2 + bias + SiLU, test.ipynb:25): K = 27 is too thin for MFMA
//                 and the layer is HBM-bound (3 B read, 2C B written per output pixel), so it is
//                 a direct f32 VALU convolution that never materialises the f32 blob.
//   yv_sppf_pool  model.9's three chained MaxPool2d(5,1,2) == clipped 5/9/13 windows, one pass,
//                 written into the channel slices the following 1x1 conv concatenates.
#include "yv_common.h"

namespace {

template <int COUT>
__global__ __launch_bounds__(256) void stem_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                   const float* __restrict__ wgt, const float* __restrict__ bias,
                                                   uint16_t* __restrict__ out, int out_ld, long long npix) {
    // weights are indexed with compile-time constants only -> the compiler keeps them on the scalar path
    // (s_load + SGPR operands): no LDS, no per-lane weight traffic.  wgt layout here: [tap*3+c][COUT].
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
            // (x * (1/255) instead of x / 255: at most one f32 ulp apart, and a division is ~10 VALU instructions - 27 of them
            //  per output pixel were a third of this kernel)
            for (int c = 0; c < 3; ++c) x[(ky * 3 + kx) * 3 + c] = ok ? (float)px[c] * (1.0f / 255.0f) : 0.f;
        }
    uint16_t* o = out + p * out_ld;
#pragma unroll
    for (int cg = 0; cg < COUT; cg += 8) {
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = bias[cg + q];
#pragma unroll
        for (int k = 0; k < 27; ++k)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = fmaf(x[k], wgt[k * COUT + cg + q], acc[q]);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = acc[q] * __builtin_amdgcn_rcpf(1.0f + __expf(-acc[q]));      // SiLU (as gemm.hip silu_f)
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

// one workgroup per (image, 8-channel group): the H x W plane of 16-byte vectors sits in LDS and the three
// MaxPool2d(5,1,2) are chained exactly like the reference module (75 LDS reads per pixel instead of 169 global)
constexpr int SPPF_MAX_HW = 1600;          // 40 x 40 (P5 of a 1280 input)
__global__ __launch_bounds__(512) void sppf_kernel(uint16_t* __restrict__ buf, int H, int W, int ld, int c) {
    __shared__ uint4 plane[2][SPPF_MAX_HW];
    const int cgs = c >> 3;
    const int b = blockIdx.x / cgs, cg = blockIdx.x - b * cgs;
    const int hw = H * W;
    uint16_t* base = buf + (size_t)b * hw * ld + cg * 8;
    for (int i = threadIdx.x; i < hw; i += blockDim.x) plane[0][i] = *(const uint4*)(base + (size_t)i * ld);
    __syncthreads();
    for (int stage = 0; stage < 3; ++stage) {
        const uint4* in = plane[stage & 1];
        uint4* outp = plane[(stage + 1) & 1];
        for (int i = threadIdx.x; i < hw; i += blockDim.x) {
            const int y = i / W, x = i - y * W;
            float m[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) m[q] = -INFINITY;
            for (int dy = -2; dy <= 2; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -2; dx <= 2; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    max8(m, in[yy * W + xx]);
                }
            }
            const uint4 v = pack8(m);
            outp[i] = v;
            *(uint4*)(base + (size_t)i * ld + (stage + 1) * c) = v;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int yv_stem_conv(const uint8_t* images, int B, int H, int W, const float* weight, const float* bias,
                            int Cout, void* out, int out_ld, void* stream) {
    if (!images || !weight || !bias || !out || B <= 0 || H <= 0 || W <= 0) return YV_ERR_ARG;
    if ((H & 1) || (W & 1) || (out_ld & 7) || out_ld < Cout) return YV_ERR_ARG;
    if (Cout != 16 && Cout != 32 && Cout != 48) return YV_ERR_LIMIT;             // YOLOv8 n / s / m stems
    const long long npix = (long long)B * (H / 2) * (W / 2);
    const dim3 grid((unsigned)((npix + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    uint16_t* o = (uint16_t*)out;
    if (Cout == 16) hipLaunchKernelGGL(stem_kernel<16>, grid, block, 0, st, images, H, W, weight, bias, o, out_ld, npix);
    else if (Cout == 32) hipLaunchKernelGGL(stem_kernel<32>, grid, block, 0, st, images, H, W, weight, bias, o, out_ld, npix);
    else hipLaunchKernelGGL(stem_kernel<48>, grid, block, 0, st, images, H, W, weight, bias, o, out_ld, npix);
    return yv_launch_status();
}

extern "C" int yv_sppf_pool(void* buf, int B, int H, int W, int ld, int c, void* stream) {
    if (!buf || B <= 0 || H <= 0 || W <= 0 || c <= 0) return YV_ERR_ARG;
    if ((c & 7) || (ld & 7) || ld < 4 * c) return YV_ERR_ARG;
    if (H * W > SPPF_MAX_HW) return YV_ERR_LIMIT;
    hipLaunchKernelGGL(sppf_kernel, dim3(B * (c / 8)), dim3(512), 0, (hipStream_t)stream, (uint16_t*)buf, H, W, ld, c);
    return yv_launch_status();
}
