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

// The same stem as an implicit GEMM on the matrix pipes (used when a row of the output is a whole number of 64-pixel
// chunks, e.g. 640 x 640 inputs): pixel values 0..255 are exact in bf16, so the product runs on INTEGER activations against
// w / 255 split into a bf16 high and low half (two MFMAs per 16 pixels x 16 channels; the sum carries 16 weight bits, the
// accumulation is f32) - against 432 scalar FMAs per output pixel on the VALU, which made stem_kernel the slowest launch of the
// detector (100 us per 32 images for 144 MB of traffic).  K = 27 -> 32 slots, ordered so that a lane's 8 slots are 8
// CONSECUTIVE bytes of one input row: lane group g < 3 holds bytes 0..7 of the 9-byte run (3 pixels x 3 channels) of kernel
// row g, group 3 holds byte 8 of the three rows and five zeros.  One wave = 64 consecutive output pixels of one row.
template <int COUT>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                        const float* __restrict__ wgt, const float* __restrict__ bias,
                                                        uint16_t* __restrict__ out, int out_ld, int chunks) {
    constexpr int NFR = COUT / 16;
    __shared__ __attribute__((aligned(16))) unsigned char stage[4 * 3 * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int Ho = H >> 1, Wo = W >> 1, cpr = Wo >> 6;
    bf16x8 wh[NFR], wl[NFR];
    float4 bv[NFR];
#pragma unroll
    for (int i = 0; i < NFR; ++i) {
        uint32_t ph[4], pl[4];
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            float hi[2], lo[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int q = q2 * 2 + e;
                const int k = fq < 3 ? fq * 9 + q : (q < 3 ? q * 9 + 8 : -1);
                const float w = k >= 0 ? wgt[k * COUT + i * 16 + fr] * (1.0f / 255.0f) : 0.f;
                hi[e] = bf16_to_f32((uint16_t)(pack_bf16x2(w, 0.f) & 0xffff));
                lo[e] = w - hi[e];
            }
            ph[q2] = pack_bf16x2(hi[0], hi[1]);
            pl[q2] = pack_bf16x2(lo[0], lo[1]);
        }
        const u32x4 vh = {ph[0], ph[1], ph[2], ph[3]}, vl = {pl[0], pl[1], pl[2], pl[3]};
        wh[i] = __builtin_bit_cast(bf16x8, vh);
        wl[i] = __builtin_bit_cast(bf16x8, vl);
        bv[i] = *(const float4*)(bias + i * 16 + fq * 4);
    }
    // grid-stride over the chunks: the weight fragments above cost every wave a dependent load round trip, so a wave takes
    // several chunks (one chunk per wave: 80 us per 32 images; the grid below: see yv_stem_conv)
    for (int chunk = blockIdx.x * 4 + wave; chunk < chunks; chunk += gridDim.x * 4) {      // (uniform per wave)
    const int row = chunk / cpr, c64 = chunk - row * cpr;    // row = b * Ho + oy
    const int b = row / Ho, oy = row - b * Ho;
    // The chunk's input - three rows of 129 pixels, 387 bytes each - is staged in a wave-private LDS window with ONE 8-byte
    // buffer load per lane and row (lane-contiguous 512 bytes; bytes before the row start / above the image are out of range
    // and read 0 = the zero padding), then picked byte by byte from LDS: as per-lane global byte loads the same 32 accesses per
    // chunk took 80-100 us per 32 images, whatever the arithmetic around them (the texture path handles a byte gather at
    // about a lane per clock).
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (size_t)b * H * W * 3), 0, H * W * 3, 0x00020000);
    const int rowb = W * 3;
    constexpr uint32_t OOB = 0x80000000u;
    const int x0 = 6 * (c64 * 64) - 3;                        // first byte of the chunk's run in a row (-3 at the left border)
    const int xa = x0 & ~7;                                   // window start, 8-byte aligned (floor also for -3 -> -8)
    unsigned char* win = stage + wave * (3 * 512);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy + ky - 1;
        const int xo = xa + lane * 8;                         // this lane's 8 bytes of the row (may start before / end after it)
        // bytes of one row only: a window that ran past the row end would read the next row's first pixels (never used:
        // the run ends at 6 ox + 5 <= 3 W - 1), before its start it must read zeros
        // (window, row start and row end are all multiples of 8 bytes - W % 128 == 0 - so no 8-byte group straddles an edge)
        const uint32_t o = (iy >= 0 && xo >= 0 && xo + 8 <= rowb) ? (uint32_t)(iy * rowb + xo) : OOB;
        const u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(rs, o, 0, 0);
        *(u32x2*)(win + ky * 512 + lane * 8) = d;             // wave-private window: LDS executes one wave's accesses in order
    }
    // LDS byte addresses of this lane's 8 slots relative to the pixel's run start (6 ox - 3 - xa)
    int loff[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ky = fq < 3 ? fq : (q < 3 ? q : 0), t = fq < 3 ? q : 8;
        loff[q] = ky * 512 + t - xa;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ox = c64 * 64 + g * 16 + fr;
        const int xb = 6 * ox - 3;
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t byte = (fq < 3 || q < 3) ? (uint32_t)win[loff[q] + xb] : 0u;
            v[q] = __float_as_uint((float)byte);
        }
        uint32_t pk[4];
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2)                        // integers <= 255 are exact in bf16: the top halves of their f32 forms
            pk[q2] = __builtin_amdgcn_perm(v[2 * q2 + 1], v[2 * q2], 0x07060302u);
        const u32x4 pa = {pk[0], pk[1], pk[2], pk[3]};
        const bf16x8 fa = __builtin_bit_cast(bf16x8, pa);
        uint16_t* o = out + ((size_t)row * Wo + ox) * out_ld + fq * 4;
#pragma unroll
        for (int i = 0; i < NFR; ++i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], fa, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], fa, acc, 0, 0, 0);
            float r[4] = {acc[0] + bv[i].x, acc[1] + bv[i].y, acc[2] + bv[i].z, acc[3] + bv[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = r[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-r[e]));      // SiLU
            *(uint2*)(o + i * 16) = make_uint2(pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3]));
        }
    }
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
    if ((W % 128) == 0 && npix < 0x7fffffffLL && (long long)H * W * 3 < 0x7fffffffLL && ((uintptr_t)images & 7) == 0) {   // matrix-pipe form
        const int chunks = (int)(npix / 64);
        const int blocks = (chunks + 3) / 4;
        const dim3 g4((unsigned)(blocks < 2048 ? blocks : 2048));          // 8 waves per SIMD in one pass, ~6 chunks per wave at 32 x 640 x 640
        if (Cout == 16) hipLaunchKernelGGL(stem_mfma_kernel<16>, g4, block, 0, st, images, H, W, weight, bias, o, out_ld, chunks);
        else if (Cout == 32) hipLaunchKernelGGL(stem_mfma_kernel<32>, g4, block, 0, st, images, H, W, weight, bias, o, out_ld, chunks);
        else hipLaunchKernelGGL(stem_mfma_kernel<48>, g4, block, 0, st, images, H, W, weight, bias, o, out_ld, chunks);
        return yv_launch_status();
    }
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
