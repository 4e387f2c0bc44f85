// Dev tool (GPU box): issue-rate of the block-scaled FP8 MFMA vs the bf16 MFMA, registers only.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(int iters, float* out) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x38383838 + i; }
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)1.0f; hb[i] = (__bf16)0.5f; }
    const int sc = 0x7f7f7f7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, sc, 0, sc);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
double run(int wgs, int iters) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<wgs, 256>>>(iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<MODE><<<wgs, 256>>>(iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop_per = MODE == 0 ? 2.0 * 16 * 16 * 128 : 2.0 * 16 * 16 * 32;
    const double tf = flop_per * 8.0 * iters * 4 * wgs / (ms * 1e-3) / 1e12;
    hipFree(out);
    return tf;
}

int main() {
    for (int wgs : {256, 512, 1024}) {
        printf("wgs %4d: mxfp8 16x16x128 %8.1f TF   bf16 16x16x32 %8.1f TF\n", wgs, run<0>(wgs, 4000), run<1>(wgs, 4000));
    }
    return 0;
}
