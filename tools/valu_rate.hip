// Issue-rate microbenchmark for gfx950 (stand-alone: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip).
// One workgroup; every wave runs ITER x 16 instructions of one kind (independent chains unless noted) between two s_memtime
// stamps.  Modes pair instruction kinds on the waves w and w + 4 (assumed to share a SIMD) to see what overlaps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum Kind { K_NONE, K_EXP, K_FMA, K_PKFMA, K_PKADD, K_CVT, K_MAX3, K_MFMA_IND, K_MFMA_DEP, K_LDS128, K_MOV, K_PKMUL, K_MFMA16, K_N };
static const char* kind_name[] = {"none", "v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32", "v_max3_f32",
                                  "mfma32x32x16 (4 acc)", "mfma32x32x16 (1 acc)", "ds_read_b128", "v_mov_b32", "v_pk_mul_f32",
                                  "mfma16x16x32 (4 acc)"};
constexpr int ITER = 256;
__device__ int g_iters = ITER;

template <int KIND>
__device__ __forceinline__ unsigned long long run_kind(float seed, float* sink, const unsigned char* lds) {
    float a[16];
    f32x2 p[8];
    f32x16 acc[4];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc4[4];
    bf16x8 fa, fb;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ld[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f32x2{seed + i, seed - i};
#pragma unroll
    for (int i = 0; i < 4; ++i) { for (int e = 0; e < 16; ++e) acc[i][e] = 0.f; for (int e = 0; e < 4; ++e) acc4[i][e] = 0.f; }
#pragma unroll
    for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(seed + e); fb[e] = (__bf16)(seed - e); }
    const unsigned laddr = (threadIdx.x & 63) * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const int iters = g_iters;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == K_EXP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if constexpr (KIND == K_FMA) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
        } else if constexpr (KIND == K_MOV) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(seed));
        } else if constexpr (KIND == K_MAX3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 15]));
        } else if constexpr (KIND == K_PKFMA) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 7]) : "v"(p[(i + 3) & 7]));
        } else if constexpr (KIND == K_PKADD) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(p[(i + 3) & 7]));
        } else if constexpr (KIND == K_PKMUL) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(p[(i + 3) & 7]));
        } else if constexpr (KIND == K_CVT) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(seed));
        } else if constexpr (KIND == K_MFMA_IND) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i & 3], 0, 0, 0);
        } else if constexpr (KIND == K_MFMA_DEP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[0], 0, 0, 0);
        } else if constexpr (KIND == K_MFMA16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc4[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc4[i & 3], 0, 0, 0);
        } else if constexpr (KIND == K_LDS128) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[i & 3]) : "v"(laddr), "i"((i & 15) * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if constexpr (KIND == K_LDS128) asm volatile("" :: "v"(ld[0]), "v"(ld[1]), "v"(ld[2]), "v"(ld[3]));
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += p[i][0] + p[i][1];
#pragma unroll
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][7] + acc4[i][1];
    if (r == 12345.678f) *sink = r;
    return t1 - t0;
}

__global__ __launch_bounds__(512) void rate_kernel(int kindA, int kindB, float seed, float* sink, unsigned long long* out, int prioA = 0, int prioB = 0) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[17408];
    for (int i = threadIdx.x; i < 17408 / 4; i += blockDim.x) ((unsigned*)lds)[i] = i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const int kind = wave < 4 ? kindA : kindB;
    const int prio = wave < 4 ? prioA : prioB;
    if (prio == 1) asm volatile("s_setprio 1"); else if (prio == 2) asm volatile("s_setprio 2"); else if (prio == 3) asm volatile("s_setprio 3");
    unsigned long long t = 0;
    switch (kind) {
        case K_EXP: t = run_kind<K_EXP>(seed, sink, lds); break;
        case K_FMA: t = run_kind<K_FMA>(seed, sink, lds); break;
        case K_PKFMA: t = run_kind<K_PKFMA>(seed, sink, lds); break;
        case K_PKADD: t = run_kind<K_PKADD>(seed, sink, lds); break;
        case K_PKMUL: t = run_kind<K_PKMUL>(seed, sink, lds); break;
        case K_CVT: t = run_kind<K_CVT>(seed, sink, lds); break;
        case K_MAX3: t = run_kind<K_MAX3>(seed, sink, lds); break;
        case K_MFMA_IND: t = run_kind<K_MFMA_IND>(seed, sink, lds); break;
        case K_MFMA_DEP: t = run_kind<K_MFMA_DEP>(seed, sink, lds); break;
        case K_MFMA16: t = run_kind<K_MFMA16>(seed, sink, lds); break;
        case K_LDS128: t = run_kind<K_LDS128>(seed, sink, lds); break;
        case K_MOV: t = run_kind<K_MOV>(seed, sink, lds); break;
        default: break;
    }
    if ((threadIdx.x & 63) == 0) out[wave] = t;
}

int main() {
    float* sink; unsigned long long* out;
    if (hipMalloc(&sink, 4) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("no device\n"); return 1; }
    auto run = [&](int a, int b, int waves) {
        std::vector<unsigned long long> h(8, 0);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(out, 0, 64);
            hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(waves * 64), 0, 0, a, b, 1.0f, sink, out);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), out, 64, hipMemcpyDeviceToHost);
        printf("  %-24s | %-24s  %d waves: ticks per instruction, wave 0..%d:", kind_name[a], waves > 4 ? kind_name[b] : "-", waves, waves - 1);
        for (int w = 0; w < waves; ++w) printf(" %5.1f", (double)h[w] / (ITER * 16.0));
        printf("\n");
    };
    // clock rate of the stamp counter against the wall clock: one long launch
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        std::vector<unsigned long long> h(8, 0);
        hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(64), 0, 0, (int)K_FMA, 0, 1.0f, sink, out); hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(64), 0, 0, (int)K_EXP, 0, 1.0f, sink, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), out, 64, hipMemcpyDeviceToHost);
        printf("stamp counter: %llu ticks in a launch of %.1f us (upper bound of the tick period: %.2f ns)\n", h[0], ms * 1e3, ms * 1e6 / (double)h[0]);
    }
    {   // calibration proper: 64 x the iterations
        int big = ITER * 64, small = ITER;
        hipMemcpyToSymbol(HIP_SYMBOL(g_iters), &big, 4);
        std::vector<unsigned long long> h(8, 0);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(64), 0, 0, (int)K_EXP, 0, 1.0f, sink, out, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), out, 64, hipMemcpyDeviceToHost);
        printf("calibration: %llu ticks in %.1f us: %.3f ns per tick (%.2f GHz)\n", h[0], ms * 1e3, ms * 1e6 / (double)h[0], (double)h[0] / (ms * 1e6));
        hipMemcpyToSymbol(HIP_SYMBOL(g_iters), &small, 4);
    }
    auto runp = [&](int a, int b, int pa, int pb) {
        std::vector<unsigned long long> h(8, 0);
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(512), 0, 0, a, b, 1.0f, sink, out, pa, pb);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), out, 64, hipMemcpyDeviceToHost);
        printf("  %-22s prio %d | %-22s prio %d :", kind_name[a], pa, kind_name[b], pb);
        for (int w = 0; w < 8; ++w) printf(" %5.1f", (double)h[w] / (ITER * 16.0));
        printf("\n");
    };
    printf("priorities (s_setprio) and placement:\n");
    runp(K_MFMA_IND, K_EXP, 0, 3); runp(K_MFMA_IND, K_FMA, 0, 3); runp(K_MFMA_IND, K_PKFMA, 0, 3); runp(K_MFMA16, K_EXP, 0, 3); runp(K_MFMA16, K_FMA, 0, 3);
    runp(K_EXP, K_MFMA_IND, 0, 0); runp(K_FMA, K_MFMA_IND, 0, 0); runp(K_FMA, K_MFMA16, 0, 0); runp(K_EXP, K_MFMA_IND, 3, 0); runp(K_FMA, K_MFMA_IND, 3, 0);
    runp(K_MFMA16, K_MFMA16, 0, 0); runp(K_MFMA16, K_MFMA_IND, 0, 0); runp(K_FMA, K_MOV, 0, 0); runp(K_LDS128, K_FMA, 0, 0);
    printf("solo (one wave per SIMD):\n");
    for (int k = 1; k < K_N; ++k) run(k, 0, 4);
    printf("same kind on both waves of a SIMD:\n");
    for (int k = 1; k < K_N; ++k) run(k, k, 8);
    printf("pairs (waves 0-3 | waves 4-7):\n");
    const int pairs[][2] = {{K_MFMA_IND, K_EXP}, {K_MFMA_IND, K_FMA}, {K_MFMA_IND, K_PKFMA}, {K_MFMA_DEP, K_EXP}, {K_EXP, K_FMA}, {K_EXP, K_PKFMA},
                            {K_MFMA_IND, K_LDS128}, {K_EXP, K_LDS128}, {K_EXP, K_CVT}, {K_MFMA16, K_EXP}};
    for (auto& pr : pairs) run(pr[0], pr[1], 8);
    return 0;
}
