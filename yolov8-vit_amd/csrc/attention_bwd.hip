// Attention backward (flash-style recompute; 197 tokens = one tile, 785 tokens = 256-row tiles; head dim 64) for the ViT
// fine-tune step (timm Attention, README.md:21-23).
//
// P is never stored: it is recomputed from Q, K and the forward's log-sum-exp.  Two kernels, each a sibling of
// the forward kernel and each free of cross-wave reductions and atomics (bitwise reproducible):
//   attn_bwd_dq   wave owns 32 QUERIES (query on the lane): S^T = K.Q^T, dP^T = V.dO^T, dS^T = P^T*(dP^T - delta),
//                 dQ^T = K^T.dS^T with dS^T fed to the MFMA straight from the accumulator registers; also emits
//                 delta[q] = sum_d dO*O for the second kernel.
//   attn_bwd_dkv  wave owns 32 KEYS (key on the lane): S = Q.K^T, dP = dO.V^T, then dV^T += dO^T.P and
//                 dK^T += Q^T.dS, again with P / dS taken from the accumulator registers as the B operand.
// S and dP are computed twice (7 products instead of 5); in exchange nothing is summed across waves.
#include "yv_common.h"

namespace {

constexpr int HD = 64;

__device__ __forceinline__ void stage_rows(unsigned char* dst, const uint16_t* base, size_t ld, int N, int NP, int tid,
                                           int T) {
    // NP rows x 64 bf16 -> 128-byte LDS rows, 16-byte chunk XOR-swizzled by ((row>>1)&7)
    for (int it = tid; it < NP * 8; it += T) {
        const int row = it >> 3, c = it & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < N) v = *(const uint4*)(base + (size_t)row * ld + c * 8);
        *(uint4*)(dst + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = v;
    }
}
__device__ __forceinline__ void stage_rows_t(unsigned char* dst, int stride, const uint16_t* base, size_t ld, int N, int NP,
                                             int tid, int T) {
    // transposed image [64][NP] (row stride `stride` bytes): dst[d][row] = src[row][d]
    for (int it = tid; it < NP * 8; it += T) {
        const int row = it >> 3, c = it & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < N) v = *(const uint4*)(base + (size_t)row * ld + c * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
            *(uint16_t*)(dst + (c * 8 + e) * stride + row * 2) = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
    }
}
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* img, int row, int chunk) {
    return *(const bf16x8*)(img + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
__device__ __forceinline__ bf16x8 frag_t(const unsigned char* img, int stride, int d, int k0) {
    const unsigned char* p = img + d * stride + k0 * 2;
    const uint2 lo = *(const uint2*)p, hi = *(const uint2*)(p + 16);
    const u32x4 pk = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(bf16x8, pk);
}

template <int NT>
__global__ __launch_bounds__(NT * 64) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ o,
                                                              const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                              int N, int H, int QB, float scale, float scale_log2e,
                                                              uint16_t* __restrict__ dqkv, float* __restrict__ delta) {
    // blockIdx.x = ((crop*H + head)*QB + query block); a query block = NT waves x 32 queries.  K/V are walked in
    // tiles of NP = 32*NT keys; the forward's log2-sum-exp makes every tile independent (no running max).
    constexpr int NP = NT * 32, TS = NP * 2 + 8, T = NT * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + NP * 128;
    unsigned char* Kt = smem + 2 * NP * 128;
    const int qb = blockIdx.x % QB, rh = blockIdx.x / QB;
    const int r = rh / H, hd = rh - r * H;
    const int D = H * HD, ld = 3 * D;
    const uint16_t* base = qkv + (size_t)r * N * ld + hd * HD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rl = lane & 31, hh = lane >> 5;
    const int q = qb * NP + wave * 32 + rl, qc = q < N ? q : N - 1;
    const uint16_t* orow = o + ((size_t)r * N + qc) * D + hd * HD;
    const uint16_t* drow = dout + ((size_t)r * N + qc) * D + hd * HD;
    bf16x8 fq[4], fdo[4];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fq[ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 16 + hh * 8);
        fdo[ks] = *(const bf16x8*)(drow + ks * 16 + hh * 8);
        const bf16x8 fo = *(const bf16x8*)(orow + ks * 16 + hh * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += (float)fdo[ks][j] * (float)fo[j];
    }
    dl += __shfl_xor(dl, 32, 64);
    if (q < N && hh == 0) delta[((size_t)r * H + hd) * N + q] = dl;
    const float lq = lse[((size_t)r * H + hd) * N + qc];
    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;

    for (int kv0 = 0; kv0 < N; kv0 += NP) {
        if (kv0 > 0) __syncthreads();
        const int nv = N - kv0;                               // valid keys of this tile (may exceed NP)
        stage_rows(Ks, base + (size_t)kv0 * ld + D, ld, nv, NP, tid, T);
        stage_rows(Vs, base + (size_t)kv0 * ld + 2 * D, ld, nv, NP, tid, T);
        stage_rows_t(Kt, TS, base + (size_t)kv0 * ld + D, ld, nv, NP, tid, T);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
            const int row = kt * 32 + rl;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, row, 2 * ks + hh), fq[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Vs, row, 2 * ks + hh), fdo[ks], dp, 0, 0, 0);
            }
            // dS^T = P^T * (dP^T - delta) * scale with P^T = exp2(S^T*c - lse[q])
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kv0 + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const float pv = key < N ? exp2f(s[e] * scale_log2e - lq) : 0.f;
                s[e] = pv * (dp[e] - dl) * scale;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 fp;
#pragma unroll
                for (int j = 0; j < 8; ++j) fp[j] = (__bf16)s[8 * st + j];
                const int key0 = kt * 32 + 16 * st + 4 * hh;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_t(Kt, TS, mt * 32 + rl, key0), fp, acc[mt], 0, 0, 0);
            }
        }
    }
    if (q < N) {
        uint16_t* dst = dqkv + ((size_t)r * N + q) * ld + hd * HD;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *(uint2*)(dst + mt * 32 + 8 * g4 + 4 * hh) =
                    make_uint2(pack_bf16x2(acc[mt][4 * g4], acc[mt][4 * g4 + 1]), pack_bf16x2(acc[mt][4 * g4 + 2], acc[mt][4 * g4 + 3]));
    }
}

template <int NT>
__global__ __launch_bounds__(NT * 64) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               int N, int H, int KB, float scale, float scale_log2e,
                                                               uint16_t* __restrict__ dqkv) {
    // blockIdx.x = ((crop*H + head)*KB + key block); wave owns 32 keys; the queries are walked in tiles of NP rows
    constexpr int NP = NT * 32, TS = NP * 2 + 8, T = NT * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Qs = smem;
    unsigned char* Os = smem + NP * 128;
    unsigned char* Qt = smem + 2 * NP * 128;
    unsigned char* Ot = Qt + 64 * TS;
    float* lse_s = (float*)(Ot + 64 * TS);
    float* del_s = lse_s + NP;
    const int kb = blockIdx.x % KB, rh = blockIdx.x / KB;
    const int r = rh / H, hd = rh - r * H;
    const int D = H * HD, ld = 3 * D;
    const uint16_t* base = qkv + (size_t)r * N * ld + hd * HD;
    const uint16_t* dbase = dout + (size_t)r * N * D + hd * HD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rl = lane & 31, hh = lane >> 5;
    const int key = kb * NP + wave * 32 + rl, kc = key < N ? key : N - 1;
    bf16x8 fk[4], fv[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fk[ks] = *(const bf16x8*)(base + (size_t)kc * ld + D + ks * 16 + hh * 8);
        fv[ks] = *(const bf16x8*)(base + (size_t)kc * ld + 2 * D + ks * 16 + hh * 8);
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) dk[mt][e] = dv[mt][e] = 0.f;

    for (int q0 = 0; q0 < N; q0 += NP) {
        if (q0 > 0) __syncthreads();
        const int nv = N - q0;
        stage_rows(Qs, base + (size_t)q0 * ld, ld, nv, NP, tid, T);
        stage_rows(Os, dbase + (size_t)q0 * D, D, nv, NP, tid, T);
        stage_rows_t(Qt, TS, base + (size_t)q0 * ld, ld, nv, NP, tid, T);
        stage_rows_t(Ot, TS, dbase + (size_t)q0 * D, D, nv, NP, tid, T);
        for (int i = tid; i < NP; i += T) {
            lse_s[i] = q0 + i < N ? lse[((size_t)r * H + hd) * N + q0 + i] : 0.f;
            del_s[i] = q0 + i < N ? delta[((size_t)r * H + hd) * N + q0 + i] : 0.f;
        }
        __syncthreads();
#pragma unroll 1
        for (int qt = 0; qt < NT; ++qt) {
            f32x16 sv, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) sv[e] = dp[e] = 0.f;
            const int row = qt * 32 + rl;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qs, row, 2 * ks + hh), fk[ks], sv, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Os, row, 2 * ks + hh), fv[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ql = qt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const bool ok = q0 + ql < N && key < N;
                const float pv = ok ? exp2f(sv[e] * scale_log2e - lse_s[ql]) : 0.f;
                sv[e] = pv;
                dp[e] = pv * (dp[e] - del_s[ql]) * scale;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 fp, fs;
#pragma unroll
                for (int j = 0; j < 8; ++j) { fp[j] = (__bf16)sv[8 * st + j]; fs[j] = (__bf16)dp[8 * st + j]; }
                const int qq = qt * 32 + 16 * st + 4 * hh;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    dv[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_t(Ot, TS, mt * 32 + rl, qq), fp, dv[mt], 0, 0, 0);
                    dk[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_t(Qt, TS, mt * 32 + rl, qq), fs, dk[mt], 0, 0, 0);
                }
            }
        }
    }
    if (key < N) {
        uint16_t* dst = dqkv + ((size_t)r * N + key) * ld + hd * HD;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = mt * 32 + 8 * g4 + 4 * hh;
                *(uint2*)(dst + D + d) = make_uint2(pack_bf16x2(dk[mt][4 * g4], dk[mt][4 * g4 + 1]),
                                                    pack_bf16x2(dk[mt][4 * g4 + 2], dk[mt][4 * g4 + 3]));
                *(uint2*)(dst + 2 * D + d) = make_uint2(pack_bf16x2(dv[mt][4 * g4], dv[mt][4 * g4 + 1]),
                                                        pack_bf16x2(dv[mt][4 * g4 + 2], dv[mt][4 * g4 + 3]));
            }
    }
}

template <int NT>
int launch_bwd(const uint16_t* qkv, const uint16_t* o, const uint16_t* dout, const float* lse, int R, int N, int H,
               float scale, uint16_t* dqkv, float* delta, hipStream_t st) {
    constexpr int NP = NT * 32, TS = NP * 2 + 8;
    const size_t lds_q = (size_t)2 * NP * 128 + 64 * TS;
    const size_t lds_kv = (size_t)2 * NP * 128 + 2 * 64 * TS + 2 * NP * 4;
    auto kq = attn_bwd_dq_kernel<NT>;
    auto kkv = attn_bwd_dkv_kernel<NT>;
    if (hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q) != hipSuccess ||
        hipFuncSetAttribute((const void*)kkv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv) != hipSuccess)
        return YV_ERR_LAUNCH;
    const float c = scale * 1.4426950408889634f;
    const int QB = (N + NP - 1) / NP;
    hipLaunchKernelGGL(kq, dim3(R * H * QB), dim3(NT * 64), lds_q, st, qkv, o, dout, lse, N, H, QB, scale, c, dqkv, delta);
    hipLaunchKernelGGL(kkv, dim3(R * H * QB), dim3(NT * 64), lds_kv, st, qkv, dout, lse, delta, N, H, QB, scale, c, dqkv);
    return yv_launch_status();
}

}  // namespace

extern "C" int yv_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, int R, int N, int H,
                                float scale, void* dqkv, float* delta_ws, void* stream) {
    if (!qkv || !out || !dout || !lse || !dqkv || !delta_ws || R <= 0 || N <= 0 || H <= 0) return YV_ERR_ARG;
    const uint16_t* q = (const uint16_t*)qkv;
    const uint16_t* o = (const uint16_t*)out;
    const uint16_t* d = (const uint16_t*)dout;
    uint16_t* g = (uint16_t*)dqkv;
    hipStream_t st = (hipStream_t)stream;
    switch (N > 256 ? 8 : (N + 31) / 32) {             // > 256 tokens: 256-row tiles on both axes
        case 1: return launch_bwd<1>(q, o, d, lse, R, N, H, scale, g, delta_ws, st);
        case 2: return launch_bwd<2>(q, o, d, lse, R, N, H, scale, g, delta_ws, st);
        case 3: return launch_bwd<3>(q, o, d, lse, R, N, H, scale, g, delta_ws, st);
        case 4: return launch_bwd<4>(q, o, d, lse, R, N, H, scale, g, delta_ws, st);
        case 5: return launch_bwd<5>(q, o, d, lse, R, N, H, scale, g, delta_ws, st);
        case 6: return launch_bwd<6>(q, o, d, lse, R, N, H, scale, g, delta_ws, st);
        case 7: return launch_bwd<7>(q, o, d, lse, R, N, H, scale, g, delta_ws, st);
        default: return launch_bwd<8>(q, o, d, lse, R, N, H, scale, g, delta_ws, st);
    }
}
