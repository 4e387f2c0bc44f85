// Wave-reduction / latency kernels around the ViT GEMMs (SURVEY.md rows B3 tail, B4):
//   yv_layernorm     timm blocks.*.norm1/2 and the final norm (eps 1e-6; README.md:21-29): f32 stream -> bf16
//   yv_cls_rows      cls_token + pos_embed[0] rows of the token stream (README.md:13-17)
//   yv_wrapper_head  Network_Wrapper.fc: ReLU -> Linear(1000,128) -> ReLU -> Linear(128,nc), ensemble mean, argmax
//                    (utils/utils.py:64-72, utils/trainClass.py:112)
// LayerNorm is HBM-bound (4 B read + 2 B written per element): one wave per row, the
// row lives in registers (two-pass mean / variance in f32), 16-byte loads, 8-byte stores.
#include "yv_common.h"

namespace {

constexpr int LN_MAXC = 4;          // float4 chunks per lane -> D <= 1024

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, size_t ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int rows, int D, float eps, uint16_t* __restrict__ y, size_t ldy,
                                                        const int32_t* __restrict__ count_dev, int rows_per_count) {
    if (count_dev) {
        long long r = (long long)count_dev[0] * rows_per_count;
        rows = r < rows ? (int)r : rows;
    }
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    const int nch = D >> 2;                       // float4 chunks in the row
    float4 v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nch) {
            v[i] = ((const float4*)xr)[c];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    uint16_t* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float4 g = ((const float4*)gamma)[c], b = ((const float4*)beta)[c];
            const float o0 = (v[i].x - mean) * rstd * g.x + b.x, o1 = (v[i].y - mean) * rstd * g.y + b.y;
            const float o2 = (v[i].z - mean) * rstd * g.z + b.z, o3 = (v[i].w - mean) * rstd * g.w + b.w;
            ((uint2*)yr)[c] = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
        }
    }
}

// LayerNorm whose output goes straight to the MXFP8 operand format of yv_linear_mxfp8 (e4m3 bytes + one E8M0 scale per 32
// consecutive elements): a block is 8 consecutive float4 chunks = 8 consecutive lanes of the wave that owns the row.
__global__ __launch_bounds__(256) void layernorm_mx_kernel(const float* __restrict__ x, size_t ldx,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int rows, int D, float eps, uint8_t* __restrict__ q, size_t ldq,
                                                           uint8_t* __restrict__ sc, long long rows_pad,
                                                           const int32_t* __restrict__ count_dev, int rows_per_count) {
    if (count_dev) {
        long long r = (long long)count_dev[0] * rows_per_count;
        rows = r < rows ? (int)r : rows;
    }
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    const int nch = D >> 2;
    float4 v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nch) {
            v[i] = ((const float4*)xr)[c];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            qq += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(qq) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c >= nch) continue;                                   // D is a multiple of 128: whole 8-lane groups drop out
        const float4 g = ((const float4*)gamma)[c], b = ((const float4*)beta)[c];
        // the bf16 rounding of the unfused path is kept, so both paths quantise the same numbers
        const float o0 = bf16_to_f32(f32_to_bf16((v[i].x - mean) * rstd * g.x + b.x));
        const float o1 = bf16_to_f32(f32_to_bf16((v[i].y - mean) * rstd * g.y + b.y));
        const float o2 = bf16_to_f32(f32_to_bf16((v[i].z - mean) * rstd * g.z + b.z));
        const float o3 = bf16_to_f32(f32_to_bf16((v[i].w - mean) * rstd * g.w + b.w));
        float amax = fmaxf(fmaxf(fabsf(o0), fabsf(o1)), fmaxf(fabsf(o2), fabsf(o3)));
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
        int e = -127;
        if (amax > 0.f) {
            int ex;
            const float mant = frexpf(amax * (1.0f / 448.0f), &ex);
            e = mant == 0.5f ? ex - 1 : ex;
            e = e < -127 ? -127 : (e > 127 ? 127 : e);
        }
        const float inv = ldexpf(1.0f, -e);
        int p = 0;
        p = __builtin_amdgcn_cvt_pk_fp8_f32(o0 * inv, o1 * inv, p, false);
        p = __builtin_amdgcn_cvt_pk_fp8_f32(o2 * inv, o3 * inv, p, true);
        *(uint32_t*)(q + (size_t)row * ldq + 4 * c) = (uint32_t)p;
        if ((lane & 7) == 0) {
            const int bk = c >> 3;
            sc[((long long)(bk >> 2) * rows_pad + row) * 4 + (bk & 3)] = (uint8_t)(e + 127);
        }
    }
}

__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, int R, int tok, int D,
                                float* __restrict__ x) {
    const int r = blockIdx.x;
    float* o = x + (size_t)r * (tok + 1) * D;
    for (int i = threadIdx.x; i < D; i += blockDim.x) o[i] = cls[i] + pos[i];
}

constexpr int WH_FEAT = 1000, WH_HID = 128, WH_MAX_NC = 32;

__global__ __launch_bounds__(256) void wrapper_head_kernel(const float* __restrict__ feats, int ldf,
                                                           const float* __restrict__ w1t, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           int nc, float scale, int accumulate,
                                                           float* __restrict__ logits, int32_t* __restrict__ labels,
                                                           const int32_t* __restrict__ r_dev) {
    // w1t is fc.1.weight TRANSPOSED: (1000, 128), so the 128 hidden units of one k are one coalesced 512-B row
    __shared__ float f[WH_FEAT];
    __shared__ float part[2][WH_HID];
    __shared__ float h[WH_HID];
    __shared__ float lg[WH_MAX_NC];
    const int r = blockIdx.x;
    const int tid = threadIdx.x;
    if (r_dev && r >= r_dev[0]) {
        // rows past the device-side crop count: logits 0, label -1 (written by the first ensemble member, so that the caller
        // need not zero / fill its output tensors with two more launches per batch)
        if (!accumulate) {
            if (tid < nc) logits[(size_t)r * nc + tid] = 0.f;
            if (tid == 0) labels[r] = -1;
        }
        return;
    }
    for (int i = tid; i < WH_FEAT; i += 256) f[i] = fmaxf(feats[(size_t)r * ldf + i], 0.f);     // ReLU
    __syncthreads();
    {
        const int u = tid & (WH_HID - 1), half = tid >> 7;                    // 2 K-halves x 128 units
        const int k0 = half * (WH_FEAT / 2);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        // 20 weight loads in flight per trip (the chain s0..s3 and its order are unchanged: same bits), 25 trips: with 4 loads
        // per trip the loop was 125 dependent L2 round trips - 28 us for 64 rows at the tail of every classifier pass
        static_assert((WH_FEAT / 2) % 20 == 0, "trip size");
        for (int k = k0; k < k0 + WH_FEAT / 2; k += 20) {
            float w[20];
#pragma unroll
            for (int j = 0; j < 20; ++j) w[j] = w1t[(size_t)(k + j) * WH_HID + u];
#pragma unroll
            for (int j = 0; j < 20; j += 4) {
                s0 = fmaf(f[k + j], w[j], s0);
                s1 = fmaf(f[k + j + 1], w[j + 1], s1);
                s2 = fmaf(f[k + j + 2], w[j + 2], s2);
                s3 = fmaf(f[k + j + 3], w[j + 3], s3);
            }
        }
        part[half][u] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (tid < WH_HID) h[tid] = fmaxf(part[0][tid] + part[1][tid] + b1[tid], 0.f);               // ReLU
    __syncthreads();
    if (tid < nc) {
        const float* wr = w2 + (size_t)tid * WH_HID;
        float s = 0.f;
        for (int k = 0; k < WH_HID; ++k) s = fmaf(h[k], wr[k], s);
        s = (s + b2[tid]) * scale;
        float* o = logits + (size_t)r * nc + tid;
        if (accumulate) s += *o;
        *o = s;
        lg[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        int best = 0;
        float bv = lg[0];
        for (int c = 1; c < nc; ++c)
            if (lg[c] > bv) { bv = lg[c]; best = c; }
        labels[r] = best;
    }
}

}  // namespace

extern "C" int yv_layernorm(const float* x, size_t ldx, const float* gamma, const float* beta, int rows, int D,
                            float eps, void* y, size_t ldy, const int32_t* count_dev, int rows_per_count,
                            void* stream) {
    if (!x || !gamma || !beta || !y || rows < 0 || D <= 0) return YV_ERR_ARG;
    if ((D & 3) || (ldx & 3) || (ldy & 3)) return YV_ERR_ARG;
    if (D > LN_MAXC * 256) return YV_ERR_LIMIT;
    if (rows == 0) return YV_OK;
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta,
                       rows, D, eps, (uint16_t*)y, ldy, count_dev, rows_per_count);
    return yv_launch_status();
}

extern "C" int yv_layernorm_mxfp8(const float* x, size_t ldx, const float* gamma, const float* beta, int rows, int D, float eps,
                                  void* q, size_t ldq, void* scales, long long rows_pad, const int32_t* count_dev,
                                  int rows_per_count, void* stream) {
    if (!x || !gamma || !beta || !q || !scales || rows < 0 || D <= 0) return YV_ERR_ARG;
    if ((D & 127) || (ldx & 3) || (ldq & 15) || rows_pad < rows || (rows_pad & 127)) return YV_ERR_ARG;
    if (D > LN_MAXC * 256) return YV_ERR_LIMIT;
    if (rows == 0) return YV_OK;
    hipLaunchKernelGGL(layernorm_mx_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, rows, D,
                       eps, (uint8_t*)q, ldq, (uint8_t*)scales, rows_pad, count_dev, rows_per_count);
    return yv_launch_status();
}

extern "C" int yv_cls_rows(const float* cls, const float* pos, int R, int tok, int D, float* x, void* stream) {
    if (!cls || !pos || !x || R < 0 || tok <= 0 || D <= 0) return YV_ERR_ARG;
    if (R == 0) return YV_OK;
    hipLaunchKernelGGL(cls_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, cls, pos, R, tok, D, x);
    return yv_launch_status();
}

extern "C" int yv_wrapper_head(const float* feats, int ldf, const float* w1t, const float* b1, const float* w2,
                               const float* b2, int R, int nc, float scale, int accumulate, float* logits,
                               int32_t* labels, const int32_t* r_dev, void* stream) {
    if (!feats || !w1t || !b1 || !w2 || !b2 || !logits || !labels || R < 0 || ldf < WH_FEAT) return YV_ERR_ARG;
    if (nc <= 0 || nc > WH_MAX_NC) return YV_ERR_LIMIT;
    if (R == 0) return YV_OK;
    hipLaunchKernelGGL(wrapper_head_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, feats, ldf, w1t, b1, w2, b2, nc,
                       scale, accumulate, logits, labels, r_dev);
    return yv_launch_status();
}
