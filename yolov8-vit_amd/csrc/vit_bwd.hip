// Backward-pass support kernels of the ViT fine-tune step (reference step order
// utils/trainClass.py:392-407: forward, build_loss, backward, SGD step).  The heavy products reuse the MFMA
// GEMM (dgrad: dY . W^T-copy, wgrad: dY^T . X^T with the token dimension as K); this file holds the
// HBM-bound glue around it:
//   yv_transpose_bf16    activations / gradients -> token-major copies for wgrad (zero padded to 64 rows)
//   yv_cast_weights      f32 master weight -> bf16 (N,K) and bf16 transposed (K,N) working copies
//   yv_cast_colsum       f32 gradient stream -> bf16 GEMM operand + per-column partial sums (bias grads)
//   yv_reduce_rows       deterministic second stage of every partial-sum reduction
//   yv_layernorm_bwd     dx += LN'(dy), partial dgamma / dbeta (wave per row, f32 statistics recomputed)
//   yv_token_reduce      d pos_embed / d cls_token = sum over crops
//   yv_head_bwd          Network_Wrapper.fc backward (ReLU-Linear-ReLU-Linear on (R,1000))
// Reductions are two-stage (partials + ordered sum): results are bitwise reproducible run to run.
#include "yv_common.h"

namespace {

// ------------------------------------------------------------------------------------------ transpose
template <typename TIN>
__global__ __launch_bounds__(256) void transpose_kernel(const TIN* __restrict__ in, int rows, int cols, long long ld_in,
                                                        uint16_t* __restrict__ out_t, long long ld_out,
                                                        uint16_t* __restrict__ out_n, long long ld_n) {
    // tile: 64 rows x 64 cols of `in`; out_t[c][r] = in[r][c]; optional straight bf16 copy out_n[r][c]
    __shared__ uint16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        uint16_t v = 0;
        if (r0 + r < rows && c0 + c < cols) {
            if constexpr (sizeof(TIN) == 4) v = f32_to_bf16(((const float*)in)[(long long)(r0 + r) * ld_in + c0 + c]);
            else v = ((const uint16_t*)in)[(long long)(r0 + r) * ld_in + c0 + c];
            if (out_n) out_n[(long long)(r0 + r) * ld_n + c0 + c] = v;
        }
        tile[r][c] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;                 // consecutive threads -> consecutive r: coalesced rows of out_t
        if (c0 + c < cols) out_t[(long long)(c0 + c) * ld_out + r0 + r] = tile[r][c];   // rows >= `rows` get zeros
    }
}

// ------------------------------------------------------------------------------ f32 stream -> bf16 + column sums
__global__ __launch_bounds__(256) void cast_colsum_kernel(const float* __restrict__ x, int rows, int cols,
                                                          uint16_t* __restrict__ y, float* __restrict__ partial,
                                                          int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    if (c >= cols) return;
    float s = 0.f;
    for (int r = r0; r < r1; ++r) {
        const float v = x[(long long)r * cols + c];
        s += v;
        if (y) y[(long long)r * cols + c] = f32_to_bf16(v);
    }
    if (partial) partial[(long long)blockIdx.y * cols + c] = s;
}

__global__ __launch_bounds__(256) void colsum_bf16_kernel(const uint16_t* __restrict__ x, int rows, int cols,
                                                          long long ld, float* __restrict__ partial, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    if (c >= cols) return;
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += bf16_to_f32(x[(long long)r * ld + c]);
    partial[(long long)blockIdx.y * cols + c] = s;
}

// vector forms of the two column-sum kernels (cols % 8 == 0 resp. % 4 == 0, 16-byte aligned rows): a workgroup covers
// 256 columns x rows_per_block rows as (column groups) x (row lanes); every thread issues all of its row loads before the
// first add (a read pass lives on its loads in flight), row lanes are combined through LDS in lane order: deterministic.
__global__ __launch_bounds__(256) void colsum_bf16_vec_kernel(const uint16_t* __restrict__ x, int rows, int cols, long long ld,
                                                              float* __restrict__ partial, int rows_per_block) {
    __shared__ float red[8][256];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;             // 32 groups of 8 columns x 8 row lanes
    const int c = blockIdx.x * 256 + cg * 8;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < cols) {
        for (int r = r0 + rl; r < r1; r += 32) {
            uint4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + u * 8;
                q[u] = rr < r1 ? *(const uint4*)(x + (long long)rr * ld + c) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t w[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s[2 * i] += bf16_to_f32((uint16_t)(w[i] & 0xffff));
                    s[2 * i + 1] += bf16_to_f32((uint16_t)(w[i] >> 16));
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[rl][cg * 8 + i] = s[i];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < cols) {
        float t = red[0][threadIdx.x];
#pragma unroll
        for (int l = 1; l < 8; ++l) t += red[l][threadIdx.x];
        partial[(long long)blockIdx.y * cols + cc] = t;
    }
}

__global__ __launch_bounds__(256) void cast_colsum_vec_kernel(const float* __restrict__ x, int rows, int cols,
                                                              uint16_t* __restrict__ y, float* __restrict__ partial,
                                                              int rows_per_block) {
    __shared__ float red[4][256];
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;             // 64 groups of 4 columns x 4 row lanes
    const int c = blockIdx.x * 256 + cg * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s[4] = {0, 0, 0, 0};
    if (c < cols) {
        for (int r = r0 + rl; r < r1; r += 16) {
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + u * 4;
                q[u] = rr < r1 ? *(const float4*)(x + (long long)rr * cols + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + u * 4;
                s[0] += q[u].x; s[1] += q[u].y; s[2] += q[u].z; s[3] += q[u].w;
                if (y && rr < r1)
                    *(uint2*)(y + (long long)rr * cols + c) = make_uint2(pack_bf16x2(q[u].x, q[u].y), pack_bf16x2(q[u].z, q[u].w));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[rl][cg * 4 + i] = s[i];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (partial && cc < cols) {
        float t = red[0][threadIdx.x];
#pragma unroll
        for (int l = 1; l < 4; ++l) t += red[l][threadIdx.x];
        partial[(long long)blockIdx.y * cols + cc] = t;
    }
}

__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ partial, int parts, int cols,
                                                          float* __restrict__ out, int accumulate,
                                                          float* __restrict__ out2 = nullptr, int split = 0) {
    // 32 columns (8 float4 groups) x 32 row-groups per workgroup; each group sums every 32nd partial row (fixed order),
    // then the 32 group sums are added in group order: deterministic.  cols is a multiple of 4 (host check).
    __shared__ float4 red[32][8];
    const int cg = threadIdx.x & 7, grp = threadIdx.x >> 3;
    const int c = blockIdx.x * 32 + cg * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cols) {
        int p = grp;
        for (; p + 96 < parts; p += 128) {                       // four partial rows per trip, loads first; order unchanged
            const float4 v0 = *(const float4*)(partial + (long long)p * cols + c);
            const float4 v1 = *(const float4*)(partial + (long long)(p + 32) * cols + c);
            const float4 v2 = *(const float4*)(partial + (long long)(p + 64) * cols + c);
            const float4 v3 = *(const float4*)(partial + (long long)(p + 96) * cols + c);
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; p < parts; p += 32) {
            const float4 v = *(const float4*)(partial + (long long)p * cols + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[grp][cg] = s;
    __syncthreads();
    if (grp == 0 && c < cols) {
        float4 t = red[0][cg];
#pragma unroll
        for (int q = 1; q < 32; ++q) { t.x += red[q][cg].x; t.y += red[q][cg].y; t.z += red[q][cg].z; t.w += red[q][cg].w; }
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = c + i;
            float* o = (out2 && ci >= split) ? out2 + (ci - split) : out + ci;      // [0, split) -> out, [split, cols) -> out2
            *o = accumulate ? *o + tv[i] : tv[i];
        }
    }
}

// columns not a multiple of 4 (small heads): one column per thread, 8 row groups
__global__ __launch_bounds__(256) void reduce_rows_scalar_kernel(const float* __restrict__ partial, int parts, int cols,
                                                                 float* __restrict__ out, int accumulate) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < cols)
        for (int p = grp; p < parts; p += 8) s += partial[(long long)p * cols + c];
    red[grp][cl] = s;
    __syncthreads();
    if (grp == 0 && c < cols) {
        float t = red[0][cl];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += red[q][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

void launch_reduce_rows(hipStream_t st, const float* partial, int parts, int cols, float* out, int accumulate,
                        float* out2 = nullptr, int split = 0) {
    if ((cols & 3) == 0 && (((uintptr_t)partial) & 15) == 0)
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((cols + 31) / 32), dim3(256), 0, st, partial, parts, cols, out, accumulate, out2,
                           split);
    else
        hipLaunchKernelGGL(reduce_rows_scalar_kernel, dim3((cols + 31) / 32), dim3(256), 0, st, partial, parts, cols, out,
                           accumulate);
}

// ---------------------------------------------------------------------------------------- LayerNorm backward
constexpr int LNB_MAXC = 4;
constexpr int LNB_ROWS = 8;           // rows per workgroup (2 per wave): 6304 tokens -> 788 workgroups (32 rows left 59 CUs idle)

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, long long ldx,
                                                            const float* __restrict__ gamma,
                                                            const uint16_t* __restrict__ dy, long long lddy, int rows, int D,
                                                            float eps, float* __restrict__ dx, long long lddx,
                                                            float* __restrict__ partial /* [blocks][2][D] */) {
    __shared__ float red[4][2][LNB_MAXC * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = D >> 2;
    float4 dg[LNB_MAXC], db[LNB_MAXC], gm[LNB_MAXC];
#pragma unroll
    for (int i = 0; i < LNB_MAXC; ++i) {
        dg[i] = db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = lane + 64 * i;
        gm[i] = c < nch ? ((const float4*)gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int rr = 0; rr < LNB_ROWS / 4; ++rr) {
        const int row = blockIdx.x * LNB_ROWS + wave * (LNB_ROWS / 4) + rr;
        if (row >= rows) break;
        const float* xr = x + row * ldx;
        const uint16_t* dr = dy + row * lddy;
        float4 v[LNB_MAXC], g[LNB_MAXC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_MAXC; ++i) {
            const int c = lane + 64 * i;
            v[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nch) {
                v[i] = ((const float4*)xr)[c];
                const uint2 d2 = ((const uint2*)dr)[c];
                g[i] = make_float4(bf16_to_f32((uint16_t)(d2.x & 0xffff)), bf16_to_f32((uint16_t)(d2.x >> 16)),
                                   bf16_to_f32((uint16_t)(d2.y & 0xffff)), bf16_to_f32((uint16_t)(d2.y >> 16)));
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
        float s1 = 0.f, s2 = 0.f;                      // sum(g*gamma), sum(g*gamma*xhat)
#pragma unroll
        for (int i = 0; i < LNB_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;        // xhat
                dg[i].x += g[i].x * v[i].x; dg[i].y += g[i].y * v[i].y; dg[i].z += g[i].z * v[i].z; dg[i].w += g[i].w * v[i].w;
                db[i].x += g[i].x; db[i].y += g[i].y; db[i].z += g[i].z; db[i].w += g[i].w;
                g[i].x *= gm[i].x; g[i].y *= gm[i].y; g[i].z *= gm[i].z; g[i].w *= gm[i].w;
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
            }
        }
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
        float* dxr = dx + row * lddx;
#pragma unroll
        for (int i = 0; i < LNB_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                float4 o = ((float4*)dxr)[c];
                o.x += rstd * (g[i].x - m1 - v[i].x * m2); o.y += rstd * (g[i].y - m1 - v[i].y * m2);
                o.z += rstd * (g[i].z - m1 - v[i].z * m2); o.w += rstd * (g[i].w - m1 - v[i].w * m2);
                ((float4*)dxr)[c] = o;
            }
        }
    }
    // cross-wave reduction of the per-lane partial dgamma / dbeta, one partial row per workgroup
#pragma unroll
    for (int i = 0; i < LNB_MAXC; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < LNB_MAXC * 256) {
            red[wave][0][c] = dg[i].x; red[wave][0][c + 1] = dg[i].y; red[wave][0][c + 2] = dg[i].z; red[wave][0][c + 3] = dg[i].w;
            red[wave][1][c] = db[i].x; red[wave][1][c + 1] = db[i].y; red[wave][1][c + 2] = db[i].z; red[wave][1][c + 3] = db[i].w;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        partial[((long long)blockIdx.x * 2) * D + c] = (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]);
        partial[((long long)blockIdx.x * 2 + 1) * D + c] = (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]);
    }
}

// --------------------------------------------------------------------------------- pos_embed / cls_token grads
__global__ __launch_bounds__(256) void token_reduce_kernel(const float* __restrict__ dx, int R, int N, int D,
                                                           float* __restrict__ out /* (N,D) */) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * D) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += dx[(long long)r * N * D + i];
    out[i] = s;
}

// ------------------------------------------------------------------------------------ Network_Wrapper.fc backward
constexpr int HB_FEAT = 1000, HB_HID = 128, HB_MAX_NC = 32;

// per crop: recompute h, dh = (W2^T dlogits) * (h > 0); keep relu(f), h, dh for the weight-gradient kernel
__global__ __launch_bounds__(256) void head_bwd_a_kernel(const float* __restrict__ feats, int ldf,
                                                         const float* __restrict__ w1t, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ dlogits,
                                                         int nc, float* __restrict__ hbuf, float* __restrict__ dhbuf) {
    __shared__ float f[HB_FEAT];
    __shared__ float part[2][HB_HID];
    const int r = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < HB_FEAT; i += 256) f[i] = fmaxf(feats[(size_t)r * ldf + i], 0.f);
    __syncthreads();
    const int u = tid & (HB_HID - 1), half = tid >> 7, k0 = half * (HB_FEAT / 2);
    float s = 0.f;
    for (int k = k0; k < k0 + HB_FEAT / 2; ++k) s = fmaf(f[k], w1t[(size_t)k * HB_HID + u], s);
    part[half][u] = s;
    __syncthreads();
    if (tid < HB_HID) {
        const float h = fmaxf(part[0][tid] + part[1][tid] + b1[tid], 0.f);
        float d = 0.f;
        for (int c = 0; c < nc; ++c) d = fmaf(dlogits[(size_t)r * nc + c], w2[(size_t)c * HB_HID + tid], d);
        hbuf[(size_t)r * HB_HID + tid] = h;
        dhbuf[(size_t)r * HB_HID + tid] = h > 0.f ? d : 0.f;
    }
}

// weight / bias gradients (sums over crops in fixed order) and d feats
__global__ __launch_bounds__(256) void head_bwd_b_kernel(const float* __restrict__ feats, int ldf,
                                                         const float* __restrict__ w1t, const float* __restrict__ hbuf,
                                                         const float* __restrict__ dhbuf, const float* __restrict__ dlogits,
                                                         int R, int nc, float* __restrict__ dw1 /* (128,1000) */,
                                                         float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                                                         uint16_t* __restrict__ dfeats /* (R,ldd) bf16 */, int ldd) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n_w1 = (long long)HB_HID * HB_FEAT, n_df = (long long)R * ldd;
    if (i < n_w1) {                                         // dW1[u][k] = sum_r dh[r][u] * relu(f[r][k])
        const int u = (int)(i / HB_FEAT), k = (int)(i - (long long)u * HB_FEAT);
        float s = 0.f;
        for (int r = 0; r < R; ++r) s = fmaf(dhbuf[(size_t)r * HB_HID + u], fmaxf(feats[(size_t)r * ldf + k], 0.f), s);
        dw1[i] = s;
    } else if (i < n_w1 + n_df) {                           // dfeat[r][k] = (sum_u dh[r][u] W1[u][k]) * (f > 0)
        const long long j = i - n_w1;
        const int r = (int)(j / ldd), k = (int)(j - (long long)r * ldd);
        float s = 0.f;
        if (k < HB_FEAT && feats[(size_t)r * ldf + k] > 0.f)
            for (int u = 0; u < HB_HID; ++u) s = fmaf(dhbuf[(size_t)r * HB_HID + u], w1t[(size_t)k * HB_HID + u], s);
        dfeats[j] = f32_to_bf16(s);
    } else {
        const long long j = i - n_w1 - n_df;
        if (j < HB_HID) {                                   // db1
            float s = 0.f;
            for (int r = 0; r < R; ++r) s += dhbuf[(size_t)r * HB_HID + j];
            db1[j] = s;
        } else if (j < HB_HID + (long long)nc * HB_HID) {   // dW2[c][u] = sum_r dlogits[r][c] * h[r][u]
            const int c = (int)((j - HB_HID) / HB_HID), u = (int)((j - HB_HID) % HB_HID);
            float s = 0.f;
            for (int r = 0; r < R; ++r) s = fmaf(dlogits[(size_t)r * nc + c], hbuf[(size_t)r * HB_HID + u], s);
            dw2[(size_t)c * HB_HID + u] = s;
        } else if (j < HB_HID + (long long)nc * HB_HID + nc) {
            const int c = (int)(j - HB_HID - (long long)nc * HB_HID);
            float s = 0.f;
            for (int r = 0; r < R; ++r) s += dlogits[(size_t)r * nc + c];
            db2[c] = s;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int yv_transpose_bf16(const void* in, int rows, int cols, long long ld_in, void* out_t, long long ld_out,
                                 void* stream) {
    if (!in || !out_t || rows <= 0 || cols <= 0 || ld_out < rows) return YV_ERR_ARG;
    dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    if ((long long)grid.y * 64 > ld_out) return YV_ERR_ARG;          // the zero padding must fit the row
    hipLaunchKernelGGL(transpose_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, rows,
                       cols, ld_in, (uint16_t*)out_t, ld_out, (uint16_t*)nullptr, 0LL);
    return yv_launch_status();
}

extern "C" int yv_cast_weights(const float* w, int N, int K, void* w_bf16, void* wt_bf16, long long ld_t, void* stream) {
    if (!w || !w_bf16 || !wt_bf16 || N <= 0 || K <= 0 || ld_t < N) return YV_ERR_ARG;
    dim3 grid((K + 63) / 64, (N + 63) / 64);
    if ((long long)grid.y * 64 > ld_t) return YV_ERR_ARG;
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, w, N, K, (long long)K,
                       (uint16_t*)wt_bf16, ld_t, (uint16_t*)w_bf16, (long long)K);
    return yv_launch_status();
}

constexpr int CS_ROWS = 32;         // rows per workgroup of the column-sum kernels (6304 rows -> 197 x cols/256 workgroups)

extern "C" size_t yv_colsum_ws_floats(int rows, int cols) { return (size_t)((rows + CS_ROWS - 1) / CS_ROWS) * (size_t)cols; }

extern "C" int yv_cast_colsum(const float* x, int rows, int cols, void* y_bf16, float* colsum, int accumulate,
                              float* ws, void* stream) {
    if (!x || rows <= 0 || cols <= 0 || (colsum && !ws)) return YV_ERR_ARG;
    const int parts = (rows + CS_ROWS - 1) / CS_ROWS;
    hipStream_t st = (hipStream_t)stream;
    if (!(cols & 3) && !((uintptr_t)x & 15) && !((uintptr_t)y_bf16 & 7))
        hipLaunchKernelGGL(cast_colsum_vec_kernel, dim3((cols + 255) / 256, parts), dim3(256), 0, st, x, rows, cols,
                           (uint16_t*)y_bf16, colsum ? ws : nullptr, CS_ROWS);
    else
        hipLaunchKernelGGL(cast_colsum_kernel, dim3((cols + 255) / 256, parts), dim3(256), 0, st, x, rows, cols,
                           (uint16_t*)y_bf16, colsum ? ws : nullptr, CS_ROWS);
    if (colsum)
        launch_reduce_rows(st, ws, parts, cols, colsum, accumulate);
    return yv_launch_status();
}

extern "C" int yv_colsum_bf16(const void* x, int rows, int cols, long long ld, float* colsum, int accumulate, float* ws,
                              void* stream) {
    if (!x || !colsum || !ws || rows <= 0 || cols <= 0) return YV_ERR_ARG;
    const int parts = (rows + CS_ROWS - 1) / CS_ROWS;
    hipStream_t st = (hipStream_t)stream;
    if (!(cols & 7) && !(ld & 7) && !((uintptr_t)x & 15))
        hipLaunchKernelGGL(colsum_bf16_vec_kernel, dim3((cols + 255) / 256, parts), dim3(256), 0, st, (const uint16_t*)x, rows,
                           cols, ld, ws, CS_ROWS);
    else
        hipLaunchKernelGGL(colsum_bf16_kernel, dim3((cols + 255) / 256, parts), dim3(256), 0, st, (const uint16_t*)x, rows,
                           cols, ld, ws, CS_ROWS);
    launch_reduce_rows(st, ws, parts, cols, colsum, accumulate);
    return yv_launch_status();
}

extern "C" size_t yv_layernorm_bwd_ws_floats(int rows, int D) {
    return (size_t)((rows + LNB_ROWS - 1) / LNB_ROWS) * 2 * (size_t)D + 2 * (size_t)D;
}

extern "C" int yv_layernorm_bwd(const float* x, long long ldx, const float* gamma, const void* dy, long long lddy,
                                int rows, int D, float eps, float* dx, long long lddx, float* dgamma, float* dbeta,
                                float* ws, void* stream) {
    if (!x || !gamma || !dy || !dx || !dgamma || !dbeta || !ws || rows <= 0 || D <= 0) return YV_ERR_ARG;
    if ((D & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3)) return YV_ERR_ARG;
    if (D > LNB_MAXC * 256) return YV_ERR_LIMIT;
    const int blocks = (rows + LNB_ROWS - 1) / LNB_ROWS;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, st, x, ldx, gamma, (const uint16_t*)dy, lddy,
                       rows, D, eps, dx, lddx, ws);
    // partial layout [block][2][D]: reduce with stride 2*D -> view as `blocks` rows of 2*D columns
    launch_reduce_rows(st, ws, blocks, 2 * D, dgamma, 0, dbeta, D);
    return yv_launch_status();
}

extern "C" int yv_token_reduce(const float* dx, int R, int N, int D, float* out, void* stream) {
    if (!dx || !out || R <= 0 || N <= 0 || D <= 0) return YV_ERR_ARG;
    const long long n = (long long)N * D;
    hipLaunchKernelGGL(token_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dx, R,
                       N, D, out);
    return yv_launch_status();
}

extern "C" int yv_head_bwd(const float* feats, int ldf, const float* w1t, const float* b1, const float* w2,
                           const float* dlogits, int R, int nc, float* dw1, float* db1, float* dw2, float* db2,
                           void* dfeats_bf16, int ldd, float* ws, void* stream) {
    if (!feats || !w1t || !b1 || !w2 || !dlogits || !dw1 || !db1 || !dw2 || !db2 || !dfeats_bf16 || !ws) return YV_ERR_ARG;
    if (R <= 0 || nc <= 0 || nc > HB_MAX_NC || ldf < HB_FEAT || ldd < HB_FEAT) return YV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* hbuf = ws;
    float* dhbuf = ws + (size_t)R * HB_HID;
    hipLaunchKernelGGL(head_bwd_a_kernel, dim3(R), dim3(256), 0, st, feats, ldf, w1t, b1, w2, dlogits, nc, hbuf, dhbuf);
    const long long items = (long long)HB_HID * HB_FEAT + (long long)R * ldd + HB_HID + (long long)nc * HB_HID + nc;
    hipLaunchKernelGGL(head_bwd_b_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, feats, ldf, w1t, hbuf,
                       dhbuf, dlogits, R, nc, dw1, db1, dw2, db2, (uint16_t*)dfeats_bf16, ldd);
    return yv_launch_status();
}
