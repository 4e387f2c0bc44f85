// Detector training kernels around the implicit-GEMM convs (the ultralytics trainer
// behind utils/trainYolo.py:13-35; layer structure docs/YOLO_TensorRT_Technical.md:160-212 with the BatchNorm
// un-folded, as ultralytics trains it: Conv = conv(no bias) -> BatchNorm2d(eps 1e-3, momentum 0.03) -> SiLU).
// Activations are NHWC bf16 "views": (rows = B*H*W, channels) with a row stride, so C2f / SPPF / neck concats are
// channel slices of one buffer, forward and backward.
//
//   yv_blob_nhwc8        u8 RGB -> bf16 /255, channels padded 3 -> 8 (the stem conv then is an ordinary conv)
//   yv_bn_stats          per-channel batch mean / rstd (+ running statistics), deterministic two-stage
//   yv_bn_act_fwd        a = SiLU(gamma * (z - mean) * rstd + beta) (+ shortcut)
//   yv_bn_act_bwd        dz, dgamma, dbeta from da (batch statistics: the mean / variance terms are included)
//   yv_view_op           copy / add / nearest-2x up / its adjoint / zero-insertion 2x (stride-2 dgrad) / zero fill
//   yv_maxpool5_bwd      adjoint of the 5x5/s1/p2 max-pool of SPPF (first maximum in scan order wins, as torch)
//   yv_im2col3           explicit (rows, 9*C) patch matrix for the 3x3 weight gradients (dW = dz^T . im2col(x))
//   yv_conv_weight_dgrad (Cout, taps, Cin) -> (Cin, flipped taps, Cout): the data gradient is a conv with this weight
// All HBM-bound element-wise / reduction kernels: 16-byte accesses over 8-channel groups, f32 math.
#include "yv_common.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf16_to_f32((uint16_t)(w[i] & 0xffff));
        f[2 * i + 1] = bf16_to_f32((uint16_t)(w[i] >> 16));
    }
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

__global__ __launch_bounds__(256) void blob_kernel(const uint8_t* __restrict__ img, long long pixels,
                                                   uint16_t* __restrict__ out) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= pixels) return;
    const uint8_t* s = img + p * 3;
    float f[8] = {s[0] * (1.0f / 255.0f), s[1] * (1.0f / 255.0f), s[2] * (1.0f / 255.0f), 0.f, 0.f, 0.f, 0.f, 0.f};
    *(uint4*)(out + p * 8) = pack8(f);
}

// ---- two-stage per-channel reductions over the rows of a (T, C) view --------------------------------------
// MODE 0: (sum z, sum z^2)      MODE 1: (sum g, sum g*xhat) with g = da * act'(u), u = gamma*xhat + beta
// Workgroup = one chunk of rows; thread = (row lane, 8-channel group); row lanes are combined in lane order through
// LDS and chunks in chunk order by the finalize kernels: bitwise reproducible.
struct RedArgs {
    const uint16_t* z; long long ldz;
    const uint16_t* da; long long ldda;
    long long T; int C; int rows_per_chunk;
    const float *mean, *rstd, *gamma, *beta;
    int act;
    float* partial;            // (chunks, 2, C)
};

template <int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(RedArgs a) {
    extern __shared__ float red[];                       // (row lanes, 2, C)
    const int cg = a.C >> 3;
    const int rp = 256 / cg;                             // row lanes (cg <= 128 checked on the host)
    const int lane_r = threadIdx.x / cg, g = threadIdx.x - lane_r * cg;
    float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane_r < rp) {
        float mu[8], rs[8], ga[8], be[8];
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                mu[i] = a.mean[g * 8 + i]; rs[i] = a.rstd[g * 8 + i]; ga[i] = a.gamma[g * 8 + i]; be[i] = a.beta[g * 8 + i];
            }
        }
        const long long r0 = (long long)blockIdx.x * a.rows_per_chunk;
        long long r1 = r0 + a.rows_per_chunk;
        r1 = r1 < a.T ? r1 : a.T;
        auto accumulate = [&](const uint4& zq, const uint4& dq) {
            float z[8];
            unpack8(zq, z);
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { s0[i] += z[i]; s1[i] += z[i] * z[i]; }
            } else {
                float d[8];
                unpack8(dq, d);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xh = (z[i] - mu[i]) * rs[i];
                    float gr = d[i];
                    if (a.act) {
                        const float u = ga[i] * xh + be[i];
                        const float sg = sigmoid_f(u);
                        gr *= sg * (1.0f + u * (1.0f - sg));
                    }
                    s0[i] += gr; s1[i] += gr * xh;
                }
            }
        };
        // four rows per trip: all eight 16-byte loads are issued before the first one is consumed (a pure read pass
        // is bound by the loads in flight per wave); rows are still accumulated in increasing order
        long long r = r0 + lane_r;
        for (; r + 3LL * rp < r1; r += 4LL * rp) {
            uint4 zq[4], dq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                zq[u] = *(const uint4*)(a.z + (r + (long long)u * rp) * a.ldz + g * 8);
                if (MODE == 1) dq[u] = *(const uint4*)(a.da + (r + (long long)u * rp) * a.ldda + g * 8);
                else dq[u] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) accumulate(zq[u], dq[u]);
        }
        for (; r < r1; r += rp) {
            const uint4 zq = *(const uint4*)(a.z + r * a.ldz + g * 8);
            const uint4 dq = MODE == 1 ? *(const uint4*)(a.da + r * a.ldda + g * 8) : make_uint4(0, 0, 0, 0);
            accumulate(zq, dq);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            red[(lane_r * 2 + 0) * a.C + g * 8 + i] = s0[i];
            red[(lane_r * 2 + 1) * a.C + g * 8 + i] = s1[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * a.C; c += 256) {
        const int which = c / a.C, ch = c - which * a.C;
        float t = 0.f;
        for (int l = 0; l < rp; ++l) t += red[(l * 2 + which) * a.C + ch];
        a.partial[((long long)blockIdx.x * 2 + which) * a.C + ch] = t;
    }
}

// Second stage: one workgroup per 8 channels; 32 chunk lanes sum every 32nd chunk (double), lane order fixes the
// final summation order (bitwise reproducible).
__device__ __forceinline__ void chunk_sums(const float* __restrict__ partial, int chunks, int C, int c, double& s, double& ss,
                                           double* sh) {
    const int lane = threadIdx.x >> 3;                       // 0..31
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int k = lane; k < chunks; k += 32) { a += partial[((long long)k * 2) * C + c]; b += partial[((long long)k * 2 + 1) * C + c]; }
    sh[(lane * 8 + (threadIdx.x & 7)) * 2] = a;
    sh[(lane * 8 + (threadIdx.x & 7)) * 2 + 1] = b;
    __syncthreads();
    s = 0.0; ss = 0.0;
    if (lane == 0)
        for (int l = 0; l < 32; ++l) { s += sh[(l * 8 + threadIdx.x) * 2]; ss += sh[(l * 8 + threadIdx.x) * 2 + 1]; }
}

__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* __restrict__ partial, int chunks, int C, long long T,
                                                             float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                                             float* __restrict__ run_mean, float* __restrict__ run_var,
                                                             float momentum) {
    __shared__ double sh[512];
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    double s, ss;
    chunk_sums(partial, chunks, C, c, s, ss, sh);
    if (threadIdx.x >= 8 || c >= C) return;
    const double m = s / (double)T;
    double var = ss / (double)T - m * m;
    var = var > 0.0 ? var : 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {                                                     // torch: unbiased variance in the running estimate
        const double unb = T > 1 ? var * (double)T / (double)(T - 1) : var;
        run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * m);
        run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ partial, int chunks, int C, long long T,
                                                           int batch_stats, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ coef /* (2, C): mean(g), mean(g*xhat) */) {
    __shared__ double sh[512];
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    double s, sx;
    chunk_sums(partial, chunks, C, c, s, sx, sh);
    if (threadIdx.x >= 8 || c >= C) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)sx;
    coef[c] = batch_stats ? (float)(s / (double)T) : 0.f;
    coef[C + c] = batch_stats ? (float)(sx / (double)T) : 0.f;
}

struct ActArgs {
    const uint16_t* z; long long ldz;
    const uint16_t* da; long long ldda;            // backward only
    const uint16_t* res; long long ldres;          // forward shortcut (optional)
    uint16_t* out; long long ldo;                  // forward: a; backward: dz
    long long T; int C;
    const float *mean, *rstd, *gamma, *beta, *coef;
    int act;
};

// thread = (row lane, 8-channel group): the per-channel constants are loaded once, then the thread walks rows
// (the first version re-read 40 per-channel scalars per element group and ran at 0.7 TB/s)
constexpr int ACT_ROWS_PER_LANE = 16;

template <bool BWD>
__global__ __launch_bounds__(256) void bn_act_kernel(ActArgs a) {
    const int cg = a.C >> 3;
    const int rp = 256 / cg;
    const int lane_r = threadIdx.x / cg, g = threadIdx.x - lane_r * cg;
    if (lane_r >= rp) return;
    float sc[8], k1[8], k2[8], gam[8], bet[8], mu[8], rs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = g * 8 + i;
        mu[i] = a.mean[c]; rs[i] = a.rstd[c]; gam[i] = a.gamma[c]; bet[i] = a.beta[c];
        sc[i] = gam[i] * rs[i];
        if (BWD) { k1[i] = a.coef[c]; k2[i] = a.coef[a.C + c]; }
    }
    const long long r0 = (long long)blockIdx.x * (rp * ACT_ROWS_PER_LANE) + lane_r;
#pragma unroll 4
    for (int k = 0; k < ACT_ROWS_PER_LANE; ++k) {
        const long long r = r0 + (long long)k * rp;
        if (r >= a.T) break;
        float z[8], o[8];
        unpack8(*(const uint4*)(a.z + r * a.ldz + g * 8), z);
        if (!BWD) {
            float rs8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (a.res) unpack8(*(const uint4*)(a.res + r * a.ldres + g * 8), rs8);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float u = gam[i] * ((z[i] - mu[i]) * rs[i]) + bet[i];
                float v = a.act ? u * sigmoid_f(u) : u;
                if (a.res) v = bf16_to_f32(f32_to_bf16(v)) + rs8[i];     // the shortcut adds the rounded activation
                o[i] = v;
            }
        } else {
            float d[8];
            unpack8(*(const uint4*)(a.da + r * a.ldda + g * 8), d);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (z[i] - mu[i]) * rs[i];
                float gr = d[i];
                if (a.act) {
                    const float u = gam[i] * xh + bet[i];
                    const float sg = sigmoid_f(u);
                    gr *= sg * (1.0f + u * (1.0f - sg));
                }
                o[i] = sc[i] * (gr - k1[i] - xh * k2[i]);
            }
        }
        *(uint4*)(a.out + r * a.ldo + g * 8) = pack8(o);
    }
}

// ---- view ops --------------------------------------------------------------------------------------------
// mode 0 copy, 1 add (dst += src), 2 nearest-2x up (dst (B,2H,2W) <- src (B,H,W)), 3 adjoint of 2 accumulated
// (dst (B,H,W) += sum of the 2x2 block of src (B,2H,2W)), 4 zero insertion (dst (B,2H,2W): [2y][2x] = src[y][x], else 0),
// 5 zero fill of dst (B,H,W), 6 zero-ring padding (dst (B,H+2,W+2): interior = src (B,H,W), ring = 0)
template <typename IDX>                                   // uint32_t whenever the element count fits: 32-bit divisions
__global__ __launch_bounds__(256) void view_op_kernel(int mode, const uint16_t* __restrict__ src, long long lds_,
                                                      uint16_t* __restrict__ dst, long long ldd, int B, int H, int W, int C) {
    const IDX cg = (IDX)(C >> 3);
    const int DH = (mode == 2 || mode == 4) ? 2 * H : (mode == 6 ? H + 2 : H), DW = (mode == 2 || mode == 4) ? 2 * W : (mode == 6 ? W + 2 : W);
    const IDX idx = (IDX)blockIdx.x * 256 + threadIdx.x;
    const IDX total = (IDX)B * DH * DW * cg;
    if (idx >= total) return;
    const int g = (int)(idx % cg);
    const IDX pi = idx / cg;                           // destination pixel
    const int x = (int)(pi % (IDX)DW);
    const IDX q = pi / (IDX)DW;
    const int y = (int)(q % (IDX)DH), b = (int)(q / (IDX)DH);
    const long long p = (long long)pi;
    uint16_t* d = dst + p * ldd + g * 8;
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (mode == 0) {
        *(uint4*)d = *(const uint4*)(src + p * lds_ + g * 8);
        return;
    } else if (mode == 1) {
        float s[8];
        unpack8(*(const uint4*)d, o);
        unpack8(*(const uint4*)(src + p * lds_ + g * 8), s);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += s[i];
    } else if (mode == 2) {
        *(uint4*)d = *(const uint4*)(src + (((long long)b * H + (y >> 1)) * W + (x >> 1)) * lds_ + g * 8);
        return;
    } else if (mode == 3) {
        unpack8(*(const uint4*)d, o);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float s[8];
                unpack8(*(const uint4*)(src + (((long long)b * 2 * H + 2 * y + dy) * (2 * W) + 2 * x + dx) * lds_ + g * 8), s);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] += s[i];
            }
    } else if (mode == 6) {
        if (y >= 1 && y <= H && x >= 1 && x <= W) {
            *(uint4*)d = *(const uint4*)(src + (((long long)b * H + (y - 1)) * W + (x - 1)) * lds_ + g * 8);
            return;
        }
    } else if (mode == 4) {
        if (!(y & 1) && !(x & 1)) {
            *(uint4*)d = *(const uint4*)(src + (((long long)b * H + (y >> 1)) * W + (x >> 1)) * lds_ + g * 8);
            return;
        }
    }
    *(uint4*)d = pack8(o);
}

// Two passes: (1) per window (= output position) the offset 0..24 of its first maximum in scan order, one byte per
// channel; (2) per input position, the 25 windows that contain it are looked up in that map (the first version
// recomputed every window's argmax per input position: 625 loads per element, 290 us per call).
__global__ __launch_bounds__(256) void maxpool5_argmax_kernel(const uint16_t* __restrict__ x, long long ldx, int B, int H, int W,
                                                              int C, uint8_t* __restrict__ amap /* (B*H*W, C) */) {
    const int cg = C >> 3;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * H * W * cg) return;
    const int g = (int)(idx % cg);
    const long long p = idx / cg;
    const int wx = (int)(p % W);
    const long long q = p / W;
    const int wy = (int)(q % H), b = (int)(q / H);
    const uint16_t* xb = x + (long long)b * H * W * ldx + g * 8;
    float best[8];
    int bo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; bo[i] = 12; }
    for (int dy = 0; dy < 5; ++dy) {
        const int yy = wy + dy - 2;
        if (yy < 0 || yy >= H) continue;
        for (int dx = 0; dx < 5; ++dx) {
            const int xx = wx + dx - 2;
            if (xx < 0 || xx >= W) continue;
            float v[8];
            unpack8(*(const uint4*)(xb + ((long long)yy * W + xx) * ldx), v);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (v[i] > best[i]) { best[i] = v[i]; bo[i] = dy * 5 + dx; }
        }
    }
    uint2 o;
    o.x = (uint32_t)bo[0] | ((uint32_t)bo[1] << 8) | ((uint32_t)bo[2] << 16) | ((uint32_t)bo[3] << 24);
    o.y = (uint32_t)bo[4] | ((uint32_t)bo[5] << 8) | ((uint32_t)bo[6] << 16) | ((uint32_t)bo[7] << 24);
    *(uint2*)(amap + p * C + g * 8) = o;
}

__global__ __launch_bounds__(256) void maxpool5_bwd_kernel(const uint8_t* __restrict__ amap, const uint16_t* __restrict__ dout,
                                                           long long lddo, uint16_t* __restrict__ din, long long lddi, int B, int H,
                                                           int W, int C) {
    const int cg = C >> 3;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * H * W * cg) return;
    const int g = (int)(idx % cg);
    const long long p = idx / cg;
    const int px = (int)(p % W);
    const long long q = p / W;
    const int py = (int)(q % H), b = (int)(q / H);
    float acc[8];
    unpack8(*(const uint4*)(din + p * lddi + g * 8), acc);
    for (int wy = py - 2; wy <= py + 2; ++wy) {
        if (wy < 0 || wy >= H) continue;
        for (int wx = px - 2; wx <= px + 2; ++wx) {
            if (wx < 0 || wx >= W) continue;
            const long long wp = ((long long)b * H + wy) * W + wx;
            const uint2 m = *(const uint2*)(amap + wp * C + g * 8);
            const int want = (py - wy + 2) * 5 + (px - wx + 2);              // this position as an offset inside that window
            float d[8];
            unpack8(*(const uint4*)(dout + wp * lddo + g * 8), d);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int off = (int)(((i < 4 ? m.x : m.y) >> (8 * (i & 3))) & 0xff);
                if (off == want) acc[i] += d[i];
            }
        }
    }
    *(uint4*)(din + p * lddi + g * 8) = pack8(acc);
}

// IDX = uint32_t whenever the element count allows it: the four divisions per thread are what this write-bound copy
// spends its instructions on, and 64-bit division is several times the cost of 32-bit
template <typename IDX>
__global__ __launch_bounds__(256) void im2col3_kernel(const uint16_t* __restrict__ x, long long ldx, int B, int Hin, int Win,
                                                      int C, int stride, int Hout, int Wout, uint16_t* __restrict__ col) {
    const IDX cg = (IDX)(C >> 3);
    const IDX idx = (IDX)blockIdx.x * 256 + threadIdx.x;
    const IDX total = (IDX)B * Hout * Wout * 9 * cg;
    if (idx >= total) return;
    const int g = (int)(idx % cg);
    const IDX t = idx / cg;
    const int tap = (int)(t % 9);
    const IDX row = t / 9;
    const int ox = (int)(row % (IDX)Wout);
    const IDX q = row / (IDX)Wout;
    const int oy = (int)(q % (IDX)Hout), b = (int)(q / (IDX)Hout);
    const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) v = *(const uint4*)(x + (((long long)b * Hin + iy) * Win + ix) * ldx + g * 8);
    *(uint4*)(col + (long long)idx * 8) = v;                     // col[row][tap][g*8..] is exactly element idx * 8
}

__global__ __launch_bounds__(256) void weight_dgrad_kernel(const uint16_t* __restrict__ w, int Cout, int taps, int Cin,
                                                           uint16_t* __restrict__ wd) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)Cout * taps * Cin;
    if (idx >= total) return;
    const int co = (int)(idx % Cout);
    long long t = idx / Cout;
    const int tp = (int)(t % taps), ci = (int)(t / taps);
    wd[idx] = w[((long long)co * taps + (taps - 1 - tp)) * Cin + ci];      // 180-degree flip of the taps
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }
inline unsigned act_blocks(long long T, int C) {
    const long long per = (long long)(256 / (C >> 3)) * ACT_ROWS_PER_LANE;
    return (unsigned)((T + per - 1) / per);
}

int chunks_for(long long T, int* rows_per_chunk) {
    long long chunks = (T + 255) / 256;
    if (chunks > 2048) chunks = 2048;
    const long long rpc = (T + chunks - 1) / chunks;
    *rows_per_chunk = (int)rpc;
    return (int)((T + rpc - 1) / rpc);
}

}  // namespace

extern "C" int yv_blob_nhwc8(const void* images, long long pixels, void* out, void* stream) {
    if (!images || !out || pixels <= 0) return YV_ERR_ARG;
    hipLaunchKernelGGL(blob_kernel, dim3(blocks_for(pixels)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)images, pixels,
                       (uint16_t*)out);
    return yv_launch_status();
}

extern "C" size_t yv_bn_ws_floats(long long T, int C) {
    int rpc;
    const int chunks = chunks_for(T > 0 ? T : 1, &rpc);
    return (size_t)chunks * 2 * C + 2 * (size_t)C;
}

static bool bn_shape_ok(long long T, int C, long long ld) { return T > 0 && C >= 8 && C <= 1024 && !(C & 7) && !(ld & 7) && ld >= C; }

extern "C" int yv_bn_stats(const void* z, long long ldz, long long T, int C, float eps, float momentum, float* mean,
                           float* rstd, float* run_mean, float* run_var, float* ws, size_t ws_floats, void* stream) {
    if (!z || !mean || !rstd || !ws || !bn_shape_ok(T, C, ldz) || ((uintptr_t)z & 15)) return YV_ERR_ARG;
    if ((run_mean == nullptr) != (run_var == nullptr)) return YV_ERR_ARG;
    if (ws_floats < yv_bn_ws_floats(T, C)) return YV_ERR_WORKSPACE;
    RedArgs a = {};
    a.z = (const uint16_t*)z; a.ldz = ldz; a.T = T; a.C = C; a.partial = ws;
    const int chunks = chunks_for(T, &a.rows_per_chunk);
    const int rp = 256 / (C >> 3);
    hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3(chunks), dim3(256), (size_t)rp * 2 * C * sizeof(float), (hipStream_t)stream, a);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, ws, chunks, C, T, eps,
                       mean, rstd, run_mean, run_var, momentum);
    return yv_launch_status();
}

extern "C" int yv_bn_act_fwd(const void* z, long long ldz, long long T, int C, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, const void* res, long long ldres, void* out,
                             long long ldo, int act, void* stream) {
    if (!z || !out || !mean || !rstd || !gamma || !beta || !bn_shape_ok(T, C, ldz) || (ldo & 7) || ldo < C) return YV_ERR_ARG;
    if (res && ((ldres & 7) || ((uintptr_t)res & 15))) return YV_ERR_ARG;
    if (((uintptr_t)z | (uintptr_t)out) & 15) return YV_ERR_ARG;
    ActArgs a = {};
    a.z = (const uint16_t*)z; a.ldz = ldz; a.res = (const uint16_t*)res; a.ldres = ldres; a.out = (uint16_t*)out; a.ldo = ldo;
    a.T = T; a.C = C; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.act = act;
    hipLaunchKernelGGL(bn_act_kernel<false>, dim3(act_blocks(T, C)), dim3(256), 0, (hipStream_t)stream, a);
    return yv_launch_status();
}

extern "C" int yv_bn_act_bwd(const void* da, long long ldda, const void* z, long long ldz, long long T, int C,
                             const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                             int batch_stats, float* dgamma, float* dbeta, void* dz, long long lddz, float* ws,
                             size_t ws_floats, void* stream) {
    if (!da || !z || !dz || !mean || !rstd || !gamma || !beta || !dgamma || !dbeta || !ws) return YV_ERR_ARG;
    if (!bn_shape_ok(T, C, ldz) || (ldda & 7) || ldda < C || (lddz & 7) || lddz < C) return YV_ERR_ARG;
    if (((uintptr_t)da | (uintptr_t)z | (uintptr_t)dz) & 15) return YV_ERR_ARG;
    if (ws_floats < yv_bn_ws_floats(T, C)) return YV_ERR_WORKSPACE;
    RedArgs r = {};
    r.z = (const uint16_t*)z; r.ldz = ldz; r.da = (const uint16_t*)da; r.ldda = ldda; r.T = T; r.C = C;
    r.mean = mean; r.rstd = rstd; r.gamma = gamma; r.beta = beta; r.act = act; r.partial = ws;
    const int chunks = chunks_for(T, &r.rows_per_chunk);
    const int rp = 256 / (C >> 3);
    float* coef = ws + (size_t)chunks * 2 * C;
    hipLaunchKernelGGL(chan_reduce_kernel<1>, dim3(chunks), dim3(256), (size_t)rp * 2 * C * sizeof(float), (hipStream_t)stream, r);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, ws, chunks, C, T,
                       batch_stats, dgamma, dbeta, coef);
    ActArgs a = {};
    a.z = (const uint16_t*)z; a.ldz = ldz; a.da = (const uint16_t*)da; a.ldda = ldda; a.out = (uint16_t*)dz; a.ldo = lddz;
    a.T = T; a.C = C; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.coef = coef; a.act = act;
    hipLaunchKernelGGL(bn_act_kernel<true>, dim3(act_blocks(T, C)), dim3(256), 0, (hipStream_t)stream, a);
    return yv_launch_status();
}

extern "C" int yv_view_op(int mode, const void* src, long long ld_src, void* dst, long long ld_dst, int B, int H, int W, int C,
                          void* stream) {
    if (mode < 0 || mode > 6 || !dst || (mode != 5 && !src) || B <= 0 || H <= 0 || W <= 0 || C < 8 || (C & 7)) return YV_ERR_ARG;
    if ((ld_dst & 7) || ld_dst < C || (mode != 5 && ((ld_src & 7) || ld_src < C))) return YV_ERR_ARG;
    if (((uintptr_t)dst & 15) || (src && ((uintptr_t)src & 15))) return YV_ERR_ARG;
    const int up = (mode == 2 || mode == 4) ? 4 : 1;
    const long long total = mode == 6 ? (long long)B * (H + 2) * (W + 2) * (C >> 3) : (long long)B * H * W * up * (C >> 3);
    if (total < (1LL << 31))
        hipLaunchKernelGGL(view_op_kernel<uint32_t>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, mode,
                           (const uint16_t*)src, ld_src, (uint16_t*)dst, ld_dst, B, H, W, C);
    else
        hipLaunchKernelGGL(view_op_kernel<unsigned long long>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, mode,
                           (const uint16_t*)src, ld_src, (uint16_t*)dst, ld_dst, B, H, W, C);
    return yv_launch_status();
}

extern "C" int yv_maxpool5_bwd(const void* x, long long ldx, const void* dout, long long lddo, void* din, long long lddi, int B,
                               int H, int W, int C, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !dout || !din || !ws || B <= 0 || H <= 0 || W <= 0 || C < 8 || (C & 7)) return YV_ERR_ARG;
    if ((ldx & 7) || (lddo & 7) || (lddi & 7) || (((uintptr_t)x | (uintptr_t)dout | (uintptr_t)din | (uintptr_t)ws) & 15)) return YV_ERR_ARG;
    if (ws_bytes < (size_t)B * H * W * C) return YV_ERR_WORKSPACE;
    const unsigned blocks = blocks_for((long long)B * H * W * (C >> 3));
    hipLaunchKernelGGL(maxpool5_argmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, ldx, B, H, W, C,
                       (uint8_t*)ws);
    hipLaunchKernelGGL(maxpool5_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)ws,
                       (const uint16_t*)dout, lddo, (uint16_t*)din, lddi, B, H, W, C);
    return yv_launch_status();
}

extern "C" int yv_im2col3(const void* x, long long ldx, int B, int Hin, int Win, int C, int stride, void* col, void* stream) {
    if (!x || !col || B <= 0 || Hin <= 0 || Win <= 0 || C < 8 || (C & 7) || (ldx & 7) || !(stride == 1 || stride == 2))
        return YV_ERR_ARG;
    if (((uintptr_t)x | (uintptr_t)col) & 15) return YV_ERR_ARG;
    const int Hout = (Hin - 1) / stride + 1, Wout = (Win - 1) / stride + 1;       // k 3, pad 1
    const long long total = (long long)B * Hout * Wout * 9 * (C >> 3);
    if (total < (1LL << 31))
        hipLaunchKernelGGL(im2col3_kernel<uint32_t>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)x, ldx, B, Hin, Win, C, stride, Hout, Wout, (uint16_t*)col);
    else
        hipLaunchKernelGGL(im2col3_kernel<unsigned long long>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)x, ldx, B, Hin, Win, C, stride, Hout, Wout, (uint16_t*)col);
    return yv_launch_status();
}

extern "C" int yv_conv_weight_dgrad(const void* w, int Cout, int taps, int Cin, void* wd, void* stream) {
    if (!w || !wd || Cout <= 0 || Cin <= 0 || !(taps == 1 || taps == 9)) return YV_ERR_ARG;
    hipLaunchKernelGGL(weight_dgrad_kernel, dim3(blocks_for((long long)Cout * taps * Cin)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)w, Cout, taps, Cin, (uint16_t*)wd);
    return yv_launch_status();
}
