// Implicit-GEMM on CDNA4 MFMA: one kernel family for
//   * Linear  out[M,N] = A[M,K] . W[N,K]^T            (timm Attention.qkv/proj, Mlp.fc1/fc2, PatchEmbed, head)
//   * Conv2d  k in {1,3}, stride {1,2}, NHWC bf16     (ultralytics Conv/C2f/SPPF/Detect)
// with the whole elementwise tail fused into the epilogue (folded-BN bias, SiLU,
// exact-erf GELU, f32 residual-stream accumulate, bf16 shortcut add, pos_embed add).
//
// Tiling (wave = 64 lanes, v_mfma_f32_16x16x32_bf16):
//   workgroup = 256 threads = 4 waves, output tile BM x BN, K step 64 (128 B per row).
//   Both operands are K-contiguous ("NT" GEMM), staged global -> registers -> LDS
//   (issue-early / write-late, two LDS buffers, one barrier per K step).  Rows are
//   128 B in LDS with the 16-byte chunk index XOR-swizzled by (row & 7): the
//   ds_read_b128 fragment reads (16 rows x one chunk per lane group) are conflict free.
//   The MFMA "A" operand is the WEIGHT tile and "B" the activation tile, i.e. the
//   accumulator holds C^T: each lane owns 4 consecutive output channels of one
//   pixel/token, so the epilogue emits 8-byte (bf16) / 16-byte (f32) stores.
//   The conv gather (im2col, zero padding, optional 2x nearest upsample and
//   2-source channel concat) happens in the global->register stage: nothing is
//   materialised.
//   blockIdx is remapped so that the 8 XCDs each walk contiguous N tiles of the same
//   M tile (activation rows stay in that XCD's L2).
#include "yv_common.h"

namespace {

constexpr int BK = 64;            // bf16 elements per K step
constexpr int THREADS = 256;

struct GemmArgs {
    // A operand (activations)
    const uint16_t* a0;
    const uint16_t* a1;          // second concat source (1x1 conv only) or null
    int lda0, lda1;              // pixel / row stride in elements
    int c0, c1;                  // channels per source (c0 + c1 = Cin); linear: c0 = K
    int up0, up1;                // nearest-2x upsample flags
    int Hin, Win;                // logical input grid (after upsample)
    int Hout, Wout, ksize, stride;
    // W operand
    const uint16_t* w;           // (N, K) bf16
    const float* bias;
    int M, N, K;
    // output
    void* out;
    int ldo;
    const uint16_t* res;         // bf16 residual view
    int ldres;
    const float* pos;            // pos_embed (tok+1, N) f32
    int tok;
    int flags;
    const int32_t* m_dev;
    int m_mul;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int MODE /*0 linear, 1 conv*/, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(THREADS) void igemm_kernel(GemmArgs g) {
    constexpr int MF = BM / WM / 16;               // activation fragments per wave
    constexpr int NF = BN / WN / 16;               // weight fragments per wave
    constexpr int A_CH = BM * 8 / THREADS;         // 16-byte chunks per thread per K step (activations)
    constexpr int W_CH = (BN * 8 + THREADS - 1) / THREADS;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [buf][A tile | W tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int M = g.M;
    if (g.m_dev) { long long md = (long long)g.m_dev[0] * g.m_mul; M = md < M ? (int)md : M; }

    // XCD-aware bijective remap of the 1-D grid
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= M) return;

    // ---- per-thread staging coordinates -------------------------------------------------
    const int ch = tid & 7;                        // chunk (8 elements) within the 64-wide K step
    const int r_in = tid >> 3;                     // 0..31
    // activation rows handled by this thread: r_in + 32*p
    int a_valid[A_CH];
    long long a_base[A_CH];                        // linear: row offset; conv: unused
    int a_b[A_CH], a_oy[A_CH], a_ox[A_CH];
#pragma unroll
    for (int p = 0; p < A_CH; ++p) {
        const int m = m0 + r_in + 32 * p;
        a_valid[p] = m < M;
        if (MODE == 0) {
            a_base[p] = (long long)m * g.lda0;
        } else {
            const int hw = g.Hout * g.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / g.Wout;
            a_b[p] = b; a_oy[p] = oy * g.stride; a_ox[p] = (rem - oy * g.Wout) * g.stride;
        }
    }
    const int pad = g.ksize >> 1;
    const int Cin = g.c0 + g.c1;

    uint4 ra[A_CH], rw[W_CH];
    auto load_tile = [&](int kt) {
        const int k = kt * BK + ch * 8;
        const bool k_ok = k < g.K;
        if (MODE == 0) {
#pragma unroll
            for (int p = 0; p < A_CH; ++p) {
                ra[p] = make_uint4(0, 0, 0, 0);
                if (k_ok && a_valid[p]) ra[p] = *(const uint4*)(g.a0 + a_base[p] + k);
            }
        } else {
            int tap = 0, cin = k;
            if (g.ksize == 3) { tap = k / Cin; cin = k - tap * Cin; }
            const int ky = g.ksize == 3 ? tap / 3 : 0;
            const int kx = g.ksize == 3 ? tap - ky * 3 : 0;
            const bool s1 = cin >= g.c0;
            const uint16_t* src = s1 ? g.a1 : g.a0;
            const int ld = s1 ? g.lda1 : g.lda0;
            const int up = s1 ? g.up1 : g.up0;
            const int cc = s1 ? cin - g.c0 : cin;
#pragma unroll
            for (int p = 0; p < A_CH; ++p) {
                ra[p] = make_uint4(0, 0, 0, 0);
                const int iy = a_oy[p] + ky - pad, ix = a_ox[p] + kx - pad;
                if (k_ok && a_valid[p] && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) {
                    const int sy = iy >> up, sx = ix >> up, sh = g.Hin >> up, sw = g.Win >> up;
                    ra[p] = *(const uint4*)(src + ((long long)(a_b[p] * sh + sy) * sw + sx) * ld + cc);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < W_CH; ++p) {
            const int rr = r_in + 32 * p;
            rw[p] = make_uint4(0, 0, 0, 0);
            if (rr < BN && k_ok && (n0 + rr) < g.N) rw[p] = *(const uint4*)(g.w + (long long)(n0 + rr) * g.K + k);
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* A = smem + buf * (A_BYTES + W_BYTES);
        unsigned char* W = A + A_BYTES;
#pragma unroll
        for (int p = 0; p < A_CH; ++p) {
            const int rr = r_in + 32 * p;
            *(uint4*)(A + rr * 128 + ((ch ^ (rr & 7)) << 4)) = ra[p];
        }
#pragma unroll
        for (int p = 0; p < W_CH; ++p) {
            const int rr = r_in + 32 * p;
            if (rr < BN) *(uint4*)(W + rr * 128 + ((ch ^ (rr & 7)) << 4)) = rw[p];
        }
    };

    f32x4 acc[NF][MF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WN, wn = wave - wm * WN;
    const int wrow_m = wm * (BM / WM), wrow_n = wn * (BN / WN);
    const int fr = lane & 15, fq = lane >> 4;

    const int nk = (g.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char* A = smem + cur * (A_BYTES + W_BYTES);
        const unsigned char* W = A + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[MF], fw[NF];
            const int kc = ks * 4 + fq;
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int rr = wrow_m + j * 16 + fr;
                fa[j] = *(const bf16x8*)(A + rr * 128 + ((kc ^ (rr & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int rr = wrow_n + i * 16 + fr;
                fw[i] = *(const bf16x8*)(W + rr * 128 + ((kc ^ (rr & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns channels n..n+3 of row m --------------------------------------
    const int flags = g.flags;
#pragma unroll
    for (int j = 0; j < MF; ++j) {
        const int m = m0 + wrow_m + j * 16 + fr;
        if (m >= M) continue;
        long long orow = m;
        const float* posrow = nullptr;
        if (flags & YV_EPI_POSEMB) {
            const int r = m / g.tok, t = m - r * g.tok;
            orow = (long long)r * (g.tok + 1) + 1 + t;
            posrow = g.pos + (long long)(1 + t) * g.N;
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int n = n0 + wrow_n + i * 16 + fq * 4;
            if (n >= g.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (flags & YV_EPI_BIAS) {
                const float4 b = *(const float4*)(g.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (flags & YV_EPI_SILU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = silu_f(v[q]);
            }
            if (flags & YV_EPI_GELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = gelu_f(v[q]);
            }
            if (flags & YV_EPI_RES_BF16) {
                const uint2 rr = *(const uint2*)(g.res + orow * g.ldres + n);
                v[0] += bf16_to_f32((uint16_t)(rr.x & 0xffff)); v[1] += bf16_to_f32((uint16_t)(rr.x >> 16));
                v[2] += bf16_to_f32((uint16_t)(rr.y & 0xffff)); v[3] += bf16_to_f32((uint16_t)(rr.y >> 16));
            }
            if (posrow) {
                const float4 pp = *(const float4*)(posrow + n);
                v[0] += pp.x; v[1] += pp.y; v[2] += pp.z; v[3] += pp.w;
            }
            if (flags & (YV_EPI_OUT_F32 | YV_EPI_RES_F32)) {
                float* o = (float*)g.out + orow * g.ldo + n;
                if (flags & YV_EPI_RES_F32) {
                    const float4 x = *(const float4*)o;
                    v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
                }
                *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                uint16_t* o = (uint16_t*)g.out + orow * g.ldo + n;
                *(uint2*)o = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
    }
}

template <int MODE, int BM, int BN, int WM, int WN>
int launch(GemmArgs& g, hipStream_t st) {
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const size_t lds = 2 * (size_t)(BM + BN) * 128;
    auto kern = igemm_kernel<MODE, BM, BN, WM, WN>;
    if (lds > 65536) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return YV_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(THREADS), lds, st, g);
    return yv_launch_status();
}

template <int MODE>
int dispatch(GemmArgs& g, hipStream_t st) {
    if (g.N > 64) return launch<MODE, 128, 128, 2, 2>(g, st);
    if (g.N > 32) return launch<MODE, 128, 64, 4, 1>(g, st);
    if (g.N > 16) return launch<MODE, 128, 32, 4, 1>(g, st);
    return launch<MODE, 128, 16, 4, 1>(g, st);
}

}  // namespace

extern "C" int yv_linear(const void* A, int lda, const void* W, const float* bias, int M, int N, int K, void* out,
                         int ldo, const float* pos, int tok, int flags, const int32_t* m_dev, int m_mul, void* stream) {
    if (!A || !W || !out || M < 0 || N <= 0 || K <= 0) return YV_ERR_ARG;
    if ((K & 7) || (lda & 7) || (N & 3) || (ldo & 3)) return YV_ERR_ARG;            // 16-byte operand chunks, 4-wide stores
    if ((flags & YV_EPI_BIAS) && !bias) return YV_ERR_ARG;
    if ((flags & YV_EPI_POSEMB) && (!pos || tok <= 0)) return YV_ERR_ARG;
    if (flags & (YV_EPI_SILU | YV_EPI_RES_BF16)) return YV_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)out) & 15) return YV_ERR_ARG;
    if (M == 0) return YV_OK;
    GemmArgs g = {};
    g.a0 = (const uint16_t*)A; g.lda0 = lda; g.c0 = K;
    g.w = (const uint16_t*)W; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.out = out; g.ldo = ldo; g.pos = pos; g.tok = tok; g.flags = flags; g.m_dev = m_dev; g.m_mul = m_mul;
    g.ksize = 1; g.stride = 1;
    return dispatch<0>(g, (hipStream_t)stream);
}

extern "C" int yv_conv2d(const yv_view* in0, const yv_view* in1, int B, int Hout, int Wout, int ksize, int stride,
                         const void* weight, const float* bias, int Cout, void* out, int out_ld, const void* res,
                         int res_ld, int flags, void* stream) {
    if (!in0 || !in0->ptr || !weight || !out || B <= 0 || Hout <= 0 || Wout <= 0 || Cout <= 0) return YV_ERR_ARG;
    if (!(ksize == 1 || ksize == 3) || !(stride == 1 || stride == 2)) return YV_ERR_ARG;
    if (in1 && in1->ptr && ksize != 1) return YV_ERR_ARG;
    const int c1 = (in1 && in1->ptr) ? in1->c : 0;
    const int Cin = in0->c + c1;
    if ((in0->c & 7) || (c1 & 7) || (in0->ld & 7) || (c1 && (in1->ld & 7)) || (Cout & 3) || (out_ld & 3))
        return YV_ERR_ARG;
    if ((flags & YV_EPI_BIAS) && !bias) return YV_ERR_ARG;
    if ((flags & YV_EPI_RES_BF16) && (!res || (res_ld & 3))) return YV_ERR_ARG;
    if (flags & (YV_EPI_GELU | YV_EPI_POSEMB | YV_EPI_RES_F32)) return YV_ERR_ARG;
    if ((long long)B * Hout * Wout > 0x7fffffffLL) return YV_ERR_LIMIT;
    GemmArgs g = {};
    g.a0 = (const uint16_t*)in0->ptr; g.lda0 = in0->ld; g.c0 = in0->c; g.up0 = in0->up;
    if (c1) { g.a1 = (const uint16_t*)in1->ptr; g.lda1 = in1->ld; g.c1 = c1; g.up1 = in1->up; }
    g.Hin = Hout * stride; g.Win = Wout * stride;
    g.Hout = Hout; g.Wout = Wout; g.ksize = ksize; g.stride = stride;
    g.w = (const uint16_t*)weight; g.bias = bias;
    g.M = B * Hout * Wout; g.N = Cout; g.K = ksize * ksize * Cin;
    g.out = out; g.ldo = out_ld; g.res = (const uint16_t*)res; g.ldres = res_ld; g.flags = flags;
    return dispatch<1>(g, (hipStream_t)stream);
}
