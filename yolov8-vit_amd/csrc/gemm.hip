// Implicit-GEMM on CDNA4 MFMA: one kernel family for
//   * Linear  out[M,N] = A[M,K] . W[N,K]^T            (timm Attention.qkv/proj, Mlp.fc1/fc2, PatchEmbed, head)
//   * Conv2d  k in {1,3}, stride {1,2}, NHWC bf16     (ultralytics Conv/C2f/SPPF/Detect)
// with the whole elementwise tail fused into the epilogue (folded-BN bias, SiLU,
// erf-form GELU (erfc approximated to 1.5e-7), f32 residual-stream accumulate, bf16 shortcut add, pos_embed add).
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
#include <string.h>
#include <type_traits>
#include <map>
#include <mutex>
#include <atomic>
#include "yv_common.h"
#include <hip/hip_ext.h>

namespace {

constexpr int BK = 64;            // bf16 elements per K step
int g_opt_p9_small = 1;              // gemm_p9_kernel: 128 / 96-row tiles allowed ("linear_p9_small")
int g_opt_p9_small_fixed = 48;       // ... and the fixed part of their cost per K tile, in rows ("linear_p9_small_fixed")
int g_opt_wgrad_split = 0;           // > 0: forced number of token slices of a matrix-shaped weight gradient (tools/wgrad_bench.py)
int g_opt_wgrad_cap = 128;          // token slices of a conv-shaped weight gradient (few output tiles, 10^5+ rows)
thread_local hipEvent_t t_time_start = nullptr, t_time_stop = nullptr;   // yv_set_launch_timing: next gemm_dma launch
int g_opt_variant = 1;            // 1 = auto; tuning knobs (yv_set_option): linear kernel variant, M-group size, persistent grid
int g_opt_group_m = 8;
int g_opt_staged = 1;
int g_opt_p8 = 3;                  // persistent kernels: 0 off, 1 8-phase kernel for wide bf16-output linears only (qkv, fc1), 2 for every
                                   // eligible linear incl. the f32 residual ones (proj, fc2), 3 the free-running kernel (gemm_p9_kernel): "linear_p8"
std::mutex g_ws_mu;
std::map<void*, std::pair<void*, size_t>> g_ws;   // per-stream split-K workspace (yv_set_workspace)
static bool ws_lookup(void* stream, void** ws, size_t* bytes) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_ws.find(stream);
    if (it == g_ws.end()) return false;
    *ws = it->second.first; *bytes = it->second.second;
    return true;
}
int g_opt_linear_splitk = 1;
int g_opt_splitk = 0;           // measured neutral end-to-end (tools/e2e_ab.py): the reduce pass costs what the shorter chain saves
constexpr int THREADS = 256;

struct GemmArgs {
    // A operand (activations)
    const uint16_t* a0;
    const uint16_t* a1;          // second concat source (1x1 conv only) or null
    int lda0, lda1;              // pixel / row stride in elements
    int seg_len;                 // gemm_tn only: X column k lives at (k / seg_len) * seg_stride + k % seg_len (0: plain)
    long long seg_stride;
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
    int group_m;
    int ldw;                     // WT kernels: row stride of the reduction-major weight (K, N)
    const float* resf;           // f32 residual source (null: read-modify-write `out`)
    uint16_t* aux;               // bf16 side buffer: SAVE_PRE target / GELU_BWD pre-activation
    int ldaux;
    int cin_shift;               // conv: log2(c0 + c1) when that is a power of two, else -1
    int tap_uniform;             // conv: (c0 + c1) % 64 == 0, a K step lies inside one tap
    int splitk;                  // conv only: K range split over `splitk` workgroups per tile (partials in `partial`)
    float* partial;              // (splitk, M, N) f32
    int staged;                  // coalesced LDS-staged epilogue usable (alignment / width checked on the host)
    int sched;                   // gemm_p8: 0 = the grid strides through the tile sequence round by round, 1 = one contiguous
                                 //          share of the sequence per XCD
    uint8_t* mxq;                // YV_EPI_OUT_MXFP8: e4m3 image of the output (row stride ldmxq bytes) ...
    long long ldmxq;
    uint8_t* mxs;                // ... and its E8M0 block scales, K-step-major (N/128, mx_rows, 4)
    long long mx_rows;
    // gemm_p9_kernel<MX>: E8M0 scales of the fp8 operands, K-step-major (K/128, rows, 4) (a0 / w then point at e4m3 bytes, lda0 in bytes)
    const uint8_t* mx_sa;
    const uint8_t* mx_sw;
    long long mx_rows_a, mx_rows_w;
};

// x * sigmoid(x) through the hardware reciprocal (1 ulp) instead of an IEEE division: the division's scale / fixup sequence was
// ~12 of the ~20 VALU instructions per output value of every detector convolution, in kernels that are VALU-issue-bound.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// 256 bytes of zeros: where a staged chunk is padding (outside the image, past K, past the last row) the conv / linear gather
// reads THIS instead of branching around the load or masking the data afterwards
__device__ __attribute__((aligned(256))) uint32_t g_zero_page[64];
// erf-form GELU through x * sigmoid(x * (p0 + p1 x^2 + p2 x^4)), coefficients fitted (minimax, |x| <= 8) against
// 0.5 x (1 + erf(x / sqrt 2)): max abs error 2.5e-5 - below half a bf16 step of the output everywhere the output exceeds
// 0.01 in magnitude.  x^2 is clamped at 64 (beyond |x| = 8 the result is x or 0 to f32 precision; the quartic would turn over).
__device__ __forceinline__ float gelu_f(float x) {
    const float x2 = fminf(x * x, 64.0f);
    float q = fmaf(-7.03039117e-4f * -1.4426950408889634f, x2, 7.40113286e-2f * -1.4426950408889634f);
    q = fmaf(q, x2, 1.59501573f * -1.4426950408889634f);
    const float e = __builtin_amdgcn_exp2f(x * q);                  // exp(-z)
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// (round 1 evaluated erfc by Abramowitz-Stegun 7.1.26: 14 operations per value against 9 here.  ONE definition for every forward
// kernel: schedules that route a linear through different kernels - full batch vs half batches - must agree bit for bit.)
// d/dx gelu(x) = Phi(x) + x * phi(x) of the erf form (erfc by Abramowitz-Stegun 7.1.26, |abs err| <= 1.5e-7); the forward's
// sigmoid fit differs from the erf form by <= 2.5e-5, i.e. this is its derivative to ~1e-4
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(x), 1.0f));
    float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, 0.5f * -0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);     // exp(-x^2/2)
    const float w = p * t * e;                                                       // 0.5*erfc(|x|/sqrt2)
    const float cdf = x >= 0.f ? 1.0f - w : w;
    return cdf + x * e * 0.39894228040143267794f;
}

template <int MF, int NF>
__device__ __forceinline__ void epilogue(const GemmArgs& g, f32x4 (&acc)[NF][MF], int M, int m0, int n0, int wrow_m,
                                         int wrow_n, int fr, int fq) {
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


// Coalesced epilogue for 64-column wave tiles (NF == 4).  The ablation of the first version showed
// the direct epilogue (8-byte stores, 32-byte row segments) costing 41 % of a K = 768 GEMM: the
// output left at ~2 TB/s.  Here each wave transposes its tile through a private, XOR-swizzled LDS
// slab (reusing the main-loop buffers after the loop's last barrier) and writes whole 128-byte
// (bf16) / 256-byte (f32) row segments with 16-byte stores; residual reads are coalesced the same way.
template <int MF>
__device__ __forceinline__ void epilogue_staged(const GemmArgs& g, f32x4 (&acc)[4][MF], int M, int m0, int n0,
                                                int wrow_m, int wrow_n, int lane, unsigned char* stage) {
    const int flags = g.flags;
    const int fr = lane & 15, fq = lane >> 4;
    const int nb = n0 + wrow_n;
    float4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = nb + i * 16 + fq * 4;
        bv[i] = ((flags & YV_EPI_BIAS) && n < g.N) ? *(const float4*)(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto value = [&](int i, int j, float* v, bool act = true) {
        v[0] = acc[i][j][0] + bv[i].x; v[1] = acc[i][j][1] + bv[i].y;
        v[2] = acc[i][j][2] + bv[i].z; v[3] = acc[i][j][3] + bv[i].w;
        if (!act) return;
        if (flags & YV_EPI_SILU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = silu_f(v[q]);
        }
        if (flags & YV_EPI_GELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = gelu_f(v[q]);
        }
    };
    if (!(flags & (YV_EPI_OUT_F32 | YV_EPI_RES_F32))) {
        // ---- bf16 output: MF*16 rows x 128 B slab -------------------------------------------------
        if (flags & YV_EPI_SAVE_PRE) {           // training: keep the pre-activation (GELU'(u) needs it)
#pragma unroll
            for (int j = 0; j < MF; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v[4];
                    value(i, j, v, false);
                    const int row = j * 16 + fr, c16 = i * 2 + (fq >> 1);
                    *(uint2*)(stage + row * 128 + ((c16 ^ (row & 7)) << 4) + (fq & 1) * 8) =
                        make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                }
#pragma unroll
            for (int it = 0; it < MF * 2; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                const int m = m0 + wrow_m + row, n = nb + ch * 8;
                const uint4 pk = *(const uint4*)(stage + row * 128 + ((ch ^ (row & 7)) << 4));
                if (m < M && n < g.N) *(uint4*)(g.aux + (long long)m * g.ldaux + n) = pk;
            }
        }
#pragma unroll
        for (int j = 0; j < MF; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[4];
                value(i, j, v);
                const int row = j * 16 + fr, c16 = i * 2 + (fq >> 1);
                *(uint2*)(stage + row * 128 + ((c16 ^ (row & 7)) << 4) + (fq & 1) * 8) =
                    make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
#pragma unroll
        for (int it = 0; it < MF * 2; ++it) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            const int m = m0 + wrow_m + row, n = nb + ch * 8;
            uint4 pk = *(const uint4*)(stage + row * 128 + ((ch ^ (row & 7)) << 4));
            if (m < M && n < g.N) {
                if (flags & YV_EPI_RES_BF16) {
                    const uint4 rr = *(const uint4*)(g.res + (long long)m * g.ldres + n);
                    const uint32_t a[4] = {pk.x, pk.y, pk.z, pk.w}, b[4] = {rr.x, rr.y, rr.z, rr.w};
                    uint32_t o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[q] = pack_bf16x2(bf16_to_f32((uint16_t)(a[q] & 0xffff)) + bf16_to_f32((uint16_t)(b[q] & 0xffff)),
                                           bf16_to_f32((uint16_t)(a[q] >> 16)) + bf16_to_f32((uint16_t)(b[q] >> 16)));
                    pk = make_uint4(o[0], o[1], o[2], o[3]);
                }
                if (flags & YV_EPI_OUT_MXFP8) {
                    // the consumer is another MXFP8 GEMM: quantise the 8 (bf16-rounded) values of this lane together with
                    // the 3 lanes that hold the rest of their 32-column block, skip the bf16 store
                    const uint32_t a[4] = {pk.x, pk.y, pk.z, pk.w};
                    float f[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { f[2 * q] = bf16_to_f32((uint16_t)(a[q] & 0xffff)); f[2 * q + 1] = bf16_to_f32((uint16_t)(a[q] >> 16)); }
                    float amax = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) amax = fmaxf(amax, fabsf(f[q]));
                    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                    int e = -127;
                    if (amax > 0.f) {
                        int ex;
                        const float mant = frexpf(amax * (1.0f / 448.0f), &ex);
                        e = mant == 0.5f ? ex - 1 : ex;
                        e = e < -127 ? -127 : (e > 127 ? 127 : e);
                    }
                    const float inv = ldexpf(1.0f, -e);
                    int p0 = 0, p1 = 0;
                    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, p0, false);
                    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, p0, true);
                    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, p1, false);
                    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, p1, true);
                    *(uint2*)(g.mxq + (long long)m * g.ldmxq + n) = make_uint2((uint32_t)p0, (uint32_t)p1);
                    if ((ch & 3) == 0) {
                        const int bk = n >> 5;
                        g.mxs[((long long)(bk >> 2) * g.mx_rows + m) * 4 + (bk & 3)] = (uint8_t)(e + 127);
                    }
                    continue;
                }
                if (flags & YV_EPI_GELU_BWD) {      // out = dg * gelu'(u), u = saved pre-activation (bf16)
                    const uint4 uu = *(const uint4*)(g.aux + (long long)m * g.ldaux + n);
                    const uint32_t a[4] = {pk.x, pk.y, pk.z, pk.w}, b[4] = {uu.x, uu.y, uu.z, uu.w};
                    uint32_t o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[q] = pack_bf16x2(bf16_to_f32((uint16_t)(a[q] & 0xffff)) * gelu_grad_f(bf16_to_f32((uint16_t)(b[q] & 0xffff))),
                                           bf16_to_f32((uint16_t)(a[q] >> 16)) * gelu_grad_f(bf16_to_f32((uint16_t)(b[q] >> 16))));
                    pk = make_uint4(o[0], o[1], o[2], o[3]);
                }
                *(uint4*)((uint16_t*)g.out + (long long)m * g.ldo + n) = pk;
            }
        }
    } else {
        // ---- f32 output / residual stream: two passes of MF*8 rows x 256 B ----------------------------
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int jj = 0; jj < MF / 2; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v[4];
                    value(i, half * (MF / 2) + jj, v);
                    const int row = jj * 16 + fr, c16 = i * 4 + fq;
                    *(float4*)(stage + row * 256 + ((c16 ^ (row & 15)) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
                }
#pragma unroll
            for (int it = 0; it < MF * 2; ++it) {
                const int row = it * 4 + (lane >> 4), ch = lane & 15;
                const int m = m0 + wrow_m + half * (MF * 8) + row, n = nb + ch * 4;
                float4 v = *(const float4*)(stage + row * 256 + ((ch ^ (row & 15)) << 4));
                if (m < M && n < g.N) {
                    long long orow = m;
                    if (flags & YV_EPI_POSEMB) {
                        const int r = m / g.tok, t = m - r * g.tok;
                        orow = (long long)r * (g.tok + 1) + 1 + t;
                        const float4 pp = *(const float4*)(g.pos + (long long)(1 + t) * g.N + n);
                        v.x += pp.x; v.y += pp.y; v.z += pp.z; v.w += pp.w;
                    }
                    float* o = (float*)g.out + orow * g.ldo + n;
                    if (flags & YV_EPI_RES_F32) {
                        const float4 x = g.resf ? *(const float4*)(g.resf + orow * g.ldo + n) : *(const float4*)o;
                        v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                    }
                    *(float4*)o = v;
                }
            }
        }
    }
}

template <int MF, int NF>
__device__ __forceinline__ void finish_tile(const GemmArgs& g, f32x4 (&acc)[NF][MF], int M, int m0, int n0, int wrow_m,
                                            int wrow_n, int lane, int wave, unsigned char* smem) {
    if constexpr (NF == 4 && (MF % 2) == 0) {
        if (g.staged) {          // wave-private slab of MF*16 rows x 128 B inside the (now idle) tile buffers
            epilogue_staged<MF>(g, acc, M, m0, n0, wrow_m, wrow_n, lane, smem + wave * (MF * 16 * 128));
            return;
        }
    }
    epilogue<MF, NF>(g, acc, M, m0, n0, wrow_m, wrow_n, lane & 15, lane >> 4);
}

template <int MODE /*0 linear, 1 conv*/, int BM, int BN, int WM, int WN, bool TWO = false /*conv with two concatenated sources*/>
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
    const int S = g.splitk > 1 ? g.splitk : 1;
    const int slice = bid % S;                 // K slices of one tile are neighbours: their A/W rows share L2
    bid /= S;
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= M) return;

    // ---- per-thread staging coordinates -------------------------------------------------
    // Everything that does not change along K is worked out once per tile: per activation row the byte offset of the output
    // pixel's centre in each source and the 3 x 3 taps that fall inside the image (9 bits), per weight row its byte offset.  A K
    // step then costs one tap decode per thread (its 8-channel chunk is fixed) and an add + a bit test per 16-byte load - the
    // first version redid the pixel arithmetic (64-bit multiplies, a division) for every load: 45 VALU instructions per load,
    // 330 per wave and step against 32 MFMAs (SQ_INSTS_VALU / SQ_INSTS_VMEM_RD over the detector's launches).
    const int ch = tid & 7;                        // chunk (8 elements) within the 64-wide K step
    const int r_in = tid >> 3;                     // 0..31
    // activation rows handled by this thread: r_in + 32*p
    int a_valid[A_CH];
    long long a_base[A_CH];                        // linear: row offset
    uint32_t a_off0[A_CH], a_off1[A_CH], a_taps[A_CH];      // conv (32-bit byte offsets: conv_impl bounds the tensors)
    const int pad = g.ksize >> 1;
    const int Cin = g.c0 + g.c1;
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
            const int cy = oy * g.stride, cx = (rem - oy * g.Wout) * g.stride;
            // (both include this lane's 8-channel chunk; a_off1 is kept as the DIFFERENCE to a_off0: a runtime choice between two
            //  register arrays would put both in scratch)
            a_off0[p] = (uint32_t)((((long long)b * (g.Hin >> g.up0) + (cy >> g.up0)) * (g.Win >> g.up0) + (cx >> g.up0)) * g.lda0 * 2) + ch * 16;
            a_off1[p] = g.c1 ? (uint32_t)((((long long)b * (g.Hin >> g.up1) + (cy >> g.up1)) * (g.Win >> g.up1) + (cx >> g.up1)) * g.lda1 * 2) + ch * 16 - a_off0[p] : 0u;
            uint32_t taps = 0;
            if (a_valid[p]) {
                if (g.ksize == 3) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int iy = cy + t / 3 - 1, ix = cx + t % 3 - 1;
                        taps |= (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) ? (1u << t) : 0u;
                    }
                } else {
                    taps = 1u;
                }
            }
            a_taps[p] = taps;
        }
    }
    uint32_t w_off[W_CH];
    bool w_ok[W_CH];
#pragma unroll
    for (int p = 0; p < W_CH; ++p) {
        const int rr = r_in + 32 * p;
        w_ok[p] = rr < BN && (n0 + rr) < g.N;
        w_off[p] = (uint32_t)(((long long)(n0 + rr) * g.K + ch * 8) * 2);
    }

    // Loads are UNCONDITIONAL.  Written as "if (valid) r = load" hipcc branches around every load and waits vmcnt(0) before the
    // next one - 8 dependent memory round trips per K step instead of 8 loads in flight (s_waitcnt vmcnt(0) in front of every
    // global_load of the first version).  Conv: buffer loads, a chunk that is padding / past K / past the last row gets an
    // out-of-range offset and reads zeros through the descriptor's range check; whatever is the same for the whole workgroup in a
    // K step (the tap's displacement, the channel base, the weight column) travels in the instruction's SCALAR offset, so a load
    // costs a bit test and a select.  The scalar offset is unsigned, hence the descriptor's base one row + one pixel before the
    // tensor.  Linear: a 64-bit address chosen between the operand and g_zero_page.
    uint4 ra[A_CH], rw[W_CH];
    const unsigned char* zp = (const unsigned char*)g_zero_page;
    typedef const __attribute__((address_space(1))) u32x4* g16_t;   // (global, not generic: a flat load would also count in lgkmcnt)
    auto ld16 = [&](bool ok, const unsigned char* ptr) __attribute__((always_inline)) {
        g16_t q = (g16_t)(ok ? ptr : zp);
        asm volatile("" : "+v"(q));                         // an opaque VALUE: hipcc would turn the choice back into two branches
        const u32x4 v = *q;
        return make_uint4(v[0], v[1], v[2], v[3]);
    };
    constexpr uint32_t OOB = 0x80000000u;
    const uint32_t bias0 = MODE == 1 && g.ksize == 3 ? (uint32_t)((g.Win + 1) * g.lda0 * 2) : 0u;
    // (copies first: a ternary between two FIELDS of the by-value kernel argument selects between their addresses; and the
    //  descriptors are built inside the lambda - captured by reference they are objects hipcc cannot keep out of memory, and the
    //  whole closure, kernel argument included, lands in scratch)
    const unsigned char* const a0p = (const unsigned char*)g.a0 - bias0;
    const unsigned char* const a1p = (const unsigned char*)g.a1;
    const unsigned char* const wp = (const unsigned char*)g.w;
    auto load_tile = [&](int kt) __attribute__((always_inline)) {
        const int kb = kt * BK;                                // (uniform)
        const int k = kb + ch * 8;
        const bool k_ok = k < g.K;
        if (MODE == 0) {
#pragma unroll
            for (int p = 0; p < A_CH; ++p)
                ra[p] = ld16(k_ok && a_valid[p], (const unsigned char*)(g.a0 + a_base[p] + k));
#pragma unroll
            for (int p = 0; p < W_CH; ++p)
                rw[p] = ld16(w_ok[p] && k_ok, (const unsigned char*)g.w + (uint32_t)(w_off[p] + (uint32_t)kb * 2));
        } else if constexpr (TWO) {
            // two concatenated sources (1 x 1 convolutions after a Concat): the source is a per-lane property in general, so this
            // instance keeps 64-bit addresses (a choice between two descriptors - as objects, as base pointers, or as two
            // branches that both load into ra[] - makes hipcc keep the arrays and the kernel argument in scratch)
            const bool s1 = k_ok && k >= g.c0;
            const unsigned char* src = (const unsigned char*)(s1 ? a1p : a0p);
            const uint32_t m1 = s1 ? 0xFFFFFFFFu : 0u;
            const uint32_t d = (uint32_t)((s1 ? k - g.c0 : k) * 2) - ch * 16;
#pragma unroll
            for (int p = 0; p < A_CH; ++p)
                ra[p] = ld16(k_ok && (a_taps[p] & 1u), src + (uint32_t)(a_off0[p] + (a_off1[p] & m1) + d));
#pragma unroll
            for (int p = 0; p < W_CH; ++p)
                rw[p] = ld16(w_ok[p] && k_ok, wp + (uint32_t)(w_off[p] + (uint32_t)kb * 2));
        } else {
            int tap = 0;                                       // (per lane only when a K step spans several taps: Cin < 64)
            uint32_t soff = bias0, dl = 0;
            if (g.ksize == 3) {
                const int kk = g.tap_uniform ? kb : k;
                tap = g.cin_shift >= 0 ? (kk >> g.cin_shift) : kk / Cin;
                const int cin = kk - tap * Cin;
                const int ky = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0), kx = tap - ky * 3;
                const uint32_t d = (uint32_t)((((ky - pad) * g.Win + (kx - pad)) * g.lda0 + cin) * 2);
                if (g.tap_uniform) soff += (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                else dl = d - ch * 16;
            } else {
                soff = (uint32_t)(kb * 2);
            }
            const auto rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a0p, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int p = 0; p < A_CH; ++p) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs0, (k_ok && ((a_taps[p] >> tap) & 1u)) ? a_off0[p] + dl : OOB, soff, 0);
                ra[p] = make_uint4(v[0], v[1], v[2], v[3]);
            }
            const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int p = 0; p < W_CH; ++p) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, (w_ok[p] && k_ok) ? w_off[p] : OOB, (uint32_t)kb * 2, 0);
                rw[p] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
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

    const int nk_all = (g.K + BK - 1) / BK;
    const int kt0 = (int)((long long)nk_all * slice / S), nk = (int)((long long)nk_all * (slice + 1) / S);
    // ONE LDS buffer: the next step's operands wait in registers while this step's MFMAs read the tile, and go to LDS between
    // two barriers.  The launches are latency-bound (a gather per step, ~0.2 us of MFMAs), so what counts is how many workgroups
    // a CU holds, and half the LDS (32 KB at 128 x 128) lets the register budget decide: 3 per CU instead of 2.
    load_tile(kt0);
    store_tile(0);
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        constexpr int cur = 0;
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
        __syncthreads();                                       // every wave has read the tile
        if (kt + 1 < nk) store_tile(0);
        __syncthreads();
    }

    if (S > 1) {                               // raw partial sums; bias / activation happen in splitk_reduce_kernel
        float* P = g.partial + (long long)slice * g.M * g.N;
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int m = m0 + wrow_m + j * 16 + fr;
            if (m >= M) continue;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int n = n0 + wrow_n + i * 16 + fq * 4;
                if (n < g.N) *(float4*)(P + (long long)m * g.N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        return;
    }
    finish_tile<MF, NF>(g, acc, M, m0, n0, wrow_m, wrow_n, lane, wave, smem);
}

// second stage of a split-K conv: sum the K slices, then the usual epilogue (bias, SiLU, bf16 shortcut, store)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int n4 = g.N >> 2;
    if (i >= (long long)g.M * n4) return;
    const int m = (int)(i / n4), n = (int)(i - (long long)m * n4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    // four slices per trip, loads first: a pure read pass lives on the loads in flight; summation order unchanged
    const float* base = g.partial + (long long)m * g.N + n;
    const long long slice = (long long)g.M * g.N;
    int s = 0;
    for (; s + 3 < g.splitk; s += 4) {
        const float4 p0 = *(const float4*)(base + (s + 0) * slice), p1 = *(const float4*)(base + (s + 1) * slice);
        const float4 p2 = *(const float4*)(base + (s + 2) * slice), p3 = *(const float4*)(base + (s + 3) * slice);
        v.x += p0.x; v.y += p0.y; v.z += p0.z; v.w += p0.w;
        v.x += p1.x; v.y += p1.y; v.z += p1.z; v.w += p1.w;
        v.x += p2.x; v.y += p2.y; v.z += p2.z; v.w += p2.w;
        v.x += p3.x; v.y += p3.y; v.z += p3.z; v.w += p3.w;
    }
    for (; s < g.splitk; ++s) {
        const float4 p = *(const float4*)(base + s * slice);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    const int flags = g.flags;
    if (flags & YV_EPI_BIAS) {
        const float4 b = *(const float4*)(g.bias + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (flags & YV_EPI_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
    if (flags & YV_EPI_RES_BF16) {
        const uint2 rr = *(const uint2*)(g.res + (long long)m * g.ldres + n);
        v.x += bf16_to_f32((uint16_t)(rr.x & 0xffff)); v.y += bf16_to_f32((uint16_t)(rr.x >> 16));
        v.z += bf16_to_f32((uint16_t)(rr.y & 0xffff)); v.w += bf16_to_f32((uint16_t)(rr.y >> 16));
    }
    if (flags & YV_EPI_OUT_F32) *(float4*)((float*)g.out + (long long)m * g.ldo + n) = v;
    else *(uint2*)((uint16_t*)g.out + (long long)m * g.ldo + n) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}


// ---------------------------------------------------------------------------------------------
// Linear fast path (K % 64 == 0): both tiles go global -> LDS by LDS-DMA (global_load_lds, 16 B per
// lane, no staging VGPRs, no ds_write pass).  One wave-instruction fills 8 rows x 128 B of the
// lane-linear LDS image, so the chunk swizzle is applied to the SOURCE address (chunk' ^ (row & 7))
// and again on the fragment read - the same involution on both sides.  Two LDS buffers; the DMA of
// K step t+1 is issued before the MFMAs of step t and drained (vmcnt(0)) at the step's only barrier.
// Rows past the end of a matrix are clamped to its last row: their results are never stored.
// ---------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

// reduction-major tiles ([64 reduction rows][128 columns] bf16, 256-byte LDS rows): chunk swizzle and the
// hardware-transposing fragment read (8 consecutive reduction elements 32*ks + 8*(lane>>4) + 0..7 of column
// col0 + (lane&15)): two ds_read_b64_tr_b16, lane 4q+p of a 16-lane group addressing row q, columns 4p..4p+3
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int ks, int lane, int col0) {
    const int fi = lane & 15, mg = lane >> 4, q = fi >> 2, p = fi & 3;
    const int row = ks * 32 + mg * 8 + q;
    const int ch = (col0 + 4 * p) >> 3, off = ((col0 + 4 * p) & 7) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(tile + row * 256 + ((ch ^ tn_swz(row)) << 4) + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(tile + (row + 4) * 256 + ((ch ^ tn_swz(row + 4)) << 4) + off));
    const u32x4 pk = {__builtin_bit_cast(u32x2, lo)[0], __builtin_bit_cast(u32x2, lo)[1],
                      __builtin_bit_cast(u32x2, hi)[0], __builtin_bit_cast(u32x2, hi)[1]};
    return __builtin_bit_cast(bf16x8, pk);
}

template <int BM, int BN, int WM, int WN, int ABL = 0, bool WT = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_dma_kernel(GemmArgs g) {
    // ABL (diagnostic builds only): 1 no in-loop DMA, 2 no MFMA, 3 no fragment reads, 4 no epilogue
    // WT: the weight operand is stored REDUCTION-major, W (K, N) with row stride g.ldw (dgrad: dX = dY . W reads the
    //     master-layout weight directly): its tile is [64 k][BN n] and its fragments are hardware-transposed reads
    constexpr int NW = WM * WN;                                // waves per workgroup
    constexpr int MF = BM / WM / 16, NF = BN / WN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    constexpr int A_INS = BM / (8 * NW), W_INS = BN / (8 * NW);   // wave-instructions per wave per K step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int M = g.M;
    if (g.m_dev) { long long md = (long long)g.m_dev[0] * g.m_mul; M = md < M ? (int)md : M; }
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    // split-K (wgrad-shaped problems: few output tiles, long K): slices of one tile are neighbouring workgroups
    const int S = g.splitk > 1 ? g.splitk : 1;
    const int slice = bid % S;
    bid /= S;
    // grouped order inside the XCD's chunk: GM consecutive M tiles share each W tile while it is hot in L2
    int tm, tn;
    {
        const int GM = g.group_m, per = GM * g.tiles_n;
        const int grp = bid / per, first = grp * GM;
        const int gsz = (g.tiles_m - first) < GM ? (g.tiles_m - first) : GM;
        const int in = bid - grp * per;
        tm = first + in % gsz;
        tn = in / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= M) return;

    // per-lane source rows (swizzled chunk) for every DMA instruction of this wave
    const int lrow = lane >> 3, lch = lane & 7;
    const uint16_t* a_src[A_INS];
    const uint16_t* w_src[W_INS];
#pragma unroll
    for (int j = 0; j < A_INS; ++j) {
        const int r = (j * NW + wave) * 8 + lrow;
        int m = m0 + r;
        m = m < g.M ? m : g.M - 1;
        a_src[j] = g.a0 + (long long)m * g.lda0 + ((lch ^ (r & 7)) << 3);
    }
#pragma unroll
    for (int j = 0; j < W_INS; ++j) {
        if constexpr (!WT) {
            const int r = (j * NW + wave) * 8 + lrow;
            int n = n0 + r;
            n = n < g.N ? n : g.N - 1;
            w_src[j] = g.w + (long long)n * g.K + ((lch ^ (r & 7)) << 3);
        } else {                                           // 4 reduction rows x 256 B per wave-instruction
            static_assert(!WT || BN == 128, "reduction-major weight tiles are 128 columns wide");
            const int r = (j * NW + wave) * 4 + (lane >> 4);
            int cn = n0 + (((lane & 15) ^ tn_swz(r)) << 3);
            cn = cn < g.N ? cn : g.N - 8;
            w_src[j] = g.w + (long long)r * g.ldw + cn;
        }
    }
    auto issue = [&](int kt, int buf) {
        unsigned char* A = smem + buf * (A_BYTES + W_BYTES);
        unsigned char* W = A + A_BYTES;
#pragma unroll
        for (int j = 0; j < A_INS; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[j] + kt * BK), (lptr_t)(A + (j * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < W_INS; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(w_src[j] + (WT ? (long long)kt * BK * g.ldw : (long long)kt * BK)),
                                             (lptr_t)(W + (j * NW + wave) * 1024), 16, 0, 0);
    };

    f32x4 acc[NF][MF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wm = wave / WN, wn = wave - wm * WN;
    const int wrow_m = wm * (BM / WM), wrow_n = wn * (BN / WN);
    const int fr = lane & 15, fq = lane >> 4;

    const int nk_all = g.K / BK;
    const int kt0 = (int)((long long)nk_all * slice / S), nk = (int)((long long)nk_all * (slice + 1) / S);
    issue(kt0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (ABL != 1 && kt + 1 < nk) issue(kt + 1, cur ^ 1);
        const unsigned char* A = smem + cur * (A_BYTES + W_BYTES);
        const unsigned char* W = A + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[MF], fw[NF];
            const int kc = ks * 4 + fq;
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int rr = wrow_m + j * 16 + fr;
                if (ABL != 3) fa[j] = *(const bf16x8*)(A + rr * 128 + ((kc ^ (rr & 7)) << 4));
                else { u32x4 z = {(uint32_t)rr, 1u, 2u, 3u}; fa[j] = __builtin_bit_cast(bf16x8, z); }
            }
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int rr = wrow_n + i * 16 + fr;
                if (ABL != 3) {
                    if constexpr (!WT) fw[i] = *(const bf16x8*)(W + rr * 128 + ((kc ^ (rr & 7)) << 4));
                    else fw[i] = tr_frag(W, ks, lane, wrow_n + i * 16);
                }
                else { u32x4 z = {(uint32_t)rr, 5u, 6u, 7u}; fw[i] = __builtin_bit_cast(bf16x8, z); }
            }
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < MF; ++j)
                    if (ABL != 2) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
                    else { asm volatile("" :: "v"(fw[i]), "v"(fa[j])); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (S > 1) {                               // raw partial sums; the epilogue runs in splitk_reduce_kernel
        float* P = g.partial + (long long)slice * g.M * g.N;
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int m = m0 + wrow_m + j * 16 + fr;
            if (m >= M) continue;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int n = n0 + wrow_n + i * 16 + fq * 4;
                if (n < g.N) *(float4*)(P + (long long)m * g.N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        return;
    }
    if (ABL != 4) finish_tile<MF, NF>(g, acc, M, m0, n0, wrow_m, wrow_n, lane, wave, smem);
    else if (acc[0][0][0] == 12345.678f) finish_tile<MF, NF>(g, acc, M, m0, n0, wrow_m, wrow_n, lane, wave, smem);
}


// ---------------------------------------------------------------------------------------------
// Convolution on the LDS-DMA GEMM structure (round 3): the implicit-GEMM gather of igemm_kernel expressed as the per-lane offset of
// a `buffer_load ... lds`.  For layers whose input channels are a multiple of 64 a K step lies inside ONE 3 x 3 tap and ONE source,
// so everything that changes along K - the tap's pixel displacement, the channel base, the source of a two-source 1 x 1 - is
// wave-uniform and travels in the instruction's scalar offset; what a lane contributes is fixed for the whole tile: the byte
// offset of its output pixel's centre (+ its swizzled 16-byte chunk) and nine tap-validity bits (padding -> an out-of-range
// offset -> zeros through the descriptor's range check).  No staging registers, no address arithmetic in the loop, no LDS stores:
// the K step is the linear kernel's (two LDS stages, DMA of step k+1 under the MFMAs of step k).  igemm_kernel was bound by the
// VALU issue of exactly that gather and staging.  Epilogues: finish_tile (bias, SiLU, bf16 shortcut, f32 output).
// Eligibility (host): (c0 + c1) % 64 == 0, c1 == 0 or (1 x 1 and c0 % 64 == 0), Cout >= 64, staged epilogue usable.
// ---------------------------------------------------------------------------------------------
template <int BN, int WM, int WN, int ST = 2 /* LDS stages: ST - 1 K steps of DMA in flight */>
__global__ __launch_bounds__(256) void cgemm_dma_kernel(GemmArgs g) {
    constexpr int BM = 128, NW = 4;
    static_assert(WM * WN == NW, "four waves");
    constexpr int MF = BM / WM / 16, NF = BN / WN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    constexpr int A_INS = BM / (8 * NW), W_INS = BN / (8 * NW);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = g.M;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= M) return;

    const int lrow = lane >> 3, lch = lane & 7;
    constexpr uint32_t OOB = 0x80000000u;
    uint32_t a_off0[A_INS], a_off1[A_INS], a_taps[A_INS], w_off[W_INS];
    const int hw = g.Hout * g.Wout;
#pragma unroll
    for (int j = 0; j < A_INS; ++j) {
        const int r = (j * NW + wave) * 8 + lrow;
        const int m = m0 + r;
        const int mc = m < M ? m : M - 1;
        const int b = mc / hw, rem = mc - b * hw;
        const int oy = rem / g.Wout;
        const int cy = oy * g.stride, cx = (rem - oy * g.Wout) * g.stride;
        const uint32_t sw = (uint32_t)((lch ^ (r & 7)) << 4);
        a_off0[j] = (uint32_t)((((long long)b * (g.Hin >> g.up0) + (cy >> g.up0)) * (g.Win >> g.up0) + (cx >> g.up0)) * g.lda0 * 2) + sw;
        a_off1[j] = g.c1 ? (uint32_t)((((long long)b * (g.Hin >> g.up1) + (cy >> g.up1)) * (g.Win >> g.up1) + (cx >> g.up1)) * g.lda1 * 2) + sw : 0u;
        uint32_t taps = 0;
        if (m < M) {
            if (g.ksize == 3) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = cy + t / 3 - 1, ix = cx + t % 3 - 1;
                    taps |= (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) ? (1u << t) : 0u;
                }
            } else {
                taps = 1u;
            }
        }
        a_taps[j] = taps;
    }
#pragma unroll
    for (int j = 0; j < W_INS; ++j) {
        const int r = (j * NW + wave) * 8 + lrow;
        int n = n0 + r;
        n = n < g.N ? n : g.N - 1;
        w_off[j] = (uint32_t)(((long long)n * g.K + ((lch ^ (r & 7)) << 3)) * 2);
    }
    // the scalar offset is unsigned: source 0's descriptor starts one row + one pixel before the tensor (3 x 3 only)
    const uint32_t bias0 = g.ksize == 3 ? (uint32_t)((g.Win + 1) * g.lda0 * 2) : 0u;
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)g.a0 - bias0), 0, 0x7fffffff, 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(g.c1 ? g.a1 : g.a0), 0, 0x7fffffff, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)g.w, 0, 0x7fffffff, 0x00020000);
    const int Cin = g.c0 + g.c1;
    const int pad = g.ksize >> 1;
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        unsigned char* A = smem + buf * (A_BYTES + W_BYTES);
        unsigned char* W = A + A_BYTES;
        const int kb = kt * BK;                                  // (uniform)
        if (g.c1 && kb >= g.c0) {                                // second source of a two-source 1 x 1
            const int so = (kb - g.c0) * 2;
#pragma unroll
            for (int j = 0; j < A_INS; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr_t)(A + (j * NW + wave) * 1024), 16,
                                                         (int)((a_taps[j] & 1u) ? a_off1[j] : OOB), so, 0, 0);
        } else {
            int tap = 0;
            uint32_t so = bias0 + (uint32_t)(kb * 2);
            if (g.ksize == 3) {
                tap = g.cin_shift >= 0 ? (kb >> g.cin_shift) : kb / Cin;
                const int cin = kb - tap * Cin;
                const int ky = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0), kx = tap - ky * 3;
                so = bias0 + (uint32_t)((((ky - pad) * g.Win + (kx - pad)) * g.lda0 + cin) * 2);
            }
#pragma unroll
            for (int j = 0; j < A_INS; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lptr_t)(A + (j * NW + wave) * 1024), 16,
                                                         (int)(((a_taps[j] >> tap) & 1u) ? a_off0[j] : OOB), (int)so, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < W_INS; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(W + (j * NW + wave) * 1024), 16, (int)w_off[j], kb * 2, 0, 0);
    };

    f32x4 acc[NF][MF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wm = wave / WN, wn = wave - wm * WN;
    const int wrow_m = wm * (BM / WM), wrow_n = wn * (BN / WN);
    const int fr = lane & 15, fq = lane >> 4;
    const int nk = g.K / BK;
    // ring of ST stages, ST - 1 K steps in flight: the launches are latency-bound (9 .. 36 K steps per tile, one L2 / HBM round trip
    // each when only the next step is in flight), and inside the pipeline the detector runs on the few CUs the classifier leaves
    // free, where a tile's latency is all that counts.  Step kt: counted wait (the ST - 2 younger steps stay in flight), ONE raw
    // barrier (step kt visible to every wave; every wave is done with step kt-1, whose stage the next issue refills), issue, MFMAs.
    constexpr int INS = A_INS + W_INS;
    int issued = 0;
#pragma unroll
    for (int p = 0; p < ST - 1; ++p)
        if (p < nk) { issue(p, p); ++issued; }
    int slot = 0, islot = (ST - 1) % ST;
    for (int kt = 0; kt < nk; ++kt) {
        // steps issued so far: `issued` (all of them once the prologue / earlier iterations ran out of steps)
        if (issued - kt - 1 >= ST - 2 && ST > 2) {
            if constexpr (ST == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INS) : "memory");
            else if constexpr (ST == 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * INS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (issued < nk) { issue(issued, islot); ++issued; islot = islot + 1 == ST ? 0 : islot + 1; }
        const unsigned char* A = smem + slot * (A_BYTES + W_BYTES);
        const unsigned char* W = A + A_BYTES;
        slot = slot + 1 == ST ? 0 : slot + 1;
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
    }
    asm volatile("s_barrier" ::: "memory");                       // the staged epilogue reuses the tile buffers
    finish_tile<MF, NF>(g, acc, M, m0, n0, wrow_m, wrow_n, lane, wave, smem);
}

// ---------------------------------------------------------------------------------------------
// MXFP8 variant of the LDS-DMA GEMM (BASELINE.json configs[4]: FP8 classifier GEMMs).  Operands are OCP e4m3 bytes
// with one E8M0 scale per 32 consecutive K elements of a row (the OCP "MX" block format); the block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 applies both scales in hardware and runs at twice the bf16 MFMA rate.  A 128-byte LDS
// row holds 128 K elements (one MFMA K step), so tile shape, DMA, swizzle and epilogues are those of gemm_dma_kernel:
// the L2->LDS traffic per flop - what bounds the bf16 main loop - is halved.
//   operand layout (measured with yv_mx_probe, tests/test_gpu_fp8.py::test_mx_mfma_layout): lane l = (row l&15, group
//   g = l>>4) holds K elements 16g..16g+15 in its first 16 bytes and 64+16g..64+16g+15 in its second 16 bytes, i.e. the
//   16-byte chunks g and 4+g of the 128-byte row; the scale of the MX block k = 32j..32j+31 of that row is byte `opsel`
//   of the scale VGPR of lane (row, group j) - so lane (row, g) loads the scale of block g of the current K step.
// Scales are read with ordinary byte loads one K step ahead (issued behind the DMA of that step, consumed after the
// step's vmcnt(0)), so they never add a wait of their own.
// ---------------------------------------------------------------------------------------------
typedef int i32x8 __attribute__((ext_vector_type(8)));

struct MxArgs {
    GemmArgs g;                 // a0 / w point at the fp8 bytes (row strides lda0 / K bytes); epilogue fields as usual
    const uint8_t* sa;          // E8M0 scales, K-step-major: (K/128, rows_a, 4) - the 4 blocks of one 128-deep K step of a row
    const uint8_t* sw;          // (K/128, rows_w, 4)                               are one aligned dword
    long long rows_a, rows_w;   // row counts of the scale arrays (multiples of 128: a tile's 128 dwords are one DMA half)
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void gemm_mx_kernel(MxArgs a) {
    const GemmArgs& g = a.g;
    constexpr int NW = WM * WN, MF = BM / WM / 16, NF = BN / WN / 16, A_BYTES = BM * 128, W_BYTES = BN * 128;
    constexpr int A_INS = BM / (8 * NW), W_INS = BN / (8 * NW);
    constexpr int S_INS = (BM + BN + 255) / 256;              // wave-instructions that fetch the tile's scale dwords
    constexpr int S_BYTES = S_INS * 1024;                      // one scale dword per tile row and K step (+ slack of the last DMA)
    constexpr int STAGE = A_BYTES + W_BYTES + S_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int M = g.M;
    if (g.m_dev) { long long md = (long long)g.m_dev[0] * g.m_mul; M = md < M ? (int)md : M; }
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    int tm, tn;
    {
        const int GM = g.group_m, per = GM * g.tiles_n;
        const int grp = bid / per, first = grp * GM;
        const int gsz = (g.tiles_m - first) < GM ? (g.tiles_m - first) : GM;
        const int in = bid - grp * per;
        tm = first + in % gsz;
        tn = in / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= M) return;
    const int lrow = lane >> 3, lch = lane & 7;
    const uint8_t* a_src[A_INS];
    const uint8_t* w_src[W_INS];
#pragma unroll
    for (int j = 0; j < A_INS; ++j) {
        const int r = (j * NW + wave) * 8 + lrow;
        int m = m0 + r; m = m < g.M ? m : g.M - 1;
        a_src[j] = (const uint8_t*)g.a0 + (long long)m * g.lda0 + ((lch ^ (r & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < W_INS; ++j) {
        const int r = (j * NW + wave) * 8 + lrow;
        int n = n0 + r; n = n < g.N ? n : g.N - 1;
        w_src[j] = (const uint8_t*)g.w + (long long)n * g.K + ((lch ^ (r & 7)) << 4);
    }
    // scale DMA: the tile's BM A dwords then its BN W dwords, 4 dwords (16 bytes) per lane, lane-linear in LDS; the
    // (BM + BN) / 256 wave-instructions are dealt to waves 0, 1, ...
    const int s_dw = (wave * 64 + lane) * 4;                   // first dword this lane would fetch if its wave takes part
    const bool s_on = wave < S_INS && s_dw < BM + BN;
    const uint8_t* s_src = s_dw < BM ? a.sa + ((long long)m0 + s_dw) * 4 : a.sw + ((long long)n0 + (s_dw - BM)) * 4;
    const long long s_step = (s_dw < BM ? a.rows_a : a.rows_w) * 4;
    auto issue = [&](int kt, int buf) {
        unsigned char* A = smem + buf * STAGE;
        unsigned char* W = A + A_BYTES;
#pragma unroll
        for (int j = 0; j < A_INS; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[j] + kt * 128), (lptr_t)(A + (j * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < W_INS; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(w_src[j] + kt * 128), (lptr_t)(W + (j * NW + wave) * 1024), 16, 0, 0);
        if (wave < S_INS) {
            // lanes past the end of the scale block re-fetch its last 16 bytes into their (unused) slot: exec stays full
            const uint8_t* sp = s_on ? s_src + kt * s_step : a.sw + (long long)n0 * 4 + kt * a.rows_w * 4;
            __builtin_amdgcn_global_load_lds((gptr_t)sp, (lptr_t)(W + W_BYTES + wave * 1024), 16, 0, 0);
        }
    };
    const int wm = wave / WN, wn = wave - wm * WN;
    const int wrow_m = wm * (BM / WM), wrow_n = wn * (BN / WN);
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[NF][MF];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = g.K >> 7;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
        const unsigned char* A = smem + cur * STAGE;
        const unsigned char* W = A + A_BYTES;
        const unsigned char* S = W + W_BYTES;
        i32x8 fa[MF], fw[NF];
        int sca[MF], scw[NF];
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int rr = wrow_m + j * 16 + fr;
            const u32x4 lo = *(const u32x4*)(A + rr * 128 + (((fq) ^ (rr & 7)) << 4));
            const u32x4 hi = *(const u32x4*)(A + rr * 128 + (((4 + fq) ^ (rr & 7)) << 4));
            fa[j] = (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            sca[j] = (int)(*(const uint32_t*)(S + rr * 4) >> (8 * fq));              // byte 0 = scale of block fq
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int rr = wrow_n + i * 16 + fr;
            const u32x4 lo = *(const u32x4*)(W + rr * 128 + (((fq) ^ (rr & 7)) << 4));
            const u32x4 hi = *(const u32x4*)(W + rr * 128 + (((4 + fq) ^ (rr & 7)) << 4));
            fw[i] = (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            scw[i] = (int)(*(const uint32_t*)(S + BM * 4 + rr * 4) >> (8 * fq));
        }
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[i], fa[j], acc[i][j], 0, 0, 0, scw[i], 0, sca[j]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    finish_tile<MF, NF>(g, acc, M, m0, n0, wrow_m, wrow_n, lane, wave, smem);
}

// one MFMA on caller-provided register images (layout probe used while bringing the MX path up; tests keep it as the
// executable statement of the operand layout)
__global__ __launch_bounds__(64) void mx_probe_kernel(const i32x8* __restrict__ a, const i32x8* __restrict__ b,
                                                      const int* __restrict__ sa, const int* __restrict__ sb, int opsel,
                                                      f32x4* __restrict__ d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (opsel == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
    else if (opsel == 1) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 1, sa[l], 1, sb[l]);
    else if (opsel == 2) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 2, sa[l], 2, sb[l]);
    else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 3, sa[l], 3, sb[l]);
    d[l] = acc;
}

// x (rows, K) bf16 -> q (rows, K) e4m3 bytes + scales (rows, K/32) E8M0: scale exponent e = ceil(log2(amax / 448)) of the
// 32-element block (so that amax * 2^-e <= 448), all-zero blocks get e = -127; q = RNE_e4m3(x * 2^-e)
__global__ __launch_bounds__(256) void quant_mx_kernel(const uint16_t* __restrict__ x, long long ldx, long long rows, int K,
                                                       uint8_t* __restrict__ q, long long ldq, uint8_t* __restrict__ sc,
                                                       long long rows_pad) {
    const int kb = K >> 5;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * kb) return;
    const long long r = idx / kb;
    const int b = (int)(idx - r * kb);
    const uint16_t* src = x + r * ldx + b * 32;
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint4 u = *(const uint4*)(src + c * 8);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[c * 8 + 2 * i] = bf16_to_f32((uint16_t)(w[i] & 0xffff));
            v[c * 8 + 2 * i + 1] = bf16_to_f32((uint16_t)(w[i] >> 16));
        }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(v[i]));
    int e = -127;
    if (amax > 0.f) {
        int ex;
        const float mant = frexpf(amax * (1.0f / 448.0f), &ex);          // amax/448 = mant * 2^ex, mant in [0.5, 1)
        e = mant == 0.5f ? ex - 1 : ex;                                   // ceil(log2(amax/448))
        e = e < -127 ? -127 : (e > 127 ? 127 : e);
    }
    const float inv = ldexpf(1.0f, -e);
    uint32_t out[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int p = 0;
        p = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i] * inv, v[4 * i + 1] * inv, p, false);
        p = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i + 2] * inv, v[4 * i + 3] * inv, p, true);
        out[i] = (uint32_t)p;
    }
    uint4* dst = (uint4*)(q + r * ldq + b * 32);
    dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
    dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
    sc[((long long)(b >> 2) * rows_pad + r) * 4 + (b & 3)] = (uint8_t)(e + 127);      // (K/128, rows_pad, 4)
}

// ---------------------------------------------------------------------------------------------
// 256 x 256 x 64 "8-phase" schedule (cdna_hip_programming.md section 5 template, re-derived for this operand
// convention; the non-persistent round-1 kernel of this shape was removed in round 3, gemm_p8_kernel below is its
// persistent form).  8 waves = 2 groups (wm = 0/1, 128 activation rows each) x 4 (64 weight rows each); one
// workgroup per CU, 128 KB of LDS = 2 stages x {A tile, W tile}.  A K tile is consumed in 4 phases, one
// 64 x 32 quadrant of the wave tile each (16 MFMAs): (m0,n0) (m0,n1) (m1,n1) (m1,n0); every phase is
//     [ds_read this phase's operand half | issue ONE half-tile of LDS-DMA] barrier [16 MFMA] barrier
// and the two wave groups run ONE BARRIER APART, so that on every SIMD one wave is in its MFMA segment while
// its partner reads LDS / issues DMA.  The DMA stream runs 3 half-tiles ahead and is retired once per K
// tile by a COUNTED s_waitcnt vmcnt(6) (never 0 in the loop), raw s_barrier (a __syncthreads would drain).
//
// Half-tiles: A-half h = rows {wm*128 + h*64 + [0,64)} (both groups), W-half h = rows {wn*64 + h*32 + [0,32)}:
// each half holds what every wave reads in ONE phase, so it dies as a unit:
//     phase 1 reads A0,W0   phase 2 reads W1   phase 3 reads A1   phase 4 reads nothing  (W0,W1 stay in VGPRs)
//     restage (one phase after the last read; reads are retired by lgkmcnt(0) BEFORE the phase's first barrier):
//     phase 1: A1(t+1)   phase 2: A0(t+2)   phase 3: W0(t+2)   phase 4: W1(t+2), then vmcnt(6) = tile t+1 landed
// RAW: a half-tile is read at the earliest one phase after the wait that retires it (two barriers later, which
// covers the one-barrier stagger between the groups).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bar() { asm volatile("s_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------
// Persistent form of the 8-phase schedule (round 2).  One workgroup per CU walks its tiles; what changes against
// a one-tile-per-workgroup kernel:
//   * the LDS-DMA stream never drains between tiles: during the last two K tiles of a tile the restage slots load the FIRST two
//     K tiles of the workgroup's next tile, so the pipeline fill (one HBM/L2 round trip per tile) and the first waits are hidden
//     behind the epilogue, and the epilogue's stores drain in the shadow of the next main loop instead of in a burst;
//   * the epilogue does not alias the stage buffers (they are being refilled): every wave transposes 16 output rows at a time
//     through a private 2 KB slab; the bias vector sits in LDS for the whole launch (a global bias load inside the epilogue
//     would make hipcc drain the in-flight DMA with vmcnt(0));
//   * operands are addressed through buffer descriptors (32-bit offsets: half the address registers of flat pointers, which
//     pays for the second - next tile - offset set; rows past M read as zeros through the range check instead of a clamp);
//   * GELU by a 9-operation sigmoid form (see gelu_fast_f).
// LDS map: [stage 0 | stage 1] 2 x 64 KB, 8 slabs x 2 KB, bias 16 KB = 160 KB exactly.
// Restrictions (checked by the host, everything else takes the 128 x 128 kernel): N % 256 == 0, N <= 4096, K % 64 == 0,
// bf16 output, flags within {BIAS, GELU}, operand images below 2 GB.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_void_t;

// MF0 / MF1: 16-row activation fragments per wave group in the first / second half of its rows; tile = 32 (MF0 + MF1) rows x
// 256 columns.  (4,4) = 256 rows is the template of the guide; the smaller instances exist for tile-count quantisation: a
// persistent grid of 256 workgroups runs ceil(tiles / 256) rounds, and e.g. the 297 tiles of a 25,216 x 768 output cost two
// rounds at 256 rows but 474 tiles = 1.85 rounds of 160-row tiles (-37 %).  The host picks the instance that minimises
// rounds x rows.  A-operand DMA slots that a smaller tile does not need are issued with an out-of-range offset (the buffer
// range check turns them into no-ops) so that every wave keeps issuing the same number of DMA instructions per phase, which
// is what the counted vmcnt relies on.
template <int MF0, int MF1, bool F32OUT, int DIAG = 0 /* tools/gemm_lab.hip only: per-segment cycle sums into g.partial */>
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmArgs g) {
    constexpr int MF = MF0 + MF1, NF = 4;
    constexpr int RG = MF * 16, BM = 2 * RG;                   // rows per wave group / per tile
    constexpr int A_BYTES = 256 * 128, W_BYTES = 256 * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int SLAB0 = 2 * STAGE, SLAB = 2048, BIAS0 = SLAB0 + 8 * SLAB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int M = g.M;
    if (g.m_dev) { long long md = (long long)g.m_dev[0] * g.m_mul; M = md < M ? (int)md : M; }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = g.N >> 8;
    const int ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    // tile schedule (blockIdx & 7 = XCD under round-robin placement - speed only, never correctness): see GemmArgs::sched
    int Lx, seq0, seq1, lid;
    if (g.sched == 1) {
        const int nx = G < 8 ? G : 8;                           // (grids smaller than 8 blocks: one sequence share per block)
        const int xcd = (int)blockIdx.x % nx;
        lid = (int)blockIdx.x / nx;
        Lx = (G - xcd + nx - 1) / nx;                           // blocks that share this part of the sequence
        int cum = 0;                                            // blocks of the parts before this one
        for (int y = 0; y < xcd; ++y) cum += (G - y + nx - 1) / nx;
        // parts proportional to their block counts: with tiles == blocks every block gets exactly one tile (equal parts would
        // hand a 29-block XCD 30 tiles and double the launch time)
        seq0 = (int)((long long)ntiles * cum / G); seq1 = (int)((long long)ntiles * (cum + Lx) / G);
    } else {
        // round r covers sequence positions [r G, r G + G); inside a round XCD x walks a contiguous chunk of it
        const int q = G >> 3, r = G & 7, x = blockIdx.x & 7;
        lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + ((int)blockIdx.x >> 3);
        Lx = G; seq0 = 0; seq1 = ntiles;
    }
    if (seq0 + lid >= seq1) return;
    {   // bias -> LDS once (plain loads, before any DMA is in flight)
        float* bl = (float*)(smem + BIAS0);
        for (int i = tid; i < g.N; i += 512) bl[i] = (g.flags & YV_EPI_BIAS) ? g.bias[i] : 0.0f;
    }
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.a0, 0, (int)(((long long)(g.M - 1) * g.lda0 + g.K) * 2), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)g.w, 0, (int)((long long)g.N * g.K * 2), 0x00020000);

    const int wm = wave >> 2, wn = wave & 3;
    const int wrow_m = wm * RG, wrow_n = wn * 64;
    const int fr = lane & 15, fq = lane >> 4;
    const int lrow = lane >> 3, lch = lane & 7;
    auto coords = [&](int seq, int& m0, int& n0) __attribute__((always_inline)) {   // grouped order: GM consecutive M tiles share a W tile
        const int GM = g.group_m, per = GM * tiles_n;
        const int grp = seq / per, first = grp * GM;
        const int gsz = (tiles_m - first) < GM ? (tiles_m - first) : GM;
        const int in = seq - grp * per;
        m0 = (first + in % gsz) * BM;
        n0 = (in / gsz) << 8;
    };
    // A half h of a stage: rows {group * RG + off_h + [0, len_h)} of both groups = 2 * len_h / 8 pieces of 8 rows; piece slots
    // s = wave * 2 + j (16 per half); slots past the piece count are dummies
    auto a_piece_row = [&](int h, int s) __attribute__((always_inline)) -> int {      // first tile row of piece s, or -1
        const int len8 = (h == 0 ? MF0 : MF1) * 2;                                      // pieces per group
        if (s >= 2 * len8) return -1;
        const int grp = s / len8, r8 = s - grp * len8;
        return grp * RG + (h == 0 ? 0 : MF0 * 16) + r8 * 8;
    };
    auto set_offsets = [&](uint32_t (&o)[4][2], int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int s_ = wave * 2 + j;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r0 = a_piece_row(h, s_);
                const int ra = r0 + lrow, m = m0 + ra;
                o[h][j] = (r0 >= 0 && m < g.M) ? (uint32_t)(((long long)m * g.lda0 + ((lch ^ (ra & 7)) << 3)) * 2) : 0x80000000u;
                const int rh = s_ * 8 + lrow;
                const int rw = (rh >> 5) * 64 + h * 32 + (rh & 31);
                o[2 + h][j] = (uint32_t)(((long long)(n0 + rw) * g.K + ((lch ^ (rw & 7)) << 3)) * 2);
            }
        }
    };
    auto lds_dst = [&](int kind, int j) __attribute__((always_inline)) -> int {   // wave-uniform destination of a piece inside a stage
        const int s_ = wave * 2 + j;
        if (kind < 2) {
            const int r0 = a_piece_row(kind, s_);
            return (r0 >= 0 ? r0 : BM) * 128;                   // dummy pieces land in the unused rows BM.. of the A region
        }
        const int rb = s_ * 8;
        return A_BYTES + ((rb >> 5) * 64 + (kind - 2) * 32 + (rb & 31)) * 128;
    };

    const int nk = g.K / BK;
    uint32_t ocur[4][2], onxt[4][2];
    int seq = seq0 + lid, m0, n0, m0n = 0, n0n = 0;
    coords(seq, m0, n0);
    set_offsets(ocur, m0, n0);
    bool has_next = seq + Lx < seq1;
    if (has_next) { coords(seq + Lx, m0n, n0n); set_offsets(onxt, m0n, n0n); }
    int gk = 0;                                                // K tiles consumed so far by this workgroup (stage = gk & 1)

    // half-tile `kind` of K tile t of the CURRENT tile into stage `st` / of K tile tt of the NEXT tile (separate functions:
    // a run-time choice between the two offset sets makes hipcc index them through scratch memory, and scratch loads count
    // in vmcnt like the DMA does)
    auto issue_cur = [&](int kind, int t, int st) __attribute__((always_inline)) {
        unsigned char* base = smem + (st & 1) * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(kind < 2 ? rsA : rsW, (lds_void_t)(base + lds_dst(kind, j)), 16,
                                                     (int)ocur[kind][j], t * 128, 0, 0);
    };
    auto issue_nxt = [&](int kind, int tt, int st) __attribute__((always_inline)) {
        if (!has_next) return;
        unsigned char* base = smem + (st & 1) * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(kind < 2 ? rsA : rsW, (lds_void_t)(base + lds_dst(kind, j)), 16,
                                                     (int)onxt[kind][j], tt * 128, 0, 0);
    };

    f32x4 acc[NF][MF];
    bf16x8 fa[4][2], fw[4][2];
    auto read_a = [&](const unsigned char* A, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < (h == 0 ? MF0 : MF1); ++j) {
            const int rr = wrow_m + (h == 0 ? 0 : MF0 * 16) + j * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[j][ks] = *(const bf16x8*)(A + rr * 128 + (((ks * 4 + fq) ^ (rr & 7)) << 4));
        }
    };
    auto read_w = [&](const unsigned char* W, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rr = wrow_n + h * 32 + i * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fw[h * 2 + i][ks] = *(const bf16x8*)(W + rr * 128 + (((ks * 4 + fq) ^ (rr & 7)) << 4));
        }
    };
    auto mma = [&](int mh, int nh) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < (mh == 0 ? MF0 : MF1); ++j)
                    acc[nh * 2 + i][(mh == 0 ? 0 : MF0) + j] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nh * 2 + i][ks], fa[j][ks], acc[nh * 2 + i][(mh == 0 ? 0 : MF0) + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto sync_reads = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        bar();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto sync_mma = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        bar();
        __builtin_amdgcn_sched_barrier(0);
    };

    // DIAG build: cycle sums per segment of a phase (s_memtime stamps; a stamp is consumed one natural lgkmcnt(0) later, so
    // that reading it never adds a wait).  [0] read + DMA issue, [1] lgkmcnt wait of phases 1-3, [2] lgkmcnt + vmcnt wait of
    // phase 4, [3] first barrier, [4] MFMA segment, [5] second barrier, [6] epilogue, [7] phases
    uint32_t dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ds0 = 0, ds1 = 0, dp1 = 0, dp2 = 0, dp3 = 0, dp4 = 0, dp_is4 = 0;
    auto stamp = [&]() __attribute__((always_inline)) -> uint32_t { return (uint32_t)__builtin_amdgcn_s_memtime(); };
    auto dg_flush = [&]() __attribute__((always_inline)) {      // right after a natural lgkmcnt(0): every older stamp has landed
        if constexpr (DIAG) {
            if (dg[7]) {
                if (dp_is4) dg[2] += dp2 - dp1; else dg[1] += dp2 - dp1;
                dg[3] += dp3 - dp2; dg[4] += dp4 - dp3; dg[5] += ds0 - dp4;
            }
            dg[0] += ds1 - ds0; dg[7] += 1; dp1 = ds1;
        }
    };
    __syncthreads();                                           // bias image complete (no DMA in flight yet: a plain barrier)
    // ---- prologue of the first tile: K tile 0 complete, first three half-tiles of K tile 1 in flight (nk >= 2) ----------
    issue_cur(0, 0, 0); issue_cur(2, 0, 0); issue_cur(3, 0, 0); issue_cur(1, 0, 0);
    issue_cur(0, 1, 1); issue_cur(2, 1, 1); issue_cur(3, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    bar();
    if (wm == 1) bar();                                        // group 1 runs one barrier behind group 0

    // one K tile = 4 phases.  TAIL 0: K tiles t+1, t+2 belong to this tile; 1: t = nk-2 (t+2 is K tile 0 of the next tile);
    // 2: t = nk-1 (t+1, t+2 are K tiles 0, 1 of the next tile).  Stage of a K tile = parity of the running counter gk.
    auto ktile = [&](int t, auto tail_c) __attribute__((always_inline)) {
        constexpr int TAIL = decltype(tail_c)::value;
        const unsigned char* A = smem + (gk & 1) * STAGE;
        const unsigned char* W = A + A_BYTES;
        auto seg_reads = [&](int is4) __attribute__((always_inline)) {     // end of a read segment (DIAG: stamped)
            if constexpr (DIAG) {
                if (!is4) ds1 = stamp();                          // phase 4 stamps in front of its vmcnt wait
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                dg_flush(); dp_is4 = is4;
                dp2 = stamp();
                __builtin_amdgcn_sched_barrier(0);
                bar();
                __builtin_amdgcn_sched_barrier(0);
                dp3 = stamp();
            } else sync_reads();
        };
        auto seg_mma = [&]() __attribute__((always_inline)) {
            if constexpr (DIAG) {
                __builtin_amdgcn_sched_barrier(0);
                dp4 = stamp();
                bar();
                __builtin_amdgcn_sched_barrier(0);
                ds0 = stamp();
            } else sync_mma();
        };
        read_a(A, 0); read_w(W, 0);
        if constexpr (TAIL == 2) issue_nxt(1, 0, gk + 1); else issue_cur(1, t + 1, gk + 1);
        seg_reads(0);
        mma(0, 0);
        seg_mma();
        read_w(W, 1);
        if constexpr (TAIL == 0) issue_cur(0, t + 2, gk); else issue_nxt(0, TAIL - 1, gk);
        seg_reads(0);
        mma(0, 1);
        seg_mma();
        read_a(A, 1);
        if constexpr (TAIL == 0) issue_cur(2, t + 2, gk); else issue_nxt(2, TAIL - 1, gk);
        seg_reads(0);
        mma(1, 1);
        seg_mma();
        if constexpr (TAIL == 0) issue_cur(3, t + 2, gk); else issue_nxt(3, TAIL - 1, gk);
        if constexpr (DIAG) ds1 = stamp();
        if (TAIL == 0 || has_next) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // three half-tiles stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        seg_reads(1);
        mma(1, 0);
        seg_mma();
        ++gk;
    };

    uint32_t de0_ = 0;
    const uint32_t dk0 = DIAG ? stamp() : 0;
    if constexpr (DIAG) ds0 = dk0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < nk - 2; ++t) ktile(t, std::integral_constant<int, 0>{});
        ktile(nk - 2, std::integral_constant<int, 1>{});
        ktile(nk - 1, std::integral_constant<int, 2>{});
        // ---- epilogue: 16 rows at a time through this wave's slab; no workgroup barrier (the groups stay one barrier apart) ----
        {
            if constexpr (DIAG) de0_ = stamp();
            unsigned char* slab = smem + SLAB0 + wave * SLAB;
            const float* bl = (const float*)(smem + BIAS0) + n0 + wrow_n + fq * 4;
            if constexpr (!F32OUT) {
                const bool gelu = g.flags & YV_EPI_GELU;
                uint16_t* outp = (uint16_t*)g.out;
#pragma unroll
                for (int j = 0; j < MF; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 bvi = *(const float4*)(bl + i * 16);
                        float v0 = acc[i][j][0] + bvi.x, v1 = acc[i][j][1] + bvi.y;
                        float v2 = acc[i][j][2] + bvi.z, v3 = acc[i][j][3] + bvi.w;
                        if (gelu) { v0 = gelu_f(v0); v1 = gelu_f(v1); v2 = gelu_f(v2); v3 = gelu_f(v3); }
                        const int c16 = i * 2 + (fq >> 1);
                        *(uint2*)(slab + fr * 128 + ((c16 ^ (fr & 7)) << 4) + (fq & 1) * 8) =
                            make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slab is wave-private: LDS executes a wave's ops in order
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int row = it * 8 + (lane >> 3), ch = lane & 7;
                        const int m = m0 + wrow_m + j * 16 + row;
                        const uint4 pk = *(const uint4*)(slab + row * 128 + ((ch ^ (row & 7)) << 4));
                        if (m < M) *(uint4*)(outp + (long long)m * g.ldo + n0 + wrow_n + ch * 8) = pk;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads retired before the next chunk overwrites the slab
                }
            } else {
                // f32 output / residual stream (x += A . W^T + b): 16 rows x 32 columns per pass (128-byte row segments).
                // The residual values are fetched for a half / quarter of the wave tile at once, in the row-segment layout of the
                // stores (8-12 independent 16-byte loads per lane in flight): fetched chunk by chunk, each of the 16 passes of a tile
                // waited for its own HBM round trip (proj: 67 us for a 39 us memory floor)
                // Loads and stores go through a buffer descriptor over the output with an out-of-range offset for rows past M
                // (reads return 0, writes are dropped) instead of `if (m < M)`: inside a branch hipcc cannot count the memory
                // operations in flight and waits vmcnt(0) before every use of a fetched residual value - and vmcnt counts STORES
                // on this chip, so each 16-row pass waited for the previous pass's stores to be acknowledged (proj 57 us / fc2
                // 131 us against 40 / 118 us with a plain bf16 epilogue).
                const bool rmw = g.flags & YV_EPI_RES_F32;
                const auto rsO = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)(((long long)(M - 1) * g.ldo + g.N) * 4), 0x00020000);
                constexpr int NPART = MF <= 5 ? 2 : 4;             // residual registers in flight: 16 * JA (the accumulators hold 16 * MF)
                constexpr int JA = (MF + NPART - 1) / NPART;
                auto out_off = [&](int j, int ip, int it) __attribute__((always_inline)) {
                    const int row = it * 8 + (lane >> 3), ch = lane & 7;
                    const int m = m0 + wrow_m + j * 16 + row;
                    return m < M ? (uint32_t)((m * g.ldo + n0 + wrow_n + ip * 32 + ch * 4) * 4) : 0x80000000u;
                };
#pragma unroll
                for (int half = 0; half < NPART; ++half) {
                    const int j0 = half * JA, jn = (MF - j0) < JA ? (MF - j0 > 0 ? MF - j0 : 0) : JA;
                    u32x4 xr[JA][2][2];
                    if (rmw) {
#pragma unroll
                        for (int jj = 0; jj < JA; ++jj)
#pragma unroll
                            for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                                for (int it = 0; it < 2; ++it)
                                    xr[jj][ip][it] = __builtin_amdgcn_raw_buffer_load_b128(rsO, jj < jn ? out_off(j0 + jj, ip, it) : 0x80000000u, 0, 0);
                    }
#pragma unroll
                    for (int jj = 0; jj < JA; ++jj) {
                        if (jj >= jn) continue;
                        const int j = j0 + jj;
#pragma unroll
                        for (int ip = 0; ip < 2; ++ip) {
#pragma unroll
                            for (int ii = 0; ii < 2; ++ii) {
                                const int i = ip * 2 + ii;
                                const float4 bvi = *(const float4*)(bl + i * 16);
                                *(float4*)(slab + fr * 128 + (((ii * 4 + fq) ^ (fr & 7)) << 4)) =
                                    make_float4(acc[i][j][0] + bvi.x, acc[i][j][1] + bvi.y, acc[i][j][2] + bvi.z, acc[i][j][3] + bvi.w);
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                            for (int it = 0; it < 2; ++it) {
                                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                                float4 v = *(const float4*)(slab + row * 128 + ((ch ^ (row & 7)) << 4));
                                if (rmw) {
                                    const u32x4 x = xr[jj][ip][it];
                                    v.x += __uint_as_float(x[0]); v.y += __uint_as_float(x[1]);
                                    v.z += __uint_as_float(x[2]); v.w += __uint_as_float(x[3]);
                                }
                                const u32x4 pk = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                                __builtin_amdgcn_raw_buffer_store_b128(pk, rsO, out_off(j, ip, it), 0, 0);
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
                    }
                }
            }
        }
        if constexpr (DIAG) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t de1 = stamp();
            dg[6] += de1 - de0_; ds0 = de1;
        }
        if (!has_next) break;
        seq += Lx;
        m0 = m0n; n0 = n0n;
#pragma unroll
        for (int k = 0; k < 4; ++k) { ocur[k][0] = onxt[k][0]; ocur[k][1] = onxt[k][1]; }
        has_next = seq + Lx < seq1;
        if (has_next) { coords(seq + Lx, m0n, n0n); set_offsets(onxt, m0n, n0n); }
    }
    if (wm == 0) bar();                                        // group 0 waits for group 1's last barrier
    if constexpr (DIAG) {
        const uint32_t dk1 = stamp();
        if (lane == 0) {
            uint32_t* o = (uint32_t*)g.partial + ((long long)blockIdx.x * 8 + wave) * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = dg[i];
            o[8] = dk1 - dk0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <int MF0, int MF1, bool F32OUT>
int launch_p8_inst2(GemmArgs& g, hipStream_t st, int n_cu) {
    constexpr int BM = 32 * (MF0 + MF1);
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = g.N / 256;
    const size_t lds = 2 * 65536 + 8 * 2048 + 16384;
    auto kern = gemm_p8_kernel<MF0, MF1, F32OUT>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return YV_ERR_LAUNCH;
    const int tiles = g.tiles_m * g.tiles_n;
    const int grid = tiles < n_cu ? tiles : n_cu;
    if (t_time_start || t_time_stop) {
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(512), (uint32_t)lds, st, t_time_start, t_time_stop, 0, g);
        t_time_start = t_time_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, g);
    }
    return yv_launch_status();
}

template <int MF0, int MF1>
int launch_p8_inst(GemmArgs& g, hipStream_t st, int n_cu) {
    // the f32 epilogue keeps a residual prefetch next to the accumulators: only the tiles up to 192 rows have the registers for it
    if constexpr (MF0 + MF1 <= 6) {
        if (g.flags & (YV_EPI_RES_F32 | YV_EPI_OUT_F32)) return launch_p8_inst2<MF0, MF1, true>(g, st, n_cu);
    }
    return launch_p8_inst2<MF0, MF1, false>(g, st, n_cu);
}

int g_opt_p8_sched = 1;            // tile schedule of the persistent kernel (GemmArgs::sched): "linear_p8_sched"
thread_local int g_opt_p8_cus = 0; // persistent grid size OF LAUNCHES MADE BY THIS THREAD; 0 = every CU ("linear_p8_cus": leave CUs to concurrent streams)
int g_opt_p8_rows = 0;             // 0 = pick the tile height per launch; 128 / 160 / 192 / 224 / 256 force it ("linear_p8_rows")

int launch_p8(GemmArgs& g, hipStream_t st) {
    static int n_cu_dev = 0;                                    // CU count of the device (same value from every thread)
    if (!n_cu_dev) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return YV_ERR_LAUNCH;
        n_cu_dev = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int n_cu = (g_opt_p8_cus > 0 && g_opt_p8_cus < n_cu_dev) ? g_opt_p8_cus : n_cu_dev;
    g.sched = g_opt_p8_sched;
    // tile height: minimise rounds x (rows + a fixed per-tile cost worth ~24 rows: epilogue, pipeline turn-around)
    int best = 256;
    if (g_opt_p8_rows) {
        best = g_opt_p8_rows < 128 ? 128 : g_opt_p8_rows;          // (96: a tile height of the free-running kernel only)
        if ((g.flags & (YV_EPI_RES_F32 | YV_EPI_OUT_F32)) && best > 192) best = 192;
    } else {
        long long best_cost = -1;
        const int cand[5] = {256, 224, 192, 160, 128};
        const bool f32out = g.flags & (YV_EPI_RES_F32 | YV_EPI_OUT_F32);
        best = f32out ? 192 : 256;
        for (int c = f32out ? 2 : 0; c < 5; ++c) {
            const long long tiles = (long long)((g.M + cand[c] - 1) / cand[c]) * (g.N / 256);
            const long long rounds = (tiles + n_cu - 1) / n_cu;
            const long long cost = rounds * (cand[c] + 24);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cand[c]; }
        }
    }
    switch (best) {
        case 224: return launch_p8_inst<4, 3>(g, st, n_cu);
        case 192: return launch_p8_inst<3, 3>(g, st, n_cu);
        case 160: return launch_p8_inst<3, 2>(g, st, n_cu);
        case 128: return launch_p8_inst<2, 2>(g, st, n_cu);
        default: return launch_p8_inst<4, 4>(g, st, n_cu);
    }
}

// ---------------------------------------------------------------------------------------------
// gemm_p9_kernel (round 3): the persistent tile walk, LDS map and LDS-DMA addressing of gemm_p8_kernel with a FREE-RUNNING main
// loop.  What the per-segment stamps of the DIAG build showed for gemm_p8_kernel (tools/gemm_lab.hip): a phase costs
// ~890 cycles where its two MFMA segments need 512; the read segment is not LDS latency but the ISSUE of the two LDS-DMA
// instructions (~85 cycles each while the four waves of a group issue theirs at once; the waits on landing DMA are ~25 cycles
// per K tile), and each of the 8 barriers of a K tile costs the last arriver ~65 cycles.  So here:
//   * ONE barrier per K tile instead of eight.  A wave's program for a K tile is P = ceil(MF / 2) phases of 16 MFMAs (two row
//     fragments x four column fragments x two 32-deep steps); the fragment reads of phase p+1 are issued at the top of phase p into
//     the other half of a two-deep register ring (16 registers each), the weight fragments of K tile t+1 replace those of K tile t
//     in place inside the last phase (after their last use), so LDS latency is covered by the wave's OWN MFMAs and the two waves of
//     a SIMD are not forced to alternate: whichever has operands issues, and the waves of a workgroup drift apart instead of
//     bursting on the LDS-DMA path together;
//   * the sync point S (end of phase P-2: vmcnt(0) + lgkmcnt(0) + barrier) retires K tile t+1 and frees the whole stage of K tile
//     t at once (its last fragment reads were issued one phase earlier); the 8 DMA instructions per wave and K tile are spread over
//     the phases that follow S (activation pieces first: they can miss L2; weight pieces last: they never do; none in the phase
//     that ends in the next S);
//   * LDS-free epilogue.  bf16 outputs: the weight rows of a wave's 64 columns are PERMUTED on the DMA source side so that MFMA
//     fragment i, row r holds column (r >> 2) * 16 + i * 4 + (r & 3): a lane's 16 accumulator values of one output row are 16
//     consecutive columns = two 16-byte stores straight from registers (64-byte row segments per wave-instruction).  f32 outputs
//     keep the plain order (a lane's four values per fragment are 16 bytes, a fragment's 16 columns one 64-byte segment).  The
//     slabs, their lgkmcnt round trips (5.5 k cycles per 256 x 256 tile) and 16 KB of LDS are gone.
// Instances: MF (16-row fragments per wave group; tile = 32 MF rows x 256 columns) in 5..8; K / 64 >= 2, even when P is odd.
// ---------------------------------------------------------------------------------------------
template <int V> using ic = std::integral_constant<int, V>;
#ifndef YV_P9_STORE_AUX
#define YV_P9_STORE_AUX 0              // cache policy bits of the bf16 output stores (2 = nt: streaming; experiment builds only)
#endif

template <int MF, bool F32OUT, int DIAG = 0 /* tools/gemm_lab.hip only: cycle sums into g.partial */,
          int EXT = 0 /* trainer epilogues of the bf16 output: 1 = YV_EPI_SAVE_PRE (fc1 forward), 2 = YV_EPI_GELU_BWD (fc2 data gradient) */,
          bool MX = false /* OCP MXFP8 operands (e4m3 bytes + one E8M0 scale per 32 K): a 128-byte LDS row is 128 K elements = ONE
                             block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) per fragment pair and K tile; same DMA, LDS images,
                             fragment reads and schedule, twice the flops per K tile; the two 1 KB scale rows of a K tile travel with
                             its activation / weight pieces (issued by waves 0 / 1) */>
__global__ __launch_bounds__(512) void gemm_p9_kernel(GemmArgs g) {
    constexpr int NF = 4, P = (MF + 1) / 2;
    static_assert(!MX || EXT == 0, "MX: no trainer epilogues");
    static_assert(EXT == 0 || (!F32OUT && MF <= 7), "aux epilogues: bf16 output, tiles of up to 224 rows (registers)");
    constexpr int MF0 = (MF + 1) / 2, MF1 = MF - MF0;          // DMA halves of the activation rows of a group (piece bookkeeping of p8)
    constexpr int RG = MF * 16, BM = 2 * RG;
    constexpr int A_BYTES = 256 * 128, SC0 = 2 * A_BYTES, STAGE = 2 * A_BYTES + (MX ? 2048 : 0), BIAS0 = 2 * STAGE;
    constexpr int ESZ = MX ? 1 : 2;                               // bytes per operand element; K elements per 128-byte row: 128 / ESZ
    constexpr int KT = 128 / ESZ;
    constexpr bool PERM = !F32OUT;
    static_assert(MF >= 3 && MF <= 8, "tile heights 96..256");
    static_assert(MF >= 5 || (!MX && EXT == 0), "96 / 128-row tiles: plain bf16 operands only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int M = g.M;
    if (g.m_dev) { long long md = (long long)g.m_dev[0] * g.m_mul; M = md < M ? (int)md : M; }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = g.N >> 8;
    const int ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    int Lx, seq0, seq1, lid;
    if (g.sched == 1) {
        const int nx = G < 8 ? G : 8;
        const int xcd = (int)blockIdx.x % nx;
        lid = (int)blockIdx.x / nx;
        Lx = (G - xcd + nx - 1) / nx;
        int cum = 0;
        for (int y = 0; y < xcd; ++y) cum += (G - y + nx - 1) / nx;
        seq0 = (int)((long long)ntiles * cum / G); seq1 = (int)((long long)ntiles * (cum + Lx) / G);
    } else {
        const int q = G >> 3, r = G & 7, x = blockIdx.x & 7;
        lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + ((int)blockIdx.x >> 3);
        Lx = G; seq0 = 0; seq1 = ntiles;
    }
    if (seq0 + lid >= seq1) return;
    {
        float* bl = (float*)(smem + BIAS0);
        for (int i = tid; i < g.N; i += 512) bl[i] = (g.flags & YV_EPI_BIAS) ? g.bias[i] : 0.0f;
    }
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.a0, 0, (int)(((long long)(g.M - 1) * g.lda0 + g.K) * ESZ), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)g.w, 0, (int)((long long)g.N * g.K * ESZ), 0x00020000);
    const int nk_all = g.K / KT;
    const auto rsSA = __builtin_amdgcn_make_buffer_rsrc((void*)(MX ? (const void*)g.mx_sa : (const void*)g.a0), 0,
                                                        MX ? (int)((long long)nk_all * g.mx_rows_a * 4) : 16, 0x00020000);
    const auto rsSW = __builtin_amdgcn_make_buffer_rsrc((void*)(MX ? (const void*)g.mx_sw : (const void*)g.w), 0,
                                                        MX ? (int)((long long)nk_all * g.mx_rows_w * 4) : 16, 0x00020000);
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)(((long long)(M - 1) * g.ldo + g.N) * (F32OUT ? 4 : 2)), 0x00020000);
    // f32 residual read from another tensor of the output's layout (trainer: x_mid = x_in + ...), else read-modify-write of `out`
    const auto rsR = __builtin_amdgcn_make_buffer_rsrc(F32OUT && g.resf ? (void*)g.resf : g.out, 0,
                                                       (int)(((long long)(M - 1) * g.ldo + g.N) * (F32OUT ? 4 : 2)), 0x00020000);
    // bf16 side tensor of the trainer epilogues (EXT): pre-activation, written (SAVE_PRE) or read (GELU_BWD)
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(EXT ? (void*)g.aux : g.out, 0, (int)(((long long)(M - 1) * (EXT ? g.ldaux : g.ldo) + g.N) * 2), 0x00020000);

    const int wm = wave >> 2, wn = wave & 3;
    const int wrow_m = wm * RG, wrow_n = wn * 64;
    const int fr = lane & 15, fq = lane >> 4;
    const int lrow = lane >> 3, lch = lane & 7;
    auto coords = [&](int seq, int& m0, int& n0) __attribute__((always_inline)) {
        const int GM = g.group_m, per = GM * tiles_n;
        const int grp = seq / per, first = grp * GM;
        const int gsz = (tiles_m - first) < GM ? (tiles_m - first) : GM;
        const int in = seq - grp * per;
        m0 = (first + in % gsz) * BM;
        n0 = (in / gsz) << 8;
    };
    auto a_piece_row = [&](int h, int s) __attribute__((always_inline)) -> int {
        const int len8 = (h == 0 ? MF0 : MF1) * 2;
        if (s >= 2 * len8) return -1;
        const int grp = s / len8, r8 = s - grp * len8;
        return grp * RG + (h == 0 ? 0 : MF0 * 16) + r8 * 8;
    };
    // DMA source offsets.  Activation pieces: per-lane byte offset of (tile row, swizzled chunk) - ONE set, pointed at the next
    // tile from K tile nk-2 on (the current tile's last activation pieces are issued in K tile nk-3); rows past M get an
    // out-of-range offset (the range check returns zeros).  Weight pieces: a lane part that never changes + the tile's n0 * K
    // in the instruction's scalar offset.
    auto set_a_offsets = [&](uint32_t (&o)[2][2], int m0, bool valid) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int s_ = wave * 2 + j;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r0 = a_piece_row(h, s_);
                const int ra = r0 + lrow, m = m0 + ra;
                o[h][j] = (valid && r0 >= 0 && m < g.M) ? (uint32_t)((long long)m * g.lda0 * ESZ + ((lch ^ (ra & 7)) << 4)) : 0x80000000u;
            }
        }
    };
    uint32_t ow[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rh = (wave * 2 + j) * 8 + lrow;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rw = (rh >> 5) * 64 + h * 32 + (rh & 31);          // LDS row of the W tile
            // PERM: LDS row (block b, fragment i, row r) holds weight row b * 64 + (r >> 2) * 16 + i * 4 + (r & 3)
            const int rsrc = PERM ? ((rw & ~63) | (((rw & 15) >> 2) << 4) | (((rw >> 4) & 3) << 2) | (rw & 3)) : rw;
            ow[h][j] = (uint32_t)(rsrc * g.K * ESZ + ((lch ^ (rw & 7)) << 4));
        }
    }
    auto lds_dst = [&](int kind, int j) __attribute__((always_inline)) -> int {
        const int s_ = wave * 2 + j;
        if (kind < 2) {
            const int r0 = a_piece_row(kind, s_);
            return (r0 >= 0 ? r0 : BM) * 128;
        }
        const int rb = s_ * 8;
        return A_BYTES + ((rb >> 5) * 64 + (kind - 2) * 32 + (rb & 31)) * 128;
    };

    const int nk = g.K / KT;
    int sa_row0 = 0;                                             // MX: first activation row of the tile the `oa` offsets point at
    uint32_t oa[2][2];
    int seq = seq0 + lid, m0, n0, m0n = 0, n0n = 0;
    coords(seq, m0, n0);
    set_a_offsets(oa, m0, true);
    sa_row0 = m0;
    bool has_next = seq + Lx < seq1;
    if (has_next) coords(seq + Lx, m0n, n0n);
    int gk = 0;                                                // K tiles consumed so far (stage of a K tile = parity)

    // piece `kind` (0, 1 activation halves; 2, 3 weight halves) of K tile k of the tile whose column origin is nb
    // MX: the K tile's 256 activation-row scale dwords (wave 0, with activation half 0) / weight-row scale dwords (wave 1, with weight
    // half 0): 1 KB each = one wave instruction; rows past the scale array read zeros (those rows are never stored)
    auto issue_a = [&](int kind, int k, int st) __attribute__((always_inline)) {
        unsigned char* base = smem + (st & 1) * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t)(base + lds_dst(kind, j)), 16, (int)oa[kind][j], k * 128, 0, 0);
        if constexpr (MX) {
            if (kind == 0 && wave == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsSA, (lds_void_t)(base + SC0), 16, lane * 16,
                                                         (int)(((long long)k * g.mx_rows_a + sa_row0) * 4), 0, 0);
        }
    };
    auto issue_w = [&](int kind, int k, int st, int nb) __attribute__((always_inline)) {
        unsigned char* base = smem + (st & 1) * STAGE;
        const int so = nb * g.K * ESZ + k * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_t)(base + lds_dst(kind, j)), 16, (int)ow[kind - 2][j], so, 0, 0);
        if constexpr (MX) {
            if (kind == 2 && wave == 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsSW, (lds_void_t)(base + SC0 + 1024), 16, lane * 16,
                                                         (int)(((long long)k * g.mx_rows_w + nb) * 4), 0, 0);
        }
    };
    // K tile t + d (d in {1, 2}) of this tile or, past its end, of the next tile
    auto issue_rel = [&](int kind, int t, int d, auto tail_c) __attribute__((always_inline)) {
        constexpr int TAIL = decltype(tail_c)::value;             // 0: t <= nk-3, 1: t = nk-2, 2: t = nk-1
        const int st = gk + d;
        if (kind < 2) issue_a(kind, TAIL + d <= 2 ? t + d : TAIL + d - 3, st);       // `oa` already points at the right tile
        else if (TAIL + d <= 2) issue_w(kind, t + d, st, n0);
        else if (has_next) issue_w(kind, TAIL + d - 3, st, n0n);
    };

    f32x4 acc[NF][MF];
    bf16x8 fa[2][2][2];                                         // [ring half][row fragment of the pair][k step]
    bf16x8 fw[4][2];                                            // [column fragment][k step] of the current K tile
    // MX: a fragment is the 8-register operand of the 128-deep MFMA (chunks fq and 4 + fq of the row), built where it is read
    i32x8 fa8[2][2], fw8[4];
    int sca[2][2], scw[4];                                       // this lane's block scale (byte 0) per row fragment
    auto read_pair = [&](auto half_c, const unsigned char* A, int pr) __attribute__((always_inline)) {
        constexpr int HALF = decltype(half_c)::value;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (2 * pr + jj >= MF) continue;
            const int rr = wrow_m + (2 * pr + jj) * 16 + fr;
            if constexpr (MX) {
                const u32x4 lo = *(const u32x4*)(A + rr * 128 + ((fq ^ (rr & 7)) << 4));
                const u32x4 hi = *(const u32x4*)(A + rr * 128 + (((4 + fq) ^ (rr & 7)) << 4));
                fa8[HALF][jj] = (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                sca[HALF][jj] = (int)(*(const uint32_t*)(A + SC0 + rr * 4) >> (8 * fq));
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fa[HALF][jj][ks] = *(const bf16x8*)(A + rr * 128 + (((ks * 4 + fq) ^ (rr & 7)) << 4));
            }
        }
    };
    auto read_w = [&](const unsigned char* W, int ks) __attribute__((always_inline)) {   // W = stage base + A_BYTES
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = wrow_n + i * 16 + fr;
            if constexpr (MX) {
                if (ks == 0) {
                    const u32x4 lo = *(const u32x4*)(W + rr * 128 + ((fq ^ (rr & 7)) << 4));
                    const u32x4 hi = *(const u32x4*)(W + rr * 128 + (((4 + fq) ^ (rr & 7)) << 4));
                    fw8[i] = (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                    // LDS row rr holds weight row (rr & ~63) | ((rr & 15) >> 2) << 4 | ((rr >> 4) & 3) << 2 | (rr & 3) when PERM
                    const int rs_ = PERM ? ((rr & ~63) | (((rr & 15) >> 2) << 4) | (((rr >> 4) & 3) << 2) | (rr & 3)) : rr;
                    scw[i] = (int)(*(const uint32_t*)(W + (SC0 - A_BYTES) + 1024 + rs_ * 4) >> (8 * fq));
                }
            } else {
                fw[i][ks] = *(const bf16x8*)(W + rr * 128 + (((ks * 4 + fq) ^ (rr & 7)) << 4));
            }
        }
    };
    auto mma = [&](auto half_c, int pr, int ks) __attribute__((always_inline)) {
        constexpr int HALF = decltype(half_c)::value;
        if constexpr (MX) { if (ks == 0) return; }                // one 128-deep MFMA per fragment pair: issued in the second slot
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                if (2 * pr + jj >= MF) continue;
                if constexpr (MX) {
                    // inline asm: around the builtin hipcc's register allocation needs ~100 more VGPRs (every instance spilled 120-360
                    // registers into the K loop; with bf16 MFMAs in its place none did).  Operands come from LDS reads (waited for by
                    // the compiler, which sees them as inputs) and a shift issued a phase earlier: no hazard window inside the string
                    // beyond the s_nop; the accumulator chains MFMA -> MFMA (no wait states) and is next read in the epilogue.
                    asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"
                                 : "+v"(acc[i][2 * pr + jj]) : "v"(fw8[i]), "v"(fa8[HALF][jj]), "v"(scw[i]), "v"(sca[HALF][jj]));
                } else {
                    acc[i][2 * pr + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i][ks], fa[HALF][jj][ks], acc[i][2 * pr + jj], 0, 0, 0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
    };
    uint32_t dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dtb = 0, dtc = 0;   // DIAG: [0] sync waits [1] barrier waits [2] main loops [3] epilogues
    auto stamp = [&]() __attribute__((always_inline)) -> uint32_t { return (uint32_t)__builtin_amdgcn_s_memtime(); };   // [4] first sync of a tile [5] tiles [6] syncs
    // DMA pieces of the window that follows a sync point: phase P-1 (of the K tile of the sync point) takes both activation halves
    // of K tile t+2, the first phases of the next K tile the weight halves of (its) K tile t+1; phase P-2 - the one that ends in
    // the next sync point - issues nothing, so the youngest piece has a whole phase to land before it is waited for
    auto dma_for_phase = [&](int p, int t, auto tail_c) __attribute__((always_inline)) {
        if constexpr (P == 2) {
            // 96 / 128-row tiles: the phase that follows the sync point is also the only one that does not end in the next: all eight
            // pieces of K tile t+2 go here (its stage - K tile t's - is free: the weight fragments of K tile t were read a K tile ago)
            if (p == 1) { issue_rel(0, t, 2, tail_c); issue_rel(1, t, 2, tail_c); issue_rel(2, t, 2, tail_c); issue_rel(3, t, 2, tail_c); }
            return;
        }
        if (p == P - 1) { issue_rel(0, t, 2, tail_c); issue_rel(1, t, 2, tail_c); return; }
        if constexpr (P == 4) { if (p < 2) issue_rel(2 + p, t, 1, tail_c); }
        else { if (p == 0) { issue_rel(2, t, 1, tail_c); issue_rel(3, t, 1, tail_c); } }
    };
    // one K tile.  PAR: ring half that holds row pair 0 of this K tile (odd P: alternates).  LASTK: t = nk - 1
    auto ktile = [&](int t, auto tail_c, auto par_c) __attribute__((always_inline)) {
        constexpr int TAIL = decltype(tail_c)::value, PAR = decltype(par_c)::value;
        if constexpr (TAIL == 1) { set_a_offsets(oa, m0n, has_next); sa_row0 = m0n; }   // from here on activation pieces belong to the next tile
        const unsigned char* A = smem + (gk & 1) * STAGE;
        const unsigned char* An = smem + ((gk + 1) & 1) * STAGE;
        auto phase = [&](auto p_c) __attribute__((always_inline)) {
            constexpr int p = decltype(p_c)::value;
            constexpr int CUR = (PAR + p) & 1, NXT = CUR ^ 1;
            if constexpr (p + 1 < P) read_pair(ic<NXT>{}, A, p + 1);
            else if constexpr (TAIL != 2) read_pair(ic<NXT>{}, An, 0);          // row pair 0 of the next K tile
            dma_for_phase(p, t, tail_c);
            mma(ic<CUR>{}, p, 0);
            if constexpr (p == P - 1 && TAIL != 2 && !MX) {
                __builtin_amdgcn_sched_barrier(0);
                read_w(An + A_BYTES, 0);                                        // in place: k step 0 of K tile t had its last use
            }
            mma(ic<CUR>{}, p, 1);
            if constexpr (p == P - 1 && TAIL != 2) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MX) read_w(An + A_BYTES, 0);                      // (MX: both halves feed the one MFMA of a fragment pair)
                else read_w(An + A_BYTES, 1);
            }
            if constexpr (p == P - 2) {
                // sync point: every DMA issued so far (all of K tile t+1) has landed, this wave's fragment reads are retired;
                // behind the barrier K tile t+1 is visible to every wave and the stage of K tile t is free
                if constexpr (DIAG) {
                    __builtin_amdgcn_sched_barrier(0);
                    const uint32_t ta = stamp();
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    if (dg[6]) dg[1] += dtc - dtb;                 // barrier wait of the previous sync point (stamp landed by now)
                    const uint32_t tb = stamp();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    dg[0] += tb - ta; if (t == 0) dg[4] += tb - ta; dg[6] += 1; dtb = tb;
                    bar();
                    __builtin_amdgcn_sched_barrier(0);
                    dtc = stamp();
                } else {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    bar();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        phase(ic<0>{});
        if constexpr (P > 1) phase(ic<1>{});
        if constexpr (P > 2) phase(ic<2>{});
        if constexpr (P > 3) phase(ic<3>{});
        ++gk;
    };

    __syncthreads();                                           // bias image complete (no DMA in flight yet)
    // ---- first tile: K tile 0 complete, the activation halves of K tile 1 in flight ---------------------------------------
    issue_a(0, 0, 0); issue_a(1, 0, 0); issue_w(2, 0, 0, n0); issue_w(3, 0, 0, n0);
    issue_a(0, 1, 1); issue_a(1, 1, 1);
    if constexpr (P == 2) {                                    // (no phase 0 issue of K tile 1's weight halves in this schedule)
        issue_w(2, 1, 1, n0); issue_w(3, 1, 1, n0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    bar();

    const uint32_t dk0 = DIAG ? stamp() : 0;
    for (;;) {
        uint32_t dt0 = 0, dt1 = 0;
        if constexpr (DIAG) dt0 = stamp();
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {   // tile prologue: weight fragments and row pair 0 of K tile 0 (landed and visible since the previous sync point)
            const unsigned char* A = smem + (gk & 1) * STAGE;
            read_w(A + A_BYTES, 0); read_w(A + A_BYTES, 1);
            read_pair(ic<0>{}, A, 0);
        }
        if constexpr (P & 1) {
            for (int t = 0; t < nk - 2; t += 2) { ktile(t, ic<0>{}, ic<0>{}); ktile(t + 1, ic<0>{}, ic<1>{}); }
            ktile(nk - 2, ic<1>{}, ic<0>{});
            ktile(nk - 1, ic<2>{}, ic<1>{});
        } else {
            for (int t = 0; t < nk - 2; ++t) ktile(t, ic<0>{}, ic<0>{});
            ktile(nk - 2, ic<1>{}, ic<0>{});
            ktile(nk - 1, ic<2>{}, ic<0>{});
        }
        // ---- epilogue: straight from the accumulators ------------------------------------------------------------------------
        if constexpr (DIAG) { dt1 = stamp(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dg[2] += dt1 - dt0; dg[5] += 1; }
        if constexpr (!F32OUT) {
            const bool gelu = g.flags & YV_EPI_GELU;
            const float* bl = (const float*)(smem + BIAS0) + n0 + wrow_n + fq * 16;
            float4 bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[i] = *(const float4*)(bl + i * 4);
            if constexpr (EXT == 2) {
                // out = bf16(acc + b) * gelu'(u), u = the pre-activation the forward saved (same rounding steps as the 128 x 128 kernel's
                // epilogue); u of two row fragments is in flight together
#pragma unroll
                for (int j0 = 0; j0 < MF; j0 += 2) {
                    u32x4 ur[2][2];
                    uint32_t offx[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int m = m0 + wrow_m + (j0 + jj) * 16 + fr;
                        offx[jj] = (j0 + jj < MF && m < M) ? (uint32_t)((m * g.ldaux + n0 + wrow_n + fq * 16) * 2) : 0x80000000u;
                        ur[jj][0] = __builtin_amdgcn_raw_buffer_load_b128(rsX, offx[jj], 0, 0);
                        ur[jj][1] = __builtin_amdgcn_raw_buffer_load_b128(rsX, offx[jj] + 16, 0, 0);
                    }
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        if (j0 + jj >= MF) continue;
                        const int j = j0 + jj;
                        const int m = m0 + wrow_m + j * 16 + fr;
                        uint32_t pk[8];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t d01 = pack_bf16x2(acc[i][j][0] + bv[i].x, acc[i][j][1] + bv[i].y);
                            const uint32_t d23 = pack_bf16x2(acc[i][j][2] + bv[i].z, acc[i][j][3] + bv[i].w);
                            const uint32_t u01 = ur[jj][i >> 1][(i & 1) * 2], u23 = ur[jj][i >> 1][(i & 1) * 2 + 1];
                            pk[2 * i] = pack_bf16x2(bf16_to_f32((uint16_t)(d01 & 0xffff)) * gelu_grad_f(bf16_to_f32((uint16_t)(u01 & 0xffff))),
                                                    bf16_to_f32((uint16_t)(d01 >> 16)) * gelu_grad_f(bf16_to_f32((uint16_t)(u01 >> 16))));
                            pk[2 * i + 1] = pack_bf16x2(bf16_to_f32((uint16_t)(d23 & 0xffff)) * gelu_grad_f(bf16_to_f32((uint16_t)(u23 & 0xffff))),
                                                        bf16_to_f32((uint16_t)(d23 >> 16)) * gelu_grad_f(bf16_to_f32((uint16_t)(u23 >> 16))));
                        }
                        const uint32_t off = m < M ? (uint32_t)((m * g.ldo + n0 + wrow_n + fq * 16) * 2) : 0x80000000u;
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4){pk[0], pk[1], pk[2], pk[3]}, rsO, off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4){pk[4], pk[5], pk[6], pk[7]}, rsO, off, 16, 0);
                    }
                }
            } else if (MX && (g.flags & YV_EPI_OUT_MXFP8)) {
                // the consumer is another MXFP8 GEMM (fc1 -> fc2): the lane's 16 consecutive (bf16-rounded) outputs + the 16 of lane ^ 16
                // are one 32-column MX block; same arithmetic as the 128 x 128 kernel's epilogue (byte-identical images)
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const int m = m0 + wrow_m + j * 16 + fr;
                    float f[16], amax = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v0 = acc[i][j][0] + bv[i].x, v1 = acc[i][j][1] + bv[i].y;
                        float v2 = acc[i][j][2] + bv[i].z, v3 = acc[i][j][3] + bv[i].w;
                        if (gelu) { v0 = gelu_f(v0); v1 = gelu_f(v1); v2 = gelu_f(v2); v3 = gelu_f(v3); }
                        f[4 * i] = bf16_to_f32(f32_to_bf16(v0)); f[4 * i + 1] = bf16_to_f32(f32_to_bf16(v1));
                        f[4 * i + 2] = bf16_to_f32(f32_to_bf16(v2)); f[4 * i + 3] = bf16_to_f32(f32_to_bf16(v3));
                    }
#pragma unroll
                    for (int q = 0; q < 16; ++q) amax = fmaxf(amax, fabsf(f[q]));
                    amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
                    int e = -127;
                    if (amax > 0.f) {
                        int ex;
                        const float mant = frexpf(amax * (1.0f / 448.0f), &ex);
                        e = mant == 0.5f ? ex - 1 : ex;
                        e = e < -127 ? -127 : (e > 127 ? 127 : e);
                    }
                    const float inv = ldexpf(1.0f, -e);
                    uint32_t q4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int pq = 0;
                        pq = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * i] * inv, f[4 * i + 1] * inv, pq, false);
                        pq = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * i + 2] * inv, f[4 * i + 3] * inv, pq, true);
                        q4[i] = (uint32_t)pq;
                    }
                    if (m < M) {
                        const int n = n0 + wrow_n + fq * 16;
                        *(uint4*)(g.mxq + (long long)m * g.ldmxq + n) = make_uint4(q4[0], q4[1], q4[2], q4[3]);
                        if (!(fq & 1)) {
                            const int bk = n >> 5;
                            g.mxs[((long long)(bk >> 2) * g.mx_rows + m) * 4 + (bk & 3)] = (uint8_t)(e + 127);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const int m = m0 + wrow_m + j * 16 + fr;
                    uint32_t pk[8];
                    if constexpr (EXT == 1) {                     // the pre-activation, bf16 (the backward's gelu'(u) reads it)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            pk[2 * i] = pack_bf16x2(acc[i][j][0] + bv[i].x, acc[i][j][1] + bv[i].y);
                            pk[2 * i + 1] = pack_bf16x2(acc[i][j][2] + bv[i].z, acc[i][j][3] + bv[i].w);
                        }
                        const uint32_t offx = m < M ? (uint32_t)((m * g.ldaux + n0 + wrow_n + fq * 16) * 2) : 0x80000000u;
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4){pk[0], pk[1], pk[2], pk[3]}, rsX, offx, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4){pk[4], pk[5], pk[6], pk[7]}, rsX, offx, 16, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v0 = acc[i][j][0] + bv[i].x, v1 = acc[i][j][1] + bv[i].y;
                        float v2 = acc[i][j][2] + bv[i].z, v3 = acc[i][j][3] + bv[i].w;
                        if (gelu) { v0 = gelu_f(v0); v1 = gelu_f(v1); v2 = gelu_f(v2); v3 = gelu_f(v3); }
                        pk[2 * i] = pack_bf16x2(v0, v1); pk[2 * i + 1] = pack_bf16x2(v2, v3);
                    }
                    const uint32_t off = m < M ? (uint32_t)((m * g.ldo + n0 + wrow_n + fq * 16) * 2) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){pk[0], pk[1], pk[2], pk[3]}, rsO, off, 0, YV_P9_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){pk[4], pk[5], pk[6], pk[7]}, rsO, off, 16, YV_P9_STORE_AUX);
                }
            }
        } else {
            const bool rmw = g.flags & YV_EPI_RES_F32;
            const float* bl = (const float*)(smem + BIAS0) + n0 + wrow_n + fq * 4;
            float4 bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[i] = *(const float4*)(bl + i * 16);
            constexpr int JA = 2;                                // row fragments whose residual values are in flight together
#pragma unroll
            for (int j0 = 0; j0 < MF; j0 += JA) {
                u32x4 xr[JA][4];
                uint32_t off[JA];
#pragma unroll
                for (int jj = 0; jj < JA; ++jj) {
                    const int m = m0 + wrow_m + (j0 + jj) * 16 + fr;
                    off[jj] = (j0 + jj < MF && m < M) ? (uint32_t)((m * g.ldo + n0 + wrow_n + fq * 4) * 4) : 0x80000000u;
                    if (rmw) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) xr[jj][i] = __builtin_amdgcn_raw_buffer_load_b128(rsR, off[jj] + i * 64, 0, 0);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < JA; ++jj) {
                    if (j0 + jj >= MF) continue;
                    const int j = j0 + jj;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v0 = acc[i][j][0] + bv[i].x, v1 = acc[i][j][1] + bv[i].y;
                        float v2 = acc[i][j][2] + bv[i].z, v3 = acc[i][j][3] + bv[i].w;
                        if (rmw) {
                            v0 += __uint_as_float(xr[jj][i][0]); v1 += __uint_as_float(xr[jj][i][1]);
                            v2 += __uint_as_float(xr[jj][i][2]); v3 += __uint_as_float(xr[jj][i][3]);
                        }
                        // the column step goes into the instruction's immediate offset, never into an SGPR soffset: behind a 16-byte
                        // store with a REGISTER soffset hipcc pads nothing before the next write of the data registers (LLVM takes that
                        // form to be free of the store-data hazard) and on gfx950 the store then read overwritten values (measured:
                        // 0.9 % of the outputs wrong, always the columns whose step needed a register: 128 and 192 bytes)
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(v0), __float_as_uint(v1), __float_as_uint(v2),
                                                                       __float_as_uint(v3)}, rsO, off[jj] + i * 64, 0, 0);
                    }
                }
            }
        }
        if constexpr (DIAG) { const uint32_t dt2 = stamp(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dg[3] += dt2 - dt1; }
        if (!has_next) break;
        seq += Lx;
        m0 = m0n; n0 = n0n;
        has_next = seq + Lx < seq1;
        if (has_next) coords(seq + Lx, m0n, n0n);
    }
    if constexpr (DIAG) {
        const uint32_t dk1 = stamp();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dg[7] = dk1 - dk0;
        if (lane == 0) {
            uint32_t* o = (uint32_t*)g.partial + ((long long)blockIdx.x * 8 + wave) * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = dg[i];
        }
    }
}

template <int MF, bool F32OUT, int EXT = 0, bool MX = false>
int launch_p9_inst(GemmArgs& g, hipStream_t st, int n_cu) {
    constexpr int BM = 32 * MF;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = g.N / 256;
    const size_t lds = 2 * 65536 + 16384 + (MX ? 4096 : 0);
    auto kern = gemm_p9_kernel<MF, F32OUT, 0, EXT, MX>;
    {   // the dynamic-LDS grant belongs to the device's copy of the kernel: once per device and instance
        static std::atomic<unsigned> granted[2] = {{0u}, {0u}};    // bit d: device d (up to 64 devices)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return YV_ERR_LAUNCH;
        if (!((granted[dev >> 5].load(std::memory_order_acquire) >> (dev & 31)) & 1u)) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return YV_ERR_LAUNCH;
            granted[dev >> 5].fetch_or(1u << (dev & 31), std::memory_order_release);
        }
    }
    const int tiles = g.tiles_m * g.tiles_n;
    const int grid = tiles < n_cu ? tiles : n_cu;
    if (t_time_start || t_time_stop) {
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(512), (uint32_t)lds, st, t_time_start, t_time_stop, 0, g);
        t_time_start = t_time_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, g);
    }
    return yv_launch_status();
}

// rows: 0 = choose (minimise rounds x (rows + per-tile cost)), else 160 / 192 / 224 / 256
int launch_p9(GemmArgs& g, hipStream_t st, int rows = 0, bool mx = false) {
    static int n_cu_dev = 0;
    if (!n_cu_dev) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return YV_ERR_LAUNCH;
        n_cu_dev = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int n_cu = (g_opt_p8_cus > 0 && g_opt_p8_cus < n_cu_dev) ? g_opt_p8_cus : n_cu_dev;
    g.sched = g_opt_p8_sched;
    const bool f32out = g.flags & (YV_EPI_RES_F32 | YV_EPI_OUT_F32);
    const bool even_nk = ((g.K / (mx ? 128 : BK)) & 1) == 0;         // odd-P instances (160 / 192 rows) walk K tiles in pairs
    const int ext = (g.flags & YV_EPI_SAVE_PRE) ? 1 : (g.flags & YV_EPI_GELU_BWD) ? 2 : 0;
    int best = rows ? rows : g_opt_p8_rows;
    if (ext && best > 224) best = 224;
    if (!best) {
        long long best_cost = -1;
        const int cand[6] = {256, 224, 192, 160, 128, 96};
        const bool small_ok = !mx && !ext && g_opt_p9_small;       // 128 / 96-row tiles: instances exist for the plain epilogues
        for (int c = 0; c < (small_ok ? 6 : 4); ++c) {
            if (cand[c] <= 192 && cand[c] >= 160 && !even_nk) continue;
            if (cand[c] > 192 && f32out && even_nk) continue;      // f32 outputs: the residual prefetch next to the accumulators spills above 192 rows
            if (cand[c] > 224 && ext) continue;                     // trainer epilogues: up to 224 rows
            if (cand[c] > 160 && mx && f32out) continue;            // MX with f32 output: 160 rows (registers)
            const long long tiles = (long long)((g.M + cand[c] - 1) / cand[c]) * (g.N / 256);
            const long long rounds = (tiles + n_cu - 1) / n_cu;
            // a K tile of a tile costs its rows + a fixed part (weight pieces, sync point); the short tiles pay the weight fetch
            // over fewer rows and are worth it only where the taller ones leave CUs idle (tools/gemm_lab.hip, LAB_M=6304)
            const long long cost = rounds * (cand[c] + (cand[c] < 160 ? g_opt_p9_small_fixed : 16));
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cand[c]; }
        }
    }
    if (best <= 192 && best >= 160 && !even_nk) best = 224;
    if (mx) switch (best) {
        case 224: return f32out ? launch_p9_inst<7, true, 0, true>(g, st, n_cu) : launch_p9_inst<7, false, 0, true>(g, st, n_cu);
        case 192: return f32out ? launch_p9_inst<6, true, 0, true>(g, st, n_cu) : launch_p9_inst<6, false, 0, true>(g, st, n_cu);
        case 160: return f32out ? launch_p9_inst<5, true, 0, true>(g, st, n_cu) : launch_p9_inst<5, false, 0, true>(g, st, n_cu);
        default: return f32out ? launch_p9_inst<8, true, 0, true>(g, st, n_cu) : launch_p9_inst<8, false, 0, true>(g, st, n_cu);
    }
    if (ext == 1) switch (best) {
        case 192: return launch_p9_inst<6, false, 1>(g, st, n_cu);
        case 160: return launch_p9_inst<5, false, 1>(g, st, n_cu);
        default: return launch_p9_inst<7, false, 1>(g, st, n_cu);
    }
    if (ext == 2) switch (best) {
        case 192: return launch_p9_inst<6, false, 2>(g, st, n_cu);
        case 160: return launch_p9_inst<5, false, 2>(g, st, n_cu);
        default: return launch_p9_inst<7, false, 2>(g, st, n_cu);
    }
    switch (best) {
        case 128: return f32out ? launch_p9_inst<4, true>(g, st, n_cu) : launch_p9_inst<4, false>(g, st, n_cu);
        case 96: return f32out ? launch_p9_inst<3, true>(g, st, n_cu) : launch_p9_inst<3, false>(g, st, n_cu);
        case 224: return f32out ? launch_p9_inst<7, true>(g, st, n_cu) : launch_p9_inst<7, false>(g, st, n_cu);
        case 192: return f32out ? launch_p9_inst<6, true>(g, st, n_cu) : launch_p9_inst<6, false>(g, st, n_cu);
        case 160: return f32out ? launch_p9_inst<5, true>(g, st, n_cu) : launch_p9_inst<5, false>(g, st, n_cu);
        default: return f32out ? launch_p9_inst<8, true>(g, st, n_cu) : launch_p9_inst<8, false>(g, st, n_cu);
    }
}


// ---------------------------------------------------------------------------------------------
// Weight-gradient GEMM ("TN"): dW[n][k] = sum_t dY[t][n] * X[t][k], t = token (the reduction index).
// Both operands are stored token-major, i.e. the reduction index is the ROW of the LDS tiles, so the MFMA
// fragments (8 consecutive reduction elements per lane) are COLUMNS of those tiles: they are read with the
// gfx950 hardware-transposing LDS read ds_read_b64_tr_b16 (two reads per fragment), and no transposed copy of
// the activations or of the incoming gradient is ever materialised.
//   tiles: [64 tokens][128 columns] bf16 = 256-byte LDS rows, filled by LDS-DMA (4 rows per wave-instruction);
//   the 16-byte chunk index is XOR-swizzled with ((row&3)<<1 | ((row>>3)&1)<<3) on the SOURCE address and on the
//   read, which spreads the 4 rows x 4 chunks a 16-lane group touches (and the two groups of a 32-lane half)
//   over distinct banks.
//   MFMA A operand = X columns (rows of the result = k), B operand = dY columns (result columns = n), so a lane
//   owns 4 consecutive k of one n: 16-byte f32 stores into dW (n, k).
// Token rows must be padded with ZERO rows up to a multiple of 64 (the trainer allocates its activations so).
// Few output tiles, long reduction -> always split over the token dimension (deterministic slice-order reduce).
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmArgs g) {
    // g.a0 = dY (T, N) ld lda0 ; g.w = X (T, K) ld lda1 ; T = g.K (multiple of 64) ; out (N, K) f32 ld ldo
    constexpr int TB = 64, TILE = TB * 256;                   // bytes per operand tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Nw = g.M, Kw = g.N, T = g.K;
    int bid = blockIdx.x;
    const int S = g.splitk > 1 ? g.splitk : 1;
    const int slice = bid % S;
    bid /= S;
    const int tn_ = bid / g.tiles_n, tk_ = bid - tn_ * g.tiles_n;
    const int n0 = tn_ * 128, k0 = tk_ * 128;

    // DMA: lane -> (row = 4*instr + lane/16, chunk' = lane%16); source chunk = chunk' ^ swz(row)
    const int lr = lane >> 4, lc = lane & 15;
    const uint16_t* ysrc[4];
    const uint16_t* xsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (j * 4 + wave) * 4 + lr;              // 0..63
        const int ch = lc ^ tn_swz(row);
        int cn = n0 + ch * 8; cn = cn < Nw ? cn : Nw - 8;      // column clamp (results of clamped columns are not stored)
        int ck = k0 + ch * 8; ck = ck < Kw ? ck : Kw - 8;
        ysrc[j] = g.a0 + (long long)row * g.lda0 + cn;
        long long xoff = ck;
        if (g.seg_len) { const int sg = ck / g.seg_len; xoff = (long long)sg * g.seg_stride + (ck - sg * g.seg_len); }
        xsrc[j] = g.w + (long long)row * g.lda1 + xoff;
    }
    auto issue = [&](int tt, int buf) {
        unsigned char* Yt = smem + buf * (2 * TILE);
        unsigned char* Xt = Yt + TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)(ysrc[j] + (long long)tt * TB * g.lda0), (lptr_t)(Yt + (j * 4 + wave) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[j] + (long long)tt * TB * g.lda1), (lptr_t)(Xt + (j * 4 + wave) * 1024), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];                                          // [k fragment][n fragment]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wk = wave >> 1, wn = wave & 1;                  // 2 x 2 waves, 64 (k) x 64 (n) each
    const int fi = lane & 15, mg = lane >> 4;

    auto frag = [&](const unsigned char* tile, int ks, int col0) -> bf16x8 { return tr_frag(tile, ks, lane, col0); };

    const int nt_all = T / TB;
    const int t0 = (int)((long long)nt_all * slice / S), t1 = (int)((long long)nt_all * (slice + 1) / S);
    if (t0 < t1) {
        issue(t0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int tt = t0; tt < t1; ++tt) {
        const int cur = (tt - t0) & 1;
        if (tt + 1 < t1) issue(tt + 1, cur ^ 1);
        const unsigned char* Yt = smem + cur * (2 * TILE);
        const unsigned char* Xt = Yt + TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fx[4], fy[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fx[i] = frag(Xt, ks, wk * 64 + i * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) fy[j] = frag(Yt, ks, wn * 64 + j * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[i], fy[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // D[row = k (lane>>4)*4 + reg][col = n (lane&15)]  ->  dW[n][k..k+3]
    float* out = S > 1 ? g.partial + (long long)slice * Nw * Kw : (float*)g.out;
    const long long ldo = S > 1 ? Kw : g.ldo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + fi;
        if (n >= Nw) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + wk * 64 + i * 16 + mg * 4;
            if (k < Kw) *(float4*)(out + (long long)n * ldo + k) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}

template <int BM, int BN, int WM, int WN, int ABL = 0, bool WT = false>
int launch_dma(GemmArgs& g, hipStream_t st) {
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const size_t lds = 2 * (size_t)(BM + BN) * 128;
    auto kern = gemm_dma_kernel<BM, BN, WM, WN, ABL, WT>;
    if (lds > 65536 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return YV_ERR_LAUNCH;
    const int S = g.splitk > 1 ? g.splitk : 1;
    if (t_time_start || t_time_stop) {
        // timestamps taken from the kernel's own dispatch packet (no extra barrier packets in the queue, unlike a pair of
        // hipEventRecord calls around the launch)
        hipExtLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n * S), dim3(WM * WN * 64), (uint32_t)lds, st, t_time_start,
                              t_time_stop, 0, g);
        t_time_start = t_time_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n * S), dim3(WM * WN * 64), lds, st, g);
    }
    if (S > 1) {
        const long long items = (long long)g.M * (g.N >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, g);
    }
    return yv_launch_status();
}

template <int MODE, int BM, int BN, int WM, int WN>
int launch(GemmArgs& g, hipStream_t st) {
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    size_t lds = (size_t)(BM + BN) * 128;
    if (lds < (size_t)THREADS / 64 * (BM / WM) * 128) lds = (size_t)THREADS / 64 * (BM / WM) * 128;   // the staged epilogue's slabs
    void (*kern)(GemmArgs) = igemm_kernel<MODE, BM, BN, WM, WN, false>;
    if constexpr (MODE == 1) { if (g.c1 > 0) kern = igemm_kernel<MODE, BM, BN, WM, WN, true>; }
    if (lds > 65536) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return YV_ERR_LAUNCH;
    }
    const int S = g.splitk > 1 ? g.splitk : 1;
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n * S), dim3(THREADS), lds, st, g);
    if (S > 1) {
        const long long items = (long long)g.M * (g.N >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, g);
    }
    return yv_launch_status();
}


// 16-byte row segments need 8-element (bf16) / 4-element (f32) aligned widths, strides and bases
bool epi_can_stage(const GemmArgs& g) {
    const bool f32 = g.flags & (YV_EPI_OUT_F32 | YV_EPI_RES_F32);
    const int al = f32 ? 4 : 8;
    if ((g.N % al) || (g.ldo % al) || ((uintptr_t)g.out & 15)) return false;
    if ((g.flags & YV_EPI_RES_BF16) && ((g.ldres % 8) || ((uintptr_t)g.res & 15))) return false;
    if ((g.flags & (YV_EPI_SAVE_PRE | YV_EPI_GELU_BWD)) && ((g.ldaux % 8) || ((uintptr_t)g.aux & 15))) return false;
    return true;
}

int g_opt_conv_dma = 8;             // convolutions with 64-aligned input channels on the LDS-DMA structure ("conv_dma": 0 off, 1..8 see dispatch)

template <int BN, int WM, int WN, int ST>
int launch_cdma(GemmArgs& g, hipStream_t st) {
    g.tiles_m = (g.M + 127) / 128;
    g.tiles_n = (g.N + BN - 1) / BN;
    const size_t lds = ST * (size_t)(128 + BN) * 128;
    auto kern = cgemm_dma_kernel<BN, WM, WN, ST>;
    if (lds > 65536) {
        static std::atomic<unsigned> granted[2] = {{0u}, {0u}};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return YV_ERR_LAUNCH;
        if (!((granted[dev >> 5].load(std::memory_order_acquire) >> (dev & 31)) & 1u)) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return YV_ERR_LAUNCH;
            granted[dev >> 5].fetch_or(1u << (dev & 31), std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(256), lds, st, g);
    return yv_launch_status();
}

template <int MODE>
int dispatch(GemmArgs& g, hipStream_t st) {
    if constexpr (MODE == 1) {
        const int Cin = g.c0 + g.c1;
        if (g_opt_conv_dma && g.N >= 64 && (Cin % 64) == 0 && (g.c1 == 0 || (g.ksize == 1 && (g.c0 % 64) == 0)) && g.staged &&
            g.splitk <= 1 && !g.m_dev && (g.K % BK) == 0)
        {
            // conv_dma: 1 = two stages (128-wide tiles for Cout > 64), 2 = three stages, 64-wide tiles (two workgroups per CU, four K
            // steps in flight per CU), 3 = three stages, 128-wide tiles (one workgroup per CU), 4 = four stages, 64-wide tiles,
            // 5 .. 8 mixtures.  Shipped: 8 = three stages / 64-wide, except the 128-channel layers of the large maps (>= 100 k output
            // pixels: the 80 x 80 head convolution at batch 32), which read their pixels once on two-stage 128-wide tiles (70 -> 57 us
            // alone; detect stage 1.175 -> 1.163 ms, YOLOv8m + ViT-L/16 1,376 -> 1,397 images/s; tools/conv_dma_ab.py)
            if (g_opt_conv_dma == 2) return launch_cdma<64, 4, 1, 3>(g, st);
            if (g_opt_conv_dma == 3) return g.N > 64 ? launch_cdma<128, 2, 2, 3>(g, st) : launch_cdma<64, 4, 1, 3>(g, st);
            if (g_opt_conv_dma == 4) return launch_cdma<64, 4, 1, 4>(g, st);
            if (g_opt_conv_dma == 5) {      // by reduction depth: deep K (>= 1024) three stages / 64-wide, shallow K two stages
                if (g.K >= 1024) return launch_cdma<64, 4, 1, 3>(g, st);
                return g.N > 64 ? launch_cdma<128, 2, 2, 2>(g, st) : launch_cdma<64, 4, 1, 2>(g, st);
            }
            if (g_opt_conv_dma == 7) {      // by rows: the large maps (>= 100 k output pixels) two stages, 128-wide where Cout allows
                if (g.M >= 100000) return g.N > 64 ? launch_cdma<128, 2, 2, 2>(g, st) : launch_cdma<64, 4, 1, 2>(g, st);
                return launch_cdma<64, 4, 1, 3>(g, st);
            }
            if (g_opt_conv_dma == 8) {      // as 7, only the 128-wide layers of the large maps
                if (g.M >= 100000 && g.N > 64) return launch_cdma<128, 2, 2, 2>(g, st);
                return launch_cdma<64, 4, 1, 3>(g, st);
            }
            if (g_opt_conv_dma == 6) {      // as 5, but 64-wide tiles everywhere
                return g.K >= 1024 ? launch_cdma<64, 4, 1, 3>(g, st) : launch_cdma<64, 4, 1, 2>(g, st);
            }
            return g.N > 64 ? launch_cdma<128, 2, 2, 2>(g, st) : launch_cdma<64, 4, 1, 2>(g, st);
        }
    }
    if (g.N > 64) return launch<MODE, 128, 128, 2, 2>(g, st);
    if (g.N > 32) return launch<MODE, 128, 64, 4, 1>(g, st);
    if (g.N > 16) return launch<MODE, 128, 32, 4, 1>(g, st);
    return launch<MODE, 128, 16, 4, 1>(g, st);
}

}  // namespace

extern "C" int yv_set_workspace(void* stream, void* ws, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (!ws || !bytes) g_ws.erase(stream);
    else g_ws[stream] = std::make_pair(ws, bytes);
    return YV_OK;
}

extern "C" int yv_mx_probe(const void* a, const void* b, const void* sa, const void* sb, int opsel, void* d, void* stream) {
    if (!a || !b || !sa || !sb || !d || opsel < 0 || opsel > 3) return YV_ERR_ARG;
    hipLaunchKernelGGL(mx_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const i32x8*)a, (const i32x8*)b,
                       (const int*)sa, (const int*)sb, opsel, (f32x4*)d);
    return yv_launch_status();
}

extern "C" int yv_quant_mxfp8(const void* x, long long ldx, long long rows, int K, void* q, long long ldq, void* scales,
                              long long rows_pad, void* stream) {
    if (!x || !q || !scales || rows <= 0 || K <= 0 || (K & 127) || (ldx & 7) || (ldq & 15)) return YV_ERR_ARG;
    if (rows_pad < rows || (rows_pad & 127)) return YV_ERR_ARG;
    if (((uintptr_t)x | (uintptr_t)q) & 15) return YV_ERR_ARG;
    const long long items = rows * (K >> 5);
    hipLaunchKernelGGL(quant_mx_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)x, ldx, rows, K, (uint8_t*)q, ldq, (uint8_t*)scales, rows_pad);
    return yv_launch_status();
}

static int linear_mx_impl(const void* Aq, long long lda, const void* Ascale, long long a_rows_pad, const void* Wq,
                          const void* Wscale, long long w_rows_pad, const float* bias, int M, int N, int K, void* out,
                          int ldo, int flags, const int32_t* m_dev, int m_mul, void* out_q, long long ldq, void* out_scales,
                          long long out_rows_pad, void* stream);

extern "C" int yv_linear_mxfp8(const void* Aq, long long lda, const void* Ascale, long long a_rows_pad, const void* Wq,
                               const void* Wscale, long long w_rows_pad, const float* bias, int M, int N, int K, void* out,
                               int ldo, int flags, const int32_t* m_dev, int m_mul, void* stream) {
    if (flags & YV_EPI_OUT_MXFP8) return YV_ERR_ARG;
    return linear_mx_impl(Aq, lda, Ascale, a_rows_pad, Wq, Wscale, w_rows_pad, bias, M, N, K, out, ldo, flags, m_dev, m_mul,
                          nullptr, 0, nullptr, 0, stream);
}

extern "C" int yv_linear_mxfp8_q(const void* Aq, long long lda, const void* Ascale, long long a_rows_pad, const void* Wq,
                                 const void* Wscale, long long w_rows_pad, const float* bias, int M, int N, int K, int flags,
                                 const int32_t* m_dev, int m_mul, void* out_q, long long ldq, void* out_scales,
                                 long long out_rows_pad, void* stream) {
    if (!out_q || !out_scales || (N & 127) || (ldq & 15) || ldq < N || out_rows_pad < M || (out_rows_pad & 127)) return YV_ERR_ARG;
    if (flags & ~(YV_EPI_BIAS | YV_EPI_GELU)) return YV_ERR_ARG;
    return linear_mx_impl(Aq, lda, Ascale, a_rows_pad, Wq, Wscale, w_rows_pad, bias, M, N, K, out_q, (int)ldq,
                          flags | YV_EPI_OUT_MXFP8, m_dev, m_mul, out_q, ldq, out_scales, out_rows_pad, stream);
}

static int linear_mx_impl(const void* Aq, long long lda, const void* Ascale, long long a_rows_pad, const void* Wq,
                          const void* Wscale, long long w_rows_pad, const float* bias, int M, int N, int K, void* out,
                          int ldo, int flags, const int32_t* m_dev, int m_mul, void* out_q, long long ldq, void* out_scales,
                          long long out_rows_pad, void* stream) {
    if (!Aq || !Ascale || !Wq || !Wscale || !out || M < 0 || N <= 0 || K <= 0) return YV_ERR_ARG;
    if ((K & 127) || (lda & 15) || (N & 7) || (ldo & 7)) return YV_ERR_ARG;             // whole 128-element K steps
    if (a_rows_pad < M || (a_rows_pad & 127) || w_rows_pad < N || (w_rows_pad & 127)) return YV_ERR_ARG;
    if (((uintptr_t)Ascale | (uintptr_t)Wscale) & 15) return YV_ERR_ARG;
    if ((flags & YV_EPI_BIAS) && !bias) return YV_ERR_ARG;
    if (flags & ~(YV_EPI_BIAS | YV_EPI_GELU | YV_EPI_RES_F32 | YV_EPI_OUT_F32 | YV_EPI_OUT_MXFP8)) return YV_ERR_ARG;
    if (((uintptr_t)Aq | (uintptr_t)Wq | (uintptr_t)out) & 15) return YV_ERR_ARG;
    if (M == 0) return YV_OK;
    MxArgs a = {};
    GemmArgs& g = a.g;
    g.a0 = (const uint16_t*)Aq; g.lda0 = (int)lda; g.w = (const uint16_t*)Wq; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.out = out; g.ldo = ldo; g.flags = flags; g.m_dev = m_dev; g.m_mul = m_mul;
    g.ksize = 1; g.stride = 1; g.splitk = 1;
    g.staged = epi_can_stage(g);
    if (!g.staged) return YV_ERR_ARG;
    g.group_m = g_opt_group_m > 0 ? g_opt_group_m : 8;
    a.sa = (const uint8_t*)Ascale; a.sw = (const uint8_t*)Wscale; a.rows_a = a_rows_pad; a.rows_w = w_rows_pad;
    g.mxq = (uint8_t*)out_q; g.ldmxq = ldq; g.mxs = (uint8_t*)out_scales; g.mx_rows = out_rows_pad;
    // persistent free-running kernel (round 3): the bf16 schedule with one block-scaled MFMA per fragment pair and K tile
    g.mx_sa = a.sa; g.mx_sw = a.sw; g.mx_rows_a = a.rows_a; g.mx_rows_w = a.rows_w;
    {
        const int nkm = K >> 7;
        const bool f32o = flags & (YV_EPI_RES_F32 | YV_EPI_OUT_F32);
        if (g_opt_p8 >= 3 && g_opt_variant == 1 && !(N & 255) && N <= 4096 && nkm >= 2 && M >= 2048 &&
            !((flags & YV_EPI_GELU) && f32o) && !((flags & YV_EPI_OUT_MXFP8) && f32o) && !(f32o && (nkm & 1)) &&
            (long long)((M + 159) / 160) * (N >> 8) >= 192 &&
            (long long)(M - 1) * lda + K < 0x7fffffffLL && (long long)N * K < 0x7fffffffLL &&
            ((long long)(M - 1) * ldo + N) * 4 < 0x7fffffffLL && (long long)(K >> 7) * a.rows_a * 4 < 0x7fffffffLL &&
            (long long)(K >> 7) * a.rows_w * 4 < 0x7fffffffLL)
            return launch_p9(g, (hipStream_t)stream, 0, true);
    }
    // (a 256 x 128 / 8-wave instance of the same template was measured on the ViT-L shapes: 4-15 % slower than two
    // 128 x 128 workgroups per CU, like its bf16 counterpart, and is not dispatched)
    g.tiles_m = (M + 127) / 128; g.tiles_n = (N + 127) / 128;
    // (a 4-stage ring with ONE workgroup per CU and three K steps in flight was built in round 1, measured 35 % slower than two
    // co-resident 2-stage workgroups, and removed in round 2: see DESIGN.md section 6, "MX deep ring")
    const int threads = 256;
    const size_t lds = 2 * (size_t)(128 * 128 * 2 + 1024);
    auto kern = gemm_mx_kernel<128, 128, 2, 2>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return YV_ERR_LAUNCH;
    if (t_time_start || t_time_stop) {
        hipExtLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(threads), (uint32_t)lds, (hipStream_t)stream, t_time_start,
                              t_time_stop, 0, a);
        t_time_start = t_time_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(threads), lds, (hipStream_t)stream, a);
    }
    return yv_launch_status();
}

extern "C" int yv_set_launch_timing(void* start_event, void* stop_event) {
    t_time_start = (hipEvent_t)start_event;
    t_time_stop = (hipEvent_t)stop_event;
    return YV_OK;
}

extern "C" int yv_set_option(const char* key, int value) {
    if (!key) return YV_ERR_ARG;
    if (!strcmp(key, "linear_variant")) { g_opt_variant = value; return YV_OK; }
    if (!strcmp(key, "wgrad_split_cap")) { g_opt_wgrad_cap = value; return YV_OK; }
    if (!strcmp(key, "wgrad_split")) { g_opt_wgrad_split = value; return YV_OK; }
    if (!strcmp(key, "linear_p9_small")) { g_opt_p9_small = value; return YV_OK; }
    if (!strcmp(key, "linear_p9_small_fixed")) { g_opt_p9_small_fixed = value; return YV_OK; }
    if (!strcmp(key, "linear_group_m")) { g_opt_group_m = value; return YV_OK; }
    if (!strcmp(key, "staged_epilogue")) { g_opt_staged = value; return YV_OK; }
    if (!strcmp(key, "linear_p8")) { g_opt_p8 = value; return YV_OK; }
    if (!strcmp(key, "linear_p8_rows")) { g_opt_p8_rows = value; return YV_OK; }
    if (!strcmp(key, "linear_p8_cus")) { g_opt_p8_cus = value; return YV_OK; }
    if (!strcmp(key, "linear_p8_sched")) { g_opt_p8_sched = value; return YV_OK; }
    if (!strcmp(key, "conv_splitk")) { g_opt_splitk = value; return YV_OK; }
    if (!strcmp(key, "linear_splitk")) { g_opt_linear_splitk = value; return YV_OK; }
    if (!strcmp(key, "conv_dma")) { g_opt_conv_dma = value; return YV_OK; }
    return YV_ERR_ARG;
}

extern "C" int yv_get_option(const char* key, int* value) {
    if (!key || !value) return YV_ERR_ARG;
    if (!strcmp(key, "linear_variant")) { *value = g_opt_variant; return YV_OK; }
    if (!strcmp(key, "wgrad_split_cap")) { *value = g_opt_wgrad_cap; return YV_OK; }
    if (!strcmp(key, "wgrad_split")) { *value = g_opt_wgrad_split; return YV_OK; }
    if (!strcmp(key, "linear_p9_small")) { *value = g_opt_p9_small; return YV_OK; }
    if (!strcmp(key, "linear_p9_small_fixed")) { *value = g_opt_p9_small_fixed; return YV_OK; }
    if (!strcmp(key, "linear_group_m")) { *value = g_opt_group_m; return YV_OK; }
    if (!strcmp(key, "staged_epilogue")) { *value = g_opt_staged; return YV_OK; }
    if (!strcmp(key, "linear_p8")) { *value = g_opt_p8; return YV_OK; }
    if (!strcmp(key, "linear_p8_rows")) { *value = g_opt_p8_rows; return YV_OK; }
    if (!strcmp(key, "linear_p8_cus")) { *value = g_opt_p8_cus; return YV_OK; }
    if (!strcmp(key, "linear_p8_sched")) { *value = g_opt_p8_sched; return YV_OK; }
    if (!strcmp(key, "conv_splitk")) { *value = g_opt_splitk; return YV_OK; }
    if (!strcmp(key, "linear_splitk")) { *value = g_opt_linear_splitk; return YV_OK; }
    if (!strcmp(key, "conv_dma")) { *value = g_opt_conv_dma; return YV_OK; }
    return YV_ERR_ARG;
}

static int linear_impl(const void* A, int lda, const void* W, const float* bias, int M, int N, int K, void* out, int ldo,
                       const float* pos, int tok, int flags, const int32_t* m_dev, int m_mul, const float* res_f32,
                       void* aux, int ldaux, hipStream_t stream) {
    if (!A || !W || !out || M < 0 || N <= 0 || K <= 0) return YV_ERR_ARG;
    if ((K & 7) || (lda & 7) || (N & 3) || (ldo & 3)) return YV_ERR_ARG;            // 16-byte operand chunks, 4-wide stores
    if ((flags & YV_EPI_BIAS) && !bias) return YV_ERR_ARG;
    if ((flags & YV_EPI_POSEMB) && (!pos || tok <= 0)) return YV_ERR_ARG;
    if (flags & (YV_EPI_SILU | YV_EPI_RES_BF16)) return YV_ERR_ARG;
    if ((flags & (YV_EPI_SAVE_PRE | YV_EPI_GELU_BWD)) && (!aux || (flags & (YV_EPI_OUT_F32 | YV_EPI_RES_F32)))) return YV_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)out) & 15) return YV_ERR_ARG;
    if (M == 0) return YV_OK;
    GemmArgs g = {};
    g.a0 = (const uint16_t*)A; g.lda0 = lda; g.c0 = K;
    g.w = (const uint16_t*)W; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.out = out; g.ldo = ldo; g.pos = pos; g.tok = tok; g.flags = flags; g.m_dev = m_dev; g.m_mul = m_mul;
    g.resf = res_f32; g.aux = (uint16_t*)aux; g.ldaux = ldaux;
    g.ksize = 1; g.stride = 1;
    g.staged = g_opt_staged && epi_can_stage(g);
    if ((flags & (YV_EPI_SAVE_PRE | YV_EPI_GELU_BWD)) && (!g.staged || (K % BK) || N <= 64)) return YV_ERR_ARG;
    if (res_f32 && (!g.staged || (K % BK) || N <= 64)) return YV_ERR_ARG;
    g.group_m = g_opt_group_m > 0 ? g_opt_group_m : 8;
    g.splitk = 1;
    if ((K % BK) == 0 && N > 64 && g_opt_variant != 0) {
        // split-K for wgrad-shaped problems (few 128x128 output tiles, long reduction): a 2304x768 weight gradient
        // over 6336 tokens is 108 tiles for 256 CUs; slicing K fills the chip.  Deterministic: partial sums go to
        // the stream's workspace and are added in slice order by splitk_reduce_kernel.
        void* ws = nullptr; size_t wsb = 0;
        // (not where the free-running kernel's 96-row tiles fill the chip in one round: the trainer's 6,304 x 768 data gradients were
        // two K slices of 300 tiles + a reduce pass; as 198 tiles of 96 x 256 the fine-tune step is 9.20 -> 9.00 ms)
        const bool p9_small_route = g_opt_variant == 1 && g_opt_p8 >= 3 && g_opt_p9_small && !(N & 255) && N <= 4096 && M >= 2048 &&
                                    K >= 128 && (long long)((M + 95) / 96) * (N >> 8) >= 128 && !(ldo & 7);
        if (g_opt_linear_splitk && !p9_small_route && !(flags & ~(YV_EPI_BIAS | YV_EPI_OUT_F32)) && !m_dev && ws_lookup((void*)stream, &ws, &wsb)) {
            const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
            const int nk = K / BK;
            int S = (int)(768 / tiles);
            if (S > nk / 4) S = nk / 4;
            if (S > 8) S = 8;
            if (S >= 2 && (size_t)S * M * N * sizeof(float) <= wsb) { g.splitk = S; g.partial = (float*)ws; }
        }
        int variant = g_opt_variant;
        if (g.splitk > 1) variant = 10;
        // auto: 128x128 tiles, two workgroups per CU (one workgroup's epilogue overlaps the other's main loop).
        // Isolated, the 8-phase 256x256 kernel is 3-13 % faster on N >= 1536, but inside the pipeline (operands
        // cold in L2, GELU / residual epilogues) the interleaved end-to-end A/B measures it 1-2 % slower.
        // persistent 8-phase kernel: wide bf16-output linears (the qkv / fc1 shapes), see gemm_p8_kernel for its restrictions
        // (the free-running kernel also takes the trainer's forms: a separate f32 residual source, YV_EPI_SAVE_PRE, YV_EPI_GELU_BWD)
        const bool p9_train = g_opt_p8 >= 3 && !(flags & ~(YV_EPI_BIAS | YV_EPI_GELU | YV_EPI_RES_F32 | YV_EPI_SAVE_PRE | YV_EPI_GELU_BWD)) &&
                              (!res_f32 || (flags & YV_EPI_RES_F32)) &&
                              (!(flags & YV_EPI_SAVE_PRE) || (flags & YV_EPI_GELU)) && !((flags & YV_EPI_GELU_BWD) && (flags & (YV_EPI_GELU | YV_EPI_SAVE_PRE))) &&
                              (!aux || ((long long)(M - 1) * ldaux + N) * 2 < 0x7fffffffLL) && (res_f32 || aux) && !((K / BK) & 1);
        const bool p8_ok = !(N & 255) && N <= 4096 && K >= 128 &&
                           (p9_train || (!(flags & ~(YV_EPI_BIAS | YV_EPI_GELU | YV_EPI_RES_F32 | YV_EPI_OUT_F32)) && !res_f32 && !aux)) &&
                           !((flags & YV_EPI_GELU) && (flags & (YV_EPI_RES_F32 | YV_EPI_OUT_F32))) && g.staged &&
                           ((long long)(M - 1) * lda + K) * 2 < 0x7fffffffLL && (long long)N * K * 2 < 0x7fffffffLL &&
                           ((long long)(M - 1) * ldo + N) * 4 < 0x7fffffffLL && !(ldo & 7) && !(lda & 7);
        if (variant == 1 && g_opt_p8 && p8_ok && M >= 2048 && (N >= 1536 || g_opt_p8 >= 2)) variant = g_opt_p8 >= 3 ? 11 : 9;
        if ((variant == 9 || variant == 11) && !p8_ok) variant = 1;
        if (variant == 9 && p9_train) variant = 1;                  // (those forms exist in the free-running kernel only)
        // a persistent grid of 256-column tiles needs enough tiles for the chip: the trainer's 6,304 x 768 products are 120 tiles of
        // 160 rows - fewer than half the CUs - and run faster as 300 tiles of 128 x 128 at two workgroups per CU (measured: the
        // fine-tune step 9.65 -> 9.9 ms with them on the persistent kernel)
        // (round 3, later: 96 / 128-row tiles of the free-running kernel - 198 tiles of 96 x 256 for those products - take them back
        // where no trainer epilogue is involved)
        if (variant == 11 && g_opt_variant == 1 && (long long)((M + 159) / 160) * (N >> 8) < 192) {
            const bool small_ok = g_opt_p9_small && !(flags & (YV_EPI_SAVE_PRE | YV_EPI_GELU_BWD));
            if (!small_ok || (long long)((M + 95) / 96) * (N >> 8) < 128) variant = 1;
        }
        // free-running form (round 3): f32 outputs have the registers for 160-row tiles only, whose three-phase K tiles are walked
        // in pairs (K / 64 even); the 8-phase kernel takes the rest
        if (variant == 11 && (flags & (YV_EPI_RES_F32 | YV_EPI_OUT_F32)) && ((K / BK) & 1)) variant = 9;
        switch (variant) {
            case 2: return launch_dma<256, 128, 4, 2>(g, stream);
            case 3: return launch_dma<256, 256, 2, 4>(g, stream);
            case 4: return launch_dma<128, 256, 2, 4>(g, stream);
            case 9: return launch_p8(g, stream);
            case 11: return launch_p9(g, stream);
            case 201: return launch_dma<256, 256, 2, 4, 1>(g, stream);
            case 202: return launch_dma<256, 256, 2, 4, 2>(g, stream);
            case 203: return launch_dma<256, 256, 2, 4, 3>(g, stream);
            case 204: return launch_dma<256, 256, 2, 4, 4>(g, stream);
            case 101: return launch_dma<128, 128, 2, 2, 1>(g, stream);
            case 102: return launch_dma<128, 128, 2, 2, 2>(g, stream);
            case 103: return launch_dma<128, 128, 2, 2, 3>(g, stream);
            case 104: return launch_dma<128, 128, 2, 2, 4>(g, stream);
            default: return launch_dma<128, 128, 2, 2>(g, stream);
        }
    }
    return dispatch<0>(g, stream);
}

extern "C" int yv_linear(const void* A, int lda, const void* W, const float* bias, int M, int N, int K, void* out,
                         int ldo, const float* pos, int tok, int flags, const int32_t* m_dev, int m_mul, void* stream) {
    return linear_impl(A, lda, W, bias, M, N, K, out, ldo, pos, tok, flags, m_dev, m_mul, nullptr, nullptr, 0,
                       (hipStream_t)stream);
}

extern "C" int yv_linear_ex(const void* A, int lda, const void* W, const float* bias, int M, int N, int K, void* out,
                            int ldo, int flags, const float* res_f32, void* aux, int ldaux, void* stream) {
    return linear_impl(A, lda, W, bias, M, N, K, out, ldo, nullptr, 0, flags, nullptr, 1, res_f32, aux, ldaux,
                       (hipStream_t)stream);
}

static int conv_impl_one(const yv_view* in0, const yv_view* in1, int B, int Hout, int Wout, int ksize, int stride,
                         const void* weight, const float* bias, int Cout, void* out, int out_ld, const void* res,
                         int res_ld, int flags, void* ws, size_t ws_bytes, void* stream) {
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
    if (ksize == 3 && in0->up) return YV_ERR_ARG;                 // the fused 2x upsample is a property of 1 x 1 (concat) inputs
    {   // the kernel addresses activations and weights with 32-bit byte offsets
        const long long px = (long long)B * Hout * stride * Wout * stride;
        // (+ one row and one pixel: the 3 x 3 descriptor starts that far before the tensor, see igemm_kernel)
        if ((px + Wout * stride + 1) * in0->ld * 2 >= 0x7fffffffLL || (c1 && px * in1->ld * 2 >= 0x7fffffffLL) ||
            (long long)Cout * ksize * ksize * Cin * 2 >= 0x7fffffffLL)
            return YV_ERR_LIMIT;
    }
    GemmArgs g = {};
    g.cin_shift = (Cin & (Cin - 1)) == 0 ? __builtin_ctz((unsigned)Cin) : -1;
    g.tap_uniform = (Cin % 64) == 0;
    g.a0 = (const uint16_t*)in0->ptr; g.lda0 = in0->ld; g.c0 = in0->c; g.up0 = in0->up;
    if (c1) { g.a1 = (const uint16_t*)in1->ptr; g.lda1 = in1->ld; g.c1 = c1; g.up1 = in1->up; }
    g.Hin = Hout * stride; g.Win = Wout * stride;
    g.Hout = Hout; g.Wout = Wout; g.ksize = ksize; g.stride = stride;
    g.w = (const uint16_t*)weight; g.bias = bias;
    g.M = B * Hout * Wout; g.N = Cout; g.K = ksize * ksize * Cin;
    g.out = out; g.ldo = out_ld; g.res = (const uint16_t*)res; g.ldres = res_ld; g.flags = flags;
    g.staged = g_opt_staged && epi_can_stage(g);
    // split-K: deep small-resolution layers (20x20 / 40x40 maps) give 100-400 tiles for 256 CUs and a serial chain of
    // 9-36 K steps per tile at one workgroup per CU; slicing K puts >= 2 workgroups on every CU and shortens the chain
    g.splitk = 1;
    if (ws && g_opt_splitk) {
        const int bn = g.N > 64 ? 128 : (g.N > 32 ? 64 : (g.N > 16 ? 32 : 16));
        const long long tiles = (long long)((g.M + 127) / 128) * ((g.N + bn - 1) / bn);
        const int nk = (g.K + BK - 1) / BK;
        int S = (int)(640 / tiles);
        if (S > nk / 2) S = nk / 2;
        if (S > 8) S = 8;
        if (S >= 2 && (size_t)S * g.M * g.N * sizeof(float) <= ws_bytes && !(g.N & 3)) {
            g.splitk = S;
            g.partial = (float*)ws;
        }
    }
    return dispatch<1>(g, (hipStream_t)stream);
}

// The kernel addresses each source with 32-bit byte offsets (< 2 GB): larger batches are taken in sub-batches, images being
// independent (e.g. the 48-channel C2f buffer of YOLOv8n at 320 x 320 passes 2 GB at 218 images).
static int conv_impl(const yv_view* in0, const yv_view* in1, int B, int Hout, int Wout, int ksize, int stride,
                     const void* weight, const float* bias, int Cout, void* out, int out_ld, const void* res,
                     int res_ld, int flags, void* ws, size_t ws_bytes, void* stream) {
    if (!in0 || !in0->ptr || B <= 0 || Hout <= 0 || Wout <= 0 || !out)
        return conv_impl_one(in0, in1, B, Hout, Wout, ksize, stride, weight, bias, Cout, out, out_ld, res, res_ld, flags, ws, ws_bytes,
                             stream);
    const long long Hin = (long long)Hout * stride, Win = (long long)Wout * stride;
    const bool two = in1 && in1->ptr;
    const long long s0 = (Hin >> in0->up) * (Win >> in0->up) * in0->ld * 2;
    const long long s1 = two ? (Hin >> in1->up) * (Win >> in1->up) * in1->ld * 2 : 0;
    const long long lead = (Win + 1) * in0->ld * 2;
    const long long cap = 0x7ffffff0LL;
    long long nb = s0 > 0 ? (cap - lead) / s0 : B;
    if (two && s1 > 0 && cap / s1 < nb) nb = cap / s1;
    if (nb >= B)
        return conv_impl_one(in0, in1, B, Hout, Wout, ksize, stride, weight, bias, Cout, out, out_ld, res, res_ld, flags, ws, ws_bytes,
                             stream);
    if (nb < 1) return YV_ERR_LIMIT;                              // a single image beyond 2 GB
    const long long esz = (flags & YV_EPI_OUT_F32) ? 4 : 2;
    for (long long b0 = 0; b0 < B; b0 += nb) {
        const int n = (int)(B - b0 < nb ? B - b0 : nb);
        yv_view v0 = *in0, v1 = two ? *in1 : yv_view{};
        v0.ptr = (unsigned char*)in0->ptr + b0 * s0;
        if (two) v1.ptr = (unsigned char*)in1->ptr + b0 * s1;
        const int rc = conv_impl_one(&v0, two ? &v1 : nullptr, n, Hout, Wout, ksize, stride, weight, bias, Cout,
                                     (unsigned char*)out + b0 * Hout * Wout * out_ld * esz, out_ld,
                                     res ? (const unsigned char*)res + b0 * Hout * Wout * res_ld * 2 : nullptr, res_ld, flags, ws, ws_bytes,
                                     stream);
        if (rc != YV_OK) return rc;
    }
    return YV_OK;
}

extern "C" int yv_linear_nn(const void* A, int lda, const void* Wkn, int ldw, const float* bias, int M, int N, int K,
                            void* out, int ldo, int flags, void* aux, int ldaux, void* stream) {
    // out[M,N] = A[M,K] . Wkn[K,N] : the weight is read in its reduction-major layout (dgrad on the master layout)
    if (!A || !Wkn || !out || M <= 0 || N <= 0 || K <= 0) return YV_ERR_ARG;
    if ((K % BK) || (lda & 7) || (ldw & 7) || (N & 7) || (ldo & 7)) return YV_ERR_ARG;
    if (flags & ~(YV_EPI_BIAS | YV_EPI_GELU_BWD)) return YV_ERR_ARG;
    if ((flags & YV_EPI_BIAS) && !bias) return YV_ERR_ARG;
    if ((flags & YV_EPI_GELU_BWD) && !aux) return YV_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)Wkn | (uintptr_t)out) & 15) return YV_ERR_ARG;
    GemmArgs g = {};
    g.a0 = (const uint16_t*)A; g.lda0 = lda; g.c0 = K;
    g.w = (const uint16_t*)Wkn; g.ldw = ldw; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.out = out; g.ldo = ldo; g.flags = flags; g.aux = (uint16_t*)aux; g.ldaux = ldaux;
    g.ksize = 1; g.stride = 1; g.splitk = 1;
    g.staged = epi_can_stage(g);
    if (!g.staged) return YV_ERR_ARG;
    g.group_m = g_opt_group_m > 0 ? g_opt_group_m : 8;
    return launch_dma<128, 128, 2, 2, 0, true>(g, (hipStream_t)stream);
}

static int wgrad_impl(const void* dY, int ldy, const void* X, int ldx, int T, int N, int K, float* dW, int ldw, int seg_len,
                      long long seg_stride, void* stream) {
    if (!dY || !X || !dW || T <= 0 || N <= 0 || K <= 0) return YV_ERR_ARG;
    if ((T & 63) || (N & 7) || (K & 7) || (ldy & 7) || (ldx & 7) || (ldw & 3)) return YV_ERR_ARG;
    if (((uintptr_t)dY | (uintptr_t)X | (uintptr_t)dW) & 15) return YV_ERR_ARG;
    GemmArgs g = {};
    g.seg_len = seg_len; g.seg_stride = seg_stride;
    g.a0 = (const uint16_t*)dY; g.lda0 = ldy; g.w = (const uint16_t*)X; g.lda1 = ldx;
    g.M = N; g.N = K; g.K = T; g.out = dW; g.ldo = ldw; g.flags = YV_EPI_OUT_F32;
    g.tiles_m = (N + 127) / 128; g.tiles_n = (K + 127) / 128;
    void* ws = nullptr; size_t wsb = 0;
    g.splitk = 1;
    if (ws_lookup(stream, &ws, &wsb)) {
        const long long tiles = (long long)g.tiles_m * g.tiles_n;
        // ViT weight gradients: 100+ tiles over 6k tokens -> S <= 9.  Conv weight gradients: 1-4 tiles over 10^5..10^6
        // output pixels -> up to 512 slices of >= 512 rows each (the partials stay tiny: S * Cout * 9*Cin floats)
        // matrix-shaped gradients: ONE round of the 512 workgroup slots (2 per CU) - a second, partly filled round costs more than
        // the longer slices of the first save (tools/wgrad_bench.py: dW 2304 x 768 over 6,336 tokens 43 us at S = 4 = 432
        // workgroups, 65 us at S = 5 = 540; round 2 took 1024 / tiles = 9: 63 us)
        int S = (int)((tiles <= 16 ? 1024 : 512) / tiles);
        if (S < 1) S = 1;
        const int cap = tiles <= 16 ? g_opt_wgrad_cap : 16;
        const int min_rows = tiles <= 16 ? 512 : 128;
        if (S > T / min_rows) S = T / min_rows;
        if (S > cap) S = cap;
        if (tiles > 16 && g_opt_wgrad_split > 0) S = g_opt_wgrad_split;
        const size_t fit = wsb / ((size_t)N * K * sizeof(float));
        if ((size_t)S > fit) S = (int)fit;
        if (S >= 2) { g.splitk = S; g.partial = (float*)ws; }
    }
    const int S = g.splitk;
    const size_t lds = 2 * 2 * 64 * 256;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(g.tiles_m * g.tiles_n * S), dim3(256), lds, (hipStream_t)stream, g);
    if (S > 1) {
        const long long items = (long long)g.M * (g.N >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g);
    }
    return yv_launch_status();
}

extern "C" int yv_wgrad(const void* dY, int ldy, const void* X, int ldx, int T, int N, int K, float* dW, int ldw,
                        void* stream) {
    return wgrad_impl(dY, ldy, X, ldx, T, N, K, dW, ldw, 0, 0, stream);
}

// 3x3 / stride 1 / pad 1 weight gradient without an im2col buffer.  Both operands are laid out over the PADDED pixel grid
// (B, H+2, W+2): there the tap (dy, dx) of pixel m is pixel m + dy * (W+2) + dx - a constant row offset - and, the
// activation being dense (row stride Cin), the three dx taps of one dy are 3*Cin CONSECUTIVE elements.  Column
// k = ((dy+1)*3 + dx+1)*Cin + ci of the virtual (T, 9*Cin) operand therefore sits at
//   Xp + (m - pitch - 1) * Cin + (k / 3Cin) * pitch * Cin + k % 3Cin,       pitch = W + 2,
// which is what gemm_tn's segment addressing reads.  dYp must be ZERO on the padding ring and on the tail rows; the
// activation ring must be zero too, and pitch + 1 readable rows of finite values must surround the activation.
extern "C" int yv_wgrad_conv3(const void* dYp, int ldy, const void* Xp, int Cin, int pitch, int T, int N, float* dW, int ldw,
                              void* stream) {
    if (!Xp || Cin < 8 || (Cin & 7) || pitch < 3) return YV_ERR_ARG;
    const uint16_t* base = (const uint16_t*)Xp - (long long)(pitch + 1) * Cin;
    return wgrad_impl(dYp, ldy, base, Cin, T, N, 9 * Cin, dW, ldw, 3 * Cin, (long long)pitch * Cin, stream);
}

extern "C" int yv_conv2d(const yv_view* in0, const yv_view* in1, int B, int Hout, int Wout, int ksize, int stride,
                         const void* weight, const float* bias, int Cout, void* out, int out_ld, const void* res,
                         int res_ld, int flags, void* stream) {
    return conv_impl(in0, in1, B, Hout, Wout, ksize, stride, weight, bias, Cout, out, out_ld, res, res_ld, flags, nullptr, 0,
                     stream);
}

extern "C" int yv_conv2d_ws(const yv_view* in0, const yv_view* in1, int B, int Hout, int Wout, int ksize, int stride,
                            const void* weight, const float* bias, int Cout, void* out, int out_ld, const void* res,
                            int res_ld, int flags, void* ws, size_t ws_bytes, void* stream) {
    return conv_impl(in0, in1, B, Hout, Wout, ksize, stride, weight, bias, Cout, out, out_ld, res, res_ld, flags, ws, ws_bytes,
                     stream);
}
