// Fused non-causal attention forward for ViT token counts (197 single-tile, 785 tiled), head dim 64:
//   O = softmax(Q K^T * scale) V              timm Attention (README.md:21-23), SURVEY.md row B3 / K10.
//
// One workgroup per (crop, head, query block); wave w owns 32 query rows.  K (row major) and V^T tiles
// live in LDS; at N = 197 the whole sequence is ONE tile (28 KB + 29 KB) so there is no online-softmax
// rescale and each wave computes its full 32 x N score block in registers; longer sequences (ViT-B/8:
// 785 tokens) walk 256-row K/V tiles with running max / sum.
//
// MFMA plan (v_mfma_f32_32x32x16_bf16, accumulators in f32):
//   S^T tile = K_tile (A: 32 keys x 64 d) . Q^T (B: 64 d x 32 queries)
//     -> the accumulator has the KEY on the register axis and the QUERY on the lane,
//        so the softmax row-reduction is in-lane + one lane^32 exchange, and
//   O^T = V^T (A: 64 d x keys) . P^T (B: keys x 32 queries)
//     takes the exponentiated accumulator registers directly as its B operand
//     (registers 8s..8s+7 -> k-step s; k order inside a step is
//      16s + 8(j>>2) + 4h + (j&3), matched by the V^T fragment addressing): P never
//     touches LDS.
//   LDS images: K rows are 128 B with the 16-byte chunk XOR-swizzled by ((row>>1)&7)
//   (conflict-free ds_read_b128 for the 32-row A fragment); V^T rows are padded to
//   2*NP+8 bytes (odd multiple of 8 B: conflict-free ds_read_b64 over 32 rows).
#include "yv_common.h"

namespace {

constexpr int HD = 64;
int g_attn_abl = 0;               // diagnostic builds (yv_attention_debug)

// optional MXFP8 image of the output (operand of the proj GEMM of yv_linear_mxfp8): e4m3 bytes + E8M0 per 32 columns
struct AttnMx { uint8_t* q; long long ldq; uint8_t* s; long long rows; };

// SINGLE (N <= NP, the ViT-x/16 case: 197 tokens in one 224-row tile): the 32-key score groups are consumed one at a time with
// an online softmax (running max / sum per query, O rescaled only when some row's maximum really grew - typically 2-3 of the
// 7 groups) instead of holding all NT groups (16 f32 registers each) across the softmax: ~100 VGPRs instead of 194, i.e. TWO
// workgroups per CU, so that one workgroup's global -> LDS staging (HBM-bound: 155 MB per launch at 128 crops) runs under the
// other's MFMA / exp work.  Round-1 form: one 7-wave workgroup per CU, staging (29 us) and compute (40 us) back to back.
template <int NT, int ABL = 0, bool SINGLE = false>
__global__ __launch_bounds__(NT * 64, SINGLE ? 4 : 1) void attention_kernel(const uint16_t* __restrict__ qkv, int N, int H, int QB,
                                                            float scale_log2e, uint16_t* __restrict__ out,
                                                            const int32_t* __restrict__ r_dev, float* __restrict__ lse, AttnMx mx) {
    // blockIdx.x = ((crop * H + head) * QB + query block); a query block = NT waves x 32 queries = NP rows.
    // Keys/values are consumed in tiles of NP rows with an online softmax (running max m, sum l, rescaled
    // O); N <= NP (ViT-x/16: 197 <= 224) is the single-tile case and pays no rescale.
    constexpr int NP = NT * 32;
    constexpr int VT_STRIDE = NP * 2 + 8;               // bytes per V^T row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;                            // NP x 128 B
    unsigned char* Vt = smem + NP * 128;                 // 64 x VT_STRIDE
    const int qb = blockIdx.x % QB;
    const int rh = blockIdx.x / QB;
    const int r = rh / H, hd = rh - r * H;
    if (r_dev && r >= r_dev[0]) return;
    const int D = H * HD, ld = 3 * D;
    const uint16_t* base = qkv + (size_t)r * N * ld + hd * HD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int T = NT * 64;

    // ---- Q fragments (B operand), straight from global --------------------------------------
    const int rl = lane & 31, hh = lane >> 5;
    const int q = qb * NP + wave * 32 + rl;
    const int qc = q < N ? q : N - 1;
    bf16x8 fq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fq[ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 16 + hh * 8);

    f32x16 o[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[mt][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int kv0 = 0; kv0 < N; kv0 += NP) {
        if (kv0 > 0) __syncthreads();                    // every wave is done reading the previous tile
        // ---- stage K (swizzled rows) and V^T (explicit transpose) -------------------------
        // All of a thread's K / V chunks are fetched before the first one is stored (NP * 8 / T = 4 trips, 8 loads in flight), from
        // a CLAMPED row instead of behind `if (key < N)`: rows past N only have to be finite (their scores are masked to -inf,
        // their probabilities are exactly 0), and a conditional load is a branch + vmcnt(0) per trip - the staging phase of the
        // first version was 4 dependent HBM round trips per workgroup.
        constexpr int TRIPS = (NP * 8 + T - 1) / T;
        uint4 kvr[TRIPS], vvr[TRIPS];
#pragma unroll
        for (int i = 0; i < TRIPS; ++i) {
            const int it = tid + i * T;
            const int kl = (it < NP * 8 ? it : NP * 8 - 1) >> 3, c = it & 7;
            int key = kv0 + kl;
            key = key < N ? key : N - 1;
            kvr[i] = vvr[i] = make_uint4(0, 0, 0, 0);
            if (ABL != 1) {
                kvr[i] = *(const uint4*)(base + (size_t)key * ld + D + c * 8);
                vvr[i] = *(const uint4*)(base + (size_t)key * ld + 2 * D + c * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < TRIPS; ++i) {
            const int it = tid + i * T;
            if (it >= NP * 8) break;
            const int kl = it >> 3, c = it & 7;
            *(uint4*)(Ks + kl * 128 + ((c ^ ((kl >> 1) & 7)) << 4)) = kvr[i];
            const uint32_t w[4] = {vvr[i].x, vvr[i].y, vvr[i].z, vvr[i].w};
            if (ABL != 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint16_t val = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
                    *(uint16_t*)(Vt + (c * 8 + e) * VT_STRIDE + kl * 2) = val;
                }
            }
        }
        __syncthreads();
        if (ABL == 2) continue;

        if constexpr (SINGLE) {
            float l_lane = 0.f;                               // this lane's share of the row sum (lane ^ 32 holds the rest)
#pragma unroll 1
            for (int kt = 0; kt < NT; ++kt) {                // rolled: unrolled, the scheduler hoists every group's MFMAs again
                f32x16 sv;
#pragma unroll
                for (int e = 0; e < 16; ++e) sv[e] = 0.f;
                const int row = kt * 32 + rl;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int c = 2 * ks + hh;
                    const bf16x8 fk = *(const bf16x8*)(Ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                    sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk, fq[ks], sv, 0, 0, 0);
                }
                if (kt * 32 + 32 > N) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        sv[e] = key < N ? sv[e] : -INFINITY;
                    }
                }
                float mx = sv[0];
#pragma unroll
                for (int e = 1; e < 16; ++e) mx = fmaxf(mx, sv[e]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run, mx);        // group 0 holds valid keys for every row -> finite from then on
                if (__any(m_new > m_run)) {                  // wave-uniform: some query's maximum grew
                    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);   // 1 for unchanged rows, 0 at the start
                    l_lane *= alpha;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) o[mt][e] *= alpha;
                    m_run = m_new;
                }
                const float mb = m_run * scale_log2e;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    sv[e] = __builtin_amdgcn_exp2f(fmaf(sv[e], scale_log2e, -mb));
                    l_lane += sv[e];
                }
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    bf16x8 fp;
#pragma unroll
                    for (int j = 0; j < 8; ++j) fp[j] = (__bf16)sv[8 * st + j];
                    const int key0 = kt * 32 + 16 * st + 4 * hh;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const unsigned char* vr = Vt + (mt * 32 + rl) * VT_STRIDE + key0 * 2;
                        const uint2 lo = *(const uint2*)vr;
                        const uint2 hi = *(const uint2*)(vr + 16);
                        const u32x4 pk = {lo.x, lo.y, hi.x, hi.y};
                        o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pk), fp, o[mt], 0, 0, 0);
                    }
                }
            }
            l_run = l_lane + __shfl_xor(l_lane, 32, 64);
            continue;
        }

        // ---- S^T = K . Q^T ------------------------------------------------------------------
        f32x16 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
            const int row = kt * 32 + rl;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = 2 * ks + hh;
                const bf16x8 fk = *(const bf16x8*)(Ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk, fq[ks], s[kt], 0, 0, 0);
            }
        }

        // ---- online softmax over this tile's keys (register axis + lane^32) -----------------
        // only a 32-key group that straddles N needs masking (197 tokens: the last of 7 groups)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (kv0 + kt * 32 + 32 > N) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kv0 + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    s[kt][e] = key < N ? s[kt][e] : -INFINITY;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[kt][e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);            // every tile holds >= 1 valid key -> finite
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);   // first tile: exp2(-inf) = 0
        const float mb = m_new * scale_log2e;
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][e], scale_log2e, -mb));    // one v_fma + one v_exp
                s[kt][e] = p;
                l += p;
            }
        l += __shfl_xor(l, 32, 64);
        l_run = l_run * alpha + l;
        m_run = m_new;
        if (kv0 > 0) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[mt][e] *= alpha;
        }

        // ---- O^T += V^T . P^T ---------------------------------------------------------------
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 fp;
#pragma unroll
                for (int j = 0; j < 8; ++j) fp[j] = (__bf16)s[kt][8 * st + j];
                const int key0 = kt * 32 + 16 * st + 4 * hh;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const unsigned char* vr = Vt + (mt * 32 + rl) * VT_STRIDE + key0 * 2;
                    const uint2 lo = *(const uint2*)vr;
                    const uint2 hi = *(const uint2*)(vr + 16);
                    const u32x4 pk = {lo.x, lo.y, hi.x, hi.y};
                    const bf16x8 fv = __builtin_bit_cast(bf16x8, pk);
                    o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv, fp, o[mt], 0, 0, 0);
                }
            }
        }
    }

    // ---- normalise and store: lane owns d = 32mt + 8g + 4hh .. +3 of its query -------------
    if (q < N) {
        if (lse && hh == 0) lse[((size_t)r * H + hd) * N + q] = m_run * scale_log2e + log2f(l_run);   // log2 domain
        const float inv = 1.0f / l_run;
        if (mx.q) {
            // a head's 64 columns are two MX blocks (d 0..31 = mt 0, d 32..63 = mt 1); a query's values of one block sit
            // in this lane and in lane ^ 32; the bf16 rounding of the ordinary output is kept
            const long long row = (long long)r * N + q;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float f[16], amax = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) { f[k] = bf16_to_f32(f32_to_bf16(o[mt][k] * inv)); amax = fmaxf(amax, fabsf(f[k])); }
                amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
                int e = -127;
                if (amax > 0.f) {
                    int ex;
                    const float mant = frexpf(amax * (1.0f / 448.0f), &ex);
                    e = mant == 0.5f ? ex - 1 : ex;
                    e = e < -127 ? -127 : (e > 127 ? 127 : e);
                }
                const float is = ldexpf(1.0f, -e);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    int p = 0;
                    p = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * g4] * is, f[4 * g4 + 1] * is, p, false);
                    p = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * g4 + 2] * is, f[4 * g4 + 3] * is, p, true);
                    *(uint32_t*)(mx.q + row * mx.ldq + hd * HD + mt * 32 + 8 * g4 + 4 * hh) = (uint32_t)p;
                }
                if (hh == 0) {
                    const int bk = hd * 2 + mt;
                    mx.s[((long long)(bk >> 2) * mx.rows + row) * 4 + (bk & 3)] = (uint8_t)(e + 127);
                }
            }
            return;
        }
        uint16_t* orow = out + ((size_t)r * N + q) * D + hd * HD;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = mt * 32 + 8 * g4 + 4 * hh;
                *(uint2*)(orow + d) = make_uint2(pack_bf16x2(o[mt][4 * g4] * inv, o[mt][4 * g4 + 1] * inv),
                                                 pack_bf16x2(o[mt][4 * g4 + 2] * inv, o[mt][4 * g4 + 3] * inv));
            }
    }
}

template <int NT>
int launch_attn(const uint16_t* qkv, int R, int N, int H, float scale, uint16_t* out, const int32_t* r_dev, float* lse,
                hipStream_t st, AttnMx mx) {
    constexpr int NP = NT * 32;
    const size_t lds = (size_t)NP * 128 + 64 * (size_t)(NP * 2 + 8);
    const bool single = N <= NP && g_attn_abl != 4;           // ablation 4: the round-1 form of the single-tile case (all groups held)
    auto kern = g_attn_abl == 1 ? attention_kernel<NT, 1> : g_attn_abl == 2 ? attention_kernel<NT, 2> : g_attn_abl == 3 ? attention_kernel<NT, 3> :
                single ? attention_kernel<NT, 0, true> : attention_kernel<NT, 0, false>;
    if (lds > 65536 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return YV_ERR_LAUNCH;
    const int QB = (N + NP - 1) / NP;
    hipLaunchKernelGGL(kern, dim3(R * H * QB), dim3(NT * 64), lds, st, qkv, N, H, QB, scale * 1.4426950408889634f, out,
                       r_dev, lse, mx);
    return yv_launch_status();
}

}  // namespace

static int attention_impl(const void* qkv, int R, int N, int H, float scale, void* out, const int32_t* r_dev, float* lse,
                          void* stream, AttnMx mx = AttnMx{nullptr, 0, nullptr, 0}) {
    if (!qkv || (!out && !mx.q) || R < 0 || N <= 0 || H <= 0) return YV_ERR_ARG;
    if ((long long)R * H * ((N + 255) / 256) > 0x7fffffffLL) return YV_ERR_LIMIT;
    if (R == 0) return YV_OK;
    const uint16_t* q = (const uint16_t*)qkv;
    uint16_t* o = (uint16_t*)out;
    hipStream_t st = (hipStream_t)stream;
    const int nt = N > 256 ? 8 : (N + 31) / 32;        // > 256 tokens: 256-row tiles, online softmax
    switch (nt) {
        case 1: return launch_attn<1>(q, R, N, H, scale, o, r_dev, lse, st, mx);
        case 2: return launch_attn<2>(q, R, N, H, scale, o, r_dev, lse, st, mx);
        case 3: return launch_attn<3>(q, R, N, H, scale, o, r_dev, lse, st, mx);
        case 4: return launch_attn<4>(q, R, N, H, scale, o, r_dev, lse, st, mx);
        case 5: return launch_attn<5>(q, R, N, H, scale, o, r_dev, lse, st, mx);
        case 6: return launch_attn<6>(q, R, N, H, scale, o, r_dev, lse, st, mx);
        case 7: return launch_attn<7>(q, R, N, H, scale, o, r_dev, lse, st, mx);
        default: return launch_attn<8>(q, R, N, H, scale, o, r_dev, lse, st, mx);
    }
}

extern "C" int yv_attention_debug(int ablate) { g_attn_abl = ablate; return YV_OK; }

extern "C" int yv_attention(const void* qkv, int R, int N, int H, float scale, void* out, const int32_t* r_dev,
                            void* stream) {
    return attention_impl(qkv, R, N, H, scale, out, r_dev, nullptr, stream);
}

extern "C" int yv_attention_mxfp8(const void* qkv, int R, int N, int H, float scale, void* out_q, long long ldq,
                                  void* out_scales, long long rows_pad, const int32_t* r_dev, void* stream) {
    if (!out_q || !out_scales || (ldq & 15) || ldq < (long long)H * 64 || rows_pad < (long long)R * N || (rows_pad & 127))
        return YV_ERR_ARG;
    if ((H & 1)) return YV_ERR_ARG;                         // whole 128-column K steps of the consumer: H*64 % 128 == 0
    return attention_impl(qkv, R, N, H, scale, nullptr, r_dev, nullptr, stream,
                          AttnMx{(uint8_t*)out_q, ldq, (uint8_t*)out_scales, rows_pad});
}

extern "C" int yv_attention_train(const void* qkv, int R, int N, int H, float scale, void* out, float* lse, void* stream) {
    if (!lse) return YV_ERR_ARG;
    return attention_impl(qkv, R, N, H, scale, out, nullptr, lse, stream);
}
