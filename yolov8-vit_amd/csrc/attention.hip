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
#include <type_traits>

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

// ---------------------------------------------------------------------------------------------
// Single-tile case, round 3 (N <= 32 NT <= 224: every ViT-x/16): a workgroup walks `per` consecutive (crop, head) items with K and V
// DOUBLE-BUFFERED in LDS, so that item i+1's 2 x NP x 128 bytes are in flight while item i is computed (the one-item kernel was a
// staging phase and a compute phase back to back, overlapped only by chance between the two workgroups of a CU):
//   * K and V rows (128-byte strips of the 3 D-wide token rows = one cache line each) go global -> LDS by LDS-DMA, 8 pieces of
//     8 rows per wave and tensor pair; no staging registers, no LDS stores, V is NOT transposed: the PV product's A operand
//     (V^T: 32 d x 16 keys) is read from the row-major image with the transposing LDS read ds_read_b64_tr_b16 (two per fragment:
//     keys kb + 4h + 0..3 and kb + 8 + 4h + 0..3 - the k order the exponentiated accumulator registers have);
//   * images: K chunk ^ ((row >> 1) & 7) (conflict-free ds_read_b128 of the 32-row A fragment, as before); V chunk ^ 4 ((row >> 1) & 1):
//     a 32-lane half of the transposing read takes rows q = 0..3 of a 4-key block, 64 bytes each; rows 0 / 2 (and 1 / 3) start
//     on the same bank and the swizzle sends them to opposite halves of the 128-byte row.  Swizzles act on the DMA's SOURCE chunk;
//   * one workgroup per CU (140 KB of LDS at NT = 7), so a wave may hold all NT score groups (16 NT registers): one softmax pass,
//     no running maximum, no rescale;
//   * one barrier per item: behind it item i has landed for every wave AND every wave is done with item i-1, whose buffer the
//     DMA of item i+1 then refills; the output stores of item i-1 stay in flight across it (counted vmcnt);
//   * output rows leave as 16-byte stores (v_permlane32_swap pairs the two half-waves' 8-byte pieces of a query row).
// Rows past N are fetched from row N-1 (finite; their scores are masked to -inf, their probabilities are exactly 0).
// ---------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* at_gptr_t;
typedef __attribute__((address_space(3))) void* at_lptr_t;
typedef __attribute__((ext_vector_type(4))) short at_s16x4;
typedef __attribute__((address_space(3))) at_s16x4* at_lds_s16x4_t;
typedef float at_f32x2 __attribute__((ext_vector_type(2)));

// transposing reads of PV step n (inline asm: see the kernel) and the counted wait for them
template <int N_>
__device__ __forceinline__ void at_vread(u32x2 (&vr)[2][4], uint32_t va0, uint32_t va1) {
    constexpr int off = (N_ >> 1) * 32 * 128 + (N_ & 1) * 16 * 128;
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t"
                 "ds_read_b64_tr_b16 %2, %5 offset:%6\n\tds_read_b64_tr_b16 %3, %5 offset:%7"
                 : "=&v"(vr[N_ & 1][0]), "=&v"(vr[N_ & 1][1]), "=&v"(vr[N_ & 1][2]), "=&v"(vr[N_ & 1][3])
                 : "v"(va0), "v"(va1), "i"(off), "i"(off + 8 * 128) : "memory");
}
template <int N_, bool LAST>
__device__ __forceinline__ void at_vwait(u32x2 (&vr)[2][4]) {
    if constexpr (LAST) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[N_ & 1][0]), "+v"(vr[N_ & 1][1]), "+v"(vr[N_ & 1][2]), "+v"(vr[N_ & 1][3]) :: "memory");
    else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vr[N_ & 1][0]), "+v"(vr[N_ & 1][1]), "+v"(vr[N_ & 1][2]), "+v"(vr[N_ & 1][3]) :: "memory");
}

template <int NT, int ABL = 0 /* timing only: 1 no K / V / Q fetch, 2 no compute, 3 no output stores */>
__global__ __launch_bounds__(NT * 64) void attention_pipe_kernel(const uint16_t* __restrict__ qkv, int N, int H, int R, int per,
                                                                  float scale_log2e, uint16_t* __restrict__ out,
                                                                  const int32_t* __restrict__ r_dev, float* __restrict__ lse, AttnMx mx) {
    constexpr int NP = NT * 32, TILE = NP * 128, BUF = 2 * TILE;
    constexpr int PIECES = 2 * NT * 4;                          // 8-row pieces of K and V together; PIECES / NT = 8 per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Re = r_dev ? (r_dev[0] < R ? r_dev[0] : R) : R;
    const int n_items = Re * H;
    const int it0 = blockIdx.x * per;
    const int it1 = it0 + per < n_items ? it0 + per : n_items;
    if (it0 >= it1) return;
    const int D = H * HD, ld = 3 * D;
    const int rl = lane & 31, hh = lane >> 5;

    // DMA source: lane (row = lane >> 3 of the piece, LDS chunk = lane & 7) fetches the chunk its LDS position holds.  Buffer form
    // (32-bit offsets; the host checks the tensor is below 2 GB): behind the global_load_lds form hipcc waited vmcnt(0) in front of
    // the first transposing LDS read of every item (it takes the DMA for a possible writer of the bytes read)
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, 0x7fffffff, 0x00020000);
    auto issue = [&](int item, int buf) __attribute__((always_inline)) {
        const int r = item / H, hd = item - r * H;
        const uint32_t base = (uint32_t)(((size_t)r * N * ld + hd * HD) * 2);      // wave-uniform: the instruction's scalar offset
        unsigned char* B = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pid = wave * 8 + j;                       // wave-uniform
            const int tensor = pid >= NT * 4 ? 1 : 0;
            const int prow = (pid - tensor * NT * 4) * 8;
            const int row = prow + (lane >> 3), c = lane & 7;
            const int key = row < N ? row : N - 1;
            const int cs = tensor ? (c ^ (((row >> 1) & 1) << 2)) : (c ^ ((row >> 1) & 7));
            const uint32_t voff = (uint32_t)((key * ld + (tensor + 1) * D + cs * 8) * 2);
            if (ABL != 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (at_lptr_t)(B + tensor * TILE + prow * 128), 16, (int)voff, (int)base, 0, 0);
        }
    };
    // Query rows: by LDS-DMA as well, into ONE buffer: a wave reads only its own 32 rows, so it refills them (item i+1) right after
    // its own S MFMAs of item i, no barrier involved.  (Ordinary loads to registers made hipcc wait vmcnt(0) - drain the DMA of item
    // i+1 - before their first use: staging and compute ran back to back, 13 + 30 us; hidden in inline asm, the loop-carried
    // registers were copied by compiler-placed moves before the data had landed.)
    auto issue_q = [&](int item) __attribute__((always_inline)) {
        const int r = item / H, hd = item - r * H;
        const uint32_t base = (uint32_t)(((size_t)r * N * ld + hd * HD) * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int prow = wave * 32 + j * 8;
            const int row = prow + (lane >> 3), c = lane & 7;
            const int qr = row < N ? row : N - 1;
            const uint32_t voff = (uint32_t)((qr * ld + ((c ^ ((row >> 1) & 7)) << 3)) * 2);
            if (ABL != 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (at_lptr_t)(smem + 2 * BUF + prow * 128), 16, (int)voff, (int)base, 0, 0);
        }
    };
    // V^T fragment addressing of the transposing read (see above): 16-lane group G = lane >> 4 takes d columns (G & 1) * 16 .. + 15
    // of the key rows kb + 4 (G >> 1) + q; lane 4 q + p of the group addresses row q, columns 4 p .. 4 p + 3
    const int tG = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tkq = 4 * (tG >> 1) + tq;
    const int tsw = ((tkq >> 1) & 1) << 2;
    const int tc = (tG & 1) * 2 + (tp >> 1);
    const int voff0 = tkq * 128 + (((0 ^ tsw) + tc) << 4) + (tp & 1) * 8;      // mt = 0: chunks 0..3
    const int voff1 = tkq * 128 + (((4 ^ tsw) + tc) << 4) + (tp & 1) * 8;      // mt = 1: chunks 4..7

    const bool plain = !lse && !mx.q && ABL == 0;                          // exactly 4 output stores per wave and item (the counted wait below)
    issue(it0, 0);
    issue_q(it0);
    for (int item = it0; item < it1; ++item) {
        const int buf = (item - it0) & 1;
        const bool more = item + 1 < it1;
        // item's K / V have landed (this wave's pieces; the barrier covers the others'); the previous item's output stores - the
        // youngest memory operations - stay in flight
        if (item > it0 && plain) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(item + 1, buf ^ 1);
        const unsigned char* Ks = smem + buf * BUF;
        const unsigned char* Vs = Ks + TILE;
        if (ABL == 2) { if (more) issue_q(item + 1); continue; }
        bf16x8 fq[4];
        {
            const int row = wave * 32 + rl;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fq[ks] = *(const bf16x8*)(smem + 2 * BUF + row * 128 + (((2 * ks + hh) ^ ((row >> 1) & 7)) << 4));
        }

        // ---- S^T = K . Q^T: all NT groups ----------------------------------------------------
        // (the four K fragments of group kt+1 are read before group kt's MFMAs: left to itself hipcc reads every fragment into the same
        // four registers right before its MFMA and waits lgkmcnt(0) for it - 28 exposed LDS round trips per item)
        f32x16 s[NT];
        bf16x8 kf[2][4];
        auto read_k = [&](int kt, bf16x8 (&f)[4]) __attribute__((always_inline)) {
            const int row = kt * 32 + rl;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) f[ks] = *(const bf16x8*)(Ks + row * 128 + (((2 * ks + hh) ^ ((row >> 1) & 7)) << 4));
        };
        read_k(0, kf[0]);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (kt + 1 < NT) read_k(kt + 1, kf[(kt + 1) & 1]);
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt & 1][ks], fq[ks], s[kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) issue_q(item + 1);                          // this wave's query rows had their last use (the fragments are in registers)
        // ---- softmax over the register axis + lane ^ 32, in two passes: the maximum first (it follows the S MFMAs group by group),
        // then exp / sum / pack of group kt+1 INTERLEAVED with the four PV MFMAs of group kt: left alone hipcc emits 28 S MFMAs, one
        // block of ~450 vector instructions, 28 PV MFMAs - neither pipe ever works beside the other (measured: 32 us of compute per
        // 128 crops for 7 us of MFMA time) ---------------------------------------------------------
        float mxv = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (kt == NT - 1) {                                   // only the last group can straddle N (N > 32 (NT - 1)); branch-free
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    s[kt][e] = key < N ? s[kt][e] : -INFINITY;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) mxv = fmaxf(mxv, s[kt][e]);
        }
        mxv = fmaxf(mxv, __shfl_xor(mxv, 32, 64));
        const float mb = mxv * scale_log2e;
        // packed fp32 for the scale and the row sums (v_pk_fma_f32 / v_pk_add_f32: half the vector instructions; two waves share a
        // SIMD's vector pipe here and it is the pipe this kernel is bound by - tools/valu_rate.hip)
        at_f32x2 l2 = {0.f, 0.f};
        const at_f32x2 sc2 = {scale_log2e, scale_log2e}, nmb2 = {-mb, -mb};
        bf16x8 fp[NT][2];
        auto expg = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                // elements e .. e + 3 of every lane are keys >= 32 kt + 8 (e >> 2): past N (last group only) they are exactly 0
                if (kt == NT - 1 && kt * 32 + 8 * (e >> 2) >= N) {
                    fp[kt][e >> 3][e & 7] = (__bf16)0.f; fp[kt][e >> 3][(e & 7) + 1] = (__bf16)0.f;
                    continue;
                }
                const at_f32x2 x = __builtin_elementwise_fma(at_f32x2{s[kt][e], s[kt][e + 1]}, sc2, nmb2);
                const at_f32x2 pp = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                l2 += pp;
                fp[kt][e >> 3][e & 7] = (__bf16)pp[0]; fp[kt][e >> 3][(e & 7) + 1] = (__bf16)pp[1];
            }
        };
        f32x16 o[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[mt][e] = 0.f;
        // The transposing reads are inline asm (their builtin makes hipcc wait vmcnt(0) - drain item i+1's DMA - before the first
        // one: it cannot see that they read the OTHER buffer), so their completion is counted by hand: the four reads of step n+1
        // are issued before the wait for step n's (lgkmcnt(4): LDS results return in order; LDS operations hipcc issues in between
        // only make the wait stricter).  Step n = (key group kt, 16-key half st); a step's reads: (mt 0, mt 1) x (keys +0..3, +8..11).
        u32x2 vr[2][4];
        const uint32_t vbase = (uint32_t)(uintptr_t)(at_lptr_t)(const_cast<unsigned char*>(Vs));
        const uint32_t va0 = vbase + voff0, va1 = vbase + voff1;
        auto pv_step = [&](auto n_c) __attribute__((always_inline)) {
            constexpr int n = decltype(n_c)::value, kt = n >> 1, st = n & 1;
            if constexpr (n + 1 < 2 * NT) at_vread<n + 1>(vr, va0, va1);
            at_vwait<n, !(n + 1 < 2 * NT)>(vr);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const u32x4 pkv = {vr[n & 1][2 * mt][0], vr[n & 1][2 * mt][1], vr[n & 1][2 * mt + 1][0], vr[n & 1][2 * mt + 1][1]};
                o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pkv), fp[kt][st], o[mt], 0, 0, 0);
            }
        };
        expg(0);
        at_vread<0>(vr, va0, va1);
        auto pv_group = [&](auto kt_c) __attribute__((always_inline)) {
            constexpr int kt = decltype(kt_c)::value;
            if constexpr (kt < NT) {
                pv_step(std::integral_constant<int, 2 * kt>{});
                pv_step(std::integral_constant<int, 2 * kt + 1>{});
                if constexpr (kt + 1 < NT) expg(kt + 1);
            }
        };
        pv_group(std::integral_constant<int, 0>{}); pv_group(std::integral_constant<int, 1>{}); pv_group(std::integral_constant<int, 2>{});
        pv_group(std::integral_constant<int, 3>{}); pv_group(std::integral_constant<int, 4>{}); pv_group(std::integral_constant<int, 5>{});
        pv_group(std::integral_constant<int, 6>{});
        const float l = (l2[0] + l2[1]) + __shfl_xor(l2[0] + l2[1], 32, 64);
        // ---- normalise and store -------------------------------------------------------------------
        const int r = item / H, hd = item - r * H;
        const int q = wave * 32 + rl;
        const float inv = 1.0f / l;
        if (lse && hh == 0 && q < N) lse[((size_t)r * H + hd) * N + q] = mb + log2f(l);              // log2 domain
        if (mx.q) {
            if (q < N) {
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
                        int pq = 0;
                        pq = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * g4] * is, f[4 * g4 + 1] * is, pq, false);
                        pq = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * g4 + 2] * is, f[4 * g4 + 3] * is, pq, true);
                        *(uint32_t*)(mx.q + row * mx.ldq + hd * HD + mt * 32 + 8 * g4 + 4 * hh) = (uint32_t)pq;
                    }
                    if (hh == 0) {
                        const int bk = hd * 2 + mt;
                        mx.s[((long long)(bk >> 2) * mx.rows + row) * 4 + (bk & 3)] = (uint8_t)(e + 127);
                    }
                }
            }
        } else {
            // a lane holds d = 32 mt + 8 g + 4 hh .. + 3 of its query; pairs of groups (g, g + 1) are exchanged between the half-waves
            // so that lane hh = 0 owns d = 8 g .. 8 g + 7 and lane hh = 1 owns d = 8 (g + 1) .. + 7: 8 stores of 16 bytes
            const auto rsO = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);   // (offsets checked by the host: < 2 GB)
            const uint32_t ob = q < N ? (uint32_t)((((size_t)r * N + q) * D + hd * HD) * 2) : 0x80000000u;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4 += 2) {
                    uint32_t a0 = pack_bf16x2(o[mt][4 * g4] * inv, o[mt][4 * g4 + 1] * inv);
                    uint32_t a1 = pack_bf16x2(o[mt][4 * g4 + 2] * inv, o[mt][4 * g4 + 3] * inv);
                    uint32_t b0 = pack_bf16x2(o[mt][4 * g4 + 4] * inv, o[mt][4 * g4 + 5] * inv);
                    uint32_t b1 = pack_bf16x2(o[mt][4 * g4 + 6] * inv, o[mt][4 * g4 + 7] * inv);
                    const auto x0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto x1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    const u32x4 pkv = {x0[0], x1[0], x0[1], x1[1]};
                    if (ABL != 3) __builtin_amdgcn_raw_buffer_store_b128(pkv, rsO, ob + (uint32_t)((mt * 32 + 8 * (g4 + hh)) * 2), 0, 0);
                    else asm volatile("" :: "v"(pkv));
                }
        }
    }
}

template <int NT>
int launch_attn_pipe(const uint16_t* qkv, int R, int N, int H, float scale, uint16_t* out, const int32_t* r_dev, float* lse,
                     hipStream_t st, AttnMx mx) {
    constexpr int NP = NT * 32;
    const size_t lds = (size_t)5 * NP * 128;                   // K, V double-buffered + this item's query rows
    auto kern = g_attn_abl == 11 ? attention_pipe_kernel<NT, 1> : g_attn_abl == 12 ? attention_pipe_kernel<NT, 2> :
                g_attn_abl == 13 ? attention_pipe_kernel<NT, 3> : attention_pipe_kernel<NT, 0>;
    if (lds > 65536 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return YV_ERR_LAUNCH;
    // items per workgroup: enough to amortise the pipeline fill, few enough that the grid still fills and balances the chip
    const int items = R * H;
    int per = items / 256;
    per = per < 1 ? 1 : (per > 6 ? 6 : per);
    const int grid = (items + per - 1) / per;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT * 64), lds, st, qkv, N, H, R, per, scale * 1.4426950408889634f, out, r_dev, lse, mx);
    return yv_launch_status();
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
    if (nt <= 7 && (g_attn_abl == 0 || g_attn_abl > 10) && (long long)R * N * H * HD * 6 < 0x7fffffffLL) {      // single tile: the pipelined kernel
        switch (nt) {
            case 1: return launch_attn_pipe<1>(q, R, N, H, scale, o, r_dev, lse, st, mx);
            case 2: return launch_attn_pipe<2>(q, R, N, H, scale, o, r_dev, lse, st, mx);
            case 3: return launch_attn_pipe<3>(q, R, N, H, scale, o, r_dev, lse, st, mx);
            case 4: return launch_attn_pipe<4>(q, R, N, H, scale, o, r_dev, lse, st, mx);
            case 5: return launch_attn_pipe<5>(q, R, N, H, scale, o, r_dev, lse, st, mx);
            case 6: return launch_attn_pipe<6>(q, R, N, H, scale, o, r_dev, lse, st, mx);
            default: return launch_attn_pipe<7>(q, R, N, H, scale, o, r_dev, lse, st, mx);
        }
    }
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
