// Fused C2f block of the detector backbone (ultralytics C2f, shortcut = True; reference call site: the YOLO(pt) forward of
// utils/utils.py:126 / test.ipynb - layers model.2 and model.4 of YOLOv8n: 160 x 160 x 32 and 80 x 80 x 64 feature maps).
//
//     y0 | y1 = SiLU(cv1 x)            1 x 1, C1 = 2c -> 2c
//     y(j+2)  = y(j+1) + SiLU(m_j.cv2 * SiLU(m_j.cv1 * y(j+1)))      two 3 x 3 convolutions c -> c, j < n
//     out     = SiLU(cv2 [y0 | y1 | .. | y(n+1)])                     1 x 1, (2 + n) c -> 2c
//
// Layer by layer these are 4 (n = 1) or 6 (n = 2) launches whose operands are the block's own intermediates: at c = 16 / 32 the
// 3 x 3 layers move 1.2 TB/s (tools/conv_layers.py: 157 + 134 us for the two blocks at batch 32, 20 + 10 us of HBM traffic).
// Here one workgroup produces a 16 x 16 output tile of one image and keeps every intermediate in LDS: the cv1 outputs over
// the tile + a halo of 2 n pixels, each bottleneck shrinking the region by 2; HBM sees x once (plus the halo overlap) and
// out once.  Pixels outside the image are ZERO in every intermediate (the 3 x 3 layers' zero padding acts on y / t).
//
// MFMA orientation (v_mfma_f32_16x16x32_bf16): A = weights (16 output channels x 32 K), B = pixels (32 K x 16 pixels), so a
// lane ends up with 4 consecutive output channels of ONE pixel (8-byte stores) and the weight fragments - the same for every
// pixel group of a phase - stay in registers, read from global memory in their (Cout, K) layout as they are.
// LDS layout of an intermediate: channel-group-major planes, [c / 8][pixels][8 channels]: the B fragment of a pixel group is
// 16 consecutive pixels x 16 bytes of one plane = 256 contiguous bytes for any tap shift (no bank conflicts, no swizzle).
// K order (tap, channel) and one f32 rounding chain per output: bias -> SiLU -> (+ residual) -> bf16, as the layer kernels'
// unstaged epilogue (gemm.hip) - tests/test_gpu_models.py compares the two paths.
#include "yv_common.h"
#include <atomic>

namespace {

struct C2fArgs {
    const uint16_t* x; long long ldx;
    int B, H, W, tiles_x, tiles_y;
    const uint16_t* w_cv1; const float* b_cv1;
    const uint16_t* w_m[4]; const float* b_m[4];
    const uint16_t* w_cv2; const float* b_cv2;
    uint16_t* out; long long ldo;
    unsigned long long* dbg;               // diagnostics (yv_c2f_debug): 8 cycle stamps of wave 0 of the first 64 workgroups
};

__device__ __forceinline__ float c2f_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

typedef __attribute__((ext_vector_type(4))) uint32_t c2f_u32x4;

__device__ __forceinline__ bf16x8 c2f_zero8() { return __builtin_bit_cast(bf16x8, (c2f_u32x4){0u, 0u, 0u, 0u}); }

template <int C, int NB, int TT> struct C2fGeo {
    static constexpr int C2F_T = TT;
    static constexpr int HALO = 2 * NB, R0 = C2F_T + 2 * HALO, PXB = C * 2;        // bytes per pixel of a c-channel array
    static constexpr int y_w(int j) { return j == 0 ? C2F_T : R0 - 4 * (j - 1); }   // width of y_j's region (y0: centre only)
    static constexpr int t_w(int j) { return R0 - 2 - 4 * j; }                       // width of bottleneck j's inner tensor
    static constexpr int y_off(int j) {                                               // byte offset of y_j in LDS: y0 y1 t0 y2 [t1 y3]
        int o = 0;
        for (int i = 0; i < j; ++i) o += y_w(i) * y_w(i) * PXB + (i >= 1 && i <= NB ? t_w(i - 1) * t_w(i - 1) * PXB : 0);
        return o;
    }
    static constexpr int t_off(int j) { return y_off(j + 1) + y_w(j + 1) * y_w(j + 1) * PXB; }
    static constexpr int BYTES = y_off(NB + 1) + y_w(NB + 1) * y_w(NB + 1) * PXB;
};

// Weight fragments + bias of one 3 x 3 layer c -> c, as the A operands of its MFMAs: loaded straight from the (Cout, 9 c) global
// layout one phase AHEAD of their use (in front of the barrier that ends the previous phase), so that their latency is not on
// the tile's critical path.
template <int C> struct C2fConvW {
    static constexpr int CPG = C / 8, KG = 9 * CPG, KS = (KG + 3) / 4, NFR = C / 16;
    bf16x8 wa[NFR][KS];
    float4 bv[NFR];
};
template <int C>
__device__ __forceinline__ void c2f_load_conv_w(C2fConvW<C>& w, const uint16_t* __restrict__ wgt, const float* __restrict__ bias, int r,
                                                int q) {
    using CW = C2fConvW<C>;
#pragma unroll
    for (int ks = 0; ks < CW::KS; ++ks) {
        const int kg = 4 * ks + q;
#pragma unroll
        for (int f = 0; f < CW::NFR; ++f)
            w.wa[f][ks] = kg < CW::KG ? *(const bf16x8*)(wgt + (size_t)(16 * f + r) * (9 * C) + 8 * kg) : c2f_zero8();
    }
#pragma unroll
    for (int f = 0; f < CW::NFR; ++f) w.bv[f] = *(const float4*)(bias + 16 * f + 4 * q);
}

// 3 x 3 / stride 1 / pad 1 layer c -> c between two LDS arrays: S (width ws) -> D (width ws - 2, the region one pixel inside).
// RES: + the bottleneck's input (array Rs, width ws + 2: the region one pixel outside S).
template <int C, int NTH, bool RES>
__device__ __forceinline__ void c2f_conv3(const C2fConvW<C>& w, const unsigned char* S, int ws, unsigned char* D,
                                          const unsigned char* Rs, int oy, int ox, int H, int W, int lane, int wave) {
    using CW = C2fConvW<C>;
    constexpr int CPG = CW::CPG, KG = CW::KG, KS = CW::KS, NFR = CW::NFR, NW = NTH / 64;
    const int r = lane & 15, q = lane >> 4;
    const int wd = ws - 2, npd = wd * wd, nps = ws * ws, wr = ws + 2;
    int off[KS];
    uint32_t vmask = 0;                                       // k steps in which this lane's K group exists (K padded to 32)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int kg = 4 * ks + q;
        const bool valid = kg < KG;
        const int tap = kg / CPG, plane = kg - tap * CPG;
        off[ks] = valid ? ((tap / 3) * ws + (tap % 3) + plane * nps) * 16 : 0;
        vmask |= valid ? 1u << ks : 0u;
    }
    const int ngroups = (npd + 15) >> 4;
    // TWO groups per trip, written as straight-line code (the second group of an odd tail repeats the last one with its stores
    // switched off): their LDS reads, MFMA chains and SiLU epilogues are independent, and with one wave per SIMD nothing else
    // hides the latency of one behind the other
    for (int g0 = wave; g0 < ngroups; g0 += 2 * NW) {
        int p_[2], py_[2], px_[2];
        bool live_[2];
        bf16x8 xb[2][KS];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int g = g0 + u * NW;
            live_[u] = g < ngroups;
            const int p = (live_[u] ? g : g0) * 16 + r;
            const int pc = p < npd ? p : npd - 1;
            p_[u] = p; py_[u] = pc / wd; px_[u] = pc - py_[u] * wd;
            const int sbase = (py_[u] * ws + px_[u]) * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                  // all fragments in flight before the first MFMA
                xb[u][ks] = *(const bf16x8*)(S + sbase + off[ks]);
                if constexpr ((KG & 3) != 0) { if (ks == KS - 1 && !((vmask >> ks) & 1)) xb[u][ks] = c2f_zero8(); }   // K padded to 32: last step only
            }
        }
        f32x4 acc[2][NFR];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int f = 0; f < NFR; ++f) acc[u][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int f = 0; f < NFR; ++f)
                    acc[u][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.wa[f][ks], xb[u][ks], acc[u][f], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = p_[u], py = py_[u], px = px_[u];
            const int iy = oy + py, ix = ox + px;
            const bool inimg = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
            for (int f = 0; f < NFR; ++f) {
                const int ch0 = 16 * f + 4 * q;
                float v[4] = {c2f_silu(acc[u][f][0] + w.bv[f].x), c2f_silu(acc[u][f][1] + w.bv[f].y), c2f_silu(acc[u][f][2] + w.bv[f].z),
                              c2f_silu(acc[u][f][3] + w.bv[f].w)};
                if constexpr (RES) {
                    const uint2 rr = *(const uint2*)(Rs + ((ch0 >> 3) * (wr * wr) + (py + 2) * wr + (px + 2)) * 16 + ((ch0 >> 2) & 1) * 8);
                    v[0] += bf16_to_f32((uint16_t)(rr.x & 0xffff)); v[1] += bf16_to_f32((uint16_t)(rr.x >> 16));
                    v[2] += bf16_to_f32((uint16_t)(rr.y & 0xffff)); v[3] += bf16_to_f32((uint16_t)(rr.y >> 16));
                }
                uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                if (!inimg) pk = make_uint2(0u, 0u);
                if (live_[u] && p < npd) *(uint2*)(D + ((ch0 >> 3) * npd + p) * 16 + ((ch0 >> 2) & 1) * 8) = pk;
            }
        }
    }
}

constexpr int C2F_NTH = 256;

template <int C, int NB, int TT>
__global__ __launch_bounds__(C2F_NTH) void c2f_fused_kernel(C2fArgs a) {
    using G = C2fGeo<C, NB, TT>;
    constexpr int T = TT, HALO = G::HALO, R0 = G::R0, CPG = C / 8, C1 = 2 * C, NTH = C2F_NTH, NW = NTH / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int H = a.H, W = a.W;
    const int oy = ty * T - HALO, ox = tx * T - HALO;          // image coordinates of pixel (0, 0) of the outermost region

    C2fConvW<C> wA, wB;                                        // the current and the next 3 x 3 layer's weights
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int sti = 0;
    auto stamp = [&]() __attribute__((always_inline)) { if (a.dbg) st_[sti] = __builtin_readcyclecounter(); ++sti; };
    stamp();
    // ---- cv1: x (global) -> y0 (centre) | y1 (whole region) ---------------------------------------------------------------
    {
        constexpr int NFR = C1 / 16, KS = C1 / 32, N0 = R0 * R0, NG = (N0 + 15) / 16, MAXG = (NG + NW - 1) / NW;
        // every input fragment of this wave's pixel groups first (one round trip to HBM / L2 for the whole phase, not one per group)
        bf16x8 xb[MAXG][KS];
#pragma unroll
        for (int i = 0; i < MAXG; ++i) {
            const int p = (wave + i * NW) * 16 + r;
            const int py = p / R0, px = p - py * R0;
            const int iy = oy + py, ix = ox + px;
            const bool inimg = p < N0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const uint16_t* src = a.x + (((long long)b * H + (inimg ? iy : 0)) * W + (inimg ? ix : 0)) * a.ldx + 8 * q;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xb[i][ks] = inimg ? *(const bf16x8*)(src + 32 * ks) : c2f_zero8();
        }
        bf16x8 wa[NFR][KS];
        float4 bv[NFR];
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wa[f][ks] = *(const bf16x8*)(a.w_cv1 + (size_t)(16 * f + r) * C1 + 32 * ks + 8 * q);
            bv[f] = *(const float4*)(a.b_cv1 + 16 * f + 4 * q);
        }
        c2f_load_conv_w<C>(wA, a.w_m[0], a.b_m[0], r, q);
        unsigned char* Y0 = smem + G::y_off(0);
        unsigned char* Y1 = smem + G::y_off(1);
#pragma unroll
        for (int i = 0; i < MAXG; ++i) {
            const int p = (wave + i * NW) * 16 + r;
            const int py = p / R0, px = p - py * R0;
            const int iy = oy + py, ix = ox + px;
            const bool inimg = p < N0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            f32x4 acc[NFR];
#pragma unroll
            for (int f = 0; f < NFR; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int f = 0; f < NFR; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[f][ks], xb[i][ks], acc[f], 0, 0, 0);
            const bool centre = py >= HALO && py < HALO + T && px >= HALO && px < HALO + T;
#pragma unroll
            for (int f = 0; f < NFR; ++f) {
                const int ch0 = 16 * f + 4 * q;
                uint2 pk = make_uint2(pack_bf16x2(c2f_silu(acc[f][0] + bv[f].x), c2f_silu(acc[f][1] + bv[f].y)),
                                      pack_bf16x2(c2f_silu(acc[f][2] + bv[f].z), c2f_silu(acc[f][3] + bv[f].w)));
                if (!inimg) pk = make_uint2(0u, 0u);
                if (ch0 < C) {
                    if (centre && p < N0) *(uint2*)(Y0 + ((ch0 >> 3) * (T * T) + (py - HALO) * T + (px - HALO)) * 16 + ((ch0 >> 2) & 1) * 8) = pk;
                } else if (p < N0) {
                    const int c0 = ch0 - C;
                    *(uint2*)(Y1 + ((c0 >> 3) * N0 + p) * 16 + ((c0 >> 2) & 1) * 8) = pk;
                }
            }
        }
    }
    // cv2's weights (A operands) and the geometry of its sources; loaded in front of the last barrier
    constexpr int NFR2 = C1 / 16, KG2 = (2 + NB) * CPG, KS2 = (KG2 + 3) / 4, KC2 = (2 + NB) * C;
    bf16x8 w2[NFR2][KS2];
    float4 bv2[NFR2];
    auto load_w2 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            const int kg = 4 * ks + q;
#pragma unroll
            for (int f = 0; f < NFR2; ++f)
                w2[f][ks] = kg < KG2 ? *(const bf16x8*)(a.w_cv2 + (size_t)(16 * f + r) * KC2 + 8 * kg) : c2f_zero8();
        }
#pragma unroll
        for (int f = 0; f < NFR2; ++f) bv2[f] = *(const float4*)(a.b_cv2 + 16 * f + 4 * q);
    };
    // ---- bottlenecks (the next layer's weights are requested in front of each barrier) ----------------------------------------
    c2f_load_conv_w<C>(wB, a.w_m[1], a.b_m[1], r, q);
    stamp();
    __syncthreads();
    stamp();
    c2f_conv3<C, NTH, false>(wA, smem + G::y_off(1), G::y_w(1), smem + G::t_off(0), nullptr, oy + 1, ox + 1, H, W, lane, wave);
    if constexpr (NB == 2) c2f_load_conv_w<C>(wA, a.w_m[2], a.b_m[2], r, q); else load_w2();
    stamp();
    __syncthreads();
    c2f_conv3<C, NTH, true>(wB, smem + G::t_off(0), G::t_w(0), smem + G::y_off(2), smem + G::y_off(1), oy + 2, ox + 2, H, W, lane, wave);
    if constexpr (NB == 2) {
        c2f_load_conv_w<C>(wB, a.w_m[3], a.b_m[3], r, q);
        __syncthreads();
        c2f_conv3<C, NTH, false>(wA, smem + G::y_off(2), G::y_w(2), smem + G::t_off(1), nullptr, oy + 3, ox + 3, H, W, lane, wave);
        load_w2();
        __syncthreads();
        c2f_conv3<C, NTH, true>(wB, smem + G::t_off(1), G::t_w(1), smem + G::y_off(3), smem + G::y_off(2), oy + 4, ox + 4, H, W, lane, wave);
    }
    stamp();
    __syncthreads();
    stamp();
    // ---- cv2: [y0 | y1 | .. | y(NB+1)] (centre pixels) -> out (global) --------------------------------------------------------
    {
        int cbase[KS2], cw[KS2];
        uint32_t vmask = 0;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            const int kg = 4 * ks + q;
            const bool valid = kg < KG2;
            vmask |= valid ? 1u << ks : 0u;
            const int arr = kg / CPG, plane = kg - arr * CPG;
            // geometry of y_arr: width, offset of the centre tile inside it
            int w_ = T, co = 0, base = G::y_off(0);
            if (arr == 1) { w_ = G::y_w(1); co = HALO; base = G::y_off(1); }
            else if (arr == 2) { w_ = G::y_w(2); co = HALO - 2; base = G::y_off(2); }
            else if (arr == 3) { w_ = G::y_w(NB + 1); co = HALO - 4; base = G::y_off(NB + 1); }
            cw[ks] = w_;
            cbase[ks] = valid ? base + (plane * w_ * w_ + co * w_ + co) * 16 : 0;
        }
        constexpr int NG2 = T * T / 16;
        for (int g0 = wave; g0 < NG2; g0 += 2 * NW) {             // two groups per trip (see c2f_conv3)
            bf16x8 xb[2][KS2];
            bool live_[2];
            int p_[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                live_[u] = g0 + u * NW < NG2;
                const int p = (live_[u] ? g0 + u * NW : g0) * 16 + r;
                p_[u] = p;
                const int py = p / T, px = p - py * T;
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) {
                    xb[u][ks] = *(const bf16x8*)(smem + cbase[ks] + (py * cw[ks] + px) * 16);
                    if constexpr ((KG2 & 3) != 0) { if (ks == KS2 - 1 && !((vmask >> ks) & 1)) xb[u][ks] = c2f_zero8(); }
                }
            }
            f32x4 acc[2][NFR2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int f = 0; f < NFR2; ++f) acc[u][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int f = 0; f < NFR2; ++f)
                        acc[u][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[f][ks], xb[u][ks], acc[u][f], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int p = p_[u];
                const int py = p / T, px = p - py * T;
                const int iy = ty * T + py, ix = tx * T + px;
                if (live_[u] && iy < H && ix < W) {
                    uint16_t* dst = a.out + (((long long)b * H + iy) * W + ix) * a.ldo + 4 * q;
#pragma unroll
                    for (int f = 0; f < NFR2; ++f)
                        *(uint2*)(dst + 16 * f) = make_uint2(pack_bf16x2(c2f_silu(acc[u][f][0] + bv2[f].x), c2f_silu(acc[u][f][1] + bv2[f].y)),
                                                             pack_bf16x2(c2f_silu(acc[u][f][2] + bv2[f].z), c2f_silu(acc[u][f][3] + bv2[f].w)));
                }
            }
        }
    }
    stamp();
    if (a.dbg && lane == 0 && wave == 0 && blockIdx.x < 64) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.dbg[blockIdx.x * 8 + i] = st_[i];
    }
}

template <int C, int NB, int TT>
int launch_c2f(C2fArgs& a, hipStream_t st) {
    constexpr int lds = C2fGeo<C, NB, TT>::BYTES;
    static_assert(lds <= 160 * 1024, "LDS");
    static std::atomic<int> granted[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return YV_ERR_LAUNCH;
    if (lds > 65536 && !granted[dev].load(std::memory_order_relaxed)) {
        if (hipFuncSetAttribute((const void*)c2f_fused_kernel<C, NB, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return YV_ERR_LAUNCH;
        granted[dev].store(1, std::memory_order_relaxed);
    }
    a.tiles_x = (a.W + TT - 1) / TT; a.tiles_y = (a.H + TT - 1) / TT;
    const long long grid = (long long)a.tiles_x * a.tiles_y * a.B;
    if (grid > 0x7fffffffLL) return YV_ERR_LIMIT;
    hipLaunchKernelGGL((c2f_fused_kernel<C, NB, TT>), dim3((unsigned)grid), dim3(C2F_NTH), lds, st, a);
    return yv_launch_status();
}
unsigned long long* g_c2f_dbg = nullptr;

}  // namespace

// diagnostics: cycle stamps of the next launches (device buffer of 64 x 8 uint64; NULL switches them off)
extern "C" int yv_c2f_debug(void* buf) { g_c2f_dbg = (unsigned long long*)buf; return YV_OK; }

extern "C" int yv_c2f_fused(const void* x, long long ldx, int B, int H, int W, int c, int n, const void* w_cv1, const float* b_cv1,
                            const void* const* w_m, const float* const* b_m, const void* w_cv2, const float* b_cv2, void* out,
                            long long ldo, void* stream) {
    if (!x || !w_cv1 || !b_cv1 || !w_m || !b_m || !w_cv2 || !b_cv2 || !out || B <= 0 || H <= 0 || W <= 0) return YV_ERR_ARG;
    if ((c != 16 && c != 32) || (n != 1 && n != 2)) return YV_ERR_ARG;
    if (ldx < 2 * c || ldo < 2 * c || (ldx & 7) || (ldo & 3) || (((uintptr_t)x | (uintptr_t)w_cv1 | (uintptr_t)w_cv2) & 15) ||
        ((uintptr_t)out & 7) || (((uintptr_t)b_cv1 | (uintptr_t)b_cv2) & 15))
        return YV_ERR_ARG;
    C2fArgs a = {};
    a.x = (const uint16_t*)x; a.ldx = ldx; a.B = B; a.H = H; a.W = W;
    a.w_cv1 = (const uint16_t*)w_cv1; a.b_cv1 = b_cv1; a.w_cv2 = (const uint16_t*)w_cv2; a.b_cv2 = b_cv2;
    for (int j = 0; j < 2 * n; ++j) {
        if (!w_m[j] || !b_m[j] || ((uintptr_t)w_m[j] & 15) || ((uintptr_t)b_m[j] & 15)) return YV_ERR_ARG;
        a.w_m[j] = (const uint16_t*)w_m[j]; a.b_m[j] = b_m[j];
    }
    a.out = (uint16_t*)out; a.ldo = ldo; a.dbg = g_c2f_dbg;
    hipStream_t st = (hipStream_t)stream;
    if (c == 16) return n == 1 ? launch_c2f<16, 1, 16>(a, st) : launch_c2f<16, 2, 16>(a, st);
    // (8 x 8 tiles for c = 32 - 53 KB of LDS - were tried: the instance needs 300 registers, so still one workgroup per CU, and the
    // halo work grows: 125 us instead of 104 for model.4 at batch 32)
    return n == 1 ? launch_c2f<32, 1, 16>(a, st) : launch_c2f<32, 2, 16>(a, st);
}
