// Dev tool (GPU box): stand-alone lab for the classifier GEMMs - no torch, starts in a second.
//   build:  make -C tools gemm_lab        (hipcc, gfx950; the binary travels with the snapshot)
//   run:    tools/build/gemm_lab [mode ...]
// It includes the product translation unit so that every kernel / launcher of csrc/gemm.hip is visible, times variants
// interleaved in one process on the ViT-B/16 shapes of the bench (random data), compares outputs bit for bit against the
// shipped kernel, and runs the DIAG instance of gemm_p8_kernel (per-segment cycle sums).
#include "../yolov8-vit_amd/csrc/gemm.hip"
#include <vector>
#include <string>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }
static inline float rndf() { return (rnd() >> 8) * (1.0f / 8388608.0f) - 1.0f; }           // [-1, 1)
static inline uint16_t h_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

struct Problem {
    const char* name; int M, N, K, flags;
    uint16_t *A, *W; float* bias; void *out, *ref; size_t out_bytes;
};

static Problem make_problem(const char* name, int M, int N, int K, int flags) {
    Problem p{name, M, N, K, flags};
    std::vector<uint16_t> a((size_t)M * K), w((size_t)N * K);
    for (auto& v : a) v = h_bf16(rndf());
    for (auto& v : w) v = h_bf16(rndf() * 0.08f);
    std::vector<float> b(N);
    for (auto& v : b) v = rndf();
    CK(hipMalloc(&p.A, a.size() * 2)); CK(hipMalloc(&p.W, w.size() * 2)); CK(hipMalloc(&p.bias, N * 4));
    CK(hipMemcpy(p.A, a.data(), a.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(p.W, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(p.bias, b.data(), N * 4, hipMemcpyHostToDevice));
    p.out_bytes = (size_t)M * N * ((flags & (YV_EPI_RES_F32 | YV_EPI_OUT_F32)) ? 4 : 2);
    CK(hipMalloc(&p.out, p.out_bytes)); CK(hipMalloc(&p.ref, p.out_bytes));
    return p;
}

static GemmArgs args_of(const Problem& p, void* out) {
    GemmArgs g = {};
    g.a0 = p.A; g.lda0 = p.K; g.c0 = p.K; g.w = p.W; g.bias = p.bias; g.M = p.M; g.N = p.N; g.K = p.K;
    g.out = out; g.ldo = p.N; g.flags = p.flags | YV_EPI_BIAS; g.m_mul = 1; g.ksize = 1; g.stride = 1;
    g.staged = 1; g.group_m = 8; g.splitk = 1;
    return g;
}

typedef std::function<int(GemmArgs&, hipStream_t)> Launcher;

static float time_us(const Problem& p, const Launcher& fn, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    GemmArgs g = args_of(p, p.out);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) { GemmArgs gg = g; if (fn(gg, 0) != YV_OK) { printf("launch failed\n"); exit(1); } }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1000.0f / iters;
}

static size_t mismatches(const Problem& p, bool verbose = false) {          // p.out vs p.ref, bytewise on host
    std::vector<unsigned char> a(p.out_bytes), b(p.out_bytes);
    CK(hipMemcpy(a.data(), p.out, p.out_bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), p.ref, p.out_bytes, hipMemcpyDeviceToHost));
    const bool f32 = p.flags & (YV_EPI_RES_F32 | YV_EPI_OUT_F32);
    const size_t es = f32 ? 4 : 2;
    size_t bad = 0; double maxd = 0; int shown = 0;
    std::vector<size_t> rowhist(256, 0), colhist(256, 0);
    for (size_t i = 0; i < p.out_bytes; i += es) {
        if (memcmp(&a[i], &b[i], es) == 0) continue;
        ++bad;
        const size_t e = i / es, r = e / p.N, c = e % p.N;
        rowhist[r & 255]++; colhist[c & 255]++;
        if (f32) {
            float x, y; memcpy(&x, &a[i], 4); memcpy(&y, &b[i], 4);
            maxd = std::max(maxd, (double)fabsf(x - y));
            if (verbose && shown < 6) { printf("     (%zu, %zu): got %.9g ref %.9g\n", r, c, x, y); ++shown; }
        }
    }
    if (bad && verbose) {
        printf("     max abs diff %.3g; mismatches by row %% 256 (16-row bins):", maxd);
        for (int k = 0; k < 16; ++k) { size_t s_ = 0; for (int q = 0; q < 16; ++q) s_ += rowhist[k * 16 + q]; printf(" %zu", s_); }
        printf("\n     by col %% 256 (16-col bins):");
        for (int k = 0; k < 16; ++k) { size_t s_ = 0; for (int q = 0; q < 16; ++q) s_ += colhist[k * 16 + q]; printf(" %zu", s_); }
        printf("\n");
    }
    return bad;
}

static void fill_out(const Problem& p, void* out) {   // residual stream start value (same for every variant)
    if (p.flags & YV_EPI_RES_F32) {
        std::vector<float> x((size_t)p.M * p.N);
        uint64_t s = rng_state; rng_state = 12345;
        for (auto& v : x) v = rndf();
        rng_state = s;
        CK(hipMemcpy(out, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    } else CK(hipMemset(out, 0, p.out_bytes));
}

struct Variant { std::string name; Launcher fn; };

template <int MF0, int MF1, bool F32OUT>
static void run_diag(const Problem& p, int n_cu) {
    GemmArgs g = args_of(p, p.out);
    uint32_t* dbg; const size_t n = (size_t)n_cu * 8 * 16;
    CK(hipMalloc(&dbg, n * 4)); CK(hipMemset(dbg, 0, n * 4));
    g.partial = (float*)dbg; g.sched = g_opt_p8_sched;
    constexpr int BM = 32 * (MF0 + MF1);
    g.tiles_m = (g.M + BM - 1) / BM; g.tiles_n = g.N / 256;
    const size_t lds = 2 * 65536 + 8 * 2048 + 16384;
    auto kern = gemm_p8_kernel<MF0, MF1, F32OUT, 1>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles = g.tiles_m * g.tiles_n, grid = tiles < n_cu ? tiles : n_cu;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, g);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(n);
    CK(hipMemcpy(h.data(), dbg, n * 4, hipMemcpyDeviceToHost));
    const char* nm[9] = {"issue", "wait123", "wait4", "barA", "mma", "barB", "epilogue", "phases", "total"};
    for (int grp = 0; grp < 2; ++grp) {
        double s[9] = {0}; int cnt = 0;
        for (int b = 0; b < grid; ++b)
            for (int w = grp * 4; w < grp * 4 + 4; ++w) {
                const uint32_t* o = &h[((size_t)b * 8 + w) * 16];
                if (!o[7]) continue;
                for (int i = 0; i < 9; ++i) s[i] += o[i];
                ++cnt;
            }
        if (!cnt) continue;
        const double ph = s[7] / cnt;
        printf("  diag %s <%d,%d> group %d: phases/wave %.0f total %.0f cyc | per phase: issue %.0f wait123 %.0f wait4(x4) %.0f barA %.0f mma %.0f barB %.0f | epilogue/tile %.0f\n",
               p.name, MF0, MF1, grp, ph, s[8] / cnt, s[0] / s[7], s[1] / s[7] * 4 / 3, s[2] / s[7] * 4, s[3] / s[7], s[4] / s[7], s[5] / s[7],
               s[6] / cnt / (ph / 4 / (p.K / 64)));
        (void)nm;
    }
    CK(hipFree(dbg));
}

template <int MF, bool F32OUT>
static void run_diag9(const Problem& p, int n_cu) {
    GemmArgs g = args_of(p, p.out);
    uint32_t* dbg; const size_t n = (size_t)n_cu * 8 * 16;
    CK(hipMalloc(&dbg, n * 4)); CK(hipMemset(dbg, 0, n * 4));
    g.partial = (float*)dbg; g.sched = g_opt_p8_sched;
    constexpr int BM = 32 * MF;
    g.tiles_m = (g.M + BM - 1) / BM; g.tiles_n = g.N / 256;
    const size_t lds = 2 * 65536 + 16384;
    auto kern = gemm_p9_kernel<MF, F32OUT, 1>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles = g.tiles_m * g.tiles_n, grid = tiles < n_cu ? tiles : n_cu;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, g);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(n);
    CK(hipMemcpy(h.data(), dbg, n * 4, hipMemcpyDeviceToHost));
    double s[8] = {0}; int cnt = 0;
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < 8; ++w) {
            const uint32_t* o = &h[((size_t)b * 8 + w) * 16];
            if (!o[5]) continue;
            for (int i = 0; i < 8; ++i) s[i] += o[i];
            ++cnt;
        }
    if (cnt) {
        const double tiles_w = s[5] / cnt, nk = p.K / 64;
        printf("  diag9 %s MF %d: tiles/wave %.2f total %.0f cyc | per tile: main %.0f (%.0f per K tile; MFMA floor %d) epilogue %.0f | per sync: wait %.0f barrier %.0f | first sync of a tile: wait %.0f\n",
               p.name, MF, tiles_w, s[7] / cnt, s[2] / s[5], s[2] / s[5] / nk, MF * 4 * 2 * 16 * 2, s[3] / s[5], s[0] / s[6], s[1] / s[6], s[4] / s[5]);
    }
    CK(hipFree(dbg));
}

int main(int argc, char** argv) {
    const int M = getenv("LAB_M") ? atoi(getenv("LAB_M")) : 25216;
    const int iters = 10, rounds = 5;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, M = %d\n", prop.name, n_cu, M);
    std::vector<Problem> ps = {make_problem("qkv", M, 2304, 768, 0), make_problem("proj", M, 768, 768, YV_EPI_RES_F32),
                               make_problem("fc1", M, 3072, 768, YV_EPI_GELU), make_problem("fc2", M, 768, 3072, YV_EPI_RES_F32)};
    if (getenv("LAB_CAL")) {
        // calibration of the FETCH_SIZE counter on THIS kernel's access pattern (run under rocprofv3 --pmc FETCH_SIZE): one column
        // tile, so no activation row has a second reader - the kernel must fetch A exactly once (+ 0.4 MB of W per XCD)
        ps.clear();
        ps.push_back(make_problem("cal256", M, 256, 768, 0));
        ps.push_back(make_problem("cal512", M, 512, 768, 0));
    }
    if (getenv("LAB_TRAIN")) {                                  // the trainer's data-gradient products with 768 outputs (bf16)
        ps.push_back(make_problem("dqkv", M, 768, 2304, 0));
        ps.push_back(make_problem("dfc1", M, 768, 3072, 0));
        ps.push_back(make_problem("dproj", M, 768, 768, 0));
    }
    std::vector<Variant> vs;
    vs.push_back({"p8 (shipped)", [](GemmArgs& g, hipStream_t st) { return launch_p8(g, st); }});
    if (getenv("LAB_R1")) vs.push_back({"dma128 (round 1)", [](GemmArgs& g, hipStream_t st) { return launch_dma<128, 128, 2, 2>(g, st); }});
    vs.push_back({"p9 auto", [](GemmArgs& g, hipStream_t st) { return launch_p9(g, st, 0); }});
    vs.push_back({"p9 256", [](GemmArgs& g, hipStream_t st) { return launch_p9(g, st, 256); }});
    vs.push_back({"p9 224", [](GemmArgs& g, hipStream_t st) { return launch_p9(g, st, 224); }});
    vs.push_back({"p9 192", [](GemmArgs& g, hipStream_t st) { return launch_p9(g, st, 192); }});
    vs.push_back({"p9 160", [](GemmArgs& g, hipStream_t st) { return launch_p9(g, st, 160); }});
    vs.push_back({"p9 128", [](GemmArgs& g, hipStream_t st) { return launch_p9(g, st, 128); }});
    vs.push_back({"p9 96", [](GemmArgs& g, hipStream_t st) { return launch_p9(g, st, 96); }});
    double tot_fl = 0; std::vector<double> tot_us(vs.size(), 0.0);
    for (auto& p : ps) {
        // reference = shipped kernel
        fill_out(p, p.ref);
        { GemmArgs g = args_of(p, p.ref); if (vs[0].fn(g, 0) != YV_OK) { printf("ref launch failed\n"); return 1; } }
        CK(hipDeviceSynchronize());
        std::vector<std::vector<float>> t(vs.size());
        for (size_t v = 0; v < vs.size(); ++v) {
            fill_out(p, p.out);
            { GemmArgs g = args_of(p, p.out); if (vs[v].fn(g, 0) != YV_OK) { printf("launch failed\n"); return 1; } }
            CK(hipDeviceSynchronize());
            const size_t bad = mismatches(p, true);
            if (bad) printf("  !! %s %s: %zu of %zu outputs differ from the shipped kernel\n", p.name, vs[v].name.c_str(), bad, (size_t)p.M * p.N);
        }
        for (int r = 0; r < rounds; ++r)
            for (size_t v = 0; v < vs.size(); ++v) t[v].push_back(time_us(p, vs[v].fn, iters));
        const double fl = 2.0 * p.M * p.N * p.K;
        tot_fl += fl;
        for (size_t v = 0; v < vs.size(); ++v) {
            std::sort(t[v].begin(), t[v].end());
            const double med = t[v][rounds / 2];
            tot_us[v] += med;
            printf("%-5s %-24s median %7.1f us  %7.1f TF/s  frac %.3f  (min %.1f us)\n", p.name, vs[v].name.c_str(), med, fl / med / 1e6,
                   fl / med / 1e6 / 2500.0, t[v][0]);
        }
        fflush(stdout);
    }
    for (size_t v = 0; v < vs.size(); ++v)
        printf("LAYER %-24s %7.1f us  flop-weighted frac %.3f\n", vs[v].name.c_str(), tot_us[v], tot_fl / tot_us[v] / 1e6 / 2500.0);
    if (!getenv("LAB_NO_DIAG") && !getenv("LAB_TRAIN") && !getenv("LAB_CAL")) {
        run_diag9<7, false>(ps[0], n_cu);
        run_diag9<5, true>(ps[1], n_cu);
        run_diag9<8, false>(ps[2], n_cu);
        run_diag9<5, true>(ps[3], n_cu);
    }
    if (getenv("LAB_DIAG8")) {
        run_diag<4, 3, false>(ps[0], n_cu);
        run_diag<4, 4, false>(ps[0], n_cu);
        run_diag<3, 2, true>(ps[1], n_cu);
        run_diag<4, 4, false>(ps[2], n_cu);
        run_diag<3, 2, true>(ps[3], n_cu);
    }
    return 0;
}
