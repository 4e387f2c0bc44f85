// Which physical CUs a hipExtStreamCreateWithCUMask bit enables on this part (stand-alone: hipcc --offload-arch=gfx950 -O3).
// A probe kernel of many short workgroups records (XCC_ID, SE, SH, CU) of every workgroup; masks with single bits / runs of bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
#include <map>
#include <string>

__global__ void probe(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
}

int main() {
    const int NB = 8192;
    uint32_t* d; hipMalloc(&d, NB * 8);
    std::vector<uint32_t> h(NB * 2);
    auto run = [&](const std::vector<int>& bits, const char* label) {
        uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b : bits) words[b >> 5] |= 1u << (b & 31);
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, 8, words) != hipSuccess) { printf("%s: create failed\n", label); return; }
        hipMemsetAsync(d, 0xff, NB * 8, st);
        hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, st, d, 20000);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost);
        std::map<int, std::set<int>> per_xcc;
        for (int i = 0; i < NB; ++i) {
            const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
        }
        int total = 0;
        std::string s;
        for (auto& kv : per_xcc) {
            total += (int)kv.second.size();
            char buf[64]; snprintf(buf, sizeof buf, " xcc%d:%zu", kv.first, kv.second.size()); s += buf;
        }
        printf("%-34s -> %3d distinct (xcc,se,sh,cu) |%s\n", label, total, s.c_str());
        if (bits.size() == 1) {            // the one XCC the bit restricts (every other XCC has an all-zero mask = unrestricted)
            for (auto& kv : per_xcc) if (kv.second.size() == 1) for (int v : kv.second) printf("      -> xcc %d se %d cu %d\n", kv.first, v >> 5, v & 15);
        }
        hipStreamDestroy(st);
    };
    std::vector<int> all; for (int i = 0; i < 256; ++i) all.push_back(i);
    run(all, "all 256 bits");
    for (int b : {0, 1, 2, 7, 8, 9, 16, 32, 33, 64, 128, 255}) { char l[32]; snprintf(l, sizeof l, "bit %d", b); run({b}, l); }
    { std::vector<int> v; for (int i = 0; i < 64; ++i) v.push_back(i); run(v, "bits 0..63"); }
    { std::vector<int> v; for (int i = 0; i < 32; ++i) v.push_back(i); run(v, "bits 0..31"); }
    { std::vector<int> v; for (int i = 0; i < 256; i += 4) v.push_back(i); run(v, "every 4th bit (64)"); }
    { std::vector<int> v; for (int i = 0; i < 256; i += 8) v.push_back(i); run(v, "every 8th bit (32)"); }
    { std::vector<int> v; for (int i = 0; i < 256; i += 2) v.push_back(i); run(v, "every 2nd bit (128)"); }
    { std::vector<int> v; for (int i = 0; i < 256; ++i) if (i % 8 < 2) v.push_back(i); run(v, "bits i%8<2 (64)"); }
    { std::vector<int> v; for (int i = 0; i < 256; ++i) if ((i / 8) % 4 == 0) v.push_back(i); run(v, "bits (i/8)%4==0 (64)"); }
    return 0;
}
