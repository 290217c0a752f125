// Which CUs does bit i of a hipExtStreamCreateWithCUMask mask enable on gfx950?  Launches 2048 workgroups on streams with a few masks and prints,
// per mask, how many workgroups ran on each XCC (HW_REG_XCC_ID) and how many distinct (XCC, SE, CU) places were used.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x] = ((xcc & 15) << 24) | (hw & 0xffffff);
        // keep the CU busy a little so that the grid spreads
        long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 300) {}
    }
}
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, words = (cus + 31) / 32, G = 2048;
    unsigned* d; CK(hipMalloc(&d, 4 * G));
    struct M { const char* name; int first, num, stride; };
    M ms[] = {{"bits 0..7", 0, 8, 1}, {"bits 0..31", 0, 32, 1}, {"bits 32..63", 32, 32, 1}, {"bits 0,8,16,..,248 (32 bits)", 0, 32, 8}, {"bits 0..47", 0, 48, 1}, {"every bit (256)", 0, 256, 1}};
    for (auto& m : ms) {
        std::vector<uint32_t> mask(words, 0u);
        for (int i = 0; i < m.num; ++i) { int c = m.first + i * m.stride; if (c < cus) mask[c / 32] |= 1u << (c % 32); }
        hipStream_t st; CK(hipExtStreamCreateWithCUMask(&st, words, mask.data()));
        CK(hipMemsetAsync(d, 0, 4 * G, st));
        hipLaunchKernelGGL(k, dim3(G), dim3(64), 0, st, d);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned> h(G); CK(hipMemcpy(h.data(), d, 4 * G, hipMemcpyDeviceToHost));
        int perX[16] = {0}; std::set<unsigned> places;
        for (unsigned v : h) { perX[v >> 24]++; places.insert(((v >> 24) << 16) | ((v >> 8) & 0xf) << 8 | ((v >> 13) & 0x7) << 12 | 0); }
        std::set<unsigned> raw; for (unsigned v : h) raw.insert(((v >> 24) << 24) | (v & 0x00ffff00));
        printf("%-32s workgroups per XCC:", m.name);
        for (int x = 0; x < 8; ++x) printf(" %4d", perX[x]);
        printf("   distinct (XCC, HW_ID[8..23]) places: %zu\n", raw.size());
        CK(hipStreamDestroy(st));
    }
    return 0;
}
