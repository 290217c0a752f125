// Microbenchmark: effective shader clock in a sparse-launch regime and latency of dependent fp64 chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void k_clock(long long* out) {
    long long c0 = __builtin_readcyclecounter();   // s_memtime: shader clock
    long long w0 = wall_clock64();                 // constant-rate counter
    double x = threadIdx.x * 1e-3 + 1.0;
    for (int i = 0; i < 20000; ++i) x = fma(x, 1.0000001, 1e-9);
    long long c1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
template <int MODE>
__global__ void k_chain(double* out, int n) {
    double x = threadIdx.x * 1e-3 + 1.0, y = x + 0.5, z = x + 0.25, w = x + 0.125;
    long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) { x = fma(x, 1.0000001, 1e-9); }                      // dependent fma f64
        if (MODE == 1) { x = fma(x, 1.0000001, 1e-9); y = fma(y, 1.0000001, 1e-9); z = fma(z, 1.0000001, 1e-9); w = fma(w, 1.0000001, 1e-9); }  // 4 independent
        if (MODE == 2) { x = __builtin_amdgcn_rsq(x) + 1.0; }                // rsq + add dependent
        if (MODE == 3) { int lo = __builtin_amdgcn_readlane(__double2loint(x), 3); int hi = __builtin_amdgcn_readlane(__double2hiint(x), 3);
                         x = fma(x, 0.5, __hiloint2double(hi, lo) * 0.25); }  // readlane -> fma dependent
        if (MODE == 4) { x = x * (1.0f / 3.0f) + 1.0; x = sqrt(x); }          // sqrt f64 chain
        if (MODE == 5) { x = 1.0 / x + 1.0; }                                 // div f64 chain
        if (MODE == 6) { float f = (float)x; f = fmaf(f, 1.0001f, 1e-6f); x = f; }
    }
    long long c1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x + y + z + w;
    if (threadIdx.x == 0) out[64] = (double)(c1 - c0) / n;
}
__global__ void k_lds(double* out, int n) {
    __shared__ double s[64];
    double x = threadIdx.x;
    s[threadIdx.x] = x;
    long long c0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        s[threadIdx.x] = x;
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        x = s[(threadIdx.x + 1) & 63] + 1.0;
    }
    long long c1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) out[64] = (double)(c1 - c0) / n;
}
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

int main() {
    long long* d; CK(hipMalloc(&d, 64)); double* o; CK(hipMalloc(&o, 1024));
    long long h[3];
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, st, d);
        CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
        printf("single-wave kernel: %lld shader cycles over %lld wall ticks (100 MHz) -> %.0f MHz, %.2f cycles per dependent fma\n",
               h[0], h[1], h[0] / (h[1] / 100.0), h[0] / 20000.0);
    }
    const char* names[] = {"dep fma f64", "4 indep fma f64 (per 4)", "rsq+add f64", "readlane x2 + mul + fma", "mul+sqrt f64", "div+add f64", "cvt+fmaf+cvt"};
    double hb[65];
#define RUN(M) hipLaunchKernelGGL(k_chain<M>, dim3(1), dim3(64), 0, st, o, 4000); CK(hipMemcpy(hb, o, sizeof(hb), hipMemcpyDeviceToHost)); printf("%-28s %.1f cycles/iter\n", names[M], hb[64]);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, st, o, 4000); CK(hipMemcpy(hb, o, sizeof(hb), hipMemcpyDeviceToHost));
    printf("LDS write+wait+read+add       %.1f cycles/iter\n", hb[64]);
    // launch overheads: back-to-back tiny kernels, events
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {1, 256, 600, 2048}) {
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, (int*)nullptr);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel grid %4d: %.2f us per launch (back-to-back, same stream)\n", grid, ms * 1000 / 200);
    }
    return 0;
}
