// Isolated timing of potrf32 / fwdsub32 (one wave, LDS-resident), cycles via s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../eqf_vio_amd/csrc/eqf_update.hpp"
using namespace eqf;
__global__ __launch_bounds__(256) void k_bench(double* out, int reps, int waves) {
    __shared__ double sK[kNB][kLdsP];
    __shared__ __attribute__((aligned(16))) double sLT[kNB][kLtP];
    __shared__ double sRd[kNB];
    __shared__ double sP[kNB][kLdsP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < kNB * kNB; e += blockDim.x) {
        const int r = e / kNB, c = e % kNB;
        sK[r][c] = (r == c ? 40.0 : 0.0) + 1.0 / (1 + r + c);
        sP[r][c] = 1.0 + 0.01 * r - 0.02 * c;
    }
    __syncthreads();
    int bad = 0;
    long long t0 = 0, t1 = 0, t2 = 0;
    if (wv < waves) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < reps; ++i) potrf32(sK, sLT, sRd, lane, &bad);
        t1 = __builtin_readcyclecounter();
        double x[kNB];
        for (int j = 0; j < kNB; ++j) x[j] = sP[lane & 31][j];
        for (int i = 0; i < reps; ++i) {
            fwdsub32(sLT, sRd, x);
            for (int j = 0; j < kNB; ++j) { x[j] = x[j] * 3.0 + 1.0; __asm__ volatile("" : "+v"(x[j])); }
        }
        t2 = __builtin_readcyclecounter();
        out[8 + lane] = x[3] + bad;
    }
    if (tid == 0) {
        out[0] = double(t1 - t0) / reps;
        out[1] = double(t2 - t1) / reps;
        out[2] = sLT[3][5];
    }
}
int main() {
    double* o; hipMalloc(&o, 1024);
    double h[3];
    for (int waves : {1, 4}) {
        hipLaunchKernelGGL(k_bench, dim3(1), dim3(256), 0, 0, o, 200, waves);
        hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
        printf("waves=%d potrf32: %.0f cycles (%.2f us)   fwdsub32: %.0f cycles (%.2f us)   check %.6f\n", waves, h[0], h[0] / 2400.0, h[1], h[1] / 2400.0, h[2]);
    }
    return 0;
}
