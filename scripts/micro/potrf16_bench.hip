// Isolated timing of potrf16 variants (one wave, 64-row panel in LDS), cold (first call) and warm (loop) cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../eqf_vio_amd/csrc/eqf_chol64.hpp"
using namespace eqf;

template <int MODE>  // 0: full (readlane bulk)  1: pivot chain only  2: one Newton step  3: LDS-broadcast bulk
__device__ __forceinline__ void variant(double (*T)[kSP], double (*colS)[kQB + 2], int base, int lane, int* bad) {
    double row[kQB];
#pragma unroll
    for (int c = 0; c < kQB; ++c) row[c] = T[lane][base + c];
    double ljPrev = 0.0;
    const bool inDiag = lane >= base && lane < base + kQB;
#pragma unroll
    for (int c = 0; c < kQB; ++c) {
        double colPrev[kQB];
        if (MODE == 3 && c >= 1) {
#pragma unroll
            for (int c2 = c + 1; c2 < kQB; ++c2) colPrev[c2] = colS[c - 1][c2];
        }
        const double d = readlane64(row[c], base + c);
        if (!(d > 0.0)) *bad = 1;
        double rd;
        if (MODE == 2) {
            double y = __builtin_amdgcn_rsq(d);
            rd = y * (1.5 - 0.5 * d * y * y);
        } else rd = rsqrtPivot(d);
        if (c >= 1 && MODE != 1) {
#pragma unroll
            for (int c2 = c + 1; c2 < kQB; ++c2)
                row[c2] = fma(-ljPrev, MODE == 3 ? colPrev[c2] : readlane64(ljPrev, base + c2), row[c2]);
        }
        const double lj = row[c] * rd;
        if (c + 1 < kQB) row[c + 1] = fma(-lj, readlane64(lj, base + c + 1), row[c + 1]);
        if (MODE == 3 && inDiag) colS[c][lane - base] = lj;
        row[c] = lj;
        ljPrev = lj;
#pragma unroll
        for (int c2 = c + 1; c2 < kQB; ++c2) __asm__ volatile("" : "+v"(row[c2]));
    }
    if (lane >= base) {
#pragma unroll
        for (int c = 0; c < kQB; ++c) T[lane][base + c] = (lane - base >= c) ? row[c] : 0.0;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_bench(double* out, int reps) {
    __shared__ double T[kSB][kSP];
    __shared__ double T0[kSB][kSP];
    __shared__ __attribute__((aligned(16))) double colS[kQB][kQB + 2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < kSB * kSB; e += blockDim.x) {
        const int r = e / kSB, c = e % kSB;
        T0[r][c] = (r == c ? 40.0 : 0.0) + 1.0 / (1 + r + c);
    }
    __syncthreads();
    int bad = 0;
    if (wv == 0) {
        for (int c = 0; c < kSB; ++c) T[lane][c] = T0[lane][c];
        long long t0 = __builtin_readcyclecounter();
        variant<MODE>(T, colS, 16, lane, &bad);
        long long t1 = __builtin_readcyclecounter();
        for (int i = 0; i < reps; ++i) {
            for (int c = 0; c < kQB; ++c) T[lane][16 + c] = T0[lane][16 + c];
            variant<MODE>(T, colS, 16, lane, &bad);
        }
        long long t2 = __builtin_readcyclecounter();
        if (lane == 0) {
            out[0] = double(t1 - t0);
            out[1] = double(t2 - t1) / reps;
            out[2] = T[40][20] + bad;
        }
    }
}
template <int MODE>
void run(const char* name, double* o) {
    double h[3];
    hipLaunchKernelGGL(k_bench<MODE>, dim3(1), dim3(256), 0, 0, o, 100);
    hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
    printf("%-28s cold %6.0f cycles   warm %6.0f cycles (%.2f us)   check %.9f\n", name, h[0], h[1], h[1] / 2400.0, h[2]);
}
int main() {
    double* o;
    hipMalloc(&o, 1024);
    run<0>("readlane bulk (product)", o);
    run<1>("pivot chain only", o);
    run<2>("readlane bulk, 1 Newton step", o);
    run<3>("LDS-broadcast bulk", o);
    run<0>("readlane bulk again", o);
    return 0;
}
