// How fast can 64 x 64 fp64 tiles be streamed (read + written back) when a tile is 64 row pieces of 512 bytes, 5 KB apart (the
// row-major chain matrices of a batch of filters), against tiles stored contiguously (32 KB each)?  512 persistent workgroups,
// the next tile's loads in flight while the current one is "computed" (a dependent FMA chain of adjustable length).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <bool TILED>
__global__ __launch_bounds__(256, 2) void k_stream(double* base, int nFilters, int nbt, int ld, long long strideF, int work, int passes) {
    const int tid = threadIdx.x, pr = tid >> 5, pc = 2 * (tid & 31);
    const long long tiles = (long long)nFilters * nbt * nbt;
    for (int p = 0; p < passes; ++p) {
        f64x2 cur[8], nxt[8];
        long long t = blockIdx.x;
        auto addr = [&](long long tt, int q) -> double* {
            const int f = int(tt / (nbt * nbt)), rc = int(tt % (nbt * nbt)), R = rc / nbt, C = rc % nbt;
            if (TILED) return base + f * strideF + ((long long)(R * nbt + C) * 64 + pr + 8 * q) * 64 + pc;
            return base + f * strideF + (long long)(R * 64 + pr + 8 * q) * ld + C * 64 + pc;
        };
        if (t < tiles)
#pragma unroll
            for (int q = 0; q < 8; ++q) nxt[q] = *reinterpret_cast<f64x2*>(addr(t, q));
        while (t < tiles) {
#pragma unroll
            for (int q = 0; q < 8; ++q) cur[q] = nxt[q];
            const long long tn = t + gridDim.x;
            if (tn < tiles)
#pragma unroll
                for (int q = 0; q < 8; ++q) nxt[q] = *reinterpret_cast<f64x2*>(addr(tn, q));
            __builtin_amdgcn_sched_barrier(0);
            for (int w = 0; w < work; ++w)
#pragma unroll
                for (int q = 0; q < 8; ++q) cur[q] = cur[q] * 1.0000001 + 1e-9;
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<f64x2*>(addr(t, q)) = cur[q];
            t = tn;
        }
    }
}

int main() {
    const int nF = 192, nbt = 10, ld = 640;  // 630 MB: beyond the 256 MB MALL
    const long long strideF = (long long)ld * ld;
    double* d;
    hipMalloc(&d, sizeof(double) * strideF * nF);
    hipMemset(d, 0, sizeof(double) * strideF * nF);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int work : {0, 50, 200}) {
        for (int tiled = 0; tiled < 2; ++tiled) {
            const int passes = 20;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (tiled) hipLaunchKernelGGL(k_stream<true>, dim3(512), dim3(256), 0, 0, d, nF, nbt, ld, strideF, work, passes);
                else hipLaunchKernelGGL(k_stream<false>, dim3(512), dim3(256), 0, 0, d, nF, nbt, ld, strideF, work, passes);
                hipEventRecord(b);
                hipEventSynchronize(b);
            }
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double bytes = 2.0 * passes * nF * nbt * nbt * 64 * 64 * 8;
            printf("work %3d  %-26s %7.1f us per pass   %5.2f TB/s (read + write)\n", work, tiled ? "tiles contiguous (32 KB)" : "row-major, rows 5 KB apart", ms * 1e3 / passes,
                   bytes / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
