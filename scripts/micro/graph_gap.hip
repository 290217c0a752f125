// Per-launch cost of a chain of dependent small kernels: plain stream launches vs a captured hipGraph.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_small(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1e-9;
}
int main() {
    const int n = 256 * 170, L = 40, reps = 50;
    double* d;
    hipMalloc(&d, n * sizeof(double));
    hipMemset(d, 0, n * sizeof(double));
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto run = [&](const char* name, auto&& body) {
        body();
        hipStreamSynchronize(s);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < reps; ++r) body();
        hipStreamSynchronize(s);
        auto t1 = std::chrono::high_resolution_clock::now();
        printf("%-32s %.2f us per kernel\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * L));
    };
    run("stream launches", [&] { for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_small, dim3(170), dim3(256), 0, s, d, n); });
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_small, dim3(170), dim3(256), 0, s, d, n);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    run("graph (40 dependent nodes)", [&] { hipGraphLaunch(ge, s); });
    run("stream launches again", [&] { for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_small, dim3(170), dim3(256), 0, s, d, n); });
    return 0;
}
