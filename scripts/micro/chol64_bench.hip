// Stand-alone driver of k_chol_step64 on one chain: checks L^-1 W against a host Cholesky and prints the cycle stamps of
// the diagonal (look-ahead) workgroup per block column.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEQF_STEP64_STAMPS
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define EQF_STEP64_STAMPS 1
#include "../../eqf_vio_amd/csrc/eqf_chol64.hpp"
using namespace eqf;
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 200;
    const int n = roundUp(eDim(N), 64), nb = n / 64, ldW = 64;
    std::vector<double> A((size_t)n * n), W((size_t)n * ldW), L;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) A[(size_t)r * n + c] = (r == c ? 3.0 + 0.01 * r : 0.0) + 1.0 / (1.0 + std::abs(r - c)) * std::cos(0.1 * (r + c));
    // make SPD: A = M M^T / n + diag
    {
        std::vector<double> M = A;
        for (int r = 0; r < n; ++r)
            for (int c = 0; c <= r; ++c) {
                double s = 0;
                for (int k = 0; k < n; ++k) s += M[(size_t)r * n + k] * M[(size_t)c * n + k];
                A[(size_t)r * n + c] = A[(size_t)c * n + r] = s / n + (r == c ? 1.0 : 0.0);
            }
    }
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < ldW; ++c) W[(size_t)r * ldW + c] = std::sin(0.3 * r + c) + (c == r % 64 ? 1 : 0);
    // host reference
    L = A;
    for (int j = 0; j < n; ++j) {
        for (int k = 0; k < j; ++k)
            for (int i = j; i < n; ++i) L[(size_t)i * n + j] -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
        const double d = std::sqrt(L[(size_t)j * n + j]);
        for (int i = j; i < n; ++i) L[(size_t)i * n + j] /= d;
    }
    std::vector<double> Y = W;
    for (int c = 0; c < ldW; ++c)
        for (int i = 0; i < n; ++i) {
            double s = Y[(size_t)i * ldW + c];
            for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * Y[(size_t)k * ldW + c];
            Y[(size_t)i * ldW + c] = s / L[(size_t)i * n + i];
        }
    double *dA, *dD, *dW, *dWO;
    Glob hg{};
    hg.updateOk = 1;
    hg.N = N;
    Glob* dg;
    int* derr;
    hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dD, sizeof(double) * (nb + 1) * kDRec); hipMalloc(&dW, sizeof(double) * n * ldW);
    hipMalloc(&dWO, sizeof(double) * n * ldW); hipMalloc(&dg, sizeof(Glob)); hipMalloc(&derr, 4);
    hipMemset(derr, 0, 4);
    hipMemcpy(dg, &hg, sizeof(Glob), hipMemcpyHostToDevice);
    ChainArgs c0{}, c1{};
    c0.g = dg; c0.A = dA; c0.D = dD; c0.W = dW; c0.WO = dWO; c0.ldA = n; c0.ldW = ldW; c0.kind = 1; c0.nbMax = nb; c0.wtMax = 1;
    c1 = c0; c1.nbMax = 0; c1.wtMax = 0;
    UpdArgs ua{};
    double* dred; hipMalloc(&dred, 256 * sizeof(double));
    ua.red = dred; ua.g = dg; ua.cap = N;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chol_step64<double, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds)));
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_factor_first64), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds)));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
        hipMemcpy(dW, W.data(), sizeof(double) * n * ldW, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_factor_first64, dim3(2, 1), dim3(256), sizeof(Step64Lds), 0, c0, c1, derr);
        for (int K = 0; K < nb; ++K) hipLaunchKernelGGL((k_chol_step64<double, 0>), dim3(chainBlocks64(nb, 1, K, 0)), dim3(256), sizeof(Step64Lds), 0, c0, c1, ua, K, 0, 0, 0, derr);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("rep %d: %d steps %.1f us total, %.2f us/step\n", rep, nb, ms * 1e3, ms * 1e3 / nb);
    }
    std::vector<double> YO((size_t)n * ldW);
    hipMemcpy(YO.data(), dWO, sizeof(double) * n * ldW, hipMemcpyDeviceToHost);
    double err = 0, ref = 0;
    for (size_t i = 0; i < YO.size(); ++i) { err = std::max(err, std::abs(YO[i] - Y[i])); ref = std::max(ref, std::abs(Y[i])); }
    int herr; hipMemcpy(&herr, derr, 4, hipMemcpyDeviceToHost);
    printf("max |Y - Yref| = %.3e (max |Yref| %.3e) errflag %d\n", err, ref, herr);
    long long st[64][16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    printf("diag workgroup, cycles: load | solve | update | factor | store\n");
    for (int K = 0; K + 1 < nb; ++K)
        printf("K=%2d  %6lld %6lld %6lld %6lld %6lld   total %6lld (%.2f us @2.4GHz)\n", K, st[K][1] - st[K][0], st[K][2] - st[K][1], st[K][3] - st[K][2],
            st[K][4] - st[K][3], st[K][5] - st[K][4], st[K][5] - st[K][0], (st[K][5] - st[K][0]) / 2400.0),
        printf("      factor stages (phase P | phase U): %lld %lld | %lld %lld | %lld %lld | %lld %lld  end %lld\n", st[K][8] - st[K][3], st[K][9] - st[K][8],
            st[K][10] - st[K][9], st[K][11] - st[K][10], st[K][12] - st[K][11], st[K][13] - st[K][12], st[K][14] - st[K][13], st[K][15] - st[K][14], st[K][4] - st[K][15]);
    return 0;
}
