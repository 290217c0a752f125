// factor64 (eqf_chol64.hpp) alone in ONE workgroup: result against a host Cholesky, cycles cold (first call of the launch) and warm, and the
// shader-clock stamps of every wave inside the factorisation (-DEQF_F64_STAMPS).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -o scripts/micro/factor64_bench scripts/micro/factor64_bench.hip
//   usage: factor64_bench [variant] [wt]      variant 0 = factor64 as shipped ; wt = 1: write-through record + stage flags (the resident kernel's use)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#ifndef NO_STAMPS
#define EQF_F64_STAMPS 1
#endif
#include "../../eqf_vio_amd/csrc/eqf_chol64.hpp"
using namespace eqf;

template <int VARIANT, bool WT>
__global__ __launch_bounds__(256) void k_f64(const double* A, double* outL, double* outW, double* Dn, int* flags, long long* stamps, long long* total, int reps, int nst) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemB[];
    const Lds64 s = ldsFull(smemB);
    const int tid = threadIdx.x;
    int bad = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int e = tid; e < kSB * kSB; e += 256) s.L[e >> 6][e & 63] = A[e];
        __syncthreads();
        factorPrologue(s, tid);
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        const int ln = tid & 63;
        factor64<WT>(s, tid, &bad, [&](int, int r, int c, f64x4& acc) { acc = ldTile(&s.L[0][0], kSP, kQB * r, kQB * c, ln); }, WT ? Dn : nullptr,
            stamps + 128 * rep, nst, [] {}, WT ? flags : nullptr, rep + 1);
        __syncthreads();
        const long long t1 = __builtin_readcyclecounter();
        if (tid == 0) total[rep] = t1 - t0;
        __syncthreads();
    }
    for (int e = tid; e < kSB * kSB; e += 256) outL[e] = s.L[e >> 6][e & 63];
    for (int e = tid; e < 4 * kQB * kQB; e += 256) outW[e] = s.Wd[e >> 8][(e >> 4) & 15][e & 15];
    if (bad && tid == 0) total[reps] = -1;
}

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0, wt = argc > 2 ? atoi(argv[2]) : 0, nst = argc > 3 ? atoi(argv[3]) : 4, reps = 6;
    const int n = 64;
    std::vector<double> A(n * n), L;
    {
        std::vector<double> M(n * n);
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) M[r * n + c] = std::cos(0.37 * r + 0.91 * c) + (r == c ? 2.0 : 0.0);
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) {
                double x = 0;
                for (int k = 0; k < n; ++k) x += M[r * n + k] * M[c * n + k];
                A[r * n + c] = x / n + (r == c ? 0.5 : 0.0);
            }
    }
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c)
            if (r >= 16 * nst || c >= 16 * nst) A[r * n + c] = r == c ? 1.0 : 0.0;
    L = A;
    for (int j = 0; j < n; ++j) {
        for (int k = 0; k < j; ++k)
            for (int i = j; i < n; ++i) L[i * n + j] -= L[i * n + k] * L[j * n + k];
        const double d = std::sqrt(L[j * n + j]);
        for (int i = j; i < n; ++i) L[i * n + j] /= d;
    }
    double *dA, *dL, *dW, *dD;
    int* dF;
    long long *dS, *dT;
    hipMalloc(&dA, 8 * n * n); hipMalloc(&dL, 8 * n * n); hipMalloc(&dW, 8 * 1024); hipMalloc(&dD, 8 * kDRec); hipMalloc(&dF, 64);
    hipMalloc(&dS, 8 * 128 * reps); hipMalloc(&dT, 8 * (reps + 1));
    hipMemset(dT, 0, 8 * (reps + 1)); hipMemset(dF, 0, 64);
    hipMemcpy(dA, A.data(), 8 * n * n, hipMemcpyHostToDevice);
    auto launch = [&](auto kern) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(sizeof(Step64Lds)));
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), sizeof(Step64Lds), 0, dA, dL, dW, dD, dF, dS, dT, reps, nst);
    };
    if (variant == 0 && !wt) launch(k_f64<0, false>);
    else if (variant == 0) launch(k_f64<0, true>);
    else { printf("unknown variant\n"); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<double> gL(n * n), gW(1024);
    std::vector<long long> st(128 * reps), tot(reps + 1);
    hipMemcpy(gL.data(), dL, 8 * n * n, hipMemcpyDeviceToHost); hipMemcpy(gW.data(), dW, 8 * 1024, hipMemcpyDeviceToHost);
    hipMemcpy(st.data(), dS, 8 * 128 * reps, hipMemcpyDeviceToHost); hipMemcpy(tot.data(), dT, 8 * (reps + 1), hipMemcpyDeviceToHost);
    double errL = 0, errW = 0;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c <= r; ++c)
            if ((r >> 4) != (c >> 4) || c <= r) errL = std::max(errL, std::abs(gL[r * n + c] - L[r * n + c]));
    for (int j = 0; j < 4; ++j)  // W_jj L_jj = I
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                double x = 0;
                for (int k = 0; k < 16; ++k) x += gW[256 * j + 16 * r + k] * (k >= c ? L[(16 * j + k) * n + 16 * j + c] : 0.0);
                errW = std::max(errW, std::abs(x - (r == c ? 1.0 : 0.0)));
            }
    {
        unsigned long long h = 1469598103934665603ULL;
        for (int r = 0; r < n; ++r)
            for (int c = 0; c <= r; ++c) { unsigned long long u; memcpy(&u, &gL[r * n + c], 8); h = (h ^ u) * 1099511628211ULL; }
        for (int e = 0; e < 1024; ++e) { unsigned long long u; memcpy(&u, &gW[e], 8); h = (h ^ u) * 1099511628211ULL; }
        std::vector<double> rec(kDRec);
        hipMemcpy(rec.data(), dD, 8 * kDRec, hipMemcpyDeviceToHost);
        unsigned long long h2 = 1469598103934665603ULL;
        for (int e = 0; e < kDRec; ++e) { unsigned long long u; memcpy(&u, &rec[e], 8); h2 = (h2 ^ u) * 1099511628211ULL; }
        double errU = 0;
        for (int r = 0; r < n; ++r) for (int c = r + 1; c < n; ++c) if ((r >> 4) == (c >> 4)) errU = std::max(errU, std::abs(gL[r * n + c]));
        printf("hash(L lower, W) %016llx  hash(record) %016llx  max |upper triangle of diagonal blocks| %.1e\n", h, wt ? h2 : 0ULL, errU);
    }
    printf("variant %d wt %d: max |L - Lref| = %.3e, max |W L - I| = %.3e, bad %lld\n", variant, wt, errL, errW, tot[reps]);
    printf("cycles per call:");
    for (int r = 0; r < reps; ++r) printf(" %lld", tot[r]);
    printf("   (%.2f us warm @2.4 GHz)\n", tot[reps - 1] / 2400.0);
    for (int rep : {0, reps - 1}) {
        const long long* S = st.data() + 128 * rep;
        const long long t0 = S[0];
        printf("rep %d stamps (cycles since wave 0 / stage 0 start): start | P done | A passed | U done | B passed | rows loaded | pivots done\n", rep);
        for (int w = 0; w < 4; ++w)
            for (int j = 0; j < 4; ++j) {
                printf("  wave %d stage %d:", w, j);
                for (int k = 0; k < 7; ++k) printf(" %7lld", S[(w * 4 + j) * 8 + k] ? S[(w * 4 + j) * 8 + k] - t0 : -1);
                printf("\n");
            }
    }
    return 0;
}
