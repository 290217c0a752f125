// A 16 x 16 diagonal block factored inside ONE 16-lane DPP row: lane r holds row r of the block AND row r of the identity that rides along
// (-> L and W = L^-T), the rank-1 multipliers reach the lanes by `row_newbcast` inside the consuming instruction (v_fmac_f64_dpp: gfx90a+
// allows DPP on 64-bit VOP2 with exactly this control) instead of a pair of v_readlane_b32 + an SGPR operand.  The four rows of the wave
// compute four copies.  Against potrf16v2's scheme (64 lanes = 64 rows of the panel, multipliers through SGPRs): instruction count per
// pivot, cycles per 16 pivots, result against a host Cholesky.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/micro/potrf16_dpp scripts/micro/potrf16_dpp.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define BCAST(N) "row_newbcast:" #N " row_mask:0xf bank_mask:0xf"
template <int N>
__device__ __forceinline__ double bcast(double v) {
    double d;
    if constexpr (N == 0) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(0) : "=v"(d) : "v"(v));
    if constexpr (N == 1) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(1) : "=v"(d) : "v"(v));
    if constexpr (N == 2) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(2) : "=v"(d) : "v"(v));
    if constexpr (N == 3) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(3) : "=v"(d) : "v"(v));
    if constexpr (N == 4) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(4) : "=v"(d) : "v"(v));
    if constexpr (N == 5) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(5) : "=v"(d) : "v"(v));
    if constexpr (N == 6) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(6) : "=v"(d) : "v"(v));
    if constexpr (N == 7) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(7) : "=v"(d) : "v"(v));
    if constexpr (N == 8) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(8) : "=v"(d) : "v"(v));
    if constexpr (N == 9) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(9) : "=v"(d) : "v"(v));
    if constexpr (N == 10) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(10) : "=v"(d) : "v"(v));
    if constexpr (N == 11) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(11) : "=v"(d) : "v"(v));
    if constexpr (N == 12) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(12) : "=v"(d) : "v"(v));
    if constexpr (N == 13) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(13) : "=v"(d) : "v"(v));
    if constexpr (N == 14) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(14) : "=v"(d) : "v"(v));
    if constexpr (N == 15) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 " BCAST(15) : "=v"(d) : "v"(v));
    return d;
}
// acc -= m * (lane N of the row's l)
template <int N>
__device__ __forceinline__ void fmsBcast(double& acc, double l, double m) {
    if constexpr (N == 1) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(1) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 2) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(2) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 3) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(3) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 4) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(4) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 5) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(5) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 6) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(6) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 7) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(7) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 8) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(8) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 9) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(9) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 10) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(10) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 11) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(11) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 12) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(12) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 13) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(13) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 14) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(14) : "+v"(acc) : "v"(l), "v"(m));
    if constexpr (N == 15) asm volatile("v_fmac_f64_dpp %0, %1, -%2 " BCAST(15) : "+v"(acc) : "v"(l), "v"(m));
}
template <int C, int C2>
__device__ __forceinline__ void updates(double* row, double* inv, double lj, double ij) {
    if constexpr (C2 < 16) {
        fmsBcast<C2>(row[C2], lj, lj);
        fmsBcast<C2>(inv[C2], lj, ij);
        updates<C, C2 + 1>(row, inv, lj, ij);
    }
}
template <int C>
__device__ __forceinline__ void pivots(double* row, double* inv) {
    if constexpr (C < 16) {
        const double d = bcast<C>(row[C]);
        const double y = __builtin_amdgcn_rsq(d);
        const double e = fma(-(d * y), y, 1.0);
        const double p = fma(0.375, e, 0.5);
        const double ay = row[C] * y, by = inv[C] * y;
        const double lj = fma(ay * e, p, ay), ij = fma(by * e, p, by);
        // (a DPP read of a VGPR the previous VALU instruction wrote needs two wait states; inline asm is opaque to the hazard recognizer)
        double ljh = lj;
        asm volatile("s_nop 1" : "+v"(ljh));
        updates<C, C + 1>(row, inv, ljh, ij);
        row[C] = ljh;
        inv[C] = ij;
        pivots<C + 1>(row, inv);
    }
}
__global__ __launch_bounds__(64) void k_dpp(const double* A, double* outL, double* outW, long long* cyc, int reps) {
    __shared__ double sA[16][17];
    const int lane = threadIdx.x, r = lane & 15;
    for (int e = lane; e < 256; e += 64) sA[e >> 4][e & 15] = A[e];
    __syncthreads();
    double row[16], inv[16];
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            row[c] = sA[r][c];
            inv[c] = (r == c) ? 1.0 : 0.0;
        }
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        pivots<0>(row, inv);
        asm volatile("" ::"v"(row[15]), "v"(inv[15]));
        const long long t1 = __builtin_readcyclecounter();
        if (lane == 0) cyc[rep] = t1 - t0;
    }
    if (lane < 16) {
        for (int c = 0; c < 16; ++c) {
            outL[r * 16 + c] = row[c];
            outW[r * 16 + c] = inv[c];
        }
    }
}
int main() {
    std::vector<double> M(256), A(256, 0.0), L(256, 0.0), W(256);
    unsigned s = 12345;
    for (auto& v : M) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / (1 << 24) - 0.5; }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            for (int k = 0; k < 16; ++k) A[i * 16 + j] += M[i * 16 + k] * M[j * 16 + k];
            if (i == j) A[i * 16 + j] += 4.0;
        }
    for (int j = 0; j < 16; ++j) {  // host Cholesky
        double d = A[j * 16 + j];
        for (int k = 0; k < j; ++k) d -= L[j * 16 + k] * L[j * 16 + k];
        L[j * 16 + j] = std::sqrt(d);
        for (int i = j + 1; i < 16; ++i) {
            double v = A[i * 16 + j];
            for (int k = 0; k < j; ++k) v -= L[i * 16 + k] * L[j * 16 + k];
            L[i * 16 + j] = v / L[j * 16 + j];
        }
    }
    double *dA, *dL, *dW;
    long long* dC;
    const int reps = 8;
    hipMalloc(&dA, 2048); hipMalloc(&dL, 2048); hipMalloc(&dW, 2048); hipMalloc(&dC, 8 * reps);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, dA, dL, dW, dC, reps);
    std::vector<double> gL(256), gW(256);
    std::vector<long long> cyc(reps);
    hipMemcpy(gL.data(), dL, 2048, hipMemcpyDeviceToHost);
    hipMemcpy(gW.data(), dW, 2048, hipMemcpyDeviceToHost);
    hipMemcpy(cyc.data(), dC, 8 * reps, hipMemcpyDeviceToHost);
    double eL = 0, eW = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j <= i; ++j) eL = std::fmax(eL, std::fabs(gL[i * 16 + j] - L[i * 16 + j]));
    // W = L^-T: rows of W times L^T = I  <=>  sum_k W[r][k] L[c][k] = delta_rc
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            double v = 0;
            for (int k = 0; k < 16; ++k) v += gW[r * 16 + k] * ((c >= k) ? L[c * 16 + k] : 0.0);
            eW = std::fmax(eW, std::fabs(v - (r == c ? 1.0 : 0.0)));
        }
    std::printf("potrf16 in one DPP row (L and W = L^-T): max |L - Lref| = %.3e, max |W L^T - I| = %.3e\ncycles per 16 pivots:", eL, eW);
    for (int i = 0; i < reps; ++i) std::printf(" %lld", cyc[i]);
    std::printf("\n");
    return 0;
}
