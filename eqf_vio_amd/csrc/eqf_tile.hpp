// Tile-local kernels for Sigma 2-D block-partitioned over a process grid (BASELINE configs[4]): every rank
// owns (3 nI x 3 nJ) tiles of the landmark-landmark part of Sigma in ITS OWN device memory (caller-owned buffers, e.g. torch
// tensors); the exchange schedule lives above the C ABI (eqf_vio_amd/tiled.py, torch.distributed over RCCL).  fp64.
//
//   k_tile_propagate   one structured Riccati step of one tile (VIOFilter.cpp:188-189 with F = [[F_bb, 0], [L, D]]):
//                        Sigma'_IJ = (D_I Sigma_IJ + L_I Sigma_bJ) D_J^T + (L_I Sigma_bb + D_I Sigma_Ib) L_J^T + T (B_I R B_J^T [+ p I])
//                      one thread per 3x3 block, a workgroup = 16 x 16 landmark pairs; the rows G_i = L_i Sigma_bb + D_i Sigma_ib of
//                      its 16 row landmarks are formed once per workgroup in LDS.  Needs nothing from any other rank: the
//                      11 x n base panel and the per-landmark blocks are replicated.
//   k_tile_downdate    C -= A^T B for A (k x m), B (k x n) row-major: the tile's share of Sigma - K C Sigma = Sigma - Y^T Y
//                      (VIOFilter.cpp:297) from the solved block rows Y_k that the update's all-gather delivers; 64 x 64 outputs per
//                      workgroup on v_mfma_f64_16x16x4_f64, operands staged through LDS in chunks of 32 rows.
#pragma once
#include "eqf_update.hpp"

namespace eqf {

struct TilePropArgs {
    double* out;
    const double* in;
    int ld, nI, nJ;
    const double *DI, *LI, *DJ, *LJ;     // [n][9] row-major 3x3 ; [3 n][11]
    const double *Sbb, *SbI, *SbJ;       // [11][11] ; [11][ldbI] ; [11][ldbJ]
    int ldbI, ldbJ;
    const double *BnI, *BnJ;             // [3 n][6] rows of the input matrix B
    double R[6];
    double T, diagNoise;                 // diagNoise = T * pointProcessVariance, added on the diagonal of blocks with i == j
    int isDiag;                          // tile (I, I): landmark i of the rows IS landmark i of the columns
};

__global__ __launch_bounds__(256) void k_tile_propagate(TilePropArgs a) {
    __shared__ double sSbb[11][12];
    __shared__ double sG[16][3][12];  // G_i rows of the workgroup's 16 row landmarks
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    if (tid < 121) sSbb[tid / 11][tid % 11] = a.Sbb[tid];
    __syncthreads();
    // ---- G_i = L_i Sigma_bb + D_i Sigma_ib  (3 x 11 per landmark): 16 x 33 entries over 256 threads
    for (int e = tid; e < 16 * 33; e += 256) {
        const int li = e / 33, rr = (e % 33) / 11, cc = e % 11, i = i0 + li;
        double g = 0.0;
        if (i < a.nI) {
            const double* L = a.LI + (long long)(3 * i + rr) * 11;
#pragma unroll
            for (int k = 0; k < 11; ++k) g = fma(L[k], sSbb[k][cc], g);
            const double* D = a.DI + (long long)i * 9 + 3 * rr;
#pragma unroll
            for (int k = 0; k < 3; ++k) g = fma(D[k], a.SbI[(long long)cc * a.ldbI + 3 * i + k], g);  // Sigma_ib[k][cc] = Sigma_bI[cc][3 i + k]
        }
        sG[li][rr][cc] = g;
    }
    __syncthreads();
    const int i = i0 + ti, j = j0 + tj;
    if (i >= a.nI || j >= a.nJ) return;
    double Di[9], Dj[9], S[9], Li[33], Lj[33], Sbj[33];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        Di[k] = a.DI[(long long)i * 9 + k];
        Dj[k] = a.DJ[(long long)j * 9 + k];
        S[k] = a.in[(long long)(3 * i + k / 3) * a.ld + 3 * j + k % 3];
    }
#pragma unroll
    for (int k = 0; k < 33; ++k) {
        Li[k] = a.LI[(long long)(3 * i) * 11 + k];
        Lj[k] = a.LJ[(long long)(3 * j) * 11 + k];
        Sbj[k] = a.SbJ[(long long)(k / 3) * a.ldbJ + 3 * j + k % 3];  // [b][c]
    }
    // M = D_i S_ij + L_i Sigma_bj
    double M[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double m = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) m = fma(Di[3 * r + k], S[3 * k + c], m);
#pragma unroll
            for (int k = 0; k < 11; ++k) m = fma(Li[11 * r + k], Sbj[3 * k + c], m);
            M[3 * r + c] = m;
        }
    double bri[18], bnj[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        bri[k] = a.BnI[(long long)(3 * i) * 6 + k] * a.R[k % 6];
        bnj[k] = a.BnJ[(long long)(3 * j) * 6 + k];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double o = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) o = fma(M[3 * r + k], Dj[3 * c + k], o);          // M D_j^T
#pragma unroll
            for (int k = 0; k < 11; ++k) o = fma(sG[ti][r][k], Lj[11 * c + k], o);        // G_i L_j^T
            double q = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) q = fma(bri[6 * r + k], bnj[6 * c + k], q);       // B_i R B_j^T
            o = fma(a.T, q, o);
            if (a.isDiag && i == j && r == c) o += a.diagNoise;
            a.out[(long long)(3 * i + r) * a.ld + 3 * j + c] = o;
        }
}

// C (m x n, ldc) -= A^T B with A (k x m, lda), B (k x n, ldb), all row-major; k a multiple of 4 is NOT required (tail rows are
// zero-filled).  grid = (ceil(n / 64), ceil(m / 64)), block = 256: each wave a 32 x 32 quadrant as 2 x 2 MFMA tiles.
__global__ __launch_bounds__(256) void k_tile_downdate(double* C, int ldc, int m, int n, const double* A, int lda, const double* B, int ldb, int k) {
    constexpr int TS = 64, KC = 32;
    __shared__ double sA[KC][TS + 1];
    __shared__ double sB[KC][TS + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int I0 = blockIdx.y * TS, J0 = blockIdx.x * TS;
    const int qi = wv >> 1, qj = wv & 1, lr = lane & 15, lk = lane >> 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[u][v][q] = 0.0;
    const int sr = tid >> 3, sc = (tid & 7) * 8;  // staging: row tid / 8 of the chunk, 8 consecutive columns
    double pa[8], pb[8];
    auto fetch = [&](int k0) {
        const int row = k0 + sr;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ca = I0 + sc + q, cb = J0 + sc + q;
            pa[q] = (row < k && ca < m) ? A[(long long)row * lda + ca] : 0.0;
            pb[q] = (row < k && cb < n) ? B[(long long)row * ldb + cb] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < k; k0 += KC) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sA[sr][sc + q] = pa[q];
            sB[sr][sc + q] = pb[q];
        }
        __syncthreads();
        if (k0 + KC < k) fetch(k0 + KC);
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            double av[2], bv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                av[u] = sA[4 * s + lk][32 * qi + 16 * u + lr];
                bv[u] = sB[4 * s + lk][32 * qj + 16 * u + lr];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[v], acc[u][v], 0, 0, 0);
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int R = I0 + 32 * qi + 16 * u + (lane >> 4) + 4 * q, Cc = J0 + 32 * qj + 16 * v + lr;
                if (R < m && Cc < n) C[(long long)R * ldc + Cc] -= acc[u][v][q];
            }
}

}  // namespace eqf
