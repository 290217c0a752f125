// Dense Riccati backend (BASELINE cfg 3): Sigma' = (F Sigma) F^T + T (P + B R B^T) with F formed densely, the two
// n^3 products on the matrix cores (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, LDS-tiled 64x64x32).
//
// This is the operation sequence the reference executes (eqf_vio/src/VIOFilter.cpp:162-189): 4 n^3 flops although
// F is 98.5 % zeros at N = 200.  It is NOT the product path (k_propagate is ~50x cheaper); it exists as a cross-check
// of the block-structured kernel and to measure the MFMA roofline on the filter's own matrices.
#pragma once
#include "eqf_propagate.hpp"
#include "eqf_update.hpp"

namespace eqf {

// One workgroup per landmark (3 rows of F) plus one for the 12 base rows: zero-fill the rows, then write the blocks.
// Also writes Bn (n x 6): the columns of sqrt(T var) * B_b, so that T * B R B^T = Bn Bn^T in the epilogue.
template <typename T>
__global__ __launch_bounds__(256) void k_dense_build(PropArgs a, T* F, T* Bn, long long fStride, long long bStride) {
    const int b = blockIdx.y;
    const Glob& G = a.gin[b];
    const ImuRec& r = a.recs ? a.recs[b] : a.inl;
    const int N = G.N, nv = kLm0 + 3 * N, ld = a.ld, cap = a.cap;
    const int blk = blockIdx.x;  // 0: base rows, 1 + i: landmark i
    if (blk > N) return;
    const double dt0 = r.stamp - G.curTime;
    if (!((G.curTime >= 0) && (dt0 > 0))) return;
    T* Fb = F + (long long)b * fStride;
    T* Bb = Bn + (long long)b * bStride;
    const int tid = threadIdx.x;
    const int r0 = blk == 0 ? 0 : kLm0 + 3 * (blk - 1), nr = blk == 0 ? kLm0 : 3;
    for (int e = tid; e < nr * nv; e += 256) {
        const int rr = r0 + e / nv, cc = e % nv;
        Fb[(long long)rr * ld + cc] = (rr == cc && rr != kBase) ? (T)1 : (T)0;
    }
    for (int e = tid; e < nr * 6; e += 256) Bb[(long long)(r0 + e / 6) * 6 + e % 6] = (T)0;
    __syncthreads();
    if (tid != 0) return;
    int bad = 0;
    StepCommon c;
    stepCommon(G, r, a, c, kPartBase | kPartRicc, &bad);
    const double sw = sqrt(c.T * a.prm.velOmegaVariance), sa = sqrt(c.T * a.prm.velAccelVariance);
    if (blk == 0) {
        for (int g = 0; g < 2; ++g)
            for (int k = 0; k < 3; ++k) {
                Fb[(long long)(6 + g) * ld + k] = (T)(-c.T * c.Bg[3 * g + k]);
                Bb[(6 + g) * 6 + k] = (T)(sw * c.Bg[3 * g + k]);
            }
        for (int v = 0; v < 3; ++v) {
            for (int k = 0; k < 3; ++k) {
                Fb[(long long)(8 + v) * ld + k] = (T)(-c.T * c.Bvw.a[3 * v + k]);
                Fb[(long long)(8 + v) * ld + 3 + k] = (T)(-c.T * c.RA.a[3 * v + k]);
                Bb[(8 + v) * 6 + k] = (T)(sw * c.Bvw.a[3 * v + k]);
                Bb[(8 + v) * 6 + 3 + k] = (T)(sa * c.RA.a[3 * v + k]);
            }
            for (int k = 0; k < 2; ++k) Fb[(long long)(8 + v) * ld + 6 + k] = (T)(c.T * c.Avg[2 * v + k]);
        }
    } else {
        const int i = blk - 1;
        const double* p0 = a.p0 + (long long)b * 3 * cap;
        const double* Q = a.Qin + (long long)b * 5 * cap;
        const LmBlocks B3 = buildBlocks(c, quat{Q[i], Q[cap + i], Q[2 * cap + i], Q[3 * cap + i]}, Q[4 * cap + i],
            mk3(p0[i], p0[cap + i], p0[2 * cap + i]));
        for (int rr = 0; rr < 3; ++rr)
            for (int k = 0; k < 3; ++k) {
                Fb[(long long)(r0 + rr) * ld + k] = (T)B3.Lw.a[3 * rr + k];
                Fb[(long long)(r0 + rr) * ld + 8 + k] = (T)B3.Lv.a[3 * rr + k];
                Fb[(long long)(r0 + rr) * ld + r0 + k] = (T)B3.D.a[3 * rr + k];
                Bb[(r0 + rr) * 6 + k] = (T)(sw * (-B3.Lw.a[3 * rr + k] / c.T));  // B_i = -Lw / T
            }
    }
    if (bad && a.errflag) atomicOr(a.errflag, 64);
}

// C = A op(B) (+ epilogue), square n x n operands with leading dimension ld, TM x 64 tile per workgroup (each wave a
// (TM/2) x 32 quadrant as (TM/32) x 2 MFMA tiles), K chunks of 32 staged through LDS, next chunk prefetched into registers
// during the MFMAs.  Interior tiles (the bulk) take unguarded contiguous reads.
//   TRANSB = false: C = A B          (G = F Sigma)
//   TRANSB = true : C = A B^T + T*P + Bn Bn^T   (Sigma' = G F^T + process noise)
template <typename T, bool TRANSB, int TM>
__global__ __launch_bounds__(256) void k_dense_gemm(const Glob* gin, const ImuRec* recs, ImuRec inl, const T* A, const T* Bm, T* Cm,
    const T* Bn, long long mStride, long long bStride, int ld, Params prm) {
    constexpr int WU = TM / 32;  // MFMA tiles per wave along the rows
    constexpr int PA = TM / 64;  // staging passes for the A tile
    const int b = blockIdx.z;
    const Glob& G = gin[b];
    const ImuRec& r = recs ? recs[b] : inl;
    const double dt0 = r.stamp - G.curTime;
    if (!((G.curTime >= 0) && (dt0 > 0))) return;
    const int nv = kLm0 + 3 * G.N;
    const int I0 = blockIdx.y * TM, J0 = blockIdx.x * 64;
    if (I0 >= nv || J0 >= nv) return;
    const T* Ab = A + (long long)b * mStride;
    const T* Bb = Bm + (long long)b * mStride;
    T* Cb = Cm + (long long)b * mStride;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int KC = 32;
    __shared__ T sA[TM][KC + 1];
    __shared__ T sB[TRANSB ? 64 : KC][TRANSB ? KC + 1 : 64 + 1];
    const int qi = wv >> 1, qj = wv & 1, lr = lane & 15, lk = lane >> 4;
    typedef MfmaT<T> MF;
    typename MF::acc_t acc[WU][2];
#pragma unroll
    for (int u = 0; u < WU; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[u][v][q] = 0;
    // staging: A (and B^T): row = tid / 4 (+ 64 per pass), 8 consecutive k from 8 * (tid % 4);  B (NN): k = tid / 8,
    // 8 consecutive columns
    const int ar = tid >> 2, ak = (tid & 3) * 8;
    const int bk = tid >> 3, bc = (tid & 7) * 8;
    T pa[PA][8], pb[8];
    const bool interiorIJ = I0 + TM <= nv && J0 + 64 <= nv;
    auto fetch = [&](int k0) {
        if (interiorIJ && k0 + KC <= nv) {
            const T* bp = TRANSB ? Bb + (long long)(J0 + ar) * ld + k0 + ak : Bb + (long long)(k0 + bk) * ld + J0 + bc;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const T* ap = Ab + (long long)(I0 + 64 * p + ar) * ld + k0 + ak;
#pragma unroll
                for (int q = 0; q < 8; ++q) pa[p][q] = ap[q];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) pb[q] = bp[q];
            return;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int K = k0 + ak + q;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int R = I0 + 64 * p + ar;
                pa[p][q] = (R < nv && K < nv) ? Ab[(long long)R * ld + K] : (T)0;
            }
            if (TRANSB) {
                const int Rb = J0 + ar;
                pb[q] = (Rb < nv && K < nv) ? Bb[(long long)Rb * ld + K] : (T)0;
            } else {
                const int Kb = k0 + bk, Cc = J0 + bc + q;
                pb[q] = (Kb < nv && Cc < nv) ? Bb[(long long)Kb * ld + Cc] : (T)0;
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < nv; k0 += KC) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int p = 0; p < PA; ++p) sA[64 * p + ar][ak + q] = pa[p][q];
            if (TRANSB) sB[ar][ak + q] = pb[q];
            else sB[bk][bc + q] = pb[q];
        }
        __syncthreads();
        if (k0 + KC < nv) fetch(k0 + KC);
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            T av[WU], bv[2];
#pragma unroll
            for (int u = 0; u < WU; ++u) av[u] = sA[(TM / 2) * qi + 16 * u + lr][4 * s + lk];
#pragma unroll
            for (int v = 0; v < 2; ++v) bv[v] = TRANSB ? sB[32 * qj + 16 * v + lr][4 * s + lk] : sB[4 * s + lk][32 * qj + 16 * v + lr];
#pragma unroll
            for (int u = 0; u < WU; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) acc[u][v] = MF::mfma(av[u], bv[v], acc[u][v]);
        }
    }
    const T Tt = (T)(G.accTime + dt0);
    const T* Bnb = Bn ? Bn + (long long)b * bStride : nullptr;
#pragma unroll
    for (int u = 0; u < WU; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int R = I0 + (TM / 2) * qi + 16 * u + MF::row(lane, q);
                const int Cc = J0 + 32 * qj + 16 * v + lr;
                if (R < nv && Cc < nv) {
                    T val = acc[u][v][q];
                    if (TRANSB) {
                        if (R == Cc && R != kBase)
                            val += Tt * (T)(R < 3 ? prm.biasOmegaProcessVariance
                                                  : (R < 6 ? prm.biasAccelProcessVariance
                                                           : (R < 8 ? prm.gravityProcessVariance
                                                                    : (R < kBase ? prm.velocityProcessVariance : prm.pointProcessVariance))));
#pragma unroll
                        for (int k = 0; k < 6; ++k) val += Bnb[(long long)R * 6 + k] * Bnb[(long long)Cc * 6 + k];
                    }
                    Cb[(long long)R * ld + Cc] = val;
                }
            }
}

}  // namespace eqf
