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

// MODE 4: LDL^T-style chain (reciprocal of the pivot on the chain, the square root off it)
__device__ __forceinline__ void variantLdl(double (*T)[kSP], int base, int lane, int* bad) {
    double u[kQB], out[kQB];
#pragma unroll
    for (int c = 0; c < kQB; ++c) u[c] = T[lane][base + c];
    double tPrev = 0.0, uPrev = 0.0;
#pragma unroll
    for (int c = 0; c < kQB; ++c) {
        const double d = readlane64(u[c], base + c);
        if (!(d > 0.0)) *bad = 1;
        double r = __builtin_amdgcn_rcp(d);
        r = fma(r, fma(-d, r, 1.0), r);
        r = fma(r, fma(-d, r, 1.0), r);
        if (c >= 1) {
#pragma unroll
            for (int c2 = c + 1; c2 < kQB; ++c2) u[c2] = fma(-tPrev, readlane64(uPrev, base + c2), u[c2]);
        }
        const double uc = u[c];
        const double t = uc * r;
        if (c + 1 < kQB) u[c + 1] = fma(-t, readlane64(uc, base + c + 1), u[c + 1]);
        out[c] = uc * rsqrtPivot(d);
        tPrev = t;
        uPrev = uc;
#pragma unroll
        for (int c2 = c + 1; c2 < kQB; ++c2) __asm__ volatile("" : "+v"(u[c2]));
    }
    if (lane >= base) {
#pragma unroll
        for (int c = 0; c < kQB; ++c) T[lane][base + c] = (lane - base >= c) ? out[c] : 0.0;
    }
}
// MODE 5: the bulk FMAs of column c-1 hand-interleaved between the dependent operations of pivot c's chain
#define SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void variantIl(double (*T)[kSP], int base, int lane, int* bad) {
    double row[kQB];
#pragma unroll
    for (int c = 0; c < kQB; ++c) row[c] = T[lane][base + c];
    double ljPrev = 0.0;
#pragma unroll
    for (int c = 0; c < kQB; ++c) {
        // bulk items of this iteration: c2 = c+1 .. 15 (column c-1), spread over 7 slots
        auto bulk = [&](int slot) {
            if (c >= 1) {
#pragma unroll
                for (int c2 = c + 1; c2 < kQB; ++c2)
                    if ((c2 - c - 1) % 7 == slot) row[c2] = fma(-ljPrev, readlane64(ljPrev, base + c2), row[c2]);
            }
        };
        const double d = readlane64(row[c], base + c);
        if (!(d > 0.0)) *bad = 1;
        double y = __builtin_amdgcn_rsq(d);
        const double h = 0.5 * d;
        SB(); bulk(0); SB();
        double t = h * y;
        SB(); bulk(1); SB();
        t = fma(-y, t, 1.5);
        SB(); bulk(2); SB();
        y = y * t;
        SB(); bulk(3); SB();
        t = h * y;
        SB(); bulk(4); SB();
        t = fma(-y, t, 1.5);
        SB(); bulk(5); SB();
        y = y * t;
        SB(); bulk(6); SB();
        const double lj = row[c] * y;
        if (c + 1 < kQB) row[c + 1] = fma(-lj, readlane64(lj, base + c + 1), row[c + 1]);
        row[c] = lj;
        ljPrev = lj;
        SB();
    }
    if (lane >= base) {
#pragma unroll
        for (int c = 0; c < kQB; ++c) T[lane][base + c] = (lane - base >= c) ? row[c] : 0.0;
    }
}
// MODE 6: all broadcasts of a column first (distinct SGPR pairs), then the FMAs
__device__ __forceinline__ void variantGrp(double (*T)[kSP], int base, int lane, int* bad) {
    double row[kQB];
#pragma unroll
    for (int c = 0; c < kQB; ++c) row[c] = T[lane][base + c];
    double ljPrev = 0.0;
#pragma unroll
    for (int c = 0; c < kQB; ++c) {
        const double d = readlane64(row[c], base + c);
        if (!(d > 0.0)) *bad = 1;
        const double rd = rsqrtPivot(d);
        if (c >= 1) {
            double bc[kQB];
#pragma unroll
            for (int c2 = c + 1; c2 < kQB; ++c2) bc[c2] = readlane64(ljPrev, base + c2);
#pragma unroll
            for (int c2 = c + 1; c2 < kQB; ++c2) __asm__ volatile("" : "+s"(bc[c2]));
#pragma unroll
            for (int c2 = c + 1; c2 < kQB; ++c2) row[c2] = fma(-ljPrev, bc[c2], row[c2]);
        }
        const double lj = row[c] * rd;
        if (c + 1 < kQB) row[c + 1] = fma(-lj, readlane64(lj, base + c + 1), row[c + 1]);
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
// MODE 9: 2 x 2 BLOCK pivots.  The trailing update A' = A - P D^-1 P^T of a two-column panel P with pivot block D = [[a,b],[b,d]]
// needs ONE reciprocal (of det D) on the dependent chain instead of two dependent reciprocal square roots; the Cholesky
// columns L = P chol(D)^-T are formed off the chain.  Same mathematics as two rank-1 steps.
__device__ __forceinline__ double rcpNewton(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
__device__ __forceinline__ void variantBlk2(double (*T)[kSP], int base, int lane, int* bad) {
    double row[kQB];
#pragma unroll
    for (int c = 0; c < kQB; ++c) row[c] = T[lane][base + c];
    double T1 = 0.0, T2 = 0.0, U1 = 0.0, U2 = 0.0;  // previous block: its bulk update (columns >= c + 2) runs one iteration later
#pragma unroll
    for (int c = 0; c < kQB; c += 2) {
        const double u1 = row[c], u2 = row[c + 1];
        const double a = readlane64(u1, base + c), b = readlane64(u1, base + c + 1), d = readlane64(u2, base + c + 1);
        // broadcasts this block's immediate update needs (independent of the reciprocal)
        double b1n[2], b2n[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (c + 2 + q < kQB) {
                b1n[q] = readlane64(u1, base + c + 2 + q);
                b2n[q] = readlane64(u2, base + c + 2 + q);
            }
        const double det = fma(a, d, -b * b);
        if (!(a > 0.0) || !(det > 0.0)) *bad = 1;
        const double rdet = rcpNewton(det);
        // deferred bulk of the previous block, in the shadow of the reciprocal
        if (c >= 2) {
            double bc1[kQB], bc2[kQB];
#pragma unroll
            for (int c2 = c + 2; c2 < kQB; ++c2) {
                bc1[c2] = readlane64(U1, base + c2);
                bc2[c2] = readlane64(U2, base + c2);
            }
#pragma unroll
            for (int c2 = c + 2; c2 < kQB; ++c2) __asm__ volatile("" : "+s"(bc1[c2]), "+s"(bc2[c2]));
#pragma unroll
            for (int c2 = c + 2; c2 < kQB; ++c2) row[c2] = fma(-T2, bc2[c2], fma(-T1, bc1[c2], row[c2]));
        }
        const double n1 = fma(u1, d, -u2 * b), n2 = fma(u2, a, -u1 * b);
        const double t1 = n1 * rdet, t2 = n2 * rdet;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (c + 2 + q < kQB) row[c + 2 + q] = fma(-t2, b2n[q], fma(-t1, b1n[q], row[c + 2 + q]));
        // Cholesky columns of the panel (off the chain)
        const double r11 = rsqrtPivot(a);
        const double l21 = b * r11;
        const double r22 = rsqrtPivot(fma(-l21, l21, d));
        const double l1 = u1 * r11;
        row[c] = l1;
        row[c + 1] = fma(-l1, l21, u2) * r22;
        T1 = t1; T2 = t2; U1 = u1; U2 = u2;
#pragma unroll
        for (int c2 = c + 2; c2 < kQB; ++c2) __asm__ volatile("" : "+v"(row[c2]));
    }
    if (lane >= base) {
#pragma unroll
        for (int c = 0; c < kQB; ++c) T[lane][base + c] = (lane - base >= c) ? row[c] : 0.0;
    }
}
// MODE 7 (round 2, measured SLOWER: 4530 vs 3749 cycles): TWO waves, static column split.  Wave A (columns 0..7) runs the pivot chain of its columns and publishes every finished
// pivot column through LDS (column, then a flag: LDS operations of one wave are performed in order); wave B (columns 8..15)
// applies those eight pivots to its columns as they arrive -- off A's instruction stream -- and then runs the pivot chain of its
// own columns.  The serial chain is unchanged (16 pivots); what leaves it is the issue-bound bulk of rank-1 updates.
__device__ __forceinline__ void twoWaveA(double (*T)[kSP], double (*cb)[kSP], volatile int* flg, int tag, int base, int lane, int* bad) {
    double row[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) row[c] = T[lane][base + c];
    double ljPrev = 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const double d = readlane64(row[c], base + c);
        if (!(d > 0.0)) *bad = 1;
        const double rd = rsqrtPivot(d);
        if (c >= 1) {
            double bc[8];
#pragma unroll
            for (int c2 = c + 1; c2 < 8; ++c2) bc[c2] = readlane64(ljPrev, base + c2);
#pragma unroll
            for (int c2 = c + 1; c2 < 8; ++c2) __asm__ volatile("" : "+s"(bc[c2]));
#pragma unroll
            for (int c2 = c + 1; c2 < 8; ++c2) row[c2] = fma(-ljPrev, bc[c2], row[c2]);
        }
        const double lj = row[c] * rd;
        cb[c][lane] = lj;
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) flg[c] = tag;
        if (c + 1 < 8) row[c + 1] = fma(-lj, readlane64(lj, base + c + 1), row[c + 1]);
        row[c] = lj;
        ljPrev = lj;
#pragma unroll
        for (int c2 = c + 1; c2 < 8; ++c2) __asm__ volatile("" : "+v"(row[c2]));
    }
    if (lane >= base) {
#pragma unroll
        for (int c = 0; c < 8; ++c) T[lane][base + c] = (lane - base >= c) ? row[c] : 0.0;
    }
}
__device__ __forceinline__ void twoWaveB(double (*T)[kSP], double (*cb)[kSP], volatile int* flg, int tag, int base, int lane, int* bad) {
    double row[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) row[c] = T[lane][base + 8 + c];
    // pivots 0..7 of wave A, as they arrive
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        while (flg[c] != tag) __builtin_amdgcn_s_sleep(0);
        __asm__ volatile("" ::: "memory");
        const double lj = cb[c][lane];
        double bc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) bc[k] = cb[c][base + 8 + k];  // uniform address: LDS broadcast
#pragma unroll
        for (int k = 0; k < 8; ++k) row[k] = fma(-lj, bc[k], row[k]);
    }
    // own pivot chain, columns 8..15
    double ljPrev = 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const double d = readlane64(row[c], base + 8 + c);
        if (!(d > 0.0)) *bad = 1;
        const double rd = rsqrtPivot(d);
        if (c >= 1) {
            double bc[8];
#pragma unroll
            for (int c2 = c + 1; c2 < 8; ++c2) bc[c2] = readlane64(ljPrev, base + 8 + c2);
#pragma unroll
            for (int c2 = c + 1; c2 < 8; ++c2) __asm__ volatile("" : "+s"(bc[c2]));
#pragma unroll
            for (int c2 = c + 1; c2 < 8; ++c2) row[c2] = fma(-ljPrev, bc[c2], row[c2]);
        }
        const double lj = row[c] * rd;
        if (c + 1 < 8) row[c + 1] = fma(-lj, readlane64(lj, base + 8 + c + 1), row[c + 1]);
        row[c] = lj;
        ljPrev = lj;
#pragma unroll
        for (int c2 = c + 1; c2 < 8; ++c2) __asm__ volatile("" : "+v"(row[c2]));
    }
    if (lane >= base) {
#pragma unroll
        for (int c = 0; c < 8; ++c) T[lane][base + 8 + c] = (lane - base >= 8 + c) ? row[c] : 0.0;
    }
}
__global__ __launch_bounds__(256) void k_bench2(double* out, int reps) {
    __shared__ double T[kSB][kSP];
    __shared__ double T0[kSB][kSP];
    __shared__ double cb[8][kSP];
    __shared__ volatile int flg[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < kSB * kSB; e += blockDim.x) {
        const int r = e / kSB, c = e % kSB;
        T0[r][c] = (r == c ? 40.0 : 0.0) + 1.0 / (1 + r + c);
    }
    if (tid < 16) flg[tid] = 0;
    __syncthreads();
    int bad = 0;
    if (wv < 2) {
        if (wv == 0) for (int c = 0; c < kSB; ++c) T[lane][c] = T0[lane][c];
    }
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    if (wv == 0) twoWaveA(T, cb, flg, 1, 16, lane, &bad);
    else if (wv == 1) twoWaveB(T, cb, flg, 1, 16, lane, &bad);
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < reps; ++i) {
        if (wv == 0) for (int c = 0; c < kQB; ++c) T[lane][16 + c] = T0[lane][16 + c];
        __syncthreads();
        if (wv == 0) twoWaveA(T, cb, flg, 2 + i, 16, lane, &bad);
        else if (wv == 1) twoWaveB(T, cb, flg, 2 + i, 16, lane, &bad);
        __syncthreads();
    }
    long long t2 = __builtin_readcyclecounter();
    if (tid == 0) {
        out[0] = double(t1 - t0);
        out[1] = double(t2 - t1) / reps;
        out[2] = T[40][20] + bad;
    }
}


// MODE 10..13: COMBINATION -- the dependent chain of the LDL^T form (reciprocal of the pivot, NEWTON Newton steps; the scaling of the
// column and its square root are off the chain), the two columns next to the pivot updated through v_readlane, everything further away
// through uniform-address LDS reads of the column published one iteration earlier (half the instructions of a readlane pair), and the
// normalisation u_c / sqrt(d_c) either inline (NORM = 0: eight more instructions per column) or deferred to the end with the sixteen
// reciprocal square roots computed side by side on lanes 0..15 (NORM = 1).
template <int NEWTON, int NORM>
__device__ __forceinline__ void variantCombo(double (*T)[kSP], double (*colS)[kQB + 2], int base, int lane, int* bad) {
    double u[kQB], out[kQB];
#pragma unroll
    for (int c = 0; c < kQB; ++c) u[c] = T[lane][base + c];
    const bool inDiag = lane >= base && lane < base + kQB;
    double tPrev = 0.0, dmine = 1.0;
#pragma unroll
    for (int c = 0; c < kQB; ++c) {
        double bc[kQB];
        if (c >= 1) {
#pragma unroll
            for (int c2 = c + 2; c2 < kQB; ++c2) bc[c2] = colS[c - 1][c2];  // (uniform address: LDS broadcast; issued first, used last)
        }
        const double d = readlane64(u[c], base + c);
        if (!(d > 0.0)) *bad = 1;
        double r = __builtin_amdgcn_rcp(d);
        r = fma(r, fma(-d, r, 1.0), r);
        if (NEWTON == 2) r = fma(r, fma(-d, r, 1.0), r);
        const double uc = u[c];
        const double b1 = (c + 1 < kQB) ? readlane64(uc, base + c + 1) : 0.0;
        const double b2 = (c + 2 < kQB) ? readlane64(uc, base + c + 2) : 0.0;
        const double t = uc * r;
        if (c + 1 < kQB) u[c + 1] = fma(-t, b1, u[c + 1]);
        if (c + 2 < kQB) u[c + 2] = fma(-t, b2, u[c + 2]);
        if (inDiag) colS[c][lane - base] = uc;
        if (c >= 1) {
#pragma unroll
            for (int c2 = c + 2; c2 < kQB; ++c2) u[c2] = fma(-tPrev, bc[c2], u[c2]);
        }
        if (NORM == 0) out[c] = uc * rsqrtPivot(d);
        else {
            out[c] = uc;
            if (lane == base + c) dmine = uc;
        }
        tPrev = t;
#pragma unroll
        for (int c2 = c + 1; c2 < kQB; ++c2) __asm__ volatile("" : "+v"(u[c2]));
    }
    if (NORM == 1) {
        const double rs = rsqrtPivot(dmine);  // lane base + c: 1 / sqrt(d_c)
#pragma unroll
        for (int c = 0; c < kQB; ++c) out[c] *= readlane64(rs, base + c);
    }
    if (lane >= base) {
#pragma unroll
        for (int c = 0; c < kQB; ++c) T[lane][base + c] = (lane - base >= c) ? out[c] : 0.0;
    }
}
template <int MODE>
__device__ __forceinline__ void dispatch(double (*T)[kSP], double (*colS)[kQB + 2], int base, int lane, int* bad) {
    if (MODE == 4) variantLdl(T, base, lane, bad);
    else if (MODE == 5) variantIl(T, base, lane, bad);
    else if (MODE == 6) variantGrp(T, base, lane, bad);
    else if (MODE == 9) variantBlk2(T, base, lane, bad);
    else if (MODE == 10) variantCombo<2, 0>(T, colS, base, lane, bad);
    else if (MODE == 11) variantCombo<1, 0>(T, colS, base, lane, bad);
    else if (MODE == 12) variantCombo<2, 1>(T, colS, base, lane, bad);
    else if (MODE == 13) variantCombo<1, 1>(T, colS, base, lane, bad);
    else variant<MODE>(T, colS, base, lane, bad);
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
        dispatch<MODE>(T, colS, 16, lane, &bad);
        long long t1 = __builtin_readcyclecounter();
        for (int i = 0; i < reps; ++i) {
            for (int c = 0; c < kQB; ++c) T[lane][16 + c] = T0[lane][16 + c];
            dispatch<MODE>(T, colS, 16, lane, &bad);
        }
        long long t2 = __builtin_readcyclecounter();
        if (lane == 0) {
            out[0] = double(t1 - t0);
            out[1] = double(t2 - t1) / reps;
            out[2] = T[40][20] + bad + 1e3 * T[63][31] + 1e6 * T[20][18];
        }
    }
}
// EXEC restricted to the first LIM lanes: does the hardware skip the 16-lane passes of a wave64 VALU instruction whose EXEC bits
// are all zero?  (The later stages of factor64 only have 48 / 32 live rows.)
template <int LIM>
__global__ __launch_bounds__(256) void k_bench_lim(double* out, int reps) {
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
        long long t1 = __builtin_readcyclecounter();
        if (lane < LIM) {
            for (int i = 0; i < reps; ++i) {
                for (int c = 0; c < kQB; ++c) T[lane][16 + c] = T0[lane][16 + c];
                variantGrp(T, 16, lane, &bad);
            }
        }
        long long t2 = __builtin_readcyclecounter();
        if (lane == 0) {
            out[0] = 0;
            out[1] = double(t2 - t1) / reps;
            out[2] = T[20][20] + bad;
        }
    }
}
template <int LIM>
void runLim(double* o) {
    double h[3];
    hipLaunchKernelGGL(k_bench_lim<LIM>, dim3(1), dim3(256), 0, 0, o, 100);
    hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
    printf("grouped broadcasts, EXEC = first %2d lanes   warm %6.0f cycles (%.2f us)   check %.9f\n", LIM, h[1], h[1] / 2400.0, h[2]);
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
    run<4>("LDL^T chain (rcp on chain)", o);
    run<5>("hand-interleaved + sched_barrier", o);
    run<6>("grouped broadcasts", o);
    run<9>("2x2 block pivots", o);
    run<10>("combo: 2 Newton, inline norm", o);
    run<11>("combo: 1 Newton, inline norm", o);
    run<12>("combo: 2 Newton, deferred norm", o);
    run<13>("combo: 1 Newton, deferred norm", o);
    run<0>("readlane bulk again", o);
    runLim<64>(o);
    runLim<48>(o);
    runLim<32>(o);
    {
        double h[3];
        hipLaunchKernelGGL(k_bench2, dim3(1), dim3(256), 0, 0, o, 100);
        hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
        printf("%-28s cold %6.0f cycles   warm %6.0f cycles (%.2f us)   check %.9f   (incl. the copy + 2 barriers per repetition)\n",
               "two waves, column split", h[0], h[1], h[1] / 2400.0, h[2]);
    }
    return 0;
}
