// IMU bursts (gfx950): K consecutive integrateUpToTime steps (VIOFilter.cpp:146-209) -- up to kBurstMax processIMUData
// calls and, optionally, the integrateUpToTime of the processVisionData call that follows them -- in TWO launches, with
// Sigma read and written once.
//
// Every step is the reference's step, in the reference's order: the linearisation of step s is taken at the state after
// step s-1 and Sigma_s = F_s Sigma_{s-1} F_s^T + T_s Q_s is formed exactly as in eqf_propagate.hpp; nothing is composed
// algebraically.  What makes it possible without a grid-wide barrier per step is the structure F = [[F_bb, 0], [L, D]]:
//     Sigma'_bb = F_bb Sigma_bb F_bb^T + ...                          needs Sigma_bb only
//     Sigma'_Ib = (L_I Sigma_bb + D_I Sigma_Ib) F_bb^T + ...          needs Sigma_bb and landmark I's own 3 x 11 panel
//     Sigma'_IJ = (D_I Sigma_IJ + L_I Sigma_bJ) D_J^T + G_I L_J^T     needs its own block and the panels of I and J
// so the 11-wide base panels evolve on their own (k_burst_build, one workgroup per 16 landmarks, which also runs the
// scalar state chain and the landmark group steps and leaves, per step and landmark, a 63-value record), and each 3x3
// landmark block then runs all K steps in registers from those records (k_burst_riccati).
//
// The per-step arithmetic does not depend on how the calls are cut into bursts: a filter replayed with other burst
// boundaries (e.g. after eqf_dump / restore) produces the same bits.
#pragma once
#include "eqf_propagate.hpp"

namespace eqf {

constexpr int kBurstMax = 16;  // steps per burst
constexpr int kBurstLm = 16;   // landmarks per builder workgroup (4 per panel wave)
// per step and landmark (element type T): D, Lw, Lv, Gn, Gv (the row constants, kBlkRec = 45 as in k_build_blocks) and
// Sw = Sigma[0:3, J], Sv = Sigma[8:11, J] (entering the step) for the column side
constexpr int kColRec = 63;
constexpr int kBuildThreads = 512;  // 8 wavefronts, see k_burst_build (a ninth would cap every wave at 168 VGPRs: spills)

struct BurstStep {
    int riccati;  // the step integrates and touches Sigma
    int pad_;
    double TtP;   // T * pointProcessVariance (diagonal process noise of the landmark blocks)
};

struct BurstArgs {
    const Glob* gin;
    Glob* gout;
    const double* p0;   // [B][3][cap]
    const double* Qin;  // [B][5][cap]
    double* Qout;
    const void* Sin;  // (T)
    void* Sout;
    const ImuRec* recs;  // IMU record of step s, filter b: recs[s * recStride + b]; nullptr -> inl[s]
    long long recStride;
    const ImuRec* visRec;  // [B] record of the closing vision step (only its stamp is used); nullptr -> inl[K - 1]
    ImuRec inl[kBurstMax];
    int K;           // steps, the closing vision step included
    int visionLast;  // step K-1 is processVisionData's integrateUpToTime(stamp, true)   (VIOFilter.cpp:233)
    int* errflag;
    long long sigmaStride;
    int cap, ld;
    void* colRec;      // [B][kBurstMax][kColRec][cap]  (T)
    void* rowRec;      // [B][kBurstMax][cap][kBlkRec]  (T)
    BurstStep* steps;  // [B][kBurstMax]
    Params prm;
};

// what stepCommon / stepGlobal / stepLandmark read of their argument block
struct StepView {
    const Params& prm;
    int isImu, doRiccati;
};

EQF_DI void waveSync() {
#ifdef __HIP_DEVICE_COMPILE__
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

template <typename T>
struct BurstLds {
    Glob glob[2];
    StepCommon com[3];
    int ricc[3];
    T F[3][11][12];   // F_bb
    T Nb[3][11][4];   // gyro columns of B, base rows
    T RA[3][9];
    T Tt[3];
    double Q[2][kBurstLm][5];
    T blk[2][kBurstLm][28];  // D, Lw, Lv
    T Sbb[3][11][12];
    T Tb[11][12];
    T P[4][12][12];   // Sigma_Ib of the panel wave's 4 landmarks (12 rows x 11 columns)
    T Gs[4][12][12];  // G_I = L_I Sigma_bb + D_I Sigma_Ib
    ImuRec rec[kBurstMax];
};

// ------------------------------------------------------------------------------------------------
// k_burst_build: grid = (max(1, ceil(N / 16)), B), block = 512 = 8 wavefronts in a software pipeline, one barrier per tick.
// In tick t
//   wave 4      scalar state of step t:            G_{t+1} = stepGlobal(G_t)                           (lane 0)
//               then Sigma_bb after step t-1
//   wave 5      common linearisation values, F_bb and the noise input rows of step t (from G_t)
//   wave 6      group step of step t-1 for the 16 landmarks:  Q_t = Q_{t-1} * lift
//   wave 7      blocks D, Lw, Lv of step t-1 (from Q_{t-1})
//   waves 0..3  panels of step t-2, 4 landmarks each: G rows -> record, Sigma_Ib after the step
// A lone wavefront retires an fp64 instruction every ~6 cycles whatever the dependencies, so the serial chains are laid
// side by side on different SIMDs; the tick is the longest stage (~2.5 k cycles) instead of their sum.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBuildThreads) void k_burst_build(BurstArgs a) {
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int L0 = blockIdx.x * kBurstLm;
    __shared__ BurstLds<T> s;
    const int N = a.gin[b].N, K = a.K, cap = a.cap, ld = a.ld;
    const bool first = blockIdx.x == 0;
    const T* Sin = static_cast<const T*>(a.Sin) + (long long)b * a.sigmaStride;
    T* Sout = static_cast<T*>(a.Sout) + (long long)b * a.sigmaStride;
    const double* p0 = a.p0 + (long long)b * 3 * cap;
    const double* Qin = a.Qin + (long long)b * 5 * cap;
    double* Qout = a.Qout + (long long)b * 5 * cap;
    T* colRec = static_cast<T*>(a.colRec) + (long long)b * kBurstMax * kColRec * cap;
    T* rowRec = static_cast<T*>(a.rowRec) + (long long)b * kBurstMax * cap * kBlkRec;
    int bad = 0;

    // ---- prologue: everything this workgroup reads of the state, issued together
    const int li = L0 + lane;                   // waves 6, 7: lane = landmark
    const bool lmOk = lane < kBurstLm && li < N;
    d3 q0 = mk3(0, 0, 1);
    // panel waves: lane -> (row r = (lane >> 4) + 4u of the wave's 12 panel rows, column c = lane & 15)
    const int pc = lane & 15, pr0 = lane >> 4;
    const int Lw0 = L0 + 4 * wv;  // first landmark of this panel wave
    if (wv < 4) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int r = pr0 + 4 * u, lm = Lw0 + r / 3;
            T v = (T)0;
            if (lm < N && pc < 11) v = Sin[(long long)(kLm0 + 3 * Lw0 + r) * ld + pc];
            if (pc < 12) s.P[wv][r][pc] = v;
        }
    } else if (wv == 4) {
        const double* src = reinterpret_cast<const double*>(a.gin + b);
        double* dst = reinterpret_cast<double*>(&s.glob[0]);
        if (lane < (int)(sizeof(Glob) / 8)) dst[lane] = src[lane];
        for (int e = lane; e < 132; e += 64) {
            const int rr = e / 12, cc = e % 12;
            s.Sbb[0][rr][cc] = (cc < 11) ? Sin[(long long)rr * ld + cc] : (T)0;
        }
    } else if (wv == 5) {
        for (int e = lane; e < K * 8; e += 64) {
            const int st = e >> 3, j = e & 7;
            const ImuRec* rp = (a.visionLast && st == K - 1) ? (a.visRec ? a.visRec + b : &a.inl[st])
                                                             : (a.recs ? a.recs + (long long)st * a.recStride + b : &a.inl[st]);
            reinterpret_cast<double*>(&s.rec[st])[j] = reinterpret_cast<const double*>(rp)[j];
        }
    } else if (wv == 6 || wv == 7) {
        if (lmOk) {
            q0 = mk3(p0[li], p0[cap + li], p0[2 * cap + li]);
            if (wv == 6) {
#pragma unroll
                for (int k = 0; k < 5; ++k) s.Q[0][lane][k] = Qin[k * cap + li];
            }
        }
    }
    __syncthreads();

    const T sw2 = (T)a.prm.velOmegaVariance, sa2 = (T)a.prm.velAccelVariance;
    // One tick loop PER ROLE (the branch on the wave index is outside the loops): inside a common loop the compiler hoists the
    // loop invariants of every role at once and the kernel needs the sum of their registers instead of the maximum.  Every
    // wave passes the same number of barriers.
    if (wv == 4) {
        for (int t = 0; t < K + 2; ++t) {
            if (t < K) {
                const int cur = t & 1;
                {
                    const double* src = reinterpret_cast<const double*>(&s.glob[cur]);
                    double* dst = reinterpret_cast<double*>(&s.glob[cur ^ 1]);
                    if (lane < (int)(sizeof(Glob) / 8)) dst[lane] = src[lane];
                }
                waveSync();
                if (lane == 0) {
                    const StepView v{a.prm, (a.visionLast && t == K - 1) ? 0 : 1, 1};
                    StepCommon c;
                    c.step = 0;
                    stepCommon(s.glob[cur], s.rec[t], v, c, kPartBase, &bad);
                    stepGlobal(s.glob[cur], &s.glob[cur ^ 1], s.rec[t], v, c, &bad);
                }
            }
            const int st = t - 1;
            if (st >= 0 && st < K) {
                const int sl = st % 3, nx = (st + 1) % 3;
                if (!s.ricc[sl]) {
                    for (int e = lane; e < 132; e += 64) s.Sbb[nx][e / 12][e % 12] = s.Sbb[sl][e / 12][e % 12];
                } else {
                    // Sigma'_bb = F_bb Sigma_bb F_bb^T + T (P_bb + B_b R B_b^T)
                    for (int e = lane; e < 121; e += 64) {
                        const int rr = e / 11, cc = e % 11;
                        T acc = 0;
#pragma unroll
                        for (int k = 0; k < 11; ++k) acc += s.F[sl][rr][k] * s.Sbb[sl][k][cc];
                        s.Tb[rr][cc] = acc;
                    }
                    waveSync();
                    const T Tt = s.Tt[sl];
                    for (int e = lane; e < 132; e += 64) {
                        const int rr = e / 12, cc = e % 12;
                        T acc = 0;
                        if (cc < 11) {
#pragma unroll
                            for (int k = 0; k < 11; ++k) acc += s.Tb[rr][k] * s.F[sl][cc][k];
                            T nz = 0;
#pragma unroll
                            for (int k = 0; k < 3; ++k) nz += sw2 * s.Nb[sl][rr][k] * s.Nb[sl][cc][k];
                            if (rr >= 8 && cc >= 8) {  // accel columns of B: rows 8:11 hold R_A
#pragma unroll
                                for (int k = 0; k < 3; ++k) nz += sa2 * s.RA[sl][3 * (rr - 8) + k] * s.RA[sl][3 * (cc - 8) + k];
                            }
                            if (rr == cc) {
                                const Params& p = a.prm;
                                nz += (T)(rr < 3 ? p.biasOmegaProcessVariance
                                                 : (rr < 6 ? p.biasAccelProcessVariance : (rr < 8 ? p.gravityProcessVariance : p.velocityProcessVariance)));
                            }
                            acc += Tt * nz;
                        }
                        s.Sbb[nx][rr][cc] = acc;
                    }
                    waveSync();
                }
            }
            __syncthreads();
        }
    } else if (wv == 5) {
        for (int t = 0; t < K + 2; ++t) {
            if (t < K) {
                const int sl = t % 3;
                if (lane == 0) {
                    const StepView v{a.prm, (a.visionLast && t == K - 1) ? 0 : 1, 1};
                    StepCommon c;
                    c.step = 0;
                    stepCommon(s.glob[t & 1], s.rec[t], v, c, kPartBase | kPartRicc | kPartLift, &bad);
                    s.com[sl] = c;
                    s.ricc[sl] = c.step;
                    s.Tt[sl] = c.step ? (T)c.T : (T)0;
                    if (first) {
                        BurstStep bs;
                        bs.riccati = c.step;
                        bs.pad_ = 0;
                        bs.TtP = c.step ? c.T * a.prm.pointProcessVariance : 0.0;
                        a.steps[b * kBurstMax + t] = bs;
                    }
                }
                waveSync();
                if (s.ricc[sl]) {
                    const StepCommon& c = s.com[sl];
                    // F_bb = I + T * [[0,0,0,0],[-B_g^w,0,0,0],[-B_v^w,-R_A, A_vg, 0]]   (VIOFilter.cpp:178-183)
                    for (int e = lane; e < 132; e += 64) {
                        const int rr = e / 12, cc = e % 12;
                        double f = (rr == cc) ? 1.0 : 0.0;
                        if (rr >= 6 && rr < 8 && cc < 3) f = -c.T * c.Bg[3 * (rr - 6) + cc];
                        if (rr >= 8) {
                            if (cc < 3) f = -c.T * c.Bvw.a[3 * (rr - 8) + cc];
                            else if (cc < 6) f = -c.T * c.RA.a[3 * (rr - 8) + cc - 3];
                            else if (cc < 8) f = c.T * c.Avg[2 * (rr - 8) + cc - 6];
                        }
                        s.F[sl][rr][cc] = (cc < 11) ? (T)f : (T)0;
                        if (cc < 4) {
                            double nb = 0.0;
                            if (cc < 3 && rr >= 6 && rr < 8) nb = c.Bg[3 * (rr - 6) + cc];
                            if (cc < 3 && rr >= 8) nb = c.Bvw.a[3 * (rr - 8) + cc];
                            s.Nb[sl][rr][cc] = (T)nb;
                        }
                    }
                    if (lane < 9) s.RA[sl][lane] = (T)c.RA.a[lane];
                }
            }
            __syncthreads();
        }
    } else if (wv == 6) {
        for (int t = 0; t < K + 2; ++t) {
            const int st = t - 1;
            if (st >= 0 && st < K && lane < kBurstLm) {
                const int cur = st & 1;
                quat Qq = quat{s.Q[cur][lane][0], s.Q[cur][lane][1], s.Q[cur][lane][2], s.Q[cur][lane][3]};
                double Qa = s.Q[cur][lane][4];
                if (lmOk && s.com[st % 3].step) {
                    const StepView v{a.prm, 1, 1};
                    quat Qo;
                    double ao;
                    stepLandmark(s.com[st % 3], v, Qq, Qa, q0, &Qo, &ao, &bad);
                    Qq = Qo;
                    Qa = ao;
                }
                s.Q[cur ^ 1][lane][0] = Qq.w; s.Q[cur ^ 1][lane][1] = Qq.x; s.Q[cur ^ 1][lane][2] = Qq.y; s.Q[cur ^ 1][lane][3] = Qq.z;
                s.Q[cur ^ 1][lane][4] = Qa;
            }
            __syncthreads();
        }
    } else if (wv == 7) {
        for (int t = 0; t < K + 2; ++t) {
            const int st = t - 1;
            if (st >= 0 && st < K && lmOk && s.ricc[st % 3]) {
                const int cur = st & 1;
                const quat Qq = quat{s.Q[cur][lane][0], s.Q[cur][lane][1], s.Q[cur][lane][2], s.Q[cur][lane][3]};
                const double Qa = s.Q[cur][lane][4];
                const LmBlocks blk = buildBlocks(s.com[st % 3], Qq, Qa, q0);
                T* cr = colRec + (long long)st * kColRec * cap + li;
                T* rr = rowRec + ((long long)st * cap + li) * kBlkRec;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const T d = (T)blk.D.a[k], lw = (T)blk.Lw.a[k], lv = (T)blk.Lv.a[k];
                    s.blk[cur][lane][k] = d;
                    s.blk[cur][lane][9 + k] = lw;
                    s.blk[cur][lane][18 + k] = lv;
                    cr[(long long)k * cap] = d;
                    cr[(long long)(9 + k) * cap] = lw;
                    cr[(long long)(18 + k) * cap] = lv;
                    rr[k] = d;
                    rr[9 + k] = lw;
                    rr[18 + k] = lv;
                }
            }
            __syncthreads();
        }
    } else {
        for (int t = 0; t < K + 2; ++t) {
            // ---- panel waves: step t-2
            const int st = t - 2;
            if (st >= 0 && st < K && s.ricc[st % 3] && Lw0 < N) {
                const int sl = st % 3, cur = st & 1;
                T g[3];
                T* cr = colRec + (long long)st * kColRec * cap;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int r = pr0 + 4 * u, jl = r / 3, rr = r - 3 * jl, lmL = 4 * wv + jl, lm = L0 + lmL;
                    const T* bk = s.blk[cur][lmL];
                    T acc = 0;
                    if (pc < 11) {
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            acc += bk[9 + 3 * rr + k] * s.Sbb[sl][k][pc] + bk[18 + 3 * rr + k] * s.Sbb[sl][8 + k][pc] +
                                   bk[3 * rr + k] * s.P[wv][3 * jl + k][pc];
                    }
                    g[u] = acc;
                    if (pc < 12) s.Gs[wv][r][pc] = acc;
                    // records: Gn / Gv for the row side, Sw / Sv (entering the step) for the column side
                    if (lm < N && (pc < 3 || (pc >= 8 && pc < 11))) {
                        const int cc = pc < 3 ? pc : pc - 8;
                        const T gv = pc < 3 ? acc + (sw2 / s.Tt[sl]) * bk[9 + 3 * rr + cc] : acc;
                        const int ko = (pc < 3 ? 27 : 36) + 3 * rr + cc;
                        cr[(long long)ko * cap + lm] = gv;
                        rowRec[((long long)st * cap + lm) * kBlkRec + ko] = gv;
                        cr[(long long)((pc < 3 ? 45 : 54) + 3 * cc + rr) * cap + lm] = s.P[wv][r][pc];
                    }
                }
                waveSync();
                // Sigma'_Ib = G_I F_bb^T - sigma_w^2 Lw_I Nb^T
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int r = pr0 + 4 * u, jl = r / 3, rr = r - 3 * jl, lmL = 4 * wv + jl;
                    const T* bk = s.blk[cur][lmL];
                    if (pc < 11) {
                        T acc = 0;
#pragma unroll
                        for (int k = 0; k < 11; ++k) acc += s.Gs[wv][r][k] * s.F[sl][pc][k];
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc -= sw2 * bk[9 + 3 * rr + k] * s.Nb[sl][pc][k];
                        g[u] = acc;
                    }
                }
                waveSync();
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (pc < 11) s.P[wv][pr0 + 4 * u][pc] = g[u];
            }
            __syncthreads();
        }
    }

    // ---- epilogue
    if (wv < 4) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int r = pr0 + 4 * u, lm = Lw0 + r / 3;
            if (lm < N && pc < 12) {
                const T v = pc < 11 ? s.P[wv][r][pc] : (T)0;
                Sout[(long long)(kLm0 + 3 * Lw0 + r) * ld + pc] = v;   // Sigma_Ib
                Sout[(long long)pc * ld + kLm0 + 3 * Lw0 + r] = v;     // Sigma_bI: its transpose
            }
        }
    } else if (wv == 4) {
        if (first) {
            const double* src = reinterpret_cast<const double*>(&s.glob[K & 1]);
            double* dst = reinterpret_cast<double*>(a.gout + b);
            if (lane < (int)(sizeof(Glob) / 8)) dst[lane] = src[lane];
        }
    } else if (wv == 6) {
        if (lmOk) {
#pragma unroll
            for (int k = 0; k < 5; ++k) Qout[k * cap + li] = s.Q[K & 1][lane][k];
        }
    }
    if (wv == 4) {
        if (first)
            for (int e = lane; e < 144; e += 64) {
                const int rr = e / 12, cc = e % 12;
                Sout[(long long)rr * ld + cc] = (rr < 11 && cc < 11) ? s.Sbb[K % 3][rr][cc] : (T)0;
            }
    }
    if (bad && a.errflag) atomicOr(a.errflag, 1);
}

// ------------------------------------------------------------------------------------------------
// k_burst_riccati: the landmark x landmark blocks, all K steps in registers.
//   Sigma'_IJ = (D_I Sigma_IJ + Lw_I Sw_J + Lv_I Sv_J) D_J^T + Gn_I Lw_J^T + Gv_I Lv_J^T  (+ T p I on the diagonal)
// grid = (ceil(N / 64), ceil(N / (4 R)), B), block = 256: lane = column landmark J, each wavefront owns R row landmarks.
// Per step a lane fetches its 45 column constants (coalesced; the fetch for step s+1 is issued before the arithmetic of
// step s) and the wave its R x 45 row constants (staged in wave-private LDS, read back as broadcasts): no workgroup
// barrier anywhere.  Per block and step 162 FMAs.
// ------------------------------------------------------------------------------------------------
template <typename T, int R>
__global__ __launch_bounds__(256) void k_burst_riccati(BurstArgs a) {
    const int b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = a.gin[b].N, K = a.K, cap = a.cap, ld = a.ld;
    const int J = blockIdx.x * 64 + lane;
    const int I0 = (blockIdx.y * 4 + wv) * R;
    if (I0 >= N) return;  // wave-uniform; the kernel has no workgroup barrier
    const int nI = min(R, N - I0);
    const bool validJ = J < N;
    const int Jc = validJ ? J : 0;
    const T* Sin = static_cast<const T*>(a.Sin) + (long long)b * a.sigmaStride;
    T* Sout = static_cast<T*>(a.Sout) + (long long)b * a.sigmaStride;
    const T* colRec = static_cast<const T*>(a.colRec) + (long long)b * kBurstMax * kColRec * cap + Jc;
    const T* rowRec = static_cast<const T*>(a.rowRec) + (long long)b * kBurstMax * cap * kBlkRec + (long long)I0 * kBlkRec;
    const BurstStep* steps = a.steps + b * kBurstMax;
    constexpr int kRowVals = R * kBlkRec, kRowTrips = (kRowVals + 63) / 64;
    __shared__ T sRow[4][2][kRowVals];

    T S[R][9];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int I = min(I0 + i, N - 1);
        const T* src = Sin + (long long)(kLm0 + 3 * I) * ld + kLm0 + 3 * Jc;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) S[i][3 * rr + cc] = src[(long long)rr * ld + cc];
    }
    T cA[kBlkRec], cB[kBlkRec], rA[kRowTrips], rB[kRowTrips];
    auto fetch = [&](int st, T* c, T* rv) {
        const T* cp = colRec + (long long)st * kColRec * cap;
#pragma unroll
        for (int k = 0; k < 27; ++k) c[k] = cp[(long long)k * cap];
#pragma unroll
        for (int k = 0; k < 18; ++k) c[27 + k] = cp[(long long)(45 + k) * cap];
        const T* rp = rowRec + (long long)st * cap * kBlkRec;
#pragma unroll
        for (int u = 0; u < kRowTrips; ++u) {
            const int e = lane + 64 * u;
            rv[u] = (e < nI * kBlkRec) ? rp[e] : (T)0;
        }
    };
    auto body = [&](int st, const T* c, const T* rv) {
        T* row = sRow[wv][st & 1];
#pragma unroll
        for (int u = 0; u < kRowTrips; ++u) {
            const int e = lane + 64 * u;
            if (e < kRowVals) row[e] = rv[u];
        }
        waveSync();
        const T TtP = (T)steps[st].TtP;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const T* rc = row + i * kBlkRec;  // wave-uniform: LDS broadcast reads
            T H[9];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    T acc = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        acc += rc[3 * rr + k] * S[i][3 * k + cc] + rc[9 + 3 * rr + k] * c[27 + 3 * k + cc] + rc[18 + 3 * rr + k] * c[36 + 3 * k + cc];
                    H[3 * rr + cc] = acc;
                }
            const bool diag = (I0 + i) == J;
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    T acc = (diag && rr == cc) ? TtP : (T)0;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        acc += H[3 * rr + k] * c[3 * cc + k] + rc[27 + 3 * rr + k] * c[9 + 3 * cc + k] + rc[36 + 3 * rr + k] * c[18 + 3 * cc + k];
                    S[i][3 * rr + cc] = acc;
                }
        }
    };
    // first step that touches Sigma, then ping-pong between the two register sets
    int st = 0;
    while (st < K && !steps[st].riccati) ++st;
    if (st < K) fetch(st, cA, rA);
    while (st < K) {
        int nx = st + 1;
        while (nx < K && !steps[nx].riccati) ++nx;
        if (nx < K) fetch(nx, cB, rB);
        body(st, cA, rA);
        st = nx;
        if (st >= K) break;
        nx = st + 1;
        while (nx < K && !steps[nx].riccati) ++nx;
        if (nx < K) fetch(nx, cA, rA);
        body(st, cB, rB);
        st = nx;
    }
    if (validJ) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (i < nI) {
                T* dst = Sout + (long long)(kLm0 + 3 * (I0 + i)) * ld + kLm0 + 3 * J;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) dst[(long long)rr * ld + cc] = S[i][3 * rr + cc];
            }
        }
    }
}

}  // namespace eqf
