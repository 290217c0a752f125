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
// so the 11-wide base panels evolve on their own (k_burst_build, one workgroup per 4 or 16 landmarks, which also runs the
// scalar state chain and the landmark group steps and leaves, per step and landmark, a 63-value record), and each 3x3
// landmark block then runs all K steps in registers from those records (k_burst_riccati_ring).  Only the blocks with I >= J are
// propagated; the others are written as their transposes (round 3).
//
// The per-step arithmetic does not depend on how the calls are cut into bursts: a filter replayed with other burst
// boundaries (e.g. after eqf_dump / restore) produces the same bits.
#pragma once
#include "eqf_propagate.hpp"
#include "eqf_handoff.hpp"

namespace eqf {

constexpr int kBurstMax = 16;  // steps per burst
constexpr int kBurstLmMax = 16;  // landmarks per builder workgroup: LM = 16 (throughput) or 4 (latency), see k_burst_build
constexpr int kLwWave = 1, kSbbWave = 2;  // k_burst_build<.., 4>: waves 1..3 carry no panel and take over three of the stages
// per step and landmark (element type T): D, Lw, Lv, Gn, Gv (the row constants, kBlkRec = 45 as in k_build_blocks) and
// Sw = Sigma[0:3, J], Sv = Sigma[8:11, J] (entering the step) for the column side
constexpr int kColRec = 63;
// Every builder publishes its step count kFlagReplicas times, a few KB apart (one store instruction, one lane per replica), and a block
// workgroup polls replica (tile index mod kFlagReplicas): a hundred pollers on ONE line saturate its memory channel, and the builders'
// write-through stores queue up behind them (measured: ticks of 3 us instead of 1.4).
constexpr int kFlagReplicas = 8;
#ifndef EQF_BURST_PUBLISH_LAG
#define EQF_BURST_PUBLISH_LAG 1  // ticks between a step's write-through stores and its publication (2: behind a counted wait, vmcnt(7); 1: a full drain that stalls 0.2 - 0.5 us of a 1.4 us tick -- and the block workgroups finish 1.4 us earlier: 35.4 -> 34.0 us per burst)
#endif
inline long long burstRecStep(long long elems) { return (elems + 31) / 32 * 32; }
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
    int ringBy;      // k_burst_riccati_ring: row tiles (of 4 R landmarks) of the launch, see ringTiles()
    int visionLast;  // step K-1 is processVisionData's integrateUpToTime(stamp, true)   (VIOFilter.cpp:233)
    int* errflag;
    long long sigmaStride;
    int cap, ld;
    void* colRec;      // [B][kBurstMax][kColRec][cap]  (T)
    void* rowRec;      // [B][kBurstMax][cap][kBlkRec]  (T)
    BurstStep* steps;  // [B][kBurstMax]
    // k_burst_fused (builder and block workgroups in ONE launch): buildFlags[b * nBuildCap + w] = epoch * 32 + s once builder workgroup w
    // has the records of steps 0 .. s-1 in memory (write-through stores, drained); nBuild = builder workgroups per filter
    int* buildFlags;   // [B][kFlagReplicas][nBuildCap (the stride of a replica)]
    int epoch, nBuild, nBuildCap;
    // elements between two steps' records: kColRec * cap and cap * kBlkRec rounded up to whole 128-byte lines (burstRecStep), so that no
    // cache line holds entries of two steps -- the fused launch reads a step's lines through the L2 while later steps are still being written
    int colStep, rowStep;
    Params prm;
    // k_burst_riccati_ring, a burst closed by a vision step whose landmark set the update will find unchanged (round 5): the block
    // workgroups also leave the update's operands that are products of Sigma' blocks -- the landmark columns of C Sigma' (rows 2i, 2i+1 of
    // YW, columns kLm0 ..) and S = C Sigma' C^T + R (SA, the lower block triangle + the blocks inside the diagonal 16 x 16 tiles) -- from the
    // blocks they hold in registers: k_update_prep64's landmark waves then read 12 columns of Sigma' instead of all of them
    // (UpdArgs::csInBurst).  csOut == 0: nothing of this.
    int csOut;
    double* YW;
    double* SA;
    const double* lmc;  // [B][15][cap]: the output matrix C_i of every landmark (rows 0..5)
    int ldY, ldS;
    long long strideY, strideS;
};

// what stepCommon / stepGlobal / stepLandmark read of their argument block
struct StepView {
    const Params& prm;
    int isImu, doRiccati;
};

// LDS-only synchronisation.  __syncthreads() / a release fence also drain the wave's GLOBAL stores (s_waitcnt vmcnt(0):
// ~6 k cycles until the records written in a tick are acknowledged), which nobody in this launch reads.
EQF_DI void waveSync() {
#ifdef __HIP_DEVICE_COMPILE__
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS operations of one wave are performed in order
#endif
}
// flags in LDS: relaxed workgroup-scope atomics.  (A `volatile` LDS access makes the compiler wait for EVERY outstanding memory
// operation of the wave -- s_waitcnt vmcnt(0) lgkmcnt(0) -- i.e. for write-through stores and prefetches that are meant to stay in flight.)
EQF_DI int ldsFlagLoad(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
EQF_DI void ldsFlagStore(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
EQF_DI void ldsBarrier() {
#ifdef __HIP_DEVICE_COMPILE__
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

#ifdef EQF_BURST_STAMPS
__device__ long long g_burstStamps[8][20][4];  // [wave][tick]: work begins / ends (workgroup 0)
#define EQF_BSTAMP(k) do { if (bxIdx == 0 && b == 0 && lane == 0 && t < 20) g_burstStamps[wv][t][k] = wall_clock64(); } while (0)  /* 100 MHz, the same clock on every CU */
#else
#define EQF_BSTAMP(k) do { } while (0)
#endif

// State-independent part of a step: functions of the IMU samples and the stamps alone.  All steps of a burst are done at
// once, one lane per step, so that the serial chain over the steps carries only what really depends on the state.
struct StepPre {
    double dt, T;
    double swT;      // sigma_w^2 / T
    int step, pad_;
    d3 wcur, acur;   // the zero-order-hold sample the step integrates (currentVelocity)
    quat lAq;        // se3Exp(dt w_cur, v) = {lAq, VA v}
    m33 VA;
    d3 vCpre;        // x_CI^ (R_CI wbar)            (linear part of Ad(T_IC^-1)(wbar, .))
    d3 oCcur;        // R_CI w_cur
    d3 vCcurPre;     // x_CI^ (R_CI w_cur)
    quat camq;       // se3Exp(-dt oCcur, v) = {camq, Vc v}
    m33 Vc;
};

template <typename T>
struct BurstLds {
    Glob glob[2];       // generic schedule: the scalar state before / after the step of the tick
    StepCommon com[4];  // rings of 4 (step u in slot u & 3): written in tick u, last read in tick u + 2
    int ricc[4];
    T F[4][11][12];     // F_bb
    T Nb[4][11][4];     // gyro columns of B, base rows
    T RA[4][9];
    T Tt[4];
    T swT[4];           // sigma_w^2 / T
    double Q[2][kBurstLmMax][5];
    T blk[4][kBurstLmMax][28];  // D, Lw, Lv of step u in slot u & 3 (written in tick u + 1, read by the panels in tick u + 2 and -- fused launch -- by the store wave in tick u + 3)
    T g2[4][4][36];     // fused launch, LM = 4: Gn, Gv, Sw, Sv of step u (record entries 27 .. 62), written by the panel wave in tick u + 2
    T Sbb[4][11][12];   // Sigma_bb ENTERING step u in slot u & 3
    T Tb[11][12];
    T Gs[4][4][3][12];  // [panel wave][landmark][row]: G_I = L_I Sigma_bb + D_I Sigma_Ib, exchanged between the column lanes
    ImuRec rec[kBurstMax];
    StepPre pre[kBurstMax];
    unsigned stepMask;
    int handStep;  // (ldsFlagLoad / ldsFlagStore) LM = 4: wave 4 has published R_A / vhat / etahat of this step (wave 3 polls it inside the tick)
};

// The two 11 x 11 pieces every step needs of its common values: F_bb = I + T [[0,0,0,0],[-B_g^w,0,0,0],[-B_v^w,-R_A,A_vg,0]]
// (VIOFilter.cpp:178-183) and the gyro columns of B on the base rows.  One wavefront; every lane owns up to three entries
// of the 11 x 12 arrays, and where an entry comes from (which double of the step's StepCommon, with which sign) is worked
// out once per launch: per step an entry costs one LDS read and a multiply.
struct FMap {
    int off[3];   // index (in doubles, from the start of StepCommon) of the source value, -1: none
    int sgn[3];   // F = diag + sgn * T * src
    int nOff[3];  // Nb = src (or 0 if -1), entries with column < 4 only
    int diag[3];
};
EQF_DI FMap burstFMap(int lane) {
    FMap m;
    constexpr int oBg = offsetof(StepCommon, Bg) / 8, oBvw = offsetof(StepCommon, Bvw) / 8, oRA = offsetof(StepCommon, RA) / 8,
                  oAvg = offsetof(StepCommon, Avg) / 8;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int e = lane + 64 * u, rr = e / 12, cc = e % 12;
        int off = -1, sg = 0, no = -1;
        if (e < 132) {
            if (rr >= 6 && rr < 8 && cc < 3) { off = oBg + 3 * (rr - 6) + cc; sg = -1; no = off; }
            if (rr >= 8) {
                if (cc < 3) { off = oBvw + 3 * (rr - 8) + cc; sg = -1; no = off; }
                else if (cc < 6) { off = oRA + 3 * (rr - 8) + cc - 3; sg = -1; }
                else if (cc < 8) { off = oAvg + 2 * (rr - 8) + cc - 6; sg = 1; }
            }
        }
        m.off[u] = off;
        m.sgn[u] = sg;
        m.nOff[u] = no;
        m.diag[u] = (e < 132 && rr == cc && cc < 11) ? 1 : 0;
    }
    return m;
}
template <typename T>
EQF_DI void burstBuildF(BurstLds<T>& s, int sl, int lane, const FMap& m) {
    const double* src = reinterpret_cast<const double*>(&s.com[sl]);
    const double Tt = s.com[sl].T;
    T* Fd = &s.F[sl][0][0];
    T* Nd = &s.Nb[sl][0][0];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int e = lane + 64 * u;
        if (e < 132) {
            const double v = m.off[u] >= 0 ? src[m.off[u]] : 0.0;
            Fd[e] = (T)((double)m.diag[u] + (double)m.sgn[u] * Tt * v);
            const int rr = e / 12, cc = e - 12 * rr;
            if (cc < 4) Nd[4 * rr + cc] = (T)(m.nOff[u] >= 0 ? v : 0.0);
        }
    }
    if (lane < 9) s.RA[sl][lane] = (T)s.com[sl].RA.a[lane];
}

// Sigma_bb after step u (slot (u + 1) & 3) from Sigma_bb entering it:  F_bb Sigma_bb F_bb^T + T (P_bb + B_b R B_b^T); one
// wavefront, lane = (row, column mod 4): a row of F_bb (then of F_bb Sigma_bb) stays in registers for the lane's three columns.
// diagVar: the process variance of the lane's row (a lane-dependent pick from the kernel arguments inside the tick loop
// becomes a vector load from the argument buffer, a microsecond each)
template <typename T>
EQF_DI void burstStepSbb(BurstLds<T>& s, int u, int lane, T sw2, T sa2, T diagVar) {
    const int sl = u & 3, nx = (u + 1) & 3;
    if (!s.ricc[sl]) {
        for (int e = lane; e < 132; e += 64) s.Sbb[nx][e / 12][e % 12] = s.Sbb[sl][e / 12][e % 12];
        return;
    }
    const int rr = min(lane >> 2, 10), c0 = lane & 3;
    const bool act = lane < 44;
    T fr[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) fr[k] = s.F[sl][rr][k];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int cc = c0 + 4 * j;  // column 11 is the structural pad: F_bb Sigma_bb is zero there
        T acc = fr[0] * s.Sbb[sl][0][cc];
#pragma unroll
        for (int k = 1; k < 11; ++k) acc = fma(fr[k], s.Sbb[sl][k][cc], acc);
        if (act) s.Tb[rr][cc] = acc;
    }
    waveSync();
    T tr[11], nr[3];
#pragma unroll
    for (int k = 0; k < 11; ++k) tr[k] = s.Tb[rr][k];
#pragma unroll
    for (int k = 0; k < 3; ++k) nr[k] = sw2 * s.Nb[sl][rr][k];
    const T Tt = s.Tt[sl];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int cc = c0 + 4 * j, cq = min(cc, 10);
        T acc = tr[0] * s.F[sl][cq][0];
#pragma unroll
        for (int k = 1; k < 11; ++k) acc = fma(tr[k], s.F[sl][cq][k], acc);
        T nz = nr[0] * s.Nb[sl][cq][0];
#pragma unroll
        for (int k = 1; k < 3; ++k) nz = fma(nr[k], s.Nb[sl][cq][k], nz);
        if (rr >= 8 && cc >= 8) {  // accel columns of B: rows 8:11 hold R_A
#pragma unroll
            for (int k = 0; k < 3; ++k) nz = fma(sa2 * s.RA[sl][3 * (rr - 8) + k], s.RA[sl][3 * (cq - 8) + k], nz);
        }
        if (rr == cc) nz += diagVar;
        if (act) s.Sbb[nx][rr][cc] = cc < 11 ? fma(Tt, nz, acc) : (T)0;
    }
    waveSync();
}

// R(w), V(w) of SE3Exp (SE3.cpp:139-164) -- eqf_math.hpp's se3Exp in two halves: {m2q(R), V} here, V v by the caller
EQF_DI void se3ExpParts(d3 w, quat* q, m33* V) {
    double A, B, C;
    expCoefficients(dot3(w, w), &A, &B, &C);
    const m33 wx = skew3(w);
    const m33 wx2 = mul33(wx, wx);
    const m33 R = add33(eye3(), add33(scl33(A, wx), scl33(B, wx2)));
    *V = add33(eye3(), add33(scl33(B, wx), scl33(C, wx2)));
    *q = m2q(R);
}

// The camera-frame part of a step's common values (stepCommon's, from the precomputed halves): v_C, U_C of the current
// sample, SE3Exp(-dt U_C).  One lane.
EQF_DI void burstCommonCam(StepCommon& c, const StepPre& pr, const Params& p) {
    m33 RcI;
#pragma unroll
    for (int i = 0; i < 9; ++i) RcI.a[i] = p.RcamI[i];
    const d3 Rv = mv33(RcI, c.vhat);
    c.vC = add(pr.vCpre, Rv);        // EqFMatrices.cpp:302-304
    c.oCcur = pr.oCcur;
    c.vCcur = add(pr.vCcurPre, Rv);  // VIOGroup.cpp:225
    if (p.useDiscreteVelocityLift) c.camInv = se3{pr.camq, mv33(pr.Vc, scl(-pr.dt, c.vCcur))};
}

// ------------------------------------------------------------------------------------------------
// k_burst_build<T, FAST, LM>: grid = (max(1, ceil(N / LM)), B), block = 512 = 8 wavefronts in a software pipeline, one LDS
// barrier per tick.  A lone wavefront retires an fp64 instruction every ~6 cycles whatever the dependencies, so the serial
// chains are laid side by side on different SIMDs; a tick costs its longest stage instead of their sum.
//
// FAST schedule (every filter of the handle initialised; the usual case).  Prologue: one lane per STEP computes what depends
// on the IMU samples only (StepPre: dt, T, the rotation / V matrices of the two SE3 exponentials, ...).  Then, in tick t
//                LM = 16 (throughput: batches, large N)                  LM = 4 (latency: the launch fits the chip)
//   wave 4       state recurrence of step t on lane 0 with X.A, X.w      the same; R_A, vhat, etahat are handed to wave 3
//                in registers + the camera-frame common values           INSIDE the tick (LDS flag; wave 4 never waits)
//   wave 3       panels                                                  camera-frame common values of step t
//   wave 5       B_g^w, B_v^w, F_bb / noise rows of step t-1,             B_g^w, B_v^w, F_bb / noise rows of step t-1
//                then Sigma_bb after step t-2
//   wave 2       panels                                                  Sigma_bb after step t-2
//   wave 6       group step of step t-1 (lane = landmark):  Q_t = Q_{t-1} * lift
//   wave 7       blocks D, Lv, Lw of step t-1 (from Q_{t-1})              blocks D, Lv
//   wave 1       panels                                                  blocks Lw
//   wave 0       panels of step t-2, 4 landmarks per panel wave (lane = (landmark, base column), the 3 x 11 panel in
//                registers): G rows -> record, Sigma_Ib after the step
// With 16 landmarks per workgroup eight busy waves share four SIMDs and every stage runs ~1.3x slower than with the stages
// spread over the otherwise idle waves of a 4-landmark workgroup (N = 200, one filter: 51 -> 43 us per burst).
// Generic schedule (some filter still waits for its first IMU sample: lazy initialisation, VIOFilter.cpp:122-124, in the
// chain): wave 4 runs stepGlobal on the LDS copy of the scalar state, wave 5 stepCommon + F_bb of step t -- the functions of
// eqf_propagate.hpp as they are (Sigma_bb after step t-1 by wave 4 / wave 2).  Both schedules do the same arithmetic per step.
// ------------------------------------------------------------------------------------------------
// OCC2 (LM = 16 only): built for two workgroups per CU -- 128 instead of 169 registers -- for launches with more workgroups than CUs
// (from 20 filters of N = 200 on): every workgroup is a chain of ticks bound by latency, a second one on the CU runs in its gaps.
// FUSE (FAST, LM = 4, T = double; k_burst_fused): the block workgroups run in the SAME launch and consume a step's records as soon as
// they are in memory.  The record entries then leave through LDS (blk, g2) and ONE wave -- wave 3, idle once the camera-frame values
// of the tick are done -- stores them with write-through stores, drains them a tick later (for free) and publishes the step count.
template <typename T, bool FAST, int LM, bool OCC2, bool FUSE>
EQF_DI void burstBuildBody(const BurstArgs& a, const int bxIdx, const int b) {
    static_assert(LM == 4 || LM == 8 || LM == 16, "role tables exist for 4, 8 and 16 landmarks per workgroup");
    static_assert(!FUSE || (FAST && LM == 4 && sizeof(T) == 8), "the fused launch exists for the latency case only");
    // LM = 4: one panel wave -- the Lw blocks, Sigma_bb and the camera-frame values get wavefronts of their own (waves 1, 2, 3).
    // LM = 8 (round 6: batches whose 8-landmark builders still have a CU each, 5 .. 10 filters of N = 200): two panel waves; Lw and Sigma_bb
    // on waves 2 and 3, the camera-frame values stay with the state recurrence on wave 4 -- six busy stages on eight waves instead of the
    // eight-on-four-SIMDs crowd of LM = 16.
    constexpr bool kLwOwn = LM <= 8, kSbbOwn = LM <= 8, kCamOwn = LM == 4;
    constexpr int kLwW = LM == 4 ? kLwWave : 2, kSbbW = LM == 4 ? kSbbWave : 3;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int L0 = bxIdx * LM;
    __shared__ BurstLds<T> s;
    const Glob& G0 = a.gin[b];
    const int N = G0.N, K = a.K, cap = a.cap, ld = a.ld;
    const bool first = bxIdx == 0;
    const T* Sin = static_cast<const T*>(a.Sin) + (long long)b * a.sigmaStride;
    T* Sout = static_cast<T*>(a.Sout) + (long long)b * a.sigmaStride;
    const double* p0 = a.p0 + (long long)b * 3 * cap;
    const double* Qin = a.Qin + (long long)b * 5 * cap;
    double* Qout = a.Qout + (long long)b * 5 * cap;
    T* colRec = static_cast<T*>(a.colRec) + (long long)b * kBurstMax * a.colStep;
    T* rowRec = static_cast<T*>(a.rowRec) + (long long)b * kBurstMax * a.rowStep;
    int bad = 0;

    // ---- prologue: everything this workgroup reads of the state, issued together
    const int li = L0 + lane;                   // waves 6, 7: lane = landmark
    const bool lmOk = lane < LM && li < N;
    d3 q0 = mk3(0, 0, 1);
    // panel waves: lane = 16 * (landmark of the wave, 0..3) + base column; the landmark's 3 x 11 panel Sigma_Ib lives in the
    // registers of its 11 column lanes for the whole burst
    const int pc = lane & 15, pi = lane >> 4;
    const int plm = L0 + 4 * wv + pi;  // this lane's landmark
    const bool pOk = 4 * wv < LM && plm < N;
    T pP[3] = {(T)0, (T)0, (T)0};
    // wave 4, FAST: the state the recurrence carries, and the constants of the origin
    quat rAq = quat{1, 0, 0, 0};
    d3 rAx = mk3(0, 0, 0), rW = mk3(0, 0, 0), rV0 = mk3(0, 0, 0), rEta0 = mk3(0, 0, 1);
    double rCd[6] = {0, 0, 0, 0, 0, 0};
    if (wv < 4) {
        if (pOk && pc < 11) {
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) pP[rr] = Sin[(long long)(kLm0 + 3 * plm + rr) * ld + pc];
        }
        if (kLwOwn && wv == kLwW && lmOk) q0 = mk3(p0[li], p0[cap + li], p0[2 * cap + li]);
    } else if (wv == 4) {
        if (FAST) {
            rAq = quat{G0.Aq[0], G0.Aq[1], G0.Aq[2], G0.Aq[3]};
            rAx = mk3(G0.Ax[0], G0.Ax[1], G0.Ax[2]);
            rW = mk3(G0.w[0], G0.w[1], G0.w[2]);
            rV0 = mk3(G0.v0[0], G0.v0[1], G0.v0[2]);
            rEta0 = mk3(G0.eta0[0], G0.eta0[1], G0.eta0[2]);
            if (lane == 0) ldsFlagStore(&s.handStep, -1);
            if (lane < 4) {  // constants of the camera offset in every slot of the ring
                StepCommon& c = s.com[lane];
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    c.RIC.a[i] = a.prm.RIC[i];
                    c.RICt.a[i] = a.prm.RICt[i];
                }
                c.xIC = mk3(a.prm.camx[0], a.prm.camx[1], a.prm.camx[2]);
#pragma unroll
                for (int i = 0; i < 6; ++i) c.Avg[i] = -kGravity * G0.cInv[i];
            }
        } else {
            const double* src = reinterpret_cast<const double*>(a.gin + b);
            double* dst = reinterpret_cast<double*>(&s.glob[0]);
            if (lane < (int)(sizeof(Glob) / 8)) dst[lane] = src[lane];
        }
    } else if (wv == 5) {
        for (int e = lane; e < K * 8; e += 64) {
            const int st = e >> 3, j = e & 7;
            const ImuRec* rp = (a.visionLast && st == K - 1) ? (a.visRec ? a.visRec + b : &a.inl[st])
                                                             : (a.recs ? a.recs + (long long)st * a.recStride + b : &a.inl[st]);
            reinterpret_cast<double*>(&s.rec[st])[j] = reinterpret_cast<const double*>(rp)[j];
        }
        for (int e = lane; e < 132; e += 64) {
            const int rr = e / 12, cc = e % 12;
            s.Sbb[0][rr][cc] = (cc < 11) ? Sin[(long long)rr * ld + cc] : (T)0;
        }
        if (FAST) {
#pragma unroll
            for (int i = 0; i < 6; ++i) rCd[i] = G0.cDiff[i];
            // ---- one lane per step: the state-independent part of every step of the burst
            waveSync();
            const int st = lane < K ? lane : K - 1;
            const ImuRec& r = s.rec[st];
            // the sample and the time the step starts from: the previous call's (every step but the last is an IMU call)
            double cv[6];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                cv[i] = st == 0 ? G0.curVel[i] : s.rec[st - (st > 0)].w[i] - G0.bias[i];
                cv[3 + i] = st == 0 ? G0.curVel[3 + i] : s.rec[st - (st > 0)].a[i] - G0.bias[3 + i];
            }
            const double ct = st == 0 ? G0.curTime : s.rec[st - (st > 0)].stamp;
            StepPre o;
            o.dt = r.stamp - ct;
            o.step = (ct >= 0) && (o.dt > 0);  // VIOFilter.cpp:147-152
            const unsigned long long bal = __ballot(o.step && lane < K);
            const unsigned mask = (unsigned)bal;
            if (lane == 0) s.stepMask = mask;
            // every step of a burst is a Riccati step: the accumulators restart after the first one (VIOFilter.cpp:192-193)
            const bool fresh = (mask & ((1u << st) - 1u)) == 0;
            o.T = (fresh ? G0.accTime : 0.0) + o.dt;  // :154
            o.wcur = mk3(cv[0], cv[1], cv[2]);
            o.acur = mk3(cv[3], cv[4], cv[5]);
            o.pad_ = 0;
            o.swT = 0.0;
            if (first && lane < K) {
                BurstStep bs;
                bs.riccati = o.step;
                bs.pad_ = 0;
                bs.TtP = o.step ? o.T * a.prm.pointProcessVariance : 0.0;
                if (FUSE) {  // (read by the block workgroups of this launch: write-through, drained by the __syncthreads() below)
                    double* dst = reinterpret_cast<double*>(a.steps + b * kBurstMax + lane);
                    hoStore8(dst, __hiloint2double(0, bs.riccati));
                    hoStore8(dst + 1, bs.TtP);
                } else {
                    a.steps[b * kBurstMax + lane] = bs;
                }
            }
            if (o.step) {
                const double invT = 1.0 / o.T;
                o.swT = a.prm.velOmegaVariance * invT;
                const d3 wbar = mk3(((fresh ? G0.accVel[0] : 0.0) + o.wcur.x * o.dt) * invT, ((fresh ? G0.accVel[1] : 0.0) + o.wcur.y * o.dt) * invT,
                    ((fresh ? G0.accVel[2] : 0.0) + o.wcur.z * o.dt) * invT);
                m33 RcI;
#pragma unroll
                for (int i = 0; i < 9; ++i) RcI.a[i] = a.prm.RcamI[i];
                const d3 xcI = mk3(a.prm.camIx[0], a.prm.camIx[1], a.prm.camIx[2]);
                o.vCpre = crs(xcI, mv33(RcI, wbar));
                o.oCcur = mv33(RcI, o.wcur);
                o.vCcurPre = crs(xcI, o.oCcur);
                se3ExpParts(scl(o.dt, o.wcur), &o.lAq, &o.VA);  // VIOGroup.cpp:214-217
                if (a.prm.useDiscreteVelocityLift) se3ExpParts(scl(-o.dt, o.oCcur), &o.camq, &o.Vc);  // :226
            }
            if (lane < K) s.pre[lane] = o;
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
    T diagVar;  // process variance of base row lane >> 2 (burstStepSbb)
    {
        const int rr = lane >> 2;
        const double v03 = a.prm.biasOmegaProcessVariance, v36 = a.prm.biasAccelProcessVariance, v68 = a.prm.gravityProcessVariance,
                     v8 = a.prm.velocityProcessVariance;
        diagVar = (T)(rr < 3 ? v03 : (rr < 6 ? v36 : (rr < 8 ? v68 : v8)));
    }
    // One tick loop PER ROLE (the branch on the wave index is outside the loops): inside a common loop the compiler hoists the
    // loop invariants of every role at once and the kernel needs the sum of their registers instead of the maximum.  Every
    // wave passes the same number of barriers.
    if (wv == 4) {
        for (int t = 0; t < K + 2; ++t) {
            EQF_BSTAMP(0);
            if (FAST) {
                if (t < K && lane == 0) {
                    const StepPre& pr = s.pre[t];
                    const int sl = t & 3;
                    StepCommon& c = s.com[sl];
                    const m33 RA = q2m(rAq);
                    const d3 vhat = mtv33(RA, sub(rV0, rW));  // X.A.R().inverse() * (v - w)   (VIOGroup.cpp:26,49)
                    const d3 etahat = mtv33(RA, rEta0);
                    c.step = pr.step;
                    s.ricc[sl] = pr.step;
                    s.Tt[sl] = pr.step ? (T)pr.T : (T)0;
                    s.swT[sl] = (T)pr.swT;
                    if (pr.step) {
                        c.dt = pr.dt;
                        c.T = pr.T;
                        c.RA = RA;
                        c.vhat = vhat;      // (the input blocks B_g^w, B_v^w of F_bb are formed from these by wave 5, one tick later)
                        c.etahat = etahat;
                    }
                    if (kCamOwn) {
                        // the camera-frame velocities of the step are wave 3's (it has no other work): hand R_A, vhat over now
                        waveSync();
                        ldsFlagStore(&s.handStep, t);
                    }
                    if (pr.step) {
                        if (!kCamOwn) burstCommonCam(c, pr, a.prm);
                        EQF_BSTAMP(2);
                        // ---- the group step of the scalar state (stepGlobal's; VIOGroup.cpp:214-222 / :182-187, :95-96)
                        const se3 lA = se3{pr.lAq, mv33(pr.VA, scl(pr.dt, vhat))};
                        d3 lw;
                        if (a.prm.useDiscreteVelocityLift) {
                            const d3 inner = add(vhat, scl(pr.dt, add(add(neg(crs(pr.wcur, vhat)), pr.acur), scl(-kGravity, etahat))));
                            lw = sub(vhat, qrot(lA.q, inner));
                        } else {
                            lw = scl(pr.dt, add(neg(pr.acur), scl(kGravity, etahat)));
                        }
                        const se3 An = se3mul(se3{rAq, rAx}, lA);
                        rW = add(rW, qrot(rAq, lw));
                        rAq = An.q;
                        rAx = An.x;
                    }
                }
            } else {
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
                if (!kSbbOwn && t >= 1 && t - 1 < K) burstStepSbb(s, t - 1, lane, sw2, sa2, diagVar);
            }
            EQF_BSTAMP(1);
            ldsBarrier();
        }
    } else if (wv == 5) {
        const FMap fmap = burstFMap(lane);
        for (int t = 0; t < K + 2; ++t) {
            EQF_BSTAMP(0);
            if (FAST) {
                if (t >= 1 && t - 1 < K && s.ricc[(t - 1) & 3]) {
                    if (lane == 0) {  // B[0:2,0:3] and B[2:5,0:3] of the step  (EqFMatrices.cpp:364-367)
                        StepCommon& c = s.com[(t - 1) & 3];
                        const m33 RA = c.RA;
                        const m33 RAg = mul33(RA, skew3(c.etahat));
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 3; ++j)
                                c.Bg[3 * i + j] = rCd[3 * i] * RAg.a[j] + rCd[3 * i + 1] * RAg.a[3 + j] + rCd[3 * i + 2] * RAg.a[6 + j];
                        c.Bvw = mul33(RA, skew3(c.vhat));
                    }
                    waveSync();
                    burstBuildF(s, (t - 1) & 3, lane, fmap);
                }
                EQF_BSTAMP(2);
                if (!kSbbOwn) {
                    waveSync();
                    if (t >= 2 && t - 2 < K) burstStepSbb(s, t - 2, lane, sw2, sa2, diagVar);
                }
            } else if (t < K) {
                const int sl = t & 3;
                if (lane == 0) {
                    const StepView v{a.prm, (a.visionLast && t == K - 1) ? 0 : 1, 1};
                    StepCommon c;
                    c.step = 0;
                    stepCommon(s.glob[t & 1], s.rec[t], v, c, kPartBase | kPartRicc | kPartLift, &bad);
                    s.com[sl] = c;
                    s.ricc[sl] = c.step;
                    s.Tt[sl] = c.step ? (T)c.T : (T)0;
                    s.swT[sl] = c.step ? (T)(a.prm.velOmegaVariance / c.T) : (T)0;
                    if (first) {
                        BurstStep bs;
                        bs.riccati = c.step;
                        bs.pad_ = 0;
                        bs.TtP = c.step ? c.T * a.prm.pointProcessVariance : 0.0;
                        a.steps[b * kBurstMax + t] = bs;
                    }
                }
                waveSync();
                if (s.ricc[sl]) burstBuildF(s, sl, lane, fmap);
            }
            EQF_BSTAMP(1);
            ldsBarrier();
        }
    } else if (wv == 6) {
        for (int t = 0; t < K + 2; ++t) {
            EQF_BSTAMP(0);
            const int st = t - 1;
            if (st >= 0 && st < K && lane < LM) {
                const int cur = st & 1;
                quat Qq = quat{s.Q[cur][lane][0], s.Q[cur][lane][1], s.Q[cur][lane][2], s.Q[cur][lane][3]};
                double Qa = s.Q[cur][lane][4];
                if (lmOk && s.com[st & 3].step) {
                    const StepView v{a.prm, 1, 1};
                    quat Qo;
                    double ao;
                    stepLandmark(s.com[st & 3], v, Qq, Qa, q0, &Qo, &ao, &bad);
                    Qq = Qo;
                    Qa = ao;
                }
                s.Q[cur ^ 1][lane][0] = Qq.w; s.Q[cur ^ 1][lane][1] = Qq.x; s.Q[cur ^ 1][lane][2] = Qq.y; s.Q[cur ^ 1][lane][3] = Qq.z;
                s.Q[cur ^ 1][lane][4] = Qa;
            }
            EQF_BSTAMP(1);
            ldsBarrier();
        }
    } else if (wv == 7) {
        for (int t = 0; t < K + 2; ++t) {
            EQF_BSTAMP(0);
            const int st = t - 1;
            if (st >= 0 && st < K && lmOk && s.ricc[st & 3]) {
                const int cur = st & 1;
                const quat Qq = quat{s.Q[cur][lane][0], s.Q[cur][lane][1], s.Q[cur][lane][2], s.Q[cur][lane][3]};
                const double Qa = s.Q[cur][lane][4];
                m33 Dm, Lvm;
                buildDLv(s.com[st & 3], Qq, Qa, q0, &Dm, &Lvm);
                T* cr = colRec + (long long)(st * a.colStep) + li;
                T* rr = rowRec + (long long)(st * a.rowStep) + (long long)li * kBlkRec;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const T d = (T)Dm.a[k], lv = (T)Lvm.a[k];
                    s.blk[st & 3][lane][k] = d;
                    s.blk[st & 3][lane][18 + k] = lv;
                    if (!FUSE) {
                        cr[(long long)k * cap] = d;
                        cr[(long long)(18 + k) * cap] = lv;
                        rr[k] = d;
                        rr[18 + k] = lv;
                    }
                }
                if (!kLwOwn) {  // (LM = 16: no wavefront to spare for Lw)
                    const StepCommon& c = s.com[st & 3];
                    const m33 Lwm = buildLw(c.T, c.RICt, c.xIC, Qq, Qa, q0);
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const T lw = (T)Lwm.a[k];
                        s.blk[st & 3][lane][9 + k] = lw;
                        cr[(long long)(9 + k) * cap] = lw;
                        rr[9 + k] = lw;
                    }
                }
            }
            EQF_BSTAMP(1);
            ldsBarrier();
        }
    } else if (kCamOwn && FAST && wv == 3) {
        // ---- the camera-frame common values of step t, inside the tick, as soon as wave 4 has handed R_A / vhat over
        // fused launch: this wave also carries the records out.  Step u's entries are complete in LDS when tick u + 2 ends and are stored at
        // the start of tick u + 3 (while wave 4 is busy with the first part of the recurrence): seven write-through store instructions.
        // Their acknowledgement takes longer than a tick (measured: a drain one tick later stalled 0.5 - 1 us), so a step is published
        // TWO ticks after its stores -- vmcnt counts in order: "at most the previous tick's seven outstanding" means the older ones are
        // in memory -- and the wait costs nothing.
        int* const myFlag = FUSE ? a.buildFlags + ((long long)b * kFlagReplicas + min(lane, kFlagReplicas - 1)) * a.nBuildCap + bxIdx : nullptr;  // lane r: replica r
        const int ebase = a.epoch * 32;
        // (a lone wave is bound by instruction issue: where a lane's seven values come from and go to is worked out once, per step there is
        // one LDS read and one store per value -- the first version did the index arithmetic per step: 0.7 us in front of the camera job)
        const T* srcL[7];   // slot 0 of the LDS ring
        int srcStride[7];   // elements per slot
        double* dstG[7];    // step 0
        int dstStride[7];
        bool on[7];
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            int l, k;
            if (p < 4) {  // column side: [step][entry][landmark]   (every instruction has active lanes: L0 < N)
                const int e = lane + 64 * p;
                k = min(e >> 2, kColRec - 1);
                l = e & 3;
                on[p] = e < 4 * kColRec && L0 + l < N;
                dstG[p] = reinterpret_cast<double*>(colRec) + (long long)k * cap + L0 + l;
                dstStride[p] = a.colStep;
            } else {  // row side: [step][landmark][entry]
                const int e = lane + 64 * (p - 4);
                l = min(e / kBlkRec, 3);
                k = e - l * kBlkRec;
                on[p] = e < 4 * kBlkRec && L0 + l < N;
                k = min(k, kBlkRec - 1);
                dstG[p] = reinterpret_cast<double*>(rowRec) + (long long)(L0 + l) * kBlkRec + k;
                dstStride[p] = a.rowStep;
            }
            srcL[p] = k < 27 ? &s.blk[0][l][k] : &s.g2[0][l][k - 27];
            srcStride[p] = k < 27 ? kBurstLmMax * 28 : 4 * 36;
        }
        auto storeRecords = [&](int u) __attribute__((always_inline)) {  // returns the store instructions issued: 7 or 0
            if (!s.ricc[u & 3]) return 0;
            const int sl = u & 3;
#pragma unroll
            for (int p = 0; p < 7; ++p)
                if (on[p]) hoStore8(dstG[p] + u * dstStride[p], (double)srcL[p][sl * srcStride[p]]);
            return 7;
        };
        int prevStores = 0;
        auto olderStoresDone = [&]() __attribute__((always_inline)) {
            if (prevStores == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else hoDrain();
        };
        for (int t = 0; t < K + 2; ++t) {
            EQF_BSTAMP(0);
            if (FUSE) {
                const int u = t - 3;
                if (EQF_BURST_PUBLISH_LAG == 2 ? u >= 2 : u >= 1) {
                    if (EQF_BURST_PUBLISH_LAG == 2) olderStoresDone();
                    else hoDrain();
                    if (lane < kFlagReplicas) hoPublish(myFlag, ebase + u - (EQF_BURST_PUBLISH_LAG - 1));  // steps 0 .. u-2 (lag 2) / u-1 (lag 1)
                }
                EQF_BSTAMP(2);
                if (u >= 0) prevStores = storeRecords(u);
                EQF_BSTAMP(3);
            }
            if (t < K) {
                while (ldsFlagLoad(&s.handStep) < t) __builtin_amdgcn_s_sleep(2);
                waveSync();
                if (lane == 0 && s.pre[t].step) burstCommonCam(s.com[t & 3], s.pre[t], a.prm);
            }
            EQF_BSTAMP(1);
            ldsBarrier();
        }
        if (FUSE) {  // the last steps: K-2 went out in the last tick, K-1 is complete in LDS now
            if (EQF_BURST_PUBLISH_LAG == 2 && K >= 3) {
                olderStoresDone();
                if (lane < kFlagReplicas) hoPublish(myFlag, ebase + K - 2);
            }
            if (EQF_BURST_PUBLISH_LAG == 1 && K >= 2) {
                hoDrain();
                if (lane < kFlagReplicas) hoPublish(myFlag, ebase + K - 1);
            }
            storeRecords(K - 1);
            hoDrain();
            if (lane < kFlagReplicas) hoPublish(myFlag, ebase + K);
        }
    } else if (kLwOwn && wv == kLwW) {
        // ---- Lw = -T B_i of step t-1: needs nothing of the state but T and the landmark's group element
        for (int t = 0; t < K + 2; ++t) {
            EQF_BSTAMP(0);
            const int st = t - 1;
            if (st >= 0 && st < K && lmOk && s.ricc[st & 3]) {
                const int cur = st & 1;
                const quat Qq = quat{s.Q[cur][lane][0], s.Q[cur][lane][1], s.Q[cur][lane][2], s.Q[cur][lane][3]};
                const double Qa = s.Q[cur][lane][4];
                const StepCommon& c = s.com[st & 3];
                const m33 Lwm = buildLw(c.T, c.RICt, c.xIC, Qq, Qa, q0);
                T* cr = colRec + (long long)(st * a.colStep) + li;
                T* rr = rowRec + (long long)(st * a.rowStep) + (long long)li * kBlkRec;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const T lw = (T)Lwm.a[k];
                    s.blk[st & 3][lane][9 + k] = lw;
                    if (!FUSE) {
                        cr[(long long)(9 + k) * cap] = lw;
                        rr[9 + k] = lw;
                    }
                }
            }
            EQF_BSTAMP(1);
            ldsBarrier();
        }
    } else if (kSbbOwn && wv == kSbbW) {
        // ---- Sigma_bb: after step t-1 (generic schedule) / t-2 (fast schedule: F_bb of a step is built one tick later)
        for (int t = 0; t < K + 2; ++t) {
            EQF_BSTAMP(0);
            const int u = FAST ? t - 2 : t - 1;
            if (u >= 0 && u < K) burstStepSbb(s, u, lane, sw2, sa2, diagVar);
            EQF_BSTAMP(1);
            ldsBarrier();
        }
    } else {
        for (int t = 0; t < K + 2; ++t) {
            EQF_BSTAMP(0);
            // ---- panel waves: step t-2
            const int st = t - 2;
            if (st >= 0 && st < K && s.ricc[st & 3] && 4 * wv < LM && L0 + 4 * wv < N) {
                const int sl = st & 3;
                const T* bk = s.blk[sl][4 * wv + pi];  // D, Lw, Lv of the lane's landmark (the same address for its 16 lanes)
                const int cq = pc < 11 ? pc : 0;
                T sb[6], g[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    sb[k] = s.Sbb[sl][k][cq];
                    sb[3 + k] = s.Sbb[sl][8 + k][cq];
                }
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    T acc = bk[9 + 3 * rr] * sb[0];
                    acc = fma(bk[9 + 3 * rr + 1], sb[1], acc);
                    acc = fma(bk[9 + 3 * rr + 2], sb[2], acc);
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc = fma(bk[18 + 3 * rr + k], sb[3 + k], acc);
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc = fma(bk[3 * rr + k], pP[k], acc);
                    g[rr] = pc < 11 ? acc : (T)0;
                    if (pc < 12) s.Gs[wv][pi][rr][pc] = g[rr];
                }
                // records: Gn / Gv for the row side, Sw / Sv (entering the step) for the column side
                if (pOk && (pc < 3 || (pc >= 8 && pc < 11))) {
                    const int cc = pc < 3 ? pc : pc - 8;
                    T* cr = colRec + (long long)(st * a.colStep) + plm;
                    T* rw = rowRec + (long long)(st * a.rowStep) + (long long)plm * kBlkRec;
                    const T swT = s.swT[sl];
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
                        const T gv = pc < 3 ? fma(swT, bk[9 + 3 * rr + cc], g[rr]) : g[rr];
                        const int ko = (pc < 3 ? 27 : 36) + 3 * rr + cc;
                        if (FUSE) {
                            s.g2[sl][pi][ko - 27] = gv;
                            s.g2[sl][pi][(pc < 3 ? 18 : 27) + 3 * cc + rr] = pP[rr];
                        } else {
                            cr[(long long)ko * cap] = gv;
                            rw[ko] = gv;
                            cr[(long long)((pc < 3 ? 45 : 54) + 3 * cc + rr) * cap] = pP[rr];
                        }
                    }
                }
                waveSync();
                // Sigma'_Ib = G_I F_bb^T - sigma_w^2 Lw_I Nb^T
                T fr[11], nb[3];
#pragma unroll
                for (int k = 0; k < 11; ++k) fr[k] = s.F[sl][cq][k];
#pragma unroll
                for (int k = 0; k < 3; ++k) nb[k] = sw2 * s.Nb[sl][cq][k];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const T* gr = s.Gs[wv][pi][rr];
                    T acc = gr[0] * fr[0];
#pragma unroll
                    for (int k = 1; k < 11; ++k) acc = fma(gr[k], fr[k], acc);
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc = fma(-bk[9 + 3 * rr + k], nb[k], acc);
                    pP[rr] = pc < 11 ? acc : (T)0;
                }
                waveSync();  // (Gs is rewritten in the next tick)
            }
            EQF_BSTAMP(1);
            ldsBarrier();
        }
    }

    // ---- epilogue
    if (wv < 4) {
        if (pOk && pc < 12) {
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                const T v = pc < 11 ? pP[rr] : (T)0;
                Sout[(long long)(kLm0 + 3 * plm + rr) * ld + pc] = v;   // Sigma_Ib
                Sout[(long long)pc * ld + kLm0 + 3 * plm + rr] = v;     // Sigma_bI: its transpose
            }
        }
    } else if (wv == 4) {
        if (first) {
            double* dst = reinterpret_cast<double*>(a.gout + b);
            if (FAST) {
                // the scalar state after the burst: a copy of G0 with what the K calls changed (stepGlobal, field by field)
                const double* src = reinterpret_cast<const double*>(a.gin + b);
                if (lane < (int)(sizeof(Glob) / 8)) dst[lane] = src[lane];
                waveSync();
                if (lane == 0) {
                    Glob* out = a.gout + b;
                    const unsigned mask = s.stepMask;
                    const int nImu = K - (a.visionLast ? 1 : 0);
                    out->Aq[0] = rAq.w; out->Aq[1] = rAq.x; out->Aq[2] = rAq.y; out->Aq[3] = rAq.z;
                    out->Ax[0] = rAx.x; out->Ax[1] = rAx.y; out->Ax[2] = rAx.z;
                    out->w[0] = rW.x; out->w[1] = rW.y; out->w[2] = rW.z;
                    if (mask) {  // VIOFilter.cpp:192-193 (every step of a burst is a Riccati step)
                        out->accTime = 0.0;
#pragma unroll
                        for (int i = 0; i < 6; ++i) out->accVel[i] = 0.0;
                    }
                    if (nImu > 0) {  // VIOFilter.cpp:129-130
                        const ImuRec& r = s.rec[nImu - 1];
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            out->curVel[i] = r.w[i] - G0.bias[i];
                            out->curVel[3 + i] = r.a[i] - G0.bias[3 + i];
                        }
                        out->curTime = r.stamp;
                    }
                    if (a.visionLast) {
                        const int ok = (mask >> (K - 1)) & 1;
                        if (ok) out->curTime = s.rec[K - 1].stamp;  // :207
                        out->updateOk = (ok && G0.initialised) ? 1 : 0;  // :234-236
                    }
                }
            } else {
                const double* src = reinterpret_cast<const double*>(&s.glob[K & 1]);
                if (lane < (int)(sizeof(Glob) / 8)) dst[lane] = src[lane];
            }
        }
    } else if (wv == 5) {
        if (first)
            for (int e = lane; e < 144; e += 64) {
                const int rr = e / 12, cc = e % 12;
                Sout[(long long)rr * ld + cc] = (rr < 11 && cc < 11) ? s.Sbb[K & 3][rr][cc] : (T)0;
            }
    } else if (wv == 6) {
        if (lmOk) {
#pragma unroll
            for (int k = 0; k < 5; ++k) Qout[k * cap + li] = s.Q[K & 1][lane][k];
        }
    }
    if (bad && a.errflag) atomicOr(a.errflag, 1);
}
template <typename T, bool FAST, int LM, bool OCC2 = false>
__global__ __launch_bounds__(kBuildThreads) __attribute__((amdgpu_waves_per_eu(OCC2 ? 4 : 2, OCC2 ? 4 : 2))) void k_burst_build(BurstArgs a) {
    burstBuildBody<T, FAST, LM, OCC2, false>(a, (int)blockIdx.x, (int)blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// k_burst_riccati_ring: the landmark x landmark blocks of a SMALL problem (the launch cannot fill the chip, every wave is
// bound by latency).  The records were written by k_burst_build on other XCDs, so the first touch of every step's
// constants is a ~2 us miss: a one-step register prefetch cannot cover it.  Here the four wavefronts of a workgroup share
// their 64 column landmarks (one row landmark each): each wave fetches a QUARTER of a step's 45 x 64 column constants, TWO
// steps ahead (two register sets of 13 values), and passes them on through a two-slot LDS ring one step before they are
// used -- loads have two full steps to arrive.  grid = (ringTiles(N, 1), B), block = 256.
// ------------------------------------------------------------------------------------------------
#ifdef EQF_BURST_STAMPS
__device__ long long g_ringStamps[4][64];
#define EQF_RSTAMP(i) do { if (tileIdx == 9 && b == 0 && lane == 0 && (i) < 64) g_ringStamps[wv][i] = wall_clock64(); } while (0)
#else
#define EQF_RSTAMP(i) do { } while (0)
#endif
// tiles of the launch for nmx landmarks: sum over the column tiles bx of the row tiles by >= 16 bx / R
inline int ringTiles(int nmx, int R, int* rowTiles) {
    const int nBy = (nmx + 4 * R - 1) / (4 * R), nBx = (nmx + 63) / 64;
    int t = 0;
    for (int bx = 0; bx < nBx; ++bx) t += std::max(nBy - bx * (16 / R), 0);
    *rowTiles = nBy;
    return t;
}
constexpr int kRingTrips = 12;  // column-constant rows per wave and step: 45 rows over 4 waves
// R = row landmarks per wavefront: 1 for the small launches described above; 4 for launches that fill the chip (2 in between: twice the
// workgroups of 4, for launches whose four-row tiles would leave CUs with one workgroup or none) -- there the point
// of the ring is that the 45 x 64 column constants of a step are fetched ONCE per workgroup (16 row landmarks) instead of once per
// wavefront, and that a lane holds 13 + 3 prefetched values per register set instead of 46: 2 wavefronts per SIMD instead of 1.
// grid = (ringTiles(N, R), B).
template <typename T, int R, bool AHEAD2 = (R == 1)>
EQF_DI void burstRingBody(const BurstArgs& a, const int tileIdx, const int b) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = a.gin[b].N, K = a.K, cap = a.cap, ld = a.ld;
    // Only the blocks on and below the diagonal (row landmark >= column landmark) are propagated; the others are their transposes and
    // are written as such at the end (below).  The launch enumerates the tiles that hold such blocks and nothing else (column tile bx
    // keeps the row tiles from by0 = 16 bx / R on: at N = 200 that is 28 of 52 tiles, and of the ragged last column tile -- 8 of 64
    // lanes -- only one), so that consecutive workgroups, which go to consecutive XCDs, all carry the same work.
    int bx = 0, by = tileIdx;
    for (;; ++bx) {
        const int cnt = a.ringBy - bx * (16 / R);
        if (by < cnt) break;
        by -= cnt;
    }
    by += bx * (16 / R);
    const int J = bx * 64 + lane;
    const int I0raw = (by * 4 + wv) * R;
    const bool validJ = J < N;
    const int I0 = min(I0raw, max(N - 1, 0)), Jc = validJ ? J : 0;  // (rows past N: the wave works on a copy of the last row, stores nothing)
    const int nI = max(min(R, N - I0), 1);
    const T* Sin = static_cast<const T*>(a.Sin) + (long long)b * a.sigmaStride;
    T* Sout = static_cast<T*>(a.Sout) + (long long)b * a.sigmaStride;
    const T* colRec = static_cast<const T*>(a.colRec) + (long long)b * kBurstMax * a.colStep + Jc;
    const T* rowRec = static_cast<const T*>(a.rowRec) + (long long)b * kBurstMax * a.rowStep + (long long)I0 * kBlkRec;
    const BurstStep* steps = a.steps + b * kBurstMax;
    constexpr int kRowVals = R * kBlkRec, kRowTrips = (kRowVals + 63) / 64;
    __shared__ T sCol[2][kBlkRec][64];
    __shared__ T sRow[4][2][kRowVals + 3];
    __shared__ T sTtP[kBurstMax];
    __shared__ int sRicc[kBurstMax];

    EQF_RSTAMP(63);
    T S[R][9];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int I = min(I0 + i, max(N - 1, 0));
        const T* src = Sin + (long long)(kLm0 + 3 * I) * ld + kLm0 + 3 * Jc;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) S[i][3 * rr + cc] = src[(long long)rr * ld + cc];
    }
    if (tid < K) {
        sTtP[tid] = (T)steps[tid].TtP;
        sRicc[tid] = steps[tid].riccati;
    }
    // this wave's share of a step's column constants: ring rows q = wv + 4 j  (source row q, or q + 18 for Sw / Sv), and the
    // row constants of its own R landmarks
    T xA[kRingTrips + kRowTrips], xB[kRingTrips + kRowTrips];
    auto fetch = [&](int st, T* x) __attribute__((always_inline)) {
        const int sc = min(st, K - 1);  // (past the end: re-read the last step, nobody uses it)
        const T* cp = colRec + (long long)(sc * a.colStep);
#pragma unroll
        for (int j = 0; j < kRingTrips; ++j) {
            const int q = min(wv + 4 * j, kBlkRec - 1);
            x[j] = cp[(long long)(q < 27 ? q : q + 18) * cap];
        }
        const T* rp = rowRec + (long long)(sc * a.rowStep);
#pragma unroll
        for (int u = 0; u < kRowTrips; ++u) x[kRingTrips + u] = rp[min(lane + 64 * u, nI * kBlkRec - 1)];
    };
    auto pass = [&](int st, const T* x) __attribute__((always_inline)) {
        const int sl = st & 1;
#pragma unroll
        for (int j = 0; j < kRingTrips; ++j) {
            const int q = wv + 4 * j;
            if (q < kBlkRec) sCol[sl][q][lane] = x[j];
        }
#pragma unroll
        for (int u = 0; u < kRowTrips; ++u) {
            const int e = lane + 64 * u;
            if (e < kRowVals) sRow[wv][sl][e] = x[kRingTrips + u];
        }
    };
    // R > 1: the row constants are wave-uniform, so they come through the SCALAR cache (constant address space: s_load into SGPRs,
    // one scalar operand per FMA) instead of 180 LDS broadcast reads per step and wave -- with 8 waves on a CU those reads alone kept
    // the LDS pipe busy longer than the FMAs keep a SIMD (7.6 k against 5.2 k cycles per step).  The vector fetch of the same records
    // one step ahead (fetch / pass above) stays as the L2 warm-up: the records were written by k_burst_build on other XCDs.
    typedef const T __attribute__((address_space(4))) CT;
    const int I0u = __builtin_amdgcn_readfirstlane(I0);
    const int nIu = __builtin_amdgcn_readfirstlane(nI);
    CT* const rowC = (CT*)(unsigned long long)(static_cast<const T*>(a.rowRec) + (long long)b * kBurstMax * a.rowStep + (long long)I0u * kBlkRec);
    auto rowConst = [&](int st, int i) __attribute__((always_inline)) {
        if constexpr (R > 1) return rowC + (long long)(st * a.rowStep) + (long long)min(i, nIu - 1) * kBlkRec;
        else return (const T*)(sRow[wv][st & 1] + i * kBlkRec);  // wave-uniform: LDS broadcast reads
    };
    auto math = [&](int st) __attribute__((always_inline)) {
        const int sl = st & 1;
        // two passes over the wave's rows so that only one group of column constants is live at a time (Sw / Sv, then D^T / Lw /
        // Lv): H_i = D_i S_i + Lw_i Sw + Lv_i Sv takes the place of S_i, then S_i' = H_i D^T + Gn_i Lw^T + Gv_i Lv^T
        {
            T Sw[9], Sv[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                Sw[k] = sCol[sl][27 + k][lane];
                Sv[k] = sCol[sl][36 + k][lane];
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const auto* rc = rowConst(st, i);
                T H[9];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        T acc = rc[3 * rr] * S[i][cc];
#pragma unroll
                        for (int k = 1; k < 3; ++k) acc = fma(rc[3 * rr + k], S[i][3 * k + cc], acc);
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc = fma(rc[9 + 3 * rr + k], Sw[3 * k + cc], acc);
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc = fma(rc[18 + 3 * rr + k], Sv[3 * k + cc], acc);
                        H[3 * rr + cc] = acc;
                    }
#pragma unroll
                for (int k = 0; k < 9; ++k) S[i][k] = H[k];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const T TtP = sTtP[st];
        {
            T c[9];  // D_J^T
#pragma unroll
            for (int k = 0; k < 9; ++k) c[k] = sCol[sl][k][lane];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const bool diag = (I0raw + i) == J;
                T O[9];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        T acc = (diag && rr == cc) ? TtP : (T)0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc = fma(S[i][3 * rr + k], c[3 * cc + k], acc);
                        O[3 * rr + cc] = acc;
                    }
#pragma unroll
                for (int k = 0; k < 9; ++k) S[i][k] = O[k];
            }
        }
#pragma unroll
        for (int grp = 1; grp < 3; ++grp) {  // + Gn_i Lw_J^T, then + Gv_i Lv_J^T (the same summation order as one pass)
            __builtin_amdgcn_sched_barrier(0);
            T c[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) c[k] = sCol[sl][9 * grp + k][lane];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const auto* rc = rowConst(st, i) + 18 + 9 * grp;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        T acc = S[i][3 * rr + cc];
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc = fma(rc[3 * rr + k], c[3 * cc + k], acc);
                        S[i][3 * rr + cc] = acc;
                    }
            }
        }
    };
    // R > 1: the same arithmetic, element by element in the same order, with the row constants as SCALAR operands.  A step's R x 45
    // constants are consumed as 5 R chunks of nine (D_i, Lw_i, Lv_i for the R rows, then Gn_i, then Gv_i), two chunks to a group;
    // group g + 1 is requested once group g has arrived and before its 54 FMAs per lane are issued (scalar loads return out of order,
    // so the only wait there is is "all of them": a request must never be outstanding when the previous group is waited for).
    auto mathS = [&](int st) __attribute__((always_inline)) {
      if constexpr (R > 1) {
        static_assert(R == 1 || R % 2 == 0, "5 R chunks in groups of two");
        const int sl = st & 1;
        CT* const base = rowC + (long long)(st * a.rowStep);
        auto chunkPtr = [&](int c) __attribute__((always_inline)) {
            const int i = c < 3 * R ? c / 3 : (c - 3 * R) % R;
            const int off = c < 3 * R ? 9 * (c % 3) : (c < 4 * R ? 27 : 36);
            return base + min(i, nIu - 1) * kBlkRec + off;
        };
        T buf[2][18];
        auto request = [&](int g) __attribute__((always_inline)) {
            CT* p0 = chunkPtr(2 * g);
            CT* p1 = chunkPtr(2 * g + 1);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                buf[g & 1][k] = p0[k];
                buf[g & 1][9 + k] = p1[k];
            }
        };
        auto arrived = [&](int g) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 18; ++k) asm volatile("" ::"s"(buf[g & 1][k]));
        };
        T H[9], cst[18];
        const T TtP = sTtP[st];
        auto chunk = [&](int c, const T* k9) __attribute__((always_inline)) {
            if (c < 3 * R) {
                const int i = c / 3, part = c % 3;
                const T* src = part == 0 ? S[i] : cst + 9 * (part - 1);  // S_i, Sw, Sv
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        T acc = part == 0 ? k9[3 * rr] * src[cc] : fma(k9[3 * rr], src[cc], H[3 * rr + cc]);
#pragma unroll
                        for (int k = 1; k < 3; ++k) acc = fma(k9[3 * rr + k], src[3 * k + cc], acc);
                        H[3 * rr + cc] = acc;
                    }
                if (part == 2) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) S[i][k] = H[k];
                }
            } else {
                const int i = (c - 3 * R) % R;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        T acc = S[i][3 * rr + cc];
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc = fma(k9[3 * rr + k], cst[3 * cc + k], acc);
                        S[i][3 * rr + cc] = acc;
                    }
            }
        };
        request(0);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            cst[k] = sCol[sl][27 + k][lane];
            cst[9 + k] = sCol[sl][36 + k][lane];
        }
#pragma unroll
        for (int g = 0; g < 5 * R / 2; ++g) {
            arrived(g);
            if (g + 1 < 5 * R / 2) request(g + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (g == 3 * R / 2) {
                // all of H_i = D_i S_i + Lw_i Sw + Lv_i Sv are in place: S_i' = H_i D_J^T (+ the landmark process noise on the diagonal)
#pragma unroll
                for (int k = 0; k < 9; ++k) cst[k] = sCol[sl][k][lane];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const bool diag = (I0raw + i) == J;
                    T O[9];
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {
                            T acc = (diag && rr == cc) ? TtP : (T)0;
#pragma unroll
                            for (int k = 0; k < 3; ++k) acc = fma(S[i][3 * rr + k], cst[3 * cc + k], acc);
                            O[3 * rr + cc] = acc;
                        }
#pragma unroll
                    for (int k = 0; k < 9; ++k) S[i][k] = O[k];
                }
#pragma unroll
                for (int k = 0; k < 9; ++k) cst[k] = sCol[sl][9 + k][lane];  // Lw_J, for + Gn_i Lw_J^T
            }
            if (g == 2 * R) {
#pragma unroll
                for (int k = 0; k < 9; ++k) cst[k] = sCol[sl][18 + k][lane];  // Lv_J, for + Gv_i Lv_J^T
            }
            chunk(2 * g, buf[g & 1]);
            chunk(2 * g + 1, buf[g & 1] + 9);
            __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    // AHEAD2 for R = 2 (round 6, measured and NOT the default): two register sets like R = 1, a step's constants requested TWO steps ahead.
    // The idea: with one set the loads of step st + 2 have one step's arithmetic to arrive, from records the builder launch wrote on other
    // XCDs a moment ago.  Measured at 4 .. 12 filters of N = 200 (profiles/r06_burst_shapes.txt): 1.5 - 2.7 us SLOWER per burst (8 filters
    // 64.6 -> 67.3 us) -- the step is not waiting for these loads; 28 more registers and a longer prologue cost more than they hide.
    auto stepMath = [&](int st) __attribute__((always_inline)) {
        if constexpr (R > 1) mathS(st);
        else math(st);
    };
    if (!AHEAD2) {
        // throughput variant: one register set, constants fetched one step ahead (the other wavefront of the SIMD covers the wait)
        fetch(0, xA);
        __builtin_amdgcn_sched_barrier(0);
        pass(0, xA);
        fetch(1, xA);
        __builtin_amdgcn_sched_barrier(0);
        ldsBarrier();
        for (int st = 0; st < K; ++st) {
            if (sRicc[st]) mathS(st);
            pass(st + 1, xA);
            fetch(st + 2, xA);
            __builtin_amdgcn_sched_barrier(0);
            ldsBarrier();
        }
    } else {
    // prologue: steps 0 and 1 in flight, step 0 handed to the ring, step 2 issued
    fetch(0, xA);
    fetch(1, xB);
    __builtin_amdgcn_sched_barrier(0);
    pass(0, xA);
    fetch(2, xA);
    __builtin_amdgcn_sched_barrier(0);
    EQF_RSTAMP(0);
    ldsBarrier();
    EQF_RSTAMP(1);
    for (int st = 0; st < K; st += 2) {
        // even step: its constants are in the ring; xB holds step st+1 (issued two steps ago), xA step st+2
        if (sRicc[st]) stepMath(st);
        EQF_RSTAMP(2 + 3 * st);
        pass(st + 1, xB);
        EQF_RSTAMP(3 + 3 * st);
        fetch(st + 3, xB);
        __builtin_amdgcn_sched_barrier(0);
        ldsBarrier();
        EQF_RSTAMP(4 + 3 * st);  // (LDS only: __syncthreads() would also drain the prefetches, loads and stores share vmcnt)
        if (st + 1 >= K) break;
        if (sRicc[st + 1]) stepMath(st + 1);
        EQF_RSTAMP(2 + 3 * (st + 1));
        pass(st + 2, xA);
        EQF_RSTAMP(3 + 3 * (st + 1));
        fetch(st + 4, xA);
        __builtin_amdgcn_sched_barrier(0);
        ldsBarrier();
        EQF_RSTAMP(4 + 3 * (st + 1));  // (LDS only: __syncthreads() would also drain the prefetches, loads and stores share vmcnt)
    }
    }
    if (validJ) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int I = I0raw + i;
            if (I < N && I >= J) {
                T* dst = Sout + (long long)(kLm0 + 3 * I) * ld + kLm0 + 3 * J;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) dst[(long long)rr * ld + cc] = S[i][3 * rr + cc];
            }
        }
    }
    // The update's operands from the same registers (BurstArgs::csOut): (C Sigma')_{IJ} = C_I Sigma'_IJ, a lane's three consecutive values of
    // two rows of YW, and S_IJ = (C Sigma')_IJ C_J^T (+ R on the diagonal) -- in k_update_prep64's expression order.  The blocks above the
    // diagonal, (C Sigma')_{JI} = C_J Sigma'_IJ^T, go out with the mirror image below; S is only read on and below the diagonal, except
    // inside the 16 x 16 tiles ON it (landmarks of one group of 8), which the lane stores itself.
    __shared__ double sCJ[6][64];
    const bool csOut = a.csOut != 0;
    double* const YWb = csOut ? a.YW + (long long)b * a.strideY : nullptr;
    if (csOut) {
        const double* lmc = a.lmc + (long long)b * 15 * cap;
        double* SAb = a.SA + (long long)b * a.strideS;
        double CJ[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) CJ[q] = lmc[(long long)q * cap + Jc];
        if (wv == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) sCJ[q][lane] = CJ[q];
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int I = I0raw + i;
            if (validJ && I < N && I >= J) {
                double CI[6], v[9];
#pragma unroll
                for (int q = 0; q < 6; ++q) CI[q] = lmc[(long long)q * cap + I];
#pragma unroll
                for (int k = 0; k < 9; ++k) v[k] = (double)S[i][k];
                double cs[6];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) cs[3 * r + c] = dot3(CI[3 * r], v[c], CI[3 * r + 1], v[3 + c], CI[3 * r + 2], v[6 + c]);
                double* yw = YWb + (long long)(2 * I) * a.ldY + kLm0 + 3 * J;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) yw[(long long)r * a.ldY + c] = cs[3 * r + c];
                double sb[4];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int t = 0; t < 2; ++t) sb[2 * r + t] = dot3(cs[3 * r], CJ[3 * t], cs[3 * r + 1], CJ[3 * t + 1], cs[3 * r + 2], CJ[3 * t + 2]);
                if (I == J) {
                    sb[0] += a.prm.measurementVariance;
                    sb[3] += a.prm.measurementVariance;
                }
                double* sa = SAb + (long long)(2 * I) * a.ldS + 2 * J;
                sa[0] = sb[0]; sa[1] = sb[1]; sa[a.ldS] = sb[2]; sa[a.ldS + 1] = sb[3];
                if (I > J && (I >> 3) == (J >> 3)) {
                    // S_JI = (C_J Sigma'_JI) C_I^T, Sigma'_JI = Sigma'_IJ^T
                    double cm[6];
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) cm[3 * r + c] = dot3(CJ[3 * r], v[3 * c], CJ[3 * r + 1], v[3 * c + 1], CJ[3 * r + 2], v[3 * c + 2]);
                    double* sm = SAb + (long long)(2 * J) * a.ldS + 2 * I;
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            sm[(long long)r * a.ldS + t] = dot3(cm[3 * r], CI[3 * t], cm[3 * r + 1], CI[3 * t + 1], cm[3 * r + 2], CI[3 * t + 2]);
                }
            }
        }
    }
    // Sigma_JI = Sigma_IJ^T for the blocks strictly below the diagonal.  Written from the registers, a lane's three rows would be 8-byte
    // stores 3 ld apart (64 cache lines per instruction: the kernel took LONGER than with every block computed twice); instead the
    // workgroup transposes its tile through the LDS ring (free now), kLanesPass column landmarks at a time, and writes rows of the
    // mirror image -- the 4 R row landmarks of the workgroup side by side, 12 R consecutive values.
    constexpr int kLanesPass = R == 4 ? 32 : 64, kWd = 12 * R, kPitch = kWd + 1, kRowsPass = 3 * kLanesPass;
    static_assert(kRowsPass * kPitch <= 2 * kBlkRec * 64, "the staging tile lives in sCol");
    T* const stage = &sCol[0][0][0];
    for (int p = 0; p < 64 / kLanesPass; ++p) {
        __syncthreads();
        if (lane / kLanesPass == p) {
            T* dst = stage + 3 * (lane % kLanesPass) * kPitch + 3 * wv * R;
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) dst[cc * kPitch + 3 * i + rr] = S[i][3 * rr + cc];
        }
        __syncthreads();
        for (int e = tid; e < kRowsPass * kWd; e += 256) {
            const int r = e / kWd, c = e % kWd;
            const int Jm = bx * 64 + p * kLanesPass + r / 3, Im = by * 4 * R + c / 3;
            if (Im < N && Im > Jm) Sout[(long long)(kLm0 + 3 * Jm + r % 3) * ld + kLm0 + 3 * Im + c % 3] = stage[r * kPitch + c];
        }
        if (csOut) {
            // rows 2 Jm, 2 Jm + 1 of C Sigma' at the workgroup's row landmarks: C_Jm times the three staged rows of Sigma'_{Jm, .}
            for (int e = tid; e < 2 * kLanesPass * kWd; e += 256) {
                const int r2 = e / kWd, c = e % kWd, jl = r2 >> 1, rr = r2 & 1;
                const int Jm = bx * 64 + p * kLanesPass + jl, Im = by * 4 * R + c / 3;
                if (Im < N && Im > Jm) {
                    const T* st3 = stage + 3 * jl * kPitch + c;
                    const double* cj = &sCJ[3 * rr][p * kLanesPass + jl];
                    YWb[(long long)(2 * Jm + rr) * a.ldY + kLm0 + 3 * Im + c % 3] =
                        dot3(cj[0], (double)st3[0], cj[64], (double)st3[kPitch], cj[128], (double)st3[2 * kPitch]);
                }
            }
        }
    }
}
template <typename T, int R = 1, bool AHEAD2 = (R == 1)>
__global__ __launch_bounds__(256, R == 1 ? 1 : 2) void k_burst_riccati_ring(BurstArgs a) {
    burstRingBody<T, R, AHEAD2>(a, (int)blockIdx.x, (int)blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// k_burst_fused: builder and block workgroups of the latency case (4 landmarks per builder, one row landmark per arithmetic wave) in
// ONE launch.  grid = (nBuild + ringTiles(N, 1), B), block = 512; workgroups [0, nBuild) are builders -- dispatched first, and they
// never wait for anybody, so the launch cannot deadlock whatever is resident -- the others propagate a tile of 4 x 64 blocks and trail
// the builders by the publication lag (five ticks) plus one fetch instead of by a whole launch.
//
// A block workgroup has seven live wavefronts and no workgroup barrier in its step loop:
//   waves 0..3  arithmetic, one row landmark each (the step of k_burst_riccati_ring<T, 1>, operation for operation: same bits);
//   wave 4      the poller: one lane per builder flag (the flags of a replica are contiguous: a poll is a line or two), publishes in LDS
//               how many steps ALL builders of the filter have in memory.  A wave that polled with its own loads would wait for its
//               prefetches at every poll (vector loads return in order);
//   waves 5, 6  the fetchers (even / odd steps): the step's 45 x 64 column constants and 4 x 45 row constants into one of three LDS slots,
//               as soon as the poller says the step is there -- independent of where the arithmetic is.  Plain loads, through the L2 (which
//               fetches a line once per XCD: with agent-scope loads every workgroup's every load was served by the memory side, 10 us per
//               step).  No stale line can be hit: the launch starts with the L2s invalidated, a line of step s is only touched once ALL
//               builders have step s in memory, and no line holds entries of two steps (BurstArgs::colStep / rowStep).
// Hand-over inside the workgroup is by LDS counters (ldsFlagLoad / ldsFlagStore): sHave (poller -> fetchers), sReady[f] (fetcher f ->
// arithmetic: its steps delivered), sDone[w] (arithmetic wave w -> fetchers: a slot is rewritten three steps later).
// ------------------------------------------------------------------------------------------------
template <typename T>
EQF_DI void burstRingFusedBody(const BurstArgs& a, const int tileIdx, const int b) {
    static_assert(sizeof(T) == 8, "the fused launch exists for fp64");
    constexpr int kSlots = 3;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = a.gin[b].N, K = a.K, cap = a.cap, ld = a.ld;
    int bx = 0, by = tileIdx;  // (the tiles on and below the diagonal, as in burstRingBody with R = 1)
    for (;; ++bx) {
        const int cnt = a.ringBy - bx * 16;
        if (by < cnt) break;
        by -= cnt;
    }
    by += bx * 16;
    const int J = bx * 64 + lane;
    const bool validJ = J < N;
    const int Jc = validJ ? J : 0;
    __shared__ T sCol[kSlots][kBlkRec][64];
    __shared__ T sRow[kSlots][4][kBlkRec + 3];
    __shared__ T sTtP[kBurstMax];
    __shared__ int sRicc[kBurstMax];
    __shared__ int sHave, sReady[2], sDone[4], sAbort;
    if (tid < 7) ldsFlagStore(tid == 0 ? &sHave : (tid < 3 ? &sReady[tid - 1] : &sDone[tid - 3]), 0);
    if (tid == 7) ldsFlagStore(&sAbort, 0);
    ldsBarrier();  // (the only barrier before the mirror-image pass: every live wave, i.e. 0 .. 6)
    const T* const colRec = static_cast<const T*>(a.colRec) + (long long)b * kBurstMax * a.colStep + Jc;
    const T* const rowRec = static_cast<const T*>(a.rowRec) + (long long)b * kBurstMax * a.rowStep;

    if (wv == 4) {
        // ---- poller
        const int ebase = a.epoch * 32;
        const int* const flags = a.buildFlags + ((long long)b * kFlagReplicas + tileIdx % kFlagReplicas) * a.nBuildCap;
        const long long t0 = wall_clock64();
        int have = 0, polls = 0;
        while (have < K) {
            int v = 1 << 20;
            for (int w = lane; w < a.nBuild; w += 64) v = min(v, __hip_atomic_load(flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ebase);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));  // (a flag of an earlier burst reads negative)
            if (v > have) {
                have = min(v, K);
                if (lane == 0) ldsFlagStore(&sHave, have);
                continue;
            }
            __builtin_amdgcn_s_sleep(12);  // ~0.3 us: a tick of the builders takes 1.4
            if ((++polls & 63) == 0 && wall_clock64() - t0 > 50000000LL) {  // 0.5 s: never hang the GPU
                // (the builders never came: the waves of this workgroup are let through -- they compute on whatever the record arrays hold --
                // but nothing of it is stored: sAbort.  The sticky bit makes the host fail the call; the handle must be reset or restored.)
                if (lane == 0) {
                    if (a.errflag) atomicOr(a.errflag, kHoErrTimeout);
                    ldsFlagStore(&sAbort, 1);
                    ldsFlagStore(&sHave, K);
                }
                break;
            }
        }
        return;
    }
    if (wv == 5 || wv == 6) {
        // ---- fetchers
        const int f = wv - 5;
        if (f == 0) {  // the step table is builder 0's (write-through stores, in memory before its first publication)
            while (ldsFlagLoad(&sHave) < 1) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            if (lane < K) {
                const double* sp = reinterpret_cast<const double*>(a.steps + b * kBurstMax + lane);
                sRicc[lane] = __double2loint(hoLoad8(sp));
                sTtP[lane] = (T)hoLoad8(sp + 1);
            }
        }
        // where this lane's three row-constant values of a step come from: the tile's four row landmarks (clamped: rows past N are
        // computed on a copy of the last row and never stored)
        long long rowOff[3];
        int rowDst[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int e = min(lane + 64 * u, 4 * kBlkRec - 1), r = e / kBlkRec, k = e - r * kBlkRec;
            rowOff[u] = (long long)min(by * 4 + r, max(N - 1, 0)) * kBlkRec + k;
            rowDst[u] = r * (kBlkRec + 3) + k;
        }
        for (int st = f; st < K; st += 2) {
            while (ldsFlagLoad(&sHave) < st + 1) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            T x[kBlkRec + 3];
            const T* cp = colRec + (long long)(st * a.colStep);
#pragma unroll
            for (int q = 0; q < kBlkRec; ++q) x[q] = cp[(long long)(q < 27 ? q : q + 18) * cap];
            const T* rp = rowRec + (long long)(st * a.rowStep);
#pragma unroll
            for (int u = 0; u < 3; ++u) x[kBlkRec + u] = rp[rowOff[u]];
            if (st >= kSlots) {  // the slot was read by step st - 3: every arithmetic wave must be past it
                const int need = st - kSlots + 1;
                while (__ballot(lane < 4 && ldsFlagLoad(&sDone[lane & 3]) < need) != 0) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
            }
            const int sl = st % kSlots;
#pragma unroll
            for (int q = 0; q < kBlkRec; ++q) sCol[sl][q][lane] = x[q];
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (lane + 64 * u < 4 * kBlkRec) (&sRow[sl][0][0])[rowDst[u]] = x[kBlkRec + u];
            waveSync();
            if (lane == 0) ldsFlagStore(&sReady[f], st + 1);
        }
        return;
    }

    // ---- arithmetic waves
    const int I0raw = by * 4 + wv;
    const int I0 = min(I0raw, max(N - 1, 0));
    const T* Sin = static_cast<const T*>(a.Sin) + (long long)b * a.sigmaStride;
    T* Sout = static_cast<T*>(a.Sout) + (long long)b * a.sigmaStride;
    EQF_RSTAMP(63);
    T S[9];
    {
        const T* src = Sin + (long long)(kLm0 + 3 * I0) * ld + kLm0 + 3 * Jc;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) S[3 * rr + cc] = src[(long long)rr * ld + cc];
    }
    const bool diag = I0raw == J;
    for (int st = 0; st < K; ++st) {
        while (ldsFlagLoad(&sReady[st & 1]) < st + 1) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        EQF_RSTAMP(2 + 3 * st);
        if (sRicc[st]) {
            // the step of k_burst_riccati_ring<T, 1>: H = D S + Lw Sw + Lv Sv in place of S, then S' = H D_J^T (+ the process noise on
            // the diagonal) + Gn Lw_J^T + Gv Lv_J^T -- the same operations in the same order
            const int sl = st % kSlots;
            const T* rc = sRow[sl][wv];  // wave-uniform: LDS broadcast reads
            {
                T Sw[9], Sv[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    Sw[k] = sCol[sl][27 + k][lane];
                    Sv[k] = sCol[sl][36 + k][lane];
                }
                T H[9];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        T acc = rc[3 * rr] * S[cc];
#pragma unroll
                        for (int k = 1; k < 3; ++k) acc = fma(rc[3 * rr + k], S[3 * k + cc], acc);
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc = fma(rc[9 + 3 * rr + k], Sw[3 * k + cc], acc);
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc = fma(rc[18 + 3 * rr + k], Sv[3 * k + cc], acc);
                        H[3 * rr + cc] = acc;
                    }
#pragma unroll
                for (int k = 0; k < 9; ++k) S[k] = H[k];
            }
            __builtin_amdgcn_sched_barrier(0);
            const T TtP = sTtP[st];
            {
                T c[9];  // D_J^T
#pragma unroll
                for (int k = 0; k < 9; ++k) c[k] = sCol[sl][k][lane];
                T O[9];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        T acc = (diag && rr == cc) ? TtP : (T)0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc = fma(S[3 * rr + k], c[3 * cc + k], acc);
                        O[3 * rr + cc] = acc;
                    }
#pragma unroll
                for (int k = 0; k < 9; ++k) S[k] = O[k];
            }
#pragma unroll
            for (int grp = 1; grp < 3; ++grp) {  // + Gn Lw_J^T, then + Gv Lv_J^T
                __builtin_amdgcn_sched_barrier(0);
                T c[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) c[k] = sCol[sl][9 * grp + k][lane];
                const T* rg = rc + 18 + 9 * grp;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        T acc = S[3 * rr + cc];
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc = fma(rg[3 * rr + k], c[3 * cc + k], acc);
                        S[3 * rr + cc] = acc;
                    }
            }
        }
        EQF_RSTAMP(3 + 3 * st);
        waveSync();  // (every LDS read of the step has returned)
        if (lane == 0) ldsFlagStore(&sDone[wv], st + 1);
        EQF_RSTAMP(4 + 3 * st);
    }
    const bool aborted = ldsFlagLoad(&sAbort) != 0;  // (set before the sHave that let this wave's last step through)
    if (!aborted && validJ && I0raw < N && I0raw >= J) {
        T* dst = Sout + (long long)(kLm0 + 3 * I0raw) * ld + kLm0 + 3 * J;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) dst[(long long)rr * ld + cc] = S[3 * rr + cc];
    }
    // Sigma_JI = Sigma_IJ^T for the blocks strictly below the diagonal, transposed through LDS (burstRingBody).  The poller and the fetchers
    // have left by now or leave without passing another barrier: s_barrier counts the live waves.
    constexpr int kWd = 12, kPitch = kWd + 1, kRowsPass = 3 * 64;
    static_assert(kRowsPass * kPitch <= kSlots * kBlkRec * 64, "the staging tile lives in sCol");
    T* const stage = &sCol[0][0][0];
    __syncthreads();
    {
        T* dst = stage + 3 * lane * kPitch + 3 * wv;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) dst[cc * kPitch + rr] = S[3 * rr + cc];
    }
    __syncthreads();
    for (int e = tid; e < kRowsPass * kWd; e += 256) {
        const int r = e / kWd, c = e % kWd;
        const int Jm = bx * 64 + r / 3, Im = by * 4 + c / 3;
        if (!aborted && Im < N && Im > Jm) Sout[(long long)(kLm0 + 3 * Jm + r % 3) * ld + kLm0 + 3 * Im + c % 3] = stage[r * kPitch + c];
    }
}

template <typename T>
__global__ __launch_bounds__(kBuildThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_burst_fused(BurstArgs a) {
    if ((int)blockIdx.x < a.nBuild) burstBuildBody<T, true, 4, false, true>(a, (int)blockIdx.x, (int)blockIdx.y);
    else if (threadIdx.x < 448) burstRingFusedBody<T>(a, (int)blockIdx.x - a.nBuild, (int)blockIdx.y);  // (wave 7 has no role)
}

}  // namespace eqf
