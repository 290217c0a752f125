// Fused EqF propagate kernel (gfx950): per-landmark linearisation blocks + block-structured Riccati
// step + group integration in ONE launch.
//
// Replaces, on the device, VIOFilter::integrateUpToTime (eqf_vio/src/VIOFilter.cpp:146-209) together
// with EqFStateMatrixA_euclid_impl / EqFInputMatrixB_euclid_impl (src/EqFMatrices.cpp:277-317,
// :346-382), stateGroupAction (src/VIOGroup.cpp:23-69), liftVelocityDiscrete / liftVelocity + VIOExp
// (src/VIOGroup.cpp:178-255) and VIOGroup::operator* (:92-110).
//
// The reference forms F = I + T*[[0,0],[-B,A0]] densely and evaluates (F Sigma) F^T with two n^3 GEMMs.
// F is  [[F_bb, 0], [L, D]]  with F_bb 11x11, L = 3N x 11 (non-zero columns: gyro-bias 0:3 and
// velocity 8:11) and D block-diagonal 3x3, so per 3x3 landmark block
//     Sigma'_IJ = (D_I Sigma_IJ + L_I Sigma_bJ) D_J^T + (L_I Sigma_bb + D_I Sigma_Ib) L_J^T + T*Q_IJ
// which is ~10 MAC per output value: the step is bound by reading and writing Sigma once (HBM / L2),
// not by flops.  One 256-thread workgroup owns a 16x16-landmark tile (48x48 values), one thread per 3x3
// block; the 11-wide base panels of the tile are staged in LDS.  Sigma is ping-ponged (in -> out) because
// every tile reads base panels that other tiles rewrite.
#pragma once
#include "eqf_device.hpp"
#include "eqf_math.hpp"

namespace eqf {

// LDS image of the few common values the base-panel code of waves 1..3 needs
struct CommonLds {
    double T;
    double Bg[6], Bvw[9], RA[9], Avg[6];
};

// per-landmark record written by k_build_blocks (element type T): D, Lw, Lv (3x3 each, already scaled by T), then the rows
// Gn = G[:,0:3] + (sigma_w^2 / T) Lw and Gv = G[:,8:11] of G = Lw Sigma[0:3,:] + Lv Sigma[8:11,:] + D Sigma_Ib
constexpr int kBlkRec = 45;

struct PropArgs {
    const Glob* gin;
    Glob* gout;
    const double* p0;    // [B][3][cap]
    const double* Qin;   // [B][5][cap]
    double* Qout;        // [B][5][cap]
    const void* Sin;     // [B][rows][ld]  (T)
    void* Sout;
    const ImuRec* recs;  // [B] device records, or nullptr -> `inl`
    ImuRec inl;
    int* errflag;
    long long sigmaStride;  // elements between filters
    int cap, ld, NT;
    int isImu;      // processIMUData (bias subtraction, lazy init, ZOH bookkeeping)
    int doRiccati;  // VIOFilter.cpp:160
    void* blk;          // [B][cap][27] (T) per-landmark blocks D, Lw, Lv written by k_build_blocks (split path)
    CommonLds* blkCommon;  // [B] common values written by k_build_blocks
    int sigmaExternal;  // the Riccati step of this call is done by the dense MFMA backend: touch no Sigma here
    Params prm;
};

// Quantities shared by every landmark of one filter at one step.
struct StepCommon {
    int step;       // integrateUpToTime does integrate (currentTime >= 0 && dt > 0)
    double dt, T;   // this call's dt; accumulated time T
    d3 wbar;        // mean angular rate over the accumulated interval
    d3 wcur, acur;  // currentVelocity (ZOH sample used for the group step)
    m33 RA;         // R_A
    d3 vhat, etahat;
    d3 vC;          // linear part of Ad(T_IC^-1) (wbar, vhat)          (EqFMatrices.cpp:302-304)
    d3 oCcur, vCcur;  // Ad(T_IC^-1) (wcur, vhat)                       (VIOGroup.cpp:225)
    se3 camInv;     // SE3Exp(-dt * Ad(T_IC^-1)(wcur, vhat))             (VIOGroup.cpp:226)
    m33 RICt;       // R_IC^T as the reference builds it: matrix of the inverse quaternion
    m33 RIC;        // R_IC
    d3 xIC;
    double Bg[6];     // B[0:2,0:3]   (EqFMatrices.cpp:364)
    m33 Bvw;          // B[2:5,0:3] = R_A vhat^x        (:367)
    double Avg[6];    // A0[2:5,0:2] = -g * InvDiff     (:289), 3x2 row-major
};

// Functions of xi0.pose alone; cached in Glob when the pose is set.
EQF_DI void poseConstants(quat P0q, double* eta0, double* cDiff, double* cInv, int* bad) {
    const d3 e = qrot(qinv(P0q), mk3(0, 0, 1));  // VIOState.cpp:90
    eta0[0] = e.x; eta0[1] = e.y; eta0[2] = e.z;
    stereoChartDiff(e, e, cDiff, bad);
    stereoChartInvDiffAtZero(e, cInv, bad);
}

// The per-step scalar chain, in three independently computable parts so that three wavefronts (three SIMDs) can
// run them side by side: a lone wavefront retires one fp64 operation per ~6 cycles whatever the dependencies.
//   kPartBase  dt, T, ZOH sample, mean rate, R_A, vhat, etahat      (everyone)
//   kPartRicc  v_C, camera-offset matrices, B / A0 base blocks     (wave 0: linearisation blocks, F_bb)
//   kPartLift  Ad(T_IC^-1)(w_cur, vhat), SE3Exp(-dt U_C)            (the landmark group step)
constexpr int kPartBase = 1, kPartRicc = 2, kPartLift = 4;
// G is the (read-only) current state; for a vision call only r.stamp is used.
// (A: anything with the members prm, isImu, doRiccati -- PropArgs, or the per-step view of the burst kernels)
template <class Args>
EQF_DI void stepCommon(const Glob& G, const ImuRec& r, const Args& a, StepCommon& c, int parts, int* bad) {
    const Params& p = a.prm;
    c.dt = r.stamp - G.curTime;
    c.step = (G.curTime >= 0) && (c.dt > 0);  // VIOFilter.cpp:147-152
    if (!c.step) return;
    c.T = G.accTime + c.dt;  // :154
    c.wcur = mk3(G.curVel[0], G.curVel[1], G.curVel[2]);
    c.acur = mk3(G.curVel[3], G.curVel[4], G.curVel[5]);
    // :155 accumulatedVelocity += currentVelocity * dt ; :169 mean = accumulatedVelocity * (1/T)
    const double invT = 1.0 / c.T;
    c.wbar = mk3((G.accVel[0] + c.wcur.x * c.dt) * invT, (G.accVel[1] + c.wcur.y * c.dt) * invT,
        (G.accVel[2] + c.wcur.z * c.dt) * invT);
    c.RA = q2m(quat{G.Aq[0], G.Aq[1], G.Aq[2], G.Aq[3]});
    // X.A.R().inverse() * v (VIOGroup.cpp:26,49): R_A^T v with the matrix already at hand
    c.vhat = mtv33(c.RA, mk3(G.v0[0] - G.w[0], G.v0[1] - G.w[1], G.v0[2] - G.w[2]));
    // constants of the origin pose: cached at initialisation; recomputed only in the (reset) corner case where a
    // step happens in the very call that initialises the pose
    double eta0[3], cDiff[6], cInv[6];
    if (G.initialised) {
#pragma unroll
        for (int i = 0; i < 3; ++i) eta0[i] = G.eta0[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            cDiff[i] = G.cDiff[i];
            cInv[i] = G.cInv[i];
        }
    } else {
        quat P0q = quat{G.P0q[0], G.P0q[1], G.P0q[2], G.P0q[3]};
        if (a.isImu)
            P0q = so3FromVectors(mk3(r.a[0] - G.bias[3], r.a[1] - G.bias[4], r.a[2] - G.bias[5]), mk3(0, 0, 1), bad);
        poseConstants(P0q, eta0, cDiff, cInv, bad);
    }
    c.etahat = mtv33(c.RA, mk3(eta0[0], eta0[1], eta0[2]));
    if (parts & (kPartRicc | kPartLift)) {
        // Ad(T_IC^-1) (w, v) = (R w ; x^ R w + R v) with the host-precomputed inverse camera offset
        m33 RcI;
#pragma unroll
        for (int i = 0; i < 9; ++i) RcI.a[i] = p.RcamI[i];
        const d3 xcI = mk3(p.camIx[0], p.camIx[1], p.camIx[2]);
        const d3 Rv = mv33(RcI, c.vhat);
        if (parts & kPartRicc) {
            const d3 Rw = mv33(RcI, c.wbar);
            c.vC = add(crs(xcI, Rw), Rv);
        }
        if (parts & kPartLift) {
            const d3 Rw = mv33(RcI, c.wcur);
            c.oCcur = Rw;
            c.vCcur = add(crs(xcI, Rw), Rv);
            if (p.useDiscreteVelocityLift) c.camInv = se3Exp(scl(-c.dt, c.oCcur), scl(-c.dt, c.vCcur));
        }
    }
    if (parts & kPartRicc) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            c.RIC.a[i] = p.RIC[i];
            c.RICt.a[i] = p.RICt[i];
        }
        c.xIC = mk3(p.camx[0], p.camx[1], p.camx[2]);
        if (a.doRiccati) {
            const m33 RAg = mul33(c.RA, skew3(c.etahat));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    c.Bg[3 * i + j] = cDiff[3 * i] * RAg.a[j] + cDiff[3 * i + 1] * RAg.a[3 + j] + cDiff[3 * i + 2] * RAg.a[6 + j];
            c.Bvw = mul33(c.RA, skew3(c.vhat));
#pragma unroll
            for (int i = 0; i < 6; ++i) c.Avg[i] = -kGravity * cInv[i];
        }
    }
}

// Linearisation blocks of one landmark, already scaled by T:
//   D = I + T*A_q (EqFMatrices.cpp:308-311), Lw = -T*B_i (:376), Lv = T*A_v (:296)
struct LmBlocks {
    m33 D, Lw, Lv;
};
// Lw = -T * B_i needs nothing of the state but T: B_i = Qhat (q^x R_IC^T + R_IC^T x_IC^x)   (EqFMatrices.cpp:371-376)
EQF_DI m33 buildLw(double T, const m33& RICt, d3 xIC, quat Qq, double Qa, d3 p0) {
    const d3 qhat = scl(1.0 / Qa, qrot(qinv(Qq), p0));  // Q^-1 p0, VIOGroup.cpp:63 / SOT3.cpp:121
    const m33 Qhat = scl33(Qa, q2m(Qq));
    const m33 Bi = mul33(Qhat, add33(mul33(skew3(qhat), RICt), mul33(RICt, skew3(xIC))));
    return scl33(-T, Bi);
}
// D = I + T*A_q and Lv = T*A_v need the common linearisation values
EQF_DI void buildDLv(const StepCommon& c, quat Qq, double Qa, d3 p0, m33* D, m33* Lv) {
    const m33 RQ = q2m(Qq);
    const d3 qhat = scl(1.0 / Qa, qrot(qinv(Qq), p0));
    const m33 Qhat = scl33(Qa, RQ);
    *Lv = scl33(-c.T, mul33(Qhat, mulT33(tr33(c.RIC), c.RA)));  // T * (-Qhat R_IC^T R_A^T)
    // A_q = -Qhat (q^x v^x - 2 v q^T + q v^T) Qhat^-1 / |q|^2 ; the scale a cancels, Qhat^-1 = R_Q^T / a
    const m33 inner = add33(mul33(skew3(qhat), skew3(c.vC)), add33(scl33(-2.0, outer3(c.vC, qhat)), outer3(qhat, c.vC)));
    const m33 Aq = scl33(-1.0 / dot3(qhat, qhat), mul33(RQ, mulT33(inner, RQ)));
    *D = add33(eye3(), scl33(c.T, Aq));
}
EQF_DI LmBlocks buildBlocks(const StepCommon& c, quat Qq, double Qa, d3 p0) {
    LmBlocks b;
    buildDLv(c, Qq, Qa, p0, &b.D, &b.Lv);
    b.Lw = buildLw(c.T, c.RICt, c.xIC, Qq, Qa, p0);
    return b;
}

// Group step of one landmark: Q_i <- Q_i * lift_i   (VIOGroup.cpp:230-240 / :188-196 + SOT3Exp; :105-107)
template <class Args>
EQF_DI void stepLandmark(const StepCommon& c, const Args& a, quat Qq, double Qa, d3 p0, quat* Qo, double* ao, int* bad) {
    const d3 qhat = scl(1.0 / Qa, qrot(qinv(Qq), p0));
    quat lq;
    double la;
    if (a.prm.useDiscreteVelocityLift) {
        const d3 q1 = se3app(c.camInv, qhat);
        lq = so3FromVectors(q1, qhat, bad);
        la = nrm3(qhat) / nrm3(q1);
    } else {
        // W_i = (omega_C + q^x v_C / |q|^2, q.v_C / |q|^2) with U_C from the CURRENT sample; VIOExp(dt * W)
        const double n2 = dot3(qhat, qhat);
        const d3 Wr = add(c.oCcur, scl(1.0 / n2, crs(qhat, c.vCcur)));
        lq = so3Exp(scl(c.dt, Wr));
        la = exp(c.dt * dot3(qhat, c.vCcur) / n2);
    }
    *Qo = qmul(Qq, lq);
    *ao = Qa * la;
}

// Scalar part of the step, one lane per filter: lazy initialisation, X.A, X.w, ZOH bookkeeping.
// `out` already holds a copy of G (made word-parallel by the calling wave); only changed fields are written, so no
// private Glob copy (which hipcc would place in scratch) is needed.
template <class Args>
EQF_DI void stepGlobal(const Glob& G, Glob* out, const ImuRec& r, const Args& a, const StepCommon& c, int* bad) {
    d3 unbW = mk3(0, 0, 0), unbA = mk3(0, 0, 0);
    if (a.isImu) {  // VIOFilter.cpp:121-124
        unbW = mk3(r.w[0] - G.bias[0], r.w[1] - G.bias[1], r.w[2] - G.bias[2]);
        unbA = mk3(r.a[0] - G.bias[3], r.a[1] - G.bias[4], r.a[2] - G.bias[5]);
        if (!G.initialised) {  // initialiseFromIMUData, VIOFilter.cpp:133-144
            const quat q0 = so3FromVectors(unbA, mk3(0, 0, 1), bad);
            out->P0q[0] = q0.w; out->P0q[1] = q0.x; out->P0q[2] = q0.y; out->P0q[3] = q0.z;
            out->P0x[0] = out->P0x[1] = out->P0x[2] = 0;
            out->v0[0] = out->v0[1] = out->v0[2] = 0;
            out->initialised = 1;
            // NB: a level start (accel along +z) makes the gravity chart singular; the reference then throws from
            // the first Riccati step (SO3.cpp:160).  The flag raised here is sticky (eqf_device_error).
            double e0[3], cd[6], ci[6];
            poseConstants(q0, e0, cd, ci, bad);
#pragma unroll
            for (int i = 0; i < 3; ++i) out->eta0[i] = e0[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                out->cDiff[i] = cd[i];
                out->cInv[i] = ci[i];
            }
        }
    }
    if (c.step) {
        // VIOFilter.cpp:154-155, :192-193
        out->accTime = a.doRiccati ? 0.0 : c.T;
#pragma unroll
        for (int i = 0; i < 6; ++i) out->accVel[i] = a.doRiccati ? 0.0 : G.accVel[i] + G.curVel[i] * c.dt;
        const se3 lA = se3Exp(scl(c.dt, c.wcur), scl(c.dt, c.vhat));  // VIOGroup.cpp:214-217 / :182-185 + :247
        d3 lw;
        if (a.prm.useDiscreteVelocityLift) {  // VIOGroup.cpp:219-222
            const d3 inner = add(c.vhat, scl(c.dt, add(add(neg(crs(c.wcur, c.vhat)), c.acur), scl(-kGravity, c.etahat))));
            lw = sub(c.vhat, qrot(lA.q, inner));
        } else {  // VIOGroup.cpp:187, :248
            lw = scl(c.dt, add(neg(c.acur), scl(kGravity, c.etahat)));
        }
        const se3 A = se3{quat{G.Aq[0], G.Aq[1], G.Aq[2], G.Aq[3]}, mk3(G.Ax[0], G.Ax[1], G.Ax[2])};
        const se3 An = se3mul(A, lA);                                      // VIOGroup.cpp:95
        const d3 wn = add(mk3(G.w[0], G.w[1], G.w[2]), qrot(A.q, lw));     // :96
        out->Aq[0] = An.q.w; out->Aq[1] = An.q.x; out->Aq[2] = An.q.y; out->Aq[3] = An.q.z;
        out->Ax[0] = An.x.x; out->Ax[1] = An.x.y; out->Ax[2] = An.x.z;
        out->w[0] = wn.x; out->w[1] = wn.y; out->w[2] = wn.z;
        out->curTime = r.stamp;  // :207
    }
    if (a.isImu) {  // VIOFilter.cpp:129-130
        out->curVel[0] = unbW.x; out->curVel[1] = unbW.y; out->curVel[2] = unbW.z;
        out->curVel[3] = unbA.x; out->curVel[4] = unbA.y; out->curVel[5] = unbA.z;
        out->curTime = r.stamp;
    } else {
        out->updateOk = (c.step && G.initialised) ? 1 : 0;  // VIOFilter.cpp:234-236
    }
}

// PRE = false: fused kernel (single small filter: one launch per step, scalar chain on waves 0..2).
// PRE = true : streaming kernel of the split path; the blocks, the group step and the scalar state were produced by
//              k_build_blocks, so this instantiation carries no fp64 scalar chain (few registers, high occupancy).
#ifdef EQF_PROP_STAMPS
__device__ long long g_propStamps[4][8];
#define EQF_PSTAMP(i) do { if ((tid & 63) == 0 && (blockIdx.x == a.NT * a.NT || blockIdx.x == a.NT + 1)) g_propStamps[(blockIdx.x == a.NT + 1 ? 2 : 0) + ((tid >> 6) == 2 ? 1 : 0)][i] = __builtin_readcyclecounter(); } while (0)
#else
#define EQF_PSTAMP(i) do { } while (0)
#endif
template <typename T, bool PRE>
__global__ __launch_bounds__(256) void k_propagate(PropArgs a) {
    const int tid = threadIdx.x;
    EQF_PSTAMP(0);
    const int b = blockIdx.y;
    const bool isExtra = (int)blockIdx.x == a.NT * a.NT;  // the base-block workgroup
    // Workgroups after the tiles carry no tile (index relative to NT^2):
    //   0              the 11 x 11 base block
    //   1              the scalar state (X.A, X.w, ZOH bookkeeping)                      [fused kernel only]
    //   2 .. 2+NT-1    "row tails":    Sigma'_Ib of one landmark group (they run that group's linearisation chain, no tile math)
    //   2+NT .. 2+2NT-1 "column tails": Sigma'_bJ of one landmark group
    //   2+2NT ..       the group step Q_i <- Q_i lift_i of 64 landmarks each              [fused kernel only]
    // Measured per-workgroup durations (N = 200, cycles): tiles 15.5 k; with the tails inside the tiles of the last tile
    // row / column that corner tile took 21.4 k, one workgroup doing both tails of a group 19.6 k, base block + scalar
    // state in one workgroup 18.4 k, the landmark group step inside the diagonal tiles held their barrier up by 3 k.
    const int rel = (int)blockIdx.x - a.NT * a.NT;
    const bool isState = rel == 1;
    const bool isRowTail = rel >= 2 && rel < 2 + a.NT;
    const bool isColTail = rel >= 2 + a.NT && rel < 2 + 2 * a.NT;
    const bool isTail = isRowTail || isColTail;
    const int tailG = isRowTail ? rel - 2 : rel - 2 - a.NT;
    const bool isLmWg = !PRE && rel >= 2 + 2 * a.NT;
    const int ti = isTail ? tailG : (rel >= 0 ? 0 : blockIdx.x / a.NT);
    const int tj = isTail ? tailG : (rel >= 0 ? 0 : blockIdx.x % a.NT);
    const int cap = a.cap, ld = a.ld;

    __shared__ T sD[32][9], sLw[32][9], sLv[32][9];  // [0,16): row landmarks I, [16,32): column landmarks J
    __shared__ T sG[16][33];                          // G_I = L_I Sigma_bb + D_I Sigma_Ib   (3 x 11)
    __shared__ T sGn[16][9];                          // G_I[:,0:3] + (sigma_w^2 / T) Lw_I  (process noise folded in)
    __shared__ T sSbJ[11][kTile];                     // Sigma[base rows, tile columns]
    __shared__ T sSIb[kTile][12];                     // Sigma[tile rows, base columns]
    __shared__ T sSbb[11][12];
    __shared__ T sF[11][12];                          // F_bb
    __shared__ T sNb[11][3];                          // rows of B (gyro columns) for the base coordinates
    __shared__ T sTmp[16][33];
    __shared__ T sTb[11][12];
    __shared__ CommonLds sC;

    const Glob& G = a.gin[b];
    const ImuRec& r = a.recs ? a.recs[b] : a.inl;
    const int N = G.N;
    // cheap, every thread: does this call integrate, and does it touch Sigma?
    const double dt0 = r.stamp - G.curTime;
    const bool step = (G.curTime >= 0) && (dt0 > 0);
    const bool riccati = step && a.doRiccati && !a.sigmaExternal;

    const T* Sin = static_cast<const T*>(a.Sin) + (long long)b * a.sigmaStride;
    T* Sout = static_cast<T*>(a.Sout) + (long long)b * a.sigmaStride;
    const double* p0 = a.p0 + (long long)b * 3 * cap;
    const double* Qin = a.Qin + (long long)b * 5 * cap;
    double* Qout = a.Qout + (long long)b * 5 * cap;
    const int I0 = ti * kTileLm, J0 = tj * kTileLm;
    int bad = 0;

    const int wv = tid >> 6, ln = tid & 63;
    // ---- waves 1..3 stage the base panels, every thread fetches its own 3x3 block (called after the scalar chains: issuing
    // these loads before the chains, with the landmark count passed by kernel argument, measured slower)
    const int bi = tid >> 4, bj = tid & 15;
    const int BI = I0 + bi, BJ = J0 + bj;
    const bool blockValid = !isExtra && !isTail && rel < 0 && BI < N && BJ < N;
    T S[9];
    auto stageAll = [&]() {
        if (tid >= 64) {
            // one flat index space over the three panels; all loads are issued before the first LDS store (one cold-miss
            // latency for the lot instead of one per trip)
            constexpr int nA = 11 * kTile, nB = kTile * 12, nC = 132, nAll = nA + nB + nC, kTrips = (nAll + 191) / 192;
            T buf[kTrips];
#pragma unroll
            for (int u = 0; u < kTrips; ++u) {
                const int e = tid - 64 + 192 * u;
                T v = (T)0;
                if (e < nA) {
                    const int rr = e / kTile, cc = e % kTile;
                    const int Cc = kLm0 + 3 * J0 + cc;
                    if (Cc < kLm0 + 3 * N) v = Sin[(long long)rr * ld + Cc];
                } else if (e < nA + nB) {
                    const int f = e - nA, rr = f / 12, cc = f % 12;
                    const int R = kLm0 + 3 * I0 + rr;
                    if (R < kLm0 + 3 * N && cc < 11) v = Sin[(long long)R * ld + cc];
                } else if (e < nAll) {
                    const int f = e - nA - nB, rr = f / 12, cc = f % 12;
                    if (cc < 11) v = Sin[(long long)rr * ld + cc];
                }
                buf[u] = v;
            }
#pragma unroll
            for (int u = 0; u < kTrips; ++u) {
                const int e = tid - 64 + 192 * u;
                if (e < nA) sSbJ[e / kTile][e % kTile] = buf[u];
                else if (e < nA + nB) sSIb[(e - nA) / 12][(e - nA) % 12] = buf[u];
                else if (e < nAll) sSbb[(e - nA - nB) / 12][(e - nA - nB) % 12] = buf[u];
            }
        }
        if (blockValid) {
            const T* src = Sin + (long long)(kLm0 + 3 * BI) * ld + kLm0 + 3 * BJ;
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) S[3 * rr + cc] = src[(long long)rr * ld + cc];
        }
    };
    if (isLmWg) {
        // ---- group step Q_i <- Q_i * lift_i of 64 landmarks (VIOGroup.cpp:230-240 / :188-196, :105-107)
        const int i = (rel - 2 - 2 * a.NT) * 64 + ln;
        if (wv == 0 && i < N) {
            const quat Qq = quat{Qin[i], Qin[cap + i], Qin[2 * cap + i], Qin[3 * cap + i]};
            const double Qa = Qin[4 * cap + i];
            quat Qo = Qq;
            double ao = Qa;
            if (step) {
                StepCommon c;
                stepCommon(G, r, a, c, kPartBase | kPartLift, &bad);
                stepLandmark(c, a, Qq, Qa, mk3(p0[i], p0[cap + i], p0[2 * cap + i]), &Qo, &ao, &bad);
            }
            Qout[i] = Qo.w; Qout[cap + i] = Qo.x; Qout[2 * cap + i] = Qo.y; Qout[3 * cap + i] = Qo.z;
            Qout[4 * cap + i] = ao;
            if (bad && a.errflag) atomicOr(a.errflag, 1);
        }
        return;
    }
    if (!PRE && wv == 1 && riccati && ln < 32 && !isExtra && !isState) {
        // ---- wave 1: the Lw blocks of the tile's landmarks -- they need nothing of the linearisation but T, so they are
        // built beside wave 0's chain instead of at its end
        const int i = (ln < 16) ? I0 + ln : J0 + ln - 16;
        m33 Lw;
#pragma unroll
        for (int k = 0; k < 9; ++k) Lw.a[k] = 0.0;
        if (i < N) {
            m33 RICt;
#pragma unroll
            for (int k = 0; k < 9; ++k) RICt.a[k] = a.prm.RICt[k];
            const quat Qq = quat{Qin[i], Qin[cap + i], Qin[2 * cap + i], Qin[3 * cap + i]};
            Lw = buildLw(G.accTime + dt0, RICt, mk3(a.prm.camx[0], a.prm.camx[1], a.prm.camx[2]), Qq, Qin[4 * cap + i],
                mk3(p0[i], p0[cap + i], p0[2 * cap + i]));
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) sLw[ln][k] = (T)Lw.a[k];
    }
    if (isState) {
        // ---- scalar state.  Word-parallel copy in -> out, then one lane patches the changed fields
        if (!PRE && wv == 0) {
            static_assert(sizeof(Glob) % 8 == 0 && sizeof(Glob) / 8 <= 64, "Glob copy is one word per lane");
            const double* src = reinterpret_cast<const double*>(&G);
            double* dst = reinterpret_cast<double*>(a.gout + b);
            if (ln < (int)(sizeof(Glob) / 8)) dst[ln] = src[ln];
            if (ln == 0) {
                StepCommon c;
                c.step = 0;
                if (step) stepCommon(G, r, a, c, kPartBase, &bad);
                stepGlobal(G, a.gout + b, r, a, c, &bad);
                if (bad && a.errflag) atomicOr(a.errflag, 1);
            }
        }
        return;
    }
    if (PRE) {
        if (riccati && tid < 32 && !isExtra) {
            const int i = (tid < 16) ? I0 + tid : J0 + tid - 16;
            const T* bp = static_cast<const T*>(a.blk) + ((long long)b * cap + i) * kBlkRec;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                sD[tid][k] = (i < N) ? bp[k] : (T)0;
                sLw[tid][k] = (i < N) ? bp[9 + k] : (T)0;
                sLv[tid][k] = (i < N) ? bp[18 + k] : (T)0;
            }
        }
        if (riccati && tid == 32) sC = a.blkCommon[b];
    } else {
    if (wv == 0 && riccati) {
        // ---- wave 0: common quantities of the linearisation + this tile's per-landmark blocks
        StepCommon c;
        stepCommon(G, r, a, c, kPartBase | kPartRicc, &bad);
        if (ln < 32 && !isExtra) {
            const int i = (ln < 16) ? I0 + ln : J0 + ln - 16;
            if (i < N) {
                const quat Qq = quat{Qin[i], Qin[cap + i], Qin[2 * cap + i], Qin[3 * cap + i]};
                m33 D, Lv;
                buildDLv(c, Qq, Qin[4 * cap + i], mk3(p0[i], p0[cap + i], p0[2 * cap + i]), &D, &Lv);
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    sD[ln][k] = (T)D.a[k];
                    sLv[ln][k] = (T)Lv.a[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k) sD[ln][k] = sLv[ln][k] = (T)0;
            }
        }
        if (ln == 32) {
            sC.T = c.T;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                sC.Bg[k] = c.Bg[k];
                sC.Avg[k] = c.Avg[k];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                sC.Bvw[k] = c.Bvw.a[k];
                sC.RA[k] = c.RA.a[k];
            }
        }
    }
    }
    if (!riccati && a.sigmaExternal && step && a.doRiccati) {
        if (bad && a.errflag) atomicOr(a.errflag, 1);
        return;  // Sigma_out was written by k_dense_gemm
    }
    if (!riccati) {
        // Sigma is not touched by this call: copy the tile through so that the ping-pong parity of all
        // filters of the batch stays in step.
        for (int e = tid; e < kTile * kTile && !isExtra && !isTail; e += 256) {
            const int rr = e / kTile, cc = e % kTile;
            const int R = kLm0 + 3 * I0 + rr, Cc = kLm0 + 3 * J0 + cc;
            if (R < kLm0 + 3 * N && Cc < kLm0 + 3 * N) Sout[(long long)R * ld + Cc] = Sin[(long long)R * ld + Cc];
        }
        if (isRowTail)
            for (int e = tid; e < kTile * 12; e += 256) {
                const int R = kLm0 + 3 * I0 + e / 12, Cc = e % 12;
                if (R < kLm0 + 3 * N) Sout[(long long)R * ld + Cc] = Sin[(long long)R * ld + Cc];
            }
        if (isColTail)
            for (int e = tid; e < 12 * kTile; e += 256) {
                const int R = e / kTile, Cc = kLm0 + 3 * J0 + e % kTile;
                if (Cc < kLm0 + 3 * N) Sout[(long long)R * ld + Cc] = Sin[(long long)R * ld + Cc];
            }
        if (isExtra && tid < 144) Sout[(long long)(tid / 12) * ld + tid % 12] = Sin[(long long)(tid / 12) * ld + tid % 12];
        if (bad && a.errflag) atomicOr(a.errflag, 1);
        return;
    }

    stageAll();
    EQF_PSTAMP(1);
    __syncthreads();
    EQF_PSTAMP(2);
    if (tid < 132) {
        const int rr = tid / 12, cc = tid % 12;
        // F_bb = I + T * [[0,0,0,0],[-B_g^w,0,0,0],[-B_v^w,-R_A, A_vg, 0]]   (VIOFilter.cpp:178-183)
        double f = (rr == cc) ? 1.0 : 0.0;
        if (rr >= 6 && rr < 8 && cc < 3) f = -sC.T * sC.Bg[3 * (rr - 6) + cc];
        if (rr >= 8) {
            if (cc < 3) f = -sC.T * sC.Bvw[3 * (rr - 8) + cc];
            else if (cc < 6) f = -sC.T * sC.RA[3 * (rr - 8) + cc - 3];
            else if (cc < 8) f = sC.T * sC.Avg[2 * (rr - 8) + cc - 6];
        }
        sF[rr][cc] = (cc < 11) ? (T)f : (T)0;
        if (cc < 3) {
            double nb = 0.0;
            if (rr >= 6 && rr < 8) nb = sC.Bg[3 * (rr - 6) + cc];
            if (rr >= 8) nb = sC.Bvw[3 * (rr - 8) + cc];
            sNb[rr][cc] = (T)nb;
        }
    }

    const T sw2 = (T)a.prm.velOmegaVariance, sa2 = (T)a.prm.velAccelVariance;
    const T Tt = (T)sC.T;

    if (!isExtra) {
    // ---- G_I = Lw_I Sigma[0:3, 0:11] + Lv_I Sigma[8:11, 0:11] + D_I Sigma_Ib ; Gn_I = G_I[:,0:3] + (sigma_w^2 / T) Lw_I
    for (int e = tid; e < 16 * 33; e += 256) {
        const int i = e / 33, rc = e % 33, rr = rc / 11, cc = rc % 11;
        T acc = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            acc += sLw[i][3 * rr + k] * sSbb[k][cc] + sLv[i][3 * rr + k] * sSbb[8 + k][cc] + sD[i][3 * rr + k] * sSIb[3 * i + k][cc];
        sG[i][rc] = acc;
        if (cc < 3) sGn[i][3 * rr + cc] = acc + (sw2 / Tt) * sLw[i][3 * rr + cc];
    }
    // ---- one 3x3 block per thread; the first product needs nothing from G_I and runs before the barrier
    T H[9];
    const int i = bi, j = bj;
    if (blockValid) {
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                T acc = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    acc += sD[i][3 * rr + k] * S[3 * k + cc] + sLw[i][3 * rr + k] * sSbJ[k][3 * j + cc] +
                           sLv[i][3 * rr + k] * sSbJ[8 + k][3 * j + cc];
                H[3 * rr + cc] = acc;
            }
    }
    __syncthreads();
    if (blockValid) {
        T* dst = Sout + (long long)(kLm0 + 3 * BI) * ld + kLm0 + 3 * BJ;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                T acc = (BI == BJ && rr == cc) ? Tt * (T)a.prm.pointProcessVariance : (T)0;  // T * P
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    acc += H[3 * rr + k] * sD[16 + j][3 * cc + k] + sGn[i][3 * rr + k] * sLw[16 + j][3 * cc + k] +
                           sG[i][rr * 11 + 8 + k] * sLv[16 + j][3 * cc + k];
                dst[(long long)rr * ld + cc] = acc;
            }
    }

    EQF_PSTAMP(3);
    // ---- Sigma'_Ib = G_I F_bb^T + T (B R B^T)_Ib      (the row-tail workgroup of landmark group I)
    if (isRowTail) {
        for (int e = tid; e < 16 * 33; e += 256) {
            const int i = e / 33, rc = e % 33, rr = rc / 11, cc = rc % 11;
            const int I = I0 + i;
            if (I >= N) continue;
            T acc = 0;
#pragma unroll
            for (int k = 0; k < 11; ++k) acc += sG[i][rr * 11 + k] * sF[cc][k];
            // T*sw2*B_I^w (B_c^w)^T with Lw = -T B^w
#pragma unroll
            for (int k = 0; k < 3; ++k) acc -= sw2 * sLw[i][3 * rr + k] * sNb[cc][k];
            Sout[(long long)(kLm0 + 3 * I + rr) * ld + cc] = acc;
        }
        if (tid < kTile) {
            const int R = kLm0 + 3 * I0 + tid;
            if (R < kLm0 + 3 * N) Sout[(long long)R * ld + 11] = (T)0;
        }
    }
    // ---- Sigma'_bJ = F_bb (Sigma_bb L_J^T + Sigma_bJ D_J^T) + T (B R B^T)_bJ   (the column-tail workgroup of landmark group J)
    if (isColTail) {
        for (int e = tid; e < 16 * 33; e += 256) {
            const int j = e / 33, rc = e % 33, cc = rc / 3, rr = rc % 3;  // G~_J[cc][rr], cc base row, rr landmark comp
            T acc = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                acc += sSbb[cc][k] * sLw[16 + j][3 * rr + k] + sSbb[cc][8 + k] * sLv[16 + j][3 * rr + k] +
                       sSbJ[cc][3 * j + k] * sD[16 + j][3 * rr + k];
            sTmp[j][rc] = acc;
        }
        __syncthreads();
        for (int e = tid; e < 16 * 33; e += 256) {
            const int j = e / 33, rc = e % 33, cc = rc / 3, rr = rc % 3;
            const int J = J0 + j;
            if (J >= N) continue;
            T acc = 0;
#pragma unroll
            for (int k = 0; k < 11; ++k) acc += sF[cc][k] * sTmp[j][3 * k + rr];
#pragma unroll
            for (int k = 0; k < 3; ++k) acc -= sw2 * sNb[cc][k] * sLw[16 + j][3 * rr + k];
            Sout[(long long)cc * ld + kLm0 + 3 * J + rr] = acc;
        }
        if (tid < kTile) {
            const int Cc = kLm0 + 3 * J0 + tid;
            if (Cc < kLm0 + 3 * N) Sout[(long long)11 * ld + Cc] = (T)0;
        }
    }
    }  // !isExtra
    // ---- Sigma'_bb = F_bb Sigma_bb F_bb^T + T (P_bb + B_b R B_b^T)     (the extra workgroup)
    if (isExtra) {
        __syncthreads();
        if (tid < 121) {
            const int rr = tid / 11, cc = tid % 11;
            T acc = 0;
#pragma unroll
            for (int k = 0; k < 11; ++k) acc += sF[rr][k] * sSbb[k][cc];
            sTb[rr][cc] = acc;
        }
        __syncthreads();
        if (tid < 144) {
            const int rr = tid / 12, cc = tid % 12;
            T acc = 0;
            if (rr < 11 && cc < 11) {
#pragma unroll
                for (int k = 0; k < 11; ++k) acc += sTb[rr][k] * sF[cc][k];
                T nz = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) nz += sw2 * sNb[rr][k] * sNb[cc][k];
                if (rr >= 8 && cc >= 8) {  // accel columns of B: rows 8:11 hold R_A
#pragma unroll
                    for (int k = 0; k < 3; ++k) nz += sa2 * (T)sC.RA[3 * (rr - 8) + k] * (T)sC.RA[3 * (cc - 8) + k];
                }
                if (rr == cc) {
                    const Params& p = a.prm;
                    nz += (T)(rr < 3 ? p.biasOmegaProcessVariance
                                     : (rr < 6 ? p.biasAccelProcessVariance : (rr < 8 ? p.gravityProcessVariance : p.velocityProcessVariance)));
                }
                acc += Tt * nz;
            }
            Sout[(long long)rr * ld + cc] = acc;  // row/col 11 stay zero
        }
    }
    EQF_PSTAMP(4);
    if (bad && a.errflag) atomicOr(a.errflag, 1);
}

// Builder of the split path.  grid = (ceil(N / 64) + 1, B), block = 128.  As in the fused kernel the fp64 scalar chain is
// spread over wavefronts (a lone wave is issue-bound, so three chains side by side cost the time of the longest one):
//   landmark workgroups  wave 0: common linearisation values + the blocks D, Lw, Lv and the G rows of 64 landmarks
//                        wave 1: the group step Q_i <- Q_i * lift_i of the same landmarks
//   last workgroup       wave 0: the scalar state (X.A, X.w, ZOH bookkeeping);  wave 1: the common values for the base panels
template <typename T>
__global__ __launch_bounds__(128) void k_build_blocks(PropArgs a) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool isState = blockIdx.x == gridDim.x - 1;
    const int i = blockIdx.x * 64 + lane;
    const int cap = a.cap;
    const Glob& G = a.gin[b];
    const ImuRec& r = a.recs ? a.recs[b] : a.inl;
    const int N = G.N;
    const double dt0 = r.stamp - G.curTime;
    const bool step = (G.curTime >= 0) && (dt0 > 0);
    const bool riccati = step && a.doRiccati && !a.sigmaExternal;
    const double* p0 = a.p0 + (long long)b * 3 * cap;
    const double* Qin = a.Qin + (long long)b * 5 * cap;
    double* Qout = a.Qout + (long long)b * 5 * cap;
    int bad = 0;
    if (isState) {
        if (wv == 0) {
            const double* src = reinterpret_cast<const double*>(&G);
            double* dst = reinterpret_cast<double*>(a.gout + b);
            if (lane < (int)(sizeof(Glob) / 8)) dst[lane] = src[lane];
            if (lane == 0) {
                StepCommon c;
                c.step = 0;
                if (step) stepCommon(G, r, a, c, kPartBase, &bad);
                stepGlobal(G, a.gout + b, r, a, c, &bad);
            }
        } else if (lane == 0 && riccati) {
            StepCommon c;
            stepCommon(G, r, a, c, kPartBase | kPartRicc, &bad);
            CommonLds cl;
            cl.T = c.T;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                cl.Bg[k] = c.Bg[k];
                cl.Avg[k] = c.Avg[k];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                cl.Bvw[k] = c.Bvw.a[k];
                cl.RA[k] = c.RA.a[k];
            }
            a.blkCommon[b] = cl;
        }
        if (bad && a.errflag) atomicOr(a.errflag, 1);
        return;
    }
    if (i >= N) return;
    const quat Qq = quat{Qin[i], Qin[cap + i], Qin[2 * cap + i], Qin[3 * cap + i]};
    const double Qa = Qin[4 * cap + i];
    const d3 q0 = mk3(p0[i], p0[cap + i], p0[2 * cap + i]);
    if (wv == 0) {
        if (riccati) {
            StepCommon c;
            stepCommon(G, r, a, c, kPartBase | kPartRicc, &bad);
            const LmBlocks blk = buildBlocks(c, Qq, Qa, q0);
            T* bp = static_cast<T*>(a.blk) + ((long long)b * cap + i) * kBlkRec;
            T D[9], Lw[9], Lv[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                D[k] = (T)blk.D.a[k];
                Lw[k] = (T)blk.Lw.a[k];
                Lv[k] = (T)blk.Lv.a[k];
                bp[k] = D[k];
                bp[9 + k] = Lw[k];
                bp[18 + k] = Lv[k];
            }
            // the two 3x3 pieces of G_I = Lw Sigma[0:3, :] + Lv Sigma[8:11, :] + D Sigma_Ib that the landmark blocks need
            // (same expression order as k_propagate's tile code)
            const T* Sin = static_cast<const T*>(a.Sin) + (long long)b * a.sigmaStride;
            const T sw2 = (T)a.prm.velOmegaVariance, Tt = (T)c.T;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c0 = half ? 8 : 0;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        T acc = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            acc += Lw[3 * rr + k] * Sin[(long long)k * a.ld + c0 + cc] + Lv[3 * rr + k] * Sin[(long long)(8 + k) * a.ld + c0 + cc] +
                                   D[3 * rr + k] * Sin[(long long)(kLm0 + 3 * i + k) * a.ld + c0 + cc];
                        bp[27 + 9 * half + 3 * rr + cc] = half ? acc : acc + (sw2 / Tt) * Lw[3 * rr + cc];
                    }
            }
        }
    } else {
        quat Qo = Qq;
        double ao = Qa;
        if (step) {
            StepCommon c;
            stepCommon(G, r, a, c, kPartBase | kPartLift, &bad);
            stepLandmark(c, a, Qq, Qa, q0, &Qo, &ao, &bad);
        }
        Qout[i] = Qo.w; Qout[cap + i] = Qo.x; Qout[2 * cap + i] = Qo.y; Qout[3 * cap + i] = Qo.z;
        Qout[4 * cap + i] = ao;
    }
    if (bad && a.errflag) atomicOr(a.errflag, 1);
}

// ------------------------------------------------------------------------------------------------
// k_riccati_stream: the landmark x landmark blocks of the structured Riccati step for throughput-bound sizes.
//   Sigma'_IJ = (D_I Sigma_IJ + Lw_I Sigma_wJ + Lv_I Sigma_vJ) D_J^T + Gn_I Lw_J^T + Gv_I Lv_J^T  (+ T p I on the diagonal)
// One lane per COLUMN landmark J (its blocks D_J, Lw_J, Lv_J and the base rows Sigma_wJ, Sigma_vJ stay in registers), a
// workgroup walks kStreamRows ROW landmarks whose constants are wave-uniform LDS broadcasts: per 3x3 block 9 loads,
// 9 stores, 162 FMAs and a handful of address instructions -- k_propagate's tile code spends ~1500 instructions on the
// same block (staging loops, integer divisions, 64-bit address arithmetic) and is issue-bound, not memory-bound.
// grid = (column strips of 256 landmarks, ceil(N / kStreamRows), B), block = 256 (4 waves = 4 strips of 64 columns).
// The workgroups of the first row chunk also write the base rows / columns of their column landmarks (and the 11 x 11 base
// block); blocks and G rows come from k_build_blocks.  The split path is these two launches.
// ------------------------------------------------------------------------------------------------
constexpr int kStreamRows = 16;
template <typename T>
__global__ __launch_bounds__(256) void k_riccati_stream(PropArgs a) {
    const int b = blockIdx.z;
    const Glob& G = a.gin[b];
    const ImuRec& r = a.recs ? a.recs[b] : a.inl;
    const int N = G.N;
    const double dt0 = r.stamp - G.curTime;
    const bool step = (G.curTime >= 0) && (dt0 > 0);
    const bool riccati = step && a.doRiccati && !a.sigmaExternal;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int I0 = blockIdx.y * kStreamRows;
    const int J = (4 * blockIdx.x + wv) * 64 + lane;
    if (I0 >= N && blockIdx.y != 0) return;  // (the first row chunk also owns the base block: it runs even without landmarks)
    const int ld = a.ld;
    const T* Sin = static_cast<const T*>(a.Sin) + (long long)b * a.sigmaStride;
    T* Sout = static_cast<T*>(a.Sout) + (long long)b * a.sigmaStride;
    const int nI = max(0, min(kStreamRows, N - I0));
    const bool validJ = J < N;
    const T* colIn = Sin + kLm0 + 3 * (validJ ? J : 0);
    T* colOut = Sout + kLm0 + 3 * (validJ ? J : 0);
    if (!riccati) {
        if (a.sigmaExternal && step && a.doRiccati) return;  // Sigma_out was written by k_dense_gemm
        // Sigma is not touched by this call: copy through (the ping-pong parity of the batch stays in step)
        if (validJ) {
            for (int i = 0; i < 3 * nI; ++i) {
                const long long ro = (long long)(kLm0 + 3 * I0 + i) * ld;
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) colOut[ro + cc] = colIn[ro + cc];
            }
            if (blockIdx.y == 0) {
                for (int cc = 0; cc < 12; ++cc)
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
                        colOut[(long long)cc * ld + rr] = colIn[(long long)cc * ld + rr];
                        Sout[(long long)(kLm0 + 3 * J + rr) * ld + cc] = Sin[(long long)(kLm0 + 3 * J + rr) * ld + cc];
                    }
            }
        }
        if (blockIdx.y == 0 && blockIdx.x == 0 && tid < 144) Sout[(long long)(tid / 12) * ld + tid % 12] = Sin[(long long)(tid / 12) * ld + tid % 12];
        return;
    }
    __shared__ T sRow[kStreamRows][kBlkRec];
    const T* blk = static_cast<const T*>(a.blk) + (long long)b * a.cap * kBlkRec;
    for (int e = tid; e < nI * kBlkRec; e += 256) sRow[e / kBlkRec][e % kBlkRec] = blk[(long long)I0 * kBlkRec + e];
    // column constants
    T DJ[9], LwJ[9], LvJ[9], SwJ[9], SvJ[9];
    {
        const T* bj = blk + (long long)(validJ ? J : 0) * kBlkRec;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            DJ[k] = bj[k];
            LwJ[k] = bj[9 + k];
            LvJ[k] = bj[18 + k];
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                SwJ[3 * rr + cc] = colIn[(long long)rr * ld + cc];
                SvJ[3 * rr + cc] = colIn[(long long)(8 + rr) * ld + cc];
            }
    }
    const T TtP = (T)a.blkCommon[b].T * (T)a.prm.pointProcessVariance;
    if (blockIdx.y == 0) {
        // ---- the first row chunk also owns the base rows / columns of its column landmarks (and, its first workgroup,
        // the 11 x 11 base block):  Sigma'_bJ = F_bb (Sigma_bb L_J^T + Sigma_bJ D_J^T) - sigma_w^2 Nb Lw_J^T, and
        // Sigma'_Jb is written as its transpose (the tile kernel evaluates G_J F_bb^T: the same number up to rounding).
        __shared__ T sF[11][12], sNb[11][3], sSbb[11][12], sTb[11][12];
        __shared__ CommonLds sC;
        if (tid == 0) sC = a.blkCommon[b];
        if (tid < 132) {
            const int rr = tid / 12, cc = tid % 12;
            sSbb[rr][cc] = (cc < 11) ? Sin[(long long)rr * ld + cc] : (T)0;
        }
        __syncthreads();
        if (tid < 132) {
            const int rr = tid / 12, cc = tid % 12;
            // F_bb = I + T * [[0,0,0,0],[-B_g^w,0,0,0],[-B_v^w,-R_A, A_vg, 0]]   (VIOFilter.cpp:178-183)
            double f = (rr == cc) ? 1.0 : 0.0;
            if (rr >= 6 && rr < 8 && cc < 3) f = -sC.T * sC.Bg[3 * (rr - 6) + cc];
            if (rr >= 8) {
                if (cc < 3) f = -sC.T * sC.Bvw[3 * (rr - 8) + cc];
                else if (cc < 6) f = -sC.T * sC.RA[3 * (rr - 8) + cc - 3];
                else if (cc < 8) f = sC.T * sC.Avg[2 * (rr - 8) + cc - 6];
            }
            sF[rr][cc] = (cc < 11) ? (T)f : (T)0;
            if (cc < 3) {
                double nb = 0.0;
                if (rr >= 6 && rr < 8) nb = sC.Bg[3 * (rr - 6) + cc];
                if (rr >= 8) nb = sC.Bvw[3 * (rr - 8) + cc];
                sNb[rr][cc] = (T)nb;
            }
        }
        __syncthreads();
        const T sw2 = (T)a.prm.velOmegaVariance, sa2 = (T)a.prm.velAccelVariance, Tt = (T)sC.T;
        if (validJ) {
            T Gt[11][3];  // Sigma_bb L_J^T + Sigma_bJ D_J^T
#pragma unroll
            for (int cc = 0; cc < 11; ++cc) {
                T sb[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    sb[k] = (cc < 3) ? SwJ[3 * cc + k] : ((cc >= 8) ? SvJ[3 * (cc - 8) + k] : colIn[(long long)cc * ld + k]);
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    T acc = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        acc += sSbb[cc][k] * LwJ[3 * rr + k] + sSbb[cc][8 + k] * LvJ[3 * rr + k] + sb[k] * DJ[3 * rr + k];
                    Gt[cc][rr] = acc;
                }
            }
#pragma unroll
            for (int cc = 0; cc < 11; ++cc)
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    T acc = 0;
#pragma unroll
                    for (int k = 0; k < 11; ++k) acc += sF[cc][k] * Gt[k][rr];
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc -= sw2 * sNb[cc][k] * LwJ[3 * rr + k];
                    colOut[(long long)cc * ld + rr] = acc;                                       // Sigma'_bJ
                    Sout[(long long)(kLm0 + 3 * J + rr) * ld + cc] = acc;                        // Sigma'_Jb
                }
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {  // the structural pad row / column 11
                colOut[(long long)11 * ld + rr] = (T)0;
                Sout[(long long)(kLm0 + 3 * J + rr) * ld + 11] = (T)0;
            }
        }
        if (blockIdx.x == 0) {
            // Sigma'_bb = F_bb Sigma_bb F_bb^T + T (P_bb + B_b R B_b^T)
            if (tid < 121) {
                const int rr = tid / 11, cc = tid % 11;
                T acc = 0;
#pragma unroll
                for (int k = 0; k < 11; ++k) acc += sF[rr][k] * sSbb[k][cc];
                sTb[rr][cc] = acc;
            }
            __syncthreads();
            if (tid < 144) {
                const int rr = tid / 12, cc = tid % 12;
                T acc = 0;
                if (rr < 11 && cc < 11) {
#pragma unroll
                    for (int k = 0; k < 11; ++k) acc += sTb[rr][k] * sF[cc][k];
                    T nz = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) nz += sw2 * sNb[rr][k] * sNb[cc][k];
                    if (rr >= 8 && cc >= 8) {  // accel columns of B: rows 8:11 hold R_A
#pragma unroll
                        for (int k = 0; k < 3; ++k) nz += sa2 * (T)sC.RA[3 * (rr - 8) + k] * (T)sC.RA[3 * (cc - 8) + k];
                    }
                    if (rr == cc) {
                        const Params& p = a.prm;
                        nz += (T)(rr < 3 ? p.biasOmegaProcessVariance
                                         : (rr < 6 ? p.biasAccelProcessVariance : (rr < 8 ? p.gravityProcessVariance : p.velocityProcessVariance)));
                    }
                    acc += Tt * nz;
                }
                Sout[(long long)rr * ld + cc] = acc;  // row/col 11 stay zero
            }
        }
    }
    T S[9];
    auto fetch = [&](int i) {
        const long long ro = (long long)(kLm0 + 3 * (I0 + i)) * ld;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) S[3 * rr + cc] = colIn[ro + (long long)rr * ld + cc];
    };
    if (nI > 0) fetch(0);
    __syncthreads();
    for (int i = 0; i < nI; ++i) {
        T Sc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Sc[k] = S[k];
        if (i + 1 < nI) fetch(i + 1);  // next block's loads fly during this block's arithmetic
        const T* rc = sRow[i];         // wave-uniform: LDS broadcast reads
        T H[9];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                T acc = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    acc += rc[3 * rr + k] * Sc[3 * k + cc] + rc[9 + 3 * rr + k] * SwJ[3 * k + cc] + rc[18 + 3 * rr + k] * SvJ[3 * k + cc];
                H[3 * rr + cc] = acc;
            }
        const long long ro = (long long)(kLm0 + 3 * (I0 + i)) * ld;
        const bool diag = (I0 + i) == J;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                T acc = (diag && rr == cc) ? TtP : (T)0;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    acc += H[3 * rr + k] * DJ[3 * cc + k] + rc[27 + 3 * rr + k] * LwJ[3 * cc + k] + rc[36 + 3 * rr + k] * LvJ[3 * cc + k];
                if (validJ) colOut[ro + (long long)rr * ld + cc] = acc;
            }
    }
}

}  // namespace eqf
