// BASELINE configs[4] / SURVEY.md 8(e) row 2: ONE filter whose Sigma is 2-D block-partitioned over a Pr x Pc process grid -- the
// per-rank device side of the closed loop (the exchange schedule lives above the C ABI: eqf_vio_amd/tiled.py, torch.distributed
// over RCCL).  fp64.
//
// What a rank holds:
//   replicated (every rank advances its own identical copy, no communication): the O(N) filter state -- Glob, p0, Q, the
//     per-landmark constants -- and the 12-row BASE PANEL  Sb = Sigma[0:12, :]  in the single-GPU path's internal index map
//     (columns 0..11 the 11 x 11 base block + the structural pad, columns 12 + 3J + c landmark J);
//   distributed: its share  Sll  of the landmark x landmark part of Sigma, ScaLAPACK style: landmark blocks I = pr, pr + Pr, ... as
//     rows and J = pc, pc + Pc, ... as columns of ONE dense local matrix (3 nlr x 3 nlc, row-major).  rowMap / colMap translate a
//     local landmark index to the global one.
//   landmark SLOTS: the partition is over N physical slots; a slot whose landmark was removed (removeOldLandmarks / removeOutliers,
//     VIOFilter.cpp:393-443) stays where it is, INACTIVE: its rows and columns of Sigma are zero with a unit diagonal block, its
//     linearisation is the identity (D = I, L = 0), its measurement rows are C = 0, delta = 0, Z = 0 -- so it is carried through both
//     factorisations as a decoupled block that contributes exact zeros, and no row or column ever moves between ranks.  New landmarks
//     take the lowest free slots (k_tl_edit_state / k_tl_edit_local).  The reference's ORDER of landmarks (insertion order) lives
//     with the caller (eqf_vio_amd/tiled.py), as a permutation over the slots: the recursion is equivariant under it.
//
// Kernels (same device functions as the single-GPU path: stepCommon / buildBlocks / stepLandmark / stepGlobal of
// eqf_propagate.hpp, liftRows / updateFinishBody of eqf_update.hpp, i.e. the same restatement of VIOFilter.cpp:146-209, :264-297,
// EqFMatrices.cpp:173-382):
//   k_tl_build     per-landmark linearisation blocks D, Lw, Lv + the two 3x3 pieces of G_I (from the base panel: Sigma_Ib is
//                  the transpose of Sigma_bI), group step, scalar state            == k_build_blocks
//   k_tl_base      Sigma'_bJ, Sigma'_bb for ALL landmarks (replicated)             == the first row chunk of k_riccati_stream
//   k_tl_riccati   Sigma'_IJ for the LOCAL blocks, in place (a 3x3 block only needs itself, the blocks of I and J and the OLD
//                  base panel, which is ping-ponged)                               == k_riccati_stream
//   k_tl_prep      residual delta_i, V_i = C0i Z_i, Z_i (one lane per landmark)    == the landmark waves of k_update_prep
//   k_tl_form_s    local S-chain operand  M = [ C_I Sigma_IJ C_J^T (+R) | C_I Sigma_IJ | C_I Sigma_Ib, delta_I, V_I ]
//   k_tl_eprep     Cholesky of Sigma_gg (5 x 5), Pg = Lg^-1 Sigma_gL, the base part of G11
//   k_tl_form_e    local E-chain operand  E = [ Sigma_IJ - Pg_I^T Pg_J | Z_I, -Pg_I^T Lg^-1 ]  (the Schur complement of Sigma_e =
//                  Sigma[6:, 6:] after its five base coordinates: EqFMatrices.cpp:239 without the explicit inverse)
//   k_tl_finish    gamma / hV / G11 -> updateFinishBody (bundleLift's 4 x 4 least squares, Delta, X <- Delta X, bias), then the
//                  base panel's share of Sigma - K C Sigma
#pragma once
#include "eqf_propagate.hpp"
#include "eqf_update.hpp"

namespace eqf {

constexpr int kTlNarrowS = 18;  // narrow right-hand sides of the S-chain: (C Sigma)_Ib (11) | delta | V (6)
constexpr int kTlNarrowE = 11;  // ... of the E-chain: Z_P (6) | E_top (5)

struct TlArgs {
    const Glob* gin;
    Glob* gout;
    const double* p0;   // [3][cap]
    const double* Qin;  // [5][cap]
    double* Qout;
    const double* SbIn;  // [12][ldb] base panel before the step
    double* SbOut;
    int ldb, cap;
    ImuRec inl;
    int isImu, doRiccati;
    double* blk;             // [cap][kBlkRec]
    CommonLds* blkCommon;    // [1]
    int* errflag;
    Params prm;
    // local share of the landmark x landmark part
    double* Sll;
    int ldl, nlr, nlc;
    const int* rowMap;  // [nlr] local row landmark -> global landmark
    const int* colMap;  // [nlc]
    const int* active;  // [cap] 1: the slot holds a landmark; 0: hole (see the header of this file)
    double* blkT;       // [27][cap] or nullptr: D, Lw, Lv of every landmark once more, transposed -- the column side of k_tl_riccati_burst
};

// ---------------------------------------------------------------------------------------------------------------------------
// grid = ceil(N / 64) + 1, block = 128 (see k_build_blocks: wave 0 blocks, wave 1 group step; last workgroup the scalar state)
__global__ __launch_bounds__(128) void k_tl_build(TlArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool isState = blockIdx.x == gridDim.x - 1;
    const int i = blockIdx.x * 64 + lane;
    const int cap = a.cap;
    const Glob& G = *a.gin;
    const ImuRec& r = a.inl;
    const int N = G.N;
    const double dt0 = r.stamp - G.curTime;
    const bool step = (G.curTime >= 0) && (dt0 > 0);
    const bool riccati = step && a.doRiccati;
    int bad = 0;
    if (isState) {
        if (wv == 0) {
            const double* src = reinterpret_cast<const double*>(&G);
            double* dst = reinterpret_cast<double*>(a.gout);
            if (lane < (int)(sizeof(Glob) / 8)) dst[lane] = src[lane];
            if (lane == 0) {
                StepCommon c;
                c.step = 0;
                if (step) stepCommon(G, r, a, c, kPartBase, &bad);
                stepGlobal(G, a.gout, r, a, c, &bad);
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
            *a.blkCommon = cl;
        }
        if (bad && a.errflag) atomicOr(a.errflag, 1);
        return;
    }
    if (i >= N) return;
    const quat Qq = quat{a.Qin[i], a.Qin[cap + i], a.Qin[2 * cap + i], a.Qin[3 * cap + i]};
    const double Qa = a.Qin[4 * cap + i];
    const d3 q0 = mk3(a.p0[i], a.p0[cap + i], a.p0[2 * cap + i]);
    const bool act = a.active[i] != 0;
    if (wv == 0) {
        if (riccati) {
            StepCommon c;
            stepCommon(G, r, a, c, kPartBase | kPartRicc, &bad);
            const LmBlocks blk = buildBlocks(c, Qq, Qa, q0);
            double* bp = a.blk + (long long)i * kBlkRec;
            double D[9], Lw[9], Lv[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                // a hole steps with the identity: its zero rows stay zero (only T p lands on its diagonal block)
                D[k] = act ? blk.D.a[k] : ((k % 4 == 0) ? 1.0 : 0.0);
                Lw[k] = act ? blk.Lw.a[k] : 0.0;
                Lv[k] = act ? blk.Lv.a[k] : 0.0;
                bp[k] = D[k];
                bp[9 + k] = Lw[k];
                bp[18 + k] = Lv[k];
                if (a.blkT) {
                    a.blkT[(long long)k * cap + i] = D[k];
                    a.blkT[(long long)(9 + k) * cap + i] = Lw[k];
                    a.blkT[(long long)(18 + k) * cap + i] = Lv[k];
                }
            }
            // G_I[:, 0:3] and G_I[:, 8:11] of G_I = Lw Sigma[0:3, :] + Lv Sigma[8:11, :] + D Sigma_Ib, Sigma_Ib[k][c] = Sb[c][12 + 3 i + k]
            // (same expression order as k_build_blocks)
            const double* Sb = a.SbIn;
            const int ld = a.ldb;
            const double sw2 = a.prm.velOmegaVariance, Tt = c.T;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c0 = half ? 8 : 0;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        double acc = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            acc += Lw[3 * rr + k] * Sb[(long long)k * ld + c0 + cc] + Lv[3 * rr + k] * Sb[(long long)(8 + k) * ld + c0 + cc] +
                                   D[3 * rr + k] * Sb[(long long)(c0 + cc) * ld + kLm0 + 3 * i + k];
                        bp[27 + 9 * half + 3 * rr + cc] = half ? acc : acc + (sw2 / Tt) * Lw[3 * rr + cc];
                    }
            }
        }
    } else {
        quat Qo = Qq;
        double ao = Qa;
        if (step && act) {
            StepCommon c;
            stepCommon(G, r, a, c, kPartBase | kPartLift, &bad);
            stepLandmark(c, a, Qq, Qa, q0, &Qo, &ao, &bad);
        }
        a.Qout[i] = Qo.w; a.Qout[cap + i] = Qo.x; a.Qout[2 * cap + i] = Qo.y; a.Qout[3 * cap + i] = Qo.z;
        a.Qout[4 * cap + i] = ao;
    }
    if (bad && a.errflag) atomicOr(a.errflag, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Base panel: Sigma'_bJ = F_bb (Sigma_bb L_J^T + Sigma_bJ D_J^T) - sigma_w^2 Nb Lw_J^T for every landmark J (one lane each), and
// Sigma'_bb = F_bb Sigma_bb F_bb^T + T (P_bb + B_b R B_b^T) (first workgroup).  grid = max(1, ceil(N / 256)), block = 256.
// Filters that do not step copy the panel through (the ping-pong parity stays in step).
__global__ __launch_bounds__(256) void k_tl_base(TlArgs a) {
    const Glob& G = *a.gin;
    const ImuRec& r = a.inl;
    const int N = G.N;
    const double dt0 = r.stamp - G.curTime;
    const bool riccati = (G.curTime >= 0) && (dt0 > 0) && a.doRiccati;
    const int tid = threadIdx.x, J = blockIdx.x * 256 + tid, ld = a.ldb;
    const double* Sin = a.SbIn;
    double* Sout = a.SbOut;
    const bool validJ = J < N;
    const double* colIn = Sin + kLm0 + 3 * (validJ ? J : 0);
    double* colOut = Sout + kLm0 + 3 * (validJ ? J : 0);
    if (!riccati) {
        if (validJ)
            for (int cc = 0; cc < 12; ++cc)
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) colOut[(long long)cc * ld + rr] = colIn[(long long)cc * ld + rr];
        if (blockIdx.x == 0 && tid < 144) Sout[(long long)(tid / 12) * ld + tid % 12] = Sin[(long long)(tid / 12) * ld + tid % 12];
        return;
    }
    __shared__ double sF[11][12], sNb[11][3], sSbb[11][12], sTb[11][12];
    __shared__ CommonLds sC;
    if (tid == 0) sC = *a.blkCommon;
    if (tid < 132) {
        const int rr = tid / 12, cc = tid % 12;
        sSbb[rr][cc] = (cc < 11) ? Sin[(long long)rr * ld + cc] : 0.0;
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
        sF[rr][cc] = (cc < 11) ? f : 0.0;
        if (cc < 3) {
            double nb = 0.0;
            if (rr >= 6 && rr < 8) nb = sC.Bg[3 * (rr - 6) + cc];
            if (rr >= 8) nb = sC.Bvw[3 * (rr - 8) + cc];
            sNb[rr][cc] = nb;
        }
    }
    __syncthreads();
    const double sw2 = a.prm.velOmegaVariance, sa2 = a.prm.velAccelVariance, Tt = sC.T;
    if (validJ) {
        const double* bj = a.blk + (long long)J * kBlkRec;
        double DJ[9], LwJ[9], LvJ[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            DJ[k] = bj[k];
            LwJ[k] = bj[9 + k];
            LvJ[k] = bj[18 + k];
        }
        double Gt[11][3];  // Sigma_bb L_J^T + Sigma_bJ D_J^T
#pragma unroll
        for (int cc = 0; cc < 11; ++cc) {
            double sb[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) sb[k] = colIn[(long long)cc * ld + k];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                double acc = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) acc += sSbb[cc][k] * LwJ[3 * rr + k] + sSbb[cc][8 + k] * LvJ[3 * rr + k] + sb[k] * DJ[3 * rr + k];
                Gt[cc][rr] = acc;
            }
        }
#pragma unroll
        for (int cc = 0; cc < 11; ++cc)
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                double acc = 0;
#pragma unroll
                for (int k = 0; k < 11; ++k) acc += sF[cc][k] * Gt[k][rr];
#pragma unroll
                for (int k = 0; k < 3; ++k) acc -= sw2 * sNb[cc][k] * LwJ[3 * rr + k];
                colOut[(long long)cc * ld + rr] = acc;
            }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) colOut[(long long)11 * ld + rr] = 0.0;
    }
    if (blockIdx.x == 0) {
        if (tid < 121) {
            const int rr = tid / 11, cc = tid % 11;
            double acc = 0;
#pragma unroll
            for (int k = 0; k < 11; ++k) acc += sF[rr][k] * sSbb[k][cc];
            sTb[rr][cc] = acc;
        }
        __syncthreads();
        if (tid < 144) {
            const int rr = tid / 12, cc = tid % 12;
            double acc = 0;
            if (rr < 11 && cc < 11) {
#pragma unroll
                for (int k = 0; k < 11; ++k) acc += sTb[rr][k] * sF[cc][k];
                double nz = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) nz += sw2 * sNb[rr][k] * sNb[cc][k];
                if (rr >= 8 && cc >= 8) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) nz += sa2 * sC.RA[3 * (rr - 8) + k] * sC.RA[3 * (cc - 8) + k];
                }
                if (rr == cc) {
                    const Params& p = a.prm;
                    nz += (rr < 3 ? p.biasOmegaProcessVariance
                                  : (rr < 6 ? p.biasAccelProcessVariance : (rr < 8 ? p.gravityProcessVariance : p.velocityProcessVariance)));
                }
                acc += Tt * nz;
            }
            Sout[(long long)rr * ld + cc] = acc;
        }
    }
}

// One Riccati step of one 3 x 3 block: out = (D_I S + Lw_I Sw_J + Lv_I Sv_J) D_J^T + Gn_I Lw_J^T + Gv_I Lv_J^T (+ diagAdd on the diagonal);
// rc = the row landmark's record (D, Lw, Lv, Gn, Gv).  ONE function for k_tl_riccati and k_tl_riccati_burst: same operations, same order.
EQF_DI void tlRiccatiBlock(const double* Sc, const double* rc, const double* DJ, const double* LwJ, const double* LvJ, const double* SwJ,
    const double* SvJ, double diagAdd, double* out) {
    double H[9];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {  // (explicit fma chain: the compiler's contraction choices must not differ between the two kernels)
                acc = fma(rc[3 * rr + k], Sc[3 * k + cc], acc);
                acc = fma(rc[9 + 3 * rr + k], SwJ[3 * k + cc], acc);
                acc = fma(rc[18 + 3 * rr + k], SvJ[3 * k + cc], acc);
            }
            H[3 * rr + cc] = acc;
        }
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            double acc = (rr == cc) ? diagAdd : 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                acc = fma(H[3 * rr + k], DJ[3 * cc + k], acc);
                acc = fma(rc[27 + 3 * rr + k], LwJ[3 * cc + k], acc);
                acc = fma(rc[36 + 3 * rr + k], LvJ[3 * cc + k], acc);
            }
            out[3 * rr + cc] = acc;
        }
}
// ---------------------------------------------------------------------------------------------------------------------------
// Local blocks, IN PLACE:  Sigma'_IJ = (D_I Sigma_IJ + Lw_I Sigma_wJ + Lv_I Sigma_vJ) D_J^T + Gn_I Lw_J^T + Gv_I Lv_J^T (+ T p I on the
// global diagonal).  One lane per local COLUMN landmark, a workgroup walks kStreamRows local ROW landmarks whose records are
// wave-uniform LDS broadcasts (the loop of k_riccati_stream).  grid = (ceil(nlc / 256), ceil(nlr / 16)), block = 256.
__global__ __launch_bounds__(256) void k_tl_riccati(TlArgs a) {
    const Glob& G = *a.gin;
    const ImuRec& r = a.inl;
    const double dt0 = r.stamp - G.curTime;
    if (!((G.curTime >= 0) && (dt0 > 0) && a.doRiccati)) return;
    const int tid = threadIdx.x;
    const int I0 = blockIdx.y * kStreamRows;
    const int jl = blockIdx.x * 256 + tid;
    const int nI = max(0, min(kStreamRows, a.nlr - I0));
    const bool validJ = jl < a.nlc;
    const int ldl = a.ldl, ldb = a.ldb;
    __shared__ double sRow[kStreamRows][kBlkRec];
    __shared__ int sGI[kStreamRows];
    for (int e = tid; e < nI * kBlkRec; e += 256) {
        const int il = e / kBlkRec, q = e % kBlkRec;
        sRow[il][q] = a.blk[(long long)a.rowMap[I0 + il] * kBlkRec + q];
    }
    if (tid < nI) sGI[tid] = a.rowMap[I0 + tid];
    const int J = validJ ? a.colMap[jl] : 0;
    double DJ[9], LwJ[9], LvJ[9], SwJ[9], SvJ[9];
    {
        const double* bj = a.blk + (long long)J * kBlkRec;
        const double* colIn = a.SbIn + kLm0 + 3 * J;
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
                SwJ[3 * rr + cc] = colIn[(long long)rr * ldb + cc];
                SvJ[3 * rr + cc] = colIn[(long long)(8 + rr) * ldb + cc];
            }
    }
    const double TtP = a.blkCommon->T * a.prm.pointProcessVariance;
    double* col = a.Sll + 3 * (validJ ? jl : 0);
    double S[9];
    auto fetch = [&](int i) {
        const long long ro = (long long)(3 * (I0 + i)) * ldl;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) S[3 * rr + cc] = col[ro + (long long)rr * ldl + cc];
    };
    if (nI > 0) fetch(0);
    __syncthreads();
    for (int i = 0; i < nI; ++i) {
        double Sc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Sc[k] = S[k];
        if (i + 1 < nI) fetch(i + 1);
        double out[9];
        tlRiccatiBlock(Sc, sRow[i], DJ, LwJ, LvJ, SwJ, SvJ, sGI[i] == J ? TtP : 0.0, out);
        const long long ro = (long long)(3 * (I0 + i)) * ldl;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc)
                if (validJ) col[ro + (long long)rr * ldl + cc] = out[3 * rr + cc];
    }
}

// K consecutive Riccati steps of the local blocks in ONE pass over Sll (IMU bursts: the calls between two vision frames + the vision call's
// integrateUpToTime).  A step of a 3 x 3 block needs the block itself, the records of its row and column landmark FOR THAT STEP and the
// base panel BEFORE that step -- all O(N) per step, produced by the K (k_tl_build, k_tl_base) pairs that ran first with one record set
// and one panel per step.  So a thread keeps kBurstRows blocks of its column landmark in registers, walks the steps, and Sll is read and
// written once instead of K times (2 |Sll| 8 bytes per burst instead of per step: 2.3 GB at N = 4000 on one GPU).  Same tlRiccatiBlock,
// same order: the result of a burst equals the K single steps.  grid = (ceil(nlc / 256), ceil(nlr / kBurstRows)), block = 256.
constexpr int kBurstRows = 4, kTlBurstMax = 16;
struct TlBurstArgs {
    int nSteps;
    const double* blk[kTlBurstMax];     // per step: [cap][kBlkRec] row records
    const double* blkT[kTlBurstMax];    // per step: [27][cap] column records
    const double* SbIn[kTlBurstMax];    // per step: the base panel before the step
    const CommonLds* common[kTlBurstMax];
    double pointVar;
    int ldb, cap;
    double* Sll;
    int ldl, nlr, nlc;
    const int* rowMap;
    const int* colMap;
};
__global__ __launch_bounds__(256) void k_tl_riccati_burst(TlBurstArgs a) {
    const int tid = threadIdx.x;
    const int I0 = blockIdx.y * kBurstRows;
    const int jl = blockIdx.x * 256 + tid;
    const int nI = max(0, min(kBurstRows, a.nlr - I0));
    const bool validJ = jl < a.nlc;
    const int ldl = a.ldl, ldb = a.ldb, cap = a.cap;
    __shared__ double sRow[kTlBurstMax][kBurstRows][kBlkRec];
    __shared__ int sGI[kBurstRows];
    __shared__ double sTtP[kTlBurstMax];
    for (int e = tid; e < a.nSteps * nI * kBlkRec; e += 256) {
        const int s_ = e / (nI * kBlkRec), r = e % (nI * kBlkRec), il = r / kBlkRec, q = r % kBlkRec;
        sRow[s_][il][q] = a.blk[s_][(long long)a.rowMap[I0 + il] * kBlkRec + q];
    }
    if (tid < nI) sGI[tid] = a.rowMap[I0 + tid];
    if (tid < a.nSteps) sTtP[tid] = a.common[tid]->T * a.pointVar;
    const int J = validJ ? a.colMap[jl] : 0;
    double* col = a.Sll + 3 * (validJ ? jl : 0);
    double S[kBurstRows][9];
#pragma unroll
    for (int i = 0; i < kBurstRows; ++i)
        if (i < nI) {
            const long long ro = (long long)(3 * (I0 + i)) * ldl;
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) S[i][3 * rr + cc] = col[ro + (long long)rr * ldl + cc];
        }
    __syncthreads();
    for (int s_ = 0; s_ < a.nSteps; ++s_) {
        double DJ[9], LwJ[9], LvJ[9], SwJ[9], SvJ[9];
        const double* bt = a.blkT[s_] + J;
        const double* colIn = a.SbIn[s_] + kLm0 + 3 * J;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            DJ[k] = bt[(long long)k * cap];
            LwJ[k] = bt[(long long)(9 + k) * cap];
            LvJ[k] = bt[(long long)(18 + k) * cap];
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                SwJ[3 * rr + cc] = colIn[(long long)rr * ldb + cc];
                SvJ[3 * rr + cc] = colIn[(long long)(8 + rr) * ldb + cc];
            }
        const double TtP = sTtP[s_];
#pragma unroll
        for (int i = 0; i < kBurstRows; ++i)
            if (i < nI) {
                double out[9];
                tlRiccatiBlock(S[i], sRow[s_][i], DJ, LwJ, LvJ, SwJ, SvJ, sGI[i] == J ? TtP : 0.0, out);
#pragma unroll
                for (int k = 0; k < 9; ++k) S[i][k] = out[k];
            }
    }
    if (!validJ) return;
#pragma unroll
    for (int i = 0; i < kBurstRows; ++i)
        if (i < nI) {
            const long long ro = (long long)(3 * (I0 + i)) * ldl;
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) col[ro + (long long)rr * ldl + cc] = S[i][3 * rr + cc];
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
struct TlUpdArgs {
    Glob* g;            // current scalar state (updated in place by the finish kernel)
    const double* p0;   // [3][cap]
    double* lmc;        // [15][cap]
    double* Q;          // [5][cap] current
    double* Sb;         // [12][ldb] current base panel (downdated in place by the finish kernel)
    int ldb, cap;
    const double* bearings;  // [N][3], in state order
    double* delta;      // [2 cap]
    double* Zrows;      // [cap][18]
    double* Vrows;      // [cap][12]
    double* Pg;         // [5][ldp]   Lg^-1 Sigma_gL
    double* Lgi;        // [25]       Lg^-1 (lower)
    int ldp;
    int* errflag;
    Params prm;
    const double* Sll;
    int ldl, nlr, nlc;
    const int* rowMap;
    const int* colMap;
    const int* active;  // [cap]
    double* M;  // S-chain operand [2 nlr][ldm]: columns [0, 2 nlc) S, [2 nlc, 5 nlc) C Sigma, [5 nlc, 5 nlc + 18) narrow
    int ldm;
    double* E;  // E-chain operand [3 nlr][lde]: columns [0, 3 nlc) Schur complement, [3 nlc, 3 nlc + 11) narrow
    int lde;
    double* G11;  // [11][11] base part of [Zt | Et]^T [Zt | Et]
};

// one lane per landmark: residual, V, Z (and, lanes 0..: nothing else).  grid = ceil(N / 64), block = 64
__global__ __launch_bounds__(64) void k_tl_prep(TlUpdArgs a) {
    const Glob& g = *a.g;
    const int N = g.N, cap = a.cap;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (!g.updateOk || i >= N) return;
    if (!a.active[i]) {  // a hole measures nothing: with C = 0 (lmc) its two rows of S are R, everything else of it exact zeros
        a.delta[2 * i] = a.delta[2 * i + 1] = 0.0;
        for (int k = 0; k < 18; ++k) a.Zrows[(long long)i * 18 + k] = 0.0;
        for (int k = 0; k < 12; ++k) a.Vrows[(long long)i * 12 + k] = 0.0;
        return;
    }
    const quat Qq = quat{a.Q[i], a.Q[cap + i], a.Q[2 * cap + i], a.Q[3 * cap + i]};
    const double Qa = a.Q[4 * cap + i];
    const d3 q0 = mk3(a.p0[i], a.p0[cap + i], a.p0[2 * cap + i]);
    const d3 y = mk3(a.bearings[3 * i], a.bearings[3 * i + 1], a.bearings[3 * i + 2]);
    // yerr = (X^-1).Q_i.R()^-1 y (outputGroupAction, VIOGroup.cpp:84,130); delta = e3ProjectSphere(R_s yerr)
    const d3 yerr = qrot(qinv(qinv(Qq)), y);
    double C[6];
    m33 Rs;
#pragma unroll
    for (int q = 0; q < 6; ++q) C[q] = a.lmc[(long long)q * cap + i];
#pragma unroll
    for (int q = 0; q < 9; ++q) Rs.a[q] = a.lmc[(long long)(6 + q) * cap + i];
    const d3 rr = mv33(Rs, yerr);
    a.delta[2 * i] = rr.x / (1 - rr.z);  // VIOState.cpp:199-204
    a.delta[2 * i + 1] = rr.y / (1 - rr.z);
    double Z[18];
    liftRows(liftCommon(g, a.prm), Qq, Qa, q0, Z);
#pragma unroll
    for (int k = 0; k < 18; ++k) a.Zrows[(long long)i * 18 + k] = Z[k];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) a.Vrows[(long long)i * 12 + 6 * r + c] = C[3 * r] * Z[c] + C[3 * r + 1] * Z[6 + c] + C[3 * r + 2] * Z[12 + c];
}

// S-chain operand of the local blocks.  grid = (ceil(nlc / 256) + 1, ceil(nlr / 16)), block = 256: lane = local column landmark,
// 16 local row landmarks per workgroup; the extra column workgroup writes the narrow right-hand sides of its rows.
__global__ __launch_bounds__(256) void k_tl_form_s(TlUpdArgs a) {
    const Glob& g = *a.g;
    if (!g.updateOk) return;
    const int tid = threadIdx.x, cap = a.cap;
    const int I0 = blockIdx.y * kStreamRows;
    const int nI = max(0, min(kStreamRows, a.nlr - I0));
    const int ldm = a.ldm;
    if (blockIdx.x == gridDim.x - 1) {
        // narrow part: rows 2 il + r: [ (C Sigma)_Ib (11) | delta | V (6) ];  (C Sigma)_Ib[r][col] = sum_c C[r][c] Sb[col][12 + 3 I + c]
        for (int e = tid; e < nI * 2 * kTlNarrowS; e += 256) {
            const int il = e / (2 * kTlNarrowS), r = (e / kTlNarrowS) % 2, col = e % kTlNarrowS;
            const int I = a.rowMap[I0 + il];
            double v;
            if (col < 11) {
                v = 0.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) v += a.lmc[(long long)(3 * r + c) * cap + I] * a.Sb[(long long)col * a.ldb + kLm0 + 3 * I + c];
            } else if (col == 11) {
                v = a.delta[2 * I + r];
            } else {
                v = a.Vrows[(long long)I * 12 + 6 * r + col - 12];
            }
            a.M[(long long)(2 * (I0 + il) + r) * ldm + 5 * a.nlc + col] = v;
        }
        return;
    }
    const int jl = blockIdx.x * 256 + tid;
    const bool validJ = jl < a.nlc;
    __shared__ double sC[kStreamRows][6];
    __shared__ int sGI[kStreamRows];
    if (tid < nI * 6) sC[tid / 6][tid % 6] = a.lmc[(long long)(tid % 6) * cap + a.rowMap[I0 + tid / 6]];
    if (tid < nI) sGI[tid] = a.rowMap[I0 + tid];
    const int J = validJ ? a.colMap[jl] : 0;
    double CJ[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) CJ[q] = a.lmc[(long long)q * cap + J];
    __syncthreads();
    if (!validJ) return;
    const double* col = a.Sll + 3 * jl;
    for (int i = 0; i < nI; ++i) {
        const long long ro = (long long)(3 * (I0 + i)) * a.ldl;
        double S[9];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) S[3 * rr + cc] = col[ro + (long long)rr * a.ldl + cc];
        const double* Ci = sC[i];
        double W[6];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) W[3 * r + c] = Ci[3 * r] * S[c] + Ci[3 * r + 1] * S[3 + c] + Ci[3 * r + 2] * S[6 + c];
        const bool diag = sGI[i] == J;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            double* mrow = a.M + (long long)(2 * (I0 + i) + r) * ldm;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                double v = W[3 * r] * CJ[3 * s] + W[3 * r + 1] * CJ[3 * s + 1] + W[3 * r + 2] * CJ[3 * s + 2];
                if (diag && r == s) v += a.prm.measurementVariance;
                mrow[2 * jl + s] = v;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) mrow[2 * a.nlc + 3 * jl + c] = W[3 * r + c];
        }
    }
}

// Cholesky of Sigma_gg = Sigma[6:11, 6:11] (gravity + velocity coordinates: the first five of Sigma_e), Pg = Lg^-1 Sigma_gL for every
// landmark column (one lane each, the 5 x 5 factor recomputed per lane: 35 flops), Lg^-1, and the base part of G11: V_g = [0 | I5]
// so G11[6:11, 6:11] = Lg^-T Lg^-1.  grid = ceil(3 N / 256), block = 256.
EQF_DI void chol5(const double* Sb, int ldb, double L[5][5], int* bad) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
#pragma unroll
        for (int i = 0; i < 5; ++i) L[i][j] = (i >= j) ? Sb[(long long)(6 + i) * ldb + 6 + j] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        double d = L[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        if (!(d > 0.0)) *bad = 1;
        const double s = sqrt(d), inv = 1.0 / s;
        L[j][j] = s;
#pragma unroll
        for (int i = j + 1; i < 5; ++i) {
            double v = L[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
            L[i][j] = v * inv;
        }
    }
}
__global__ __launch_bounds__(256) void k_tl_eprep(TlUpdArgs a) {
    const Glob& g = *a.g;
    if (!g.updateOk) return;
    const int N = g.N;
    const int col = blockIdx.x * 256 + threadIdx.x;
    double L[5][5];
    int bad = 0;
    chol5(a.Sb, a.ldb, L, &bad);
    if (col < 3 * N) {
        double x[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            double v = a.Sb[(long long)(6 + i) * a.ldb + kLm0 + col];
#pragma unroll
            for (int k = 0; k < i; ++k) v -= L[i][k] * x[k];
            x[i] = v / L[i][i];
            a.Pg[(long long)i * a.ldp + col] = x[i];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double Li[5][5];  // Lg^-1 column by column
#pragma unroll
        for (int e = 0; e < 5; ++e) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                double v = (i == e) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; ++k) v -= L[i][k] * Li[k][e];
                Li[i][e] = v / L[i][i];
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int e = 0; e < 5; ++e) a.Lgi[5 * i + e] = Li[i][e];
        for (int r = 0; r < 11; ++r)
            for (int c = 0; c < 11; ++c) {
                double v = 0.0;
                if (r >= 6 && c >= 6)
                    for (int q = 0; q < 5; ++q) v += Li[q][r - 6] * Li[q][c - 6];
                a.G11[11 * r + c] = v;
            }
        if (bad && a.errflag) atomicOr(a.errflag, 4);
    }
}

// E-chain operand of the local blocks: E_IJ = Sigma_IJ - Pg_I^T Pg_J; narrow rows [ Z_I (6) | -Pg_I^T Lg^-1 (5) ].
// grid = (ceil(nlc / 256) + 1, ceil(nlr / 16)), block = 256.
__global__ __launch_bounds__(256) void k_tl_form_e(TlUpdArgs a) {
    const Glob& g = *a.g;
    if (!g.updateOk) return;
    const int tid = threadIdx.x;
    const int I0 = blockIdx.y * kStreamRows;
    const int nI = max(0, min(kStreamRows, a.nlr - I0));
    const int lde = a.lde;
    __shared__ double sP[kStreamRows][15];  // Pg[0:5][3 I + r] as [r][q]
    __shared__ double sL[25];
    if (tid < nI * 15) {
        const int il = tid / 15, r = (tid % 15) / 5, q = tid % 5;
        sP[il][5 * r + q] = a.Pg[(long long)q * a.ldp + 3 * a.rowMap[I0 + il] + r];
    }
    if (blockIdx.x == gridDim.x - 1) {
        if (tid < 25) sL[tid] = a.Lgi[tid];
        __syncthreads();
        for (int e = tid; e < nI * 3 * kTlNarrowE; e += 256) {
            const int il = e / (3 * kTlNarrowE), r = (e / kTlNarrowE) % 3, col = e % kTlNarrowE;
            const int I = a.rowMap[I0 + il];
            double v;
            if (col < 6) {
                v = a.Zrows[(long long)I * 18 + 6 * r + col];
            } else {
                v = 0.0;
#pragma unroll
                for (int q = 0; q < 5; ++q) v -= sP[il][5 * r + q] * sL[5 * q + col - 6];
            }
            a.E[(long long)(3 * (I0 + il) + r) * lde + 3 * a.nlc + col] = v;
        }
        return;
    }
    __syncthreads();
    const int jl = blockIdx.x * 256 + tid;
    if (jl >= a.nlc) return;
    const int J = a.colMap[jl];
    double PJ[15];  // [c][q]
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 5; ++q) PJ[5 * c + q] = a.Pg[(long long)q * a.ldp + 3 * J + c];
    const double* col = a.Sll + 3 * jl;
    for (int i = 0; i < nI; ++i) {
        const long long ro = (long long)(3 * (I0 + i)) * a.ldl;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                double v = col[ro + (long long)rr * a.ldl + cc];
#pragma unroll
                for (int q = 0; q < 5; ++q) v -= sP[i][5 * rr + q] * PJ[5 * cc + q];
                a.E[(long long)(3 * (I0 + i) + rr) * lde + 3 * jl + cc] = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// After the two chains: acc [18][ldacc] (global landmark order, 3 N columns) = sum_k Yn_k^T Y_k, Gnn [18][18] = sum_k Yn_k^T Yn_k with
// Yn = [Yb (11) | z | Vt (6)], G11 [11][11].  gamma_b = Gnn[0:11, 11], gamma_L = acc[11, :], hV = Gnn[12:18, 11]; then bundleLift /
// Delta / X <- Delta X / bias (updateFinishBody), and the base panel's downdate: Sigma_bL -= acc[0:11, :], Sigma_bb -= Gnn[0:11, 0:11].
// ONE workgroup of 256 threads.
struct TlFinArgs {
    UpdArgs u;  // g, p0, Q, cap, dbgGamma, dbgGammaTot, prm, errflag are used
    const double* acc;
    int ldacc;
    const double* Gnn;  // [18][18]
    const double* G11;  // [11][11]
    double* Sb;
    int ldb;
};
__global__ __launch_bounds__(256) void k_tl_finish(TlFinArgs a) {
    const Glob& g = *a.u.g;
    if (!g.updateOk || g.N == 0) return;
    const int N = g.N, tid = threadIdx.x;
    __shared__ double sRed[256];
    double* gam = a.u.dbgGamma;
    for (int c = tid; c < kLm0 + 3 * N; c += 256) {
        double v = 0.0;
        if (c < 11) v = a.Gnn[kTlNarrowS * c + 11];
        else if (c >= kLm0) v = a.acc[(long long)11 * a.ldacc + c - kLm0];
        gam[c] = v;
    }
    if (tid < 6) sRed[tid] = a.Gnn[kTlNarrowS * (12 + tid) + 11];
    if (tid < 121) sRed[8 + tid] = a.G11[tid];
    // the base panel's share of Sigma - Y^T Y (uses nothing the lift changes)
    for (int e = tid; e < 11 * 3 * N; e += 256) {
        const int r = e / (3 * N), c = e % (3 * N);
        a.Sb[(long long)r * a.ldb + kLm0 + c] -= a.acc[(long long)r * a.ldacc + c];
    }
    if (tid < 121) a.Sb[(long long)(tid / 11) * a.ldb + tid % 11] -= a.Gnn[kTlNarrowS * (tid / 11) + tid % 11];
    __threadfence_block();
    __syncthreads();
    updateFinishBody(a.u, 0, sRed);
}

// ---------------------------------------------------------------------------------------------------------------------------
// addNewLandmarks on an EMPTY state (VIOFilter.cpp:345-391 with :361-366's initialSceneDepth branch): p0 = y depth, Q = identity,
// constants; base panel columns zero.  grid = ceil(n / 128), block = 128.
__global__ void k_tl_append(Glob* g0, Glob* g1, int n, double depth, int cap, const double* bearings, double* p0, double* Q0, double* Q1, double* lmc,
    double* Sb0, double* Sb1, int ldb, int* active, int* errflag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) g0->N = g1->N = n;
    if (i >= n) return;
    active[i] = 1;
    const double* y = bearings + 3 * i;
    d3 p = mk3(y[0] * depth, y[1] * depth, y[2] * depth);
#if defined(__HIP_DEVICE_COMPILE__)
    __asm__ volatile("" : "+v"(p.x), "+v"(p.y), "+v"(p.z));  // (the constants must come from p0 AS STORED: eqf_churn.hpp, k_append)
#endif
    p0[i] = p.x; p0[cap + i] = p.y; p0[2 * cap + i] = p.z;
    for (double* Q : {Q0, Q1}) {
        Q[i] = 1.0; Q[cap + i] = 0.0; Q[2 * cap + i] = 0.0; Q[3 * cap + i] = 0.0; Q[4 * cap + i] = 1.0;
    }
    double cst[15];
    int bad = 0;
    landmarkConstants(p, cst, &bad);
    for (int c = 0; c < 15; ++c) lmc[(long long)c * cap + i] = cst[c];
    for (double* Sb : {Sb0, Sb1})
        for (int r = 0; r < 12; ++r)
            for (int c = 0; c < 3; ++c) Sb[(long long)r * ldb + kLm0 + 3 * i + c] = 0.0;
    if (bad && errflag) atomicOr(errflag, 16);
}
// local blocks of a freshly appended landmark set: initialPointVariance on the global diagonal, zero elsewhere
__global__ void k_tl_init_local(double* Sll, int ldl, int nlr, int nlc, const int* rowMap, const int* colMap, double pointVar) {
    const int jl = blockIdx.x * blockDim.x + threadIdx.x, il = blockIdx.y;
    if (jl >= nlc || il >= nlr) return;
    const bool diag = rowMap[il] == colMap[jl];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Sll[(long long)(3 * il + r) * ldl + 3 * jl + c] = (diag && r == c) ? pointVar : 0.0;
}
// ---------------------------------------------------------------------------------------------------------------------------
// Landmark churn on slots (removeOldLandmarks / removeOutliers / addNewLandmarks, VIOFilter.cpp:345-443; which slots is the caller's
// decision).  mark[i] = 1: slot i loses its landmark, 2: slot i receives the landmark with bearing bearings[3 i ..] at `depth` (the
// median scene depth, :358-366), 0: untouched.  k_tl_edit_state: the replicated part -- origin landmark, group element (identity,
// :377-381), constants, base panel columns (zero, :385-386), the slot's flag; newN slots are in use afterwards.  grid = ceil(n / 128).
__global__ void k_tl_edit_state(Glob* g0, Glob* g1, int n, int newN, const int* mark, double depth, int cap, const double* bearings, double* p0,
    double* Q0, double* Q1, double* lmc, double* Sb0, double* Sb1, int ldb, int* active, int* errflag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) g0->N = g1->N = newN;
    if (i >= n) return;
    const int m = mark[i];
    if (m == 0) return;
    int bad = 0;
    d3 p = mk3(1.0, 0.0, 0.0);  // (a hole's origin landmark is never evaluated; any finite point away from the chart pole e3)
    double cst[15];
    if (m == 2) {
        const double* y = bearings + 3 * i;
        p = mk3(y[0] * depth, y[1] * depth, y[2] * depth);
#if defined(__HIP_DEVICE_COMPILE__)
        __asm__ volatile("" : "+v"(p.x), "+v"(p.y), "+v"(p.z));  // (as in k_tl_append)
#endif
        landmarkConstants(p, cst, &bad);
    } else {
        for (int c = 0; c < 15; ++c) cst[c] = (c >= 6 && (c - 6) % 4 == 0) ? 1.0 : 0.0;  // C = 0, R_s = I
    }
    active[i] = m == 2 ? 1 : 0;
    p0[i] = p.x; p0[cap + i] = p.y; p0[2 * cap + i] = p.z;
    for (double* Q : {Q0, Q1}) {
        Q[i] = 1.0; Q[cap + i] = 0.0; Q[2 * cap + i] = 0.0; Q[3 * cap + i] = 0.0; Q[4 * cap + i] = 1.0;
    }
    for (int c = 0; c < 15; ++c) lmc[(long long)c * cap + i] = cst[c];
    for (double* Sb : {Sb0, Sb1})
        for (int r = 0; r < 12; ++r)
            for (int c = 0; c < 3; ++c) Sb[(long long)r * ldb + kLm0 + 3 * i + c] = 0.0;
    if (bad && errflag) atomicOr(errflag, 16);
}
// ... and the rank's share of the landmark x landmark part: every 3 x 3 block in a marked row or column <- 0 (removeRows / removeCols of
// :421-427, resp. the zero off-diagonal blocks of :384-386), the diagonal block of a marked slot <- I (hole) or initialPointVariance I
// (:387-388).  grid = (ceil(nlc / 128), nlr), block = 128.
__global__ void k_tl_edit_local(double* Sll, int ldl, int nlr, int nlc, const int* rowMap, const int* colMap, const int* mark, double pointVar) {
    const int jl = blockIdx.x * blockDim.x + threadIdx.x, il = blockIdx.y;
    if (jl >= nlc || il >= nlr) return;
    const int I = rowMap[il], J = colMap[jl];
    const int mi = mark[I], mj = mark[J];
    if (!(mi | mj)) return;
    const double dv = (I == J) ? (mi == 2 ? pointVar : 1.0) : 0.0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Sll[(long long)(3 * il + r) * ldl + 3 * jl + c] = (r == c) ? dv : 0.0;
}
// constants after a state injection
__global__ void k_tl_restore(Glob* g, const double* p0, double* lmc, int cap, int* active, int* errflag) {
    Glob& s = *g;
    int bad = 0;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = tid; i < s.N; i += gridDim.x * blockDim.x) {
        active[i] = 1;
        double cst[15];
        landmarkConstants(mk3(p0[i], p0[cap + i], p0[2 * cap + i]), cst, &bad);
        for (int c = 0; c < 15; ++c) lmc[(long long)c * cap + i] = cst[c];
    }
    if (tid == 0 && s.initialised) {
        double e0[3], cd[6], ci[6];
        poseConstants(quat{s.P0q[0], s.P0q[1], s.P0q[2], s.P0q[3]}, e0, cd, ci, &bad);
        for (int i = 0; i < 3; ++i) s.eta0[i] = e0[i];
        for (int i = 0; i < 6; ++i) {
            s.cDiff[i] = cd[i];
            s.cInv[i] = ci[i];
        }
    }
    if (bad && errflag) atomicOr(errflag, 32);
}
// stateEstimate = stateGroupAction(X, xi0) (VIOFilter.cpp:304, VIOGroup.cpp:23-45): out = q(4) x(3) v(3) p(3N)
__global__ void k_tl_state_estimate(const Glob* g, const double* p0, const double* Q, int cap, double* out) {
    const Glob& s = *g;
    const se3 P0 = se3{quat{s.P0q[0], s.P0q[1], s.P0q[2], s.P0q[3]}, mk3(s.P0x[0], s.P0x[1], s.P0x[2])};
    const se3 A = se3{quat{s.Aq[0], s.Aq[1], s.Aq[2], s.Aq[3]}, mk3(s.Ax[0], s.Ax[1], s.Ax[2])};
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid == 0) {
        const se3 P = se3mul(P0, A);
        const d3 v = qrot(qinv(A.q), mk3(s.v0[0] - s.w[0], s.v0[1] - s.w[1], s.v0[2] - s.w[2]));
        out[0] = P.q.w; out[1] = P.q.x; out[2] = P.q.y; out[3] = P.q.z;
        out[4] = P.x.x; out[5] = P.x.y; out[6] = P.x.z;
        out[7] = v.x; out[8] = v.y; out[9] = v.z;
    }
    for (int i = tid; i < s.N; i += gridDim.x * blockDim.x) {
        const quat Qq = quat{Q[i], Q[cap + i], Q[2 * cap + i], Q[3 * cap + i]};
        const d3 qh = scl(1.0 / Q[4 * cap + i], qrot(qinv(Qq), mk3(p0[i], p0[cap + i], p0[2 * cap + i])));
        out[10 + 3 * i] = qh.x; out[11 + 3 * i] = qh.y; out[12 + 3 * i] = qh.z;
    }
}

}  // namespace eqf
