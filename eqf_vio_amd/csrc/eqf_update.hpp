// EqF vision update on the device (gfx950): innovation, gain, innovation lift and covariance downdate.
//
// Replaces VIOFilter::processVisionData's numerical core (eqf_vio/src/VIOFilter.cpp:264-297) together
// with measureSystemState (src/VIOState.cpp:58-70), outputGroupAction (src/VIOGroup.cpp:71-90),
// outputCoordinateChart (src/VisionMeasurement.cpp:24-34), EqFOutputMatrixC_euclid_impl
// (src/EqFMatrices.cpp:319-344), bundleLift (:173-252) and liftTotalSpaceInnovationDiscrete (:254-275).
//
// The reference evaluates  S = C Sigma C^T + R,  K = Sigma C^T S^-1 (LU inverse),  Sigma - K C Sigma  and,
// inside bundleLift, an explicit (5+3N)^2 inverse of Sigma[6:,6:].  Here the same quantities come from two
// blocked Cholesky chains with extra right-hand-side columns and NO explicit inverse / back substitution:
//
//   S-chain:  S = L L^T ;  Y = L^-1 [ C Sigma | delta | V ]        (V = C0 Z_P, 6 columns)
//             gamma = K delta = Y^T z  (z = L^-1 delta) ;  Sigma <- Sigma - Y^T Y
//   E-chain:  Sigma_e = Sigma[6:,6:] = Le Le^T ;  [Zt | Et] = Le^-1 [ Z_P | E_top ]
//             Z_P = D * pHatMat * Ad(P0) (6 columns: the weighted-least-squares regressors of bundleLift
//             before the K_par / K_perp split), E_top = first five unit vectors.
//   bundleLift's normal equations then are, with G6 = Zt^T Zt, T65 = Zt^T Et, hV = (L^-1 V)^T z:
//             coeff^T W coeff = Kpar^T G6 Kpar
//             coeff^T W obs   = Kpar^T ( -(hV - T65 gamma_e[0:5]) - G6 DeltaU_fixed )
//   because D*alpha = -gamma_landmarks and Sigma_e^-1 gamma_e = C0^T S^-1 delta (gamma_e = Sigma_e C0^T S^-1 delta).
//
// All factorisation work is fp64 (v_mfma_f64_16x16x4_f64 for the 32x32x32 block products): Sigma has a
// condition number of 1e6..1e8 while landmarks converge, fp32 cannot carry it.
#pragma once
#include <type_traits>

#include "eqf_device.hpp"
#include "eqf_math.hpp"

namespace eqf {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int roundUp(int x, int m) { return (x + m - 1) / m * m; }
// chain dimensions of one filter
__host__ __device__ inline int sDim(int N) { return 2 * N; }              // measurement dim m
__host__ __device__ inline int eDim(int N) { return 6 + 3 * N; }          // internal Sigma_e order (incl. pad at e = 5)
__host__ __device__ inline int yCols(int N) { return kLm0 + 3 * N + 6; }  // C Sigma (delta in col 11) | V

struct UpdArgs {
    Glob* g;             // current scalar state [B] (updated in place by updateFinishBody)
    const double* p0;    // [B][3][cap]
    const double* lmc;   // [B][15][cap] per-landmark constants of the origin landmark: C0i (6), chart rotation R_s (9)
    double* Q;           // [B][5][cap] current group landmarks (updated in place by finish)
    const void* Sin;     // Sigma (T) current
    void* Sout;          // Sigma (T) next
    long long sigmaStride;
    int cap, ld;
    // measurement: bearings[b][k][3], landmark i of the state reads k = perm ? perm[b][i] : i
    const double* bearings;
    long long bearStride;  // doubles between filters
    const int* perm;       // [B][cap] or nullptr
    // S-chain buffers (fp64): work matrix A, factor L, right-hand sides W (work) and Wout (solved)
    double *SA, *SL, *YW, *YO;
    int ldS, ldY;
    long long strideS, strideY;
    // E-chain buffers
    double *EA, *EL, *ZW, *ZO;
    int ldE, ldZ;
    long long strideE, strideZ;
    // outputs for getters / tests
    double* dbgDelta;  // [B][2*cap]
    double* dbgGamma;  // [B][12+3*cap] internal index map
    double* dbgGammaTot;  // [B][9+3*cap]
    double* red;          // [B][256]: hV (6) at 0, G11 = [Zt|Et]^T [Zt|Et] (11x11) at 8
    int* errflag;
    int* resCounters;     // [B][4] work counters of k_chol_resident (zeroed by the prep launch), or nullptr
    int pad;              // chain dimensions are padded (with identity) to multiples of this: 32 (k_chol_step) or 64 (k_chol_step64)
    int csInBurst;        // the landmark columns of C Sigma and S were left by the burst's block kernel (BurstArgs::csOut): the landmark waves
                          // build the 12 leading columns, the residual, V and the padding only
    int eFromSigma;       // split 64-wide chain, fp64: the E-chain's tiles are first read straight from Sigma (EA = Sigma[6:, 6:] is a plain
                          // offset there) -- prep only copies the LAST block row, the one that holds the identity padding
    Params prm;
};

// ------------------------------------------------------------------------------------------------
// Per-landmark pieces shared by prep kernels
// ------------------------------------------------------------------------------------------------
// C0i (2x3) = 1/|q| * stereoSphereChartDiff(y, y) * (I - y y^T), y = q/|q|   (EqFMatrices.cpp:332-339)
EQF_DI void outputBlockC(d3 p0, double* C, int* bad) {
    const double n = nrm3(p0);
    const d3 y = scl(1.0 / n, p0);
    double D[6];
    stereoChartDiff(y, y, D, bad);
    const double s = 1.0 / n;
    const m33 P = add33(eye3(), scl33(-1.0, outer3(y, y)));
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            C[3 * r + c] = (s * D[3 * r]) * P.a[c] + (s * D[3 * r + 1]) * P.a[3 + c] + (s * D[3 * r + 2]) * P.a[6 + c];
}

// Constants of one origin landmark p0 (they change only when the landmark set changes): C0i and the matrix of the
// chart rotation R_s = SO3FromVectors(-y0, e3), y0 = p0/|p0|, that the residual chart uses (VisionMeasurement.cpp:30,
// VIOState.cpp:230-234).  Layout: out[0..5] = C0i row-major, out[6..14] = R_s row-major.
EQF_DI void landmarkConstants(d3 p0, double* out, int* bad) {
    outputBlockC(p0, out, bad);
    const m33 Rs = q2m(sphereRotQ(unit3(p0), bad));
#pragma unroll
    for (int k = 0; k < 9; ++k) out[6 + k] = Rs.a[k];
}

// Rows of Z_P for one landmark: Qhat_i R_C^T [ (x0 - pHat)^x R0 , R0 ]   (3x6)   (EqFMatrices.cpp:221-235)
struct LiftCommon {
    m33 RCt;   // R_C^T, R_C = R_Phat R_IC
    se3 PC;    // xiHat.pose * cameraOffset
    m33 R0;    // xi0.pose rotation
    d3 x0;
};
EQF_DI LiftCommon liftCommon(const Glob& g, const Params& p) {
    LiftCommon L;
    const se3 P0 = se3{quat{g.P0q[0], g.P0q[1], g.P0q[2], g.P0q[3]}, mk3(g.P0x[0], g.P0x[1], g.P0x[2])};
    const se3 A = se3{quat{g.Aq[0], g.Aq[1], g.Aq[2], g.Aq[3]}, mk3(g.Ax[0], g.Ax[1], g.Ax[2])};
    const se3 cam = se3{quat{p.camq[0], p.camq[1], p.camq[2], p.camq[3]}, mk3(p.camx[0], p.camx[1], p.camx[2])};
    const se3 Phat = se3mul(P0, A);  // stateGroupAction: pose * X.A  (VIOGroup.cpp:25)
    L.RCt = q2m(qinv(qmul(Phat.q, cam.q)));
    L.PC = se3mul(Phat, cam);
    L.R0 = q2m(P0.q);
    L.x0 = P0.x;
    return L;
}
EQF_DI void liftRows(const LiftCommon& L, quat Qq, double Qa, d3 p0, double* Z /*[18]*/) {
    const d3 qhat = scl(1.0 / Qa, qrot(qinv(Qq), p0));
    const d3 pHat = se3app(L.PC, qhat);
    const m33 M = mul33(scl33(Qa, q2m(Qq)), L.RCt);  // X.Q[i].asMatrix3d() * R_C^T
    const m33 left = mul33(M, mul33(skew3(sub(L.x0, pHat)), L.R0));
    const m33 right = mul33(M, L.R0);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Z[6 * r + c] = left.a[3 * r + c];
            Z[6 * r + 3 + c] = right.a[3 * r + c];
        }
}

// ------------------------------------------------------------------------------------------------
// k_update_prep: one WAVEFRONT per landmark.  Builds delta_i, C0i, the two rows of C*Sigma (lanes stride the
// Sigma columns: coalesced 3-row reads), the two rows of S and of V; extra workgroups copy Sigma_e.
// grid.x = lmBlocks + eBlocks, grid.y = B, block = 256 (4 waves = 4 landmarks).
// ------------------------------------------------------------------------------------------------
// Dynamic LDS: wpb (waves per workgroup) x 2 rows x nvPad doubles for one column chunk of the C*Sigma rows.
constexpr int kPrepLmChunk = 512;
// WT (k_chol_resident's prep roles: the results are consumed by other workgroups of the SAME launch): every global store is an 8-byte
// agent-scope store (write-through, eqf_handoff.hpp); the caller drains and publishes.  bx / b: the workgroup's index in the prep grid.
template <bool WT>
EQF_DI void prepStore(double* p, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
#else
    *p = v;
#endif
}
template <typename T, bool WT = false>
EQF_DI void updatePrepBody(const UpdArgs& a, int bx, int b, int lmBlocks, int wpb, int nvPad, double* sCSraw) {
    const Glob& g = a.g[b];
    if (!g.updateOk) return;
    const int N = g.N;
    if (N == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cap = a.cap, ld = a.ld;
    const T* Sin = static_cast<const T*>(a.Sin) + (long long)b * a.sigmaStride;
    const double* p0 = a.p0 + (long long)b * 3 * cap;
    const double* Q = a.Q + (long long)b * 5 * cap;
    int bad = 0;

    if (bx >= lmBlocks) {
        // ---- E-chain operand: EA = Sigma[6:,6:] (pad e=5 -> identity), ZW = [Z_P | E_top], 32 rows per workgroup
        const int ne = eDim(N), nep = roundUp(ne, a.pad);
        const int r0 = (bx - lmBlocks) * kNB;
        if (r0 >= nep) return;
        double* EA = a.EA + (long long)b * a.strideE;
        double* ZW = a.ZW + (long long)b * a.strideZ;
        // Four rows per wave and trip, lanes stride the columns, ten column groups (all columns up to N = 211): 40 independent
        // loads in flight.  Sigma was written by the previous launch on other XCDs -- a row-by-row, group-by-group copy pays
        // the ~2 us first-touch miss two dozen times in a row and made these workgroups the longest of the launch.
        const int nw = (int)blockDim.x >> 6;
        constexpr int kRowsTrip = 4, kColGroups = 10;
        // (a.eFromSigma: only the last 64-row block row is copied, 6 MB less traffic per filter and update at N = 200)
        const bool copyRows = a.eFromSigma != 2 && (!a.eFromSigma || r0 + kNB > nep - 64);  // (2: k_chol_resident, OCC2 build, reads every tile from Sigma)
        for (int rb = wv; rb < (copyRows ? kNB : 0); rb += nw * kRowsTrip) {
            for (int c0 = lane; c0 < nep; c0 += 64 * kColGroups) {
                // raw loads first, conversion afterwards: with T = float a convert right behind each load makes hipcc wait
                // for every load separately
                T v[kRowsTrip][kColGroups];
#pragma unroll
                for (int q = 0; q < kRowsTrip; ++q) {
                    const int rr = r0 + rb + q * nw;
#pragma unroll
                    for (int u = 0; u < kColGroups; ++u) {
                        const int c2 = c0 + 64 * u;
                        const bool in = rb + q * nw < kNB && rr < ne && c2 < ne && rr != 5 && c2 != 5;
                        v[q][u] = Sin[in ? (long long)(6 + rr) * ld + 6 + c2 : 0];
                    }
                }
#pragma unroll
                for (int q = 0; q < kRowsTrip; ++q) {
                    const int rr = r0 + rb + q * nw;
#pragma unroll
                    for (int u = 0; u < kColGroups; ++u) {
                        const int c2 = c0 + 64 * u;
                        const bool in = rr < ne && c2 < ne && rr != 5 && c2 != 5;
                        if (rb + q * nw < kNB && c2 < nep) prepStore<WT>(&EA[(long long)rr * a.ldE + c2], in ? (double)v[q][u] : ((rr == c2) ? 1.0 : 0.0));
                    }
                }
            }
        }
        if (tid < kNB) {
            const int rr = r0 + tid;
            double row[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) row[c] = 0.0;
            if (rr >= 6 && rr < ne) {
                const int i = (rr - 6) / 3, comp = (rr - 6) % 3;
                const LiftCommon L = liftCommon(g, a.prm);
                double Z[18];
                liftRows(L, quat{Q[i], Q[cap + i], Q[2 * cap + i], Q[3 * cap + i]}, Q[4 * cap + i],
                    mk3(p0[i], p0[cap + i], p0[2 * cap + i]), Z);
#pragma unroll
                for (int c = 0; c < 6; ++c) row[c] = (comp == 0) ? Z[c] : (comp == 1 ? Z[6 + c] : Z[12 + c]);
            }
            if (rr < 5) row[6 + rr] = 1.0;  // E_top
            for (int c = 0; c < a.ldZ; ++c) prepStore<WT>(&ZW[(long long)rr * a.ldZ + c], (c < 16) ? row[c] : 0.0);
        }
        return;
    }

    // ---- landmark waves.  The two C*Sigma rows of a landmark are staged through LDS in column chunks that cover
    // kPrepLmChunk landmarks each (one chunk up to N = 512; larger N loops), so LDS use does not grow with N.
    double* sCS0 = sCSraw + (long long)(2 * wv) * nvPad;
    double* sCS1 = sCS0 + nvPad;
    const int i = bx * wpb + wv;
    const int m = sDim(N), mp = roundUp(m, a.pad);
    const int yc = yCols(N), ycp = roundUp(yc, a.pad);
    double* SA = a.SA + (long long)b * a.strideS;
    double* YW = a.YW + (long long)b * a.strideY;
    double C[6] = {0, 0, 0, 0, 0, 0};
    double dl[2] = {0, 0};
    double V[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) V[k] = 0.0;
    const bool valid = i < N;
    const double* lmc = a.lmc + (long long)b * 15 * cap;
    const T* s0 = Sin + (long long)(kLm0 + 3 * (valid ? i : 0)) * ld;
    // The landmark's three Sigma rows: the first kPrepGroups * 64 columns (all of them up to N = 209) are requested before
    // anything else -- Sigma was written by the previous launch on other XCDs, every access is a ~2 us miss, and the
    // scalar chain below (residual, chart, lift rows) runs in its shadow.
    constexpr int kPrepGroups = 10;
    T vr[kPrepGroups][3];
    const bool lead = a.csInBurst != 0;  // (only the 12 leading columns)
    {
        const int colHi0 = lead ? kLm0 : kLm0 + 3 * min(N, kPrepLmChunk);
#pragma unroll
        for (int u = 0; u < kPrepGroups; ++u) {
            const int cc = min(lane + 64 * u, colHi0 - 1);
            if (u == 0 || !lead) {
#pragma unroll
                for (int q = 0; q < 3; ++q) vr[u][q] = s0[(long long)q * ld + cc];
            }
        }
    }
    if (valid) {
        const quat Qq = quat{Q[i], Q[cap + i], Q[2 * cap + i], Q[3 * cap + i]};
        const double Qa = Q[4 * cap + i];
        const d3 q0 = mk3(p0[i], p0[cap + i], p0[2 * cap + i]);
        const int k = a.perm ? a.perm[(long long)b * cap + i] : i;
        const double* yb = a.bearings + (long long)b * a.bearStride + 3 * k;
        const d3 y = mk3(yb[0], yb[1], yb[2]);
        // yerr = (X^-1).Q_i.R()^-1 y (outputGroupAction, VIOGroup.cpp:84,130); delta = e3ProjectSphere(R_s yerr)
        const d3 yerr = qrot(qinv(qinv(Qq)), y);
        m33 Rs;
#pragma unroll
        for (int q = 0; q < 6; ++q) C[q] = lmc[(long long)q * cap + i];
#pragma unroll
        for (int q = 0; q < 9; ++q) Rs.a[q] = lmc[(long long)(6 + q) * cap + i];
        const d3 rr = mv33(Rs, yerr);
        dl[0] = rr.x / (1 - rr.z);  // VIOState.cpp:199-204
        dl[1] = rr.y / (1 - rr.z);
        double Z[18];
        liftRows(liftCommon(g, a.prm), Qq, Qa, q0, Z);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) V[6 * r + c] = dot3(C[3 * r], Z[c], C[3 * r + 1], Z[6 + c], C[3 * r + 2], Z[12 + c]);
    }
    for (int q0 = 0; q0 < (lead ? 1 : N); q0 += kPrepLmChunk) {
        const int q1 = min(N, q0 + kPrepLmChunk);
        const int colLo = (q0 == 0) ? 0 : kLm0 + 3 * q0, colHi = lead ? kLm0 : kLm0 + 3 * q1;
        if (valid) {
            // rows 2i, 2i+1 of C*Sigma for this column chunk: lanes stride the columns (coalesced 3-row reads)
            // (4 column groups per trip: 12 independent loads in flight -- Sigma was written by the previous launch on
            // other XCDs, every access is a ~2 us miss)
            for (int col = colLo + lane; col < colHi; col += 64 * kPrepGroups) {
                // raw loads first (clamped column, always in bounds), conversion afterwards -- see above; the very first
                // trip is already in flight
                if (!(q0 == 0 && col == lane)) {
#pragma unroll
                    for (int u = 0; u < kPrepGroups; ++u) {
                        const int cc = min(col + 64 * u, colHi - 1);
#pragma unroll
                        for (int q = 0; q < 3; ++q) vr[u][q] = s0[(long long)q * ld + cc];
                    }
                }
#pragma unroll
                for (int u = 0; u < kPrepGroups; ++u) {
                    const int cc = col + 64 * u;
                    if (cc < colHi) {
                        const double v0 = (double)vr[u][0], v1 = (double)vr[u][1], v2 = (double)vr[u][2];
                        sCS0[cc - colLo] = dot3(C[0], v0, C[1], v1, C[2], v2);
                        sCS1[cc - colLo] = dot3(C[3], v0, C[4], v1, C[5], v2);
                    }
                }
            }
        }
        __syncthreads();
        if (valid) {
            for (int col = colLo + lane; col < colHi; col += 64) {  // right-hand sides: C Sigma, delta in column 11
                const bool isz = (col == 11);
                prepStore<WT>(&YW[(long long)(2 * i) * a.ldY + col], isz ? dl[0] : sCS0[col - colLo]);
                prepStore<WT>(&YW[(long long)(2 * i + 1) * a.ldY + col], isz ? dl[1] : sCS1[col - colLo]);
            }
            // S[2i+r][2j+s] = sum_c CS[r][12+3j+c] C_j[s][c]  (+ measurementVariance on the diagonal)
            for (int j = q0 + lane; j < (lead ? 0 : q1); j += 64) {
                double Cj[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) Cj[q] = lmc[(long long)q * cap + j];
                const double* c0 = &sCS0[kLm0 + 3 * j - colLo];
                const double* c1 = &sCS1[kLm0 + 3 * j - colLo];
                double s00 = dot3(c0[0], Cj[0], c0[1], Cj[1], c0[2], Cj[2]);
                const double s01 = dot3(c0[0], Cj[3], c0[1], Cj[4], c0[2], Cj[5]);
                const double s10 = dot3(c1[0], Cj[0], c1[1], Cj[1], c1[2], Cj[2]);
                double s11 = dot3(c1[0], Cj[3], c1[1], Cj[4], c1[2], Cj[5]);
                if (j == i) {
                    s00 += a.prm.measurementVariance;
                    s11 += a.prm.measurementVariance;
                }
                double* r0p = SA + (long long)(2 * i) * a.ldS + 2 * j;
                prepStore<WT>(r0p, s00); prepStore<WT>(r0p + 1, s01);
                prepStore<WT>(r0p + a.ldS, s10); prepStore<WT>(r0p + a.ldS + 1, s11);
            }
        }
        __syncthreads();
    }
    const int nvv = kLm0 + 3 * N;
    if (valid) {
        for (int col = nvv + lane; col < ycp; col += 64) {  // V (6 columns) and zero padding
            prepStore<WT>(&YW[(long long)(2 * i) * a.ldY + col], (col < nvv + 6) ? V[col - nvv] : 0.0);
            prepStore<WT>(&YW[(long long)(2 * i + 1) * a.ldY + col], (col < nvv + 6) ? V[6 + col - nvv] : 0.0);
        }
        for (int j = N + lane; j < mp / 2; j += 64) {  // padding columns of S
            double* r0p = SA + (long long)(2 * i) * a.ldS + 2 * j;
            prepStore<WT>(r0p, 0.0); prepStore<WT>(r0p + 1, 0.0); prepStore<WT>(r0p + a.ldS, 0.0); prepStore<WT>(r0p + a.ldS + 1, 0.0);
        }
        if (lane == 0 && a.dbgDelta) {
            a.dbgDelta[(long long)b * 2 * cap + 2 * i] = dl[0];
            a.dbgDelta[(long long)b * 2 * cap + 2 * i + 1] = dl[1];
        }
    } else if (2 * i < mp) {
        // padding rows of the S-chain: identity in S, zero right-hand sides
        for (int col = lane; col < ycp; col += 64) {
            prepStore<WT>(&YW[(long long)(2 * i) * a.ldY + col], 0.0);
            prepStore<WT>(&YW[(long long)(2 * i + 1) * a.ldY + col], 0.0);
        }
        for (int col = lane; col < mp; col += 64) {
            prepStore<WT>(&SA[(long long)(2 * i) * a.ldS + col], (col == 2 * i) ? 1.0 : 0.0);
            prepStore<WT>(&SA[(long long)(2 * i + 1) * a.ldS + col], (col == 2 * i + 1) ? 1.0 : 0.0);
        }
    }
    if (bad && a.errflag) atomicOr(a.errflag, 2);
}

// ------------------------------------------------------------------------------------------------
// Small helpers of the factorisation kernels (eqf_chol64.hpp)
// ------------------------------------------------------------------------------------------------
// Broadcast of a double held by lane `src` (compile-time constant after unrolling) through SGPRs.
EQF_DI double readlane64(double v, int src) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
#else
    (void)src;
    return v;
#endif
}
// 1/sqrt(d): hardware v_rsq_f64 seed + two Newton steps (full fp64 accuracy, ~8 dependent instructions).
EQF_DI double rsqrtPivot(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(d);
#else
    double y = 1.0 / sqrt(d);
#endif
    const double h = 0.5 * d;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return y;
}

// a / sqrt(d) for the pivot chains' column scaling, with as few DEPENDENT fp64 operations behind v_rsq_f64 as accuracy allows: the chain of
// a pivot is a lone wave issuing in order, every dependent operation is its full latency.  v_rsq_f64 is good to 2^-24.2
// (scripts/micro/rsq_acc.hip); ONE third-order step y1 = y (1 + e / 2 + 3 e^2 / 8), e = 1 - d y^2, takes that to 2^-72, below the rounding
// of the result, and folded into the product with a it is four dependent operations -- t = d y; e = 1 - t y; {p = 1/2 + 3/8 e, q = (a y) e};
// a y + q p -- against seven for two Newton steps (three each) and the product.  Relative error of the result <= 1.5 ulp, as before.
EQF_DI double scaleRsqrtPivot(double a, double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(d);
#else
    const double y = 1.0 / sqrt(d);
#endif
    const double ay = a * y;
    const double e = fma(-(d * y), y, 1.0);
    const double p = fma(0.375, e, 0.5);
    return fma(ay * e, p, ay);
}

struct ChainArgs {
    Glob* g;
    double *A, *D, *W, *WO;   // work matrix, diagonal factors [nb][2][32][32] (L_kk, inv L_kk), rhs work, rhs solved
    int ldA, ldW;
    long long strideA, strideD, strideW;
    int kind;     // 0: S-chain, 1: E-chain
    int nbMax;    // tiles per edge launched for A
    int wtMax;    // right-hand-side column tiles launched
    // in-launch hand-off of the diagonal-factor records (k_chol_step64<T, 3>, eqf_handoff.hpp): flags[b * strideF + K] == epoch
    // once D[K] of this update has been published
    int* flags;
    long long strideF;
    int epoch;
};

// One of two argument blocks by a run-time flag, FIELD BY FIELD on values.  (`second ? c1 : c0` on the structs -- or on a member
// of them, an lvalue -- is compiled as a select between two ADDRESSES: both structs are copied to scratch memory at kernel
// entry and every later use is a scratch load.)
template <typename V>
EQF_DI V pickValue(bool second, V v0, V v1) {
    return second ? v1 : v0;
}
EQF_DI ChainArgs pickChain(bool second, const ChainArgs& c0, const ChainArgs& c1) {
    ChainArgs ch;
    ch.g = pickValue(second, c0.g, c1.g);
    ch.A = pickValue(second, c0.A, c1.A);
    ch.D = pickValue(second, c0.D, c1.D);
    ch.W = pickValue(second, c0.W, c1.W);
    ch.WO = pickValue(second, c0.WO, c1.WO);
    ch.ldA = pickValue(second, c0.ldA, c1.ldA);
    ch.ldW = pickValue(second, c0.ldW, c1.ldW);
    ch.strideA = pickValue(second, c0.strideA, c1.strideA);
    ch.strideD = pickValue(second, c0.strideD, c1.strideD);
    ch.strideW = pickValue(second, c0.strideW, c1.strideW);
    ch.kind = pickValue(second, c0.kind, c1.kind);
    ch.nbMax = pickValue(second, c0.nbMax, c1.nbMax);
    ch.wtMax = pickValue(second, c0.wtMax, c1.wtMax);
    ch.flags = pickValue(second, c0.flags, c1.flags);
    ch.strideF = pickValue(second, c0.strideF, c1.strideF);
    ch.epoch = pickValue(second, c0.epoch, c1.epoch);
    return ch;
}

// ------------------------------------------------------------------------------------------------
// updateFinishBody (one workgroup per filter; runs inside the last chain launch, or as the extra workgroup of k_downdate):
// gamma = Y^T z, the 4x4 weighted least squares of bundleLift,
// Delta = liftTotalSpaceInnovationDiscrete(Gamma), X <- Delta * X, bias += gamma[0:6].
// ------------------------------------------------------------------------------------------------
EQF_DI void solve4(double M[4][4], double* rhs, double* x) {
    // Gaussian elimination with partial pivoting (the reference uses Householder QR on the same 4x4 system).  Fully
    // unrolled with static indices -- row exchanges are conditional swaps -- so that everything stays in registers (a
    // permutation vector would make M dynamically indexed, i.e. live in scratch memory).
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            const bool sw = fabs(M[i][k]) > fabs(M[k][k]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double u = M[k][j], v = M[i][j];
                M[k][j] = sw ? v : u;
                M[i][j] = sw ? u : v;
            }
            const double u = rhs[k], v = rhs[i];
            rhs[k] = sw ? v : u;
            rhs[i] = sw ? u : v;
        }
        const double inv = 1.0 / M[k][k];
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            const double l = M[i][k] * inv;
#pragma unroll
            for (int j = k; j < 4; ++j) M[i][j] -= l * M[k][j];
            rhs[i] -= l * rhs[k];
        }
    }
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        double sacc = rhs[i];
#pragma unroll
        for (int j = i + 1; j < 4; ++j) sacc -= M[i][j] * x[j];
        x[i] = sacc / M[i][i];
    }
}

// red: hV (6) at 0, G11 = [Zt|Et]^T [Zt|Et] (11 x 11) at 8 (global a.red of filter b, or an LDS copy of it)
// phase 0: everything.  1: only what needs gamma alone -- the per-landmark part of Delta (threads >= 64), and thread 0 touches the scalars
// its serial part will read; 2: only thread 0's serial part (weighted least squares, X <- Delta X, bias).  k_chol_resident runs phase 1
// while the E-chain's last diagonal factor is still being computed and phase 2 behind it.
EQF_DI void updateFinishBody(const UpdArgs& a, int b, const double* red, int phase = 0) {
    Glob& g = a.g[b];
    if (!g.updateOk || g.N == 0) return;
    const int N = g.N, cap = a.cap;
    const int tid = threadIdx.x;
    double* gam = a.dbgGamma + (long long)b * (kLm0 + 3 * cap);  // gamma = Y^T z
    double* gT = a.dbgGammaTot ? a.dbgGammaTot + (long long)b * (9 + 3 * cap) : nullptr;
    int bad = 0;
    // The weighted least squares + the scalar part of X <- Delta X is serial work for one lane; the per-landmark part
    // of Delta only needs gamma, so it runs on the other wavefronts at the same time.
    if (tid == 0 && phase == 1) {
        // (first touch of the lines thread 0 reads in phase 2: ~1 us of misses taken here)
        double sink = g.v0[0] + g.eta0[0] + g.cInv[0] + g.Aq[0] + g.Ax[0] + g.w[0] + g.bias[0] + gam[0] + gam[8];
        if (sink == 1.2345e300 && gT) gT[0] = sink;
    } else if (tid == 0) {
        double G6[36], T65[30], hV[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) hV[i] = red[i];
#pragma unroll
        for (int i = 0; i < 36; ++i) G6[i] = red[8 + 11 * (i / 6) + i % 6];             // Zt^T Zt
#pragma unroll
        for (int i = 0; i < 30; ++i) T65[i] = red[8 + 11 * (i / 5) + 6 + i % 5];        // Zt^T Et
        const d3 v0 = mk3(g.v0[0], g.v0[1], g.v0[2]);
        const d3 gv = mk3(gam[8], gam[9], gam[10]);
        double dU[6] = {0, 0, 0, 0, 0, 0};
        se3 DA;
        d3 Dw;
        if (a.prm.useInnovationLift) {  // bundleLift, EqFMatrices.cpp:173-252
            const d3 eta0 = unit3(mk3(g.eta0[0], g.eta0[1], g.eta0[2]));
            const double gg0 = gam[6], gg1 = gam[7];
            // stereoSphereChartInvDiff(0, eta0) is cached in g.cInv (3x2)
            const d3 t = mk3(g.cInv[0] * gg0 + g.cInv[1] * gg1, g.cInv[2] * gg0 + g.cInv[3] * gg1, g.cInv[4] * gg0 + g.cInv[5] * gg1);
            const d3 dUw = neg(crs(eta0, t));
            const d3 fx = sub(dUw, scl(dot3(eta0, dUw), eta0));  // DeltaU_fixed = K_perp DeltaU = ((I - eta eta^T) dUw ; 0)
            const double dUf[3] = {fx.x, fx.y, fx.z};
            const double eta[3] = {eta0.x, eta0.y, eta0.z};
            // rhs6 = -(hV - T65 gamma_e[0:5]) - G6 [dUf; 0]
            double rhs6[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double sacc = -hV[c];
#pragma unroll
                for (int q = 0; q < 5; ++q) sacc += T65[5 * c + q] * gam[6 + q];
#pragma unroll
                for (int q = 0; q < 3; ++q) sacc -= G6[6 * c + q] * dUf[q];
                rhs6[c] = sacc;
            }
            // normal equations in the K_par basis, columns (eta;0), (0;e1), (0;e2), (0;e3):  M = Kp^T G6 Kp
            double M[4][4], rhs[4], sol[4];
            double Gw_eta[6];  // G6[:, 0:3] eta
#pragma unroll
            for (int p = 0; p < 6; ++p) Gw_eta[p] = G6[6 * p] * eta[0] + G6[6 * p + 1] * eta[1] + G6[6 * p + 2] * eta[2];
            M[0][0] = eta[0] * Gw_eta[0] + eta[1] * Gw_eta[1] + eta[2] * Gw_eta[2];
            rhs[0] = eta[0] * rhs6[0] + eta[1] * rhs6[1] + eta[2] * rhs6[2];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                M[0][1 + j] = eta[0] * G6[3 + j] + eta[1] * G6[6 + 3 + j] + eta[2] * G6[12 + 3 + j];
                M[1 + j][0] = Gw_eta[3 + j];
                rhs[1 + j] = rhs6[3 + j];
#pragma unroll
                for (int i = 0; i < 3; ++i) M[1 + i][1 + j] = G6[6 * (3 + i) + 3 + j];
            }
            solve4(M, rhs, sol);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                dU[i] = dUf[i] + eta[i] * sol[0];
                dU[3 + i] = sol[1 + i];
            }
            DA = se3Exp(mk3(dU[0], dU[1], dU[2]), mk3(dU[3], dU[4], dU[5]));
            if (a.prm.useDiscreteInnovationLift) Dw = sub(v0, qrot(DA.q, add(v0, gv)));  // EqFMatrices.cpp:254-258
            else Dw = sub(neg(gv), crs(mk3(dU[0], dU[1], dU[2]), v0));                  // :69-79
        } else {  // VIOExp(liftInnovation(gamma_e, xi0)), :35-49
            const d3 eta = mk3(g.eta0[0], g.eta0[1], g.eta0[2]);
            const d3 t = mk3(g.cInv[0] * gam[6] + g.cInv[1] * gam[7], g.cInv[2] * gam[6] + g.cInv[3] * gam[7],
                g.cInv[4] * gam[6] + g.cInv[5] * gam[7]);
            const d3 Uw = neg(crs(eta, t));
            DA = se3Exp(Uw, mk3(0, 0, 0));
            Dw = sub(neg(gv), crs(Uw, v0));
        }
        const se3 A = se3{quat{g.Aq[0], g.Aq[1], g.Aq[2], g.Aq[3]}, mk3(g.Ax[0], g.Ax[1], g.Ax[2])};
        const se3 An = se3mul(DA, A);                                        // X = Delta * X  (VIOFilter.cpp:296, VIOGroup.cpp:95)
        const d3 wn = add(Dw, qrot(DA.q, mk3(g.w[0], g.w[1], g.w[2])));       // :96
        g.Aq[0] = An.q.w; g.Aq[1] = An.q.x; g.Aq[2] = An.q.y; g.Aq[3] = An.q.z;
        g.Ax[0] = An.x.x; g.Ax[1] = An.x.y; g.Ax[2] = An.x.z;
        g.w[0] = wn.x; g.w[1] = wn.y; g.w[2] = wn.z;
#pragma unroll
        for (int i = 0; i < 6; ++i) g.bias[i] += gam[i];  // VIOFilter.cpp:295
        if (gT) {
#pragma unroll
            for (int i = 0; i < 6; ++i) gT[i] = dU[i];
            gT[6] = gv.x; gT[7] = gv.y; gT[8] = gv.z;
        }
    } else if (tid >= 64 && phase != 2) {
        // ---- per-landmark part of Delta and Q_i <- Delta_i Q_i   (VIOGroup.cpp:105-107)
        double* Q = a.Q + (long long)b * 5 * cap;
        const double* p0 = a.p0 + (long long)b * 3 * cap;
        const bool discreteLm = a.prm.useInnovationLift && a.prm.useDiscreteInnovationLift;
        for (int i = tid - 64; i < N; i += 192) {
            const d3 qi = mk3(p0[i], p0[cap + i], p0[2 * cap + i]);
            const d3 gq = mk3(gam[kLm0 + 3 * i], gam[kLm0 + 3 * i + 1], gam[kLm0 + 3 * i + 2]);
            quat dq;
            double da;
            if (discreteLm) {  // EqFMatrices.cpp:262-270
                const d3 q1 = add(qi, gq);
                dq = so3FromVectors(q1, qi, &bad);
                da = nrm3(qi) / nrm3(q1);
            } else {  // :84-93 / :54-63 then SOT3Exp
                const double n2 = dot3(qi, qi);
                dq = so3Exp(scl(-1.0 / n2, crs(qi, gq)));
                da = exp(-dot3(qi, gq) / n2);
            }
            const quat Qq = quat{Q[i], Q[cap + i], Q[2 * cap + i], Q[3 * cap + i]};
            const quat Qn = qmul(dq, Qq);
            Q[i] = Qn.w; Q[cap + i] = Qn.x; Q[2 * cap + i] = Qn.y; Q[3 * cap + i] = Qn.z;
            Q[4 * cap + i] = da * Q[4 * cap + i];
            if (gT) {
                gT[9 + 3 * i] = gq.x; gT[10 + 3 * i] = gq.y; gT[11 + 3 * i] = gq.z;
            }
        }
    }
    if (bad && a.errflag) atomicOr(a.errflag, 8);
}

// ------------------------------------------------------------------------------------------------
// k_downdate: Sigma_out = Sigma_in - Y^T Y on the matrix cores.  64x64 output tile per workgroup, each of
// the 4 waves owns a 32x32 quadrant as 2x2 MFMA 16x16 tiles; the operands Y[:, I], Y[:, J] stream from
// L2 (Y is m x n fp64, 2 MB at N = 200).  T = double: v_mfma_f64_16x16x4_f64; T = float:
// v_mfma_f32_16x16x4_f32 on Y rounded to fp32.  Filters whose update was skipped copy Sigma through.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct MfmaT;
template <>
struct MfmaT<double> {
    typedef f64x4 acc_t;
    static EQF_DI acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static EQF_DI int row(int lane, int q) { return (lane >> 4) + 4 * q; }
};
template <>
struct MfmaT<float> {
    typedef f32x4 acc_t;
    static EQF_DI acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static EQF_DI int row(int lane, int q) { return 4 * (lane >> 4) + q; }
};

// Symmetric, LDS-staged version: only tiles with J0 >= I0 are computed (grid.x enumerates the upper triangle of
// tiles), the mirror tile is written from the same accumulators.  Y row chunks of 32 x 64 values per operand are
// staged through LDS with the next chunk's global loads issued before the MFMAs of the current one.
// TS = output tile edge: 64 (each wave a 32x32 quadrant as 2x2 MFMA tiles; best operand reuse, used when there are
// enough tiles to fill the chip) or 32 (each wave one 16x16 MFMA tile; 4x more workgroups for a single small filter).
// One symmetric tile pair of the downdate; lds: 2 * 32 * (TS + 1) elements of T.  All 256 threads.
// DEPTH = chunks in flight ahead of the one the matrix cores work on.  1 where the launch is bound by throughput (the per-column launches of
// a large batch: the other workgroups of the CU cover the wait); 4 inside k_chol_resident, whose downdate tiles are few and LATE -- a tile is
// 13 dependent 2 us fetches of Y written on other XCDs a moment ago, against 0.3 us of MFMAs per chunk.
// Gate (64 x 64 tiles only): gate(C, ti, tj) is called by EVERY WAVE before it asks for the first rows of block row C of Y (64 rows = four
// chunks: exactly the look-ahead) and returns false if those rows will never come.  k_chol_resident passes a wait for the two Y tiles (C, ti),
// (C, tj) of its S-chain: the downdate then runs BEHIND the right-hand-side roles block row by block row instead of starting when the last
// Y tile is out (8 filters of N = 200: the launch ended 21 us after the innovation lift, with 45 us of downdate behind the last Y tile).
// Same chunks in the same order: the same bits.  After a failed gate nothing is written.
struct DdNoGate {
    EQF_DI bool operator()(int, int, int) const { return true; }
};
template <typename T, int TS, int DEPTH = 1, typename Gate = DdNoGate>
EQF_DI void downdateTile(const UpdArgs& a, int nt, int b, int tileIdx, T* lds, Gate gate = Gate()) {
    constexpr int WM = TS / 32;  // MFMA tiles per wave and dimension
    const Glob& g = a.g[b];
    const int N = g.N;
    const int nv = kLm0 + 3 * N;
    // tile pair (ti <= tj) from the linear index over the upper triangle
    int ti = 0, rem = tileIdx;
    while (rem >= nt - ti) {
        rem -= nt - ti;
        ++ti;
    }
    const int tj = ti + rem;
    const int I0 = ti * TS, J0 = tj * TS;
    if (I0 >= nv || J0 >= nv) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ld = a.ld;
    const T* Sin = static_cast<const T*>(a.Sin) + (long long)b * a.sigmaStride;
    T* Sout = static_cast<T*>(a.Sout) + (long long)b * a.sigmaStride;
    if (!g.updateOk || N == 0) {
        for (int e = tid; e < TS * TS; e += 256) {
            const int R = I0 + e / TS, Cc = J0 + e % TS;
            if (R < nv && Cc < nv) {
                Sout[(long long)R * ld + Cc] = Sin[(long long)R * ld + Cc];
                if (ti != tj) Sout[(long long)Cc * ld + R] = Sin[(long long)Cc * ld + R];
            }
        }
        return;
    }
    const int mp = roundUp(sDim(N), a.pad);
    const double* Y = a.YO + (long long)b * a.strideY;
    const int ldY = a.ldY;
    int late = 0;
    constexpr int KC = 32;                 // rows of Y per chunk
    T (*sI)[TS + 1] = reinterpret_cast<T (*)[TS + 1]>(lds);                    // Y[k0 + r][I0 + c]
    T (*sJ)[TS + 1] = reinterpret_cast<T (*)[TS + 1]>(lds + KC * (TS + 1));    // Y[k0 + r][J0 + c]
    const int qi = wv >> 1, qj = wv & 1;
    const int lr = lane & 15, lk = lane >> 4;
    typedef MfmaT<T> MF;
    typename MF::acc_t acc[WM][WM];
#pragma unroll
    for (int u = 0; u < WM; ++u)
#pragma unroll
        for (int v = 0; v < WM; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[u][v][q] = 0;
    // staging assignment: thread -> row tid / 8 (0..31), PT consecutive columns starting at PT * (tid % 8)
    constexpr int PT = TS / 8;
    const int sr = tid >> 3, sc = (tid & 7) * PT;
    if constexpr (TS == 64) {
        // ---- 64 x 64 tiles (round 3): chunks of 16 rows of Y, two LDS buffers (one barrier per chunk), four chunks in flight in
        // registers.  Layout chosen for the LDS, whose pipe a 2 x 2 wave tile keeps busy a quarter of the time at the full MFMA rate:
        // row pitch 80 doubles (= 16 modulo the 32-double bank window), so the two k rows a half-wavefront reads for an MFMA operand fall
        // on disjoint banks (pitch 65 made every operand read a 2-way conflict: SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE), and a
        // half-wavefront stages one whole 64-value row (lane l: columns 2 l, 2 l + 1 -- one 512-byte global read, one conflict-free write).
#ifndef EQF_DD_DEPTH
#define EQF_DD_DEPTH 8  // (two block rows of look-ahead: 4 -> 8 gave +0.5 .. 1 % at every batch size, profiles/r05_downdate_depth.txt)
#endif
        constexpr int KC2 = 16, PITCH = 80, D = std::is_same<Gate, DdNoGate>::value ? 4 : EQF_DD_DEPTH;
        static_assert(2 * 2 * KC2 * PITCH >= TS * (TS + 1), "the epilogue's transposed tile lives in the same LDS");
        const int h = lane >> 5, c2 = 2 * (lane & 31);
        const bool fastI = I0 > 11 && I0 + TS <= nv, fastJ = J0 > 11 && J0 + TS <= nv;
        typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));
        double pa[D][4], pb[D][4];
        auto fetch2 = [&](int d, int chunk) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double* yr = Y + (long long)(chunk * KC2 + 4 * wv + 2 * j + h) * ldY;
                if (fastI) {
                    const f64x2u v = *reinterpret_cast<const f64x2u*>(yr + I0 + c2);
                    pa[d][2 * j] = v.x; pa[d][2 * j + 1] = v.y;
                } else {
                    // column 11 of Y holds z, not C*Sigma: rows/cols 11 of Sigma are structurally zero
                    const int ci = I0 + c2;
                    pa[d][2 * j] = (ci < nv && ci != 11) ? yr[ci] : 0.0;
                    pa[d][2 * j + 1] = (ci + 1 < nv && ci + 1 != 11) ? yr[ci + 1] : 0.0;
                }
                if (fastJ) {
                    const f64x2u v = *reinterpret_cast<const f64x2u*>(yr + J0 + c2);
                    pb[d][2 * j] = v.x; pb[d][2 * j + 1] = v.y;
                } else {
                    const int cj = J0 + c2;
                    pb[d][2 * j] = (cj < nv && cj != 11) ? yr[cj] : 0.0;
                    pb[d][2 * j + 1] = (cj + 1 < nv && cj + 1 != 11) ? yr[cj + 1] : 0.0;
                }
            }
        };
        auto stage2 = [&](int d, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = 4 * wv + 2 * j + h;
                T* da = lds + ((buf * 2 + 0) * KC2 + r) * PITCH + c2;
                T* db = lds + ((buf * 2 + 1) * KC2 + r) * PITCH + c2;
                da[0] = (T)pa[d][2 * j]; da[1] = (T)pa[d][2 * j + 1];
                db[0] = (T)pb[d][2 * j]; db[1] = (T)pb[d][2 * j + 1];
            }
        };
        const int nc = mp / KC2;  // (mp is a multiple of 64)
        static_assert((D * KC2) % 64 == 0, "the look-ahead is whole 64-row block rows of Y: one gate per block row, at its first chunk");
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < nc) {
                if ((d & 3) == 0 && !late && !gate(d >> 2, ti, tj)) late = 1;
                fetch2(d, d);
            }
        stage2(0, 0);
        if (D < nc) {
            if (!late && !gate(D >> 2, ti, tj)) late = 1;
            fetch2(0, D);
        }
        __syncthreads();
        for (int c0 = 0; c0 < nc; c0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int c = c0 + d;
                if (c < nc) {  // (uniform)
                    const int nd = (d + 1) % D;
                    if (c + 1 < nc) {
                        stage2(nd, (c + 1) & 1);
                        if (c + 1 + D < nc) {
                            if (((c + 1 + D) & 3) == 0 && !late && !gate((c + 1 + D) >> 2, ti, tj)) late = 1;  // (wave-uniform)
                            fetch2(nd, c + 1 + D);
                        }
                    }
                    const T* sa = lds + ((c & 1) * 2 + 0) * KC2 * PITCH + 16 * WM * qi + lr;
                    const T* sb = lds + ((c & 1) * 2 + 1) * KC2 * PITCH + 16 * WM * qj + lr;
#pragma unroll
                    for (int s = 0; s < KC2 / 4; ++s) {
                        T av[WM], bv[WM];
#pragma unroll
                        for (int u = 0; u < WM; ++u) {
                            av[u] = sa[(4 * s + lk) * PITCH + 16 * u];
                            bv[u] = sb[(4 * s + lk) * PITCH + 16 * u];
                        }
#pragma unroll
                        for (int u = 0; u < WM; ++u)
#pragma unroll
                            for (int v = 0; v < WM; ++v) acc[u][v] = MF::mfma(av[u], bv[v], acc[u][v]);
                    }
                    __syncthreads();
                }
            }
        }
    } else {
        double pi[DEPTH][PT], pj[DEPTH][PT];
        auto fetch = [&](int d, int k0) __attribute__((always_inline)) {
            const double* yr = Y + (long long)(k0 + sr) * ldY;
    #pragma unroll
            for (int q = 0; q < PT; ++q) {
                const int ci = I0 + sc + q, cj = J0 + sc + q;
                // column 11 of Y holds z, not C*Sigma: rows/cols 11 of Sigma are structurally zero
                pi[d][q] = (ci < nv && ci != 11) ? yr[ci] : 0.0;
                pj[d][q] = (cj < nv && cj != 11) ? yr[cj] : 0.0;
            }
        };
    #pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d * KC < mp) fetch(d, d * KC);
        for (int k0 = 0; k0 < mp; k0 += KC * DEPTH) {
    #pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int kk = k0 + d * KC;
                if (kk < mp) {  // (uniform)
                    __syncthreads();
    #pragma unroll
                    for (int q = 0; q < PT; ++q) {
                        sI[sr][sc + q] = (T)pi[d][q];
                        sJ[sr][sc + q] = (T)pj[d][q];
                    }
                    __syncthreads();
                    if (kk + KC * DEPTH < mp) fetch(d, kk + KC * DEPTH);
    #pragma unroll
                    for (int s = 0; s < KC / 4; ++s) {
                        T av[WM], bv[WM];
    #pragma unroll
                        for (int u = 0; u < WM; ++u) {
                            av[u] = sI[4 * s + lk][16 * WM * qi + 16 * u + lr];
                            bv[u] = sJ[4 * s + lk][16 * WM * qj + 16 * u + lr];
                        }
    #pragma unroll
                        for (int u = 0; u < WM; ++u)
    #pragma unroll
                            for (int v = 0; v < WM; ++v) acc[u][v] = MF::mfma(av[u], bv[v], acc[u][v]);
                    }
                }
            }
        }
    }
    if constexpr (!std::is_same<Gate, DdNoGate>::value) {
        if (__syncthreads_or(late)) return;  // (a gate failed: Sigma_out keeps what it holds)
    }
    // ---- epilogue.  Every read of Sigma_in is issued before the first write of Sigma_out (a load the compiler cannot prove
    // independent of the previous store waits for that store's acknowledgement: 32 dependent round trips per thread), and
    // the mirror tile goes through LDS so that its rows are read and written as rows (512 contiguous bytes per wavefront
    // instead of 64 lines).
    T sv[WM][WM][4];
#pragma unroll
    for (int u = 0; u < WM; ++u)
#pragma unroll
        for (int v = 0; v < WM; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int R = I0 + 16 * WM * qi + 16 * u + MF::row(lane, q);
                const int Cc = J0 + 16 * WM * qj + 16 * v + lr;
                sv[u][v][q] = (R < nv && Cc < nv) ? Sin[(long long)R * ld + Cc] : (T)0;
            }
    constexpr int MT = TS * TS / 256;  // mirror-tile elements per thread
    T mv[MT];
    if (ti != tj) {
#pragma unroll
        for (int w = 0; w < MT; ++w) {
            const int e = tid + 256 * w, mr = J0 + e / TS, mc = I0 + e % TS;  // mirror element (row in J, column in I)
            mv[w] = (mr < nv && mc < nv) ? Sin[(long long)mr * ld + mc] : (T)0;
        }
        __syncthreads();  // (the last chunk's operands have been read)
        T (*sM)[TS + 1] = reinterpret_cast<T (*)[TS + 1]>(lds);  // sM[c][r] = acc(r, c)   (TS x (TS + 1) <= 2 * KC * (TS + 1))
#pragma unroll
        for (int u = 0; u < WM; ++u)
#pragma unroll
            for (int v = 0; v < WM; ++v)
#pragma unroll
                for (int q = 0; q < 4; ++q) sM[16 * WM * qj + 16 * v + lr][16 * WM * qi + 16 * u + MF::row(lane, q)] = acc[u][v][q];
    }
#pragma unroll
    for (int u = 0; u < WM; ++u)
#pragma unroll
        for (int v = 0; v < WM; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int R = I0 + 16 * WM * qi + 16 * u + MF::row(lane, q);
                const int Cc = J0 + 16 * WM * qj + 16 * v + lr;
                if (R < nv && Cc < nv) Sout[(long long)R * ld + Cc] = sv[u][v][q] - acc[u][v][q];
            }
    if (ti != tj) {
        __syncthreads();
        const T (*sM)[TS + 1] = reinterpret_cast<const T (*)[TS + 1]>(lds);
#pragma unroll
        for (int w = 0; w < MT; ++w) {
            const int e = tid + 256 * w, mr = J0 + e / TS, mc = I0 + e % TS;
            if (mr < nv && mc < nv) Sout[(long long)mr * ld + mc] = mv[w] - sM[e / TS][e % TS];
        }
    }
}

// withFinish = 0: no innovation-lift workgroup (it ran in the E-chain's last launch).
template <typename T, int TS>
__global__ __launch_bounds__(256) void k_downdate(UpdArgs a, int nt, int withFinish = 1) {
    const int b = blockIdx.y;
    if (withFinish && blockIdx.x == gridDim.x - 1) {
        // the innovation lift / X <- Delta X / bias update is independent of the downdate: one extra workgroup of this
        // launch does it (saves a kernel boundary; both only need gamma and the reduced products)
        updateFinishBody(a, b, a.red + (long long)b * 256);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char sBufDd[];
    downdateTile<T, TS>(a, nt, b, (int)blockIdx.x, reinterpret_cast<T*>(sBufDd));
}
template <typename T, int TS>
constexpr int downdateLdsBytes() { return int(sizeof(T)) * (TS == 64 ? 2 * 2 * 16 * 80 : 2 * 32 * (TS + 1)); }

}  // namespace eqf
