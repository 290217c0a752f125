// k_chol_step64: the two Cholesky chains of the vision update in 64-wide block columns (gfx950, fp64).
//
// Latency variant of k_chol_step (eqf_update.hpp) for ONE small filter, where a frame is bound by the number of
// dependent launches and by the serial pivot chain, not by flops: half as many launches (n/64 instead of n/32), and the
// serial part of a launch -- factoring the next 64x64 diagonal block -- is organised so that nothing is solved twice:
//
//   diagonal block  factored in four 16-column stages: one wave holds the 64 rows of the panel in registers (lane =
//                   row), so the right-looking 16-column potrf scales / updates the rows below the diagonal block in
//                   the same instruction stream (L_rj = A_rj L_jj^-T for free) -- and 16 otherwise idle lanes carry the
//                   rows of the identity, which leave as W_jj^T = L_jj^-T, the inverse the other workgroups need.  The
//                   16x16 tiles of the next column are updated on the matrix cores between two stages; all other
//                   trailing tiles are updated by the idle waves WHILE the next 16 columns are being factored.
//   panel blocks    X = A_RK L_KK^-T (and Y_K = L_KK^-1 R_K) as a 4-stage blocked substitution made of 16x16x16
//                   products only, in the transposed form X_j^T = W_jj (A_j^T - sum_{i<j} L_ji X_i^T): the accumulator
//                   layout of v_mfma_f64_16x16x4_f64 (row = (lane>>4) + 4 reg, col = lane & 15) IS its B-operand layout
//                   for the matrix held in the accumulator, so each result feeds the next product straight from
//                   registers; only L / W blocks are read from LDS.  Every wave owns a 16-row (16-column) strip, no
//                   barriers, no long unrolled scalar code.  (A launch executes its code exactly once, so straight-line
//                   scalar code runs at instruction-fetch / LDS-latency speed: an unrolled 32-column forward
//                   substitution measured 27 cycles per FMA.)
//                   Every workgroup solves the two panel blocks its tile needs while the diagonal workgroup factors
//                   the next block (one-step look-ahead as in k_chol_step).
//
// Same mathematics as k_chol_step (see the header of eqf_update.hpp: S-chain Y = L^-1 [C Sigma | delta | V], E-chain
// [Zt | Et] = Le^-1 [Z_P | E_top], replacing S.inverse() / Sigma_e.inverse() of VIOFilter.cpp:276-277 and
// EqFMatrices.cpp:239); chain dimensions are padded to multiples of 64 with identity (UpdArgs::pad = 64).
#pragma once
#include "eqf_handoff.hpp"
#include "eqf_update.hpp"

namespace eqf {

constexpr int kSB = 64;        // block-column width
constexpr int kSP = kSB + 1;   // LDS pitch of a 64x64 tile  (kSB + 2 would make the MFMA operand reads conflict-free: measured, one filter +3 us -- the column accesses of the pivot chain want the odd pitch -- batches within noise)
constexpr int kQB = 16;        // sub-block (one MFMA tile)
constexpr int kWP = kQB + 1;   // LDS pitch of a 16x16 inverse block
constexpr int kDRec = kSB * kSB + 4 * kQB * kQB;  // doubles per diagonal-factor record in ChainArgs::D:
// [0, 4096) L_KK row-major 64x64 (lower block triangle) ; [4096, 5120) W_jj = L_jj^-1, j = 0..3, row-major 16x16

struct Step64Lds {
    double P[kSB][kSP];   // A_RK -> L_RK (A tiles) / R_K -> Y_K (rhs tiles)
    double Q[kSB][kSP];   // A_CK -> L_CK
    double L[kSB][kSP];   // L_KK ; the diagonal workgroup factors the next block in here
    double Wd[4][kQB][kWP];
    double D0[kQB][kWP];  // copy of the leading 16x16 block of the matrix being factored (read by the identity-row wave)
    double Zs[kSB][kWP];  // rhs workgroups of the S-chain: the innovation column z_K = L_KK^-1 delta_K (column 0)
    double redL[256];     // last E-chain rhs workgroup: the reduced products handed to the innovation lift
};

// Pointer view of the LDS workspace.  Two layouts:
//   full   (fused / panel launches): Step64Lds as is                                         (119 KB, 1 workgroup / CU)
//   update (update-only launches of the split chain): Q | P (the diagonal workgroup factors in it: L aliases P) | Wd | D0
//                                                                                             (77 KB, 2 workgroups / CU)
struct Lds64 {
    double (*P)[kSP];
    double (*Q)[kSP];
    double (*L)[kSP];
    double (*Wd)[kQB][kWP];
    double (*D0)[kWP];
    double (*Zs)[kWP];
    double* redL;
    double* zv;  // tail layout only: [0, 64) the innovation column of this block row, [64, 128) z of the previous block row
};
constexpr int kLdsUpdateBytes = int(sizeof(double)) * (2 * kSB * kSP + 4 * kQB * kWP + kQB * kWP);
EQF_DI Lds64 ldsFull(unsigned char* smem) {
    Step64Lds* f = reinterpret_cast<Step64Lds*>(smem);
    return Lds64{f->P, f->Q, f->L, f->Wd, f->D0, f->Zs, f->redL, nullptr};
}
// what factor64 alone needs (the first diagonal blocks, factored inside the prep launch): L | Wd | D0   (44 KB)
constexpr int kLdsFactorBytes = int(sizeof(double)) * (kSB * kSP + 4 * kQB * kWP + kQB * kWP);
EQF_DI Lds64 ldsFactor(unsigned char* smem) {
    double* d = reinterpret_cast<double*>(smem);
    double (*L)[kSP] = reinterpret_cast<double (*)[kSP]>(d);
    double (*Wd)[kQB][kWP] = reinterpret_cast<double (*)[kQB][kWP]>(d + kSB * kSP);
    double (*D0)[kWP] = reinterpret_cast<double (*)[kWP]>(d + kSB * kSP + 4 * kQB * kWP);
    return Lds64{nullptr, nullptr, L, Wd, D0, nullptr, nullptr, nullptr};
}
EQF_DI Lds64 ldsUpdate(unsigned char* smem) {
    double* d = reinterpret_cast<double*>(smem);
    double (*Q)[kSP] = reinterpret_cast<double (*)[kSP]>(d);
    double (*P)[kSP] = reinterpret_cast<double (*)[kSP]>(d + kSB * kSP);
    double (*Wd)[kQB][kWP] = reinterpret_cast<double (*)[kQB][kWP]>(d + 2 * kSB * kSP);
    double (*D0)[kWP] = reinterpret_cast<double (*)[kWP]>(d + 2 * kSB * kSP + 4 * kQB * kWP);
    return Lds64{P, Q, P, Wd, D0, nullptr, nullptr, nullptr};
}
// tail (update launches that also solve the next block column, PHASE 3): the update layout + zv + redL  (80.5 KB, 2 / CU)
constexpr int kLdsTailBytes = kLdsUpdateBytes + int(sizeof(double)) * (128 + 256);
EQF_DI Lds64 ldsTail(unsigned char* smem) {
    Lds64 s = ldsUpdate(smem);
    double* d = reinterpret_cast<double*>(smem) + kLdsUpdateBytes / int(sizeof(double));
    s.zv = d;
    s.redL = d + 128;
    return s;
}

// k_chol_resident on a grid larger than the chip: Q | P | Wd | D0 | redL with L ALIASING Q (78 KB: two workgroups per CU).  Valid there
// because every role is done with its panel operand Q before it touches L: the record of D[K] (hoLoadRecord / the staged loads) arrives
// after the panel loop, and a row head factors its diagonal tile in L while `pre` still reads P.
constexpr int kLdsRes2Bytes = int(sizeof(double)) * (2 * kSB * kSP + 4 * kQB * kWP + kQB * kWP + 256);
EQF_DI Lds64 ldsRes2(unsigned char* smem) {
    double* d = reinterpret_cast<double*>(smem);
    double (*Q)[kSP] = reinterpret_cast<double (*)[kSP]>(d);
    double (*P)[kSP] = reinterpret_cast<double (*)[kSP]>(d + kSB * kSP);
    double (*Wd)[kQB][kWP] = reinterpret_cast<double (*)[kQB][kWP]>(d + 2 * kSB * kSP);
    double (*D0)[kWP] = reinterpret_cast<double (*)[kWP]>(d + 2 * kSB * kSP + 4 * kQB * kWP);
    return Lds64{P, Q, Q, Wd, D0, nullptr, d + 2 * kSB * kSP + 4 * kQB * kWP + kQB * kWP, nullptr};
}

EQF_DI void chainDims64(const ChainArgs& ch, int N, int* nb, int* wt) {
    if (ch.kind == 0) {
        *nb = roundUp(sDim(N), kSB) / kSB;
        *wt = roundUp(yCols(N), kSB) / kSB;
    } else {
        *nb = roundUp(eDim(N), kSB) / kSB;
        *wt = 1;
    }
}

// Workgroups one chain needs in launch K (host and device use the same count): the index space only covers what is left
// of the matrix -- block rows / columns K+1.. -- so late launches do not spawn thousands of workgroups that exit at once.
//   phase 0 (fused):  rem x rem A tiles (lower triangle used) + wt x (rem + 1) rhs tiles (column K solves, K+1.. update)
//   phase 1 (panel):  rem A blocks (R, K)                    + wt rhs tiles (column K)
//   phase 2 (update): rem x rem A tiles                      + wt x rem rhs tiles
//   phase 3 (update + solve of block column K+1 in the same launch): as phase 2
__host__ __device__ inline int chainBlocks64(int nbMax, int wtMax, int K, int phase) {
    const int rem = nbMax - K - 1;
    if (rem < 0) return 0;
    if (phase == 1) return rem + wtMax;
    return rem * rem + wtMax * (phase == 0 ? rem + 1 : rem);
}

// One 16x16 output tile on v_mfma_f64_16x16x4_f64:  acc += sgn * sum_{k<KD} A(i0+i, k) * B(k, j0+j).
//   A(i,k) = Am[i * lda + k] ;  B(k,j) = TB ? Bm[j * ldb + k] : Bm[k * ldb + j]
template <bool TB, int KD>
EQF_DI f64x4 mmTile(f64x4 acc, const double* Am, int lda, int i0, const double* Bm, int ldb, int j0, int lane, double sgn) {
    const int lr = lane & 15, lk = lane >> 4;
    double av[KD / 4], bv[KD / 4];
#pragma unroll
    for (int s = 0; s < KD / 4; ++s) {  // all operand reads first: one LDS latency per tile, not one per MFMA
        av[s] = Am[(i0 + lr) * lda + 4 * s + lk];
        bv[s] = TB ? Bm[(j0 + lr) * ldb + 4 * s + lk] : Bm[(4 * s + lk) * ldb + j0 + lr];
    }
#pragma unroll
    for (int s = 0; s < KD / 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sgn * av[s], bv[s], acc, 0, 0, 0);
    return acc;
}
// acc += sgn * A * Bm with the 16x16 matrix Bm held in ANOTHER accumulator tile: register s of the accumulator layout
// (element [(lane>>4) + 4s][lane & 15]) is exactly the B operand of k-step s (element [4s + (lane>>4)][lane & 15]).
//   A(i,k) = Am[(i0 + i) * lda + k0 + k], 16 x 16
EQF_DI f64x4 mmRegB(f64x4 acc, const double* Am, int lda, int i0, int k0, const f64x4& Breg, int lane, double sgn) {
    const int lr = lane & 15, lk = lane >> 4;
    double av[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) av[s] = Am[(i0 + lr) * lda + k0 + 4 * s + lk];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sgn * av[s], Breg[s], acc, 0, 0, 0);
    return acc;
}
// 16x16 tile in the accumulator layout (row = (lane >> 4) + 4 * reg, col = lane & 15) <-> memory
EQF_DI f64x4 ldTile(const double* M, int ld, int r0, int c0, int lane) {
    f64x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = M[(r0 + (lane >> 4) + 4 * q) * ld + c0 + (lane & 15)];
    return v;
}
EQF_DI void stTile(const f64x4& v, double* M, int ld, int r0, int c0, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) M[(r0 + (lane >> 4) + 4 * q) * ld + c0 + (lane & 15)] = v[q];
}

#ifdef EQF_STEP64_STAMPS
__device__ long long g_stamps[64][16];
#define EQF_STAMP(i) do { if (diagNext && tid == 0 && !second) g_stamps[K][i] = __builtin_readcyclecounter(); } while (0)
#define EQF_FSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.y == 0) g_stamps[40][i] = __builtin_readcyclecounter(); } while (0)
#else
#define EQF_STAMP(i) do { } while (0)
#define EQF_FSTAMP(i) do { } while (0)
#endif
// -DEQF_F64_STAMPS: shader-clock stamps of every wave inside factor64 (scripts/micro/factor64_bench.hip, scripts/res_stamps.py):
// [wave][stage][0 stage start, 1 phase P done, 2 barrier A passed, 3 phase U done, 4 barrier B passed].  Kept in LDS (a global store per
// stamp would sit in front of every __syncthreads' vmcnt(0)) and copied out at the end of the factorisation.
#ifdef EQF_F64_STAMPS
__shared__ long long sF64Stamps[4][4][8];
#define EQF_F64STAMP(j, k) do { if ((threadIdx.x & 63) == 0) sF64Stamps[threadIdx.x >> 6][j][k] = __builtin_readcyclecounter(); } while (0)
#else
#define EQF_F64STAMP(j, k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------------------------
// factor64, round 4.  Shader-clock stamps of every wave (scripts/micro/factor64_bench.hip) showed that of the 5.8 k cycles a 16-column
// stage of the round-3 version cost (8.66 us per block, removed late in round 4), the sixteen pivots were 2.7 k: the rest was the
// INSTRUCTION COUNT of the lone pivot wave around them -- a wave on its own issues one instruction per ~5 cycles -- namely 0.95 k cycles
// to load its sixteen columns (exec-masked 8-byte loads, selects for the identity / idle lanes, SGPR spills), 0.9-1.1 k to write them back
// (a select pair + store per value, the transposed scatter of the inverse block), and 1.0 k for two barriers with a tile update of its own
// in between.  Now the pivot wave does nothing but: 8 ds_read2_b64 from a per-lane row address (real rows in s.L, the identity rows in
// s.Wd[j], which is preset to I and receives L_jj^-T in the same eight ds_write2_b64 as the real rows), sixteen pivots, 8 ds_write2_b64,
// two barriers.  Everything else moved to the other waves:
//   waves 2, 3  own the six 16 x 16 tiles (r, c), 1 <= c <= r, IN REGISTERS from `pre` to their last update (v1: LDS round trip per update):
//               after barrier A of stage j the tiles of column j+1 get update j and go to LDS (the only thing between two pivot chains),
//               the others get it during the next stage's pivots;
//   wave 1      carries the identity rows of stage 0 (as before) and from then on does the chores one stage behind the pivots: W_jj back to
//               row-major, zeros above the diagonal of L_jj, columns j of the record to global memory (16-byte stores), drain, stage flag.
// Same operations on every element in the same order as the round-3 version: bitwise the same L and W (checked hash against hash).
// Pre-condition (factorPrologue): s.D0 = leading 16 x 16 block of s.L, s.Wd[0] = I; post-condition: s.L lower block triangle, s.Wd row-major,
// all LDS writes visible (ends with a barrier).
EQF_DI void factorPrologue(const Lds64& s, int tid) {
    if (tid < kQB * kQB) {
        s.D0[tid >> 4][tid & 15] = s.L[tid >> 4][tid & 15];
        s.Wd[0][tid >> 4][tid & 15] = ((tid >> 4) == (tid & 15)) ? 1.0 : 0.0;
    }
}
// (the same for callers whose wave 0 writes s.D0 from registers: s.Wd[0] = I by the 64 lanes of another wave)
EQF_DI void factorPrologueW(const Lds64& s, int lane) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s.Wd[0][(lane >> 4) + 4 * k][lane & 15] = ((lane >> 4) + 4 * k == (lane & 15)) ? 1.0 : 0.0;
}
#if defined(__HIP_DEVICE_COMPILE__)
#define EQF_LDS_BARRIER() __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define EQF_LDS_BARRIER() do { } while (0)
#endif
// The pivot wave's stage: sixteen values from rowPtr (8-byte aligned LDS address of this lane's row: a real row, an identity row, or any
// harmless row for an idle lane), the right-looking 16-column factorisation with the pivots in lanes 0..15 (see potrf16), sixteen values
// back to the same address (lanes with `store`).  scatterW: lanes 48.. instead scatter their row as a COLUMN of Wj (row-major inverse).
// (Round 4, measured with scripts/micro/factor64_bench.hip and not kept: the next pivot formed as a wave-uniform d' = a - l^2 from two
// broadcasts, one of them off the dependent chain -- one v_readlane round trip less per pivot, bitwise the same value: 15.67 k -> 16.03 k
// cycles per block, three more instructions per pivot cost more than the shorter chain gains; the multipliers of the rank-1 updates from
// LDS (one 8-byte store per pivot, one ds_read2_b64 per two multipliers instead of four v_readlane_b32): 160 instructions fewer per stage and
// 16.9 k cycles -- the waits for LDS sit in front of the chain's instructions, a lone wave issues in order.  v_rsq_f64 is good to 2^-24.2,
// one Newton step leaves 4.2e-15 = 38 ulp, two 2.4e-16 (scripts/micro/rsq_acc.hip): both steps stay.)
EQF_DI void potrf16v2(double* rowPtr, bool store, bool scatterW, double (*Wj)[kWP], int lane, int* bad, int stampStage = -1) {
    double row[kQB];
#pragma unroll
    for (int c = 0; c < kQB; ++c) row[c] = rowPtr[c];
    double ljPrev = 0.0;
    if (stampStage >= 0) EQF_F64STAMP(stampStage, 5);
#pragma unroll
    for (int c = 0; c < kQB; ++c) {
        const double d = readlane64(row[c], c);
        if (c >= 1) {
            // all broadcasts of the column first (distinct SGPR pairs, pinned), then the FMAs: one v_readlane -> VALU hazard per column
            double bc[kQB];
#pragma unroll
            for (int c2 = c + 1; c2 < kQB; ++c2) bc[c2] = readlane64(ljPrev, c2);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int c2 = c + 1; c2 < kQB; ++c2) __asm__ volatile("" : "+s"(bc[c2]));
#endif
#pragma unroll
            for (int c2 = c + 1; c2 < kQB; ++c2) row[c2] = fma(-ljPrev, bc[c2], row[c2]);
        }
        const double lj = scaleRsqrtPivot(row[c], d);
        if (c + 1 < kQB) row[c + 1] = fma(-lj, readlane64(lj, c + 1), row[c + 1]);
        row[c] = lj;
        ljPrev = lj;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int c2 = c + 1; c2 < kQB; ++c2) __asm__ volatile("" : "+v"(row[c2]));
#endif
    }
    // a pivot that is not positive (or not a number) leaves NaNs in every later diagonal entry -- rsq of a negative number is NaN, of
    // zero infinite, and 0 * inf, x - inf * inf end as NaN / -inf under the next square root -- so ONE test of the last diagonal entry
    // replaces sixteen (which cost the lone pivot wave 32 instructions per stage)
    if (!(readlane64(row[kQB - 1], kQB - 1) > 0.0)) *bad = 1;
    if (stampStage >= 0) EQF_F64STAMP(stampStage, 6);
    if (scatterW) {
        if (store && lane < 48) {
#pragma unroll
            for (int c = 0; c < kQB; ++c) rowPtr[c] = row[c];
        }
        if (lane >= 48) {
#pragma unroll
            for (int c = 0; c < kQB; ++c) Wj[c][lane - 48] = row[c];
        }
    } else if (store) {
#pragma unroll
        for (int c = 0; c < kQB; ++c) rowPtr[c] = row[c];
    }
}
// 16 x 16 block of LDS transposed in place by ONE wave (all reads before the first write)
EQF_DI void transpose16(double (*Wj)[kWP], int lane) {
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = Wj[(lane + 64 * k) >> 4][lane & 15];
#pragma unroll
    for (int k = 0; k < 4; ++k) Wj[lane & 15][(lane + 64 * k) >> 4] = v[k];
}
// zeros above the diagonal of the 16 x 16 block at (base, base) of s.L, one wave, two unmasked stores per lane: the 120 entries in eight
// groups of sixteen -- row g (15 - g entries) together with row 14 - g (g + 1 entries); row 7 pairs with itself (eight entries written twice)
EQF_DI void zeroUpper16(const Lds64& s, int base, int lane) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int g = (lane >> 4) + 4 * k, idx = lane & 15;
        const bool first = idx < 15 - g;
        const int r = first ? g : 14 - g, c = first ? g + 1 + idx : 15 - (idx - (15 - g));
        s.L[base + r][base + c] = 0.0;
    }
}
// Columns [16 j, 16 j + 16) of s.L and W_jj (row-major in s.Wd[j]) -> record Dn, by ONE wave: ten 16-byte stores per lane, every LDS read
// issued before the first store.  (The entries above the diagonal of L_jj are whatever LDS holds: zeroUpper16 comes first.)
template <bool WT>
EQF_DI void storeStageWave(const Lds64& s, double* Dn, int j, int lane) {
    double v[10][2];
    const int r0 = lane >> 3, c0 = kQB * j + 2 * (lane & 7);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        v[u][0] = s.L[r0 + 8 * u][c0];
        v[u][1] = s.L[r0 + 8 * u][c0 + 1];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = 2 * (lane + 64 * u);
        v[8 + u][0] = s.Wd[j][e >> 4][e & 15];
        v[8 + u][1] = s.Wd[j][e >> 4][(e & 15) + 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        double* p = Dn + (r0 + 8 * u) * kSB + c0;
        if (WT) hoStore16(p, v[u][0], v[u][1]);
        else { p[0] = v[u][0]; p[1] = v[u][1]; }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        double* p = Dn + kSB * kSB + kQB * kQB * j + 2 * (lane + 64 * u);
        if (WT) hoStore16(p, v[8 + u][0], v[8 + u][1]);
        else { p[0] = v[8 + u][0]; p[1] = v[8 + u][1]; }
    }
}
// The same by all 256 threads (the last stage and the identity padding behind it), with the zeros above the diagonal of L_jj supplied
// here -- wave 1 may be zeroing them in LDS at the same time
template <bool WT>
EQF_DI void storeDiagColumns16(const Lds64& s, double* Dn, int j, int t, int nthr) {
    for (int ch = t; ch < kSB * 8; ch += nthr) {
        const int r = ch >> 3, c = kQB * j + 2 * (ch & 7);
        const bool dg = (r >> 4) == j;
        const double v0 = (dg && c > r) ? 0.0 : s.L[r][c], v1 = (dg && c + 1 > r) ? 0.0 : s.L[r][c + 1];
        double* p = Dn + r * kSB + c;
        if (WT) hoStore16(p, v0, v1);
        else { p[0] = v0; p[1] = v1; }
    }
    for (int ch = t; ch < kQB * 8; ch += nthr) {
        const int e = 2 * ch;
        double* p = Dn + kSB * kSB + kQB * kQB * j + e;
        if (WT) hoStore16(p, s.Wd[j][e >> 4][e & 15], s.Wd[j][e >> 4][(e & 15) + 1]);
        else { p[0] = s.Wd[j][e >> 4][e & 15]; p[1] = s.Wd[j][e >> 4][(e & 15) + 1]; }
    }
}
// The tile work of wave W = 2, 3 (compile-time tile tables, stages unrolled: every LDS address is a constant plus the lane's offset).
//   slot: tile   wave 2: (1,1) (3,1) (3,2)   wave 3: (2,1) (2,2) (3,3)
template <int W, bool WT, typename Pre>
EQF_DI void factorTiles(const Lds64& s, int lane, Pre pre, int nStages) {
    constexpr int TR[3] = {W == 2 ? 1 : 2, W == 2 ? 3 : 2, 3}, TC[3] = {1, W == 2 ? 1 : 2, W == 2 ? 2 : 3};
    double* const L0 = &s.L[0][0];
    // Wd[1..3] = I: the identity rows the pivot wave picks up in stage j >= 1 (and W_jj of the chain's identity padding, j >= nStages)
    for (int e = lane + 64 * (W - 2); e < 3 * kQB * kQB; e += 128) s.Wd[1 + (e >> 8)][(e >> 4) & 15][e & 15] = (((e >> 4) & 15) == (e & 15)) ? 1.0 : 0.0;
    // `pre(wave, r, c, acc)` delivers tile (r, c): two tiles per wave now -- all of column 1, which the first barrier is followed by, and
    // (2,2) -- the third in the shadow of the SECOND sixteen pivots (all six at once took waves 2, 3 longer than the first sixteen pivots:
    // 5.0 k against 3.5 k cycles; the column-1 tiles alone left wave 3 with 5.1 k cycles in the second stage)
    f64x4 acc[3];
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (TC[k] < nStages) pre(W, TR[k], TC[k], acc[k]);
    if (WT) hoDrain();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < nStages) {
            if (j == 1 && TC[2] < nStages) pre(W, TR[2], TC[2], acc[2]);
            if (j > 0) {
                // update j-1 of the tiles that are not due yet, in the shadow of the pivots of stage j
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (TC[k] > j && TC[k] < nStages)
                        acc[k] = mmTile<true, kQB>(acc[k], L0 + kQB * (j - 1), kSP, kQB * TR[k], L0 + kQB * (j - 1), kSP, kQB * TC[k], lane, -1.0);
            }
            EQF_F64STAMP(j, 1);
            EQF_LDS_BARRIER();  // ---- A: columns j of L are in LDS
            EQF_F64STAMP(j, 2);
            if (j + 1 < nStages) {
                // update j of the tiles of column j+1, which the next sixteen pivots start from: the only thing between two pivot chains
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (TC[k] == j + 1) {
                        acc[k] = mmTile<true, kQB>(acc[k], L0 + kQB * j, kSP, kQB * TR[k], L0 + kQB * j, kSP, kQB * (j + 1), lane, -1.0);
                        stTile(acc[k], L0, kSP, kQB * TR[k], kQB * (j + 1), lane);
                    }
                EQF_F64STAMP(j, 3);
                EQF_LDS_BARRIER();  // ---- B
                EQF_F64STAMP(j, 4);
            }
        }
    }
}
// `pre(wave, r, c, acc)`: waves 2 and 3 ask for the 16 x 16 tile (r, c), c >= 1, of the block in the accumulator layout, each once, in the
// shadow of the pivots (the diagonal workgroup's remaining trailing-update tiles; preFromLds: they already are in s.L).  `mid()`: thread 0, once, after the first stage's barrier A, by which
// every thread has drained its write-through stores (WT): k_chol_resident publishes the solved block L_{R,R-1} there.
// `stageFlag` (WT only; nullptr = none): stageFlag[j] <- epoch as soon as columns [16 j, 16 j + 16) of L and W_jj are in the record Dn,
// j = 0..nStages-2 -- one stage behind the pivot chain.
// One loop per wave role (the wave index is branched on OUTSIDE the stage loops: inside a common loop every wave pays for the address
// induction variables and the branches of all roles -- a quarter of what was left between two pivot chains).  Every role passes the same
// barriers: A_0 B_0 A_1 B_1 .. A_{n-1}, then the tail's.
template <bool WT = false, typename Pre, typename Mid>
EQF_DI void factor64(const Lds64& s, int tid, int* bad, Pre pre, double* Dn, long long* st, int nStages, Mid mid, int* stageFlag = nullptr,
    int epoch = 0) {
    const int lane = tid & 63, wv = tid >> 6;
    double* const L0 = &s.L[0][0];
    const int jl = nStages - 1;
    if (wv == 0) {
        // ---- the pivot wave
        double* rp = L0 + lane * kSP;  // stage 0: row `lane`, columns 0..15
#pragma unroll 1
        for (int j = 0; j < nStages; ++j) {
            EQF_F64STAMP(j, 0);
            if (j > 0) {
                const int base = kQB * j;
                rp = lane < kSB - base ? L0 + (lane + base) * kSP + base : &s.Wd[j][lane >= 48 ? lane - 48 : 0][0];
            }
            potrf16v2(rp, j == 0 || lane < kSB - kQB * j || lane >= 48, j == jl && j > 0, s.Wd[j], lane, bad, j);
            if (WT && j == 0) hoDrain();
            EQF_F64STAMP(j, 1);
            EQF_LDS_BARRIER();  // A
            EQF_F64STAMP(j, 2);
            if (j == 0 && lane == 0) mid();
            if (j == jl) break;
            EQF_LDS_BARRIER();  // B
            EQF_F64STAMP(j, 4);
        }
    } else if (wv == 1) {
        // ---- stage 0: the identity rows next to a copy of the leading block (the pivot wave has no idle lanes there); then the chores,
        // one stage behind the pivots: W_jj^T -> W_jj, zeros above the diagonal of L_jj, columns j of the record, drain, stage flag
        EQF_F64STAMP(0, 0);
        potrf16v2(lane < kQB ? &s.D0[lane][0] : &s.Wd[0][lane >= 48 ? lane - 48 : 0][0], lane >= 48, false, s.Wd[0], lane, bad);
        if (WT) hoDrain();
        EQF_F64STAMP(0, 1);
        EQF_LDS_BARRIER();  // A_0
#pragma unroll 1
        for (int j = 1; j < nStages; ++j) {
            EQF_LDS_BARRIER();  // B_{j-1}
            EQF_F64STAMP(j, 0);
            transpose16(s.Wd[j - 1], lane);
            zeroUpper16(s, kQB * (j - 1), lane);
            EQF_F64STAMP(j, 5);
            if (Dn) {
                storeStageWave<WT>(s, Dn, j - 1, lane);
                EQF_F64STAMP(j, 6);
                if (WT && stageFlag) {
                    hoDrain();
                    if (lane == 0) hoPublish(stageFlag + (j - 1), epoch);
                }
            }
            EQF_F64STAMP(j, 1);
            EQF_LDS_BARRIER();  // A_j
        }
    } else if (wv == 2) {
        factorTiles<2, WT>(s, lane, pre, nStages);
    } else {
        factorTiles<3, WT>(s, lane, pre, nStages);
    }
    // ---- tail.  The last real stage jl: its inverse block is already row-major (the pivot wave scattered it) unless jl == 0 (wave 1 held
    // the identity rows: transposed)
    if (jl == 0) {
        if (wv == 1) transpose16(s.Wd[0], lane);
        EQF_LDS_BARRIER();
    }
    if (wv == 1) zeroUpper16(s, kQB * jl, lane);
#ifdef EQF_F64_STAMPS
    if (st) {
        EQF_LDS_BARRIER();
        if (tid < 128) st[tid] = (&sF64Stamps[0][0][0])[tid];
    }
#endif
    if (Dn)
        for (int j = jl; j < 4; ++j) storeDiagColumns16<WT>(s, Dn, j, tid, 256);
    EQF_LDS_BARRIER();
}
// (the block is complete in s.L: the tiles come from there)
EQF_DI void factor64(const Lds64& s, int tid, int* bad, double* Dn = nullptr, long long* st = nullptr, int nStages = 4) {
    const int lane = tid & 63;
    double* const L0 = &s.L[0][0];
    factor64<false>(s, tid, bad, [&](int, int r, int c, f64x4& acc) { acc = ldTile(L0, kSP, kQB * r, kQB * c, lane); }, Dn, st, nStages, [] {});
}

// stages of the 64-wide block starting at column c0 of a chain of real order n that hold a real column
EQF_DI int realStages(int n, int c0) { return max(1, min(4, (min(kSB, n - c0) + kQB - 1) / kQB)); }

// Panel solve of one 16-wide strip of M in place, four 16x16 stages chained through the accumulators:
//   TR = true : the strip is rows x0..x0+15,    M <- M L^-T   (held transposed:  X_j^T = W_jj (A_j^T - sum L_ji X_i^T))
//   TR = false: the strip is columns x0..x0+15, M <- L^-1 M   (                  Y_j   = W_jj (R_j   - sum L_ji Y_i))
template <bool TR>
EQF_DI void solveStrip(double* M, int ld, const Lds64& s, int x0, int lane) {
    f64x4 Z[4], X[4];
    const int lc = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) Z[j][q] = TR ? M[(x0 + lc) * ld + kQB * j + lg + 4 * q] : M[(kQB * j + lg + 4 * q) * ld + x0 + lc];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
        X[j] = mmRegB(zero, &s.Wd[j][0][0], kWP, 0, 0, Z[j], lane, 1.0);
#pragma unroll
        for (int j2 = j + 1; j2 < 4; ++j2) Z[j2] = mmRegB(Z[j2], &s.L[0][0], kSP, kQB * j2, kQB * j, X[j], lane, -1.0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (TR) M[(x0 + lc) * ld + kQB * j + lg + 4 * q] = X[j][q];
            else M[(kQB * j + lg + 4 * q) * ld + x0 + lc] = X[j][q];
        }
}

// z <- L^-1 z for one 64-vector in LDS, by ONE wave (lanes 0..15 carry a 16-block), with the inverse diagonal blocks:
//   z_j = W_jj (z_j - sum_{k < 16 j} L[16 j + r][k] z_k).  LDS traffic of one wave is ordered; the asm statements keep the
// compiler from moving loads across the in-place stores.
EQF_DI void solveVec64(double* z, const Lds64& s, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
#define EQF_LDS_ORDER() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define EQF_LDS_ORDER() do { } while (0)
#endif
    const int r = lane & 15;
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        double t = z[kQB * j + r];
        for (int k = 0; k < kQB * j; ++k) t = fma(-s.L[kQB * j + r][k], z[k], t);
        EQF_LDS_ORDER();
        if (lane < kQB) z[kQB * j + r] = t;
        EQF_LDS_ORDER();
        double zz = 0.0;
#pragma unroll
        for (int c = 0; c < kQB; ++c) zz = fma(s.Wd[j][r][c], z[kQB * j + c], zz);
        EQF_LDS_ORDER();
        if (lane < kQB) z[kQB * j + r] = zz;
        EQF_LDS_ORDER();
    }
#undef EQF_LDS_ORDER
}

// The first diagonal blocks of the two chains, formed straight from Sigma and factored INSIDE the prep launch (two extra
// workgroups per filter that finish in the shadow of the landmark waves), so that the first chain launch is an ordinary
// one:   kind 0:  S_00 = C Sigma C^T + R for the first 32 landmarks   (same expression order as k_update_prep)
//        kind 1:  Sigma_e[0:64, 0:64]                                 (identity at the pad index 5 and beyond n_e)
// WT: the record is handed over INSIDE a launch (k_chol_resident's role F0: write-through stores; the caller drains and publishes D[0]'s flag)
template <typename T, bool WT = false>
EQF_DI void factorFirstFromSigma(const UpdArgs& a, const ChainArgs& ch, int b, const Lds64& s, int* bad) {
    const Glob& g = a.g[b];
    if (ch.kind == 0 && a.resCounters && threadIdx.x < 4) a.resCounters[4 * b + threadIdx.x] = 0;  // (k_chol_resident's work counters)
    if (!g.updateOk || g.N == 0) return;
    const int N = g.N, cap = a.cap, ld = a.ld, tid = threadIdx.x;
    const T* Sin = static_cast<const T*>(a.Sin) + (long long)b * a.sigmaStride;
    // (every global read of a role is issued before the first use: the block was written by the previous launch on other
    // XCDs, a read-use-read-use loop would pay the ~2 us first-touch miss once per trip)
    if (ch.kind == 0) {
        const double* lmc = a.lmc + (long long)b * 15 * cap;
        const double mv = a.prm.measurementVariance;
        double Ci[4][6], Cj[4][6], v[4][9];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u, i = min(e >> 5, N - 1), j = min(e & 31, N - 1);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                Ci[u][q] = lmc[(long long)q * cap + i];
                Cj[u][q] = lmc[(long long)q * cap + j];
            }
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) v[u][3 * q + c] = (double)Sin[(long long)(kLm0 + 3 * i + q) * ld + kLm0 + 3 * j + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u, i = e >> 5, j = e & 31;
            double o[4] = {0.0, 0.0, 0.0, 0.0};
            if (i < N && j < N) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    double cs[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) cs[c] = Ci[u][3 * r] * v[u][c] + Ci[u][3 * r + 1] * v[u][3 + c] + Ci[u][3 * r + 2] * v[u][6 + c];
#pragma unroll
                    for (int q = 0; q < 2; ++q) o[2 * r + q] = cs[0] * Cj[u][3 * q] + cs[1] * Cj[u][3 * q + 1] + cs[2] * Cj[u][3 * q + 2];
                }
                if (i == j) {
                    o[0] += mv;
                    o[3] += mv;
                }
            } else if (i == j) {
                o[0] = o[3] = 1.0;
            }
            s.L[2 * i][2 * j] = o[0]; s.L[2 * i][2 * j + 1] = o[1];
            s.L[2 * i + 1][2 * j] = o[2]; s.L[2 * i + 1][2 * j + 1] = o[3];
        }
    } else {
        const int ne = eDim(N);
        T v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u, rr = e >> 6, cc = e & 63;
            const bool in = rr < ne && cc < ne && rr != 5 && cc != 5;
            v[u] = Sin[(long long)(6 + (in ? rr : 0)) * ld + 6 + (in ? cc : 0)];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u, rr = e >> 6, cc = e & 63;
            const bool in = rr < ne && cc < ne && rr != 5 && cc != 5;
            s.L[rr][cc] = in ? (double)v[u] : (rr == cc ? 1.0 : 0.0);
        }
    }
    __syncthreads();
    factorPrologue(s, tid);
    __syncthreads();
    const int nst = realStages(ch.kind == 0 ? sDim(N) : eDim(N), 0);
    if (WT) {
        const int ln = tid & 63;
        double* const L0 = &s.L[0][0];
        factor64<true>(s, tid, bad, [&](int, int r, int c, f64x4& acc) { acc = ldTile(L0, kSP, kQB * r, kQB * c, ln); }, ch.D + (long long)b * ch.strideD, nullptr, nst,
            [] {});
    } else {
        factor64(s, tid, bad, ch.D + (long long)b * ch.strideD, nullptr, nst);
    }
}
// k_update_prep + the two first-block workgroups (grid.x = lmBlocks + eBlocks + 2)
#ifdef EQF_PREP_STAMPS
__device__ long long g_prepStamps[512][2];  // per workgroup: first / last cycle
#define EQF_PREPSTAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 512 && blockIdx.y == 0) g_prepStamps[blockIdx.x][k] = wall_clock64(); } while (0)
#else
#define EQF_PREPSTAMP(k) do { } while (0)
#endif
// OCC2: compiled for two workgroups per CU (256 registers instead of 268: 60 bytes of scratch in the first-block workgroups) and launched
// with the 44 KB LDS the landmark waves / factor64 need instead of the 119 KB image -- for batches whose prep launch is larger than the
// chip (4 filters of N = 200 on: 16 filters 55 -> 47 us, 64 filters 166 -> 122 us).  One or two filters keep the one-per-CU build: there the
// two first-block workgroups ARE the launch's critical path and the spill costs them 1.8 us (12.5 -> 14.3 us).
template <typename T, bool OCC2 = false>
__global__ __launch_bounds__(256, OCC2 ? 2 : 1) void k_update_prep64(UpdArgs a, ChainArgs cS, ChainArgs cE, int lmBlocks, int eBlocks, int wpb, int nvPad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    const int role = (int)blockIdx.x - (lmBlocks + eBlocks);
    EQF_PREPSTAMP(0);
    if (role < 0) {
        updatePrepBody<T>(a, (int)blockIdx.x, (int)blockIdx.y, lmBlocks, wpb, nvPad, reinterpret_cast<double*>(smem64));
        __syncthreads();
        EQF_PREPSTAMP(1);
        return;
    }
    int bad = 0;
    factorFirstFromSigma<T>(a, role == 0 ? cS : cE, blockIdx.y, ldsFactor(smem64), &bad);
    if (bad && a.errflag && threadIdx.x == 0) atomicOr(a.errflag, 4);
    __syncthreads();
    EQF_PREPSTAMP(1);
}
// The two first-block workgroups as a launch of their own (grid = (2, B)): used when the prep launch has more workgroups
// than the chip has CUs -- there the 119 KB of LDS these two need would cost every prep workgroup its occupancy.
// Stand-alone variant (tests / microbenchmarks without a prep launch): factor A_00 of each chain from ChainArgs::A.
inline __global__ __launch_bounds__(256) void k_factor_first64(ChainArgs c0, ChainArgs c1, int* errflag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    const Lds64 s = ldsFull(smem64);
    const ChainArgs& ch = blockIdx.x ? c1 : c0;
    const int b = blockIdx.y, tid = threadIdx.x;
    if (ch.nbMax == 0 || !ch.g[b].updateOk || ch.g[b].N == 0) return;
    const double* A = ch.A + (long long)b * ch.strideA;
    for (int e = tid; e < kSB * kSB; e += 256) s.L[e >> 6][e & 63] = A[(long long)(e >> 6) * ch.ldA + (e & 63)];
    __syncthreads();
    factorPrologue(s, tid);
    __syncthreads();
    int bad = 0;
    factor64(s, tid, &bad, ch.D + (long long)b * ch.strideD);
    if (bad && errflag && tid == 0) atomicOr(errflag, 4);
}

// grid.x = sum over the two chains of nbMax^2 (A tiles) + wtMax * nbMax (rhs tiles), nbMax / wtMax in 64-blocks,
//          + the tiles of the covariance downdate in the ONE launch that carries it (ddNt > 0);
// grid.y = B; block = 256; dynamic LDS = sizeof(Step64Lds).
//
// What used to be three more launches rides along (they only exist as separate kernels for the 32-wide path):
//   * gamma = Y^T z, hV = (L^-1 V)^T z and G11 = [Zt|Et]^T [Zt|Et] (k_update_reduce) are accumulated block row by block
//     row by the rhs workgroups that produce Y_K / Z_K (the S-chain ones solve the 64 entries of z_K along);
//   * Sigma <- Sigma - Y^T Y (k_downdate) runs as extra workgroups of the first launch after the S-chain has finished,
//     next to the remaining E-chain steps (the S-chain is always the shorter one);
//   * the innovation lift / X <- Delta X (updateFinishBody) is done by the E-chain's rhs workgroup in its last step.
// embedFinish = 0 (some filter of the batch has chains of equal length): the host launches k_downdate afterwards.
//
// PHASE 0: fused launch (above): every tile workgroup solves the two panel blocks it needs itself -- right when a launch
//          is bound by the serial diagonal chain (one / a few small filters).
// PHASE 1 + PHASE 2: the split chain for throughput (many tiles per launch): a panel launch solves every block of column
//          K ONCE, in place (A_RK <- L_RK, Y_K -> WO), an update launch then only multiplies: no redundant solves
//          (2.25x fewer MFMAs per tile) and the 77 KB LDS layout lets two workgroups share a CU.
// PHASE 3: an update launch that ALSO solves block column K+1 (the next panel launch folded in: one launch per block column
//          instead of two).  The workgroups of column K+1 keep their freshly updated tile in registers, wait INSIDE the
//          launch for the diagonal workgroup to publish D[K+1] (eqf_handoff.hpp: write-through record + epoch flag, 2.6 us
//          for the 40 KB record), then solve in place.  Only those (nb - K - 2 + wt) workgroups per chain wait, and the
//          diagonal workgroup they wait for has the lowest block index of its filter's chain, so it is dispatched first.
// ---- PHASE 3 (update + solve of block column K+1 in one launch): how a launch's workgroups are laid out.
// Per filter and chain with rem = nbMax - K - 1 block rows left and m = rem - 1:
//   1 diagonal workgroup  (tile (K+1, K+1): update, factor, publish D[K+1])
//   m + wt tails          (tiles (R, K+1), R > K+1, and the rhs tiles of block row K+1: update, wait for D[K+1], solve)
//   m (m + 1) / 2 + wt m  pure updates -- NOT one workgroup each: nStream workgroups per filter walk the list with the
//                         operands of the next tile in flight while the matrix cores work on the current one.
// A pure update is 128 KB of traffic for 1.7 us of matrix-core time; as one workgroup per tile (load everything, then
// compute, then store, two workgroups per CU) a tile cost 8.4 CU-microseconds -- a batch of 64 filters ran its chain
// launches at 19 % of the fp64 MFMA rate whether its data sat in HBM or in the MALL.
__host__ __device__ inline void step3Counts(int nbMax, int wtMax, int K, int* nTail, int* nUpd) {
    const int rem = nbMax - K - 1, m = rem > 1 ? rem - 1 : 0;
    *nTail = rem >= 1 ? m + wtMax : 0;
    *nUpd = m * (m + 1) / 2 + wtMax * m;
}

// Element x of a workgroup class that starts at linear workgroup index o (per filters of `per` workgroups each) -> filter b and
// index i within the filter, such that all workgroups of a filter share (o + x) % 8: observed, for speed only, workgroup L of a
// launch runs on XCD L % 8 -- a filter's panel blocks (and, in the downdate, its Y) are then fetched into ONE 4 MB L2 and
// reused there instead of crossing the fabric once per tile.  Any placement is correct.
EQF_DI void xcdSplit(long long o, long long x, int per, int Bn, int* b, int* i) {
    if ((Bn & 7) == 0) {
        const int r = int((o + x) & 7);
        const long long q = x >> 3;
        *b = r + 8 * int(q / per);
        *i = int(q % per);
    } else {
        *b = int(x / per);
        *i = int(x % per);
    }
}

struct StreamTile {
    double* Ct;
    const double* Pg;
    const double* Qg;
    int ldcp, ldq;  // leading dimension of Ct and Pg (the same matrix family), of Qg
    int isW;
    const double* CtL;  // where the tile's current value is READ: Ct, or -- block column 0 of the E-chain, UpdArgs::eFromSigma -- Sigma
    int ldL;
};

// update tiles first, first + stride, ... of filter b in launch K (see step3Counts); 256 threads, LDS: s.P, s.Q
EQF_DI void streamUpdates64(const ChainArgs& c0, const ChainArgs& c1, const UpdArgs& a, int b, int K, int first, int stride, const Lds64& s) {
    const Glob& g = c0.g[b];
    if (!g.updateOk || g.N == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int nbS, wtS, nbE, wtE;
    chainDims64(c0, g.N, &nbS, &wtS);
    chainDims64(c1, g.N, &nbE, &wtE);
    const int mS = max(c0.nbMax - K - 2, 0), mE = max(c1.nbMax - K - 2, 0);
    const int nAS = mS * (mS + 1) / 2, nS = nAS + c0.wtMax * mS, nAE = mE * (mE + 1) / 2, total = nS + nAE + c1.wtMax * mE;
    // (field-by-field selects: a reference picked between the two argument structs at run time makes the compiler copy both to
    // scratch memory)
    double* const AS = c0.A + (long long)b * c0.strideA;
    double* const AE = c1.A + (long long)b * c1.strideA;
    double* const WS = c0.W + (long long)b * c0.strideW;
    double* const WE = c1.W + (long long)b * c1.strideW;
    const double* const YS = c0.WO + (long long)b * c0.strideW;
    const double* const YE = c1.WO + (long long)b * c1.strideW;
    const int ldAS = c0.ldA, ldAE = c1.ldA, ldWS = c0.ldW, ldWE = c1.ldW;
    // E-chain A tiles in launch 0: their values are still Sigma's (fp64 only; the last block row comes from EA: identity padding)
    const bool eSrc = a.eFromSigma && K == 0;
    const double* Se = static_cast<const double*>(a.Sin) + (long long)b * a.sigmaStride + 6LL * a.ld + 6;
    const int ldSg = a.ld;
    auto decode = [&](int u, StreamTile& t) -> bool {
        const bool second = u >= nS;
        const int nb = second ? nbE : nbS, wt = second ? wtE : wtS, m = second ? mE : mS, nA = second ? nAE : nAS;
        int v = second ? u - nS : u;
        double* A = second ? AE : AS;
        const int ldA = second ? ldAE : ldAS, ldW = second ? ldWE : ldWS;
        const bool isA = v < nA;
        // A tile (R, C) of the lower triangle, R, C >= K + 2  |  rhs tile: column tile tt of block row C
        int R = 0, C = 0, tt = 0;
        if (isA) {
            int r = 0;
            while (v >= r + 1) {
                v -= r + 1;
                ++r;
            }
            R = K + 2 + r;
            C = K + 2 + v;
        } else {
            v -= nA;
            tt = v / max(m, 1);
            C = K + 2 + v % max(m, 1);
        }
        // (every field assigned once, from values: fields written in the two arms of a branch end up in scratch memory)
        const int ldcp = isA ? ldA : ldW;
        double* Cb = isA ? A : (second ? WE : WS);
        const double* Pb = isA ? A : (second ? YE : YS);
        t.Ct = Cb + (long long)((isA ? R : C) * kSB) * ldcp + (isA ? C : tt) * kSB;
        t.Pg = Pb + (long long)((isA ? R : K) * kSB) * ldcp + (isA ? K : tt) * kSB;  // (R == C: the same block as Qg, read twice)
        t.Qg = A + (long long)(C * kSB) * ldA + K * kSB;
        t.ldcp = ldcp;
        t.ldq = ldA;
        t.isW = isA ? 0 : 1;
        const bool fromS = eSrc && second && isA && R < nb - 1;
        t.CtL = fromS ? Se + (long long)(R * kSB) * ldSg + C * kSB : t.Ct;
        t.ldL = fromS ? ldSg : ldcp;
        return isA ? R < nb : (tt < wt && C < nb);
    };
    // a contiguous chunk of the list per workgroup: consecutive tiles share their P operand (A tiles of one block row: L_RK;
    // rhs tiles of one column tile: Y_Kt), which then stays in LDS
    const int chunk = (total + stride - 1) / stride, uEnd = min(total, (first + 1) * chunk);
    StreamTile cur, nxt;
    int u = first * chunk;
    bool have = false;
    while (u < uEnd && !(have = decode(u, nxt))) ++u;
    if (!have) return;
    // operands as 16-byte pairs: thread -> rows (tid >> 5) + 8 q, columns 2 (tid & 31), 2 (tid & 31) + 1
    const int pr = tid >> 5, pc = 2 * (tid & 31);
    f64x4 nAcc[4];
    f64x2 nP[8], nQ[8];
    auto issue = [&](const StreamTile& t, bool withP) {
#pragma unroll
        for (int q = 0; q < 8; ++q) nQ[q] = *reinterpret_cast<const f64x2*>(t.Qg + (long long)(pr + 8 * q) * t.ldq + pc);
        if (withP) {
#pragma unroll
            for (int q = 0; q < 8; ++q) nP[q] = *reinterpret_cast<const f64x2*>(t.Pg + (long long)(pr + 8 * q) * t.ldcp + pc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) nAcc[i][q] = t.CtL[(long long)(kQB * wv + (lane >> 4) + 4 * q) * t.ldL + kQB * i + (lane & 15)];
    };
    issue(nxt, true);
    bool newP = true;  // the tile in flight brings its own P block (otherwise the one in LDS is its P too)
    // Order inside an iteration: [operands of tile i -> LDS] [result of tile i-1 -> memory] [fetch of tile i+1] [arithmetic of
    // tile i].  The wait counter retires in order: with the stores issued BEFORE the next fetch, waiting for that fetch at the
    // top of the next iteration only has stores in front of it that are a whole tile's arithmetic old.
    f64x4 res[4];
    double* resPtr = nullptr;
    int resLd = 0;
    for (;;) {
        cur = nxt;
        f64x4 acc[4];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            s.Q[pr + 8 * q][pc] = nQ[q][0];
            s.Q[pr + 8 * q][pc + 1] = nQ[q][1];
        }
        if (newP) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                s.P[pr + 8 * q][pc] = nP[q][0];
                s.P[pr + 8 * q][pc + 1] = nP[q][1];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = nAcc[i];
        __syncthreads();
        if (resPtr) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) resPtr[(long long)(kQB * wv + (lane >> 4) + 4 * q) * resLd + kQB * i + (lane & 15)] = res[i][q];
        }
        ++u;
        have = false;
        while (u < uEnd && !(have = decode(u, nxt))) ++u;
        newP = have && nxt.Pg != cur.Pg;
        if (have) issue(nxt, newP);
        __builtin_amdgcn_sched_barrier(0);  // (the next tile's loads stay ahead of this tile's arithmetic)
        if (cur.isW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mmTile<false, kSB>(acc[i], &s.Q[0][0], kSP, kQB * wv, &s.P[0][0], kSP, kQB * i, lane, -1.0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mmTile<true, kSB>(acc[i], &s.P[0][0], kSP, kQB * wv, &s.Q[0][0], kSP, kQB * i, lane, -1.0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) res[i] = acc[i];
        resPtr = cur.Ct;
        resLd = cur.ldcp;
        if (!have) break;
        __syncthreads();  // every wave is done with s.P / s.Q
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) resPtr[(long long)(kQB * wv + (lane >> 4) + 4 * q) * resLd + kQB * i + (lane & 15)] = res[i][q];
}

#ifdef EQF_CHOL_WG_STAMPS
__device__ long long g_cholWg[16][256][2];  // [launch K][workgroup]: first / last cycle
__device__ int g_cholWgInfo[16][256][4];    // second chain?, isW, R, C
struct CholWgStamp {
    int K;
    __device__ CholWgStamp(int k) : K(k) {
        if (threadIdx.x == 0 && blockIdx.y == 0 && K < 16 && blockIdx.x < 256) g_cholWg[K][blockIdx.x][0] = __builtin_readcyclecounter();
    }
    __device__ ~CholWgStamp() {
        __syncthreads();
        if (threadIdx.x == 0 && blockIdx.y == 0 && K < 16 && blockIdx.x < 256) g_cholWg[K][blockIdx.x][1] = __builtin_readcyclecounter();
    }
};
#endif
template <typename T, int PHASE>
__global__ __launch_bounds__(256, (PHASE == 2 || PHASE == 3) ? 2 : 1) void k_chol_step64(ChainArgs c0, ChainArgs c1, UpdArgs a, int K, int ddNt, int ddSmall,
    int embedFinish, int* errflag, int nStream = 0, int tailsLast = 0) {
#ifdef EQF_CHOL_WG_STAMPS
    CholWgStamp wgStamp(K);
#endif
    int b = blockIdx.y;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    bool second = false, isW = false;
    int R = 0, C = 0;  // A tile (R,C) or rhs tile (t = R, C)
    if (PHASE == 3) {
        // dispatch order = linear workgroup index: [diagonal workgroups of every filter | tails | streams | downdate tiles]
        // tailsLast = 0: as written; 1: streams before tails; 2: filter-major, see below
        const int Bn = gridDim.y;
        long long x = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        int tS, uS, tE, uE;
        step3Counts(c0.nbMax, c0.wtMax, K, &tS, &uS);
        step3Counts(c1.nbMax, c1.wtMax, K, &tE, &uE);
        const int nT = tS + tE;
        const long long nDiag = 2LL * Bn, nTail = (long long)nT * Bn, nStr = (long long)nStream * Bn;
        int cls = 0;  // 0 diagonal, 1 tail, 2 stream, 3 downdate
        int bx = 0;   // filter, classes 1 and 2
        if (x >= nDiag) {
            x -= nDiag;
            if (tailsLast == 2) {
                // filter-major: [streams of filter 0 | tails of filter 0 | streams of filter 1 | ...] -- the latency-bound tails of one
                // filter share the chip with the bandwidth-bound streams of its neighbours, and all but the first few filters' tails
                // find D[K+1] published when they start
                const int per = nStream + nT;
                if (x < (long long)per * Bn) {
                    bx = int(x / per);
                    x %= per;
                    if (x < nStream) cls = 2;
                    else { x -= nStream; cls = 1; }
                } else { x -= (long long)per * Bn; cls = 3; }
            } else {
                const long long nFirst = tailsLast ? nStr : nTail, nSecond = tailsLast ? nTail : nStr;
                long long o = nDiag;
                if (x < nFirst) cls = tailsLast ? 2 : 1;
                else if (x - nFirst < nSecond) { x -= nFirst; o += nFirst; cls = tailsLast ? 1 : 2; }
                else { x -= nFirst + nSecond; o += nFirst + nSecond; cls = 3; }
                int ix = 0;
                if (cls == 1) xcdSplit(o, x, max(nT, 1), Bn, &bx, &ix);
                if (cls == 2) xcdSplit(o, x, max(nStream, 1), Bn, &bx, &ix);
                if (cls == 3) {
                    const int ddTiles = ddNt * (ddNt + 1) / 2;
                    if (ddTiles == 0) return;
                    xcdSplit(o, x, ddTiles, Bn, &bx, &ix);
                    if (bx >= Bn) return;
                    if (ddSmall) downdateTile<T, 32>(a, ddNt, bx, ix, reinterpret_cast<T*>(smem64));
                    else downdateTile<T, 64>(a, ddNt, bx, ix, reinterpret_cast<T*>(smem64));
                    return;
                }
                x = ix;
            }
        }
        if (cls == 3) {
            const int ddTiles = ddNt * (ddNt + 1) / 2;
            if (ddTiles == 0) return;
            b = int(x / ddTiles);
            if (b >= Bn) return;
            if (ddSmall) downdateTile<T, 32>(a, ddNt, b, int(x % ddTiles), reinterpret_cast<T*>(smem64));
            else downdateTile<T, 64>(a, ddNt, b, int(x % ddTiles), reinterpret_cast<T*>(smem64));
            return;
        }
        if (cls == 2) {
            streamUpdates64(c0, c1, a, bx, K, int(x), nStream, ldsTail(smem64));
            return;
        }
        if (cls == 0) {
            b = int(x >> 1);
            second = (x & 1) != 0;
            R = C = K + 1;
        } else {
            b = bx;
            int i = int(x);
            second = i >= tS;
            if (second) i -= tS;
            const int m = max(pickValue(second, c0.nbMax, c1.nbMax) - K - 2, 0);
            C = K + 1;
            if (i < m) R = K + 2 + i;
            else {
                isW = true;
                R = i - m;
            }
        }
    } else {
        const int n0 = chainBlocks64(c0.nbMax, c0.wtMax, K, PHASE);
        const int nChain = n0 + chainBlocks64(c1.nbMax, c1.wtMax, K, PHASE);
        if ((int)blockIdx.x >= nChain) {  // covariance downdate tile
            if (ddSmall) downdateTile<T, 32>(a, ddNt, b, (int)blockIdx.x - nChain, reinterpret_cast<T*>(smem64));
            else downdateTile<T, 64>(a, ddNt, b, (int)blockIdx.x - nChain, reinterpret_cast<T*>(smem64));
            return;
        }
        second = (int)blockIdx.x >= n0;
    }
    const ChainArgs ch = pickChain(second, c0, c1);
    const Glob& g = ch.g[b];
    if (!g.updateOk || g.N == 0) return;
    int nb, wt;
    chainDims64(ch, g.N, &nb, &wt);
    if (K >= nb) return;
    const int rem = ch.nbMax - K - 1;
    if (rem < 0) return;
    if (PHASE == 3) {
        if (rem < 1) return;
        if (isW ? (R >= wt || C >= nb) : (R >= nb)) return;
    } else {
        int idx = second ? (int)blockIdx.x - chainBlocks64(c0.nbMax, c0.wtMax, K, PHASE) : (int)blockIdx.x;
        if (PHASE == 1) {
            if (idx < rem) {  // panel launch: the blocks (R, K) below the diagonal
                R = K + 1 + idx;
                C = K;
                if (R >= nb) return;
            } else {
                isW = true;
                R = idx - rem;
                C = K;
                if (R >= wt) return;
            }
        } else if (idx < rem * rem) {
            R = K + 1 + idx / rem;
            C = K + 1 + idx % rem;
            if (R >= nb || C > R) return;
        } else {
            idx -= rem * rem;
            isW = true;
            const int cols = PHASE == 0 ? rem + 1 : rem;  // block rows of the right-hand sides still in play
            R = idx / cols;
            C = (PHASE == 0 ? K : K + 1) + idx % cols;
            if (R >= wt || C >= nb) return;
        }
    }
    double* A = ch.A + (long long)b * ch.strideA;
    double* D = ch.D + (long long)b * ch.strideD;
    double* W = ch.W + (long long)b * ch.strideW;
    double* WO = ch.WO + (long long)b * ch.strideW;
    const int ldA = ch.ldA, ldW = ch.ldW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    constexpr bool UPD = PHASE == 2 || PHASE == 3;  // update launches of the split chain: panel blocks arrive solved
    const Lds64 s = PHASE == 3 ? ldsTail(smem64) : (PHASE == 2 ? ldsUpdate(smem64) : ldsFull(smem64));
    int bad = 0;
#ifdef EQF_CHOL_WG_STAMPS
    if (threadIdx.x == 0 && blockIdx.y == 0 && K < 16 && blockIdx.x < 256) {
        g_cholWgInfo[K][blockIdx.x][0] = second;
        g_cholWgInfo[K][blockIdx.x][1] = isW;
        g_cholWgInfo[K][blockIdx.x][2] = R;
        g_cholWgInfo[K][blockIdx.x][3] = C;
    }
#endif
    const bool diagNext = PHASE != 1 && !isW && R == C && C == K + 1;
    const bool needQ = C > K;
    const bool needP = isW || R != C;
    const bool solveOnly = isW && C == K;
    const bool panelA = PHASE == 1 && !isW;  // solve in place, no update
    // PHASE 3: this workgroup's tile belongs to block column K+1 (rhs: block row K+1): it is solved in this launch
    const bool tail = PHASE == 3 && C == K + 1 && !diagNext;

    // ---- which 16x16 tiles of the 64x64 output tile this wave owns: its 16-row strip (tiles (wv, 0..3)); the diagonal
    // workgroup needs the lower triangle only and the first column first: slot 0 = (wv, 0), then the other tiles spread
    // over waves 2, 3 (they run while waves 0, 1 factor the first 16 columns)
    int tr[4], tc[4], nt;
    if (!diagNext) {
        nt = 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { tr[i] = wv; tc[i] = i; }
    } else {
        // deferred tiles (1,1) (2,1) (3,1) (2,2) (3,2) (3,3): wave 2 takes the even, wave 3 the odd ones
        nt = wv >= 2 ? 4 : 1;
        tr[0] = wv; tc[0] = 0;
        tr[1] = wv == 2 ? 1 : 2; tc[1] = 1;
        tr[2] = wv == 2 ? 3 : 2; tc[2] = wv == 2 ? 1 : 2;
        tr[3] = 3;               tc[3] = wv == 2 ? 2 : 3;
    }

    EQF_STAMP(0);
    if (second && isW && C == K && K == nb - 1) EQF_FSTAMP(0);
    // ---- every global read of the launch is issued up front (the data was written by the previous launch on other
    // XCDs: each access is a ~2 us miss, so they must all be in flight together)
    double* Ct = isW ? (W + (long long)(C * kSB) * ldW + R * kSB) : (A + (long long)(R * kSB) * ldA + C * kSB);
    const int ldc = isW ? ldW : ldA;
    // (UpdArgs::eFromSigma: in launch 0 the E-chain's A tiles still hold Sigma's values and are read there -- EA = Sigma[6:, 6:] is a
    // plain offset; the last block row, which holds the identity padding, comes from EA as ever)
    const bool eSrc = a.eFromSigma && K == 0 && second && !isW && R < nb - 1 && (PHASE == 1 || PHASE == 3);
    const double* Se = static_cast<const double*>(a.Sin) + (long long)b * a.sigmaStride + 6LL * a.ld + 6;
    const double* CtL = eSrc ? Se + (long long)(R * kSB) * a.ld + C * kSB : Ct;
    const int ldcL = eSrc ? a.ld : ldc;
    f64x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            acc[i][q] = (solveOnly || panelA || i >= nt) ? 0.0 : CtL[(long long)(kQB * tr[i] + (lane >> 4) + 4 * q) * ldcL + kQB * tc[i] + (lane & 15)];
    // running sums of the reductions (rhs workgroups): previous value of this thread's entry, fetched with everything else
    double prevSum = 0.0;
    double* sumPtr = nullptr;
    const bool rhsSolve = solveOnly || (tail && isW);  // this workgroup solves a block row of right-hand sides ...
    const int Kp = tail ? K + 1 : K;                   // ... namely block row Kp
    if (rhsSolve) {
        const int nvv = kLm0 + 3 * g.N;
        if (ch.kind == 0) {
            const int col = R * kSB + tid;
            if (tid < kSB && col < nvv) sumPtr = a.dbgGamma + (long long)b * (kLm0 + 3 * a.cap) + col;
            else if (tid < kSB && col < nvv + 6) sumPtr = a.red + (long long)b * 256 + col - nvv;
        } else if (tid < 121) {
            sumPtr = a.red + (long long)b * 256 + 8 + tid;
        }
        if (sumPtr && Kp) prevSum = *sumPtr;
    }
    const double* Dk = D + (long long)K * kDRec;
    // (update launches of the split chain read the SOLVED blocks: Y_K from WO, L_RK / L_CK in place in A)
    const double* Pg = isW ? ((UPD ? WO : W) + (long long)(K * kSB) * ldW + R * kSB)
                           : ((eSrc && panelA) ? Se + (long long)(R * kSB) * a.ld + K * kSB : A + (long long)(R * kSB) * ldA + K * kSB);
    const double* Qg = A + (long long)(C * kSB) * ldA + K * kSB;
    const int ldp = isW ? ldW : ((eSrc && panelA) ? a.ld : ldA);
    double rL[16], rW[4], rP[16], rQ[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = tid + 256 * u, rr = e >> 6, cc = e & 63;
        rL[u] = UPD ? 0.0 : Dk[e];
        rP[u] = needP ? Pg[(long long)rr * ldp + cc] : 0.0;
        rQ[u] = needQ ? Qg[(long long)rr * ldA + cc] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) rW[u] = UPD ? 0.0 : Dk[kSB * kSB + tid + 256 * u];
    // PHASE 3, S-chain rhs tails: the innovation column of block row K+1 (through K-1) and z_K of block row K
    double rZ = 0.0;
    if (tail && isW && ch.kind == 0 && tid < 2 * kSB)
        rZ = tid < kSB ? W[(long long)((K + 1) * kSB + tid) * ldW + 11] : WO[(long long)(K * kSB + tid - kSB) * ldW + 11];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = tid + 256 * u, rr = e >> 6, cc = e & 63;
        if (!UPD) s.L[rr][cc] = rL[u];
        s.P[rr][cc] = rP[u];
        s.Q[rr][cc] = rQ[u];
    }
    if (!UPD) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            s.Wd[e >> 8][(e >> 4) & 15][e & 15] = rW[u];
        }
    }
    if (tail && isW && ch.kind == 0 && tid < 2 * kSB) s.zv[tid] = rZ;
    if (solveOnly && ch.kind == 0) {  // the innovation column: rows of this block row, column 11 of the right-hand sides
        for (int e = tid; e < kSB * kQB; e += 256)
            s.Zs[e >> 4][e & 15] = ((e & 15) == 0) ? W[(long long)(K * kSB + (e >> 4)) * ldW + 11] : 0.0;
    }
    __syncthreads();
    EQF_STAMP(1);
    if (second && isW && C == K && K == nb - 1) EQF_FSTAMP(1);
    // ---- panel blocks: each wave solves its 16-row strip of P and of Q (16-column strip of the rhs block)
    if (!UPD) {
        if (needP) {
            if (isW) solveStrip<false>(&s.P[0][0], kSP, s, kQB * wv, lane);
            else solveStrip<true>(&s.P[0][0], kSP, s, kQB * wv, lane);
        }
        if (needQ && PHASE == 0) solveStrip<true>(&s.Q[0][0], kSP, s, kQB * wv, lane);
        if (solveOnly && ch.kind == 0 && wv == 0) solveStrip<false>(&s.Zs[0][0], kWP, s, 0, lane);
        __syncthreads();
    }
    EQF_STAMP(2);
    if (second && isW && C == K && K == nb - 1) EQF_FSTAMP(2);

    // Epilogue of a solved block row of right-hand sides (s.P = Y_Kp tile R): store it, add its share of the reductions, and
    // -- last block row of the E-chain -- run the innovation lift.  z = L_KpKp^-1 delta_Kp, element r at z[r * zs].
    auto rhsEpilogue = [&](const double* z, int zs) {
        for (int e = tid; e < kSB * kSB; e += 256) WO[(long long)(Kp * kSB + (e >> 6)) * ldW + R * kSB + (e & 63)] = s.P[e >> 6][e & 63];
        const double* red = a.red + (long long)b * 256;
        if (ch.kind == 0) {
            // gamma[col] += Y_K[:, col] . z_K ; the six columns after nv are hV ; column 11 is z itself (gamma[11] = 0)
            if (sumPtr) {
                double v = 0.0;
#pragma unroll 8
                for (int r = 0; r < kSB; ++r) v = fma(s.P[r][tid], z[r * zs], v);
                *sumPtr = (R * kSB + tid == 11) ? 0.0 : prevSum + v;
            }
        } else {
            // G11 += [Zt|Et]_K^T [Zt|Et]_K ; in the chain's last step the totals go straight to the innovation lift
            const bool last = embedFinish && Kp == nb - 1;
            if (sumPtr) {
                const int q0 = tid / 11, q1 = tid % 11;
                double v = 0.0;
#pragma unroll 8
                for (int r = 0; r < kSB; ++r) v = fma(s.P[r][q0], s.P[r][q1], v);
                const double tot = prevSum + v;
                *sumPtr = tot;
                if (last) s.redL[8 + tid] = tot;
            } else if (last && tid >= 128 && tid < 134) {
                s.redL[tid - 128] = red[tid - 128];
            }
            if (last) {
                __syncthreads();
                EQF_FSTAMP(3);
                updateFinishBody(a, b, s.redL);
                __syncthreads();
                EQF_FSTAMP(4);
            }
        }
    };
    if (panelA) {
        for (int e = tid; e < kSB * kSB; e += 256) A[(long long)(R * kSB + (e >> 6)) * ldA + K * kSB + (e & 63)] = s.P[e >> 6][e & 63];
    } else if (solveOnly) {
        rhsEpilogue(PHASE == 3 ? nullptr : &s.Zs[0][0], kWP);
    } else {
        // ---- trailing update of this tile: A_RC -= L_RK L_CK^T  /  R_C -= L_CK Y_K
        const double (*Am)[kSP] = isW ? s.Q : (needP ? s.P : s.Q);
        auto upd = [&](int i) {
            if (isW) acc[i] = mmTile<false, kSB>(acc[i], &Am[0][0], kSP, kQB * tr[i], &s.P[0][0], kSP, kQB * tc[i], lane, -1.0);
            else acc[i] = mmTile<true, kSB>(acc[i], &Am[0][0], kSP, kQB * tr[i], &s.Q[0][0], kSP, kQB * tc[i], lane, -1.0);
        };
        if (!diagNext && !tail) {
#pragma unroll
            for (int i = 0; i < 4; ++i) upd(i);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) Ct[(long long)(kQB * tr[i] + (lane >> 4) + 4 * q) * ldc + kQB * tc[i] + (lane & 15)] = acc[i][q];
        } else if (!diagNext) {
            // ---- PHASE 3, block column K+1: update, then solve in this launch as soon as D[K+1] is published
            if (isW && ch.kind == 0 && wv == 1) {
                // innovation column of this block row through K:  delta' = delta - L_{K+1,K} z_K   (L_{K+1,K} = s.Q)
                double t = s.zv[lane];
#pragma unroll 8
                for (int k = 0; k < kSB; ++k) t = fma(-s.Q[lane][k], s.zv[kSB + k], t);
                s.zv[lane] = t;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) upd(i);
            __syncthreads();  // every wave is done with s.P / s.Q as operands
#pragma unroll
            for (int i = 0; i < 4; ++i) stTile(acc[i], &s.P[0][0], kSP, kQB * tr[i], kQB * tc[i], lane);
            const double* Dn = D + (long long)(K + 1) * kDRec;
            if (tid == 0 && !hoWait(ch.flags + (long long)b * ch.strideF + K + 1, ch.epoch, errflag)) bad = 8;  // (timed out: flagged, no hang)
            __syncthreads();
            Lds64 sT = s;
            sT.L = s.Q;  // the factor of the diagonal block goes where L_CK was
            {
                v4i32 v[10];
                hoLoad16x10(reinterpret_cast<const char*>(Dn) + 16 * tid, v);
#pragma unroll
                for (int u = 0; u < 10; ++u) {
                    const int e = 2 * (tid + 256 * u);
                    if (e < kSB * kSB) {
                        sT.L[e >> 6][e & 63] = hoLo(v[u]);
                        sT.L[e >> 6][(e & 63) + 1] = hoHi(v[u]);
                    } else {
                        const int q = e - kSB * kSB;
                        s.Wd[q >> 8][(q >> 4) & 15][q & 15] = hoLo(v[u]);
                        s.Wd[q >> 8][(q >> 4) & 15][(q & 15) + 1] = hoHi(v[u]);
                    }
                }
            }
            __syncthreads();
            if (isW) {
                solveStrip<false>(&s.P[0][0], kSP, sT, kQB * wv, lane);
                if (ch.kind == 0 && wv == 3) solveVec64(s.zv, sT, lane);
                __syncthreads();
                rhsEpilogue(s.zv, 1);
            } else {
                solveStrip<true>(&s.P[0][0], kSP, sT, kQB * wv, lane);
                __syncthreads();
                for (int e = tid; e < kSB * kSB; e += 256) A[(long long)(R * kSB + (e >> 6)) * ldA + (K + 1) * kSB + (e & 63)] = s.P[e >> 6][e & 63];
            }
        } else {
            // ---- look-ahead: factor the freshly updated diagonal block for launch K + 1 in s.L (nobody reads it any more)
            upd(0);
            stTile(acc[0], &s.L[0][0], kSP, kQB * tr[0], kQB * tc[0], lane);
            if (wv == 0) stTile(acc[0], &s.D0[0][0], kWP, 0, 0, lane);
            if (wv == 1) factorPrologueW(s, lane);
            __syncthreads();
            EQF_STAMP(3);
            // (slot i of this wave is tile (tr[i], tc[i]): the table of factorTiles)
            auto pre = [&](int, int r, int c, f64x4& t) {
#pragma unroll
                for (int i = 1; i < 4; ++i)
                    if (tr[i] == r && tc[i] == c) {
                        upd(i);
                        t = acc[i];
                    }
            };
            // (PHASE 3 publishes the whole record at the end, write-through, instead of plain stores along the way)
            double* Dnext = D + (long long)(K + 1) * kDRec;
#ifdef EQF_STEP64_STAMPS
            factor64<false>(s, tid, &bad, pre, PHASE == 3 ? nullptr : Dnext, !second ? &g_stamps[K][8] : nullptr, realStages(ch.kind == 0 ? sDim(g.N) : eDim(g.N), kSB * (K + 1)), [] {});
#else
            factor64<false>(s, tid, &bad, pre, PHASE == 3 ? nullptr : Dnext, nullptr, realStages(ch.kind == 0 ? sDim(g.N) : eDim(g.N), kSB * (K + 1)), [] {});
#endif
            EQF_STAMP(4);
            if (PHASE == 3) {
                // hand D[K+1] to the workgroups of block column K+1 of THIS launch (eqf_handoff.hpp)
#pragma unroll
                for (int u = 0; u < 10; ++u) {
                    const int e = 2 * (tid + 256 * u);
                    double v0, v1;
                    if (e < kSB * kSB) {
                        v0 = s.L[e >> 6][e & 63];
                        v1 = s.L[e >> 6][(e & 63) + 1];
                    } else {
                        const int q = e - kSB * kSB;
                        v0 = s.Wd[q >> 8][(q >> 4) & 15][q & 15];
                        v1 = s.Wd[q >> 8][(q >> 4) & 15][(q & 15) + 1];
                    }
                    hoStore16(Dnext + e, v0, v1);
                }
                hoDrain();
                __syncthreads();
                if (tid == 0) hoPublish(ch.flags + (long long)b * ch.strideF + K + 1, ch.epoch);
            }
            EQF_STAMP(5);
        }
    }
    if (bad && errflag && tid == 0) atomicOr(errflag, bad == 8 ? kHoErrTimeout : 4);
}

}  // namespace eqf
