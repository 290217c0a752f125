// Tile-local kernels for Sigma 2-D block-partitioned over a process grid (BASELINE configs[4]): every rank
// owns (3 nI x 3 nJ) tiles of the landmark-landmark part of Sigma in ITS OWN device memory (caller-owned buffers, e.g. torch
// tensors); the exchange schedule lives above the C ABI (eqf_vio_amd/tiled.py, torch.distributed over RCCL).  fp64.
//
//   k_tile_propagate   one structured Riccati step of one tile (VIOFilter.cpp:188-189 with F = [[F_bb, 0], [L, D]]):
//                        Sigma'_IJ = (D_I Sigma_IJ + L_I Sigma_bJ) D_J^T + (L_I Sigma_bb + D_I Sigma_Ib) L_J^T + T (B_I R B_J^T [+ p I])
//                      one thread per 3x3 block, a workgroup = 16 x 16 landmark pairs; the rows G_i = L_i Sigma_bb + D_i Sigma_ib of
//                      its 16 row landmarks are formed once per workgroup in LDS.  Needs nothing from any other rank: the
//                      11 x n base panel and the per-landmark blocks are replicated.
//   k_tile_gemm_tn     C += alpha A^T B for A (k x m), B (k x n) row-major: every dense product of the distributed update -- the trailing
//                      updates of the two factorisations, the right-hand sides, the downdate Sigma - K C Sigma = Sigma - Y^T Y
//                      (VIOFilter.cpp:297) and the reductions -- 128 x 128 outputs per workgroup on v_mfma_f64_16x16x4_f64, operands
//                      double-buffered through LDS in chunks of 16 rows, XCD-contiguous tile order.
//   k_tile_potrf       Cholesky of ONE n x n diagonal block of the distributed factorisation (S_kk / the Schur complement of
//                      Sigma_e, VIOFilter.cpp:276, EqFMatrices.cpp:239), in place, by one workgroup in 64-wide block columns with the
//                      building blocks of the single-GPU path (eqf_chol64.hpp: factor64, solveStrip, mmTile).  Leaves, per 64-wide
//                      block column, the diagonal-factor record the panel solves use (L_jj + the inverses of its four 16 x 16
//                      diagonal blocks, kDRec doubles).  The diagonal block is the serial part of a block column of the distributed
//                      algorithm: one workgroup is what it offers.
//   k_tile_trsm        the panel blocks and the block row of right-hand sides against that factor:  B <- B L^-T (right; B is m x n)
//                      or B <- L^-1 B (left; B is n x m); one workgroup per 64 rows (columns) of B, register-chained MFMA
//                      substitution per 64 x 64 block, the earlier blocks of the strip applied first.
#pragma once
#include <type_traits>
#include "eqf_chol64.hpp"

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

// ---- C (m x n, ldc) += alpha A^T B with A (k x m, lda), B (k x n, ldb), all row-major (every dense product of the distributed
// update has this form once the factorisation keeps block ROWS: trailing updates U_ki^T U_kj, right-hand sides U_ki^T Y_kt, the
// downdate Y_kI^T Y_kJ, the reductions Yn_k^T [Y_k | Yn_k]).  fp64 on v_mfma_f64_16x16x4_f64.
// A 16x16x4 f64 MFMA occupies a SIMD's matrix pipe for 64 cycles (78.6 TFLOP/s over 1024 SIMDs), so the kernel is laid out for the
// MEMORY side: a 64 x 64 output tile asks for 8 flop per staged operand byte = 9.8 TB/s of L2 traffic at peak -- more than the
// fabric delivers; a 128 x 128 tile halves that.  One workgroup = 128 x 128 outputs, 4 waves of 64 x 64 (4 x 4 MFMA tiles, 128
// accumulator registers), K in chunks of 16 rows: the next chunk's 32 KB travel global -> registers while the 64 MFMAs of the
// current one (4096 matrix-pipe cycles per wave) run from LDS, one barrier per chunk (double-buffered LDS, row pitch 144 doubles:
// rows 128 B apart modulo the 256 B bank window, so the four k rows of an operand fetch fall on disjoint banks).
// Tile order: the linear workgroup index is dealt to the 8 XCDs round-robin by the hardware (L % 8); it is remapped so that each XCD
// walks a CONTIGUOUS range of tiles in a grouped order (8 tile rows at a time, column by column): the panels of A and B a XCD's 32
// resident workgroups read at any time are a handful of 128-column slices that stay in its 4 MB L2.
// mask (rb > 0): C is the A-part of a block-cyclic local matrix whose strictly-lower blocks are never read (upper block rows of a
// Cholesky): tiles whose rows all lie in blocks I > the blocks J of all their columns are skipped; row r belongs to global block
// (rblk0 + r / rb) * Pr + pr, column c to (cblk0 + c / cb) * Pc + pc.
struct GemmMask {
    int rb, cb;          // rows / columns per block (0 = no mask)
    int rblk0, Pr, pr;   // local row block index of C's first row, process grid rows, this rank's grid row
    int cblk0, Pc, pc;
};
constexpr int kGemmTile = 128, kGemmKC = 16, kGemmPitch = 144;
constexpr int kGemmLdsBytes = 2 * 2 * kGemmKC * kGemmPitch * 8;
// Masked launches walk ONLY the tiles the mask leaves, in the same XCD-contiguous grouped order as unmasked ones (round 4; until then they
// were dealt round-robin over the whole rectangle, which put each XCD on one tile row per group -- every B panel fetched by all eight L2s).
// The host lays the walk out (a mask is a staircase: row ti is active from column tile first[ti] on, first[] nondecreasing) and hands it over
// by value in the kernel arguments: first[] per tile row and the number of active tiles in front of every group of kGemmGroup rows.
constexpr int kGemmGroup = 8, kGemmPlanRows = 256;
struct GemmPlan {
    int tact;                                        // active tiles (0: no plan -- unmasked launch, or more than kGemmPlanRows tile rows)
    int before[kGemmPlanRows / kGemmGroup + 1];      // active tiles in the groups before group g
    int first[kGemmPlanRows];                        // first active column tile of tile row ti (tn: none)
};
// tile (ti, tj) holds something the mask keeps
EQF_DI bool gemmTileActive(const GemmMask& mk, int ti, int tj, int n) {
    const int I0 = ti * kGemmTile, J0 = tj * kGemmTile;
    const int Ilo = (mk.rblk0 + I0 / mk.rb) * mk.Pr + mk.pr;
    const int Jend = (J0 + kGemmTile < n ? J0 + kGemmTile : n) - 1;
    const int Jhi = (mk.cblk0 + Jend / mk.cb) * mk.Pc + mk.pc;
    return Ilo <= Jhi;
}
inline void gemmMakePlan(const GemmMask& mk, int tm, int tn, int n, GemmPlan* pl) {
    pl->tact = 0;
    if (mk.rb <= 0 || tm > kGemmPlanRows) return;
    int total = 0;
    for (int ti = 0; ti < tm; ++ti) {
        if (ti % kGemmGroup == 0) pl->before[ti / kGemmGroup] = total;
        int c = ti > 0 ? pl->first[ti - 1] : 0;  // (nondecreasing in ti)
        while (c < tn && !gemmTileActive(mk, ti, c, n)) ++c;
        pl->first[ti] = c;
        total += tn - c;
    }
    pl->before[(tm + kGemmGroup - 1) / kGemmGroup] = total;
    pl->tact = total;
}
// DIRECT (round 6, a build of its own): in the straight-line chunks the next chunk's operand rows go global -> LDS without passing through
// registers (global_load_lds_dwordx4: one instruction of a wave lands one 1 KB row at that row's padded LDS address) -- no staging stores,
// 32 registers less in flight; the wave waits for its own copies (vmcnt) in front of the chunk's barrier.
typedef const void __attribute__((address_space(1)))* gemm_gptr_t;
typedef void __attribute__((address_space(3)))* gemm_lptr_t;
template <bool DIRECT>
__global__ __launch_bounds__(256, 2) void k_tile_gemm_tn(double* C, int ldc, int m, int n, const double* A, int lda, const double* B, int ldb, int k,
    double alpha, GemmMask mk, int tm, int tn, GemmPlan pl) {
    extern __shared__ __attribute__((aligned(16))) double sGemm[];  // [2][2][KC][pitch]: buffer, operand
    // ---- tile of this workgroup (XCD-contiguous grouped order: the hardware deals workgroup L to XCD L % 8; XCD x walks tiles [x per, (x + 1) per))
    constexpr int G = kGemmGroup;
    const int L = blockIdx.x;
    int ti, tj;
    if (pl.tact > 0) {
        const int per = (pl.tact + 7) / 8;
        int id = (L & 7) * per + (L >> 3);
        if ((L >> 3) >= per || id >= pl.tact) return;
        int g = 0;
        while (pl.before[g + 1] <= id) ++g;
        id -= pl.before[g];
        const int first = g * G, gsz = min(tm - first, G);
        // inside the group: column by column from the first column its top row keeps; a column holds the rows whose staircase has reached it
        // (the top `a` rows: first[] is nondecreasing), from the last row's first column on all gsz of them
        tj = pl.first[first];
        const int cfull = pl.first[first + gsz - 1];
        ti = -1;
        while (tj < cfull) {
            int a = 0;
            for (int r = 0; r < gsz; ++r) a += pl.first[first + r] <= tj;
            if (id < a) {
                ti = first + id;
                break;
            }
            id -= a;
            ++tj;
        }
        if (ti < 0) {
            tj = cfull + id / gsz;
            ti = first + id % gsz;
        }
    } else {
        // (a mask without a plan -- more than kGemmPlanRows tile rows --: round-robin over the rectangle, inactive tiles leave at once)
        const int T = tm * tn, per = (T + 7) / 8;
        const int id = mk.rb > 0 ? L : (L & 7) * per + (L >> 3);
        if ((L >> 3) >= per || id >= T) return;
        const int grp = id / (G * tn), first = grp * G, gsz = min(tm - first, G);
        ti = first + (id % (G * tn)) % gsz;
        tj = (id % (G * tn)) / gsz;
        if (mk.rb > 0 && !gemmTileActive(mk, ti, tj, n)) return;
    }
    const int I0 = ti * kGemmTile, J0 = tj * kGemmTile;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wi = wv >> 1, wj = wv & 1, lr = lane & 15, lk = lane >> 4;
    // 16-row / 16-column sub-tiles of this wave that hold anything (ragged edges, narrow products)
    const int mu = max(0, min(4, (m - I0 - 64 * wi + 15) / 16)), nv = max(0, min(4, (n - J0 - 64 * wj + 15) / 16));
    f64x4 acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[u][v][q] = 0.0;
    // staging: wave wv takes rows wv, wv + 4, wv + 8, wv + 12 of the chunk, lane l the columns 2 l, 2 l + 1: one load instruction of a wave
    // reads one whole 1 KB row of the operand slice (fully coalesced), and writes it to LDS as one conflict-free row
    const int sc = 2 * lane;
    const bool fullA = I0 + kGemmTile <= m, fullB = J0 + kGemmTile <= n;
    typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));  // 8-byte aligned pair: views may start at odd columns
    double pa[8], pb[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = k0 + wv + 4 * r;
            const double* ar = A + (long long)row * lda + I0 + sc;
            const double* br = B + (long long)row * ldb + J0 + sc;
            if (row < k && fullA) {
                const f64x2u v = *reinterpret_cast<const f64x2u*>(ar);
                pa[2 * r] = v.x; pa[2 * r + 1] = v.y;
            } else {
                pa[2 * r] = (row < k && I0 + sc < m) ? ar[0] : 0.0;
                pa[2 * r + 1] = (row < k && I0 + sc + 1 < m) ? ar[1] : 0.0;
            }
            if (row < k && fullB) {
                const f64x2u v = *reinterpret_cast<const f64x2u*>(br);
                pb[2 * r] = v.x; pb[2 * r + 1] = v.y;
            } else {
                pb[2 * r] = (row < k && J0 + sc < n) ? br[0] : 0.0;
                pb[2 * r + 1] = (row < k && J0 + sc + 1 < n) ? br[1] : 0.0;
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* da = sGemm + ((buf * 2 + 0) * kGemmKC + wv + 4 * r) * kGemmPitch + sc;
            double* db = sGemm + ((buf * 2 + 1) * kGemmKC + wv + 4 * r) * kGemmPitch + sc;
            *reinterpret_cast<f64x2*>(da) = f64x2{pa[2 * r], pa[2 * r + 1]};
            *reinterpret_cast<f64x2*>(db) = f64x2{pb[2 * r], pb[2 * r + 1]};
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    const int nc = (k + kGemmKC - 1) / kGemmKC;
    auto fetchFull = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long row = k0 + wv + 4 * r;
            const f64x2u va = *reinterpret_cast<const f64x2u*>(A + row * lda + I0 + sc);
            const f64x2u vb = *reinterpret_cast<const f64x2u*>(B + row * ldb + J0 + sc);
            pa[2 * r] = va.x; pa[2 * r + 1] = va.y;
            pb[2 * r] = vb.x; pb[2 * r + 1] = vb.y;
        }
    };
    // operands of k step s of the chunk in LDS buffer `buf`: four 16-row slices of A and of B, one double per lane each
    auto operands = [&](int buf, int s, double* av, double* bv) {
        const double* sa = sGemm + (buf * 2 + 0) * kGemmKC * kGemmPitch + 64 * wi + lr + (4 * s + lk) * kGemmPitch;
        const double* sb = sGemm + (buf * 2 + 1) * kGemmKC * kGemmPitch + 64 * wj + lr + (4 * s + lk) * kGemmPitch;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            av[u] = sa[16 * u];
            bv[u] = sb[16 * u];
        }
    };
    auto mfma16 = [&](const double* av, const double* bv, auto fullTag) {
        constexpr bool FULL = decltype(fullTag)::value;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (FULL || (u < mu && v < nv)) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[v], acc[u][v], 0, 0, 0);
    };
    // Whole tiles: chunks c with c + 1 entirely inside k run a straight-line body (no predicate anywhere: one basic block per chunk, in which
    // the scheduler places the global loads, the LDS reads and the LDS writes in the shadow of the matrix instructions -- a 16x16x4 f64 MFMA
    // holds the pipe for 64 cycles, fifteen issue slots each; it also takes the staging stores and the barrier in front of the last dozen
    // MFMAs).  The predicated body below tests (u < mu && v < nv) per MFMA and every fetched element: dozens of scalar branches per chunk
    // during which the wave feeds nothing to the matrix pipe -- 50.7 -> 56.0 TFLOP/s at 12000 x 12000 x 750 (MFMA-busy 0.65 -> 0.74).
    // Measured and not kept (same rate): two chunks in flight (the loads are not late), the loop rotated around the barrier with the next
    // chunk's first operands read behind it.
    const bool whole = fullA && fullB && mu == 4 && nv == 4;
    const int nFast = whole ? max(0, k / kGemmKC - 1) : 0;
    int c = 0;
    if constexpr (DIRECT) {
        for (; c < nFast; ++c) {
            const int nb_ = (c + 1) & 1;  // (that buffer was read in chunk c - 1: the barrier at its end has been passed)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long row = (c + 1) * kGemmKC + wv + 4 * r;
                __builtin_amdgcn_global_load_lds((gemm_gptr_t)(A + row * lda + I0 + sc),
                    (gemm_lptr_t)(sGemm + ((nb_ * 2 + 0) * kGemmKC + wv + 4 * r) * kGemmPitch), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gemm_gptr_t)(B + row * ldb + J0 + sc),
                    (gemm_lptr_t)(sGemm + ((nb_ * 2 + 1) * kGemmKC + wv + 4 * r) * kGemmPitch), 16, 0, 0);
            }
#pragma unroll
            for (int s_ = 0; s_ < kGemmKC / 4; ++s_) {
                double av[4], bv[4];
                operands(c & 1, s_, av, bv);
                mfma16(av, bv, std::true_type{});
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        for (; c < nFast; ++c) {
            fetchFull((c + 1) * kGemmKC);
#pragma unroll
            for (int s_ = 0; s_ < kGemmKC / 4; ++s_) {
                double av[4], bv[4];
                operands(c & 1, s_, av, bv);
                mfma16(av, bv, std::true_type{});
            }
            stage((c + 1) & 1);
            __syncthreads();
        }
    }
    for (; c < nc; ++c) {
        if (c + 1 < nc) fetch((c + 1) * kGemmKC);  // in flight during this chunk's MFMAs
#pragma unroll
        for (int s_ = 0; s_ < kGemmKC / 4; ++s_) {
            double av[4], bv[4];
            operands(c & 1, s_, av, bv);
            mfma16(av, bv, std::false_type{});
        }
        if (c + 1 < nc) stage((c + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int R = I0 + 64 * wi + 16 * u + lk + 4 * q, Cc = J0 + 64 * wj + 16 * v + lr;
                if (R < m && Cc < n) {
                    // C += alpha acc as a fire-and-forget global_atomic_add_f64: the wave does not wait for C to arrive (a read-modify-write
                    // epilogue is a load -> fma -> store chain per element while the tile's matrix-pipe slot sits empty: 5-10 % of the kernel
                    // by ablation), the L2 does the addition.  Every element of C has exactly ONE writer in a launch, so the result is
                    // deterministic, and for alpha = +-1 (every product of the update) it is the same rounding as fma(alpha, acc, C).
                    unsafeAtomicAdd(C + (long long)R * ldc + Cc, alpha * acc[u][v][q]);
                }
            }
}


// ---- C (n x n, symmetric in exact arithmetic, blocks of rb): every element below the BLOCK diagonal <- its mirror image,
// C[r][c] = C[c][r] for r / rb > c / rb.  Completes a masked k_tile_gemm_tn on a rank whose local matrix is symmetric (square process
// grid, diagonal rank: the partitioned downdate then costs half the flops).  grid = (ceil(n / 64), ceil(n / 64)), block = 256; the
// transpose goes through LDS so that both the reads and the writes are row-contiguous.
__global__ __launch_bounds__(256) void k_tile_mirror(double* C, int ld, int n, int rb) {
    const int R0 = blockIdx.y * 64, C0 = blockIdx.x * 64;
    if ((R0 + 63 < n ? R0 + 63 : n - 1) / rb <= C0 / rb) return;  // no element of this tile is below the block diagonal
    __shared__ double sT[64][65];
    const int tid = threadIdx.x;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int i = e >> 6, j = e & 63;  // source element (C0 + i, R0 + j)
        sT[i][j] = (C0 + i < n && R0 + j < n) ? C[(long long)(C0 + i) * ld + R0 + j] : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
        const int i = e >> 6, j = e & 63, r = R0 + i, c = C0 + j;
        if (r < n && c < n && r / rb > c / rb) C[(long long)r * ld + c] = sT[j][i];
    }
}

// ---- n x n block, lower triangle, row-major with leading dimension ld: A <- L (A = L L^T); drec[ceil(n / 64)][kDRec].
// One workgroup of 256 threads, LDS = Step64Lds.  info: or-ed with 1 if a pivot is not positive.
__global__ __launch_bounds__(256) void k_tile_potrf(double* A, int ld, int n, double* drec, int* info) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemT[];
    const Lds64 s = ldsFull(smemT);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nb = (n + kSB - 1) / kSB;
    int bad = 0;
    // block (rb, cb) of A <-> LDS; rows / columns past n read as the identity (diagonal blocks) or zero
    auto loadBlock = [&](double (*dst)[kSP], int rb, int cb, bool diag) {
        for (int e = tid; e < kSB * kSB; e += 256) {
            const int r = e >> 6, c = e & 63, gr = kSB * rb + r, gc = kSB * cb + c;
            double v = 0.0;
            if (gr < n && gc < n) v = (!diag || c <= r) ? A[(long long)gr * ld + gc] : 0.0;
            else if (diag && r == c) v = 1.0;
            dst[r][c] = v;
        }
    };
    auto storeBlock = [&](const double (*src)[kSP], int rb, int cb, bool diag) {
        for (int e = tid; e < kSB * kSB; e += 256) {
            const int r = e >> 6, c = e & 63, gr = kSB * rb + r, gc = kSB * cb + c;
            if (gr < n && gc < n && (!diag || c <= r)) A[(long long)gr * ld + gc] = src[r][c];
        }
    };
    for (int kb = 0; kb < nb; ++kb) {
        loadBlock(s.L, kb, kb, true);
        __syncthreads();
        factorPrologue(s, tid);
        __syncthreads();
        factor64(s, tid, &bad, drec + (long long)kb * kDRec, nullptr, realStages(n, kSB * kb));
        __syncthreads();
        storeBlock(s.L, kb, kb, true);
        // panel: A_rb,kb <- A_rb,kb L_kk^-T
        for (int rb = kb + 1; rb < nb; ++rb) {
            loadBlock(s.P, rb, kb, false);
            __syncthreads();
            solveStrip<true>(&s.P[0][0], kSP, s, kQB * wv, lane);
            __syncthreads();
            storeBlock(s.P, rb, kb, false);
            __syncthreads();
        }
        __threadfence_block();  // (the panel is read back below by other threads of this workgroup)
        __syncthreads();
        // trailing update of the lower triangle: A_rb,cb -= L_rb,kb L_cb,kb^T
        for (int rb = kb + 1; rb < nb; ++rb) {
            loadBlock(s.P, rb, kb, false);
            for (int cb = kb + 1; cb <= rb; ++cb) {
                loadBlock(s.Q, cb, kb, false);
                __syncthreads();
                f64x4 acc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int gr = kSB * rb + kQB * wv + (lane >> 4) + 4 * q, gc = kSB * cb + kQB * i + (lane & 15);
                        acc[i][q] = (gr < n && gc < n) ? A[(long long)gr * ld + gc] : 0.0;
                    }
                    acc[i] = mmTile<true, kSB>(acc[i], &s.P[0][0], kSP, kQB * wv, &s.Q[0][0], kSP, kQB * i, lane, -1.0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int gr = kSB * rb + kQB * wv + (lane >> 4) + 4 * q, gc = kSB * cb + kQB * i + (lane & 15);
                        if (gr < n && gc < n && gc <= gr) A[(long long)gr * ld + gc] = acc[i][q];  // (lower triangle only)
                    }
                }
                __syncthreads();
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (bad && info && tid == 0) atomicOr(info, 1);
}

// ---- blocked potrf for larger diagonal blocks, one launch per phase of a 64-wide block column kb (eqf_tile_potrf drives them):
//   k_tile_potrf on the 64 x 64 diagonal block (one workgroup)  ->  k_tile_trsm<right> on the panel below it (one workgroup per 64
//   rows)  ->  k_tile_potrf_trail: the lower triangle of the trailing matrix, A_rb,cb -= L_rb,kb L_cb,kb^T, one workgroup per 64 x 64
//   tile.  The single-workgroup kernel above spends its time in exactly this trailing update (n^3 / 3 flops on one CU: 2.2 ms at
//   n = 750); spread over the chip a block column costs three short launches.  grid = t (t + 1) / 2 with t = nb - kb - 1.
constexpr int kTrailLdsBytes = 2 * kSB * kSP * 8;
__global__ __launch_bounds__(256) void k_tile_potrf_trail(double* A, int ld, int n, int kb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemTr[];  // two 64 x kSP blocks (66 KB: dynamic)
    double (*sP)[kSP] = reinterpret_cast<double (*)[kSP]>(smemTr);
    double (*sQ)[kSP] = sP + kSB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // tile (r, c), r >= c, of the trailing lower triangle, row-major enumeration
    int r = 0, idx = blockIdx.x;
    while (idx >= r + 1) {
        idx -= r + 1;
        ++r;
    }
    const int rb = kb + 1 + r, cb = kb + 1 + idx;
    for (int e = tid; e < kSB * kSB; e += 256) {
        const int rr = e >> 6, cc = e & 63, gc = kSB * kb + cc;
        const int gr = kSB * rb + rr, gq = kSB * cb + rr;
        sP[rr][cc] = (gr < n && gc < n) ? A[(long long)gr * ld + gc] : 0.0;
        sQ[rr][cc] = (gq < n && gc < n) ? A[(long long)gq * ld + gc] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f64x4 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gr = kSB * rb + kQB * wv + (lane >> 4) + 4 * q, gc = kSB * cb + kQB * i + (lane & 15);
            acc[q] = (gr < n && gc < n) ? A[(long long)gr * ld + gc] : 0.0;
        }
        acc = mmTile<true, kSB>(acc, &sP[0][0], kSP, kQB * wv, &sQ[0][0], kSP, kQB * i, lane, -1.0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gr = kSB * rb + kQB * wv + (lane >> 4) + 4 * q, gc = kSB * cb + kQB * i + (lane & 15);
            if (gr < n && gc < n && gc <= gr) A[(long long)gr * ld + gc] = acc[q];
        }
    }
}

// ---- B <- B L^-T (right = 1: B is m x n, a workgroup owns 64 rows) or B <- L^-1 B (right = 0: B is n x m, a workgroup owns 64
// columns), L = lower triangle of A with the records of k_tile_potrf.  grid = ceil(m / 64), block = 256, LDS = Step64Lds.
__global__ __launch_bounds__(256) void k_tile_trsm(const double* A, int ld, int n, const double* drec, double* B, int ldb, int m, int right) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemT[];
    const Lds64 s = ldsFull(smemT);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nb = (n + kSB - 1) / kSB, s0 = blockIdx.x * kSB;  // first row (right) / column (left) of the strip
    // strip block kb <-> (row, column) of B: right: rows s0.., columns 64 kb.. ; left: rows 64 kb.., columns s0..
    auto bAt = [&](int kb, int r, int c, long long* idx) -> bool {
        const int gr = right ? s0 + r : kSB * kb + r, gc = right ? kSB * kb + c : s0 + c;
        *idx = (long long)gr * ldb + gc;
        return right ? (gr < m && gc < n) : (gr < n && gc < m);
    };
    // The strip is a chain of nb (nb + 1) / 2 block products, each behind two 32 KB operand blocks: the blocks of product (kb, j + 1) -- or
    // (kb + 1, 0) -- travel global -> REGISTERS while product (kb, j) runs from LDS (round 4: a step took 10 us, nearly all of it the
    // exposed latency of the loads; a block-row solve of the partitioned filter is one such chain per 64 columns, 0.77 ms at n = 500).
    double pP[16], pQ[16];
    auto issue = [&](int kb, int j) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u, r = e >> 6, c = e & 63;
            long long idx;
            pP[u] = bAt(j, r, c, &idx) ? B[idx] : 0.0;
            const int gr = kSB * kb + r, gc = kSB * j + c;
            pQ[u] = (gr < n && gc < n) ? A[(long long)gr * ld + gc] : 0.0;
        }
    };
    auto toLds = [&]() {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = tid + 256 * u;
            s.P[e >> 6][e & 63] = pP[u];
            s.Q[e >> 6][e & 63] = pQ[u];
        }
    };
    for (int kb = 0; kb < nb; ++kb) {
        // the strip's block kb in the accumulator layout (wave wv: rows 16 wv.., four 16-column tiles)
        f64x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                long long idx;
                acc[i][q] = bAt(kb, kQB * wv + (lane >> 4) + 4 * q, kQB * i + (lane & 15), &idx) ? B[idx] : 0.0;
            }
        if (kb > 0) issue(kb, 0);  // (the strip's solved blocks j < kb were written by this workgroup: fenced at the end of their step)
        for (int j = 0; j < kb; ++j) {
            // P <- the strip's solved block j ; Q <- L_kb,j
            toLds();
            __syncthreads();
            if (j + 1 < kb) issue(kb, j + 1);
            if (right) {  // X_kb -= X_j L_kb,j^T
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mmTile<true, kSB>(acc[i], &s.P[0][0], kSP, kQB * wv, &s.Q[0][0], kSP, kQB * i, lane, -1.0);
            } else {  // Y_kb -= L_kb,j Y_j
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mmTile<false, kSB>(acc[i], &s.Q[0][0], kSP, kQB * wv, &s.P[0][0], kSP, kQB * i, lane, -1.0);
            }
            __syncthreads();
        }
        // L_kb,kb and its inverse blocks
        const double* Dk = drec + (long long)kb * kDRec;
        for (int e = tid; e < kSB * kSB; e += 256) s.L[e >> 6][e & 63] = Dk[e];
        for (int e = tid; e < 4 * kQB * kQB; e += 256) s.Wd[e >> 8][(e >> 4) & 15][e & 15] = Dk[kSB * kSB + e];
#pragma unroll
        for (int i = 0; i < 4; ++i) stTile(acc[i], &s.P[0][0], kSP, kQB * wv, kQB * i, lane);
        __syncthreads();
        if (right) solveStrip<true>(&s.P[0][0], kSP, s, kQB * wv, lane);
        else solveStrip<false>(&s.P[0][0], kSP, s, kQB * wv, lane);
        __syncthreads();
        for (int e = tid; e < kSB * kSB; e += 256) {
            long long idx;
            if (bAt(kb, e >> 6, e & 63, &idx)) B[idx] = s.P[e >> 6][e & 63];
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// The downdate's product on the INTEGER matrix pipe (round 6; opt-in, eqf_tf_set_option "downdate_slices"): C -= A^T B for A (k x m), B (k x n)
// row-major (VIOFilter.cpp:297 as Sigma - Y^T Y), with every column of A and B scaled by a power of two and cut into S signed 7-bit slices
// (int8, |q| <= 64): the S (S + 1) / 2 slice pairs (ta, tb) with ta + tb < S are multiplied on v_mfma_i32_32x32x32_i8 -- int32 accumulation is
// EXACT (k * S * 64^2 < 2^31 for k <= 70 000) -- and the S accumulators of an element (one per ta + tb) are recombined in fp64.  What is lost
// is only what the slices do not hold: the bits of an entry below 2^-(6 + 7 (S - 1)) of its column's largest entry (S = 5: 34 bits, 6: 41, 7:
// 48).  Sigma itself stays fp64 in memory, and unlike an fp32 product nothing is lost in the accumulation -- the two things DESIGN.md section
// 2 measured fp32 to fail on.  What the filter needs was measured on the bench stream with this truncation put into the fp64 restatement
// (scripts/slice_precision_study.py, profiles/r06_slice_precision_study*.txt) and with this kernel in the filter (profiles/r06_i8_downdate_error.txt):
// the truncation alone would allow S = 5 (5e-6), but the slice pairs the product drops (ta + tb >= S: products of the LOWER slices, of the
// truncation's size) add up coherently over Y's correlated columns -- S = 5: 1.4e-4 .. 9e-4, misses; S = 6: 2e-6 .. 6e-5; S = 7: 1e-8 .. 8e-8.
//   k_i8_colexp   per column the exponent of its largest |entry| (frexp), as an atomicMax over row slabs (expo zeroed by k_i8_zero; stored + 2048)
//   k_i8_split<S> the slices in MFMA FRAGMENT order: for a 32-column tile ct, a 32-row chunk kc and slice t one 1 KB block whose lane l holds
//                 column ct * 32 + (l & 31), rows kc * 32 + 16 (l >> 5) .. + 16  -- the operand layout of v_mfma_i32_32x32x32_i8; block index
//                 ((ct * nKc + kc) * S + t).  Rows past k and columns past m are zero.
//   k_i8_gemm<S>  512 threads = 8 waves as 4 x 2, workgroup tile 128 (rows of C) x 64, a wave owns ONE 32 x 32 MFMA tile with S accumulators
//                 (two waves per SIMD); a chunk's 4 S + 2 S fragment blocks go global -> LDS directly (global_load_lds_dwordx4: the global
//                 layout IS the LDS image), three LDS buffers (chunk kc + 2 in flight while kc is multiplied), one raw s_barrier per chunk
//                 behind a counted vmcnt.  mk.rb > 0: the first maskCols columns of C are masked as in k_tile_gemm_tn (GemmMask: tiles entirely
//                 below the block diagonal are skipped -- the downdate's symmetric local matrix, which k_tile_mirror completes, and the
//                 factorisations' upper block rows); columns from maskCols on (right-hand sides) are always formed.  Epilogue: C[i][j] += alpha 2^(eA[i] + eB[j]) sum_d acc_d 2^-(12 + 7 d).
// Measured (scripts/micro/i8_split_gemm.hip, profiles/r06_i8_split_gemm_v2.txt): S = 5: 84 - 108 fp64-equivalent TFLOP/s at the downdate's
// shapes (1.3 - 1.6 POPS of int8) against 51 - 57 for k_tile_gemm_tn in the same run.
typedef int i8v4 __attribute__((ext_vector_type(4)));
typedef int i8v16 __attribute__((ext_vector_type(16)));
constexpr int kI8Bits = 7;

__global__ __launch_bounds__(256) void k_i8_colexp(const double* X, int K, int M, int ld, int* expo) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), kq = threadIdx.x >> 6;
    const int k0 = blockIdx.y * 512, k1 = min(k0 + 512, K);
    double mx = 0.0;
    if (c < M)
        for (int k = k0 + kq; k < k1; k += 4) mx = fmax(mx, fabs(X[(long long)k * ld + c]));
    __shared__ double sm[4][64];
    sm[kq][threadIdx.x & 63] = mx;
    __syncthreads();
    if (kq == 0 && c < M) {
        mx = fmax(fmax(sm[0][threadIdx.x], sm[1][threadIdx.x]), fmax(sm[2][threadIdx.x], sm[3][threadIdx.x]));
        if (mx > 0.0) {
            int e = 0;
            frexp(mx, &e);  // mx = f 2^e, f in [0.5, 1): |x| 2^-e < 1
            atomicMax(expo + c, e + 2048);
        }
    }
}

__global__ __launch_bounds__(256) void k_i8_zero(int* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0;
}

template <int S>
__global__ __launch_bounds__(256) void k_i8_split(const double* X, int K, int M, int ld, const int* expo, signed char* out, int nKc) {
    const int ct = blockIdx.x, kc = blockIdx.y * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (kc >= nKc) return;
    const int c = ct * 32 + (l & 31), k0 = kc * 32 + (l >> 5) * 16;
    const int es = c < M ? expo[c] : 0;
    const double sc = (c < M && es > 0) ? ldexp(1.0, -(es - 2048)) : 0.0;
    signed char q[S][16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = k0 + j;
        double r = (c < M && k < K) ? X[(long long)k * ld + c] * sc : 0.0;  // |r| < 1, exact (power-of-two scale)
        double w = 64.0, wi = 0.015625;                                    // 2^6, then 2^13, 2^20, ...
#pragma unroll
        for (int t = 0; t < S; ++t) {
            const double qq = rint(r * w);  // |qq| <= 64
            q[t][j] = (signed char)(int)qq;
            r = fma(-qq, wi, r);  // exact (wi = 1 / w, a power of two)
            w *= 128.0;
            wi *= 0.0078125;
        }
    }
#pragma unroll
    for (int t = 0; t < S; ++t) {
        int4 v;
        v.x = (unsigned char)q[t][0] | ((unsigned char)q[t][1] << 8) | ((unsigned char)q[t][2] << 16) | ((unsigned)(unsigned char)q[t][3] << 24);
        v.y = (unsigned char)q[t][4] | ((unsigned char)q[t][5] << 8) | ((unsigned char)q[t][6] << 16) | ((unsigned)(unsigned char)q[t][7] << 24);
        v.z = (unsigned char)q[t][8] | ((unsigned char)q[t][9] << 8) | ((unsigned char)q[t][10] << 16) | ((unsigned)(unsigned char)q[t][11] << 24);
        v.w = (unsigned char)q[t][12] | ((unsigned char)q[t][13] << 8) | ((unsigned char)q[t][14] << 16) | ((unsigned)(unsigned char)q[t][15] << 24);
        reinterpret_cast<int4*>(out)[(((size_t)ct * nKc + kc) * S + t) * 64 + l] = v;
    }
}

typedef const void __attribute__((address_space(1)))* i8gptr_t;
typedef void __attribute__((address_space(3)))* i8lptr_t;
template <int S>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_i8_gemm(const signed char* As, const signed char* Bs,
    const int* eA, const int* eB, double* C, int M, int N, int ldc, int nKc, double alpha, GemmMask mk, int maskCols) {
    constexpr int kFrag = 6 * S, kPerWave = (kFrag + 7) / 8, kSlots = kPerWave * 8;  // (every wave issues the same number of copies: one vmcnt)
    __shared__ int4 sm[3][kSlots * 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv >> 1, wc = wv & 1;
    const int ctA0 = blockIdx.y * 4, ctB0 = blockIdx.x * 2;
    if (mk.rb > 0 && ctB0 * 32 + 63 < maskCols) {  // (uniform) a tile inside the masked columns, entirely below the block diagonal: nobody reads it
        const int Ilo = (mk.rblk0 + (ctA0 * 32) / mk.rb) * mk.Pr + mk.pr;
        const int Jhi = (mk.cblk0 + (ctB0 * 32 + 63) / mk.cb) * mk.Pc + mk.pc;
        if (Ilo > Jhi) return;
    }
    const int4* gA = reinterpret_cast<const int4*>(As);
    const int4* gB = reinterpret_cast<const int4*>(Bs);
    auto stage = [&](int kc, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < kPerWave; ++j) {
            const int slot = wv + 8 * j;
            const int blk = slot < kFrag ? slot : 0;  // (pad slots re-read block 0 into LDS nobody looks at)
            const int4* src;
            if (blk < 4 * S) {
                const int ct = blk / S, t = blk - ct * S;
                src = gA + (((size_t)(ctA0 + ct) * nKc + kc) * S + t) * 64 + lane;
            } else {
                const int b2 = blk - 4 * S, ct = b2 / S, t = b2 - ct * S;
                src = gB + (((size_t)(ctB0 + ct) * nKc + kc) * S + t) * 64 + lane;
            }
            __builtin_amdgcn_global_load_lds((i8gptr_t)src, (i8lptr_t)&sm[buf][slot * 64], 16, 0, 0);
        }
    };
    i8v16 acc[S];
#pragma unroll
    for (int d = 0; d < S; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0;
    stage(0, 0);
    if (nKc > 1) stage(1, 1);
    // chunk 0 complete (the older kPerWave of this wave's copies), then everybody's: the barrier
    if (nKc > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPerWave) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kc = 0; kc < nKc; ++kc) {
        const int buf = kc % 3;
        if (kc + 2 < nKc) stage(kc + 2, (kc + 2) % 3);  // (that buffer was read in iteration kc - 1: the barrier at its end has been passed)
        i8v4 a[S], b[S];
#pragma unroll
        for (int t = 0; t < S; ++t) {
            const int4 va = sm[buf][(wr * S + t) * 64 + lane];
            const int4 vb = sm[buf][(4 * S + wc * S + t) * 64 + lane];
            a[t] = i8v4{va.x, va.y, va.z, va.w};
            b[t] = i8v4{vb.x, vb.y, vb.z, vb.w};
        }
#pragma unroll
        for (int ta = 0; ta < S; ++ta)
#pragma unroll
            for (int tb = 0; tb + ta < S; ++tb) acc[ta + tb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ta], b[tb], acc[ta + tb], 0, 0, 0);
        // chunk kc + 1 must be in LDS before anybody reads it: this wave's copies of it are the older ones of what it has in flight; and
        // every read of this chunk has returned before its buffer is restaged two iterations on
        if (kc + 2 < nKc) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPerWave) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const int j = (ctB0 + wc) * 32 + (lane & 31);
    const int ibase = (ctA0 + wr) * 32;
    const int ej = j < N ? eB[j] - 2048 : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = ibase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (i < M && j < N) {
            double v = 0.0;
#pragma unroll
            for (int d = S - 1; d >= 0; --d) v += ldexp((double)acc[d][r], -(12 + kI8Bits * d));  // smallest terms first
            const int ei = eA[i];
            if (ei > 0 && eB[j] > 0) {  // (a column that is all zero has no exponent and contributes nothing)
                double* dst = C + (long long)i * ldc + j;
                *dst += alpha * ldexp(v, ei - 2048 + ej);
            }
        }
    }
}

}  // namespace eqf
