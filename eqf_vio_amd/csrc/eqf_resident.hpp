// k_chol_resident: the WHOLE factorisation part of one vision update -- both Cholesky chains with their right-hand
// sides, the reductions, the covariance downdate and the innovation lift -- as ONE launch (gfx950, fp64).
//
// Same mathematics and the same building blocks as k_chol_step64 (eqf_chol64.hpp: 64-wide block columns, register-chained
// MFMA panel solves, the staged 16-column diagonal factorisation); what changes is WHERE the data lives and how the steps
// are ordered.  k_chol_step64 is one launch per block column: every launch re-reads and re-writes all trailing tiles
// through global memory (8.5 MB per launch, 85 MB per update at N = 200) and every block column pays a kernel boundary.
// Here every 64x64 tile of the two chain matrices has ONE workgroup that keeps it in registers from the first to the last
// update that touches it; only SOLVED blocks (L_RK, Y_Kt) and the diagonal-factor records D[K] travel, once each, through
// global memory with the in-launch hand-off of eqf_handoff.hpp (write-through stores + epoch flags).
//
// Roles (per chain; S-chain: nb block rows, wt right-hand-side column tiles; E-chain: wt = 1):
//   H(R), R = 1..nb-1   "row head": tiles (R, R-1) and (R, R).  Applies the panels K < R-1 to both, then -- the only serial
//                       part -- waits for D[R-1], solves L_{R,R-1}, publishes it, updates and factors the diagonal tile,
//                       publishes D[R].  One hand-off per block column on the critical path.
//   T(R,C), C <= R-2    interior tile: applies the panels K < C, solves with D[C], publishes L_RC.
//   W(t,C)              right-hand-side tile (block row C, column tile t): applies K < C, solves with D[C], publishes
//                       Y_Ct; S-chain tiles also carry the innovation column z along and leave their share of
//                       gamma = Y^T z; the E-chain's leave their share of G11 = [Zt|Et]^T [Zt|Et].
//   The last right-hand-side workgroup of the E-chain waits for every share, sums them IN BLOCK-ROW ORDER (deterministic)
//   and runs the innovation lift.  The downdate tiles Sigma - Y^T Y are workgroups of their own BEHIND the roles in the grid: they
//   wait for the S-chain's last Y tile and overlap the tail of the (longer) E-chain.
// Deadlock freedom: the roles are numbered in dependency order -- a workgroup only ever waits for LOWER numbers (roles: the groups before
// theirs; downdate tiles: the S-chain's roles).  On a co-resident grid everything is resident.  On a grid larger than the chip the number
// of a workgroup is a TICKET drawn when it starts (ResArgs::ticket, round 6): whoever holds number i runs, and all lower numbers run or are
// done -- no assumption about the order in which the hardware starts workgroups (rounds 3-5: the block index, i.e. "in order").  (Until late in round 3 a co-resident grid ran the downdate in the finished role
// workgroups, which waited for higher block indices while holding their CUs: removed.)  Every wait is bounded (eqf_handoff.hpp,
// 0.5 s): a timeout raises the sticky device error flag (bit 128 -> EQF_ERR_NUMERIC from eqf_device_error) and the workgroup that saw it
// publishes nothing more, so the downdate tiles never see the S-chain complete: Sigma_out is not overwritten from stale operands.
// The host uses this kernel at every size whose chains are of unequal length (eqf_capi.hip: since the build for two workgroups per CU,
// OCC2 below, it beats the per-column launches from one filter of N = 200 to 96 of them and to one filter of N = 4000).
#pragma once
#include "eqf_chol64.hpp"

namespace eqf {

struct ResRole {
    int kind;  // 0 S-chain, 1 E-chain
    int role;  // 0 H, 1 T, 2 W, 6 F0 (the chain's first diagonal block, from Sigma)
    int R, C;  // H: R ; T: (R, C) ; W: R = column tile t, C = block row
};

struct ResArgs {
    ChainArgs c0, c1;      // flags = D-record flags (epoch valued)
    UpdArgs a;
    const ResRole* roles;  // [gridDim.x]
    int* readyA;           // [B][2][nbCap * nbCap]  panel block (R, K) of chain c solved     (epoch valued)
    int* readyY;           // [B][2][nbCap * wtCap]  right-hand-side tile (block row C, tile t) solved
    int* counters;         // [B][4]: 0 = S-chain Y tiles out, 1 = next downdate tile (zeroed by the prep launch)
    double* gammaPart;     // [B][nbCap][ldY]   share of block row C in gamma / hV
    double* g11Part;       // [B][nbCap][128]   share of block row C in G11
    int nbCap, wtCap;
    int ddNt;              // 64 x 64 downdate tiles per edge (the tile workgroups behind the roles)
    int* errflag;
    int nRoles, nDdTiles;  // role workgroups, downdate-tile workgroups behind them (per filter)
    int rolesPerRow;       // grid.x = batch * rolesPerRow
    int waitD0;            // bit `kind`: D[0] of that chain is produced inside this launch (role F0): its consumers wait for the flag like for any D[K]
    // ---- the prep work INSIDE this launch (co-resident grids): nPrep = lmBlocks + eBlocks workgroups per filter in FRONT of the roles do
    // what k_update_prep64's landmark waves and Z-row workgroups do (updatePrepBody, write-through), each raising prepFlags[b][i] = epoch;
    // the roles that read S, the right-hand sides or Z_P wait for all of them and read through agent-scope loads.  0: a prep launch came first.
    int nPrep, lmBlocks, prepWpb, prepNvPad;
    int nFront;            // roles[0 .. nFront) sit IN FRONT of the prep roles in the grid (they need nothing of the prep work: the first diagonal blocks and the
                           // E-chain's first dependency group), the prep roles follow, then roles[nFront ..)
    int* prepFlags;        // [B][nPrepCap]
    int nPrepCap;
    int eFromSigma;        // the E-chain's tiles are first read straight from Sigma (EA = Sigma[6:, 6:] with the pad row / column 5 and
                           // the rows past n_e as the identity): the prep launch does not copy them
    int* stageFlags;       // [B][2][nbCap][4]  stage j of D[K] of chain c is in the record (epoch valued; factor64's stageFlag)
    // Grids larger than the chip (the PIPEH builds): WHICH role of its filter a workgroup plays is DRAWN when it starts --
    // atomicAdd(ticket[32 b], 1) - ticketBase, one counter per filter on a line of its own (the counters run on from launch to launch, the
    // host keeps the base; the filter itself stays the block index's: with a batch that is a multiple of 8 a filter's workgroups share an
    // XCD, i.e. an L2) -- instead of read off its block index.  Whoever holds ticket i of a filter is running, and every role it can wait
    // for is a lower ticket of the same filter, i.e. running or done: the roles are free of deadlock in whatever order the hardware starts
    // the workgroups (rounds 3-5 relied on "in the order of their linear index", which HIP does not promise).  One exception keeps the
    // time-out as its only guard: batches that are multiples of 8 above 8 hand a downdate tile to a workgroup of ANOTHER filter of the same
    // XCD (the L2-friendly order below), which then waits for roles whose tickets it cannot vouch for.  nullptr: the block index
    // (co-resident grids: everything is resident, nothing to order).  (One counter for the whole grid was measured first: 8 filters 164.6 ->
    // 181.8 us per update, 64 filters 982 -> 1054 -- 512 workgroups arrive at once and a device-scope atomic on one address takes ~35 ns,
    // and the arrival order scatters a filter's roles over the XCDs.)
    unsigned* ticket;
    unsigned ticketBase;
};

// ---- 64x64 block <-> LDS through write-through / L1-bypassing 16-byte accesses.  Thread t handles row t / 4, 16 doubles
// starting at column 16 (t % 4): one address register per block and immediate offsets (the asm operand limit).
EQF_DEV void hoLoadBlocks2(const double* srcA, int ldA, double (*dstA)[kSP], const double* srcB, int ldB, double (*dstB)[kSP], int tid) {
    const int r = tid >> 2, c = (tid & 3) * 16;
    const char* pa = reinterpret_cast<const char*>(srcA + (long long)r * ldA + c);
    const char* pb = reinterpret_cast<const char*>(srcB + (long long)r * ldB + c);
    v4i32 a[8], b[8];
    asm volatile(
        "global_load_dwordx4 %0, %16, off sc0 sc1\n\tglobal_load_dwordx4 %1, %16, off offset:16 sc0 sc1\n\t"
        "global_load_dwordx4 %2, %16, off offset:32 sc0 sc1\n\tglobal_load_dwordx4 %3, %16, off offset:48 sc0 sc1\n\t"
        "global_load_dwordx4 %4, %16, off offset:64 sc0 sc1\n\tglobal_load_dwordx4 %5, %16, off offset:80 sc0 sc1\n\t"
        "global_load_dwordx4 %6, %16, off offset:96 sc0 sc1\n\tglobal_load_dwordx4 %7, %16, off offset:112 sc0 sc1\n\t"
        "global_load_dwordx4 %8, %17, off sc0 sc1\n\tglobal_load_dwordx4 %9, %17, off offset:16 sc0 sc1\n\t"
        "global_load_dwordx4 %10, %17, off offset:32 sc0 sc1\n\tglobal_load_dwordx4 %11, %17, off offset:48 sc0 sc1\n\t"
        "global_load_dwordx4 %12, %17, off offset:64 sc0 sc1\n\tglobal_load_dwordx4 %13, %17, off offset:80 sc0 sc1\n\t"
        "global_load_dwordx4 %14, %17, off offset:96 sc0 sc1\n\tglobal_load_dwordx4 %15, %17, off offset:112 sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]), "=&v"(b[0]), "=&v"(b[1]),
          "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]), "=&v"(b[7])
        : "v"(pa), "v"(pb)
        : "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        dstA[r][c + 2 * u] = hoLo(a[u]);
        dstA[r][c + 2 * u + 1] = hoHi(a[u]);
        dstB[r][c + 2 * u] = hoLo(b[u]);
        dstB[r][c + 2 * u + 1] = hoHi(b[u]);
    }
}
EQF_DEV void hoLoadBlock(const double* src, int ld, double (*dst)[kSP], int tid) {
    const int r = tid >> 2, c = (tid & 3) * 16;
    const char* pa = reinterpret_cast<const char*>(src + (long long)r * ld + c);
    v4i32 a[8];
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc0 sc1\n\t"
        "global_load_dwordx4 %2, %8, off offset:32 sc0 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:48 sc0 sc1\n\t"
        "global_load_dwordx4 %4, %8, off offset:64 sc0 sc1\n\tglobal_load_dwordx4 %5, %8, off offset:80 sc0 sc1\n\t"
        "global_load_dwordx4 %6, %8, off offset:96 sc0 sc1\n\tglobal_load_dwordx4 %7, %8, off offset:112 sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7])
        : "v"(pa)
        : "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        dst[r][c + 2 * u] = hoLo(a[u]);
        dst[r][c + 2 * u + 1] = hoHi(a[u]);
    }
}
// LDS block -> global, write-through; the caller drains, synchronises and publishes the flag
// (coalesced: 32 consecutive lanes cover one 512-byte row)
EQF_DEV void hoStoreBlock(double* dst, int ld, const double (*src)[kSP], int tid) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = tid + 256 * u, r = p >> 5, c = 2 * (p & 31);
        hoStore16(dst + (long long)r * ld + c, src[r][c], src[r][c + 1]);
    }
}
// diagonal-factor record (L_KK 64x64 + four inverse 16x16 blocks) global -> s.L / s.Wd
EQF_DEV void hoLoadRecord(const double* Dk, const Lds64& s, int tid) {
    v4i32 v[10];
    hoLoad16x10(reinterpret_cast<const char*>(Dk) + 16 * tid, v);
#pragma unroll
    for (int u = 0; u < 10; ++u) {
        const int e = 2 * (tid + 256 * u);
        if (e < kSB * kSB) {
            s.L[e >> 6][e & 63] = hoLo(v[u]);
            s.L[e >> 6][(e & 63) + 1] = hoHi(v[u]);
        } else {
            const int q = e - kSB * kSB;
            s.Wd[q >> 8][(q >> 4) & 15][q & 15] = hoLo(v[u]);
            s.Wd[q >> 8][(q >> 4) & 15][(q & 15) + 1] = hoHi(v[u]);
        }
    }
}
// ONE 16-column stage of a diagonal-factor record: columns [16 j, 16 j + 16) of L_KK (64 rows, 8 KB) -> s.L, W_jj (2 KB) -> s.Wd[j].
// Three 16-byte loads per thread, in flight together.
EQF_DEV void hoLoadStage(const double* Dk, int j, const Lds64& s, int tid) {
    const int w0 = tid, w1 = tid + 256;  // word (16 bytes) w: row w / 8, doubles 2 (w % 8) .. +1 of the column block
    const char* p0 = reinterpret_cast<const char*>(Dk + (w0 >> 3) * kSB + kQB * j + 2 * (w0 & 7));
    const char* p1 = reinterpret_cast<const char*>(Dk + (w1 >> 3) * kSB + kQB * j + 2 * (w1 & 7));
    const char* p2 = reinterpret_cast<const char*>(Dk + kSB * kSB + kQB * kQB * j + 2 * (tid & 127));
    v4i32 a, b, c;
    asm volatile(
        "global_load_dwordx4 %0, %3, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off sc0 sc1\n\tglobal_load_dwordx4 %2, %5, off sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a), "=&v"(b), "=&v"(c)
        : "v"(p0), "v"(p1), "v"(p2)
        : "memory");
    s.L[w0 >> 3][kQB * j + 2 * (w0 & 7)] = hoLo(a);
    s.L[w0 >> 3][kQB * j + 2 * (w0 & 7) + 1] = hoHi(a);
    s.L[w1 >> 3][kQB * j + 2 * (w1 & 7)] = hoLo(b);
    s.L[w1 >> 3][kQB * j + 2 * (w1 & 7) + 1] = hoHi(b);
    if (tid < 128) {
        s.Wd[j][tid >> 3][2 * (tid & 7)] = hoLo(c);
        s.Wd[j][tid >> 3][2 * (tid & 7) + 1] = hoHi(c);
    }
}
// Stages 0, 1, 2 together (nine loads in flight: one round trip) and the inverse block of stage 3 on its own -- what the row head of the
// next block column needs of D[K]: L_ji for i < j and the four W_jj.
EQF_DEV void hoLoadStages012(const double* Dk, const Lds64& s, int tid) {
    const int w0 = tid, w1 = tid + 256;
    const char* pl0 = reinterpret_cast<const char*>(Dk + (w0 >> 3) * kSB + 2 * (w0 & 7));
    const char* pl1 = reinterpret_cast<const char*>(Dk + (w1 >> 3) * kSB + 2 * (w1 & 7));
    const char* pw = reinterpret_cast<const char*>(Dk + kSB * kSB + 2 * (tid & 127));
    const char* pw2 = pw + 2 * kQB * kQB * 8;  // (W_22: beyond the 13-bit immediate offset)
    v4i32 a[3], b[3], c[3];
    asm volatile(
        "global_load_dwordx4 %0, %9, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off offset:128 sc0 sc1\n\t"
        "global_load_dwordx4 %2, %9, off offset:256 sc0 sc1\n\tglobal_load_dwordx4 %3, %10, off sc0 sc1\n\t"
        "global_load_dwordx4 %4, %10, off offset:128 sc0 sc1\n\tglobal_load_dwordx4 %5, %10, off offset:256 sc0 sc1\n\t"
        "global_load_dwordx4 %6, %11, off sc0 sc1\n\tglobal_load_dwordx4 %7, %11, off offset:2048 sc0 sc1\n\t"
        "global_load_dwordx4 %8, %12, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2])
        : "v"(pl0), "v"(pl1), "v"(pw), "v"(pw2)
        : "memory");
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        s.L[w0 >> 3][kQB * j + 2 * (w0 & 7)] = hoLo(a[j]);
        s.L[w0 >> 3][kQB * j + 2 * (w0 & 7) + 1] = hoHi(a[j]);
        s.L[w1 >> 3][kQB * j + 2 * (w1 & 7)] = hoLo(b[j]);
        s.L[w1 >> 3][kQB * j + 2 * (w1 & 7) + 1] = hoHi(b[j]);
        if (tid < 128) {
            s.Wd[j][tid >> 3][2 * (tid & 7)] = hoLo(c[j]);
            s.Wd[j][tid >> 3][2 * (tid & 7) + 1] = hoHi(c[j]);
        }
    }
}
// All four stages at once (ten loads in flight: ONE round trip): what a row head asks for when D[K] is already complete on its arrival.
EQF_DEV void hoLoadStages0123(const double* Dk, const Lds64& s, int tid) {
    const int w0 = tid, w1 = tid + 256;
    const char* pl0 = reinterpret_cast<const char*>(Dk + (w0 >> 3) * kSB + 2 * (w0 & 7));
    const char* pl1 = reinterpret_cast<const char*>(Dk + (w1 >> 3) * kSB + 2 * (w1 & 7));
    const char* pw = reinterpret_cast<const char*>(Dk + kSB * kSB + 2 * (tid & 127));
    const char* pw2 = pw + 2 * kQB * kQB * 8;  // (W_22, W_33: beyond the 13-bit immediate offset)
    v4i32 a[3], b[3], c[4];
    asm volatile(
        "global_load_dwordx4 %0, %10, off sc0 sc1\n\tglobal_load_dwordx4 %1, %10, off offset:128 sc0 sc1\n\t"
        "global_load_dwordx4 %2, %10, off offset:256 sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\t"
        "global_load_dwordx4 %4, %11, off offset:128 sc0 sc1\n\tglobal_load_dwordx4 %5, %11, off offset:256 sc0 sc1\n\t"
        "global_load_dwordx4 %6, %12, off sc0 sc1\n\tglobal_load_dwordx4 %7, %12, off offset:2048 sc0 sc1\n\t"
        "global_load_dwordx4 %8, %13, off sc0 sc1\n\tglobal_load_dwordx4 %9, %13, off offset:2048 sc0 sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
        : "v"(pl0), "v"(pl1), "v"(pw), "v"(pw2)
        : "memory");
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        s.L[w0 >> 3][kQB * j + 2 * (w0 & 7)] = hoLo(a[j]);
        s.L[w0 >> 3][kQB * j + 2 * (w0 & 7) + 1] = hoHi(a[j]);
        s.L[w1 >> 3][kQB * j + 2 * (w1 & 7)] = hoLo(b[j]);
        s.L[w1 >> 3][kQB * j + 2 * (w1 & 7) + 1] = hoHi(b[j]);
    }
    if (tid < 128) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s.Wd[j][tid >> 3][2 * (tid & 7)] = hoLo(c[j]);
            s.Wd[j][tid >> 3][2 * (tid & 7) + 1] = hoHi(c[j]);
        }
    }
}
EQF_DEV void hoLoadW3(const double* Dk, const Lds64& s, int tid) {
    const char* pw = reinterpret_cast<const char*>(Dk + kSB * kSB + 3 * kQB * kQB + 2 * (tid & 127));
    v4i32 c;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(c) : "v"(pw) : "memory");
    if (tid < 128) {
        s.Wd[3][tid >> 3][2 * (tid & 7)] = hoLo(c);
        s.Wd[3][tid >> 3][2 * (tid & 7) + 1] = hoHi(c);
    }
}
EQF_DEV void hoStoreRecord(double* Dk, const Lds64& s, int tid) {
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
        hoStore16(Dk + e, v0, v1);
    }
}

// Thread 0 waits for up to three flags (nullptr = none).  Returns through `bad` (8 = timed out).
// (once a wait has failed -- *bad == 8 -- the later ones of this workgroup are skipped: it publishes nothing more anyway)
EQF_DEV void hoWait3(const int* f0, const int* f1, const int* f2, int epoch, int tid, int* bad, int* err) {
    if (tid == 0 && *bad != 8) {
        if (f0 && !hoWait(f0, epoch, err)) *bad = 8;
        if (*bad != 8 && f1 && !hoWait(f1, epoch, err)) *bad = 8;
        if (*bad != 8 && f2 && !hoWait(f2, epoch, err)) *bad = 8;
    }
    __syncthreads();
}

// Panel solve X = A L_KK^-T of a 64 x 64 tile held in accumulators (wave wv: rows 16 wv .. +15) against a diagonal factor D[K] that ANOTHER
// workgroup is factoring right now: its record arrives in four 16-column stages (factor64's stageFlag), and the substitution
// X_j^T = W_jj (A_j^T - sum_{i<j} L_ji X_i^T) only needs stage j for its step j.  The first three steps run in the shadow of the producer's
// pivot chain; what is left after its last pivot is one flag, 2 KB (W_33) and one 16 x 16 x 16 product -- not the whole 40 KB record and the
// whole solve.  (A consumer usually gets here about a microsecond before the record is complete, so the first three stages are out: one
// wait, one round trip for the three of them, then the last stage on its own.)  Same operations in the same order as solveStrip<true>:
// bitwise the same block.  Result in s.P; all 256 threads; ends with a barrier.
EQF_DEV void stagedPanelSolve(const f64x4 (&acc)[4], const Lds64& s, const double* Dk, const int* stageIn, const int* flagWhole, int epoch, int tid,
    int* bad, int* err) {
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) stTile(acc[i], &s.P[0][0], kSP, kQB * wv, kQB * i, lane);
    // (round 5) A head whose panels came late finds D[K] COMPLETE: then stage by stage buys nothing and costs a second round trip -- in steady
    // state every other head of a chain arrives late (a head that could hide its solve shortens the next one's panel time: stamps in
    // profiles/r05_res_stamps_N200.txt), so one look at the whole-record flag decides between "four stages in one round trip" and the
    // staged path.  The same products in the same order either way.
    __shared__ int sWholeThere;
    if (tid == 0) sWholeThere = (*bad != 8 && __hip_atomic_load(flagWhole, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) ? 1 : 0;
    __syncthreads();
    f64x4 Z[4], X[4];
    const int lc = lane & 15, lg = lane >> 4, x0 = kQB * wv;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) Z[j][q] = s.P[x0 + lc][kQB * j + lg + 4 * q];
    const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
    if (sWholeThere) {
        hoLoadStages0123(Dk, s, tid);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            X[j] = mmRegB(zero, &s.Wd[j][0][0], kWP, 0, 0, Z[j], lane, 1.0);
#pragma unroll
            for (int j2 = j + 1; j2 < 4; ++j2) Z[j2] = mmRegB(Z[j2], &s.L[0][0], kSP, kQB * j2, kQB * j, X[j], lane, -1.0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) s.P[x0 + lc][kQB * j + lg + 4 * q] = X[j][q];
        __syncthreads();
        return;
    }
    hoWait3(stageIn + 2, nullptr, nullptr, epoch, tid, bad, err);
    hoLoadStages012(Dk, s, tid);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        X[j] = mmRegB(zero, &s.Wd[j][0][0], kWP, 0, 0, Z[j], lane, 1.0);
#pragma unroll
        for (int j2 = j + 1; j2 < 4; ++j2) Z[j2] = mmRegB(Z[j2], &s.L[0][0], kSP, kQB * j2, kQB * j, X[j], lane, -1.0);
    }
    hoWait3(flagWhole, nullptr, nullptr, epoch, tid, bad, err);
    hoLoadW3(Dk, s, tid);
    __syncthreads();
    X[3] = mmRegB(zero, &s.Wd[3][0][0], kWP, 0, 0, Z[3], lane, 1.0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) s.P[x0 + lc][kQB * j + lg + 4 * q] = X[j][q];
    __syncthreads();
}

// ---- software-pipelined panel loop (round 3): the two 64 x 64 operand blocks of panel K+1 travel global -> REGISTERS while the matrix
// cores work on panel K from LDS.  The loads are 8-byte agent-scope atomic loads (hoLoad8), not the 16-byte asm loads of hoLoadBlocks2:
// the compiler tracks them, so they may stay in flight across the products (an asm load must carry its s_waitcnt inside the statement).
// Thread t holds elements t + 256 u, u = 0..15, of each block (a wavefront reads 512 contiguous bytes per instruction).
// Why: a workgroup that is dispatched late (a batch on a grid larger than the chip) finds ALL its panels waiting and used to apply them
// one after the other at 5 us each (flag poll + 64 KB round trip + 1.7 us of MFMAs + two barriers); pipelined a panel costs what the
// longer of the two takes.
struct PanelRegs {
    double p[16], q[16];
};
EQF_DEV void panelIssue(PanelRegs& r, const double* srcP, int ldP, const double* srcQ, int ldQ, int tid) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = tid + 256 * u, row = e >> 6, col = e & 63;
        r.p[u] = hoLoad8(srcP + (long long)row * ldP + col);
        r.q[u] = hoLoad8(srcQ + (long long)row * ldQ + col);
    }
}
// (one block of the pair at a time: at the frontier the two flags of a panel come microseconds apart -- see the right-hand-side roles)
EQF_DEV void panelIssueP(PanelRegs& r, const double* srcP, int ldP, int tid) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = tid + 256 * u, row = e >> 6, col = e & 63;
        r.p[u] = hoLoad8(srcP + (long long)row * ldP + col);
    }
}
EQF_DEV void panelIssueQ(PanelRegs& r, const double* srcQ, int ldQ, int tid) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = tid + 256 * u, row = e >> 6, col = e & 63;
        r.q[u] = hoLoad8(srcQ + (long long)row * ldQ + col);
    }
}
EQF_DEV void panelToLds(const PanelRegs& r, double (*dstP)[kSP], double (*dstQ)[kSP], int tid) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = tid + 256 * u, row = e >> 6, col = e & 63;
        dstP[row][col] = r.p[u];
        dstQ[row][col] = r.q[u];
    }
}
EQF_DEV bool hoProbe3(const int* f0, const int* f1, const int* f2, int epoch) {
    bool ok = __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch &&
              __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
    if (f2) ok = ok && __hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
    return ok;
}
// non-blocking look at two flags (one lane): both published?
EQF_DEV bool hoProbe2(const int* f0, const int* f1, int epoch) {
    return __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch &&
           __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
}

#ifdef EQF_RES_STAMPS
#ifndef EQF_STAMP_B
#define EQF_STAMP_B 0  // the filter whose workgroups leave stamps (-DEQF_STAMP_B=(gridDim.x-1): the last one of a batch)
#endif
__device__ long long g_resStamps[2][16][16];  // [chain][R][phase] wall-clock (100 MHz) stamps of the row heads, filter 0
#define EQF_HSTAMP(i) do { if (tid == 0 && b == EQF_STAMP_B && R < 16) g_resStamps[role.kind][R][i] = wall_clock64(); } while (0)
#define EQF_WSTAMP(i) do { if (tid == 0 && b == EQF_STAMP_B && !isS && C == nb - 1) g_resStamps[1][15][i] = wall_clock64(); } while (0)
#else
#define EQF_HSTAMP(i) do { } while (0)
#define EQF_WSTAMP(i) do { } while (0)
#endif
#if defined(EQF_RES_STAMPS) && defined(EQF_F64_STAMPS)
__device__ long long g_resF64[2][16][128];  // [chain][R]: factor64's per-wave stamps (sF64Stamps) of the row heads, filter EQF_STAMP_B
#define EQF_F64_ST() ((b == EQF_STAMP_B && R < 16) ? &g_resF64[role.kind][R][0] : nullptr)
#else
#define EQF_F64_ST() nullptr
#endif
// PIPEH: the row heads apply their old panels with the pipelined loop of the interior tiles as well.  Off for a grid that is co-resident
// (one or two small filters: every head is there from the start and applies each panel the moment it appears; the 64 prefetch registers
// cost the pivot chain 4 us per update through the register allocation), on for a batch on a grid larger than the chip: there a head is
// dispatched when R - 1 panels are already waiting, and at 6.5 us per panel (flag, 64 KB round trip, 3.4 us of MFMAs, two barriers) the heads of
// the later block columns were still catching up when their diagonal factor was due (wall-clock stamps, 8 filters: H(8) 9 us late).
// OCC2: the build for grids MANY times the chip (from ~4.5 roles per CU with a batch of 8 or more, ~6 otherwise): two workgroups per CU --
// 237 registers, and the 78 KB LDS view in which L aliases Q (ldsRes2).  Twice the workgroups in flight reach twice as many block columns
// ahead of the pivot chains: 8 filters 181 -> 170 us per update, 12: 254 -> 220, 16: 312 -> 255, 20: 401 -> 315, N = 1000 1.54 -> 1.28 ms.
// On lightly oversubscribed grids two workgroups on a CU only get in each other's way (4 filters 134 -> 153 us, N = 600 455 -> 475).
// FOLD: the build with the prep roles inside (ResArgs::nPrep).  Round 4: co-resident grids (one filter).  Round 5: also with PIPEH / OCC2 for
// batches on grids larger than the chip -- filter index fastest, so the prep workgroups of ALL filters are dispatched first, and they wait for
// nobody: the dependency order of the block indices holds as before.  A build of its own because the mere presence of the prep
// role and of the agent-scope loads of S / the right-hand sides cost the other builds 3 + 7 us per update (124 -> 134: this kernel is
// bound by a chain of latencies and feels every change of its code layout).
#ifdef EQF_WAIT_STATS
template <typename T, bool PIPEH = false, bool OCC2 = false, bool FOLD = false, bool TICKET = false>
__device__ __forceinline__ void residentBody(const ResArgs& ra) {
#else
// TICKET (round 6): the build that draws its role from ResArgs::ticket.  A build of its own for the same reason as FOLD: with the draw compiled
// into the default kernels, the SAME box ran them 0.9 % (2, 4 filters) to 2.3 % (16, 64 filters, N = 1000) slower than round 5's library
// (profiles/r06_ab_against_r05_tickets_auto.txt) -- half of it with the draw switched off.
template <typename T, bool PIPEH = false, bool OCC2 = false, bool FOLD = false, bool TICKET = false>
__global__ __launch_bounds__(256, OCC2 ? 2 : 1) void k_chol_resident(ResArgs ra) {
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smemR[];
    // grid = (batch, roles): the filter index runs FASTEST in dispatch order, so that a grid larger than the chip advances all filters
    // together, dependency group by dependency group (role-major order would run the filters one after the other), and with a batch
    // that is a multiple of 8 every workgroup of filter b runs on XCD b mod 8
    // (role index and filter from the linear workgroup index: grid = (B * rolesPerRow, rows), filter fastest; one row unless the roles and
    // downdate tiles of a filter are more than 32768 -- N > ~2700)
    const int nB = (int)gridDim.x / ra.rolesPerRow;
    int roleIdxAll = (int)blockIdx.x / nB + ra.rolesPerRow * (int)blockIdx.y;
    const int bIdx = (int)blockIdx.x % nB;
    static_assert(!TICKET || (PIPEH && !FOLD), "the ticket build exists for grids larger than the chip with a prep launch in front");
    if constexpr (TICKET) {
        // (see ResArgs::ticket) the role index inside the filter is the order of ARRIVAL among the filter's workgroups
        int* const sT = reinterpret_cast<int*>(smemR);
        if (threadIdx.x == 0) *sT = (int)(atomicAdd(ra.ticket + 32 * bIdx, 1u) - ra.ticketBase);
        __syncthreads();
        roleIdxAll = *sT;
        __syncthreads();  // (the first bytes of the dynamic LDS belong to the role from here on)
    }
    if (FOLD && roleIdxAll >= ra.nFront && roleIdxAll < ra.nFront + ra.nPrep) {
        // ---- a prep role (see ResArgs::nPrep): among the lowest block indices of the grid -- nothing they need is produced in this launch
        EQF_STAT_CLASS(4);
        const int prepIdx = roleIdxAll - ra.nFront;
        const Glob& gp = ra.a.g[bIdx];
        if (gp.updateOk && gp.N != 0) {
            updatePrepBody<T, true>(ra.a, prepIdx, bIdx, ra.lmBlocks, ra.prepWpb, ra.prepNvPad, reinterpret_cast<double*>(smemR));
            hoDrain();
        }
        __syncthreads();
        if (threadIdx.x == 0) hoPublish(ra.prepFlags + (long long)bIdx * ra.nPrepCap + prepIdx, ra.c0.epoch);
        return;
    }
    const int roleIdx = roleIdxAll - ((FOLD && roleIdxAll >= ra.nFront) ? ra.nPrep : 0);
    if (roleIdx >= ra.nRoles + ra.nDdTiles) return;
    if (roleIdx >= ra.nRoles) {
        // ---- a downdate tile of filter blockIdx.x.  These workgroups have the HIGHEST block indices: they are dispatched when
        // the role workgroups in front of them have been, i.e. during the last block columns of the E-chain, and they wait -- for LOWER
        // block indices only -- until the S-chain's last Y tile is out (it usually is).  The downdate overlaps the tail of the longer chain.
        // Which tile of which filter: with a batch that is a multiple of 8 the workgroups of filter b all run on XCD b mod 8, and in dispatch
        // order (filter fastest) an XCD would hold the same few tiles of ALL its filters at a time -- at 64 filters 8 x 2.3 MB of Y against 4 MB
        // of L2: every panel came from the memory side again (3520 tiles x 458 KB = 1.6 GB per update, the 330 us tail of the launch at the
        // MALL's rate; profiles/r05_res_stamps_B64.txt).  So within an XCD the tiles run filter by filter: one filter's Y at a time.
        EQF_STAT_CLASS(5);
        int bb = bIdx, tile = roleIdx - ra.nRoles;
        if ((nB & 7) == 0 && nB > 8) {
            const int seq = tile * (nB >> 3) + (bIdx >> 3);
            bb = (bIdx & 7) + 8 * (seq / ra.nDdTiles);
            tile = seq % ra.nDdTiles;
        }
        const int t = threadIdx.x;
        const Glob& gg = ra.a.g[bb];
        int late = 0;
#ifdef EQF_RES_STAMPS
        const int slot = tile == 0 ? 0 : (tile == 27 ? 1 : (tile == 54 ? 2 : -1));
        if (t == 0 && bb == EQF_STAMP_B && slot >= 0) g_resStamps[0][15][3 * slot] = wall_clock64();
#endif
        constexpr bool kGated = !(FOLD && !PIPEH);  // the 64 x 64 tiles follow the S-chain block row by block row (downdateTile's Gate)
        const int* const ryAll = ra.readyY + ((long long)bb * 2 + 0) * ra.nbCap * ra.wtCap;
        if (!kGated && gg.updateOk && gg.N != 0) {
            // (the Y tiles of the S-chain's LAST block row carry the epoch when they are out, and a tile of block row C is only solved after
            // the tiles above it: one flag per column tile says "all of Y".  A counter said the same until round 4; it had to be zeroed by
            // an earlier launch, which the prep roles of this launch are not.)
            int nS, wS;
            chainDims64(ra.c0, gg.N, &nS, &wS);
            // (ONE lane, long sleeps: on a co-resident grid these 55 workgroups poll from the first microsecond on -- with every lane of
            // ten polling at hoWait's rate the flags of the chains' own hand-offs got slower: 124 -> 133 us per update)
            const int* ry = ryAll + (nS - 1) * ra.wtCap;
            if (t == 0) {
                const long long t0 = wall_clock64();
                for (int i = 0; i < wS && !late; ++i)
                    while (__hip_atomic_load(ry + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ra.c0.epoch) {
                        __builtin_amdgcn_s_sleep(32);
                        if (hoAborted(ra.errflag)) {  // (some hand-off of this launch timed out: nobody will finish the S-chain)
                            late = 1;
                            break;
                        }
                        if (wall_clock64() - t0 > 50000000LL) {  // 0.5 s
                            if (ra.errflag) atomicOr(ra.errflag, kHoErrTimeout);
                            late = 1;
                            break;
                        }
                    }
            }
        }
        if (__syncthreads_or(late)) return;  // (Sigma_out is left untouched: the sticky flag makes the host fail the update)
#ifdef EQF_RES_STAMPS
        if (t == 0 && bb == EQF_STAMP_B && slot >= 0) g_resStamps[0][15][3 * slot + 1] = wall_clock64();
#endif
        // Round 5, grids larger than the chip: the tile workgroups are dispatched while the S-chain is still running (8 filters of N = 200:
        // 65 .. 120 us into a launch whose last Y tile comes at 128), so each wave waits for the two Y tiles of a block row right before it
        // asks for that block row's first chunk -- one lane per wave, long sleeps (see above) -- and the downdate ends a few microseconds after
        // the S-chain's last right-hand-side tile instead of 45 us after it.
        // (A tile that is dispatched when the S-chain is through -- all of them on a grid many times the chip: 64 filters -- would still pay
        // two dependent flag reads per block row, ~2.5 us each time with every wave of the workgroup behind the next barrier: 7 block rows, a
        // third of the tile's 50 us.  So every look also reads the column tiles' flags of the LAST block row, in the same round trip: a tile
        // of block row C is only solved after the tiles above it, so once those two are out nothing is looked at again.)
        int nSrows = 1, wSdummy = 0;
        if (kGated && gg.updateOk && gg.N != 0) chainDims64(ra.c0, gg.N, &nSrows, &wSdummy);
        int allOut = 0;
        auto gate = [&](int C, int ti, int tj) -> bool {
            if (allOut) return true;
#ifdef EQF_WAIT_STATS
            const long long tGate0 = wall_clock64();
#endif
            int ok = 1, all = 0;
            if ((t & 63) == 0) {
                const long long t0 = wall_clock64();
                const int* f0 = ryAll + C * ra.wtCap + ti;
                const int* f1 = ryAll + C * ra.wtCap + tj;
                const int* l0 = ryAll + (nSrows - 1) * ra.wtCap + ti;
                const int* l1 = ryAll + (nSrows - 1) * ra.wtCap + tj;
                const int v0 = __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), v1 = __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int w0 = __hip_atomic_load(l0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w1 = __hip_atomic_load(l1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                all = (w0 == ra.c0.epoch && w1 == ra.c0.epoch) ? 1 : 0;
                for (int i = 0; i < 2 && ok && !all; ++i) {
                    if ((i ? v1 : v0) == ra.c0.epoch) continue;
                    const int* fl = i ? f1 : f0;
#ifndef EQF_GATE_SLEEP
#define EQF_GATE_SLEEP 32
#endif
                    while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ra.c0.epoch) {
                        __builtin_amdgcn_s_sleep(EQF_GATE_SLEEP);
                        if (hoAborted(ra.errflag)) {
                            ok = 0;
                            break;
                        }
                        if (wall_clock64() - t0 > 50000000LL) {  // 0.5 s
                            if (ra.errflag) atomicOr(ra.errflag, kHoErrTimeout);
                            ok = 0;
                            break;
                        }
                    }
                }
            }
            asm volatile("" ::: "memory");
#ifdef EQF_WAIT_STATS
            if ((t & 63) == 0) EQF_STAT_WAIT(wall_clock64() - tGate0);
#endif
            allOut = __builtin_amdgcn_readfirstlane(all);
            return __builtin_amdgcn_readfirstlane(ok) != 0;
        };
        // (FOLD: 32 x 32 tiles -- with the S-chain starting behind the prep roles the downdate ends the launch, and a 64 x 64 tile is 28 us
        // of dependent fetches for one workgroup; four times the workgroups, each a third of that)
        // (a grid larger than the chip -- FOLD with PIPEH, round 5 -- keeps the 64 x 64 tiles: there the tile workgroups are many and throughput counts)
        if constexpr (!kGated) downdateTile<T, 32>(ra.a, ra.ddNt, bb, tile, reinterpret_cast<T*>(smemR));
        else if (gg.updateOk && gg.N != 0) downdateTile<T, 64, 4>(ra.a, ra.ddNt, bb, tile, reinterpret_cast<T*>(smemR), gate);
        else downdateTile<T, 64, 4>(ra.a, ra.ddNt, bb, tile, reinterpret_cast<T*>(smemR));
#ifdef EQF_RES_STAMPS
        __syncthreads();
        if (t == 0 && bb == EQF_STAMP_B && slot >= 0) g_resStamps[0][15][3 * slot + 2] = wall_clock64();
#endif
        return;
    }
    // (bit 128 of the sticky device error word: a hand-off of this launch -- or of an earlier one: the handle must be reset -- timed out.
    // Role workgroups that start after that leave at once; nothing they would publish could be complete.)
    if (hoAborted(ra.errflag)) return;
    const ResRole role = ra.roles[roleIdx];
    EQF_STAT_CLASS((role.role == 6 ? 3 : role.role) + 8 * role.kind);
    const int b = bIdx;
    constexpr bool fold = FOLD;
    const ChainArgs& ch = role.kind ? ra.c1 : ra.c0;
    const UpdArgs& a = ra.a;
    const Glob& g = ch.g[b];
    // a filter whose update is switched off (speculative outlier gate) or that has no landmarks: no factorisation work, but its
    // Sigma still has to reach the other ping-pong buffer -- the downdate loop below copies it
    const bool active = g.updateOk && g.N != 0;
    if (!active) return;  // (its Sigma reaches the other ping-pong buffer through the downdate tile workgroups, which copy it)
    int nb, wt;
    chainDims64(ch, g.N, &nb, &wt);
    int nbS, wtS, nbE, wtE;
    chainDims64(ra.c0, g.N, &nbS, &wtS);
    chainDims64(ra.c1, g.N, &nbE, &wtE);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int epoch = ch.epoch;
    const Lds64 s = OCC2 ? ldsRes2(smemR) : ldsFull(smemR);
    double* A = ch.A + (long long)b * ch.strideA;
    double* D = ch.D + (long long)b * ch.strideD;
    double* W = ch.W + (long long)b * ch.strideW;
    double* WO = ch.WO + (long long)b * ch.strideW;
    const int ldA = ch.ldA, ldW = ch.ldW;
    const int nbCap = ra.nbCap, wtCap = ra.wtCap;
    int* readyA = ra.readyA + ((long long)b * 2 + role.kind) * nbCap * nbCap;
    int* readyY = ra.readyY + ((long long)b * 2 + role.kind) * nbCap * wtCap;
    const int* flagD = ch.flags + (long long)b * ch.strideF;
    int* counters = ra.counters + (long long)b * 4;
    int bad = 0;
    // element (i, j) of tile (R, C) of the chain matrix BEFORE any of this update's operations
    const bool srcSigma = ra.eFromSigma && role.kind == 1;
    const T* Sg = static_cast<const T*>(a.Sin) + (long long)b * a.sigmaStride;
    const int neE = eDim(g.N), ldS = a.ld;
    auto tile0 = [&](int R, int C, int i, int j) __attribute__((always_inline)) {
        if (!srcSigma) return fold ? hoLoad8(A + (long long)(R * kSB + i) * ldA + C * kSB + j) : A[(long long)(R * kSB + i) * ldA + C * kSB + j];
        const int r = R * kSB + i, c = C * kSB + j;
        const bool in = r < neE && c < neE && r != 5 && c != 5;
        const double v = (double)Sg[in ? (long long)(6 + r) * ldS + 6 + c : 0];
        return in ? v : (r == c ? 1.0 : 0.0);
    };

    // (fold) everything the prep roles of this launch write -- S, both chains' right-hand sides -- is there once all of them have published
    auto waitPrep = [&]() {
        if (!fold) return;
        for (int i = tid; i < ra.nPrep; i += 256)
            if (!hoWait(ra.prepFlags + (long long)b * ra.nPrepCap + i, epoch, ra.errflag)) bad = 8;
        bad = __syncthreads_or(bad) ? 8 : 0;
    };
    const bool waitD0 = (ra.waitD0 >> role.kind) & 1;

    if (role.role == 6) {
        // =========================================================================================== F0: the chain's first diagonal block
        // (what two extra workgroups of the prep launch do otherwise; here for the E-chain when its diagonal-factor chain runs as a launch
        // of its own NEXT to the prep launch -- it needs nothing of it: Sigma_e is Sigma[6:, 6:])
        // (Tried: the first row head factors D[0] itself and keeps it in LDS -- no flag, no 40 KB round trip into the chain -- 134.8 -> 140-145 us
        // per update: the inlined factorisation costs EVERY row head more than the first one saves.)
        factorFirstFromSigma<T, true>(a, ch, b, s, &bad);
        hoDrain();
        __syncthreads();
        // (published whatever the pivots said, like every D[R] of the row heads: a non-positive pivot is reported through bit 4 at the end of
        // the kernel and the chain runs on -- withholding the flag turned a numeric error into a 0.5 s stall of every role of the chain.
        // `bad` is the pivot wave's verdict: v_readlane broadcasts make it uniform over wave 0, thread 0 included; F0 waits for nobody.)
        if (tid == 0) hoPublish(ch.flags + (long long)b * ch.strideF, epoch);
    } else if (role.role == 0) {
        // =========================================================================================== H(R)
        const int R = role.R;
        if (R >= nb) return;
        // tile (R, R-1): wave wv owns the 16-row strip (tiles (wv, 0..3)); tile (R, R): the lower triangle in the
        // diagonal-workgroup layout of k_chol_step64 (slot 0 = (wv, 0), the deferred tiles on waves 2, 3)
        int tr[4], tc[4], nt;
        nt = wv >= 2 ? 4 : 1;
        tr[0] = wv; tc[0] = 0;
        tr[1] = wv == 2 ? 1 : 2; tc[1] = 1;
        tr[2] = wv == 2 ? 3 : 2; tc[2] = wv == 2 ? 1 : 2;
        tr[3] = 3;               tc[3] = wv == 2 ? 2 : 3;
        if (!srcSigma) waitPrep();
        f64x4 a1[4], a2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a1[i][q] = tile0(R, R - 1, kQB * wv + (lane >> 4) + 4 * q, kQB * i + (lane & 15));
                a2[i][q] = i < nt ? tile0(R, R, kQB * tr[i] + (lane >> 4) + 4 * q, kQB * tc[i] + (lane & 15)) : 0.0;
            }
        EQF_HSTAMP(0);
        // (A dry run of the serial part during the idle time before the panels arrive -- to take the instruction-cache misses of
        // code a workgroup executes exactly once off the critical path -- was tried and measured: no gain.)
        if constexpr (PIPEH) {
            if (R > 2) {
                __shared__ int sAheadH;
                PanelRegs pr;
                const double* rowP = A + (long long)(R * kSB) * ldA;
                const double* rowQ = A + (long long)((R - 1) * kSB) * ldA;
                const int nK = R - 2;  // panels 0 .. R-3; the last one, K = R-2, below as always
                hoWait3(readyA + R * nbCap, readyA + (R - 1) * nbCap, nullptr, epoch, tid, &bad, ra.errflag);
                panelIssue(pr, rowP, ldA, rowQ, ldA, tid);
                bool probe = (tid == 0 && nK > 1) ? hoProbe2(readyA + R * nbCap + 1, readyA + (R - 1) * nbCap + 1, epoch) : false;
                for (int K = 0; K < nK; ++K) {
                    panelToLds(pr, s.P, s.Q, tid);
                    if (tid == 0) sAheadH = probe ? 1 : 0;
                    __syncthreads();
                    const bool ahead = K + 1 < nK && sAheadH;
                    if (ahead) {
                        panelIssue(pr, rowP + (K + 1) * kSB, ldA, rowQ + (K + 1) * kSB, ldA, tid);
                        probe = (tid == 0 && K + 2 < nK) ? hoProbe2(readyA + R * nbCap + K + 2, readyA + (R - 1) * nbCap + K + 2, epoch) : false;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) a1[i] = mmTile<true, kSB>(a1[i], &s.P[0][0], kSP, kQB * wv, &s.Q[0][0], kSP, kQB * i, lane, -1.0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < nt) a2[i] = mmTile<true, kSB>(a2[i], &s.P[0][0], kSP, kQB * tr[i], &s.P[0][0], kSP, kQB * tc[i], lane, -1.0);
                    __syncthreads();
                    if (K + 1 < nK && !ahead) {
                        hoWait3(readyA + R * nbCap + K + 1, readyA + (R - 1) * nbCap + K + 1, nullptr, epoch, tid, &bad, ra.errflag);
                        panelIssue(pr, rowP + (K + 1) * kSB, ldA, rowQ + (K + 1) * kSB, ldA, tid);
                        probe = (tid == 0 && K + 2 < nK) ? hoProbe2(readyA + R * nbCap + K + 2, readyA + (R - 1) * nbCap + K + 2, epoch) : false;
                    }
                }
            }
        }
        for (int K = PIPEH ? max(R - 2, 0) : 0; K + 1 < R; ++K) {
            if (K + 2 < R) {
                hoWait3(readyA + R * nbCap + K, readyA + (R - 1) * nbCap + K, nullptr, epoch, tid, &bad, ra.errflag);
                hoLoadBlocks2(A + (long long)(R * kSB) * ldA + K * kSB, ldA, s.P, A + (long long)((R - 1) * kSB) * ldA + K * kSB, ldA, s.Q, tid);
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i) a1[i] = mmTile<true, kSB>(a1[i], &s.P[0][0], kSP, kQB * wv, &s.Q[0][0], kSP, kQB * i, lane, -1.0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nt) a2[i] = mmTile<true, kSB>(a2[i], &s.P[0][0], kSP, kQB * tr[i], &s.P[0][0], kSP, kQB * tc[i], lane, -1.0);
                __syncthreads();
            } else {
                // The last panel, K = R-2.  L_{R-1,R-2} is the block the PREVIOUS row head solved a moment ago: it is the last thing to
                // arrive, and only the tile (R, R-1) needs it.  So this row's own block L_{R,R-2} (an interior tile, out long before) is
                // applied to the diagonal tile first, and what is left behind the late block is one 64 x 64 x 64 product instead of two
                // (measured: 7 us from that flag to "panels applied" before -- a second critical path as long as the pivot chain's).
                EQF_HSTAMP(9);
                hoWait3(readyA + R * nbCap + K, nullptr, nullptr, epoch, tid, &bad, ra.errflag);
                hoLoadBlock(A + (long long)(R * kSB) * ldA + K * kSB, ldA, s.P, tid);
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nt) a2[i] = mmTile<true, kSB>(a2[i], &s.P[0][0], kSP, kQB * tr[i], &s.P[0][0], kSP, kQB * tc[i], lane, -1.0);
                hoWait3(readyA + (R - 1) * nbCap + K, nullptr, nullptr, epoch, tid, &bad, ra.errflag);
                EQF_HSTAMP(10);
                hoLoadBlock(A + (long long)((R - 1) * kSB) * ldA + K * kSB, ldA, s.Q, tid);
                __syncthreads();
                EQF_HSTAMP(13);
#pragma unroll
                for (int i = 0; i < 4; ++i) a1[i] = mmTile<true, kSB>(a1[i], &s.P[0][0], kSP, kQB * wv, &s.Q[0][0], kSP, kQB * i, lane, -1.0);
                __syncthreads();
            }
        }
        // ---- the serial part: D[R-1] -> L_{R,R-1} -> diagonal tile -> D[R]
        EQF_HSTAMP(1);
        int* const stageOut = ra.stageFlags ? ra.stageFlags + (((long long)b * 2 + role.kind) * nbCap + R) * 4 : nullptr;
        if (R - 1 == 0 || !ra.stageFlags) {
            // D[0] comes complete from the prep launch
            hoWait3((R - 1 > 0 || waitD0) ? flagD + (R - 1) : nullptr, nullptr, nullptr, epoch, tid, &bad, ra.errflag);
            EQF_HSTAMP(2);
            hoLoadRecord(D + (long long)(R - 1) * kDRec, s, tid);
#pragma unroll
            for (int i = 0; i < 4; ++i) stTile(a1[i], &s.P[0][0], kSP, kQB * wv, kQB * i, lane);
            __syncthreads();
            EQF_HSTAMP(3);
            solveStrip<true>(&s.P[0][0], kSP, s, kQB * wv, lane);
            __syncthreads();
        } else {
            // D[R-1] is being factored by the previous row head RIGHT NOW: stage by stage (stagedPanelSolve, above)
            stagedPanelSolve(a1, s, D + (long long)(R - 1) * kDRec, ra.stageFlags + (((long long)b * 2 + role.kind) * nbCap + (R - 1)) * 4,
                flagD + (R - 1), epoch, tid, &bad, ra.errflag);
        }
        EQF_HSTAMP(4);
        // first column of the diagonal tile, then the factorisation with the other tiles deferred to waves 2, 3; the solved
        // block leaves for the other workgroups meanwhile (stores are asynchronous)
        hoStoreBlock(A + (long long)(R * kSB) * ldA + (R - 1) * kSB, ldA, s.P, tid);
        a2[0] = mmTile<true, kSB>(a2[0], &s.P[0][0], kSP, kQB * tr[0], &s.P[0][0], kSP, kQB * tc[0], lane, -1.0);
        __syncthreads();  // (every wave has finished reading s.L / s.Wd of D[R-1])
        EQF_HSTAMP(5);
        stTile(a2[0], &s.L[0][0], kSP, kQB * tr[0], kQB * tc[0], lane);
        if (wv == 0) stTile(a2[0], &s.D0[0][0], kWP, 0, 0, lane);
        if (wv == 1) factorPrologueW(s, lane);
        __syncthreads();
        // (slot i of this wave is tile (tr[i], tc[i]): the table of factorTiles)
        auto pre = [&](int, int r, int c, f64x4& t) {
#pragma unroll
            for (int i = 1; i < 4; ++i)
                if (tr[i] == r && tc[i] == c) t = mmTile<true, kSB>(a2[i], &s.P[0][0], kSP, kQB * tr[i], &s.P[0][0], kSP, kQB * tc[i], lane, -1.0);
        };
        EQF_HSTAMP(6);
        // the solved block's stores drain in the shadow of the first 16 pivots; it is published right after them.  (Tried in round 3:
        // waves 2, 3 alone store the block and the second of them to have drained publishes it right after its deferred tiles, 1.5 us
        // earlier for the next row head -- 138 -> 142 us per update: sixteen write-through stores per thread and a drain in front of the
        // first stage's barrier cost the pivot chain more than the next head gains.)
        // (factor64 calls this from thread 0 alone, after the first stage's barrier, by which every thread has drained its stores)
        auto mid = [&] {
            if (bad != 8) hoPublish(readyA + R * nbCap + (R - 1), epoch);
            EQF_HSTAMP(12);
        };
        factor64<true>(s, tid, &bad, pre, D + (long long)R * kDRec, EQF_F64_ST(), realStages(ch.kind == 0 ? sDim(g.N) : eDim(g.N), kSB * R), mid,
            stageOut, epoch);
        EQF_HSTAMP(7);
        hoDrain();
        __syncthreads();
        if (tid == 0 && bad != 8) hoPublish(ch.flags + (long long)b * ch.strideF + R, epoch);
        EQF_HSTAMP(8);
    } else if (role.role == 1) {
        // =========================================================================================== T(R, C)
        const int R = role.R, C = role.C;
        if (R >= nb) return;
        if (!srcSigma) waitPrep();
        f64x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][q] = tile0(R, C, kQB * wv + (lane >> 4) + 4 * q, kQB * i + (lane & 15));
        if (C > 0) {
            // pipelined panel loop (panelIssue above): panel K+1 is fetched during the products of panel K whenever its two blocks are
            // already out (a look at the flags one iteration ahead, never a wait); at the frontier -- the newest panel not published
            // yet -- it degenerates to wait, load, multiply as before
            __shared__ int sAhead;
            PanelRegs pr;
            const double* rowP = A + (long long)(R * kSB) * ldA;
            const double* rowQ = A + (long long)(C * kSB) * ldA;
            hoWait3(readyA + R * nbCap, readyA + C * nbCap, nullptr, epoch, tid, &bad, ra.errflag);
            panelIssue(pr, rowP, ldA, rowQ, ldA, tid);
            bool probe = (tid == 0 && C > 1) ? hoProbe2(readyA + R * nbCap + 1, readyA + C * nbCap + 1, epoch) : false;
            for (int K = 0; K < C; ++K) {
                panelToLds(pr, s.P, s.Q, tid);
                if (tid == 0) sAhead = probe ? 1 : 0;
                __syncthreads();
                const bool ahead = K + 1 < C && sAhead;
                if (ahead) {
                    panelIssue(pr, rowP + (K + 1) * kSB, ldA, rowQ + (K + 1) * kSB, ldA, tid);
                    probe = (tid == 0 && K + 2 < C) ? hoProbe2(readyA + R * nbCap + K + 2, readyA + C * nbCap + K + 2, epoch) : false;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mmTile<true, kSB>(acc[i], &s.P[0][0], kSP, kQB * wv, &s.Q[0][0], kSP, kQB * i, lane, -1.0);
                __syncthreads();
                if (K + 1 < C && !ahead) {
                    hoWait3(readyA + R * nbCap + K + 1, readyA + C * nbCap + K + 1, nullptr, epoch, tid, &bad, ra.errflag);
                    panelIssue(pr, rowP + (K + 1) * kSB, ldA, rowQ + (K + 1) * kSB, ldA, tid);
                    probe = (tid == 0 && K + 2 < C) ? hoProbe2(readyA + R * nbCap + K + 2, readyA + C * nbCap + K + 2, epoch) : false;
                }
            }
        }
        // (Round 3 also tried the staged consumption of stagedPanelSolve for the tile T(C+2, C), whose block is the other thing the row
        // head H(C+2) waits for: no change, 137.8 against 138.3 us per update -- once the heads consume D stage by stage the pivot chain
        // and its hand-off are the critical path again, 13.6 us per block column: factor64 10.3, store issue + first column 1.1, flag +
        // W_33 0.9, last solve step 0.3, publish 0.45.)
        // Round 5: it is tried again for T(C+2, C) only, because the stamps changed: the block this tile publishes is the row head H(C+2)'s OWN
        // block of its last panel, and that head's work on it (a 32 KB load and the diagonal tile's products, 3.7 us) now ends 1.4 us AFTER the
        // other block of the panel -- the one the previous head publishes behind its first pivots -- is out: the later this tile publishes, the
        // later the head sees that block (profiles/r05_res_stamps_N200.txt).  Consumed stage by stage, D[C] costs this tile one round trip of
        // 2 KB and one 16 x 16 x 16 product behind the producer's last pivot instead of the whole 40 KB record and the whole solve.
#ifndef EQF_T_STAGED
#define EQF_T_STAGED 1
#endif
        if (EQF_T_STAGED && C > 0 && C + 2 == R && ra.stageFlags) {
            stagedPanelSolve(acc, s, D + (long long)C * kDRec, ra.stageFlags + (((long long)b * 2 + role.kind) * nbCap + C) * 4, flagD + C, epoch, tid, &bad,
                ra.errflag);
        } else {
            hoWait3((C > 0 || waitD0) ? flagD + C : nullptr, nullptr, nullptr, epoch, tid, &bad, ra.errflag);
            hoLoadRecord(D + (long long)C * kDRec, s, tid);
#pragma unroll
            for (int i = 0; i < 4; ++i) stTile(acc[i], &s.P[0][0], kSP, kQB * wv, kQB * i, lane);
            __syncthreads();
            solveStrip<true>(&s.P[0][0], kSP, s, kQB * wv, lane);
            __syncthreads();
        }
        hoStoreBlock(A + (long long)(R * kSB) * ldA + C * kSB, ldA, s.P, tid);
        hoDrain();
        __syncthreads();
        if (tid == 0 && bad != 8) hoPublish(readyA + R * nbCap + C, epoch);
        if (C + 2 == R) EQF_HSTAMP(11);
    } else {
        // =========================================================================================== W(t, C)
        const int t = role.R, C = role.C;
        if (t >= wt || C >= nb) return;
        const double* Tg = W + (long long)(C * kSB) * ldW + t * kSB;
        waitPrep();
        f64x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double* p = Tg + (long long)(kQB * wv + (lane >> 4) + 4 * q) * ldW + kQB * i + (lane & 15);
                acc[i][q] = fold ? hoLoad8(p) : *p;
            }
        const bool isS = ch.kind == 0;
        double* zv = s.redL;  // [0, 64) the innovation column delta of this block row, [64, 128) z of block row K
        const int nvv = kLm0 + 3 * g.N;
        // The last right-hand-side workgroup of the E-chain is the END of the update's critical path (D[nb-1] -> its solve -> G11 -> innovation
        // lift): what its epilogue needs of the S-chain -- the complete sums, which the S-chain's last block row carries; it finished several
        // block columns ago -- is collected while this workgroup waits for its LAST panel (1.4 us off the path), and its own Y tile, share
        // of G11 and flag, which nobody reads, are not stored, drained and published any more (1.6 us).
        const bool last = ch.kind == 1 && C == nb - 1;
        // (the co-resident kernel only: in the variant for grids larger than the chip the same changes cost 3.5 % -- 8 filters 181 -> 187 us)
        const bool lastFast = last && !PIPEH;
        bool collected = false;
        auto collect = [&]() {
            collected = true;
            const int* ryS = ra.readyY + ((long long)b * 2 + 0) * nbCap * wtCap;
            if (tid < wtS && !hoWait(ryS + (nbS - 1) * wtCap + tid, epoch, ra.errflag)) bad = 8;
            bad = __syncthreads_or(bad) ? 8 : 0;  // (a timeout seen by ANY of the polling threads is reported below by thread 0)
            double* gam = a.dbgGamma + (long long)b * (kLm0 + 3 * a.cap);
            const int ldY = ra.c0.ldW;
            double gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int col = tid + 256 * u;
                gv[u] = col < nvv + 6 ? hoLoad8(ra.gammaPart + ((long long)b * nbCap + nbS - 1) * ldY + col) : 0.0;
            }
            for (int col = tid + 1024; col < nvv + 6; col += 256) {  // (more than 1024 columns: N > 335)
                const double v = hoLoad8(ra.gammaPart + ((long long)b * nbCap + nbS - 1) * ldY + col);
                if (col < nvv) gam[col] = v;
                else s.redL[col - nvv] = v;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int col = tid + 256 * u;
                if (col < nvv) gam[col] = col == 11 ? 0.0 : gv[u];
                else if (col < nvv + 6) s.redL[col - nvv] = gv[u];
            }
            if (lastFast) {  // (the part of the innovation lift that only needs gamma, now: updateFinishBody)
                __syncthreads();
                updateFinishBody(a, b, s.redL, 1);
            }
        };
        if (isS && tid < kSB) zv[tid] = fold ? hoLoad8(W + (long long)(C * kSB + tid) * ldW + 11) : W[(long long)(C * kSB + tid) * ldW + 11];
        if (C > 0) {
            // pipelined panel loop (see panelIssue): Q <- L_{C,K}, P <- Y_{K,t}; the S-chain also carries z_K along
            __shared__ int sAheadW;
            PanelRegs pr;
            double zNext = 0.0;
            const double* rowQ = A + (long long)(C * kSB) * ldA;
            const double* colP = WO + t * kSB;
            auto issue = [&](int K) {
                panelIssue(pr, colP + (long long)(K * kSB) * ldW, ldW, rowQ + K * kSB, ldA, tid);
                if (isS && tid < kSB) zNext = hoLoad8(WO + (long long)(K * kSB + tid) * ldW + 11);
            };
            auto look = [&](int K) {
                return (tid == 0 && K < C) ? hoProbe3(readyA + C * nbCap + K, readyY + K * wtCap + t, isS ? readyY + K * wtCap : nullptr, epoch) : false;
            };
            // At the frontier -- the newest panel not complete yet -- its two blocks come from different producers at different times: L_{C,K}
            // from the row head H(C) / an interior tile a few microseconds after D[K], Y_{K,t} from the right-hand-side tile above, which
            // itself waited for the one above it: on a grid larger than the chip the right-hand-side tiles of a column are a chain of
            // their own that lags the row heads (8 filters of N = 200: the S-chain's last Y tile 36 us after its last D,
            // profiles/r05_res_stamps_B8_before.txt).  So L_{C,K} is asked for as soon as ITS flag is up and only Y_{K,t} travels behind the
            // late flag: 32 KB on the chain instead of 64.
            auto issueSplit = [&](int K) {
                hoWait3(readyA + C * nbCap + K, nullptr, nullptr, epoch, tid, &bad, ra.errflag);
                panelIssueQ(pr, rowQ + K * kSB, ldA, tid);
                hoWait3(readyY + K * wtCap + t, isS ? readyY + K * wtCap : nullptr, nullptr, epoch, tid, &bad, ra.errflag);
                panelIssueP(pr, colP + (long long)(K * kSB) * ldW, ldW, tid);
                if (isS && tid < kSB) zNext = hoLoad8(WO + (long long)(K * kSB + tid) * ldW + 11);
            };
            issueSplit(0);
            bool probe = look(1);
            for (int K = 0; K < C; ++K) {
                panelToLds(pr, s.P, s.Q, tid);
                if (isS && tid < kSB) zv[kSB + tid] = zNext;
                if (tid == 0) sAheadW = probe ? 1 : 0;
                __syncthreads();
                const bool ahead = K + 1 < C && sAheadW;
                if (ahead) {
                    issue(K + 1);
                    probe = look(K + 2);
                }
                if (isS && wv == 1) {  // delta -= L_CK z_K
                    double d = zv[lane];
#pragma unroll 8
                    for (int k = 0; k < kSB; ++k) d = fma(-s.Q[lane][k], zv[kSB + k], d);
                    zv[lane] = d;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mmTile<false, kSB>(acc[i], &s.Q[0][0], kSP, kQB * wv, &s.P[0][0], kSP, kQB * i, lane, -1.0);
                __syncthreads();
                if (K + 1 < C && !ahead) {
                    if (lastFast && K + 2 == C) collect();  // (the last panel is not out yet: this wait is idle time on the critical path's side)
                    issueSplit(K + 1);
                    probe = look(K + 2);
                }
            }
        }
        if (last && !collected) collect();
        EQF_WSTAMP(0);
        hoWait3((C > 0 || waitD0) ? flagD + C : nullptr, nullptr, nullptr, epoch, tid, &bad, ra.errflag);
        EQF_WSTAMP(1);
        hoLoadRecord(D + (long long)C * kDRec, s, tid);
        // running sums of the reductions: block row C adds its share to what block row C-1 of the same column tile left
        // (published together with that tile, which this workgroup has already waited for): a fixed summation order
        double prevSum = 0.0;
        if (C > 0) {
            if (isS && tid < kSB && t * kSB + tid < nvv + 6) prevSum = hoLoad8(ra.gammaPart + ((long long)b * nbCap + C - 1) * ldW + t * kSB + tid);
            if (!isS && tid < 121) prevSum = hoLoad8(ra.g11Part + ((long long)b * nbCap + C - 1) * 128 + tid);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) stTile(acc[i], &s.P[0][0], kSP, kQB * wv, kQB * i, lane);
        __syncthreads();
        solveStrip<false>(&s.P[0][0], kSP, s, kQB * wv, lane);
        if (isS && wv == 3) solveVec64(zv, s, lane);
        __syncthreads();
        EQF_WSTAMP(2);
        if (!lastFast) hoStoreBlock(WO + (long long)(C * kSB) * ldW + t * kSB, ldW, s.P, tid);
        double g11Tot = 0.0;
        if (isS) {
            // gamma[col] = sum_C Y_C[:, col] . z_C (columns nvv .. nvv+5: hV), through block row C
            const int col = t * kSB + tid;
            if (tid < kSB && col < nvv + 6) {
                double v = 0.0;
#pragma unroll 8
                for (int r = 0; r < kSB; ++r) v = fma(s.P[r][tid], zv[r], v);
                hoStore8(ra.gammaPart + ((long long)b * nbCap + C) * ldW + col, prevSum + v);
            }
        } else if (tid < 121) {
            const int q0 = tid / 11, q1 = tid % 11;
            double v = 0.0;
#pragma unroll 8
            for (int r = 0; r < kSB; ++r) v = fma(s.P[r][q0], s.P[r][q1], v);
            g11Tot = prevSum + v;
            if (!lastFast) hoStore8(ra.g11Part + ((long long)b * nbCap + C) * 128 + tid, g11Tot);
        }
        if (!lastFast) {
            hoDrain();
            __syncthreads();
            if (tid == 0 && bad != 8) {
                hoPublish(readyY + C * wtCap + t, epoch);
                if (isS) __hip_atomic_fetch_add(counters + 0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        EQF_WSTAMP(3);
        if (last) {
            // ---- the innovation lift (the sums of the S-chain were collected above)
            if (tid < 121) s.redL[8 + tid] = g11Tot;
            __syncthreads();
            EQF_WSTAMP(4);
            updateFinishBody(a, b, s.redL, lastFast ? 2 : 0);
            __syncthreads();
            EQF_WSTAMP(5);
        }
    }
    if (bad && ra.errflag && tid == 0) atomicOr(ra.errflag, bad == 8 ? kHoErrTimeout : 4);  // (`bad == 8` is this kernel's LOCAL code for a failed wait)
    // A hand-off that timed out left this workgroup with stale operands.  The sticky flag (bit 128, include/eqf_vio_amd.h) makes the host
    // report EQF_ERR_NUMERIC, and the workgroup has published NOTHING since (`bad != 8` at every publish above): the workgroups that depend
    // on it time out in turn, the count of finished Y tiles stays short, the downdate tiles give up -- Sigma_out is not overwritten with
    // a plausible-looking wrong matrix.

    // ---- covariance downdate Sigma - Y^T Y: the tile workgroups behind the roles (top of the kernel).  Until late in round 3 a co-resident
    // grid did it here instead -- every workgroup that was done with its role waited, holding its CU, until all Y tiles were out and then
    // shared the tiles from a counter: the one wait of the kernel for HIGHER block indices (safe only with full co-residency; the advisor's
    // round-2 finding) and 2 us slower than the tile workgroups (136.7 -> 134.8 us per update).  Removed.  (Also tried, for grids larger
    // than the chip: finished role workgroups take tiles only if the last Y tile is already out and leave otherwise -- 2 filters 294 us
    // against 135 + 41 us with a follow-up launch; too few workgroups finish after the S-chain.)
}
#ifdef EQF_WAIT_STATS
template <typename T, bool PIPEH = false, bool OCC2 = false, bool FOLD = false, bool TICKET = false>
__global__ __launch_bounds__(256, OCC2 ? 2 : 1) void k_chol_resident(ResArgs ra) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        sStatClass = 7;
        sStatWait = 0;
    }
    __syncthreads();
    residentBody<T, PIPEH, OCC2, FOLD, TICKET>(ra);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&g_waitStats[sStatClass & 15][0], sStatWait);
        atomicAdd(&g_waitStats[sStatClass & 15][1], (unsigned long long)(wall_clock64() - t0));
        atomicAdd(&g_waitStats[sStatClass & 15][2], 1ull);
    }
}
#endif

}  // namespace eqf
