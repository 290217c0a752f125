"""Sigma 2-D block-partitioned over a process grid (BASELINE configs[4]: N = 4000 landmarks, Sigma = 1.15 GB fp64, SURVEY.md
8(e) row 2) -- the exchange schedule and the tile-local mathematics, one process per GPU over torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

What is partitioned.  Sigma in the reference's index map (eqf_vio/src/VIOFilter.cpp:54-57): 11 base coordinates, then 3 per
landmark.  Landmarks are cut into blocks of `bl`; the landmark-landmark part of Sigma is an nb x nb grid of (3 bl x 3 bl) tiles
dealt block-cyclically over a Pr x Pc process grid (tile (I, J) lives on process (I mod Pr, J mod Pc); 2 x 4 for one 8-GPU
node).  The 11 x n base panel (88 KB per 1000 landmarks) is REPLICATED on every rank, and so is the O(N) filter state.

Riccati propagate (VIOFilter.cpp:160-194), F = I + T A_b = [[F_bb, 0], [L, D]] with D block-diagonal:
    Sigma'_IJ = (D_I Sigma_IJ + L_I Sigma_bJ) D_J^T + (L_I Sigma_bb + D_I Sigma_Ib) L_J^T + Q_IJ      tile-local
    Sigma'_bJ = F_bb (Sigma_bb L_J^T + Sigma_bJ D_J^T) + Q_bJ ,  Sigma'_bb = F_bb Sigma_bb F_bb^T + Q_bb   replicated
  -> NO communication: every rank advances its own tiles and its copy of the base panel.

Update (VIOFilter.cpp:264-297) in the Cholesky form the single-GPU path uses (csrc/eqf_update.hpp):
    S_IJ = C_I Sigma_IJ C_J^T (+ R)          tile-local (C is block diagonal, EqFMatrices.cpp:319-344)
    S = L L^T, Y = L^-1 [C Sigma | delta]    blocked right-looking Cholesky over the process grid, block column k:
        1. the owner of S_kk factors it and broadcasts L_kk                                   (one broadcast, 2bl x 2bl)
        2. the owners solve their panel blocks L_ik = S_ik L_kk^-T and right-hand-side tiles Y_kt = L_kk^-1 W_kt
        3. the panel column and the block row Y_k are gathered to every rank                  (one all-gather each)
        4. every rank updates its own trailing tiles S_ij -= L_ik L_jk^T, W_it -= L_ik Y_kt, and -- with Y_k at hand --
           its share of the downdate Sigma_IJ -= Y_kI^T Y_kJ and of gamma = K delta = sum_k Y_k^T z_k   (no further traffic)
    the base panel is downdated from the same Y_k on every rank (replicated, identical arithmetic).
  Per update every rank receives the lower half of L once (m^2/2 values) and Y once (m n values): at N = 4000 that is
  0.26 + 0.77 GB against 4.6e12 flops / 8 -- communication-bound on xGMI unless panels are restricted to the process rows /
  columns that need them (SUMMA); the schedule below gathers to all ranks because that is the simplest correct one.
bundleLift's weights (EqFMatrices.cpp:239, Sigma_e = Sigma[6:, 6:]) come from the same distributed solver after the five base
coordinates of Sigma_e have been eliminated locally (a Schur complement every rank can form from the replicated panel).

Status: design + exchange schedule validated on CPU with gloo against the single-process fp64 reference filter
(tests/test_tiled.py, tile-local mathematics in torch).  On a GPU the two operations that touch every tile every step -- the
Riccati step and the downdate -- run in hand-written tile kernels behind the C ABI (eqf_tile_propagate, eqf_tile_downdate:
csrc/eqf_tile.hpp, called on the torch tensors' device pointers; ProcessGrid(..., kernels=TileKernels(dev))), checked on the
MI355X against the torch path (tests/test_gpu_tiled.py); the panel operations of the distributed Cholesky (potrf / trsm of
2bl x 2bl blocks) are still torch (rocBLAS / rocSOLVER).  A measured 8-GPU run is outstanding: no 8-GPU node has been available
to this build.
"""
import ctypes

import torch


class TileKernels:
    """The tile kernels of csrc/eqf_tile.hpp through the C ABI, on torch CUDA tensors (their device pointers) and torch's
    current stream, so that they order with the torch operations around them."""

    def __init__(self, device_index=0):
        from . import binding

        self.lib = binding.lib()
        self.dev = int(device_index)
        self._dp = ctypes.POINTER(ctypes.c_double)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr())

    def propagate(self, inp, nI, nJ, D_I, L_I, D_J, L_J, Sbb, SbI, ldbI, SbJ, ldbJ, BnI, BnJ, R6, T, diag_noise, is_diag):
        out = torch.empty_like(inp)
        r6 = (ctypes.c_double * 6)(*[float(x) for x in R6])
        rc = self.lib.eqf_tile_propagate(self.dev, self._stream(), self._p(out), self._p(inp), inp.stride(0), nI, nJ, self._p(D_I), self._p(L_I),
                                         self._p(D_J), self._p(L_J), self._p(Sbb), self._p(SbI), ldbI, self._p(SbJ), ldbJ, self._p(BnI),
                                         self._p(BnJ), ctypes.cast(r6, self._dp), float(T), float(diag_noise), int(is_diag))
        if rc:
            raise RuntimeError(f"eqf_tile_propagate failed with status {rc}")
        return out

    def downdate(self, C, A, B):
        """C -= A^T B in place (C m x n, A k x m, B k x n; row-major, unit column stride)."""
        if A.stride(1) != 1:
            A = A.contiguous()  # (LAPACK-backed torch solves hand back column-major strides)
        if B.stride(1) != 1:
            B = B.contiguous()
        assert C.stride(1) == 1
        rc = self.lib.eqf_tile_downdate(self.dev, self._stream(), self._p(C), C.stride(0), C.shape[0], C.shape[1], self._p(A), A.stride(0),
                                        self._p(B), B.stride(0), A.shape[0])
        if rc:
            raise RuntimeError(f"eqf_tile_downdate failed with status {rc}")


    DREC = 64 * 64 + 4 * 16 * 16  # doubles per 64-wide block column: L_jj and the inverses of its four 16 x 16 diagonal blocks

    def potrf(self, A):
        """Cholesky of the n x n block A (lower triangle read): returns (L, drec); L = the lower-triangular factor (fresh tensor),
        drec = the diagonal-factor records trsm() multiplies with."""
        n = A.shape[0]
        L = torch.tril(A).contiguous()
        drec = torch.empty(((n + 63) // 64) * self.DREC, dtype=torch.float64, device=A.device)
        info = torch.zeros(1, dtype=torch.int32, device=A.device)
        rc = self.lib.eqf_tile_potrf(self.dev, self._stream(), self._p(L), L.stride(0), n, self._p(drec), self._p(info))
        if rc:
            raise RuntimeError(f"eqf_tile_potrf failed with status {rc}")
        self._info = info  # checked by the caller when it synchronises anyway (dist_chol_solve: after the broadcast)
        return L, drec

    def trsm(self, L, drec, B, right):
        """right: B (m x n) <- B L^-T ; left: B (n x m) <- L^-1 B.  Returns the solved block (B is copied first)."""
        X = B.contiguous().clone()
        m = X.shape[0] if right else X.shape[1]
        rc = self.lib.eqf_tile_trsm(self.dev, self._stream(), self._p(L), L.stride(0), L.shape[0], self._p(drec), self._p(X), X.stride(0), m,
                                    1 if right else 0)
        if rc:
            raise RuntimeError(f"eqf_tile_trsm failed with status {rc}")
        return X


class ProcessGrid:
    """Pr x Pc process grid over a torch.distributed group; tile (I, J) -> rank (I mod Pr) * Pc + (J mod Pc)."""

    def __init__(self, dist, Pr, Pc, device="cpu", kernels=None):
        self.dist, self.Pr, self.Pc = dist, Pr, Pc
        self.kernels = kernels  # TileKernels (GPU) or None (tile mathematics in torch)
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        assert self.world == Pr * Pc
        self.pr, self.pc = divmod(self.rank, Pc)
        self.device = device

    def owner(self, I, J):
        return (I % self.Pr) * self.Pc + (J % self.Pc)

    def mine(self, I, J):
        return self.owner(I, J) == self.rank

    def bcast(self, t, src):
        if self.dist is not None and self.world > 1:
            self.dist.broadcast(t, src=src)
        return t

    def allgather_blocks(self, mine, shape):
        """mine: {key: tensor(shape)} of the blocks this rank contributes; returns {key: tensor} of everybody's.  One
        all-gather of a padded [slots, *shape] buffer + an int key table (ragged counts per rank)."""
        if self.dist is None or self.world == 1:
            return dict(mine)
        cnt = torch.tensor([len(mine)], dtype=torch.int64, device=self.device)
        cnts = [torch.zeros_like(cnt) for _ in range(self.world)]
        self.dist.all_gather(cnts, cnt)
        slots = max(int(c.item()) for c in cnts)
        if slots == 0:
            return {}
        buf = torch.zeros((slots,) + tuple(shape), dtype=torch.float64, device=self.device)
        keys = torch.full((slots, 2), -1, dtype=torch.int64, device=self.device)
        for s, (k, v) in enumerate(sorted(mine.items())):
            buf[s] = v
            keys[s, 0], keys[s, 1] = k
        bufs = [torch.empty_like(buf) for _ in range(self.world)]
        keyss = [torch.empty_like(keys) for _ in range(self.world)]
        self.dist.all_gather(bufs, buf)
        self.dist.all_gather(keyss, keys)
        out = {}
        for r in range(self.world):
            for s in range(int(cnts[r].item())):
                out[(int(keyss[r][s, 0]), int(keyss[r][s, 1]))] = bufs[r][s]
        return out


class TiledSigma:
    """The rank's share of Sigma: owned landmark tiles {(I, J): (3bl x 3bl)}, replicated base block Sbb (11 x 11) and base panel
    Sb (11 x 3N)."""

    def __init__(self, grid, N, bl):
        assert N % bl == 0, "the prototype wants whole landmark blocks"
        self.g, self.N, self.bl, self.nb = grid, N, bl, N // bl
        self.t = {}
        self.Sbb = None
        self.Sb = None

    @classmethod
    def from_dense(cls, grid, Sigma, bl):
        """Scatter by slicing a dense Sigma every rank holds (test set-up / restart from a single-GPU snapshot)."""
        S = torch.as_tensor(Sigma, dtype=torch.float64, device=grid.device)
        N = (S.shape[0] - 11) // 3
        ts = cls(grid, N, bl)
        ts.Sbb = S[:11, :11].clone()
        ts.Sb = S[:11, 11:].clone()
        w = 3 * bl
        for I in range(ts.nb):
            for J in range(ts.nb):
                if grid.mine(I, J):
                    ts.t[(I, J)] = S[11 + I * w:11 + (I + 1) * w, 11 + J * w:11 + (J + 1) * w].clone()
        return ts

    def to_dense(self):
        """Gather to a dense Sigma on every rank (tests, snapshots)."""
        w = 3 * self.bl
        allt = self.g.allgather_blocks(self.t, (w, w))
        n = 11 + 3 * self.N
        S = torch.zeros((n, n), dtype=torch.float64, device=self.g.device)
        S[:11, :11] = self.Sbb
        S[:11, 11:] = self.Sb
        S[11:, :11] = self.Sb.T
        for (I, J), v in allt.items():
            S[11 + I * w:11 + (I + 1) * w, 11 + J * w:11 + (J + 1) * w] = v
        return S

    def cols(self, J):
        w = 3 * self.bl
        return slice(J * w, (J + 1) * w)


def propagate(ts, Fbb, L, D, Qbb, Bn, Rdiag, T, point_var):
    """One Riccati step, tile-local (no communication).  Fbb 11 x 11; L (3N x 11) = rows of F below the base block; D (N, 3, 3)
    the diagonal blocks of F; process noise Q = T (P + Bn R Bn^T) with Bn = (n x 6) input matrix (first 6 rows zero), R = diag
    (6,), P = diag(..., point_var on every landmark coordinate); Qbb = its 11 x 11 base block."""
    bl, nb = ts.bl, ts.nb
    w = 3 * bl
    Dm = [torch.block_diag(*D[I * bl:(I + 1) * bl]) for I in range(nb)]
    Lr = [L[I * w:(I + 1) * w] for I in range(nb)]
    BR = Bn * Rdiag  # (n x 6) columns scaled
    Sbb, Sb = ts.Sbb, ts.Sb
    new = {}
    kern = ts.g.kernels
    if kern is not None:
        Dc, Lc, Bc, Sbbc, Sbc = D.contiguous(), L.contiguous(), Bn.contiguous(), Sbb.contiguous(), Sb.contiguous()
        for (I, J), S_IJ in ts.t.items():
            new[(I, J)] = kern.propagate(S_IJ.contiguous(), bl, bl, Dc[I * bl:], Lc[I * w:], Dc[J * bl:], Lc[J * w:], Sbbc, Sbc[:, I * w:],
                                         Sbc.stride(0), Sbc[:, J * w:], Sbc.stride(0), Bc[11 + I * w:], Bc[11 + J * w:], Rdiag.tolist(), T,
                                         T * point_var, I == J)
    for (I, J), S_IJ in ([] if kern is not None else ts.t.items()):
        SIb = Sb[:, ts.cols(I)].T  # Sigma_Ib = Sigma_bI^T
        G_I = Lr[I] @ Sbb + Dm[I] @ SIb
        Q_IJ = T * (BR[11 + I * w:11 + (I + 1) * w] @ Bn[11 + J * w:11 + (J + 1) * w].T)
        if I == J:
            Q_IJ = Q_IJ + T * point_var * torch.eye(w, dtype=torch.float64, device=S_IJ.device)
        new[(I, J)] = (Dm[I] @ S_IJ + Lr[I] @ Sb[:, ts.cols(J)]) @ Dm[J].T + G_I @ Lr[J].T + Q_IJ
    Sb_new = torch.empty_like(Sb)
    for J in range(nb):
        Sb_new[:, ts.cols(J)] = Fbb @ (Sbb @ Lr[J].T + Sb[:, ts.cols(J)] @ Dm[J].T) + T * (BR[:11] @ Bn[11 + J * w:11 + (J + 1) * w].T)
    ts.Sbb = Fbb @ Sbb @ Fbb.T + Qbb
    ts.Sb = Sb_new
    ts.t = new


def dist_chol_solve(grid, nb, A, Wt, Wn, bs, wt, on_row=None):
    """Blocked right-looking Cholesky of the SPD matrix A (nb x nb blocks of bs, lower blocks {(i, j), i >= j} on their owners)
    with right-hand sides: wide tiles Wt {(i, t)} (bs x wt, owner (i mod Pr, t mod Pc)) and one narrow tile per block row Wn
    {(i, 0)} (bs x nn, owner (i mod Pr, 0)).  A and W are consumed.  After block column k the solved block row is handed to
    on_row(k, Yk_wide {t: bs x wt}, Yk_narrow) ON EVERY RANK (this is where the downdate and the reductions hang)."""
    nn = next(iter(Wn.values())).shape[1] if Wn else 0
    nn_all = torch.tensor([nn], dtype=torch.int64, device=grid.device)
    if grid.dist is not None and grid.world > 1:
        grid.dist.all_reduce(nn_all, op=grid.dist.ReduceOp.MAX)
    nn = int(nn_all.item())
    kern = grid.kernels
    for k in range(nb):
        # 1. diagonal block (GPU: hand-written k_tile_potrf, which also leaves the records the panel solves multiply with)
        Lkk = torch.empty((bs, bs), dtype=torch.float64, device=grid.device)
        drec = torch.empty(((bs + 63) // 64) * TileKernels.DREC, dtype=torch.float64, device=grid.device) if kern is not None else None
        if grid.mine(k, k):
            if kern is not None:
                Lkk, drec = kern.potrf(A.pop((k, k)))
            else:
                Lkk = torch.linalg.cholesky(A.pop((k, k))).contiguous()  # (LAPACK hands back column-major strides)
        grid.bcast(Lkk, grid.owner(k, k))
        if kern is not None:
            grid.bcast(drec, grid.owner(k, k))

        def solve_right(Bm):  # B Lkk^-T
            return kern.trsm(Lkk, drec, Bm, True) if kern is not None else torch.linalg.solve_triangular(Lkk, Bm.T, upper=False).T

        def solve_left(Bm):  # Lkk^-1 B
            return kern.trsm(Lkk, drec, Bm, False) if kern is not None else torch.linalg.solve_triangular(Lkk, Bm, upper=False)

        # 2. panel blocks and this block row of right-hand sides, on their owners
        pan = {}
        for (i, j) in [key for key in A if key[1] == k]:
            pan[(i, k)] = solve_right(A.pop((i, j)))
        yw = {}
        for (i, t) in [key for key in Wt if key[0] == k]:
            yw[(k, t)] = solve_left(Wt.pop((i, t)))
        yn = {}
        if (k, 0) in Wn:
            yn[(k, 0)] = solve_left(Wn.pop((k, 0)))
        # 3. everybody gets the panel column and the block row
        pan = grid.allgather_blocks(pan, (bs, bs))
        yw = grid.allgather_blocks(yw, (bs, wt))
        yn = grid.allgather_blocks(yn, (bs, nn)) if nn else {}
        # 4. trailing updates of what this rank owns
        for (i, j) in A:
            if j > k:
                A[(i, j)] -= pan[(i, k)] @ pan[(j, k)].T
        for (i, t) in Wt:
            if i > k:
                Wt[(i, t)] -= pan[(i, k)] @ yw[(k, t)]
        for (i, _) in Wn:
            if i > k:
                Wn[(i, 0)] -= pan[(i, k)] @ yn[(k, 0)]
        if on_row is not None:
            on_row(k, {t: v for (_, t), v in yw.items()}, yn.get((k, 0)))


def update(ts, C, delta, meas_var):
    """Sigma <- Sigma - K C Sigma and gamma = K delta (VIOFilter.cpp:276-297) over the process grid.  C (N, 2, 3): the blocks of
    EqFOutputMatrixC (one per landmark, acting on its three coordinates); delta (2N,).  Returns gamma (11 + 3N,), the same on
    every rank.  Sigma is downdated in place from the solved block rows as they arrive."""
    g, bl, nb = ts.g, ts.bl, ts.nb
    w, bs = 3 * bl, 2 * bl
    dev = g.device
    Cm = [torch.block_diag(*C[I * bl:(I + 1) * bl]) for I in range(nb)]  # (2bl x 3bl)
    # S tiles (lower) and right-hand sides, all tile-local
    A, Wt, Wn = {}, {}, {}
    for (I, J), S_IJ in ts.t.items():
        CS = Cm[I] @ S_IJ  # (C Sigma)_IJ
        Wt[(I, J)] = CS
        if I >= J:
            Sij = CS @ Cm[J].T
            if I == J:
                Sij = Sij + meas_var * torch.eye(bs, dtype=torch.float64, device=dev)
            A[(I, J)] = Sij
    d = torch.as_tensor(delta, dtype=torch.float64, device=dev)
    for I in range(nb):
        if g.mine(I, 0):
            Wn[(I, 0)] = torch.cat([Cm[I] @ ts.Sb[:, ts.cols(I)].T, d[I * bs:(I + 1) * bs, None]], dim=1)  # [(C Sigma)_Ib | delta_I]
    n = 11 + 3 * ts.N
    gamma = torch.zeros(n, dtype=torch.float64, device=dev)
    Sb_dd = torch.zeros_like(ts.Sb)
    Sbb_dd = torch.zeros_like(ts.Sbb)

    def on_row(k, Yw, Yn):
        z = Yn[:, 11]
        Yb = Yn[:, :11]
        gamma[:11] += Yb.T @ z
        Sbb_dd.add_(Yb.T @ Yb)
        for J in range(nb):
            gamma[11 + J * w:11 + (J + 1) * w] += Yw[J].T @ z
            Sb_dd[:, ts.cols(J)] += Yb.T @ Yw[J]
        for (I, J) in ts.t:
            if g.kernels is not None:
                g.kernels.downdate(ts.t[(I, J)], Yw[I], Yw[J])
            else:
                ts.t[(I, J)] -= Yw[I].T @ Yw[J]

    dist_chol_solve(g, nb, A, Wt, Wn, bs, w, on_row)
    ts.Sb = ts.Sb - Sb_dd
    ts.Sbb = ts.Sbb - Sbb_dd
    return gamma


def sigma_e_quadratic_form(ts, V):
    """G = V^T Sigma_e^-1 V for Sigma_e = Sigma[6:, 6:] (bundleLift's weights, EqFMatrices.cpp:239) and V ((5 + 3N) x q) given
    on every rank: the five base coordinates are eliminated locally from the replicated panel, the Schur complement of the
    landmark tiles goes through the distributed solver, the q x q result is accumulated from the solved block rows."""
    g, bl, nb = ts.g, ts.bl, ts.nb
    w = 3 * bl
    V = torch.as_tensor(V, dtype=torch.float64, device=g.device)
    q = V.shape[1]
    Sgg = ts.Sbb[6:, 6:]
    Lg = torch.linalg.cholesky(Sgg)
    Pg = torch.linalg.solve_triangular(Lg, ts.Sb[6:, :], upper=False)   # Lg^-1 Sigma_gL   (5 x 3N)
    Vg = torch.linalg.solve_triangular(Lg, V[:5], upper=False)           # Lg^-1 V_g
    G = Vg.T @ Vg
    A, Wn = {}, {}
    for (I, J), S_IJ in ts.t.items():
        if I >= J:
            A[(I, J)] = S_IJ - Pg[:, ts.cols(I)].T @ Pg[:, ts.cols(J)]
    for I in range(nb):
        if g.mine(I, 0):
            Wn[(I, 0)] = V[5 + I * w:5 + (I + 1) * w] - Pg[:, ts.cols(I)].T @ Vg
    acc = torch.zeros((q, q), dtype=torch.float64, device=g.device)

    def on_row(k, Yw, Yn):
        acc.add_(Yn.T @ Yn)

    dist_chol_solve(g, nb, A, {}, Wn, w, w, on_row)
    return G + acc
