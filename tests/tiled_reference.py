"""The exchange schedule of the 2-D block-partitioned filter as an EXECUTABLE SPECIFICATION in Python -- test infrastructure.

Until round 5 this file was the product's host loop (eqf_vio_amd/tiled.py); the product's loop is now C++ behind the C ABI
(csrc/eqf_tiledf.hip, eqf_tf_*), written after this one operation for operation.  It stays for two jobs: (1) the CPU tests drive it with a
numpy double of the per-rank kernels over gloo process grids (tests/test_tiled.py, tests/tiled_double.py) -- the schedule's block-cyclic
index arithmetic, the SUMMA-restricted broadcasts, landmark churn on slots, against the dense oracle; (2) on the GPU the C++ loop is
compared with it bit for bit (tests/test_gpu_tiled.py).  The text below is the round-4 module documentation.

One EqF filter with Sigma 2-D block-partitioned over a process grid -- BASELINE configs[4] (N = 4000 landmarks, Sigma = 1.15 GB
fp64), SURVEY.md 8(e) row 2.  One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on a GPU node, "gloo" in
the CPU tests); the host side of eqf_vio/include/eqf_vio/VIOFilter.h:41-88 for this configuration: `TiledFilter.processIMUData`,
`processVisionData`, `stateEstimate`, `stateCovariance`.

What is partitioned.  Sigma in the reference's index map (eqf_vio/src/VIOFilter.cpp:54-57): 11 base coordinates, then 3 per
landmark.  Landmarks are cut into blocks of `bl`; process (pr, pc) of the Pr x Pc grid owns the landmark blocks I = pr, pr + Pr, ...
as rows and J = pc, pc + Pc, ... as columns of ONE dense local matrix Sll (3 nlr x 3 nlc, ScaLAPACK's block-cyclic local storage), so
every step is a handful of launches over the whole local matrix.  The 11-row base panel Sigma[0:11, :] (88 KB per 1000 landmarks) and
the O(N) filter state are REPLICATED: every rank advances its own identical copy (csrc/eqf_tiled.hpp, the device functions of the
single-GPU path).  Pr must divide Pc (1 x 1, 1 x 2, 2 x 2, 2 x 4 for one 8-GPU node, 1 x 8).

Riccati propagate (VIOFilter.cpp:160-194), F = I + T A_b = [[F_bb, 0], [L, D]] with D block-diagonal:
    Sigma'_IJ = (D_I Sigma_IJ + L_I Sigma_bJ) D_J^T + (L_I Sigma_bb + D_I Sigma_Ib) L_J^T + Q_IJ      local, in place
    Sigma'_bJ = F_bb (Sigma_bb L_J^T + Sigma_bJ D_J^T) + Q_bJ ,  Sigma'_bb = F_bb Sigma_bb F_bb^T + Q_bb   replicated
  -> NO communication.

Update (VIOFilter.cpp:264-297) in the Cholesky form of the single-GPU path (csrc/eqf_update.hpp): S = U^T U, Y = U^-T [C Sigma | C
Sigma_b, delta, V], gamma = Y^T z, Sigma <- Sigma - Y^T Y; bundleLift's weights (EqFMatrices.cpp:239) from a second factorisation, of
the Schur complement of Sigma_e = Sigma[6:, 6:] after its five base coordinates.  Both run through `_chain`: a right-looking blocked
Cholesky that keeps block ROWS (upper factor), so that every dense product has the one form C += alpha A^T B (eqf_tile_gemm_tn).
Per block row k (owner process row k mod Pr):
    1. the owner of the diagonal block factors it (eqf_tile_potrf) and broadcasts L_kk ALONG ITS PROCESS ROW;
    2. that process row solves its pieces of block row k in place:  [U_k,k+1.. | Y_k] = L_kk^-1 [A_k,k+1.. | W_k]  (eqf_tile_trsm);
    3. every rank of the row broadcasts its solved piece DOWN ITS PROCESS COLUMN  (-> the "B operand" of every product);
    4. the ranks (pr, c) with c = pr mod Pr re-broadcast the piece they just received ALONG THEIR PROCESS ROW; interleaved these give
       the "A operand": the blocks U_ki / Y_kI of the rank's own ROW blocks (this needs Pr | Pc);
    5. trailing updates of the local matrix, the rank's share of the downdate Sigma_IJ -= Y_kI^T Y_kJ and of the reductions -- no
       further traffic.
  A rank receives (1/Pr + 1/Pc) of every block row instead of all of it (SUMMA-restricted; the round-2 prototype gathered every panel
  to every rank): at N = 4000 on 2 x 4 that is 0.75 x (0.26 + 0.77) GB per update.

Landmark churn (VIOFilter.cpp:345-443: removeOldLandmarks, removeOutliers, addNewLandmarks) works on SLOTS.  The partition is over
physical landmark slots; a removed landmark leaves an INACTIVE slot where it was (zero rows / columns of Sigma with a unit diagonal
block, identity linearisation, no measurement rows: both factorisations carry it along as a decoupled block of exact zeros), a new
landmark takes the lowest free slot -- so no row or column of Sigma ever moves between ranks, and only the ranks that own a slot's blocks
touch them.  The reference's landmark ORDER (insertion order, VIOFilter.cpp:211-230) is kept here, on the host, as the permutation
`slot_of`; the recursion is equivariant under it, and every getter of TiledFilter answers in the reference's order.  The decisions --
which ids left, which bearings fail the gate, the median scene depth -- are O(N) on the replicated state and identical on every rank.

The tile mathematics is NOT here: `HipBackend` calls the HIP kernels through the C ABI (include/eqf_vio_amd.h, eqf_tiled_* /
eqf_tile_*) on torch CUDA tensors' device pointers, and fails loudly without the library or a GPU.  The CPU tests drive the same
schedule with a test double of the backend (tests/tiled_double.py) over gloo.
"""
import numpy as np
import torch

DREC = 64 * 64 + 4 * 16 * 16  # doubles per 64-wide block column of a diagonal-factor record (DREC)

NARROW_S = 18  # (C Sigma)_Ib (11) | delta | V (6)
NARROW_E = 11  # Z_P (6) | E_top (5)


class BlockCyclic:
    """Geometry of the partition: N landmarks in blocks of bl, block I on process row I mod Pr, block J on process column J mod Pc."""

    def __init__(self, N, bl, Pr, Pc, pr, pc):
        assert N >= 1 and bl >= 1 and Pc % Pr == 0, "the process grid needs Pr | Pc"
        self.N, self.bl, self.Pr, self.Pc, self.pr, self.pc = N, bl, Pr, Pc, pr, pc
        self.nb = (N + bl - 1) // bl
        self.row_blocks = list(range(pr, self.nb, Pr))
        self.col_blocks = list(range(pc, self.nb, Pc))
        self.rowMap = self.landmarks_of(self.row_blocks)
        self.colMap = self.landmarks_of(self.col_blocks)
        self.nlr, self.nlc = len(self.rowMap), len(self.colMap)

    def block_size(self, b):
        return min(self.bl, self.N - b * self.bl)

    def landmarks_of(self, blocks):
        out = [np.arange(b * self.bl, b * self.bl + self.block_size(b), dtype=np.int32) for b in blocks]
        return np.concatenate(out) if out else np.zeros(0, dtype=np.int32)

    def ncols_of(self, c):
        """landmarks in the local columns of process column c"""
        return sum(self.block_size(b) for b in range(c, self.nb, self.Pc))

    @staticmethod
    def blocks_upto(k, p, P):
        """number of blocks b <= k with b mod P == p"""
        return (k - p) // P + 1 if k >= p else 0


class ProcessGrid:
    """Pr x Pc process grid over a torch.distributed group, rank = pr * Pc + pc, with one sub-group per process row / column."""

    def __init__(self, dist, Pr, Pc, device="cpu"):
        self.dist, self.Pr, self.Pc, self.device = dist, Pr, Pc, device
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        assert self.world == Pr * Pc and Pc % Pr == 0
        self.pr, self.pc = divmod(self.rank, Pc)
        self.row_group = self.col_group = None
        if dist is not None and self.world > 1:
            # (new_group is collective over the whole job: every rank creates every group, in the same order)
            for r in range(Pr):
                grp = dist.new_group([r * Pc + c for c in range(Pc)])
                if r == self.pr:
                    self.row_group = grp
            for c in range(Pc):
                grp = dist.new_group([r * Pc + c for r in range(Pr)])
                if c == self.pc:
                    self.col_group = grp

    def bcast_row(self, t, root_pc):
        if self.Pc > 1:
            self.dist.broadcast(t, src=self.pr * self.Pc + root_pc, group=self.row_group)
        return t

    def bcast_col(self, t, root_pr):
        if self.Pr > 1:
            self.dist.broadcast(t, src=root_pr * self.Pc + self.pc, group=self.col_group)
        return t

    def allgather_row(self, t):
        """every rank of my process row contributes t (same shape); returns the Pc pieces.  Done as Pc broadcasts: small, once per update,
        and it works for device tensors on every backend (gloo has no device all_gather)."""
        if self.Pc == 1:
            return [t]
        out = []
        for c in range(self.Pc):
            piece = t if c == self.pc else torch.empty_like(t)
            self.dist.broadcast(piece, src=self.pr * self.Pc + c, group=self.row_group)
            out.append(piece)
        return out

    def allgather_all(self, t):
        if self.world == 1:
            return [t]
        out = []
        for r in range(self.world):
            piece = t if r == self.rank else torch.empty_like(t)
            self.dist.broadcast(piece, src=r)
            out.append(piece)
        return out


class TiledFilter:
    """VIOFilter (VIOFilter.h:41-88) for one filter whose Sigma is partitioned over `grid`.  Every rank of the grid makes the same calls
    with the same arguments.  `backend`: HipBackend (the product path), or the CPU test double.  `capacity`: landmark slots the local
    storage is sized for (default: the backend's capacity)."""

    def __init__(self, grid, backend, block_landmarks, capacity=None):
        self.g, self.be, self.bl = grid, backend, int(block_landmarks)
        self.cap = int(capacity if capacity is not None else backend.cap)
        # the two factorisations of an update are independent: they run side by side on two streams, each with its own exchange buffers
        # and -- on more than one rank -- its own process groups (two communicators: collectives of different streams must not share one)
        # Two RCCL communicators with kernels in flight on different streams of one process can deadlock when the ranks' GPUs schedule them in
        # different orders, and this schedule has never run on more than one GPU: over nccl with more than one rank the chains run one after
        # the other unless EQF_TILED_OVERLAP_CHAINS=1 asks for it (one rank, or gloo -- host-blocking collectives --: side by side).
        import os

        backend_name = grid.dist.get_backend() if (grid.dist is not None and grid.world > 1) else ""
        env = os.environ.get("EQF_TILED_OVERLAP_CHAINS")
        self.overlap_chains = (env != "0") if env is not None else not (grid.world > 1 and backend_name == "nccl")
        self.gE = ProcessGrid(grid.dist, grid.Pr, grid.Pc, grid.device) if grid.world > 1 else grid
        self.geo = None
        self.Sll = self.M = self.E = None
        # landmark bookkeeping (host; identical on every rank): ids in the REFERENCE's order (X.id, VIOFilter.cpp:211-230), the slot of
        # each, and which slots are taken.  nslots = slots in use = 1 + the highest taken slot (at least 1 once storage exists).
        self.ids = None
        self.slot_of = np.zeros(0, dtype=np.int64)
        self.taken = np.zeros(self.cap, dtype=bool)
        self.nslots = 0
        self.churn_stats = dict(removed_old=0, removed_outliers=0, added=0)
        self._queue, self._mirror_time = [], None  # IMU calls waiting for their burst; the filter's time as the queued calls leave it
        self.lookahead = True  # factor the next diagonal block on a second stream in the shadow of the trailing update (_chain)
        self.phase_ms = None  # set to a dict to collect GPU time per phase (bench.py): {"propagate": ms, "prep": ms, "chain_S": ...}
        self._pending = []

    class _Phase:
        """torch.cuda event bracket around a phase of a call, summed into TiledFilter.phase_ms when the filter is asked for its timings
        (no synchronisation inside the loop)."""

        def __init__(self, tf, name):
            self.tf, self.name = tf, name
            self.on = tf.phase_ms is not None and getattr(tf.be, "device", None) is not None and tf.be.device.type == "cuda"

        def __enter__(self):
            if self.on:
                self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.a.record()

        def __exit__(self, *exc):
            if self.on:
                self.b.record()
                self.tf._pending.append((self.name, self.a, self.b))

    def collect_phases(self):
        if self._pending:
            torch.cuda.synchronize()
            for name, a, b in self._pending:
                self.phase_ms[name] = self.phase_ms.get(name, 0.0) + a.elapsed_time(b)
            self._pending = []
        return self.phase_ms

    # ---- storage: allocated ONCE for `cap` slots; the working set is a view of it for the slots in use.  Growing the number of slots
    # never moves a block: only the globally last block is ragged, so the local position of a slot does not depend on how many follow it.
    def _alloc(self):
        g, be, cap = self.g, self.be, self.cap
        full = BlockCyclic(cap, self.bl, g.Pr, g.Pc, g.pr, g.pc)
        r16 = lambda x: (x + 15) // 16 * 16
        self._Sll_buf = be.zeros(max(3 * full.nlr, 1), r16(max(3 * full.nlc, 1)))
        self._M_buf = be.empty(max(2 * full.nlr, 1), r16(5 * full.nlc + NARROW_S))
        self._E_buf = be.empty(max(3 * full.nlr, 1), r16(3 * full.nlc + NARROW_E))
        self.G11 = be.zeros(11, 11)
        # exchange buffers: a solved block row piece per process column of my row (B operand / contributions to the A operand)
        bsmax = 3 * min(self.bl, cap)
        wmax = {c: 5 * full.ncols_of(c) + NARROW_S for c in range(g.Pc)}
        wmax_e = {c: 3 * full.ncols_of(c) + NARROW_E for c in range(g.Pc)}
        mine = [c for c in range(g.Pc) if c == g.pc or c % g.Pr == g.pr]
        self._bufs = {  # per chain: solved block row pieces, the diagonal factor + records (double-buffered), the interleaved row operand
            "S": dict(buf={c: be.empty(bsmax * wmax[c]) for c in mine},
                      pack=[be.empty(bsmax * bsmax + ((bsmax + 63) // 64) * DREC) for _ in range(2)], aopA=be.empty(bsmax, max(3 * full.nlr, 1))),
            "E": dict(buf={c: be.empty(bsmax * wmax_e[c]) for c in mine},
                      pack=[be.empty(bsmax * bsmax + ((bsmax + 63) // 64) * DREC) for _ in range(2)], aopA=be.empty(bsmax, max(3 * full.nlr, 1))),
        }
        self._aopW = be.empty(bsmax, max(3 * full.nlr, 1))
        # the downdate Sigma_IJ -= sum_k Y_kI^T Y_kJ is ONE product per update (K = m = 2 N: Sll is read and written once instead of once
        # per block row, and the product's prologue / epilogue are amortised): the solved block rows are kept -- the columns of my
        # process column (B operand) and of my row blocks (A operand; the same matrix on a symmetric rank)
        self.symmetric = g.Pr == g.Pc and g.pr == g.pc  # my row blocks ARE my column blocks: the local matrix is symmetric
        self._Yc_buf = be.empty(2 * cap, max(3 * full.nlc, 1))
        self._Yr_buf = self._Yc_buf if self.symmetric else be.empty(2 * cap, max(3 * full.nlr, 1))
        self._accS_buf = be.zeros(NARROW_S, 3 * full.nlc + NARROW_S)
        self._accE = be.zeros(NARROW_E, NARROW_E)

    def _set_slots(self, n):
        """the working set for n >= 1 slots in use: geometry (host + device) and the views of the storage"""
        g, be = self.g, self.be
        if self.geo is not None and self.geo.N == n:
            return
        if self.geo is None:
            self._alloc()
        self.geo = geo = BlockCyclic(n, self.bl, g.Pr, g.Pc, g.pr, g.pc)
        be.set_geometry(geo)
        self.Sll = self._Sll_buf[: 3 * geo.nlr, : 3 * geo.nlc]
        self.M = self._M_buf[: 2 * geo.nlr, : 5 * geo.nlc + NARROW_S]
        self.E = self._E_buf[: 3 * geo.nlr, : 3 * geo.nlc + NARROW_E]
        self._wmax = {c: 5 * geo.ncols_of(c) + NARROW_S for c in range(g.Pc)}
        self._wmax_e = {c: 3 * geo.ncols_of(c) + NARROW_E for c in range(g.Pc)}
        self._Yc = self._Yc_buf[: 2 * n, : 3 * geo.nlc]
        self._Yr = self._Yc if self.symmetric else self._Yr_buf[: 2 * n, : 3 * geo.nlr]
        self._accS = self._accS_buf[:, : 3 * geo.nlc + NARROW_S]

    # ---- VIOFilter::processIMUData (VIOFilter.cpp:120-131)
    # IMU calls are QUEUED (up to 16 - 1 of them) and leave for the device with the next vision call, getter or full queue as
    # one burst: every call keeps its own linearisation, but the local blocks of Sigma are read and written once per burst instead of once
    # per call (eqf_tiled_propagate_burst).  The status a call returns is the reference's control flow (VIOFilter.cpp:120-131, :146-152),
    # mirrored here: it only depends on the stamps.  burst = False: one launch sequence per call, as before.
    burst = True

    def processIMUData(self, stamp, omega, accel):
        if not (self.burst and hasattr(self.be, "propagate_burst")):
            with self.be.main(), self._Phase(self, "propagate"):
                return self.be.propagate(stamp, omega, accel, True, self.Sll)
        if self._mirror_time is None:
            self._mirror_time = self.be.time()
        st = 1 if self._mirror_time < 0 else (2 if not (stamp - self._mirror_time > 0) else 0)  # EQF_SKIPPED_BEFORE_FIRST_IMU / _NONPOSITIVE_DT
        self._mirror_time = float(stamp)
        self._queue.append((float(stamp), np.array(omega, dtype=np.float64), np.array(accel, dtype=np.float64)))
        if len(self._queue) >= self.be.BURST_MAX - 1:
            self._flush()
        return st

    def _flush(self, vision_stamp=None):
        """the queued IMU calls (and the vision call's integration) -> the device; returns the status of the vision call's integration"""
        if not self._queue and vision_stamp is None:
            return 0
        recs, self._queue = self._queue, []
        with self.be.main(), self._Phase(self, "propagate"):
            status = self.be.propagate_burst(recs, vision_stamp, self.Sll)
        if vision_stamp is not None and status[-1] == 0:
            self._mirror_time = float(vision_stamp)
        return status[-1]

    # ---- VIOFilter::processVisionData (VIOFilter.cpp:232-302)
    def processVisionData(self, stamp, ids, bearings):
        with self.be.main():
            return self._process_vision(stamp, ids, bearings)

    def _process_vision(self, stamp, ids, bearings):
        ids = np.asarray(ids, dtype=np.int64)
        y = np.asarray(bearings, dtype=np.float64).reshape(-1, 3)
        if len(ids) != len(y) or (len(ids) > 1 and not np.all(np.diff(ids) > 0)):
            raise ValueError("bearings must come with strictly ascending ids (VIOFilter.cpp:239-240)")
        if self.burst and hasattr(self.be, "propagate_burst"):
            st = self._flush(stamp)  # the queued IMU calls + :233 integrateUpToTime, one pass over the local blocks
        else:
            with self._Phase(self, "propagate"):
                st = self.be.propagate(stamp, None, None, False, self.Sll)  # :233 integrateUpToTime
        if st != 0:
            return st  # :234-236
        with self._Phase(self, "churn"):
            y_slots = self._churn(ids, y)  # :242-249
        if y_slots is None:
            return 4  # EQF_SKIPPED_NO_BEARINGS, :258-259
        self._update(y_slots)
        return 0

    def _churn(self, ids, y):
        """removeOldLandmarks, removeOutliers, addNewLandmarks (VIOFilter.cpp:242-249, :345-443) on slots.  Returns the bearings in SLOT
        order (holes carry a dummy the device ignores), or None when no landmark is left to update with."""
        be = self.be
        have = self.ids if self.ids is not None else np.zeros(0, dtype=np.int64)
        pos = np.searchsorted(ids, have)  # ids ascending: where each state id sits in the measurement, if it does
        pos_c = np.minimum(pos, max(len(ids) - 1, 0))
        seen = (ids[pos_c] == have) if len(ids) else np.zeros(len(have), dtype=bool)  # removeOldLandmarks :393-419
        new_k = np.nonzero(~np.isin(ids, have))[0]  # measurement entries without a landmark, ascending ids (:211-230 puts them last)
        keep = seen.copy()
        depth = be.initial_scene_depth()
        thr = be.outlier_threshold()
        gate = thr < 2.0 and seen.any()  # (no chord of unit vectors is longer than 2: such a threshold switches the gate off, no readback)
        if gate or (len(new_k) and seen.any()):
            p = np.asarray(be.state_estimate()["p"], dtype=np.float64).reshape(-1, 3)[self.slot_of]  # reference order
            if gate:  # removeOutliers :429-443: chord between the measured and the expected bearing
                yhat = p / np.linalg.norm(p, axis=1, keepdims=True)
                chord = np.linalg.norm(y[pos_c] - yhat, axis=1)
                keep &= ~(chord > thr)
            if len(new_k) and keep.any():  # median scene depth of what is left, :353-366 (nth_element at size / 2)
                d2 = np.sort(np.sum(p[keep] * p[keep], axis=1))
                depth = float(np.sqrt(d2[len(d2) // 2]))
        n_old = int((~seen).sum())
        n_out = int((seen & ~keep).sum())
        remove_slots = self.slot_of[~keep]
        taken = self.taken.copy()
        taken[remove_slots] = False
        free = np.nonzero(~taken)[0]
        if len(new_k) > len(free):
            raise RuntimeError(f"{int(taken.sum()) + len(new_k)} landmarks in view, the partitioned filter was created for {self.cap}")
        add_slots = free[: len(new_k)]  # lowest free slots first: holes are refilled before the partition grows
        taken[add_slots] = True
        top = np.nonzero(taken)[0]
        nslots = max(int(top[-1]) + 1 if len(top) else 0, 1)
        if self.ids is None and len(new_k) == 0:
            return None  # nothing yet, nothing to add: no storage either
        if len(remove_slots) or len(add_slots):
            self._set_slots(max(self.nslots, nslots))
            be.edit_landmarks(remove_slots, add_slots, y[new_k], depth, nslots, self.Sll)
            self._set_slots(nslots)
            self.ids = np.concatenate([have[keep], ids[new_k]])
            self.slot_of = np.concatenate([self.slot_of[keep], add_slots]).astype(np.int64)
            self.taken, self.nslots = taken, nslots
            self.churn_stats["removed_old"] += n_old
            self.churn_stats["removed_outliers"] += n_out
            self.churn_stats["added"] += len(new_k)
        if len(self.ids) == 0:
            return None
        # the measurement in slot order: landmarks that stayed, then the new ones (:211-230 matchMeasurementsToState)
        y_slots = np.zeros((self.nslots, 3))
        y_slots[:, 2] = 1.0
        y_slots[self.slot_of] = np.concatenate([y[pos_c[keep]], y[new_k]]) if len(self.ids) else y[:0]
        return y_slots

    def initialise_from(self, st):
        """Restart from a single-GPU snapshot (FilterBatch.dump_state(); every rank holds the dense Sigma once, here)."""
        N = len(st["ids"])
        with self.be.main():
            self._initialise_from(st, N)

    def _initialise_from(self, st, N):
        if N > self.cap:
            raise RuntimeError(f"snapshot with {N} landmarks, the partitioned filter was created for {self.cap}")
        self._set_slots(max(N, 1))
        S = torch.as_tensor(np.asarray(st["sigma"]), dtype=torch.float64)
        rows = torch.as_tensor(np.repeat(3 * self.geo.rowMap.astype(np.int64), 3) + np.tile(np.arange(3), self.geo.nlr) + 11)
        cols = torch.as_tensor(np.repeat(3 * self.geo.colMap.astype(np.int64), 3) + np.tile(np.arange(3), self.geo.nlc) + 11)
        if N and self.geo.nlr and self.geo.nlc:
            self.Sll.copy_(S[rows][:, cols].to(self.Sll.device))
        self.be.set_state(st)
        self.ids = np.asarray(st["ids"], dtype=np.int64)
        self._queue, self._mirror_time = [], None
        self.slot_of = np.arange(N, dtype=np.int64)  # the snapshot's order is the reference's: slot i = landmark i
        self.taken = np.zeros(self.cap, dtype=bool)
        self.taken[:N] = True
        self.nslots = N

    # ---- the update
    def _update(self, y):
        be, geo = self.be, self.geo
        with self._Phase(self, "prep"):
            be.update_prep(y, self.Sll, self.M, self.E, self.G11)  # E and M are formed from the PRE-update Sigma (VIOFilter.cpp:285 before :297)
        nA = 2 * geo.nlc
        self._accS.zero_()
        self._accE.zero_()

        def hook_s(k, bk, Bop, off, contributions):
            # Bop[:, off:] = [Y_k (3 nlc) | Yn_k (18)] of my process column; the rank's share of the downdate and of the reductions
            Yw = Bop[:, off: off + 3 * geo.nlc]
            Yn = Bop[:, off + 3 * geo.nlc: off + 3 * geo.nlc + NARROW_S]
            r0 = 2 * k * geo.bl
            self._Yc[r0: r0 + bk].copy_(Yw)
            if not self.symmetric:
                YI = self._rows_operand(contributions, 3, lambda c, wc: (wc - 3 * geo.ncols_of(c) - NARROW_S, 0), bk, self._aopW, all_blocks=True)
                self._Yr[r0: r0 + bk].copy_(YI)
            be.gemm_tn(self._accS, Yn, Bop[:, off:], 1.0)               # [Sigma_b's downdate ; gamma_L ; .. | Gnn] += Yn_k^T [Y_k | Yn_k]

        def hook_e(k, bk, Bop, off, contributions):
            En = Bop[:, off: off + NARROW_E]
            be.gemm_tn(self._accE, En, En, 1.0)

        # the E-chain (bundleLift's weights) needs nothing of the S-chain: it runs on its own stream next to it.  It is bound by its serial
        # diagonal blocks, the S-chain and the downdate by the matrix cores -- side by side they take little more than the longer one
        prepared = be.record()
        e_done = None
        if self.overlap_chains:
            # The two chains are enqueued ALTERNATELY, block row by block row: a chain is a few hundred launches, and enqueued one chain
            # after the other the second stream sat idle until the host was through with the first -- 32 of an update's 84 ms under the
            # profiler (scripts/queue_summary.py), the update was bound by the HOST's launch rate, not by the GPU.
            stepsE = self._chain_steps(self.E, 3, 3 * geo.nlc, hook_e, self._wmax_e, self.gE, self._bufs["E"], be.aux_side)
            stepsS = self._chain_steps(self.M, 2, nA, hook_s, self._wmax, self.g, self._bufs["S"], be.side)
            phE, phS = self._Phase(self, "chain_E"), self._Phase(self, "chain_S")
            with be.aux():
                be.wait(prepared)
                phE.__enter__()
            phS.__enter__()
            doneE = doneS = False
            while not (doneE and doneS):
                if not doneE:
                    with be.aux():
                        doneE = next(stepsE, None) is None
                if not doneS:
                    doneS = next(stepsS, None) is None
            with be.aux():
                phE.__exit__(None, None, None)
                e_done = be.record()
            phS.__exit__(None, None, None)
        else:
            with self._Phase(self, "chain_S"):
                self._chain(self.M, 2, nA, hook_s, self._wmax, self.g, self._bufs["S"], be.side)
        with self._Phase(self, "downdate"):
            # Sigma_IJ -= Y_I^T Y_J (VIOFilter.cpp:297), one product; on a symmetric rank only the blocks on and above the block
            # diagonal are computed and the rest is mirrored
            if geo.nlr and geo.nlc:
                w3 = 3 * geo.bl
                if self.symmetric:
                    be.gemm_tn(self.Sll, self._Yr, self._Yc, -1.0, mask=(w3, w3, 0, 1, 0, 0, 1, 0))
                    be.mirror_lower(self.Sll, w3)
                else:
                    be.gemm_tn(self.Sll, self._Yr, self._Yc, -1.0)
        if self.overlap_chains:
            be.wait(e_done)
        else:
            with self._Phase(self, "chain_E"):
                self._chain(self.E, 3, 3 * geo.nlc, hook_e, self._wmax_e, self.g, self._bufs["E"], be.side)
        with self._Phase(self, "finish"):
            # gamma_L and the base panel's downdate live with the process COLUMNS: gather them along the process row, global landmark order
            acc = self._gather_columns(self._accS[:, : 3 * geo.nlc])
            Gnn = self._accS[:, 3 * geo.nlc:].contiguous()
            G11 = (self.G11 + self._accE).contiguous()
            be.update_finish(acc, Gnn, G11)
        self._frames_since_check = getattr(self, "_frames_since_check", 0) + 1
        if self.check_every and self._frames_since_check >= self.check_every:
            self.check()

    check_every = 1  # frames between two looks at the factorisations' pivot flag (a look synchronises the stream)

    def check(self):
        """Raises if a pivot of S or Sigma_e was not positive since the last look (synchronises)."""
        self._frames_since_check = 0
        if self.be.factor_info():
            raise ArithmeticError("a pivot of S or Sigma_e was not positive (distributed factorisation)")

    def _gather_columns(self, mine):
        g, geo = self.g, self.geo
        rows = mine.shape[0]
        wmax = 3 * max(geo.ncols_of(c) for c in range(g.Pc))
        pad = self.be.zeros(rows, wmax)
        pad[:, : mine.shape[1]] = mine
        parts = g.allgather_row(pad)
        out = self.be.empty(rows, 3 * geo.N)
        for c, part in enumerate(parts):
            o = 0
            for b in range(c, geo.nb, g.Pc):
                w = 3 * geo.block_size(b)
                out[:, 3 * b * geo.bl: 3 * b * geo.bl + w] = part[:, o: o + w]
                o += w
        return out

    def _rows_operand(self, contributions, unit, part_of, bk, buf, all_blocks, k=None):
        """The A operand of the products of block row k: for each of MY local row blocks (all of them, or the trailing ones i > k) the
        (bk x unit * size) block of the solved block row -- found in the piece of the process column c = i mod Pc, which the rank (pr, c)
        re-broadcast along the process row.  contributions: {c: (piece (bk x w_c), jl0_c)}; part_of(c, w_c) -> (column offset of the part
        inside the piece, 1 if the part starts at local block jl0_c else 0)."""
        g, geo = self.g, self.geo
        q = g.Pc // g.Pr
        bsF = unit * geo.bl
        il0 = 0 if all_blocks else BlockCyclic.blocks_upto(k, g.pr, g.Pr)
        ncol = unit * geo.nlr - il0 * bsF
        if ncol <= 0:
            return None
        if q == 1:
            # one contributor, c = pr: its local column blocks ARE my local row blocks, in order -> a view, no copy
            piece, jl0 = contributions[g.pr]
            off, trailing = part_of(g.pr, piece.shape[1])
            start = off + ((il0 - jl0) * bsF if trailing else il0 * bsF)
            return piece[:, start: start + ncol]
        out = buf[:bk, : unit * geo.nlr]
        for s in range(q):
            c = g.pr + g.Pr * s
            piece, jl0 = contributions[c]
            off, trailing = part_of(c, piece.shape[1])
            # my local row block ilb = s + t q  <->  local column block t of process column c
            for ilb in range(s, len(geo.row_blocks), q):
                if ilb < il0:
                    continue
                t = (ilb - s) // q
                w = unit * geo.block_size(geo.row_blocks[ilb])
                src = off + ((t - jl0) if trailing else t) * bsF
                out[:, ilb * bsF: ilb * bsF + w] = piece[:, src: src + w]
        return out[:, il0 * bsF:]

    def _chain(self, X, unit, nA, hook, wmax, g, bufs, side):
        """all block rows of _chain_steps, one after the other"""
        for _ in self._chain_steps(X, unit, nA, hook, wmax, g, bufs, side):
            pass

    def _chain_steps(self, X, unit, nA, hook, wmax, g, bufs, side):
        """(a generator: one block row per step, so that the caller can feed two factorisations to their streams alternately)
        Blocked right-looking Cholesky by block ROWS of the SPD matrix in X[:, :nA] (upper blocks, block size unit * bl, block-cyclic
        over the grid) with the right-hand sides X[:, nA:]; X is consumed.  hook(k, bk, Bop, off, contributions) runs on every rank once
        block row k is solved: Bop[:, off:] holds the right-hand-side part of my process column.
        Look-ahead: the diagonal block is the serial part (one workgroup, eqf_tile_potrf).  As soon as block row k is solved, the owner of
        block (k+1, k+1) applies row k to a COPY of that block and factors the copy on a second stream, in the shadow of the trailing
        update of step k; step k+1 then starts from the finished factor."""
        geo, be = self.geo, self.be
        bsF = unit * geo.bl
        W = X.shape[1]
        ahead = None  # event: the look-ahead factor of the current block is in bufs["pack"][k & 1]
        for k in range(geo.nb):
            prk, pck = k % g.Pr, k % g.Pc
            bk = unit * geo.block_size(k)
            klr, klc = k // g.Pr, k // g.Pc
            jl0 = BlockCyclic.blocks_upto(k, g.pc, g.Pc)
            c0 = min(jl0 * bsF, nA)
            width = W - c0
            Bop = bufs["buf"][g.pc][: bk * width].view(bk, width)
            if g.pr == prk:
                # 1. the diagonal block, L_kk and its records along the process row
                nrec = ((bk + 63) // 64) * DREC
                pack = bufs["pack"][k & 1][: bk * bk + nrec]
                Lkk, drec = pack[: bk * bk].view(bk, bk), pack[bk * bk:]
                if g.pc == pck:
                    if ahead is not None:
                        be.wait(ahead)
                        ahead = None
                    else:
                        Lkk.copy_(X[klr * bsF: klr * bsF + bk, klc * bsF: klc * bsF + bk])
                        be.potrf(Lkk, drec)
                g.bcast_row(pack, pck)
                # 2. my piece of block row k
                R = X[klr * bsF: klr * bsF + bk, c0:]
                be.trsm_left(Lkk, drec, R)
                Bop.copy_(R)
            # 3. down the process column
            g.bcast_col(Bop, prk)
            # 4. along the process row, from the ranks whose column blocks are this process row's row blocks
            contributions = {}
            for c in range(g.pr, g.Pc, g.Pr):
                jl0c = BlockCyclic.blocks_upto(k, c, g.Pc)
                wc = wmax[c] - min(jl0c * bsF, unit * geo.ncols_of(c))
                piece = Bop if c == g.pc else bufs["buf"][c][: bk * wc].view(bk, wc)
                g.bcast_row(piece, c)
                contributions[c] = (piece, jl0c)
            # 5. trailing updates of what this rank owns: rows of blocks i > k, columns from block jl0 on
            il0 = BlockCyclic.blocks_upto(k, g.pr, g.Pr)
            if il0 * bsF < X.shape[0]:
                Ua = self._rows_operand(contributions, unit, lambda c, wc: (0, 1), bk, bufs["aopA"], all_blocks=False, k=k)
                if self.lookahead and k + 1 < geo.nb and g.pr == (k + 1) % g.Pr and g.pc == (k + 1) % g.Pc:
                    # look-ahead: block (k+1, k+1) is the first trailing block of my rows and of my columns
                    b1 = unit * geo.block_size(k + 1)
                    nrec1 = ((b1 + 63) // 64) * DREC
                    pack1 = bufs["pack"][(k + 1) & 1][: b1 * b1 + nrec1]
                    L1, drec1 = pack1[: b1 * b1].view(b1, b1), pack1[b1 * b1:]
                    L1.copy_(X[il0 * bsF: il0 * bsF + b1, c0: c0 + b1])
                    ready = be.record()
                    with side():
                        be.wait(ready)
                        be.gemm_tn(L1, Ua[:, :b1], Bop[:, :b1], -1.0)
                        be.potrf(L1, drec1)
                        ahead = be.record()
                Ct = X[il0 * bsF:, c0:]
                if nA - c0 > 0:
                    be.gemm_tn(Ct[:, : nA - c0], Ua, Bop[:, : nA - c0], -1.0, mask=(bsF, bsF, il0, g.Pr, g.pr, jl0, g.Pc, g.pc))
                be.gemm_tn(Ct[:, nA - c0:], Ua, Bop[:, nA - c0:], -1.0)
            hook(k, bk, Bop, nA - c0, contributions)
            yield k

    # ---- getters
    def getTime(self):
        self._flush()
        return self.be.time()

    def _coords(self, unit, base):
        """coordinates of the landmarks, reference order, in a slot-ordered vector with `unit` entries per slot after `base` leading ones"""
        return (base + unit * np.repeat(self.slot_of, unit) + np.tile(np.arange(unit), len(self.slot_of))).astype(np.int64)

    def stateEstimate(self):
        """VIOFilter::stateEstimate (:304): landmarks in the reference's order"""
        self._flush()
        e = dict(self.be.state_estimate())
        e["p"] = np.asarray(e["p"]).reshape(-1, 3)[self.slot_of]
        e["ids"] = self.ids.copy() if self.ids is not None else np.zeros(0, dtype=np.int64)
        return e

    def bias(self):
        self._flush()
        return self.be.bias()

    def lastUpdate(self):
        """delta (2 N), gamma (11 + 3 N), Gamma (9 + 3 N) of the last update, landmarks in the reference's order"""
        lu = self.be.last_update()
        out = {"delta": np.asarray(lu["delta"])[self._coords(2, 0)]}
        out["gamma"] = np.concatenate([np.asarray(lu["gamma"])[:11], np.asarray(lu["gamma"])[self._coords(3, 11)]])
        G = lu.get("Gamma")
        out["Gamma"] = None if G is None else np.concatenate([np.asarray(G)[:9], np.asarray(G)[self._coords(3, 9)]])
        return out

    def stateCovariance(self):
        """Dense Sigma (reference index map and landmark order) gathered to every rank -- tests and snapshots
        (VIOFilter::stateCovariance, :306-309)."""
        self._flush()
        with self.be.main():
            S = self._state_covariance()
        idx = np.concatenate([np.arange(11), self._coords(3, 11)])
        return S[np.ix_(idx, idx)]

    def slotCovariance(self):
        """Dense Sigma over ALL slots in use, holes included (slot order) -- tests of the hole invariants."""
        self._flush()
        with self.be.main():
            return self._state_covariance()

    def _state_covariance(self):
        g, geo = self.g, self.geo
        N = geo.N
        n = 11 + 3 * N
        S = np.zeros((n, n))
        base = np.asarray(self.be.base_rows())[:, :n]
        S[:, :11] = base.T  # (only the base ROWS are kept: the columns are their transpose)
        S[:11, :] = base
        rmax = 3 * max(len(BlockCyclic(N, self.bl, g.Pr, g.Pc, r, 0).rowMap) for r in range(g.Pr))
        cmax = 3 * max(geo.ncols_of(c) for c in range(g.Pc))
        pad = self.be.zeros(rmax, cmax)
        pad[: 3 * geo.nlr, : 3 * geo.nlc] = self.Sll
        for rank, part in enumerate(g.allgather_all(pad)):
            r, c = divmod(rank, g.Pc)
            og = BlockCyclic(N, self.bl, g.Pr, g.Pc, r, c)
            rows = (np.repeat(3 * og.rowMap.astype(np.int64), 3) + np.tile(np.arange(3), og.nlr)) + 11
            cols = (np.repeat(3 * og.colMap.astype(np.int64), 3) + np.tile(np.arange(3), og.nlc)) + 11
            S[np.ix_(rows, cols)] = part[: 3 * og.nlr, : 3 * og.nlc].cpu().numpy()
        return S
