"""ctypes binding of the PER-RANK device side of the 2-D block-partitioned filter (include/eqf_vio_amd.h: eqf_tiled_* -- replicated O(N)
state, base panel, local blocks -- and eqf_tile_* -- the dense tile kernels of the distributed factorisations) on torch CUDA tensors'
device pointers.  The filter's host loop is C++ (csrc/eqf_tiledf.hip, eqf_tf_*; glue: eqf_vio_amd/tiled.py); this class is what the
kernel-level tests, the microbenchmarks and the Python reference of the schedule (tests/tiled_reference.py) drive the kernels with.
Fails loudly without the library or a GPU: there is no CPU fallback."""
import ctypes

import numpy as np
import torch


class HipBackend:
    """The per-rank device side through the C ABI: an eqf_tiled handle (replicated state + base panel) and the dense tile kernels, on
    torch CUDA tensors of `device_index` and torch's current stream."""

    DREC = 64 * 64 + 4 * 16 * 16  # doubles per 64-wide block column of a diagonal-factor record

    def __init__(self, settings, capacity, device_index=0, reserve_cus=None, cu_range=None):
        """cu_range = (first_cu, num_cus): confine EVERY stream of this rank to that slice of the GPU (experiments with several ranks on one
        device, scripts/tiled_cumask.py: main streams on [first + reserve, first + num), look-ahead streams on [first, first + reserve))."""
        from . import binding

        self.b = binding
        self.lib = binding.lib()  # raises when libeqf_vio_amd.so is missing: there is no CPU fallback
        if isinstance(settings, dict):
            settings = binding.settings_from_dict(settings)
        self.settings = settings
        self.dev = int(device_index)
        self.device = torch.device("cuda", self.dev)
        self.cap = int(capacity)
        self._h = ctypes.c_void_p()
        binding._check(self.lib.eqf_tiled_create(ctypes.byref(settings), self.cap, self.dev, ctypes.byref(self._h)), "eqf_tiled_create")
        self._stream = None
        self._info = torch.zeros(1, dtype=torch.int32, device=self.device)
        # Two streams with disjoint CU sets: `reserve` CUs for the look-ahead factorisation of the next diagonal block (side()), all the
        # others for everything else (main()).  EQF_TILED_RESERVE_CUS=0: no reservation -- main() is torch's current stream and the
        # look-ahead only runs when the trailing update happens to leave room.
        import os

        self.reserve = int(os.environ.get("EQF_TILED_RESERVE_CUS", "24")) if reserve_cus is None else int(reserve_cus)
        # The streams are made on first use (round 6): a CU-masked stream is a hardware queue of its own, the GPU slows down once a process has
        # used more than a handful of them (scripts/handle_age_probe.py), and a backend that only answers for a TiledFilter's handle (adopt())
        # never runs anything on its own streams.
        self._raw = []
        self._cu_range = cu_range
        self._main = self._side = self._aux = self._aux_side = None
        self._sync_stream()

    def _masked(self, i):
        """stream i of (main, side, aux, aux_side): main / aux on all CUs but the reserved ones, side / aux_side on the reserved ones"""
        p = ctypes.c_void_p()
        if self._cu_range is None:
            first, count, comp = 0, self.reserve, 1 if i % 2 == 0 else 0
        else:
            cr = self._cu_range
            first, count, comp = (cr[0] + self.reserve, cr[1] - self.reserve, 0) if i % 2 == 0 else (cr[0], self.reserve, 0)
        self.b._check(self.lib.eqf_stream_create_masked(self.dev, first, count, comp, ctypes.byref(p)), "eqf_stream_create_masked")
        self._raw.append(p)
        return torch.cuda.ExternalStream(p.value, device=self.device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            if not getattr(self, "_adopted", False):  # (an adopted handle is its TiledFilter's)
                self.lib.eqf_tiled_destroy(self._h)
            self._h = ctypes.c_void_p()
            for p in getattr(self, "_raw", []):
                self.lib.eqf_stream_destroy(self.dev, p)
            self._raw = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def adopt(self, handle):
        """From now on answer for the eqf_tiled handle of a TiledFilter (eqf_tf_tiled_handle): the getters of the replicated state then show
        the FILTER's state, in slot order.  The handle stays the filter's (its stream is not touched; do not call the state-changing entry
        points through an adopted handle)."""
        if self._h.value:
            self.lib.eqf_tiled_destroy(self._h)
        self._h = ctypes.c_void_p(handle)
        self._adopted = True

    def release(self):
        if getattr(self, "_adopted", False):
            self._h = ctypes.c_void_p()
            self._adopted = False

    # ---- plumbing
    def _sync_stream(self):
        s = torch.cuda.current_stream(self.dev).cuda_stream
        if s != self._stream:
            self.b._check(self.lib.eqf_tiled_set_stream(self._h, ctypes.c_void_p(s)), "eqf_tiled_set_stream")
            self._stream = s
        return ctypes.c_void_p(s)

    def _cur(self):
        """torch's current stream, for the dense tile kernels (they take the stream as an argument; the handle's own stream -- the
        state kernels -- only follows torch's stream outside side())"""
        return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    # ---- a second stream for the look-ahead factorisation of the next diagonal block (TiledFilter._chain)
    def main(self):
        """context: the stream every call of the filter runs on (all CUs but the reserved ones)"""
        import contextlib

        if self.reserve <= 0:
            return contextlib.nullcontext()
        if self._main is None:
            self._main = self._masked(0)
        return torch.cuda.stream(self._main)

    def side(self):
        if self._side is None:
            self._side = self._masked(1) if self.reserve > 0 else torch.cuda.Stream(device=self.device)
        return torch.cuda.stream(self._side)

    def aux(self):
        """context: a second 'main' stream (same CU set), for the factorisation that runs next to the other one"""
        if self._aux is None:
            self._aux = self._masked(2) if self.reserve > 0 else torch.cuda.Stream(device=self.device)
        return torch.cuda.stream(self._aux)

    def aux_side(self):
        if self._aux_side is None:
            self._aux_side = self._masked(3) if self.reserve > 0 else torch.cuda.Stream(device=self.device)
        return torch.cuda.stream(self._aux_side)

    def record(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        return ev

    def wait(self, ev):
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p()

    @staticmethod
    def _dp(a):
        return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64, device=self.device)

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float64, device=self.device)

    # ---- replicated state + local blocks
    def set_geometry(self, geo):
        rm, cm = np.ascontiguousarray(geo.rowMap, dtype=np.int32), np.ascontiguousarray(geo.colMap, dtype=np.int32)
        ip = ctypes.POINTER(ctypes.c_int)
        self.b._check(self.lib.eqf_tiled_set_geometry(self._h, len(rm), rm.ctypes.data_as(ip), len(cm), cm.ctypes.data_as(ip)), "eqf_tiled_set_geometry")

    def propagate(self, stamp, omega, accel, is_imu, Sll):
        self._sync_stream()
        w = np.ascontiguousarray(omega if omega is not None else np.zeros(3), dtype=np.float64)
        a = np.ascontiguousarray(accel if accel is not None else np.zeros(3), dtype=np.float64)
        ld = Sll.stride(0) if Sll is not None else 0
        return self.b._check(self.lib.eqf_tiled_propagate(self._h, float(stamp), self._dp(w), self._dp(a), int(bool(is_imu)), self._p(Sll), ld),
                             "eqf_tiled_propagate")

    def add_landmarks(self, bearings, Sll):
        self._sync_stream()
        y = np.ascontiguousarray(bearings, dtype=np.float64).reshape(-1, 3)
        self.b._check(self.lib.eqf_tiled_add_landmarks(self._h, len(y), self._dp(y), self._p(Sll), Sll.stride(0) if Sll is not None else 0),
                      "eqf_tiled_add_landmarks")

    BURST_MAX = 16

    def propagate_burst(self, records, vision_stamp, Sll):
        """records: [(stamp, omega, accel), ...] IMU calls; vision_stamp: the stamp of the vision call whose integrateUpToTime closes the burst
        (or None).  One pass over Sll for all of them (eqf_tiled_propagate_burst).  Returns the status of every call."""
        self._sync_stream()
        K = len(records) + (1 if vision_stamp is not None else 0)
        assert 1 <= K <= self.BURST_MAX
        stamps = np.zeros(K)
        w, a = np.zeros((K, 3)), np.zeros((K, 3))
        for k, (st, om, ac) in enumerate(records):
            stamps[k], w[k], a[k] = st, om, ac
        if vision_stamp is not None:
            stamps[K - 1] = vision_stamp
        status = np.zeros(K, dtype=np.int32)
        ld = Sll.stride(0) if Sll is not None else 0
        self.b._check(self.lib.eqf_tiled_propagate_burst(self._h, K, self._dp(stamps), self._dp(w), self._dp(a), int(vision_stamp is not None), self._p(Sll),
                                                         ld, status.ctypes.data_as(ctypes.POINTER(ctypes.c_int))), "eqf_tiled_propagate_burst")
        return [int(x) for x in status]

    def edit_landmarks(self, remove_slots, add_slots, add_bearings, depth, new_num_slots, Sll):
        """removeLandmarkAtIndex for remove_slots, then addNewLandmarks into add_slots (include/eqf_vio_amd.h: eqf_tiled_edit_landmarks);
        the geometry in force must cover max(old, new) slots."""
        self._sync_stream()
        rs = np.ascontiguousarray(remove_slots, dtype=np.int32).reshape(-1)
        ads = np.ascontiguousarray(add_slots, dtype=np.int32).reshape(-1)
        y = np.ascontiguousarray(add_bearings, dtype=np.float64).reshape(-1, 3)
        assert len(y) == len(ads)
        ip = ctypes.POINTER(ctypes.c_int)
        self.b._check(self.lib.eqf_tiled_edit_landmarks(self._h, len(rs), rs.ctypes.data_as(ip), len(ads), ads.ctypes.data_as(ip), self._dp(y), float(depth),
                                                        int(new_num_slots), self._p(Sll), Sll.stride(0) if Sll is not None else 0),
                      "eqf_tiled_edit_landmarks")

    def initial_scene_depth(self):
        return float(self.settings.initialSceneDepth)

    def update_prep(self, bearings, Sll, M, E, G11):
        self._sync_stream()
        y = np.ascontiguousarray(bearings, dtype=np.float64).reshape(-1, 3)
        self.b._check(self.lib.eqf_tiled_update_prep(self._h, self._dp(y), self._p(Sll), Sll.stride(0), self._p(M), M.stride(0), self._p(E),
                                                     E.stride(0), self._p(G11)), "eqf_tiled_update_prep")

    def update_finish(self, acc, Gnn, G11):
        self._sync_stream()
        assert acc.stride(1) == 1 and Gnn.is_contiguous() and G11.is_contiguous()
        self.b._check(self.lib.eqf_tiled_update_finish(self._h, self._p(acc), acc.stride(0), self._p(Gnn), self._p(G11)), "eqf_tiled_update_finish")

    # ---- dense tile kernels (csrc/eqf_tile.hpp)
    def potrf(self, Akk, drec=None):
        """In place: lower triangle of the (n x n) view Akk <- L.  Returns (fills) the diagonal-factor records trsm() multiplies with."""
        n = Akk.shape[0]
        if drec is None:
            drec = torch.empty(((n + 63) // 64) * self.DREC, dtype=torch.float64, device=self.device)
        self.b._check(self.lib.eqf_tile_potrf(self.dev, self._cur(), self._p(Akk), Akk.stride(0), n, self._p(drec), self._p(self._info)),
                      "eqf_tile_potrf")
        return drec

    TRSM_SPLIT = 6  # block rows of 64 from which a solve is split in two (see trsm_left)

    def trsm_left(self, L, drec, Bm):
        """In place: Bm (n x m view) <- L^-1 Bm.
        eqf_tile_trsm is one workgroup per 64-column strip, and a strip is a CHAIN of nb (nb + 1) / 2 block products (nb = n / 64): its time
        is that chain's latency whatever the width.  So from TRSM_SPLIT block rows on the solve is split once, [L11 0; L21 L22]:
        X1 = L11^-1 B1 (a chain of a quarter of the products), B2 -= L21 X1 as ONE product on the whole chip (eqf_tile_gemm_tn, with L21
        transposed into a scratch operand), X2 = L22^-1 B2 -- the records of L22's block columns are the tail of L's."""
        n = L.shape[0]
        nb = (n + 63) // 64
        if nb < self.TRSM_SPLIT or Bm.shape[1] < 256:
            self._trsm_launch(L, drec, Bm)
            return
        h = 64 * (nb // 2)
        self._trsm_launch(L[:h, :h], drec, Bm[:h])
        key = (n - h, h, self._cur().value)  # (one scratch operand per shape AND stream: the two chains solve side by side)
        if not hasattr(self, "_l21t"):
            self._l21t = {}
        if key not in self._l21t:
            self._l21t[key] = self.empty(h, n - h)
        l21t = self._l21t[key]
        l21t.copy_(L[h:, :h].t())
        self.gemm_tn(Bm[h:], l21t, Bm[:h], -1.0)
        self._trsm_launch(L[h:, h:], drec[(h // 64) * self.DREC:], Bm[h:])

    def _trsm_launch(self, L, drec, Bm):
        self.b._check(self.lib.eqf_tile_trsm(self.dev, self._cur(), self._p(L), L.stride(0), L.shape[0], self._p(drec), self._p(Bm),
                                             Bm.stride(0), Bm.shape[1], 0), "eqf_tile_trsm")

    def gemm_tn(self, Cm, A, B, alpha, mask=None):
        """Cm (m x n view) += alpha A^T B; A (k x m), B (k x n) views with unit column stride.  mask = (rb, cb, rblk0, Pr, pr, cblk0, Pc,
        pc): skip tiles entirely below the block diagonal of a block-cyclic local matrix."""
        m, n = Cm.shape
        k = A.shape[0]
        if m == 0 or n == 0 or k == 0:
            return
        assert A.shape[1] == m and B.shape == (k, n) and Cm.stride(1) == 1 and A.stride(1) == 1 and B.stride(1) == 1
        mk = mask if mask is not None else (0, 0, 0, 1, 0, 0, 1, 0)
        self.b._check(self.lib.eqf_tile_gemm_tn(self.dev, self._cur(), self._p(Cm), Cm.stride(0), m, n, self._p(A), A.stride(0), self._p(B),
                                                B.stride(0), k, float(alpha), *[int(x) for x in mk]), "eqf_tile_gemm_tn")

    def downdate_i8(self, Cm, A, B, slices, mask_rb=0):
        """Cm (m x n view) -= A^T B on the integer matrix pipe from `slices` 7-bit slices of the operands' columns, exact accumulation
        (eqf_tile_downdate_i8, include/eqf_vio_amd_debug.h); A is B (the same view): one split."""
        import torch

        m, n = Cm.shape
        k = A.shape[0]
        same = A.data_ptr() == B.data_ptr() and A.shape == B.shape and A.stride(0) == B.stride(0)
        need = int(self.lib.eqf_tile_i8_workspace_bytes(m, n, k, int(slices), int(same)))
        assert need > 0
        ws = torch.empty(need, dtype=torch.uint8, device=Cm.device)
        self.b._check(self.lib.eqf_tile_downdate_i8(self.dev, self._cur(), self._p(Cm), Cm.stride(0), m, n, self._p(A), A.stride(0), self._p(B),
                                                    B.stride(0), k, int(slices), int(mask_rb), ws.data_ptr(), need), "eqf_tile_downdate_i8")
        torch.cuda.current_stream(Cm.device).synchronize()  # (the workspace is a temporary of this call)

    def gemm_tn_i8(self, Cm, A, B, slices, mask=None, mask_cols=0):
        """Cm (m x n view) -= A^T B on the integer matrix pipe behind gemm_tn's block mask; the mask covers the first mask_cols columns
        (eqf_tile_gemm_tn_i8, include/eqf_vio_amd_debug.h).  A a column range of B at a multiple of 32: one split."""
        import torch

        m, n = Cm.shape
        k = A.shape[0]
        need = int(self.lib.eqf_tile_i8_workspace_bytes(m, n, k, int(slices), 0))
        assert need > 0
        ws = torch.empty(need, dtype=torch.uint8, device=Cm.device)
        mk = mask if mask is not None else (0, 0, 0, 1, 0, 0, 1, 0)
        self.b._check(self.lib.eqf_tile_gemm_tn_i8(self.dev, self._cur(), self._p(Cm), Cm.stride(0), m, n, self._p(A), A.stride(0), self._p(B),
                                                   B.stride(0), k, int(slices), *[int(x) for x in mk], int(mask_cols), ws.data_ptr(), need),
                      "eqf_tile_gemm_tn_i8")
        torch.cuda.current_stream(Cm.device).synchronize()  # (the workspace is a temporary of this call)

    def mirror_lower(self, Cm, rb):
        """Cm (n x n view): every element below the block diagonal (blocks of rb) <- its mirror image."""
        self.b._check(self.lib.eqf_tile_mirror(self.dev, self._cur(), self._p(Cm), Cm.stride(0), Cm.shape[0], int(rb)), "eqf_tile_mirror")

    def factor_info(self):
        """non-zero if a pivot of any diagonal block since the last call was not positive (synchronises)"""
        v = int(self._info.item())
        self._info.zero_()
        return v

    # ---- getters (synchronise)
    def num_landmarks(self):
        return self.lib.eqf_tiled_num_landmarks(self._h)

    def time(self):
        t = ctypes.c_double()
        self.lib.eqf_tiled_get_time(self._h, ctypes.byref(t))
        return t.value

    def device_error(self):
        return self.lib.eqf_tiled_device_error(self._h)

    def outlier_threshold(self):
        return float(self.settings.outlierThreshold)

    def state_estimate(self):
        N = self.num_landmarks()
        q, x, v, p = np.zeros(4), np.zeros(3), np.zeros(3), np.zeros((max(N, 1), 3))
        self.b._check(self.lib.eqf_tiled_get_state_estimate(self._h, self._dp(q), self._dp(x), self._dp(v), self._dp(p)), "eqf_tiled_get_state_estimate")
        return {"q": q, "x": x, "v": v, "p": p[:N]}

    def origin(self):
        N = self.num_landmarks()
        q, x, v, p = np.zeros(4), np.zeros(3), np.zeros(3), np.zeros((max(N, 1), 3))
        self.b._check(self.lib.eqf_tiled_get_origin(self._h, self._dp(q), self._dp(x), self._dp(v), self._dp(p)), "eqf_tiled_get_origin")
        return {"q": q, "x": x, "v": v, "p": p[:N]}

    def group(self):
        N = self.num_landmarks()
        Aq, Ax, w, Qq, Qa = np.zeros(4), np.zeros(3), np.zeros(3), np.zeros((max(N, 1), 4)), np.zeros(max(N, 1))
        self.b._check(self.lib.eqf_tiled_get_group(self._h, self._dp(Aq), self._dp(Ax), self._dp(w), self._dp(Qq), self._dp(Qa)), "eqf_tiled_get_group")
        return {"Aq": Aq, "Ax": Ax, "w": w, "Qq": Qq[:N], "Qa": Qa[:N]}

    def bias(self):
        b6 = np.zeros(6)
        self.b._check(self.lib.eqf_tiled_get_bias(self._h, self._dp(b6)), "eqf_tiled_get_bias")
        return b6

    def integrator(self):
        cv, av, at, ini = np.zeros(6), np.zeros(6), ctypes.c_double(), ctypes.c_int()
        self.b._check(self.lib.eqf_tiled_get_integrator(self._h, self._dp(cv), self._dp(av), ctypes.byref(at), ctypes.byref(ini)), "eqf_tiled_get_integrator")
        return {"currentVelocity": cv, "accumulatedVelocity": av, "accumulatedTime": at.value, "initialised": bool(ini.value)}

    def last_update(self):
        N = self.num_landmarks()
        d, g, G = np.zeros(2 * N), np.zeros(11 + 3 * N), np.zeros(9 + 3 * N)
        self.b._check(self.lib.eqf_tiled_get_last_update(self._h, self._dp(d), self._dp(g), self._dp(G)), "eqf_tiled_get_last_update")
        return {"delta": d, "gamma": g, "Gamma": G}

    def base_rows(self):
        N = self.num_landmarks()
        out = np.zeros((11, 11 + 3 * N))
        self.b._check(self.lib.eqf_tiled_get_base(self._h, self._dp(out), out.shape[1]), "eqf_tiled_get_base")
        return out

    def set_state(self, st):
        """st: a snapshot as FilterBatch.dump_state() makes it (ids, origin, group, bias, sigma, time, currentVelocity, accumulatedVelocity,
        accumulatedTime, initialised); only the first 11 rows of sigma are taken (the replicated base panel)."""
        N = len(st["ids"])
        o, g = st["origin"], st["group"]

        def arr(a, shape):
            out = np.zeros(shape)
            if N:
                out[...] = np.asarray(a, dtype=np.float64).reshape(shape)
            return np.ascontiguousarray(out)

        p0, Qq, Qa = arr(o["p"], (max(N, 1), 3)), arr(g["Qq"], (max(N, 1), 4)), arr(g["Qa"], (max(N, 1),))
        sb = np.ascontiguousarray(np.asarray(st["sigma"], dtype=np.float64)[:11])
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        arrs = [f(x) for x in (o["q"], o["x"], o["v"], p0, g["Aq"], g["Ax"], g["w"], Qq, Qa, st["bias"], sb)]
        cv, av = f(st["currentVelocity"]), f(st["accumulatedVelocity"])
        self.b._check(self.lib.eqf_tiled_set_state(self._h, N, *[self._dp(a) for a in arrs], sb.shape[1], float(st["time"]), self._dp(cv),
                                                   self._dp(av), float(st["accumulatedTime"]), int(st["initialised"])), "eqf_tiled_set_state")
