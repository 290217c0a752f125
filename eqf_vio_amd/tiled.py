"""One EqF filter with Sigma 2-D block-partitioned over a process grid -- BASELINE configs[4] (N = 4000 landmarks, Sigma = 1.15 GB
fp64), SURVEY.md 8(e) row 2.  One process per GPU; the host side of eqf_vio/include/eqf_vio/VIOFilter.h:41-88 for this configuration:
`TiledFilter.processIMUData`, `processVisionData`, `stateEstimate`, `stateCovariance`.

This module is GLUE.  The filter's host loop -- the landmark bookkeeping on slots, the IMU bursts, the two distributed Cholesky
factorisations of an update with their look-ahead, the covariance downdate -- is C++ behind the C ABI (csrc/eqf_tiledf.hip: eqf_tf_*,
include/eqf_vio_amd.h); what is here is (1) the ctypes binding of those entry points and (2) `ProcessGrid`: the ONE callback the C++
loop needs from its host, "broadcast this device buffer along my process row / column, ordered on this HIP stream", answered with
torch.distributed (backend "nccl" = RCCL over xGMI on a GPU node, "gloo" in the tests).  A C++ host answers the same callback with
ncclBroadcast.  The per-rank kernels can also be driven directly (eqf_vio_amd/tiled_backend.py: HipBackend); the schedule
as an executable specification in Python, with a numpy double of the kernels for CPU process grids, is tests/tiled_reference.py.
"""
import ctypes as C

import numpy as np
import torch

from . import binding
from .tiled_backend import HipBackend  # noqa: F401  (re-exported: kernel-level tests and scripts)

_BCAST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)


class _Comm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("bcast", _BCAST)]


class _DevBuf:
    """`nbytes` bytes of device memory at `ptr` as a CUDA-array-interface object (-> torch.as_tensor without a copy)."""

    def __init__(self, ptr, count, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 3, "strides": None}


def _lib():
    L = binding.lib()
    if not getattr(L, "_eqf_tf_bound", False):
        vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
        L.eqf_tf_create.argtypes = [C.POINTER(binding.Settings), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_Comm), C.POINTER(vp)]
        L.eqf_tf_destroy.argtypes = [vp]
        L.eqf_tf_destroy.restype = None
        L.eqf_tf_set_option.argtypes = [vp, C.c_char_p, C.c_int]
        L.eqf_tf_process_imu.argtypes = [vp, C.c_double, dp, dp]
        L.eqf_tf_process_vision.argtypes = [vp, C.c_double, C.c_int, ip, dp]
        for name in ("eqf_tf_synchronize", "eqf_tf_check", "eqf_tf_device_error", "eqf_tf_num_landmarks", "eqf_tf_num_slots"):
            getattr(L, name).argtypes = [vp]
        L.eqf_tf_get_ids.argtypes = [vp, ip, ip]
        L.eqf_tf_get_time.argtypes = [vp, dp]
        L.eqf_tf_get_state_estimate.argtypes = [vp, dp, dp, dp, dp]
        L.eqf_tf_get_bias.argtypes = [vp, dp]
        L.eqf_tf_get_last_update.argtypes = [vp, dp, dp, dp]
        L.eqf_tf_get_sigma.argtypes = [vp, dp, C.c_int, C.c_int]
        L.eqf_tf_set_state.argtypes = [vp, C.c_int, ip] + [dp] * 11 + [C.c_int, C.c_double, dp, dp, C.c_double, C.c_int]
        L.eqf_tf_get_churn_stats.argtypes = [vp, C.POINTER(C.c_longlong)]
        L.eqf_tf_local_matrix.argtypes = [vp, C.POINTER(dp), ip, ip, ip]
        L.eqf_tf_get_phases.argtypes = [vp, dp]
        L.eqf_tf_phase_name.argtypes = [C.c_int]
        L.eqf_tf_phase_name.restype = C.c_char_p
        L.eqf_tf_last_error.argtypes = [vp]
        L.eqf_tf_last_error.restype = C.c_char_p
        L.eqf_tf_tiled_handle.argtypes = [vp]
        L.eqf_tf_tiled_handle.restype = vp
        L._eqf_tf_bound = True
    return L


class ProcessGrid:
    """Pr x Pc process grid over a torch.distributed group, rank = pr * Pc + pc, with one sub-group per process row / column -- and a second
    set for the E-chain of an update, which runs next to the S-chain on its own stream (two collectives of ONE communicator must not be in
    flight at the same time).  `bcast` is the callback of eqf_tf_comm (include/eqf_vio_amd.h)."""

    def __init__(self, dist, Pr, Pc, device="cpu"):
        self.dist, self.Pr, self.Pc, self.device = dist, Pr, Pc, device
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        assert self.world == Pr * Pc and Pc % Pr == 0, "the process grid needs Pr | Pc"
        self.pr, self.pc = divmod(self.rank, Pc)
        self.groups = [[None, None, None], [None, None, None]]  # [chain][row, column, everybody]
        if dist is not None and self.world > 1:
            # (new_group is collective over the whole job: every rank creates every group, in the same order)
            for chain in range(2):
                for r in range(Pr):
                    grp = dist.new_group([r * Pc + c for c in range(Pc)])
                    if r == self.pr:
                        self.groups[chain][0] = grp
                for c in range(Pc):
                    grp = dist.new_group([r * Pc + c for r in range(Pr)])
                    if c == self.pc:
                        self.groups[chain][1] = grp
        self.error = None
        # what went through the callback (bench.py / tests: the exchange volume of an update against DESIGN.md section 7's SUMMA-restricted figure)
        self.bcast_calls = 0
        self.bcast_bytes = 0           # every broadcast this rank took part in
        self.bcast_bytes_received = 0  # ... of which it was not the root
        self._cb = _BCAST(self._bcast)
        self.comm = _Comm(None, self._cb)

    def backend_name(self):
        return self.dist.get_backend() if (self.dist is not None and self.world > 1) else ""

    def _bcast(self, ctx, group, chain, root, buf, nbytes, stream):
        try:
            t = torch.as_tensor(_DevBuf(buf, nbytes // 8), device=self.device)
            src = (self.pr * self.Pc + root) if group == 0 else ((root * self.Pc + self.pc) if group == 1 else root)
            self.bcast_calls += 1
            self.bcast_bytes += nbytes
            if src != self.rank:
                self.bcast_bytes_received += nbytes
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device)):
                self.dist.broadcast(t, src=src, group=self.groups[chain][group] if group < 2 else None)
            return 0
        except Exception as e:  # (an exception must not unwind through the C++ frames)
            self.error = e
            return 1


class TiledFilter:
    """VIOFilter (VIOFilter.h:41-88) for one filter whose Sigma is partitioned over `grid`; every rank of the grid makes the same calls with
    the same arguments.  `backend`: a HipBackend -- its settings, capacity, device and CU reservation are taken, and from then on it answers
    for the filter's replicated state (its getters, in SLOT order) -- or a settings dict with capacity= / device_index= given."""

    def __init__(self, grid, backend, block_landmarks, capacity=None, device_index=None, reserve_cus=None):
        L = self.lib = _lib()
        self.g, self.bl = grid, int(block_landmarks)
        if isinstance(backend, HipBackend):
            self.be = backend
            settings, dev = backend.settings, backend.dev
            cap = int(capacity if capacity is not None else backend.cap)
            reserve = backend.reserve if reserve_cus is None else int(reserve_cus)
        else:
            self.be = None
            settings = binding.settings_from_dict(backend) if isinstance(backend, dict) else backend
            dev, cap = int(device_index or 0), int(capacity)
            reserve = -1 if reserve_cus is None else int(reserve_cus)
        self.settings, self.cap, self.dev = settings, cap, dev
        self.device = torch.device("cuda", dev)
        torch.cuda.init()
        self._h = C.c_void_p()
        comm = C.byref(grid.comm) if grid.world > 1 else None
        binding._check(L.eqf_tf_create(C.byref(settings), cap, self.bl, grid.Pr, grid.Pc, grid.rank, dev, reserve, comm, C.byref(self._h)), "eqf_tf_create")
        if self.be is not None:
            self.be.adopt(L.eqf_tf_tiled_handle(self._h))
        # The two factorisations of an update run side by side on two stream pairs -- on one rank, and over gloo (the process-grid tests: the
        # ranks share one GPU).  Every exchange of the handle is issued on ONE stream in program order (csrc/eqf_tiledf.hip: bcast), so a
        # communicator only ever sees an ordered sequence; still, the interleaved form has never run over RCCL on a node, so over nccl the
        # chains run one after the other unless EQF_TILED_OVERLAP_CHAINS=1 asks for the overlap (the library's own default, eqf_tf_create).
        import os

        env = os.environ.get("EQF_TILED_OVERLAP_CHAINS")
        self.overlap_chains = (env != "0") if env is not None else not (grid.world > 1 and grid.backend_name() == "nccl")
        self._phases_on = False

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            if self.be is not None:
                self.be.release()
            self.lib.eqf_tf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if self.g.error is not None:
            e, self.g.error = self.g.error, None
            raise e
        if rc < 0:
            msg = self.lib.eqf_tf_last_error(self._h).decode()
            if rc == binding.ERR_NUMERIC:
                raise ArithmeticError(msg or "a pivot of S or Sigma_e was not positive (distributed factorisation)")
            if rc == binding.ERR_CAPACITY:
                raise RuntimeError(msg or "capacity exceeded")
            if rc == binding.ERR_UNSORTED:
                raise ValueError("bearings must come with strictly ascending ids (VIOFilter.cpp:239-240)")
            raise binding.EqfError(rc, what)
        return rc

    def _opt(self, name, value):
        self._check(self.lib.eqf_tf_set_option(self._h, name.encode(), int(value)), "eqf_tf_set_option")

    # ---- options (attributes, as the Python loop had them)
    # 0 (default): the covariance downdate on the fp64 matrix cores; 5 / 6 / 7: on the integer matrix pipe from that many 7-bit slices (round 6)
    downdate_slices = property(lambda s: getattr(s, "_dd_slices", 0), lambda s, v: (setattr(s, "_dd_slices", int(v)), s._opt("downdate_slices", v))[0])
    # the same for the trailing products of the two factorisations (five slices are enough there: scripts/slice_precision_study_chain.py)
    chain_slices = property(lambda s: getattr(s, "_chain_slices", 0), lambda s, v: (setattr(s, "_chain_slices", int(v)), s._opt("chain_slices", v))[0])
    # > 0: block-row solves split recursively down to this many 64-row blocks (0, the default: one split -- the recursion measured slower)
    trsm_leaf = property(lambda s: getattr(s, "_trsm_leaf", 0), lambda s, v: (setattr(s, "_trsm_leaf", int(v)), s._opt("trsm_leaf", v))[0])
    # percent of the S-chain's block rows whose share of the downdate runs on the E-chain's stream while the chains are still going
    # a block row's solve as one product with the explicit inverse of its diagonal factor (csrc/eqf_tiledf.hip: solveInverse)
    solve_inverse = property(lambda s: getattr(s, "_solve_inverse", False), lambda s, v: (setattr(s, "_solve_inverse", bool(v)), s._opt("solve_inverse", v))[0])
    downdate_early = property(lambda s: getattr(s, "_dd_early", 0), lambda s, v: (setattr(s, "_dd_early", int(v)), s._opt("downdate_early", v))[0])
    overlap_chains = property(lambda s: s._overlap, lambda s, v: (setattr(s, "_overlap", bool(v)), s._opt("overlap_chains", v))[0])
    lookahead = property(lambda s: getattr(s, "_lookahead", True), lambda s, v: (setattr(s, "_lookahead", bool(v)), s._opt("lookahead", v))[0])
    burst = property(lambda s: getattr(s, "_burst", True), lambda s, v: (setattr(s, "_burst", bool(v)), s._opt("burst", v))[0])
    check_every = property(lambda s: getattr(s, "_check_every", 1), lambda s, v: (setattr(s, "_check_every", int(v)), s._opt("check_every", v))[0])

    @property
    def phase_ms(self):
        return self._phase_ms if self._phases_on else None

    @phase_ms.setter
    def phase_ms(self, v):
        self._phases_on = v is not None
        self._phase_ms = {} if v is not None else None
        self._opt("profiling", 1 if v is not None else 0)

    def collect_phases(self):
        ms = np.zeros(7)
        self._check(self.lib.eqf_tf_get_phases(self._h, binding._p(ms)), "eqf_tf_get_phases")
        self._phase_ms = {self.lib.eqf_tf_phase_name(i).decode(): float(ms[i]) for i in range(7) if ms[i] > 0}
        return self._phase_ms

    # ---- VIOFilter::processIMUData / processVisionData (VIOFilter.cpp:120-131, :232-302)
    def processIMUData(self, stamp, omega, accel):
        w, a = np.ascontiguousarray(omega, dtype=np.float64), np.ascontiguousarray(accel, dtype=np.float64)
        return self._check(self.lib.eqf_tf_process_imu(self._h, float(stamp), binding._p(w), binding._p(a)), "eqf_tf_process_imu")

    def processVisionData(self, stamp, ids, bearings):
        ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(-1)
        y = np.ascontiguousarray(bearings, dtype=np.float64).reshape(-1, 3)
        if len(ids) != len(y):
            raise ValueError("bearings must come with strictly ascending ids (VIOFilter.cpp:239-240)")
        return self._check(self.lib.eqf_tf_process_vision(self._h, float(stamp), len(ids), ids.ctypes.data_as(C.POINTER(C.c_int)), binding._p(y)),
                           "eqf_tf_process_vision")

    def check(self):
        """Raises if a pivot of S or Sigma_e was not positive since the last look (synchronises)."""
        self._check(self.lib.eqf_tf_check(self._h), "eqf_tf_check")

    def synchronize(self):
        self._check(self.lib.eqf_tf_synchronize(self._h), "eqf_tf_synchronize")

    def device_error(self):
        return self.lib.eqf_tf_device_error(self._h)

    # ---- bookkeeping the tests and the bench look at
    @property
    def ids(self):
        n = self.lib.eqf_tf_num_landmarks(self._h)
        out = np.zeros(max(n, 1), dtype=np.int32)
        self.lib.eqf_tf_get_ids(self._h, out.ctypes.data_as(C.POINTER(C.c_int)), None)
        return out[:n].astype(np.int64)

    @property
    def slot_of(self):
        n = self.lib.eqf_tf_num_landmarks(self._h)
        ids, sl = np.zeros(max(n, 1), dtype=np.int32), np.zeros(max(n, 1), dtype=np.int32)
        self.lib.eqf_tf_get_ids(self._h, ids.ctypes.data_as(C.POINTER(C.c_int)), sl.ctypes.data_as(C.POINTER(C.c_int)))
        return sl[:n].astype(np.int64)

    @property
    def nslots(self):
        return self.lib.eqf_tf_num_slots(self._h)

    @property
    def taken(self):
        t = np.zeros(self.cap, dtype=bool)
        t[self.slot_of] = True
        return t

    @property
    def churn_stats(self):
        s = (C.c_longlong * 3)()
        self.lib.eqf_tf_get_churn_stats(self._h, s)
        return dict(removed_old=int(s[0]), removed_outliers=int(s[1]), added=int(s[2]))

    @property
    def Sll(self):
        """the rank's local matrix (3 nlr x 3 nlc) as a torch view of the handle's device memory"""
        p, r, c, ld = C.POINTER(C.c_double)(), C.c_int(), C.c_int(), C.c_int()
        self.lib.eqf_tf_local_matrix(self._h, C.byref(p), C.byref(r), C.byref(c), C.byref(ld))
        if not p or r.value == 0 or c.value == 0:
            return torch.zeros(0, 0, dtype=torch.float64, device=self.device)
        flat = torch.as_tensor(_DevBuf(C.cast(p, C.c_void_p).value, r.value * ld.value), device=self.device)
        return flat.view(r.value, ld.value)[:, : c.value]

    # ---- getters (reference order)
    def getTime(self):
        t = C.c_double()
        self._check(self.lib.eqf_tf_get_time(self._h, C.byref(t)), "eqf_tf_get_time")
        return t.value

    def stateEstimate(self):
        """VIOFilter::stateEstimate (:304): landmarks in the reference's order"""
        n = self.lib.eqf_tf_num_landmarks(self._h)
        q, x, v, p = np.zeros(4), np.zeros(3), np.zeros(3), np.zeros((max(n, 1), 3))
        self._check(self.lib.eqf_tf_get_state_estimate(self._h, binding._p(q), binding._p(x), binding._p(v), binding._p(p)), "eqf_tf_get_state_estimate")
        return {"q": q, "x": x, "v": v, "p": p[:n], "ids": self.ids}

    def bias(self):
        b = np.zeros(6)
        self._check(self.lib.eqf_tf_get_bias(self._h, binding._p(b)), "eqf_tf_get_bias")
        return b

    def lastUpdate(self):
        """delta (2 N), gamma (11 + 3 N), Gamma (9 + 3 N) of the last update, landmarks in the reference's order"""
        n = self.lib.eqf_tf_num_landmarks(self._h)
        d, g, G = np.zeros(max(2 * n, 1)), np.zeros(11 + 3 * n), np.zeros(9 + 3 * n)
        self._check(self.lib.eqf_tf_get_last_update(self._h, binding._p(d), binding._p(g), binding._p(G)), "eqf_tf_get_last_update")
        return {"delta": d[: 2 * n], "gamma": g, "Gamma": G}

    def _sigma(self, slot_order):
        n = 11 + 3 * (self.nslots if slot_order else self.lib.eqf_tf_num_landmarks(self._h))
        S = np.zeros((n, n))
        self._check(self.lib.eqf_tf_get_sigma(self._h, binding._p(S), n, int(slot_order)), "eqf_tf_get_sigma")
        return S

    def stateCovariance(self):
        """Dense Sigma (reference index map and landmark order) gathered to every rank (VIOFilter::stateCovariance, :306-309); collective."""
        return self._sigma(False)

    def slotCovariance(self):
        """Dense Sigma over ALL slots in use, holes included (slot order) -- tests of the hole invariants; collective."""
        return self._sigma(True)

    def initialise_from(self, st):
        """Restart from a single-GPU snapshot (FilterBatch.dump_state(); every rank holds the dense Sigma once, here)."""
        ids = np.ascontiguousarray(st["ids"], dtype=np.int32)
        N = len(ids)
        o, g = st["origin"], st["group"]

        def arr(a, shape):
            out = np.zeros(shape)
            if N:
                out[...] = np.asarray(a, dtype=np.float64).reshape(shape)
            return np.ascontiguousarray(out)

        p0, Qq, Qa = arr(o["p"], (max(N, 1), 3)), arr(g["Qq"], (max(N, 1), 4)), arr(g["Qa"], (max(N, 1),))
        S = np.ascontiguousarray(st["sigma"], dtype=np.float64)
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        arrs = [f(x) for x in (o["q"], o["x"], o["v"], p0, g["Aq"], g["Ax"], g["w"], Qq, Qa, st["bias"], S)]
        cv, av = f(st["currentVelocity"]), f(st["accumulatedVelocity"])
        self._check(self.lib.eqf_tf_set_state(self._h, N, ids.ctypes.data_as(C.POINTER(C.c_int)), *[binding._p(a) for a in arrs], S.shape[1],
                                              float(st["time"]), binding._p(cv), binding._p(av), float(st["accumulatedTime"]), int(st["initialised"])),
                    "eqf_tf_set_state")
