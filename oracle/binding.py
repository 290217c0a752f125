"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/liboracle.so (the fp64 C++ CPU oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

_SCALARS = [
    "biasOmegaProcessVariance", "biasAccelProcessVariance", "gravityProcessVariance", "velocityProcessVariance",
    "pointProcessVariance", "velOmegaVariance", "velAccelVariance", "measurementVariance", "initialGravityVariance",
    "initialVelocityVariance", "initialPointVariance", "initialBiasOmegaVariance", "initialBiasAccelVariance",
    "initialSceneDepth", "outlierThreshold",
]
_FLAGS = ["useInnovationLift", "useDiscreteInnovationLift", "useDiscreteVelocityLift", "fastRiccati"]
# VIOFilterSettings.h:29-50
_DEFAULTS = dict(zip(_SCALARS, [0.001] * 5 + [0.1, 0.1, 0.1, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.01]))
_DEFAULTS.update(dict(useInnovationLift=True, useDiscreteInnovationLift=True, useDiscreteVelocityLift=True, fastRiccati=False))


class OracleSettings(C.Structure):
    _fields_ = [
        ("v", C.c_double * 15),
        ("flags", C.c_int * 4),
        ("initialAccelBias", C.c_double * 3),
        ("initialOmegaBias", C.c_double * 3),
        ("cameraOffset_x", C.c_double * 3),
        ("cameraOffset_q", C.c_double * 4),
    ]


def make_settings(d):
    """d: dict with the reference's setting names (+ cameraOffset_x / cameraOffset_q [w,x,y,z])."""
    s = OracleSettings()
    for i, k in enumerate(_SCALARS):
        s.v[i] = float(d.get(k, _DEFAULTS[k]))
    for i, k in enumerate(_FLAGS):
        s.flags[i] = int(bool(d.get(k, _DEFAULTS[k])))
    for name, n, dflt in (("initialAccelBias", 3, [0, 0, 0]), ("initialOmegaBias", 3, [0, 0, 0]),
                          ("cameraOffset_x", 3, [0, 0, 0]), ("cameraOffset_q", 4, [1, 0, 0, 0])):
        arr = np.asarray(d.get(name, dflt), dtype=float)
        for i in range(n):
            getattr(s, name)[i] = arr[i]
    return s


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "eqf_oracle.cpp")
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(OracleSettings)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_set_structured.argtypes = [C.c_void_p, C.c_int]
        L.oracle_process_imu.argtypes = [C.c_void_p, C.c_double, dp, dp]
        L.oracle_process_vision.argtypes = [C.c_void_p, C.c_double, C.c_int, ip, dp]
        L.oracle_num_landmarks.argtypes = [C.c_void_p]
        L.oracle_get_time.restype = C.c_double
        L.oracle_get_time.argtypes = [C.c_void_p]
        L.oracle_get_ids.argtypes = [C.c_void_p, ip]
        for f in ("oracle_get_sigma", "oracle_get_bias", "oracle_get_xi0", "oracle_get_estimate", "oracle_get_group"):
            getattr(L, f).argtypes = [C.c_void_p, dp]
        L.oracle_set_sigma.argtypes = [C.c_void_p, dp]
        L.oracle_set_state.argtypes = [C.c_void_p, C.c_int, ip, dp, dp, dp, dp, C.c_double, dp, dp, C.c_double, C.c_int]
        L.oracle_get_last.argtypes = [C.c_void_p, dp, dp, dp]
        L.oracle_matrices.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp, dp, dp]
        L.oracle_bundle_lift.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp, dp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleFilter:
    """The reference's VIOFilter API on the C++ fp64 oracle.

    structured=False: the reference's dense operation sequence (the parity anchor and the "cpu_baseline").
    structured=True:  the same equations without the structural-zero work and with Cholesky-form S^-1 / Sigma_e^-1
                      ("cpu_structured"); pinned against the dense form by tests/test_oracle_structured.py."""

    def __init__(self, settings_dict, structured=False):
        self._s = make_settings(settings_dict)
        self._h = C.c_void_p(lib().oracle_create(C.byref(self._s)))
        self.structured = bool(structured)
        if structured:
            lib().oracle_set_structured(self._h, 1)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_destroy(self._h)
            self._h = None

    def processIMUData(self, stamp, omega, accel):
        w = np.ascontiguousarray(omega, dtype=np.float64)
        a = np.ascontiguousarray(accel, dtype=np.float64)
        rc = lib().oracle_process_imu(self._h, float(stamp), _dp(w), _dp(a))
        if rc:
            raise ValueError("The vectors cannot be exactly opposing.")

    def processVisionData(self, stamp, ids, bearings):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        y = np.ascontiguousarray(bearings, dtype=np.float64)
        rc = lib().oracle_process_vision(self._h, float(stamp), len(ids), ids.ctypes.data_as(C.POINTER(C.c_int)), _dp(y))
        if rc:
            raise ValueError("The vectors cannot be exactly opposing.")

    @property
    def N(self):
        return lib().oracle_num_landmarks(self._h)

    def getTime(self):
        return lib().oracle_get_time(self._h)

    def ids(self):
        out = np.zeros(self.N, dtype=np.int32)
        lib().oracle_get_ids(self._h, out.ctypes.data_as(C.POINTER(C.c_int)))
        return out

    def stateCovariance(self):
        n = 11 + 3 * self.N
        out = np.zeros((n, n))
        lib().oracle_get_sigma(self._h, _dp(out))
        return out

    def setCovariance(self, S):
        S = np.ascontiguousarray(S, dtype=np.float64)
        assert S.shape == (11 + 3 * self.N,) * 2
        lib().oracle_set_sigma(self._h, _dp(S))

    def set_state(self, st):
        """Inject a snapshot in the format of eqf_vio_amd.binding.FilterBatch.dump_state (kernel-level parity tests)."""
        ids = np.ascontiguousarray(st["ids"], dtype=np.int32)
        o, g = st["origin"], st["group"]
        state = np.ascontiguousarray(pack_state(o["q"], o["x"], o["v"], o["p"]))
        group = np.ascontiguousarray(pack_group(g["Aq"], g["Ax"], g["w"], g["Qq"], g["Qa"]))
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (st["bias"], st["sigma"], st["currentVelocity"], st["accumulatedVelocity"])]
        lib().oracle_set_state(self._h, len(ids), ids.ctypes.data_as(C.POINTER(C.c_int)), _dp(state), _dp(group), _dp(arrs[0]),
                               _dp(arrs[1]), float(st["time"]), _dp(arrs[2]), _dp(arrs[3]), float(st["accumulatedTime"]),
                               int(st["initialised"]))

    def bias(self):
        out = np.zeros(6)
        lib().oracle_get_bias(self._h, _dp(out))
        return out

    def _state(self, fn):
        out = np.zeros(10 + 3 * self.N)
        fn(self._h, _dp(out))
        return dict(q=out[0:4].copy(), x=out[4:7].copy(), v=out[7:10].copy(), p=out[10:].reshape(-1, 3).copy())

    def xi0(self):
        return self._state(lib().oracle_get_xi0)

    def stateEstimate(self):
        return self._state(lib().oracle_get_estimate)

    def group(self):
        out = np.zeros(10 + 5 * self.N)
        lib().oracle_get_group(self._h, _dp(out))
        Q = out[10:].reshape(-1, 5)
        return dict(Aq=out[0:4].copy(), Ax=out[4:7].copy(), w=out[7:10].copy(), Qq=Q[:, 0:4].copy(), Qa=Q[:, 4].copy())

    def last_update(self):
        N = self.N
        delta = np.zeros(2 * N)
        gamma = np.zeros(11 + 3 * N)
        Gamma = np.zeros(9 + 3 * N)
        k = lib().oracle_get_last(self._h, _dp(delta), _dp(gamma), _dp(Gamma))
        if k == 0:
            return None
        return dict(delta=delta, gamma=gamma, Gamma=Gamma)


def pack_group(Aq, Ax, w, Qq, Qa):
    N = len(Qa)
    out = np.zeros(10 + 5 * N)
    out[0:4], out[4:7], out[7:10] = Aq, Ax, w
    out[10:] = np.hstack([np.asarray(Qq).reshape(N, 4), np.asarray(Qa).reshape(N, 1)]).reshape(-1)
    return out


def pack_state(q, x, v, p):
    p = np.asarray(p).reshape(-1, 3)
    out = np.zeros(10 + 3 * len(p))
    out[0:4], out[4:7], out[7:10] = q, x, v
    out[10:] = p.reshape(-1)
    return out


def matrices(group, state, camq, camx, omega):
    """A0 (5+3N)^2, B (5+3N)x6, C0 2N x (5+3N) from the C++ oracle (EqFMatrices.cpp:277-382)."""
    N = (len(state) - 10) // 3
    ne = 5 + 3 * N
    A0 = np.zeros((ne, ne))
    B = np.zeros((ne, 6))
    C0 = np.zeros((2 * N, ne))
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (group, state, camq, camx, omega)]
    rc = lib().oracle_matrices(N, *[_dp(a) for a in arrs], _dp(A0), _dp(B), _dp(C0))
    if rc:
        raise ValueError("The vectors cannot be exactly opposing.")
    return A0, B, C0


def bundle_lift(group, state, camq, camx, base, SigmaE):
    N = (len(state) - 10) // 3
    out = np.zeros(9 + 3 * N)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (group, state, camq, camx, base, SigmaE)]
    lib().oracle_bundle_lift(N, *[_dp(a) for a in arrs], _dp(out))
    return out
