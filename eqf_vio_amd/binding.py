"""ctypes binding of libeqf_vio_amd.so (the C ABI declared in include/eqf_vio_amd.h).

There is no CPU fallback: if the shared library is missing or no MI355X is visible, loading /
eqf_create fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (EQF_VIO_AMD_LIB: an instrumented build of the same library for the stamp scripts, scripts/README.md -- never a fallback)
LIB_PATH = os.environ.get("EQF_VIO_AMD_LIB") or os.path.join(_HERE, "libeqf_vio_amd.so")

EQF_OK = 0
SKIPPED_BEFORE_FIRST_IMU = 1
SKIPPED_NONPOSITIVE_DT = 2
SKIPPED_NOT_INITIALISED = 3
SKIPPED_NO_BEARINGS = 4
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_UNSORTED, ERR_NUMERIC, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7
PRECISION_F64, PRECISION_F32 = 0, 1
PROF_CLASSES = 11

_ERR_NAMES = {
    -1: "EQF_ERR_INVALID", -2: "EQF_ERR_NO_DEVICE", -3: "EQF_ERR_HIP", -4: "EQF_ERR_CAPACITY",
    -5: "EQF_ERR_UNSORTED", -6: "EQF_ERR_NUMERIC", -7: "EQF_ERR_UNSUPPORTED",
}


class EqfError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"{what}: {_ERR_NAMES.get(code, code)}")
        self.code = code


class Settings(C.Structure):
    """eqf_settings == VIOFilter::Settings (eqf_vio/include/eqf_vio/VIOFilterSettings.h:28-54)."""

    _fields_ = [
        ("biasOmegaProcessVariance", C.c_double), ("biasAccelProcessVariance", C.c_double),
        ("gravityProcessVariance", C.c_double), ("velocityProcessVariance", C.c_double),
        ("pointProcessVariance", C.c_double), ("velOmegaVariance", C.c_double), ("velAccelVariance", C.c_double),
        ("measurementVariance", C.c_double), ("initialGravityVariance", C.c_double),
        ("initialVelocityVariance", C.c_double), ("initialPointVariance", C.c_double),
        ("initialBiasOmegaVariance", C.c_double), ("initialBiasAccelVariance", C.c_double),
        ("initialSceneDepth", C.c_double), ("outlierThreshold", C.c_double),
        ("useInnovationLift", C.c_int), ("useDiscreteInnovationLift", C.c_int), ("useDiscreteVelocityLift", C.c_int),
        ("fastRiccati", C.c_int),
        ("initialAccelBias", C.c_double * 3), ("initialOmegaBias", C.c_double * 3),
        ("cameraOffset_x", C.c_double * 3), ("cameraOffset_q", C.c_double * 4),
    ]


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

EXPORTED_SYMBOLS = [
    "eqf_settings_default", "eqf_create", "eqf_destroy", "eqf_reset", "eqf_process_imu", "eqf_process_vision",
    "eqf_stream_upload", "eqf_stream_imu", "eqf_stream_vision", "eqf_synchronize", "eqf_get_time", "eqf_num_landmarks",
    "eqf_get_ids", "eqf_get_state_estimate", "eqf_get_origin", "eqf_get_group", "eqf_get_bias", "eqf_get_sigma",
    "eqf_set_sigma", "eqf_set_state", "eqf_set_camera_offset", "eqf_get_integrator", "eqf_get_last_update", "eqf_debug_get_blocks", "eqf_device_error", "eqf_debug_drop_role", "eqf_debug_option", "eqf_debug_launch_shape", "eqf_set_dense_propagate", "eqf_set_imu_burst", "eqf_profile_enable",
    "eqf_profile_get", "eqf_profile_class_name", "eqf_version", "eqf_build_info", "eqf_tile_propagate", "eqf_tile_downdate", "eqf_tile_potrf", "eqf_tile_trsm", "eqf_tile_gemm_tn", "eqf_tile_mirror", "eqf_tile_downdate_i8", "eqf_tile_gemm_tn_i8", "eqf_tile_i8_workspace_bytes", "eqf_stream_create_masked", "eqf_stream_destroy",
    "eqf_tiled_create", "eqf_tiled_destroy", "eqf_tiled_set_stream", "eqf_tiled_set_geometry", "eqf_tiled_propagate", "eqf_tiled_add_landmarks",
    "eqf_tiled_edit_landmarks", "eqf_tiled_propagate_burst", "eqf_tiled_stage_bearings", "eqf_tiled_pingpong",
    "eqf_tiled_update_prep", "eqf_tiled_update_finish", "eqf_tiled_synchronize", "eqf_tiled_num_landmarks", "eqf_tiled_get_time",
    "eqf_tiled_device_error", "eqf_tiled_get_state_estimate", "eqf_tiled_get_origin", "eqf_tiled_get_group", "eqf_tiled_get_bias",
    "eqf_tiled_get_last_update", "eqf_tiled_get_integrator", "eqf_tiled_get_base", "eqf_tiled_set_state",
    "eqf_tf_create", "eqf_tf_destroy", "eqf_tf_set_option", "eqf_tf_process_imu", "eqf_tf_process_vision", "eqf_tf_synchronize", "eqf_tf_check",
    "eqf_tf_device_error", "eqf_tf_num_landmarks", "eqf_tf_num_slots", "eqf_tf_get_ids", "eqf_tf_get_time", "eqf_tf_get_state_estimate",
    "eqf_tf_get_bias", "eqf_tf_get_last_update", "eqf_tf_get_sigma", "eqf_tf_set_state", "eqf_tf_get_churn_stats", "eqf_tf_local_matrix",
    "eqf_tf_get_phases", "eqf_tf_phase_name", "eqf_tf_last_error", "eqf_tf_tiled_handle", "eqf_tf_graph_launches",
]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  eqf_vio_amd has no CPU fallback."
            )
        # torch wheels ship their own HIP runtime (torch/lib/libamdhip64.so).  A process must end up with ONE runtime: if this library pulled
        # in the system's first, a later `import torch` + torch.cuda initialisation sees "No HIP GPUs" (measured on the MI355X box: the
        # partitioned filter, whose device memory is torch's, created after a FilterBatch).  So when torch is installed it is imported
        # first and this library binds to the runtime torch loaded; hosts without torch get the system runtime as before.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.eqf_settings_default.argtypes = [C.POINTER(Settings)]
        L.eqf_settings_default.restype = None
        L.eqf_create.argtypes = [C.POINTER(Settings), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.eqf_destroy.argtypes = [vp]
        L.eqf_destroy.restype = None
        L.eqf_reset.argtypes = [vp]
        L.eqf_process_imu.argtypes = [vp, _dp, _dp, _dp, _ip]
        L.eqf_process_vision.argtypes = [vp, _dp, _ip, _ip, _dp, C.c_int, _ip]
        L.eqf_stream_upload.argtypes = [vp, C.c_int, _dp, C.c_int, _dp, C.c_int, _ip, _dp]
        L.eqf_stream_imu.argtypes = [vp, C.c_int]
        L.eqf_stream_vision.argtypes = [vp, C.c_int]
        L.eqf_synchronize.argtypes = [vp]
        L.eqf_get_time.argtypes = [vp, _dp]
        L.eqf_num_landmarks.argtypes = [vp, C.c_int]
        L.eqf_get_ids.argtypes = [vp, C.c_int, _ip]
        L.eqf_get_state_estimate.argtypes = [vp, C.c_int, _dp, _dp, _dp, _dp]
        L.eqf_get_origin.argtypes = [vp, C.c_int, _dp, _dp, _dp, _dp]
        L.eqf_get_group.argtypes = [vp, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.eqf_get_bias.argtypes = [vp, C.c_int, _dp]
        L.eqf_get_sigma.argtypes = [vp, C.c_int, _dp, C.c_int]
        L.eqf_set_sigma.argtypes = [vp, C.c_int, _dp, C.c_int]
        L.eqf_get_last_update.argtypes = [vp, C.c_int, _dp, _dp, _dp]
        L.eqf_set_state.argtypes = [vp, C.c_int, C.c_int, _ip] + [_dp] * 11 + [C.c_int, C.c_double, _dp, _dp, C.c_double, C.c_int]
        L.eqf_get_integrator.argtypes = [vp, C.c_int, _dp, _dp, _dp, _ip]
        L.eqf_debug_get_blocks.argtypes = [vp, C.c_int, _dp, _dp, _dp]
        L.eqf_device_error.argtypes = [vp]
        L.eqf_debug_drop_role.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.eqf_debug_option.argtypes = [vp, C.c_char_p, C.c_int]
        if hasattr(L, "eqf_debug_launch_shape"):  # (an older build loaded through EQF_VIO_AMD_LIB for an A/B run may predate it)
            L.eqf_debug_launch_shape.argtypes = [vp, C.POINTER(C.c_int)]
        L.eqf_set_dense_propagate.argtypes = [vp, C.c_int]
        L.eqf_set_imu_burst.argtypes = [vp, C.c_int]
        L.eqf_profile_enable.argtypes = [vp, C.c_int]
        L.eqf_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_longlong), _dp]
        L.eqf_profile_class_name.argtypes = [C.c_int]
        L.eqf_profile_class_name.restype = C.c_char_p
        L.eqf_version.restype = C.c_char_p
        L.eqf_build_info.restype = C.c_char_p
        vpp = C.c_void_p
        L.eqf_tile_propagate.argtypes = [C.c_int, vpp, vpp, vpp, C.c_int, C.c_int, C.c_int, vpp, vpp, vpp, vpp, vpp, vpp, C.c_int, vpp, C.c_int,
                                         vpp, vpp, _dp, C.c_double, C.c_double, C.c_int]
        L.eqf_tile_downdate.argtypes = [C.c_int, vpp, vpp, C.c_int, C.c_int, C.c_int, vpp, C.c_int, vpp, C.c_int, C.c_int]
        L.eqf_tile_potrf.argtypes = [C.c_int, vpp, vpp, C.c_int, C.c_int, vpp, vpp]
        L.eqf_tile_trsm.argtypes = [C.c_int, vpp, vpp, C.c_int, C.c_int, vpp, vpp, C.c_int, C.c_int, C.c_int]
        L.eqf_tile_gemm_tn.argtypes = [C.c_int, vpp, vpp, C.c_int, C.c_int, C.c_int, vpp, C.c_int, vpp, C.c_int, C.c_int, C.c_double] + [C.c_int] * 8
        L.eqf_tile_mirror.argtypes = [C.c_int, vpp, vpp, C.c_int, C.c_int, C.c_int]
        if hasattr(L, "eqf_tile_downdate_i8"):  # (an older build loaded through EQF_VIO_AMD_LIB may predate it)
            L.eqf_tile_i8_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
            L.eqf_tile_i8_workspace_bytes.restype = C.c_size_t
            L.eqf_tile_downdate_i8.argtypes = [C.c_int, vpp, vpp, C.c_int, C.c_int, C.c_int, vpp, C.c_int, vpp, C.c_int, C.c_int, C.c_int, C.c_int, vpp, C.c_size_t]
        if hasattr(L, "eqf_tile_gemm_tn_i8"):
            L.eqf_tile_gemm_tn_i8.argtypes = [C.c_int, vpp, vpp, C.c_int, C.c_int, C.c_int, vpp, C.c_int, vpp, C.c_int, C.c_int, C.c_int] + [C.c_int] * 9 + [vpp, C.c_size_t]
        L.eqf_stream_create_masked.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vpp)]
        L.eqf_stream_destroy.argtypes = [C.c_int, vpp]
        # the 2-D block-partitioned filter (BASELINE configs[4]); device buffers are plain pointers (torch tensors' data_ptr)
        L.eqf_tiled_create.argtypes = [C.POINTER(Settings), C.c_int, C.c_int, C.POINTER(vp)]
        L.eqf_tiled_destroy.argtypes = [vp]
        L.eqf_tiled_destroy.restype = None
        L.eqf_tiled_set_stream.argtypes = [vp, vpp]
        L.eqf_tiled_set_geometry.argtypes = [vp, C.c_int, _ip, C.c_int, _ip]
        L.eqf_tiled_propagate.argtypes = [vp, C.c_double, _dp, _dp, C.c_int, vpp, C.c_int]
        L.eqf_tiled_add_landmarks.argtypes = [vp, C.c_int, _dp, vpp, C.c_int]
        L.eqf_tiled_propagate_burst.argtypes = [vp, C.c_int, _dp, _dp, _dp, C.c_int, vpp, C.c_int, _ip]
        L.eqf_tiled_edit_landmarks.argtypes = [vp, C.c_int, _ip, C.c_int, _ip, _dp, C.c_double, C.c_int, vpp, C.c_int]
        L.eqf_tiled_update_prep.argtypes = [vp, _dp, vpp, C.c_int, vpp, C.c_int, vpp, C.c_int, vpp]
        L.eqf_tiled_update_finish.argtypes = [vp, vpp, C.c_int, vpp, vpp]
        L.eqf_tiled_synchronize.argtypes = [vp]
        L.eqf_tiled_num_landmarks.argtypes = [vp]
        L.eqf_tiled_get_time.argtypes = [vp, _dp]
        L.eqf_tiled_device_error.argtypes = [vp]
        L.eqf_tiled_get_state_estimate.argtypes = [vp, _dp, _dp, _dp, _dp]
        L.eqf_tiled_get_origin.argtypes = [vp, _dp, _dp, _dp, _dp]
        L.eqf_tiled_get_group.argtypes = [vp, _dp, _dp, _dp, _dp, _dp]
        L.eqf_tiled_get_bias.argtypes = [vp, _dp]
        L.eqf_tiled_get_last_update.argtypes = [vp, _dp, _dp, _dp]
        L.eqf_tiled_get_integrator.argtypes = [vp, _dp, _dp, _dp, _ip]
        L.eqf_tiled_get_base.argtypes = [vp, _dp, C.c_int]
        L.eqf_tiled_set_state.argtypes = [vp, C.c_int] + [_dp] * 11 + [C.c_int, C.c_double, _dp, _dp, C.c_double, C.c_int]
        _lib = L
    return _lib


def build_info():
    """Which library this process loaded and whether it was built from the sources beside it: sha256 of the .so, the source hash the
    library carries (eqf_build_info) and the same hash recomputed now over csrc/*.hip, csrc/*.hpp, include/*.h, csrc/Makefile."""
    import glob
    import hashlib

    L = lib()
    with open(LIB_PATH, "rb") as f:
        so = hashlib.sha256(f.read()).hexdigest()
    embedded = L.eqf_build_info().decode().split("=", 1)[1]
    csrc = os.path.join(_HERE, "csrc")
    names = sorted([os.path.basename(p) for p in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp"))]
                   + ["../../include/eqf_vio_amd.h", "../../include/eqf_vio_amd_debug.h", "Makefile"])  # (the order of the Makefile's $(sort ...))
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(csrc, n), "rb") as f:
            h.update(f.read())
    now = h.hexdigest()
    return {"library": os.path.relpath(LIB_PATH, os.path.dirname(_HERE)), "so_sha256": so, "src_sha256_in_library": embedded,
            "src_sha256_now": now, "library_matches_sources": embedded == now}


def default_settings():
    s = Settings()
    lib().eqf_settings_default(C.byref(s))
    return s


def settings_from_dict(d):
    """Reference setting names -> eqf_settings; unknown keys raise."""
    s = default_settings()
    for k, v in d.items():
        if k in ("initialAccelBias", "initialOmegaBias", "cameraOffset_x", "cameraOffset_q"):
            arr = getattr(s, k)
            for i, x in enumerate(np.asarray(v, dtype=float)):
                arr[i] = x
        elif hasattr(s, k):
            setattr(s, k, int(bool(v)) if k.startswith("use") or k == "fastRiccati" else float(v))
        else:
            raise AttributeError(k)
    return s


def _p(a):
    return a.ctypes.data_as(_dp)


def _check(rc, what):
    if rc < 0:
        raise EqfError(rc, what)
    return rc


class FilterBatch:
    """A batch of independent EqF filters on one MI355X (one handle, one HIP stream)."""

    def __init__(self, settings, capacity, batch=1, device=0, precision=PRECISION_F64):
        if isinstance(settings, dict):
            settings = settings_from_dict(settings)
        self.settings = settings
        self.B = int(batch)
        self.cap = int(capacity)
        self._h = C.c_void_p()
        _check(lib().eqf_create(C.byref(settings), self.cap, self.B, int(device), int(precision), C.byref(self._h)), "eqf_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().eqf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs
    def process_imu(self, stamps, omega, accel):
        st = np.ascontiguousarray(np.broadcast_to(np.asarray(stamps, dtype=np.float64), (self.B,)))
        w = np.ascontiguousarray(np.broadcast_to(np.asarray(omega, dtype=np.float64), (self.B, 3)))
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(accel, dtype=np.float64), (self.B, 3)))
        status = np.zeros(self.B, dtype=np.int32)
        _check(lib().eqf_process_imu(self._h, _p(st), _p(w), _p(a), status.ctypes.data_as(_ip)), "eqf_process_imu")
        return status

    def process_vision(self, stamps, ids, bearings, nb=None):
        """ids: (B, stride) or (stride,) ascending; bearings: (B, stride, 3) or (stride, 3)."""
        ids = np.asarray(ids, dtype=np.int32)
        y = np.asarray(bearings, dtype=np.float64)
        if ids.ndim == 1:
            ids = np.broadcast_to(ids, (self.B, ids.shape[0]))
        if y.ndim == 2:
            y = np.broadcast_to(y, (self.B,) + y.shape)
        ids = np.ascontiguousarray(ids)
        y = np.ascontiguousarray(y)
        stride = ids.shape[1]
        nbv = np.full(self.B, stride, dtype=np.int32) if nb is None else np.ascontiguousarray(nb, dtype=np.int32)
        st = np.ascontiguousarray(np.broadcast_to(np.asarray(stamps, dtype=np.float64), (self.B,)))
        status = np.zeros(self.B, dtype=np.int32)
        _check(
            lib().eqf_process_vision(self._h, _p(st), nbv.ctypes.data_as(_ip), ids.ctypes.data_as(_ip), _p(y), stride,
                                     status.ctypes.data_as(_ip)),
            "eqf_process_vision",
        )
        return status

    def stream_upload(self, imu, vstamps, ids, bearings):
        """imu (K,B,7) or (K,7); vstamps (F,B) or (F,); ids (nb,); bearings (F,B,nb,3) or (F,nb,3)."""
        imu = np.asarray(imu, dtype=np.float64)
        if imu.ndim == 2:
            imu = np.broadcast_to(imu[:, None, :], (imu.shape[0], self.B, 7))
        vs = np.asarray(vstamps, dtype=np.float64)
        if vs.ndim == 1:
            vs = np.broadcast_to(vs[:, None], (vs.shape[0], self.B))
        y = np.asarray(bearings, dtype=np.float64)
        if y.ndim == 3:
            y = np.broadcast_to(y[:, None], (y.shape[0], self.B) + y.shape[1:])
        imu, vs, y = np.ascontiguousarray(imu), np.ascontiguousarray(vs), np.ascontiguousarray(y)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        _check(
            lib().eqf_stream_upload(self._h, imu.shape[0], _p(imu), vs.shape[0], _p(vs), len(ids), ids.ctypes.data_as(_ip), _p(y)),
            "eqf_stream_upload",
        )

    def stream_imu(self, k):
        return _check(lib().eqf_stream_imu(self._h, int(k)), "eqf_stream_imu")

    def stream_vision(self, f):
        return _check(lib().eqf_stream_vision(self._h, int(f)), "eqf_stream_vision")

    def set_dense_propagate(self, on=True):
        """Riccati step as dense F Sigma F^T on the matrix cores (BASELINE cfg 3 cross-check backend)."""
        _check(lib().eqf_set_dense_propagate(self._h, int(bool(on))), "eqf_set_dense_propagate")

    def set_imu_burst(self, max_steps):
        """IMU calls are queued and launched as bursts of up to `max_steps` steps (0: one launch per call); see
        include/eqf_vio_amd.h."""
        _check(lib().eqf_set_imu_burst(self._h, int(max_steps)), "eqf_set_imu_burst")

    def synchronize(self):
        _check(lib().eqf_synchronize(self._h), "eqf_synchronize")

    def reset(self):
        _check(lib().eqf_reset(self._h), "eqf_reset")

    # ---- outputs
    def num_landmarks(self, b=0):
        return _check(lib().eqf_num_landmarks(self._h, b), "eqf_num_landmarks")

    def get_time(self):
        t = np.zeros(self.B)
        _check(lib().eqf_get_time(self._h, _p(t)), "eqf_get_time")
        return t

    def ids(self, b=0):
        out = np.zeros(self.num_landmarks(b), dtype=np.int32)
        _check(lib().eqf_get_ids(self._h, b, out.ctypes.data_as(_ip)), "eqf_get_ids")
        return out

    def _state(self, fn, b, what):
        N = self.num_landmarks(b)
        q, x, v, p = np.zeros(4), np.zeros(3), np.zeros(3), np.zeros((max(N, 1), 3))
        _check(fn(self._h, b, _p(q), _p(x), _p(v), _p(p)), what)
        return dict(q=q, x=x, v=v, p=p[:N])

    def state_estimate(self, b=0):
        return self._state(lib().eqf_get_state_estimate, b, "eqf_get_state_estimate")

    def origin(self, b=0):
        return self._state(lib().eqf_get_origin, b, "eqf_get_origin")

    def group(self, b=0):
        N = self.num_landmarks(b)
        Aq, Ax, w = np.zeros(4), np.zeros(3), np.zeros(3)
        Qq, Qa = np.zeros((max(N, 1), 4)), np.zeros(max(N, 1))
        _check(lib().eqf_get_group(self._h, b, _p(Aq), _p(Ax), _p(w), _p(Qq), _p(Qa)), "eqf_get_group")
        return dict(Aq=Aq, Ax=Ax, w=w, Qq=Qq[:N], Qa=Qa[:N])

    def bias(self, b=0):
        out = np.zeros(6)
        _check(lib().eqf_get_bias(self._h, b, _p(out)), "eqf_get_bias")
        return out

    def sigma(self, b=0):
        n = 11 + 3 * self.num_landmarks(b)
        out = np.zeros((n, n))
        _check(lib().eqf_get_sigma(self._h, b, _p(out), n), "eqf_get_sigma")
        return out

    def set_sigma(self, S, b=0):
        S = np.ascontiguousarray(S, dtype=np.float64)
        n = 11 + 3 * self.num_landmarks(b)
        assert S.shape == (n, n)
        _check(lib().eqf_set_sigma(self._h, b, _p(S), n), "eqf_set_sigma")

    def dump_state(self, b=0):
        """Lossless snapshot of filter b (everything eqf_set_state needs)."""
        cv, av, at, ini = np.zeros(6), np.zeros(6), np.zeros(1), np.zeros(1, dtype=np.int32)
        _check(lib().eqf_get_integrator(self._h, b, _p(cv), _p(av), _p(at), ini.ctypes.data_as(_ip)), "eqf_get_integrator")
        return dict(ids=self.ids(b), origin=self.origin(b), group=self.group(b), bias=self.bias(b), sigma=self.sigma(b),
                    time=float(self.get_time()[b]), currentVelocity=cv, accumulatedVelocity=av, accumulatedTime=float(at[0]),
                    initialised=int(ini[0]))

    def restore_state(self, st, b=0):
        """Inverse of dump_state (checkpoint / resume, cross-backend state injection)."""
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
        cv = np.ascontiguousarray(st["currentVelocity"], dtype=np.float64)
        av = np.ascontiguousarray(st["accumulatedVelocity"], dtype=np.float64)
        arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (o["q"], o["x"], o["v"], p0, g["Aq"], g["Ax"], g["w"], Qq, Qa, st["bias"], S)]
        _check(
            lib().eqf_set_state(self._h, b, N, ids.ctypes.data_as(_ip), *[_p(a) for a in arrs], S.shape[0], float(st["time"]), _p(cv), _p(av),
                                float(st["accumulatedTime"]), int(st["initialised"])),
            "eqf_set_state",
        )

    def set_camera_offset(self, q, x):
        q, x = np.ascontiguousarray(q, dtype=np.float64), np.ascontiguousarray(x, dtype=np.float64)
        _check(lib().eqf_set_camera_offset(self._h, _p(q), _p(x)), "eqf_set_camera_offset")

    def last_update(self, b=0):
        N = self.num_landmarks(b)
        delta, gamma, Gamma = np.zeros(max(2 * N, 1)), np.zeros(11 + 3 * N), np.zeros(9 + 3 * N)
        _check(lib().eqf_get_last_update(self._h, b, _p(delta), _p(gamma), _p(Gamma)), "eqf_get_last_update")
        return dict(delta=delta[: 2 * N], gamma=gamma, Gamma=Gamma)

    def debug_blocks(self, b=0):
        """Linearisation blocks of the last single-step split-path launch + the C0i blocks (include/eqf_vio_amd.h)."""
        N = self.num_landmarks(b)
        common, rec, c0 = np.zeros(31), np.zeros((max(N, 1), 27)), np.zeros((max(N, 1), 6))
        _check(lib().eqf_debug_get_blocks(self._h, b, _p(common), _p(rec), _p(c0)), "eqf_debug_get_blocks")
        return dict(T=common[0], Bg=common[1:7].reshape(2, 3), Bvw=common[7:16].reshape(3, 3), RA=common[16:25].reshape(3, 3),
                    Avg=common[25:31].reshape(3, 2), D=rec[:N, 0:9].reshape(N, 3, 3), Lw=rec[:N, 9:18].reshape(N, 3, 3),
                    Lv=rec[:N, 18:27].reshape(N, 3, 3), C0=c0[:N].reshape(N, 2, 3))

    def device_error(self):
        return lib().eqf_device_error(self._h)

    def debug_drop_role(self, kind, role=0, R=0, C_=0):
        """Fault injection (tests): one role of the update launch leaves without publishing (kind < 0: off); include/eqf_vio_amd_debug.h."""
        _check(lib().eqf_debug_drop_role(self._h, int(kind), int(role), int(R), int(C_)), "eqf_debug_drop_role")

    def debug_option(self, name, value):
        """Developer toggle by name (include/eqf_vio_amd_debug.h: eqf_debug_option), e.g. ``"cs_in_burst"``."""
        _check(lib().eqf_debug_option(self._h, name.encode(), int(value)), "eqf_debug_option")

    def launch_shape(self):
        """Shape of the most recent IMU burst (include/eqf_vio_amd_debug.h: eqf_debug_launch_shape)."""
        a = (C.c_int * 8)()
        _check(lib().eqf_debug_launch_shape(self._h, a), "eqf_debug_launch_shape")
        return dict(builder_landmarks=a[0], rows_per_wave=a[1], fused=bool(a[2]), cs_out=bool(a[3]), builder_workgroups=a[4],
                    block_workgroups=a[5], steps=a[6])

    # ---- profiling
    def profile_enable(self, on=True):
        _check(lib().eqf_profile_enable(self._h, int(bool(on))), "eqf_profile_enable")

    def profile(self):
        out = {}
        for c in range(PROF_CLASSES):
            n = C.c_longlong()
            ms = C.c_double()
            _check(lib().eqf_profile_get(self._h, c, C.byref(n), C.byref(ms)), "eqf_profile_get")
            out[lib().eqf_profile_class_name(c).decode()] = (n.value, ms.value)
        return out
