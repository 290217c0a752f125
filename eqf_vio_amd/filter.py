"""Python mirror of the reference's VIOFilter interface on the MI355X path.

Same member names and argument meaning as eqf_vio/include/eqf_vio/VIOFilter.h:64-88 so that parity tests read
like tests of the reference: VIOFilter(settings), processIMUData, processVisionData, getTime, stateEstimate,
stateCovariance.  All arithmetic happens in the HIP kernels behind the C ABI (no CPU fallback).
"""
from dataclasses import dataclass, field

import numpy as np

from . import binding


@dataclass
class IMUVelocity:
    """eqf_vio/include/eqf_vio/IMUVelocity.h:24-37"""

    stamp: float = 0.0
    omega: np.ndarray = field(default_factory=lambda: np.zeros(3))
    accel: np.ndarray = field(default_factory=lambda: np.zeros(3))


@dataclass
class VisionMeasurement:
    """eqf_vio/include/eqf_vio/VisionMeasurement.h:24-28: bearings sorted by ascending id."""

    stamp: float = 0.0
    ids: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    bearings: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))

    @property
    def numberOfBearings(self):
        return len(self.ids)


@dataclass
class VIOState:
    """eqf_vio/include/eqf_vio/VIOState.h:51-60 with the pose as (quaternion wxyz, position)."""

    pose_q: np.ndarray
    pose_x: np.ndarray
    velocity: np.ndarray
    bodyLandmarks: np.ndarray  # (N, 3)
    ids: np.ndarray


@dataclass
class AuxiliaryFilterData:
    """eqf_vio/include/eqf_vio/VIOFilter.h:30-39 (the fields the filter reads: VIOFilter.cpp:74-82)."""

    initialAttitude: np.ndarray = field(default_factory=lambda: np.array([1.0, 0.0, 0.0, 0.0]))  # quaternion wxyz
    initialPosition: np.ndarray = field(default_factory=lambda: np.zeros(3))
    initialTime: float = 0.0
    cameraOffset_q: np.ndarray = field(default_factory=lambda: np.array([1.0, 0.0, 0.0, 0.0]))
    cameraOffset_x: np.ndarray = field(default_factory=lambda: np.zeros(3))


def _qmul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
        a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1],
    ])


def _qrot(q, v):
    u, w = np.asarray(q[1:]), q[0]
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


class VIOFilter:
    """Drop-in for the reference's VIOFilter on one MI355X (batch of one)."""

    def __init__(self, settings, capacity=256, device=0, precision=binding.PRECISION_F64, auxiliaryData=None):
        self._fb = binding.FilterBatch(settings, capacity, 1, device, precision)
        self._settings = self._fb.settings
        if auxiliaryData is not None:  # VIOFilter(AuxiliaryFilterData, Settings), VIOFilter.cpp:51-58
            self.setAuxiliaryData(auxiliaryData)

    def setAuxiliaryData(self, aux):
        """VIOFilter.cpp:74-82: origin pose from the given attitude/position, zero velocity, camera offset replaced,
        filter marked initialised (no gravity alignment at the first IMU sample)."""
        st = self._fb.dump_state(0)
        st["origin"]["q"] = np.asarray(aux.initialAttitude, dtype=np.float64)
        st["origin"]["x"] = np.asarray(aux.initialPosition, dtype=np.float64)
        st["origin"]["v"] = np.zeros(3)
        st["initialised"] = 1
        self._fb.set_camera_offset(aux.cameraOffset_q, aux.cameraOffset_x)
        self._cam = (np.asarray(aux.cameraOffset_q, dtype=np.float64), np.asarray(aux.cameraOffset_x, dtype=np.float64))
        self._fb.restore_state(st, 0)

    def setInertialPoints(self, points, ids):
        """VIOFilter.cpp:93-118: replace the landmark set by points given in the inertial frame (ids as given);
        Q_i = identity, Sigma = initialPointVariance * I outside the 11x11 base block."""
        points = np.asarray(points, dtype=np.float64).reshape(-1, 3)
        ids = np.asarray(ids, dtype=np.int32)
        N = len(ids)
        st = self._fb.dump_state(0)
        cq, cx = getattr(self, "_cam", (np.asarray(self._settings.cameraOffset_q[:]), np.asarray(self._settings.cameraOffset_x[:])))
        pq, px = st["origin"]["q"], st["origin"]["x"]
        tq, tx = _qmul(pq, cq), px + _qrot(pq, cx)  # xi0.pose * xi0.cameraOffset
        tqi = np.array([tq[0], -tq[1], -tq[2], -tq[3]])
        st["origin"]["p"] = np.array([_qrot(tqi, p - tx) for p in points]).reshape(-1, 3)
        st["ids"] = ids
        st["group"]["Qq"] = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (N, 1))
        st["group"]["Qa"] = np.ones(N)
        n = 11 + 3 * N
        S = np.eye(n) * float(self._settings.initialPointVariance)
        S[:11, :11] = st["sigma"][:11, :11]
        st["sigma"] = S
        self._fb.restore_state(st, 0)

    def processIMUData(self, imu):
        """VIOFilter.cpp:120-131"""
        return int(self._fb.process_imu([imu.stamp], imu.omega, imu.accel)[0])

    def processVisionData(self, meas):
        """VIOFilter.cpp:232-302"""
        return int(self._fb.process_vision([meas.stamp], meas.ids, meas.bearings)[0])

    def getTime(self):
        return float(self._fb.get_time()[0])

    def stateEstimate(self):
        s = self._fb.state_estimate(0)
        return VIOState(s["q"], s["x"], s["v"], s["p"], self._fb.ids(0))

    def stateCovariance(self):
        return self._fb.sigma(0)

    def reset(self):
        self._fb.reset()

    @property
    def batch(self):
        return self._fb
