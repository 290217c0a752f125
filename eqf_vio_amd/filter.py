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


class VIOFilter:
    """Drop-in for the reference's VIOFilter on one MI355X (batch of one)."""

    def __init__(self, settings, capacity=256, device=0, precision=binding.PRECISION_F64):
        self._fb = binding.FilterBatch(settings, capacity, 1, device, precision)

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
