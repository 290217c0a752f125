"""Deterministic synthetic IMU + bearing stream.

IMU @200 Hz = exact body angular rate and specific force of an analytic trajectory + white noise +
constant bias; bearings @20 Hz of N world-fixed landmarks seen from the EuRoC cam0 extrinsics.
Sign conventions follow the reference's system model (eqf_vio/src/VIOState.cpp:26-56):
    R' = R w^x,  p' = R v,  v' = -w x v + a - g R^T e3     (v body-frame velocity, e3 "up")
so the accelerometer reads  a = R^T (p'' + g e3).
The filter itself models no field of view; landmarks are drawn in a +-35 deg cone about the camera's
optical axis so they stay in front of the camera for the whole run.
"""
from dataclasses import dataclass

import numpy as np

GRAVITY_CONSTANT = 9.81  # eqf_vio/include/eqf_vio/IMUVelocity.h:22

# EuRoC cam0 extrinsics, eqf_vio/EQVIO_config_template.yaml:21-29 ("xw": x, then quaternion w x y z)
CAM_OFFSET_X = np.array([-0.0216401454975, -0.064676986768, 0.00981073058949])
CAM_OFFSET_Q = np.array([0.7123014606690344, -0.007707179755538301, 0.010499323370588468, 0.7017528002920512])


def _quat_to_matrix(q):
    w, x, y, z = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ]
    )


# Ry(-90 deg): maps body x to world z
R_BASE = np.array([[0.0, 0.0, -1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0]])


@dataclass
class SyntheticStream:
    imu: np.ndarray  # (K, 7): stamp, wx, wy, wz, ax, ay, az
    vision_stamps: np.ndarray  # (F,)
    ids: np.ndarray  # (N,) ascending int32
    bearings: np.ndarray  # (F, N, 3) unit vectors, camera frame
    true_pos: np.ndarray  # (F, 3) true position at the vision stamps
    true_att: np.ndarray  # (F, 3, 3) true attitude at the vision stamps
    landmarks_world: np.ndarray  # (N, 3)

    def events(self):
        """Event order of the reference's offline runner (eqf_vio/src/main.cpp:111-170):
        IMU first while imu.stamp < meas.stamp, otherwise vision.  Yields ("imu", k) / ("vision", f)."""
        k, f = 0, 0
        K, F = len(self.imu), len(self.vision_stamps)
        while k < K and f < F:
            if self.imu[k, 0] < self.vision_stamps[f]:
                yield ("imu", k)
                k += 1
            else:
                yield ("vision", f)
                f += 1


def _trajectory(t):
    """Position p(t), its derivatives, attitude R(t) and body rate w(t); all analytic."""
    t = np.asarray(t, dtype=float)
    amp = np.array([0.5, 0.5, 0.3])
    frq = 2 * np.pi * np.array([0.2, 0.3, 0.4])
    # sin^3 profiles: position, velocity and acceleration all start at zero (the vehicle starts at
    # rest, like the EuRoC sequences), yet stay smooth, bounded and analytic.
    sn, cs = np.sin(frq * t[:, None]), np.cos(frq * t[:, None])
    p = amp * sn**3
    pd = amp * frq * 3 * sn**2 * cs
    pdd = amp * frq * frq * (6 * sn * cs**2 - 3 * sn**3)
    # attitude: R = Rz(c) Ry(b) Rx(a), each +-10 deg
    aamp = np.deg2rad(10.0)
    afrq = 2 * np.pi * np.array([0.25, 0.35, 0.15])
    sa_, ca_ = np.sin(afrq * t[:, None]), np.cos(afrq * t[:, None])
    ang = aamp * sa_**3
    angd = aamp * afrq * 3 * sa_**2 * ca_
    a, b, c = ang[:, 0], ang[:, 1], ang[:, 2]
    ad, bd, cd = angd[:, 0], angd[:, 1], angd[:, 2]
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    R = np.empty((len(t), 3, 3))
    R[:, 0, 0] = cc * cb
    R[:, 0, 1] = cc * sb * sa - sc * ca
    R[:, 0, 2] = cc * sb * ca + sc * sa
    R[:, 1, 0] = sc * cb
    R[:, 1, 1] = sc * sb * sa + cc * ca
    R[:, 1, 2] = sc * sb * ca - cc * sa
    R[:, 2, 0] = -sb
    R[:, 2, 1] = cb * sa
    R[:, 2, 2] = cb * ca
    # body angular rate for ZYX Euler angles (roll a, pitch b, yaw c)
    w = np.stack([ad - cd * sb, bd * ca + cd * cb * sa, -bd * sa + cd * cb * ca], axis=1)
    # constant base attitude: body x "up" (as on the EuRoC vehicle).  The reference's gravity chart
    # stereoSphereChart(., pole = R0^T e3) is singular when the initial body z axis is exactly up
    # (SO3FromVectors(-e3, e3) throws, libs/core/src/SO3.cpp:160), so a level start is avoided.
    R = R @ R_BASE
    w = w @ R_BASE
    return p, pd, pdd, R, w


def make_stream(N, seed=1234, duration=11.0, imu_rate=200.0, cam_rate=20.0, imu_noise_var=1e-4,
                gyro_bias=0.01, accel_bias=0.05, bearing_noise=1e-3):
    """Build the stream.  duration includes the 1 s warm-up of SURVEY.md section 8d."""
    rng = np.random.default_rng(seed)
    K = int(round(duration * imu_rate))
    t_imu = np.arange(K) / imu_rate
    p, pd, pdd, R, w = _trajectory(t_imu)
    g_e3 = np.array([0.0, 0.0, GRAVITY_CONSTANT])
    acc = np.einsum("kji,kj->ki", R, pdd + g_e3)  # R^T (p'' + g e3)
    sd = np.sqrt(imu_noise_var)
    imu = np.empty((K, 7))
    imu[:, 0] = t_imu
    imu[:, 1:4] = w + gyro_bias + sd * rng.standard_normal((K, 3))
    imu[:, 4:7] = acc + accel_bias + sd * rng.standard_normal((K, 3))

    F = int(np.floor((duration - 0.0025) * cam_rate))
    t_cam = np.arange(F) / cam_rate + 0.0025  # +2.5 ms so dt > 0 at every vision call
    pc, _, _, Rc, _ = _trajectory(t_cam)

    R_IC = _quat_to_matrix(CAM_OFFSET_Q)
    x_IC = CAM_OFFSET_X
    # landmarks: cone about the camera optical axis at t = 0 (pose = identity at t = 0)
    lm_cam = np.empty((N, 3))
    i = 0
    while i < N:
        ang = np.deg2rad(35.0) * np.sqrt(rng.uniform())
        az = rng.uniform(0, 2 * np.pi)
        d = np.array([np.sin(ang) * np.cos(az), np.sin(ang) * np.sin(az), np.cos(ang)])
        if d[2] >= 1 - 1e-6:  # chart pole of SO3FromVectors(-y, e3), libs/core/src/SO3.cpp:159-161
            continue
        lm_cam[i] = d * rng.uniform(2.0, 8.0)
        i += 1
    p0, _, _, R0, _ = _trajectory(np.array([0.0]))
    lm_world = (R0[0] @ (R_IC @ lm_cam.T + x_IC[:, None])).T + p0[0]

    bearings = np.empty((F, N, 3))
    for f in range(F):
        body = (Rc[f].T @ (lm_world - pc[f]).T).T
        cam = (R_IC.T @ (body - x_IC).T).T
        y = cam / np.linalg.norm(cam, axis=1, keepdims=True)
        # isotropic tangent-plane perturbation, renormalised
        nz = bearing_noise * rng.standard_normal((N, 3))
        nz -= np.sum(nz * y, axis=1, keepdims=True) * y
        y = y + nz
        bearings[f] = y / np.linalg.norm(y, axis=1, keepdims=True)
    assert np.all(bearings[:, :, 2] > 0.0)
    return SyntheticStream(imu, t_cam, np.arange(N, dtype=np.int32), bearings, pc, Rc, lm_world)


def template_settings_dict():
    """Filter tunables of eqf_vio/EQVIO_config_template.yaml:1-29 with the bench overrides of
    SURVEY.md section 8d (fastRiccati=false is already the template value; outlierThreshold raised)."""
    return dict(
        initialGravityVariance=1.0,
        initialVelocityVariance=1.0,
        initialPointVariance=5000.0,
        biasOmegaProcessVariance=0.0001,
        biasAccelProcessVariance=0.0001,
        gravityProcessVariance=0.01,
        velocityProcessVariance=0.1,
        pointProcessVariance=0.001,
        measurementVariance=0.003,
        velOmegaVariance=0.0001,
        velAccelVariance=0.0001,
        initialBiasOmegaVariance=1.0,
        initialBiasAccelVariance=1.0,
        initialSceneDepth=1.0,
        outlierThreshold=1e9,
        fastRiccati=False,
        useInnovationLift=True,
        useDiscreteInnovationLift=True,
        useDiscreteVelocityLift=True,
        cameraOffset_x=CAM_OFFSET_X.copy(),
        cameraOffset_q=CAM_OFFSET_Q.copy(),
    )


def churn_measurements(stream, seed=7, max_visible=None, outlier_frames=(), outlier_angle=0.05):
    """Per-frame (ids, bearings) with landmarks entering and leaving the field of view, to exercise the
    reference's landmark bookkeeping (eqf_vio/src/VIOFilter.cpp:345-443).  Landmark j is visible on a random
    window of frames; ids stay sorted ascending (VIOFilter.cpp:239-240).  On `outlier_frames` one visible
    bearing is rotated by `outlier_angle` rad (caught by removeOutliers when the threshold is the default)."""
    rng = np.random.default_rng(seed)
    F, N = stream.bearings.shape[0], stream.bearings.shape[1]
    start = rng.integers(0, max(1, F // 2), size=N)
    length = rng.integers(max(2, F // 4), F, size=N)
    start[: max(2, N // 3)] = 0  # some landmarks are there from the first frame
    length[:3] = F  # ... and three stay for the whole run: with fewer than two landmarks bundleLift's 4x4
    # normal equations (EqFMatrices.cpp:240-242) are rank deficient and any solver's answer is arbitrary
    out = []
    for f in range(F):
        vis = np.where((start <= f) & (f < start + length))[0]
        if max_visible is not None and len(vis) > max_visible:
            vis = np.sort(rng.choice(vis, size=max_visible, replace=False))
        y = stream.bearings[f, vis].copy()
        if f in outlier_frames and len(vis) > 0:
            k = int(rng.integers(0, len(vis)))
            axis = np.cross(y[k], np.array([1.0, 0.3, -0.2]))
            axis /= np.linalg.norm(axis)
            y[k] = y[k] * np.cos(outlier_angle) + np.cross(axis, y[k]) * np.sin(outlier_angle)
            y[k] /= np.linalg.norm(y[k])
        out.append((stream.ids[vis].astype(np.int32), y))
    return out
