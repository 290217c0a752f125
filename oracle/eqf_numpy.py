"""TEST INFRASTRUCTURE ONLY -- fp64 numpy restatement of the eqf_vio EqF propagate/update hot path.

This file is the *checker*, never the product: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  It restates, in own code, the algorithm of
pvangoor/eqf_vio (reference paths below are relative to /root/reference/eqf_vio):

    libs/core/src/SO3.cpp, SE3.cpp, SOT3.cpp   Lie groups (quaternion backed)
    src/VIOState.cpp                           charts, measurement, projectToManifold
    src/VIOGroup.cpp                           group product/inverse/actions, velocity lifts
    src/VisionMeasurement.cpp                  output chart
    src/EqFMatrices.cpp                        A0, B, C0, innovation lifts, bundleLift
    src/VIOFilter.cpp                          the filter (processIMUData / processVisionData)

PARITY UNPINNED at the filter level: the reference cannot be built here (Eigen 3 and yaml-cpp are
absent) and its own tests hold no golden vector for VIOFilter.  What pins this
restatement instead: (1) the reference's property tests (eqf_vio/test/*.cpp) restated in
tests/test_oracle_properties.py, (2) agreement to ~1e-12 with the independently written C++
restatement oracle/eqf_oracle.cpp, (3) line-by-line following of VIOFilter.cpp.

Third-party arithmetic: the reference delegates quaternion<->matrix conversion, quaternion products,
LU inverse and Householder QR to Eigen 3 (version unpinned, eqf_vio/CMakeLists.txt:12).  The
quaternion formulas below restate Eigen's published algorithms (Eigen/src/Geometry/Quaternion.h:
toRotationMatrix, quaternionbase_assign_impl<Other,3,3>, _transformVector, inverse, operator*).

Conventions: quaternions are arrays [w, x, y, z]; SE3 = (q, x); SOT3 = (q, a).
"""
import numpy as np

GRAVITY_CONSTANT = 9.81  # include/eqf_vio/IMUVelocity.h:22
SIGMA_BASE_SIZE = 11  # include/eqf_vio/VIOFilter.h:28
E3 = np.array([0.0, 0.0, 1.0])


# ----------------------------------------------------------------------------------------------
# Quaternions (Eigen semantics) and SO3 -- libs/core/src/SO3.cpp
# ----------------------------------------------------------------------------------------------
def quat_identity():
    return np.array([1.0, 0.0, 0.0, 0.0])


def quat_to_matrix(q):
    """Eigen QuaternionBase::toRotationMatrix (used by SO3::asMatrix, SO3.cpp:94)."""
    w, x, y, z = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array(
        [
            [1 - (tyy + tzz), txy - twz, txz + twy],
            [txy + twz, 1 - (txx + tzz), tyz - twx],
            [txz - twy, tyz + twx, 1 - (txx + tyy)],
        ]
    )


def quat_from_matrix(m):
    """Eigen matrix->quaternion (used by SO3::fromMatrix, SO3.cpp:100)."""
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1] = (m[2, 1] - m[1, 2]) * t
        q[2] = (m[0, 2] - m[2, 0]) * t
        q[3] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[1 + i] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[k, j] - m[j, k]) * t
        q[1 + j] = (m[j, i] + m[i, j]) * t
        q[1 + k] = (m[k, i] + m[i, k]) * t
    return q


def quat_mul(a, b):
    """Eigen quaternion product (SO3.cpp:76)."""
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by + ay * bw + az * bx - ax * bz,
            aw * bz + az * bw + ax * by - ay * bx,
        ]
    )


def quat_inverse(q):
    """Eigen QuaternionBase::inverse = conjugate / squaredNorm (SO3.cpp:82)."""
    n2 = float(q @ q)
    return np.array([q[0], -q[1], -q[2], -q[3]]) / n2


def quat_rotate(q, v):
    """Eigen _transformVector (SO3.cpp:66)."""
    u = q[1:4]
    uv = np.cross(u, v)
    uv = uv + uv
    return v + q[0] * uv + np.cross(u, uv)


def skew(v):
    """SO3.cpp:110-114."""
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def vex(O):
    """SO3.cpp:116-120."""
    return np.array([O[2, 1], O[0, 2], O[1, 0]])


def so3_exp(w):
    """SO3.cpp:122-140 -> quaternion."""
    th = np.linalg.norm(w)
    if abs(th) >= 1e-8:
        A = np.sin(th) / th
        B = (1 - np.cos(th)) / th**2
    else:
        A, B = 1.0, 0.5
    wx = skew(w)
    return quat_from_matrix(np.eye(3) + A * wx + B * wx @ wx)


def so3_log(q):
    """SO3.cpp:142-153."""
    R = quat_to_matrix(q)
    theta = np.arccos((np.trace(R) - 1.0) / 2.0)
    coefficient = 0.5
    if abs(theta) >= 1e-6:
        coefficient = theta / (2.0 * np.sin(theta))
    return vex(coefficient * (R - R.T))


class AntipodalError(ValueError):
    """std::domain_error of SO3FromVectors (SO3.cpp:160-161)."""


def so3_from_vectors_matrix(origin, dest):
    """SO3.cpp:155-161, the rotation matrix before it is re-stored as a quaternion."""
    o = origin / np.linalg.norm(origin)
    d = dest / np.linalg.norm(dest)
    v = np.cross(o, d)
    c = float(o @ d)
    if abs(1 + c) <= 1e-8:
        raise AntipodalError("The vectors cannot be exactly opposing.")
    vx = skew(v)
    return np.eye(3) + (vx + 1.0 / (1.0 + c) * vx @ vx)


def so3_from_vectors(origin, dest):
    """SO3.cpp:155-167 -> quaternion."""
    return quat_from_matrix(so3_from_vectors_matrix(origin, dest))


# ----------------------------------------------------------------------------------------------
# SE3 = (q, x) -- libs/core/src/SE3.cpp ; SOT3 = (q, a) -- libs/core/src/SOT3.cpp
# ----------------------------------------------------------------------------------------------
class SE3:
    __slots__ = ("q", "x")

    def __init__(self, q=None, x=None):
        self.q = quat_identity() if q is None else np.array(q, dtype=float)
        self.x = np.zeros(3) if x is None else np.array(x, dtype=float)

    def copy(self):
        return SE3(self.q.copy(), self.x.copy())

    def R(self):
        return quat_to_matrix(self.q)

    def apply(self, p):  # SE3.cpp:63
        return quat_rotate(self.q, p) + self.x

    def __mul__(self, other):  # SE3.cpp:71-76
        return SE3(quat_mul(self.q, other.q), self.x + quat_rotate(self.q, other.x))

    def inverse(self):  # SE3.cpp:80-83 (R inverted first, then x = -(R^-1 x))
        qi = quat_inverse(self.q)
        return SE3(qi, -quat_rotate(qi, self.x))

    def adjoint(self):  # SE3.cpp:95-103
        R = self.R()
        Ad = np.zeros((6, 6))
        Ad[0:3, 0:3] = R
        Ad[3:6, 0:3] = skew(self.x) @ R
        Ad[3:6, 3:6] = R
        return Ad


def se3_exp(u):
    """SE3.cpp:139-164."""
    w = u[0:3]
    v = u[3:6]
    th = np.linalg.norm(w)
    if abs(th) >= 1e-12:
        A = np.sin(th) / th
        B = (1 - np.cos(th)) / th**2
        C = (1 - A) / th**2
    else:
        A, B, C = 1.0, 0.5, 1.0 / 6.0
    wx = skew(w)
    R = np.eye(3) + A * wx + B * wx @ wx
    V = np.eye(3) + B * wx + C * wx @ wx
    return SE3(quat_from_matrix(R), V @ v)


def se3_log(P):
    """SE3.cpp:166-187."""
    Omega = skew(so3_log(P.q))
    theta = np.linalg.norm(vex(Omega))
    coefficient = 1.0 / 12.0
    if abs(theta) > 1e-8:
        coefficient = 1 / (theta * theta) * (1 - (theta * np.sin(theta)) / (2 * (1 - np.cos(theta))))
    VInv = np.eye(3) - 0.5 * Omega + coefficient * Omega @ Omega
    return np.concatenate([vex(Omega), VInv @ P.x])


class SOT3:
    __slots__ = ("q", "a")

    def __init__(self, q=None, a=1.0):
        self.q = quat_identity() if q is None else np.array(q, dtype=float)
        self.a = float(a)

    def copy(self):
        return SOT3(self.q.copy(), self.a)

    def apply(self, p):  # SOT3.cpp:69
        return self.a * quat_rotate(self.q, p)

    def __mul__(self, other):  # SOT3.cpp:77-82
        return SOT3(quat_mul(self.q, other.q), self.a * other.a)

    def inverse(self):  # SOT3.cpp:86-89
        return SOT3(quat_inverse(self.q), 1.0 / self.a)

    def as_matrix3(self):  # SOT3.cpp:107-110
        return self.a * quat_to_matrix(self.q)

    def apply_inverse(self, p):  # SOT3.cpp:121
        return (1.0 / self.a) * quat_rotate(quat_inverse(self.q), p)


def sot3_exp(w):  # SOT3.cpp:127-132
    return SOT3(so3_exp(w[0:3]), np.exp(w[3]))


def sot3_log(T):  # SOT3.cpp:134-139
    return np.concatenate([so3_log(T.q), [np.log(T.a)]])


# ----------------------------------------------------------------------------------------------
# States, measurements -- include/eqf_vio/VIOState.h, IMUVelocity.h, VisionMeasurement.h
# ----------------------------------------------------------------------------------------------
class VIOState:
    """VIOState.h:51-60.  landmarks: (N,3) float array p, ids: (N,) int array."""

    def __init__(self, pose=None, velocity=None, p=None, ids=None, cameraOffset=None):
        self.pose = SE3() if pose is None else pose
        self.velocity = np.zeros(3) if velocity is None else np.array(velocity, dtype=float)
        self.p = np.zeros((0, 3)) if p is None else np.array(p, dtype=float).reshape(-1, 3)
        self.ids = np.zeros(0, dtype=np.int64) if ids is None else np.array(ids, dtype=np.int64)
        self.cameraOffset = SE3() if cameraOffset is None else cameraOffset

    def copy(self):
        return VIOState(self.pose.copy(), self.velocity.copy(), self.p.copy(), self.ids.copy(), self.cameraOffset.copy())


class VIOManifoldState:
    """VIOState.h:43-49."""

    def __init__(self, gravityDir, velocity, p, ids, cameraOffset):
        self.gravityDir = np.array(gravityDir, dtype=float)
        self.velocity = np.array(velocity, dtype=float)
        self.p = np.array(p, dtype=float).reshape(-1, 3)
        self.ids = np.array(ids, dtype=np.int64)
        self.cameraOffset = cameraOffset


def project_to_manifold(Xi):
    """VIOState.cpp:88-95."""
    if isinstance(Xi, VIOManifoldState):
        return Xi
    return VIOManifoldState(quat_rotate(quat_inverse(Xi.pose.q), E3), Xi.velocity, Xi.p, Xi.ids, Xi.cameraOffset)


class IMUVelocity:
    """IMUVelocity.h:24-37, IMUVelocity.cpp:20-58."""

    def __init__(self, stamp=0.0, omega=None, accel=None):
        self.stamp = float(stamp)
        self.omega = np.zeros(3) if omega is None else np.array(omega, dtype=float)
        self.accel = np.zeros(3) if accel is None else np.array(accel, dtype=float)

    def __add__(self, other):
        if isinstance(other, IMUVelocity):  # IMUVelocity.cpp:35-41
            return IMUVelocity(self.stamp if self.stamp > 0 else other.stamp, self.omega + other.omega, self.accel + other.accel)
        return IMUVelocity(self.stamp, self.omega + other[0:3], self.accel + other[3:6])  # :43-49

    def __sub__(self, vec):  # :51
        return self + (-np.asarray(vec))

    def __mul__(self, c):  # :53-58
        return IMUVelocity(self.stamp, self.omega * c, self.accel * c)


class VIOGroup:
    """VIOGroup.h:24-33.  Q: list of SOT3."""

    def __init__(self, A=None, w=None, Q=None, ids=None):
        self.A = SE3() if A is None else A
        self.w = np.zeros(3) if w is None else np.array(w, dtype=float)
        self.Q = [] if Q is None else Q
        self.ids = np.zeros(0, dtype=np.int64) if ids is None else np.array(ids, dtype=np.int64)

    @staticmethod
    def identity(ids=()):  # VIOGroup.cpp:112-122
        return VIOGroup(SE3(), np.zeros(3), [SOT3() for _ in ids], np.array(ids, dtype=np.int64))

    def copy(self):
        return VIOGroup(self.A.copy(), self.w.copy(), [Qi.copy() for Qi in self.Q], self.ids.copy())

    def __mul__(self, other):  # VIOGroup.cpp:92-110
        assert len(self.Q) == len(other.Q)
        return VIOGroup(
            self.A * other.A,
            self.w + quat_rotate(self.A.q, other.w),
            [a * b for a, b in zip(self.Q, other.Q)],
            self.ids.copy(),
        )

    def inverse(self):  # VIOGroup.cpp:124-134
        Ai = self.A.inverse()
        return VIOGroup(Ai, -quat_rotate(quat_inverse(self.A.q), self.w), [Qi.inverse() for Qi in self.Q], self.ids.copy())


class VIOAlgebra:
    """VIOGroup.h:35-45.  W: (N,4)."""

    def __init__(self, U=None, u=None, W=None, ids=None):
        self.U = np.zeros(6) if U is None else np.array(U, dtype=float)
        self.u = np.zeros(3) if u is None else np.array(u, dtype=float)
        self.W = np.zeros((0, 4)) if W is None else np.array(W, dtype=float).reshape(-1, 4)
        self.ids = np.zeros(0, dtype=np.int64) if ids is None else np.array(ids, dtype=np.int64)

    def __mul__(self, c):
        return VIOAlgebra(self.U * c, self.u * c, self.W * c, self.ids)

    __rmul__ = __mul__

    def __neg__(self):
        return VIOAlgebra(-self.U, -self.u, -self.W, self.ids)

    def __add__(self, o):
        return VIOAlgebra(self.U + o.U, self.u + o.u, self.W + o.W, self.ids)

    def __sub__(self, o):
        return self + (-o)


# ----------------------------------------------------------------------------------------------
# Sphere charts -- src/VIOState.cpp:199-251
# ----------------------------------------------------------------------------------------------
def e3_project_sphere(eta):  # :199-204
    return (eta - E3)[0:2] / (1 - eta[2])


def e3_project_sphere_inv(y):  # :206-211
    ybar = np.array([y[0], y[1], 0.0])
    return E3 + 2.0 / (ybar @ ybar + 1) * (ybar - E3)


def e3_project_sphere_diff(eta):  # :213-220
    I23 = np.eye(3)[0:2, :]
    D = I23 @ (np.eye(3) * (1 - eta[2]) + np.outer(eta - E3, E3))
    return (1 - eta[2]) ** -2.0 * D


def e3_project_sphere_inv_diff(y):  # :222-228
    D = np.zeros((3, 2))
    n2 = float(y @ y)
    D[0:2, :] = np.eye(2) * (n2 + 1.0) - 2 * np.outer(y, y)
    D[2, :] = 2 * y
    return 2.0 * (n2 + 1.0) ** -2.0 * D


def _sphere_rot(pole):
    """SO3FromVectors(-pole, e3) as the quaternion the reference stores (VIOState.cpp:231)."""
    return so3_from_vectors(-pole, E3)


def stereo_sphere_chart(eta, pole):  # :230-234
    return e3_project_sphere(quat_rotate(_sphere_rot(pole), eta))


def stereo_sphere_chart_inv(y, pole):  # :236-240
    return quat_rotate(quat_inverse(_sphere_rot(pole)), e3_project_sphere_inv(y))


def stereo_sphere_chart_diff(eta, pole):  # :242-246
    q = _sphere_rot(pole)
    return e3_project_sphere_diff(quat_rotate(q, eta)) @ quat_to_matrix(q)


def stereo_sphere_chart_inv_diff(y, pole):  # :248-251
    q = _sphere_rot(pole)
    return quat_to_matrix(quat_inverse(q)) @ e3_project_sphere_inv_diff(y)


# ----------------------------------------------------------------------------------------------
# State charts, measurement, system integrator -- src/VIOState.cpp
# ----------------------------------------------------------------------------------------------
def euclid_coordinate_chart(xi, xi0):  # :97-111
    N = len(xi0.ids)
    eps = np.zeros(5 + 3 * N)
    eps[0:2] = stereo_sphere_chart(xi.gravityDir, xi0.gravityDir)
    eps[2:5] = xi.velocity - xi0.velocity
    eps[5:] = (xi.p - xi0.p).reshape(-1)
    return eps


def euclid_coordinate_chart_inv(eps, xi0):  # :113-128
    return VIOManifoldState(
        stereo_sphere_chart_inv(eps[0:2], xi0.gravityDir),
        xi0.velocity + eps[2:5],
        xi0.p + eps[5:].reshape(-1, 3),
        xi0.ids,
        xi0.cameraOffset,
    )


def measure_system_state(state):  # :58-70  -> (N,3) unit bearings
    return state.p / np.linalg.norm(state.p, axis=1, keepdims=True)


def integrate_system_function(state, velocity, dt):  # :26-56 (test oracle only)
    poseVel = np.concatenate([velocity.omega, state.velocity])
    new = VIOState()
    new.pose = state.pose * se3_exp(dt * poseVel)
    new.velocity = state.velocity + dt * (
        -skew(velocity.omega) @ state.velocity
        + velocity.accel
        - quat_rotate(quat_inverse(state.pose.q), np.array([0, 0, GRAVITY_CONSTANT]))
    )
    U_C = state.cameraOffset.inverse().adjoint() @ poseVel  # == vee(T_IC^-1 wedge(U) T_IC)
    camInv = se3_exp(-dt * U_C)
    new.p = np.array([camInv.apply(p) for p in state.p]).reshape(-1, 3)
    new.ids = state.ids.copy()
    new.cameraOffset = state.cameraOffset.copy()
    return new


def output_coordinate_chart(y, y0):  # src/VisionMeasurement.cpp:24-34 ; y, y0 (N,3)
    delta = np.zeros(2 * len(y))
    for i in range(len(y)):
        delta[2 * i : 2 * i + 2] = stereo_sphere_chart(y[i], y0[i])
    return delta


def output_coordinate_chart_inv(delta, y0):  # :36-50
    return np.array([stereo_sphere_chart_inv(delta[2 * i : 2 * i + 2], y0[i]) for i in range(len(y0))])


# ----------------------------------------------------------------------------------------------
# Group actions and lifts -- src/VIOGroup.cpp
# ----------------------------------------------------------------------------------------------
def state_group_action(X, state):
    """VIOGroup.cpp:23-69 (both overloads)."""
    assert len(X.Q) == len(state.ids)
    RAinv = quat_inverse(X.A.q)
    newp = np.array([Q.inverse().apply(p) for Q, p in zip(X.Q, state.p)]).reshape(-1, 3)
    vel = quat_rotate(RAinv, state.velocity - X.w)
    if isinstance(state, VIOManifoldState):
        return VIOManifoldState(quat_rotate(RAinv, state.gravityDir), vel, newp, state.ids, state.cameraOffset)
    return VIOState(state.pose * X.A, vel, newp, state.ids.copy(), state.cameraOffset.copy())


def output_group_action(X, y):
    """VIOGroup.cpp:71-90: y_i -> Q_i.R()^-1 y_i."""
    return np.array([quat_rotate(quat_inverse(Q.q), yi) for Q, yi in zip(X.Q, y)]).reshape(-1, 3)


def lift_velocity(state, velocity):
    """VIOGroup.cpp:178-207."""
    state = project_to_manifold(state)
    U = np.concatenate([velocity.omega, state.velocity])
    u = -velocity.accel + state.gravityDir * GRAVITY_CONSTANT
    U_C = state.cameraOffset.inverse().adjoint() @ U
    omega_C, v_C = U_C[0:3], U_C[3:6]
    W = np.zeros((len(state.ids), 4))
    for i, p in enumerate(state.p):
        n2 = float(p @ p)
        W[i, 0:3] = omega_C + skew(p) @ v_C / n2
        W[i, 3] = float(p @ v_C) / n2
    return VIOAlgebra(U, u, W, state.ids)


def lift_velocity_discrete(state, velocity, dt):
    """VIOGroup.cpp:209-243."""
    state = project_to_manifold(state)
    AVel = np.concatenate([velocity.omega, state.velocity])
    A = se3_exp(dt * AVel)
    w = state.velocity - quat_rotate(
        A.q,
        state.velocity
        + dt * (-skew(velocity.omega) @ state.velocity + velocity.accel - state.gravityDir * GRAVITY_CONSTANT),
    )
    U_C = state.cameraOffset.inverse().adjoint() @ AVel
    camInv = se3_exp(-dt * U_C)
    Q = []
    for p0 in state.p:
        p1 = camInv.apply(p0)
        Q.append(SOT3(so3_from_vectors(p1 / np.linalg.norm(p1), p0 / np.linalg.norm(p0)), np.linalg.norm(p0) / np.linalg.norm(p1)))
    return VIOGroup(A, w, Q, state.ids)


def vio_exp(lam):
    """VIOGroup.cpp:245-255."""
    return VIOGroup(se3_exp(lam.U), lam.u.copy(), [sot3_exp(Wi) for Wi in lam.W], lam.ids)


# ----------------------------------------------------------------------------------------------
# EqF matrices and innovation lifts -- src/EqFMatrices.cpp
# ----------------------------------------------------------------------------------------------
def eqf_state_matrix_A(X, xi0, imuVel):
    """EqFMatrices.cpp:277-317."""
    xi0 = project_to_manifold(xi0)
    N = len(xi0.ids)
    A0t = np.zeros((5 + 3 * N, 5 + 3 * N))
    A0t[2:5, 0:2] = -stereo_sphere_chart_inv_diff(np.zeros(2), xi0.gravityDir) * GRAVITY_CONSTANT  # :289
    R_IC = quat_to_matrix(xi0.cameraOffset.q)
    R_Ahat = quat_to_matrix(X.A.q)
    for i in range(N):
        Qhat = quat_to_matrix(X.Q[i].q) * X.Q[i].a
        A0t[5 + 3 * i : 8 + 3 * i, 2:5] = -Qhat @ R_IC.T @ R_Ahat.T
    xi_hat = state_group_action(X, xi0)
    U_I = np.concatenate([imuVel.omega, xi_hat.velocity])
    U_C = xi0.cameraOffset.inverse().adjoint() @ U_I
    v_C = U_C[3:6]
    for i in range(N):
        Qhat = quat_to_matrix(X.Q[i].q) * X.Q[i].a
        qhat = xi_hat.p[i]
        A_qi = (
            -Qhat
            @ (skew(qhat) @ skew(v_C) - 2 * np.outer(v_C, qhat) + np.outer(qhat, v_C))
            @ np.linalg.inv(Qhat)
            * (1 / float(qhat @ qhat))
        )
        A0t[5 + 3 * i : 8 + 3 * i, 5 + 3 * i : 8 + 3 * i] = A_qi
    return A0t


def eqf_output_matrix_C(xi0):
    """EqFMatrices.cpp:319-344."""
    xi0 = project_to_manifold(xi0)
    N = len(xi0.ids)
    C0 = np.zeros((2 * N, 5 + 3 * N))
    for i in range(N):
        qi0 = xi0.p[i]
        nrm = np.linalg.norm(qi0)
        yi0 = qi0 / nrm
        C0[2 * i : 2 * i + 2, 5 + 3 * i : 8 + 3 * i] = (
            1 / nrm * stereo_sphere_chart_diff(yi0, yi0) @ (np.eye(3) - np.outer(yi0, yi0))
        )
    return C0


def eqf_input_matrix_B(X, xi0):
    """EqFMatrices.cpp:346-382."""
    xi0 = project_to_manifold(xi0)
    N = len(xi0.ids)
    Bt = np.zeros((5 + 3 * N, 6))
    xi_hat = state_group_action(X, xi0)
    R_A = quat_to_matrix(X.A.q)
    Bt[0:2, 0:3] = stereo_sphere_chart_diff(xi0.gravityDir, xi0.gravityDir) @ R_A @ skew(xi_hat.gravityDir)
    Bt[2:5, 0:3] = R_A @ skew(xi_hat.velocity)
    Bt[2:5, 3:6] = R_A
    RT_IC = quat_to_matrix(quat_inverse(xi0.cameraOffset.q))
    x_IC = xi0.cameraOffset.x
    for i in range(N):
        Qhat = quat_to_matrix(X.Q[i].q) * X.Q[i].a
        Bt[5 + 3 * i : 8 + 3 * i, 0:3] = Qhat @ (skew(xi_hat.p[i]) @ RT_IC + RT_IC @ skew(x_IC))
    return Bt


def lift_innovation(baseInnovation, xi0):
    """EqFMatrices.cpp:35-67 (2-arg)."""
    xi0 = project_to_manifold(xi0)
    N = len(xi0.ids)
    Delta = VIOAlgebra(ids=xi0.ids)
    Delta.U = np.zeros(6)
    Delta.U[0:3] = -skew(xi0.gravityDir) @ stereo_sphere_chart_inv_diff(np.zeros(2), xi0.gravityDir) @ baseInnovation[0:2]
    gamma_v = baseInnovation[2:5]
    Delta.u = -gamma_v - skew(Delta.U[0:3]) @ xi0.velocity
    Delta.W = np.zeros((N, 4))
    for i in range(N):
        g = baseInnovation[5 + 3 * i : 8 + 3 * i]
        q = xi0.p[i]
        n2 = float(q @ q)
        Delta.W[i, 0:3] = -np.cross(q, g) / n2
        Delta.W[i, 3] = -float(q @ g) / n2
    return Delta


def lift_total_space_innovation(totalInnovation, xi0):
    """EqFMatrices.cpp:69-96."""
    N = len(xi0.ids)
    Delta = VIOAlgebra(ids=xi0.ids)
    Delta.U = totalInnovation[0:6].copy()
    Delta.u = -totalInnovation[6:9] - skew(Delta.U[0:3]) @ xi0.velocity
    Delta.W = np.zeros((N, 4))
    for i in range(N):
        g = totalInnovation[9 + 3 * i : 12 + 3 * i]
        q = xi0.p[i]
        n2 = float(q @ q)
        Delta.W[i, 0:3] = -np.cross(q, g) / n2
        Delta.W[i, 3] = -float(q @ g) / n2
    return Delta


def _wls_pieces(baseInnovation, xi0, X, DeltaU0):
    """Shared body of EqFMatrices.cpp:98-171 and :173-252 (the two functions duplicate it)."""
    xiHat = state_group_action(X, xi0)
    eta0 = project_to_manifold(xi0).gravityDir
    eta0 = eta0 / np.linalg.norm(eta0)
    N = len(xi0.ids)
    KPara = np.zeros((6, 4))
    KPara[0:3, 0] = eta0
    KPara[3:6, 1:4] = np.eye(3)
    KPerp = np.zeros((6, 6))
    KPerp[0:3, 0:3] = np.eye(3) - np.outer(eta0, eta0)
    R_Cq = quat_mul(xiHat.pose.q, xiHat.cameraOffset.q)
    R_CTransMat = quat_to_matrix(quat_inverse(R_Cq))
    AdP0 = xi0.pose.adjoint()
    DeltaUFixed = KPerp @ DeltaU0
    coeffMat = np.zeros((3 * N, 4))
    observationVec = np.zeros(3 * N)
    D = np.zeros((5 + 3 * N, 3 * N))
    PC = xiHat.pose * xiHat.cameraOffset
    for i in range(N):
        g = baseInnovation[5 + 3 * i : 8 + 3 * i]
        pHat = PC.apply(xiHat.p[i])
        alpha = -quat_rotate(R_Cq, X.Q[i].inverse().apply(g))
        pHatMat = np.zeros((3, 6))
        pHatMat[:, 0:3] = -skew(pHat)
        pHatMat[:, 3:6] = np.eye(3)
        observationVec[3 * i : 3 * i + 3] = alpha - pHatMat @ AdP0 @ DeltaUFixed
        coeffMat[3 * i : 3 * i + 3, :] = pHatMat @ AdP0 @ KPara
        D[5 + 3 * i : 8 + 3 * i, 3 * i : 3 * i + 3] = X.Q[i].as_matrix3() @ R_CTransMat
    return KPara, DeltaUFixed, coeffMat, observationVec, D


def _qr_solve(M, b):
    """Eigen householderQr().solve() on a square system (EqFMatrices.cpp:160-163, :240-242)."""
    Qm, Rm = np.linalg.qr(M)
    return np.linalg.solve(Rm, Qm.T @ b)


def bundle_lift(baseInnovation, xi0, X, Sigma):
    """EqFMatrices.cpp:173-252."""
    eta0 = project_to_manifold(xi0).gravityDir
    eta0 = eta0 / np.linalg.norm(eta0)
    DeltaU = np.zeros(6)
    DeltaU[0:3] = -skew(eta0) @ stereo_sphere_chart_inv_diff(np.zeros(2), eta0) @ baseInnovation[0:2]
    KPara, DeltaUFixed, coeffMat, observationVec, D = _wls_pieces(baseInnovation, xi0, X, DeltaU)
    weightMat = D.T @ np.linalg.inv(Sigma) @ D  # :239
    sol = _qr_solve(coeffMat.T @ weightMat @ coeffMat, coeffMat.T @ weightMat @ observationVec)
    DeltaU = DeltaUFixed + KPara @ sol
    N = len(xi0.ids)
    lifted = np.zeros(9 + 3 * N)
    lifted[6:] = baseInnovation[2:]
    lifted[0:6] = DeltaU
    return lifted


def lift_innovation_wls(baseInnovation, xi0, X, Sigma):
    """EqFMatrices.cpp:98-171 (4-arg liftInnovation; used by the reference's tests only)."""
    Delta = lift_innovation(baseInnovation, xi0)
    KPara, DeltaUFixed, coeffMat, observationVec, D = _wls_pieces(baseInnovation, xi0, X, Delta.U)
    weightMat = D.T @ np.linalg.inv(Sigma) @ D
    sol = _qr_solve(coeffMat.T @ weightMat @ coeffMat, coeffMat.T @ weightMat @ observationVec)
    Delta.U = DeltaUFixed + KPara @ sol
    Delta.u = -baseInnovation[2:5] - skew(Delta.U[0:3]) @ xi0.velocity
    return Delta


def lift_total_space_innovation_discrete(totalInnovation, xi0):
    """EqFMatrices.cpp:254-275."""
    A = se3_exp(totalInnovation[0:6])
    w = xi0.velocity - quat_rotate(A.q, xi0.velocity + totalInnovation[6:9])
    Q = []
    for i in range(len(xi0.ids)):
        qi = xi0.p[i]
        qi1 = qi + totalInnovation[9 + 3 * i : 12 + 3 * i]
        Q.append(
            SOT3(so3_from_vectors(qi1 / np.linalg.norm(qi1), qi / np.linalg.norm(qi)), np.linalg.norm(qi) / np.linalg.norm(qi1))
        )
    return VIOGroup(A, w, Q, xi0.ids)


# ----------------------------------------------------------------------------------------------
# Settings and the filter -- include/eqf_vio/VIOFilterSettings.h, src/VIOFilter.cpp
# ----------------------------------------------------------------------------------------------
class Settings:
    """VIOFilterSettings.h:28-54 (defaults :29-50)."""

    def __init__(self, **kw):
        self.biasOmegaProcessVariance = 0.001
        self.biasAccelProcessVariance = 0.001
        self.gravityProcessVariance = 0.001
        self.velocityProcessVariance = 0.001
        self.pointProcessVariance = 0.001
        self.velOmegaVariance = 0.1
        self.velAccelVariance = 0.1
        self.measurementVariance = 0.1
        self.initialGravityVariance = 1.0
        self.initialVelocityVariance = 1.0
        self.initialPointVariance = 1.0
        self.initialBiasOmegaVariance = 1.0
        self.initialBiasAccelVariance = 1.0
        self.initialSceneDepth = 1.0
        self.outlierThreshold = 0.01
        self.useInnovationLift = True
        self.useDiscreteInnovationLift = True
        self.useDiscreteVelocityLift = True
        self.fastRiccati = False
        self.initialAccelBias = np.zeros(3)
        self.initialOmegaBias = np.zeros(3)
        self.cameraOffset = SE3()
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


class VIOFilter:
    """src/VIOFilter.cpp -- dense fp64, operation order of the reference."""

    def __init__(self, settings):  # VIOFilter.cpp:60-73
        s = self.settings = settings
        self.inputBias = np.zeros(6)
        self.xi0 = VIOState()
        self.X = VIOGroup.identity()
        self.Sigma = np.eye(SIGMA_BASE_SIZE)
        self.Sigma[0:3, 0:3] = np.eye(3) * s.initialBiasOmegaVariance
        self.Sigma[3:6, 3:6] = np.eye(3) * s.initialBiasAccelVariance
        self.Sigma[6:8, 6:8] = np.eye(2) * s.initialGravityVariance
        self.Sigma[8:11, 8:11] = np.eye(3) * s.initialVelocityVariance
        self.xi0.cameraOffset = s.cameraOffset.copy()
        self.inputBias[0:3] = s.initialOmegaBias
        self.inputBias[3:6] = s.initialAccelBias
        self.initialisedFlag = False
        self.currentTime = -1.0
        self.currentVelocity = IMUVelocity()
        self.accumulatedVelocity = IMUVelocity()
        self.accumulatedTime = 0.0
        self.last = {}  # internals of the most recent update, for kernel-level parity tests

    # -- VIOFilter.cpp:74-82.  aux: object with initialAttitude (quaternion wxyz), initialPosition, cameraOffset (SE3)
    def setAuxiliaryData(self, aux):
        self.auxData = aux
        self.xi0.pose = SE3(np.array(aux.initialAttitude, dtype=float), np.array(aux.initialPosition, dtype=float))
        self.xi0.velocity = np.zeros(3)
        self.initialisedFlag = True
        self.xi0.cameraOffset = aux.cameraOffset.copy()

    # -- VIOFilter.cpp:93-118: landmarks given in the inertial frame; Sigma = initialPointVariance * I with the
    # 11x11 base block kept.
    def setInertialPoints(self, points, ids):
        points = np.asarray(points, dtype=float).reshape(-1, 3)
        N = len(points)
        tf = (self.xi0.pose * self.xi0.cameraOffset).inverse()
        self.xi0.p = np.array([tf.apply(pt) for pt in points]).reshape(-1, 3)
        self.xi0.ids = np.array(ids, dtype=np.int64)
        self.X.Q = [SOT3() for _ in range(N)]
        self.X.ids = np.array(ids, dtype=np.int64)
        n = SIGMA_BASE_SIZE + 3 * N
        Sn = np.eye(n) * self.settings.initialPointVariance
        Sn[0:SIGMA_BASE_SIZE, 0:SIGMA_BASE_SIZE] = self.Sigma[0:SIGMA_BASE_SIZE, 0:SIGMA_BASE_SIZE]
        self.Sigma = Sn

    # -- outputs (VIOFilter.cpp:304-309, :343)
    def stateEstimate(self):
        return state_group_action(self.X, self.xi0)

    def stateCovariance(self):
        return self.Sigma

    def getTime(self):
        return self.currentTime

    # -- VIOFilter.cpp:120-144
    def processIMUData(self, imu):
        unbiased = imu - self.inputBias
        if not self.initialisedFlag:
            self.initialiseFromIMUData(unbiased)
        self.integrateUpToTime(imu.stamp, not self.settings.fastRiccati)
        self.currentVelocity = unbiased
        self.currentTime = imu.stamp

    def initialiseFromIMUData(self, imu):
        self.xi0.pose = SE3()
        self.xi0.velocity = np.zeros(3)
        self.initialisedFlag = True
        g = imu.accel / np.linalg.norm(imu.accel)
        self.xi0.pose.q = so3_from_vectors(g, E3)

    # -- VIOFilter.cpp:146-209
    def integrateUpToTime(self, newTime, doRiccati=True):
        if self.currentTime < 0:
            return False
        dt = newTime - self.currentTime
        if dt <= 0:
            return False
        s = self.settings
        self.accumulatedTime += dt
        self.accumulatedVelocity = self.accumulatedVelocity + self.currentVelocity * dt
        N = len(self.xi0.ids)
        currentState = self.stateEstimate()
        if doRiccati:
            n = self.Sigma.shape[0]
            PMat = np.eye(n)
            PMat[0:3, 0:3] *= s.biasOmegaProcessVariance
            PMat[3:6, 3:6] *= s.biasAccelProcessVariance
            PMat[6:8, 6:8] *= s.gravityProcessVariance
            PMat[8:11, 8:11] *= s.velocityProcessVariance
            PMat[11:, 11:] *= s.pointProcessVariance
            A0t = eqf_state_matrix_A(self.X, self.xi0, self.accumulatedVelocity * (1.0 / self.accumulatedTime))
            Bt = eqf_input_matrix_B(self.X, self.xi0)
            R = np.eye(6)
            R[0:3, 0:3] *= s.velOmegaVariance
            R[3:6, 3:6] *= s.velAccelVariance
            A0tBiased = np.zeros((n, n))
            A0tBiased[6:, 6:] = A0t
            A0tBiased[6:, 0:6] = -Bt
            F = np.eye(n) + A0tBiased * self.accumulatedTime
            BtBiased = np.zeros((n, 6))
            BtBiased[6:, :] = Bt
            self.Sigma = self.accumulatedTime * (PMat + BtBiased @ R @ BtBiased.T) + F @ self.Sigma @ F.T
            self.accumulatedVelocity = IMUVelocity()
            self.accumulatedTime = 0.0
        if s.useDiscreteVelocityLift:
            self.X = self.X * lift_velocity_discrete(currentState, self.currentVelocity, dt)
        else:
            self.X = self.X * vio_exp(dt * lift_velocity(currentState, self.currentVelocity))
        self.currentTime = newTime
        return True

    # -- VIOFilter.cpp:211-230
    def matchMeasurementsToState(self, ids, y):
        stateIds = list(self.X.ids)
        out_ids = np.zeros(len(ids), dtype=np.int64)
        out_y = np.zeros((len(ids), 3))
        newPos = len(stateIds) - 1
        for idv, yv in zip(ids, y):
            if idv in stateIds:
                idx = stateIds.index(idv)
            else:
                newPos += 1
                idx = newPos
            out_ids[idx] = idv
            out_y[idx] = yv
        return out_ids, out_y

    # -- VIOFilter.cpp:232-302
    def processVisionData(self, stamp, ids, y):
        """ids: ascending int ids; y: (nb,3) unit bearings in the camera frame."""
        ids = np.array(ids, dtype=np.int64)
        y = np.array(y, dtype=float).reshape(-1, 3)
        self.last = {}
        if not self.integrateUpToTime(stamp) or not self.initialisedFlag:
            return
        assert np.all(np.diff(ids) >= 0)
        self.removeOldLandmarks(ids)
        m_ids, m_y = self.matchMeasurementsToState(ids, y)
        m_ids, m_y = self.removeOutliers(m_ids, m_y)
        self.addNewLandmarks(m_ids, m_y)
        assert np.array_equal(m_ids, self.X.ids)
        if len(m_ids) == 0:
            return
        s = self.settings
        y0 = measure_system_state(self.xi0)
        yerr = output_group_action(self.X.inverse(), m_y)
        delta = output_coordinate_chart(yerr, y0)
        C0 = eqf_output_matrix_C(self.xi0)
        N = len(self.xi0.ids)
        QMat = s.measurementVariance * np.eye(2 * N)
        C0Biased = np.zeros((2 * N, C0.shape[1] + 6))
        C0Biased[:, 6:] = C0
        S = C0Biased @ self.Sigma @ C0Biased.T + QMat
        K = self.Sigma @ C0Biased.T @ np.linalg.inv(S)
        baseInnovationBiased = K @ delta
        gammaE = baseInnovationBiased[6:]
        gammaBias = baseInnovationBiased[0:6]
        Gamma = None
        if s.useInnovationLift:
            Gamma = bundle_lift(gammaE, self.xi0, self.X, self.Sigma[6:, 6:])
            if s.useDiscreteInnovationLift:
                Delta = lift_total_space_innovation_discrete(Gamma, self.xi0)
            else:
                Delta = vio_exp(lift_total_space_innovation(Gamma, self.xi0))
        else:
            Delta = vio_exp(lift_innovation(gammaE, self.xi0))
        self.last = dict(delta=delta, C0=C0, S=S, K=K, gamma=baseInnovationBiased, Gamma=Gamma, Sigma_prior=self.Sigma.copy())
        self.inputBias = self.inputBias + gammaBias
        self.X = Delta * self.X
        self.Sigma = self.Sigma - K @ C0Biased @ self.Sigma

    # -- VIOFilter.cpp:345-391
    def addNewLandmarks(self, m_ids, m_y):
        stateIds = set(int(i) for i in self.X.ids)
        new = [k for k, idv in enumerate(m_ids) if int(idv) not in stateIds]
        if not new:
            return
        est = self.stateEstimate().p
        d2 = np.sort(np.sum(est * est, axis=1))
        medianDepth = self.settings.initialSceneDepth
        if len(d2) > 0:
            medianDepth = d2[len(d2) // 2] ** 0.5  # nth_element at size/2
        newp = m_y[new] * medianDepth
        self.xi0.p = np.vstack([self.xi0.p, newp])
        self.xi0.ids = np.concatenate([self.xi0.ids, m_ids[new]])
        self.X.ids = np.concatenate([self.X.ids, m_ids[new]])
        self.X.Q = self.X.Q + [SOT3() for _ in new]
        og = self.Sigma.shape[0]
        k = 3 * len(new)
        Sn = np.zeros((og + k, og + k))
        Sn[0:og, 0:og] = self.Sigma
        Sn[og:, og:] = np.eye(k) * self.settings.initialPointVariance
        self.Sigma = Sn

    # -- VIOFilter.cpp:393-427
    def removeOldLandmarks(self, meas_ids):
        meas = set(int(i) for i in meas_ids)
        lost = [i for i, idv in enumerate(self.X.ids) if int(idv) not in meas]
        for li in reversed(lost):
            self.removeLandmarkAtIndex(li)

    def removeLandmarkAtIndex(self, idx):
        self.xi0.p = np.delete(self.xi0.p, idx, axis=0)
        self.xi0.ids = np.delete(self.xi0.ids, idx)
        self.X.ids = np.delete(self.X.ids, idx)
        del self.X.Q[idx]
        r = SIGMA_BASE_SIZE + 3 * idx
        self.Sigma = np.delete(np.delete(self.Sigma, slice(r, r + 3), axis=0), slice(r, r + 3), axis=1)

    # -- VIOFilter.cpp:429-443
    def removeOutliers(self, m_ids, m_y):
        yHat = measure_system_state(self.stateEstimate())
        m_ids = list(m_ids)
        m_y = list(m_y)
        for i in range(len(yHat) - 1, -1, -1):
            if np.linalg.norm(m_y[i] - yHat[i]) > self.settings.outlierThreshold:
                self.removeLandmarkAtIndex(i)
                del m_ids[i]
                del m_y[i]
        return np.array(m_ids, dtype=np.int64), np.array(m_y, dtype=float).reshape(-1, 3)
