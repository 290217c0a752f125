"""The reference's property tests (eqf_vio/test/*.cpp), restated in own code against the fp64 oracle.

The reference's tests are unseeded randomised properties on N = 5 landmarks (TEST_REPS = 25, NEAR_ZERO = 1e-12,
eqf_vio/test/CMakeLists.txt:30-31).  Here they run seeded, with fewer repetitions, against oracle/eqf_numpy.py;
the C++ oracle is tied to the numpy one in test_oracle_cross.py.  These properties are what pins the oracle:
the reference has no golden vectors for this path.
"""
import numpy as np
import pytest
import scipy.linalg

from oracle import eqf_numpy as O

NEAR_ZERO = 1e-12
REPS = 6
IDS = [0, 1, 2, 3, 4]
N = len(IDS)


def rq(rng):
    q = rng.standard_normal(4)
    return q / np.linalg.norm(q)


def random_state(rng):  # testing_utilities.cpp:23-42
    return O.VIOState(O.SE3(rq(rng), rng.uniform(-1, 1, 3)), rng.uniform(-1, 1, 3), rng.uniform(-1, 1, (N, 3)), IDS,
                      O.SE3(rq(rng), np.zeros(3)))


def random_group(rng):  # testing_utilities.cpp:67-81
    return O.VIOGroup(O.SE3(rq(rng), rng.uniform(-1, 1, 3)), rng.uniform(-1, 1, 3),
                      [O.SOT3(rq(rng), rng.uniform(1, 6)) for _ in IDS], IDS)


def random_velocity(rng):
    return O.IMUVelocity(0, rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3))


def log_norm(X):  # testing_utilities.cpp:83-91
    return np.linalg.norm(O.se3_log(X.A)) + np.linalg.norm(X.w) + sum(np.linalg.norm(O.sot3_log(Q)) for Q in X.Q)


def state_distance(a, b):
    d = np.linalg.norm(a.velocity - b.velocity) + np.sum(np.linalg.norm(a.p - b.p, axis=1))
    if isinstance(a, O.VIOManifoldState):
        return d + np.linalg.norm(a.gravityDir - b.gravityDir)
    return d + np.linalg.norm(O.se3_log(a.pose.inverse() * b.pose))


def assert_fd_converges(f0, fd, lo=1, hi=7, floor=1e-8):
    """(f(dt e) - f(0))/dt -> J e: the error must not grow as dt shrinks (until roundoff, like the reference's
    'if (dist > 1e-8)' guards) and must be first order small."""
    prev = 1e8
    for i in range(lo, hi + 1):
        dist = fd(10.0 ** -i)
        if dist > floor:
            assert dist <= prev * (1 + 1e-9), (i, dist, prev)
        prev = max(dist, floor)
    assert fd(1e-5) < 1e-3


# ---------------------------------------------------------------- test_common.cpp
def test_skew_vex_and_exponentials():
    rng = np.random.default_rng(1)
    for _ in range(REPS):
        v, w = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        assert np.allclose(O.skew(v) @ w, np.cross(v, w), atol=1e-15)  # :27-38
        assert np.allclose(O.vex(O.skew(v)), v)
        R = O.quat_to_matrix(O.so3_exp(w))  # :40-58 SO3Exp vs expm
        assert np.abs(R - scipy.linalg.expm(O.skew(w))).max() < 1e-8
        assert np.abs(O.so3_log(O.so3_exp(w)) - w).max() < 1e-8
        u = rng.uniform(-1, 1, 6)  # :60-95 SE3Exp vs expm of the wedge
        T = O.se3_exp(u)
        W = np.zeros((4, 4))
        W[0:3, 0:3] = O.skew(u[0:3])
        W[0:3, 3] = u[3:6]
        E = scipy.linalg.expm(W)
        assert np.abs(O.quat_to_matrix(T.q) - E[0:3, 0:3]).max() < 1e-8 and np.abs(T.x - E[0:3, 3]).max() < 1e-8
        assert np.abs(O.se3_log(T) - u).max() < 1e-8


def test_so3_from_vectors():  # test_common.cpp:97-116
    rng = np.random.default_rng(2)
    for _ in range(REPS):
        v, w = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        q = O.so3_from_vectors(v, w)
        R = O.quat_to_matrix(q)
        assert np.abs(R @ (v / np.linalg.norm(v)) - w / np.linalg.norm(w)).max() < 1e-8
        assert np.abs(R.T @ R - np.eye(3)).max() < 1e-8
    with pytest.raises(O.AntipodalError):  # libs/core/src/SO3.cpp:160-161
        O.so3_from_vectors(np.array([0, 0, -1.0]), np.array([0, 0, 1.0]))


def test_se3_product_stays_orthonormal():  # test_common.cpp:118-159
    rng = np.random.default_rng(3)
    T = O.SE3()
    for _ in range(1000):
        T = T * O.se3_exp(0.1 * rng.uniform(-1, 1, 6))
    R = O.quat_to_matrix(T.q)
    assert np.abs(R.T @ R - np.eye(3)).max() < 1e-8


# ---------------------------------------------------------------- test_VIOGroup.cpp
def test_group_axioms():
    rng = np.random.default_rng(4)
    for _ in range(REPS):
        X1, X2, X3 = random_group(rng), random_group(rng), random_group(rng)
        assert log_norm(X1 * X1.inverse()) < 1e-10  # :26-36 (1e-12 in the reference, a in [1,6] amplifies)
        assert log_norm(X1.inverse() * X1) < 1e-10
        assert log_norm(((X1 * X2) * X3) * (X1 * (X2 * X3)).inverse()) < 1e-10  # :38-48
        I = O.VIOGroup.identity(IDS)
        assert log_norm((X1 * I) * X1.inverse()) < 1e-10  # :50-60
        assert log_norm((I * X1) * X1.inverse()) < 1e-10


# ---------------------------------------------------------------- test_VIOGroupActions.cpp
def test_state_action_is_right_action_and_output_equivariance():
    rng = np.random.default_rng(5)
    for _ in range(REPS):
        X1, X2, xi = random_group(rng), random_group(rng), random_state(rng)
        a = O.state_group_action(X2, O.state_group_action(X1, xi))  # :28-44
        b = O.state_group_action(X1 * X2, xi)
        assert state_distance(a, b) < 1e-10
        y = rng.standard_normal((N, 3))
        y /= np.linalg.norm(y, axis=1, keepdims=True)
        ya = O.output_group_action(X2, O.output_group_action(X1, y))  # :46-66
        yb = O.output_group_action(X1 * X2, y)
        assert np.abs(ya - yb).max() < 1e-10
        # h(phi(X, xi)) = rho(X, h(xi))   :68-92
        h1 = O.measure_system_state(O.state_group_action(X1, xi))
        h2 = O.output_group_action(X1, O.measure_system_state(xi))
        assert np.abs(h1 - h2).max() < 1e-10


# ---------------------------------------------------------------- test_VIOLift.cpp
def state_vec_diff(a, b):  # testing_utilities.cpp:44-58
    return np.concatenate([O.se3_log(a.pose.inverse() * b.pose), b.velocity - a.velocity, (b.p - a.p).reshape(-1)])


def test_velocity_lift_matches_system_to_first_order():  # :28-54
    rng = np.random.default_rng(6)
    for _ in range(REPS):
        xi0, vel = random_state(rng), random_velocity(rng)
        lam = O.lift_velocity(xi0, vel)

        def fd(dt):
            xi1 = O.integrate_system_function(xi0, vel, dt)
            xi2 = O.state_group_action(O.vio_exp(dt * lam), xi0)
            return np.linalg.norm(state_vec_diff(xi0, xi1) / dt - state_vec_diff(xi0, xi2) / dt)

        assert_fd_converges(None, fd, lo=0, hi=7)


def test_discrete_velocity_lift_is_exact_on_the_manifold():  # :56-76
    rng = np.random.default_rng(7)
    for _ in range(REPS):
        Xi0, vel = random_state(rng), random_velocity(rng)
        xi1 = O.project_to_manifold(O.integrate_system_function(Xi0, vel, 0.1))
        xi0 = O.project_to_manifold(Xi0)
        xi2 = O.state_group_action(O.lift_velocity_discrete(xi0, vel, 0.1), xi0)
        assert state_distance(xi1, xi2) < NEAR_ZERO * 10


@pytest.mark.parametrize("which", ["plain", "wls", "total"])
def test_innovation_lifts_reproject_to_the_base_innovation(which):  # :78-197
    rng = np.random.default_rng(8)
    for _ in range(2):
        Xi0 = random_state(rng)
        xi0 = O.project_to_manifold(Xi0)
        X = random_group(rng)
        M = rng.uniform(-1, 1, (5 + 3 * N, 5 + 3 * N))
        Sigma = M @ M.T
        for j in range(0, 5 + 3 * N, 3):
            base = np.zeros(5 + 3 * N)
            base[j] = 1.0
            if which == "plain":
                lifted = O.lift_innovation(base, xi0)
            elif which == "wls":
                lifted = O.lift_innovation_wls(base, Xi0, X, Sigma)
            else:
                lifted = O.lift_total_space_innovation(O.bundle_lift(base, Xi0, X, Sigma), Xi0)

            def fd(dt):
                xi1 = O.state_group_action(O.vio_exp(dt * lifted), xi0)
                return np.linalg.norm(O.euclid_coordinate_chart(xi1, xi0) / dt - base)

            assert_fd_converges(None, fd, lo=1, hi=6, floor=1e-7)


def test_bundle_lift_equals_four_argument_lift():  # :199-219
    rng = np.random.default_rng(9)
    for _ in range(2):
        Xi0, X = random_state(rng), random_group(rng)
        M = rng.uniform(-1, 1, (5 + 3 * N, 5 + 3 * N))
        Sigma = M @ M.T
        for j in range(5 + 3 * N):
            base = np.zeros(5 + 3 * N)
            base[j] = 0.1
            l1 = O.lift_total_space_innovation(O.bundle_lift(base, Xi0, X, Sigma), Xi0)
            l2 = O.lift_innovation_wls(base, Xi0, X, Sigma)
            assert log_norm(O.vio_exp(l1 - l2)) < NEAR_ZERO * 1e3


def test_discrete_and_continuous_innovation_lifts_converge():  # :221-252
    rng = np.random.default_rng(10)
    Xi0, X = random_state(rng), random_group(rng)
    M = rng.uniform(-1, 1, (5 + 3 * N, 5 + 3 * N))
    Sigma = M @ M.T
    for j in range(0, 5 + 3 * N, 2):
        base = np.zeros(5 + 3 * N)
        base[j] = 0.1
        total = O.bundle_lift(base, Xi0, X, Sigma)
        prev = 1e8
        for i in range(1, 7):
            dt = 10.0 ** -i
            d1 = O.lift_total_space_innovation_discrete(dt * total, Xi0)
            d2 = O.vio_exp(O.lift_total_space_innovation(dt * total, Xi0))
            dist = log_norm(d1.inverse() * d2)
            if dist < 1e-10:
                break
            assert dist <= prev
            prev = dist


# ---------------------------------------------------------------- test_CoordinateCharts.cpp
def test_sphere_charts_round_trip_and_differentials():
    rng = np.random.default_rng(11)
    for _ in range(REPS):
        eta = rng.standard_normal(3)
        eta /= np.linalg.norm(eta)
        pole = rng.standard_normal(3)
        pole /= np.linalg.norm(pole)
        y = O.e3_project_sphere(eta)  # :26-50
        assert np.abs(O.e3_project_sphere_inv(y) - eta).max() < 1e-10
        ys = O.stereo_sphere_chart(eta, pole)  # :52-80
        assert np.abs(O.stereo_sphere_chart_inv(ys, pole) - eta).max() < 1e-10
        assert np.abs(O.stereo_sphere_chart(pole, pole)).max() < 1e-12  # the pole maps to the origin
        # differentials by finite differences :82-160
        D = O.stereo_sphere_chart_diff(eta, pole)
        Di = O.stereo_sphere_chart_inv_diff(ys, pole)
        h = 1e-6
        for k in range(3):
            e = np.zeros(3)
            e[k] = h
            # derivative along the sphere's tangent directions only: project e
            t = e - (e @ eta) * eta
            n1 = (eta + t) / np.linalg.norm(eta + t)
            fd = (O.stereo_sphere_chart(n1, pole) - ys)
            assert np.abs(fd - D @ (n1 - eta)).max() < 1e-9
        for k in range(2):
            e = np.zeros(2)
            e[k] = h
            fd = (O.stereo_sphere_chart_inv(ys + e, pole) - eta) / h
            assert np.abs(fd - Di[:, k]).max() < 1e-5
        assert np.abs(D @ Di - np.eye(2)).max() < 1e-9


def test_state_and_output_charts_round_trip():  # :162-220
    rng = np.random.default_rng(12)
    for _ in range(REPS):
        xi0 = O.project_to_manifold(random_state(rng))
        xi = O.project_to_manifold(random_state(rng))
        eps = O.euclid_coordinate_chart(xi, xi0)
        back = O.euclid_coordinate_chart_inv(eps, xi0)
        assert state_distance(back, xi) < 1e-9
        y0 = O.measure_system_state(xi0)
        y = O.measure_system_state(xi)
        d = O.output_coordinate_chart(y, y0)
        assert np.abs(O.output_coordinate_chart_inv(d, y0) - y).max() < 1e-9


# ---------------------------------------------------------------- test_EqFMatrices.cpp
def test_state_matrix_A_is_the_differential_of_the_error_dynamics():  # :28-96
    rng = np.random.default_rng(13)
    X, xi0, vel = random_group(rng), O.project_to_manifold(random_state(rng)), random_velocity(rng)
    A0 = O.eqf_state_matrix_A(X, xi0, vel)

    def a0(eps):
        xi_hat = O.state_group_action(X, xi0)
        xi = O.state_group_action(X, O.euclid_coordinate_chart_inv(eps, xi0))
        L = O.lift_velocity(xi, vel) - O.lift_velocity(xi_hat, vel)
        xi_e1 = O.state_group_action(X.inverse(), O.state_group_action(O.vio_exp(L), xi_hat))
        return O.euclid_coordinate_chart(xi_e1, xi0)

    assert np.linalg.norm(a0(np.zeros(5 + 3 * N))) < 1e-11
    dirs = [np.eye(5 + 3 * N)[j] for j in range(0, 5 + 3 * N, 2)] + [rng.uniform(-1, 1, 5 + 3 * N) for _ in range(4)]
    for e in dirs:
        assert_fd_converges(None, lambda dt: np.linalg.norm(a0(dt * e) / dt - A0 @ e), lo=1, hi=7, floor=1e-6)


def test_input_matrix_B_is_the_differential_wrt_the_velocity():  # :98-152
    rng = np.random.default_rng(14)
    X, xi0, vel = random_group(rng), O.project_to_manifold(random_state(rng)), random_velocity(rng)
    B = O.eqf_input_matrix_B(X, xi0)

    def b0(v6):
        xi_hat = O.state_group_action(X, xi0)
        L = O.lift_velocity(xi_hat, vel + v6) - O.lift_velocity(xi_hat, vel)
        xi_e1 = O.state_group_action(X.inverse(), O.state_group_action(O.vio_exp(L), xi_hat))
        return O.euclid_coordinate_chart(xi_e1, xi0)

    assert np.linalg.norm(b0(np.zeros(6))) < 1e-11
    for e in list(np.eye(6)) + [rng.uniform(-1, 1, 6) for _ in range(4)]:
        assert_fd_converges(None, lambda dt: np.linalg.norm(b0(dt * e) / dt - B @ e), lo=1, hi=5, floor=1e-8)


def test_output_matrix_C_is_the_differential_of_the_output():  # :154-217
    rng = np.random.default_rng(15)
    xi0 = O.project_to_manifold(random_state(rng))
    C0 = O.eqf_output_matrix_C(xi0)
    y0 = O.measure_system_state(xi0)

    def c0(eps):
        return O.output_coordinate_chart(O.measure_system_state(O.euclid_coordinate_chart_inv(eps, xi0)), y0)

    assert np.linalg.norm(c0(np.zeros(5 + 3 * N))) < 1e-11
    dirs = [np.eye(5 + 3 * N)[j] for j in range(5 + 3 * N)] + [rng.uniform(-1, 1, 5 + 3 * N) for _ in range(4)]
    for e in dirs:
        assert_fd_converges(None, lambda dt: np.linalg.norm(c0(dt * e) / dt - C0 @ e), lo=1, hi=6, floor=1e-8)


# ---------------------------------------------------------------- closed forms used by the kernels
def test_closed_forms_used_on_the_device():
    rng = np.random.default_rng(16)
    for _ in range(REPS):
        xi0 = O.project_to_manifold(random_state(rng))
        X = random_group(rng)
        vel = random_velocity(rng)
        A0 = O.eqf_state_matrix_A(X, xi0, vel)
        xi_hat = O.state_group_action(X, xi0)
        v_C = (xi0.cameraOffset.inverse().adjoint() @ np.concatenate([vel.omega, xi_hat.velocity]))[3:6]
        for i in range(N):
            RQ = O.quat_to_matrix(X.Q[i].q)
            q = xi_hat.p[i]
            inner = O.skew(q) @ O.skew(v_C) - 2 * np.outer(v_C, q) + np.outer(q, v_C)
            Aq = -RQ @ inner @ RQ.T / (q @ q)  # the scale a_i cancels: no 3x3 inverse needed
            assert np.abs(Aq - A0[5 + 3 * i:8 + 3 * i, 5 + 3 * i:8 + 3 * i]).max() < 1e-11


def test_auxiliary_start_places_landmarks_in_the_camera_frame():
    """setAuxiliaryData + setInertialPoints (VIOFilter.cpp:74-118): xi0 landmarks = (pose * cameraOffset)^-1 * p,
    Sigma = initialPointVariance * I outside the base block, no gravity alignment at the first IMU sample."""
    from types import SimpleNamespace

    rng = np.random.default_rng(5)
    att = rng.normal(size=4)
    att /= np.linalg.norm(att)
    cq = rng.normal(size=4)
    cq /= np.linalg.norm(cq)
    pos, cx = rng.normal(size=3), 0.1 * rng.normal(size=3)
    cam_pts = rng.normal(size=(7, 3)) + np.array([0, 0, 5.0])
    T = O.SE3(att, pos) * O.SE3(cq, cx)
    pts = np.array([T.apply(p) for p in cam_pts])
    f = O.VIOFilter(O.Settings(initialPointVariance=12.5, initialVelocityVariance=3.0))
    f.setAuxiliaryData(SimpleNamespace(initialAttitude=att, initialPosition=pos, cameraOffset=O.SE3(cq, cx)))
    f.setInertialPoints(pts, np.arange(7) + 40)
    est = f.stateEstimate()
    assert np.abs(est.p - cam_pts).max() < 1e-13
    assert np.array_equal(est.ids, np.arange(7) + 40)
    S = f.stateCovariance()
    assert S.shape == (32, 32)
    assert np.array_equal(S[11:, 11:], 12.5 * np.eye(21)) and not S[:11, 11:].any()
    assert S[8, 8] == 3.0
    # already initialised: the first IMU sample must not re-align the attitude with gravity (VIOFilter.cpp:123-125)
    f.processIMUData(O.IMUVelocity(0.0, np.zeros(3), np.array([0.0, 9.81, 0.0])))
    assert np.array_equal(f.xi0.pose.q, att)
