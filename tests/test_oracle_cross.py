"""Ties the C++ oracle (oracle/eqf_oracle.cpp, the timed CPU baseline) to the numpy restatement and to the
committed golden vectors.  CPU only."""
import numpy as np
import pytest

from helpers import load_golden, rel_fro, run_oracle_on_golden
from oracle import eqf_numpy as O


def _rq(rng):
    q = rng.standard_normal(4)
    return q / np.linalg.norm(q)


@pytest.mark.parametrize("N", [1, 5, 17])
def test_eqf_matrices_cpp_equals_numpy(oracle_lib, N):
    rng = np.random.default_rng(100 + N)
    ids = np.arange(N)
    camq, camx = _rq(rng), rng.uniform(-0.1, 0.1, 3)
    Xi = O.VIOState(O.SE3(_rq(rng), rng.uniform(-1, 1, 3)), rng.uniform(-1, 1, 3), rng.uniform(-1, 1, (N, 3)) + [0, 0, 3.0], ids,
                    O.SE3(camq, camx))
    X = O.VIOGroup(O.SE3(_rq(rng), rng.uniform(-1, 1, 3)), rng.uniform(-1, 1, 3), [O.SOT3(_rq(rng), rng.uniform(1, 6)) for _ in ids], ids)
    omega = rng.uniform(-1, 1, 3)
    A0 = O.eqf_state_matrix_A(X, Xi, O.IMUVelocity(0, omega, np.zeros(3)))
    B = O.eqf_input_matrix_B(X, Xi)
    C0 = O.eqf_output_matrix_C(Xi)
    g = oracle_lib.pack_group(X.A.q, X.A.x, X.w, [Q.q for Q in X.Q], [Q.a for Q in X.Q])
    s = oracle_lib.pack_state(Xi.pose.q, Xi.pose.x, Xi.velocity, Xi.p)
    A0c, Bc, C0c = oracle_lib.matrices(g, s, camq, camx, omega)
    assert np.abs(A0c - A0).max() < 1e-12 * max(1, np.abs(A0).max())
    assert np.abs(Bc - B).max() < 1e-12 * max(1, np.abs(B).max())
    assert np.abs(C0c - C0).max() < 1e-12
    if N < 2:
        return  # bundleLift's 4x4 normal equations (yaw + position) are rank deficient with one landmark
    M = rng.uniform(-1, 1, (5 + 3 * N, 5 + 3 * N))
    Sig = M @ M.T + np.eye(5 + 3 * N)
    base = rng.uniform(-1, 1, 5 + 3 * N)
    G = O.bundle_lift(base, Xi, X, Sig)
    Gc = oracle_lib.bundle_lift(g, s, camq, camx, base, Sig)
    assert np.abs(G - Gc).max() < 1e-9 * max(1, np.abs(G).max())


@pytest.mark.parametrize("name", ["stream_N5", "stream_N25", "churn_N12", "flags_continuous_N6", "flags_nolift_fast_N6"])
def test_cpp_oracle_reproduces_golden_vectors(oracle_lib, name):
    d, settings = load_golden(name)
    frames, f, internals = run_oracle_on_golden(oracle_lib, d, settings, capture=(1, 4))
    g = d["frames"]
    assert frames.shape == g.shape
    assert np.array_equal(frames[:, -1], g[:, -1])  # landmark counts frame by frame
    assert np.abs(frames[:, :16] - g[:, :16]).max() < 1e-8  # pose, velocity, bias
    assert np.abs(frames[:, 16] / g[:, 16] - 1).max() < 1e-8  # |Sigma|_F
    assert np.array_equal(f.ids(), d["final_ids"])
    assert rel_fro(f.stateCovariance(), d["final_sigma"]) < 1e-8
    for k, lu in internals.items():
        if f"delta_{k}" in d.files and lu is not None:
            assert np.abs(lu["delta"] - d[f"delta_{k}"]).max() < 1e-9
            assert np.abs(lu["gamma"] - d[f"gamma_{k}"]).max() < 1e-7 * max(1, np.abs(d[f"gamma_{k}"]).max())


def test_antipodal_level_start_raises_like_the_reference(oracle_lib):
    """A perfectly level first accelerometer sample makes the gravity chart singular: SO3FromVectors(-e3, e3)
    throws std::domain_error (libs/core/src/SO3.cpp:160) once the Riccati step runs."""
    f = oracle_lib.OracleFilter({})
    f.processIMUData(0.0, [0, 0, 0], [0, 0, 9.81])
    with pytest.raises(ValueError):
        f.processIMUData(0.005, [0, 0, 0], [0, 0, 9.81])
