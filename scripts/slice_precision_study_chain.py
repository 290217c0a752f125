#!/usr/bin/env python
"""Would the int8 slices also do for the FACTORISATION's products?  (round 6; CPU only, numpy restatement; the follow-up question of
scripts/slice_precision_study.py.)  The partitioned filter spends most of an update in the trailing updates of its two blocked Cholesky
factorisations (U_ki^T U_kj and the right-hand sides' U_ki^T Y_k), the same kind of product as the downdate.  Here the S-chain -- S = L L^T with
the right-hand side C Sigma, VIOFilter.cpp:276-277 -- is redone in 64-wide block rows with every trailing product formed as the int8 kernel
forms it (sliced_product: the columns' 7-bit slices, the slice pairs ta + tb < S, each pair exact), K, gamma and Sigma+ follow from that factor;
the E-chain (bundleLift's Sigma_e^-1) and the downdate itself stay exact.  Worst Sigma error / pose error against the unmodified restatement.
    python scripts/slice_precision_study_chain.py [N=200] [seconds=2] > profiles/r06_slice_precision_study_chain.txt"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

os.environ.setdefault("OPENBLAS_NUM_THREADS", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.linalg as sla  # noqa: E402

from eqf_vio_amd import synth  # noqa: E402
from oracle import eqf_numpy as O  # noqa: E402


def slices_of(X, S):
    mx = np.abs(X).max(axis=0)
    e = np.where(mx > 0, np.frexp(np.maximum(mx, 1e-300))[1], 0)
    r = X * np.ldexp(1.0, -e)
    sl, w = [], 64.0
    for _ in range(S):
        q = np.rint(r * w)
        sl.append(q / w)
        r = r - q / w
        w *= 128.0
    return sl, np.ldexp(1.0, e)


def sliced_product(A, B, S):
    """A^T B as k_i8_gemm forms it (A: k x m, B: k x n)."""
    if S is None:
        return A.T @ B
    sa, ea = slices_of(A, S)
    sb, eb = slices_of(B, S)
    G = np.zeros((A.shape[1], B.shape[1]))
    for ta in range(S):
        for tb in range(S - ta):
            G += sa[ta].T @ sb[tb]
    return G * np.outer(ea, eb)


def chain(Smat, Rhs, S, bs=64):
    """Blocked right-looking Cholesky by block rows with right-hand sides: returns Y = L^-1 Rhs and L; trailing products sliced."""
    A, B = Smat.copy(), Rhs.copy()
    n = A.shape[0]
    L = np.zeros_like(A)
    Y = np.zeros_like(B)
    for k0 in range(0, n, bs):
        k1 = min(k0 + bs, n)
        Lkk = np.linalg.cholesky(A[k0:k1, k0:k1])
        L[k0:k1, k0:k1] = Lkk
        Yk = sla.solve_triangular(Lkk, B[k0:k1], lower=True)
        Y[k0:k1] = Yk
        if k1 < n:
            U = sla.solve_triangular(Lkk, A[k0:k1, k1:], lower=True)
            L[k1:, k0:k1] = U.T
            A[k1:, k1:] -= sliced_product(U, U, S)
            B[k1:] -= sliced_product(U, Yk, S)
    return L, Y


def run(args):
    N, seconds, cs, dd = args
    ES = None
    if isinstance(cs, tuple):
        cs, ES = cs
    st = synth.make_stream(N, seed=1234, duration=seconds)
    d = synth.template_settings_dict()
    cx, cq = d.pop("cameraOffset_x"), d.pop("cameraOffset_q")
    s = O.Settings(**d)
    s.cameraOffset = O.SE3(cq, cx)
    f = O.VIOFilter(s)
    inv0 = np.linalg.inv

    def inv_chain(Smat):
        # (the restatement forms K = Sigma C^T S^-1 with this inverse: give it the inverse of the factor the sliced chain produces)
        if (cs is None and ES is None) or Smat.shape[0] < 128:
            return inv0(Smat)
        if Smat.shape[0] == 5 + 3 * N and ES is not None:  # Sigma_e inside bundleLift (EqFMatrices.cpp:239): the E-chain
            L, _ = chain(Smat, np.zeros((Smat.shape[0], 1)), ES)
            Li = sla.solve_triangular(L, np.eye(L.shape[0]), lower=True)
            return Li.T @ Li
        if Smat.shape[0] != 2 * N:
            return inv0(Smat)
        if cs is None:
            return inv0(Smat)
        L, _ = chain(Smat, np.zeros((Smat.shape[0], 1)), cs)
        Li = sla.solve_triangular(L, np.eye(L.shape[0]), lower=True)
        return Li.T @ Li

    out = []
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]
            f.processIMUData(O.IMUVelocity(r[0], r[1:4], r[4:7]))
        else:
            O.np.linalg.inv = inv_chain
            try:
                f.processVisionData(st.vision_stamps[k], st.ids, st.bearings[k])
            finally:
                O.np.linalg.inv = inv0
            if cs is not None and f.last and f.last["Sigma_prior"].shape == f.Sigma.shape:
                # the covariance as the device forms it: Sigma - Y^T Y with Y = L^-1 (C Sigma) OUT OF THE SLICED CHAIN (its right-hand-side
                # products U_ki^T Y_k sliced as well); the product Y^T Y itself exact here (dd = None) or sliced too (dd = slices)
                Sp, S_, K = f.last["Sigma_prior"], f.last["S"], f.last["K"]
                _, Y = chain(S_, (K @ S_).T, cs)
                f.Sigma = Sp - sliced_product(Y, Y, dd)
            e = f.stateEstimate()
            out.append((f.Sigma.copy(), np.concatenate([e.pose.q, e.pose.x])))
    return ((cs, ES) if ES is not None else cs, dd), out


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    variants = [(None, None), (5, None), (6, None), (7, None), (6, 6), (7, 6), ((None, 5), None), ((None, 6), None), ((5, 5), 6), ((6, 6), 6)]
    with ProcessPoolExecutor(max_workers=4) as ex:
        res = dict(ex.map(run, [(N, seconds, cs, dd) for cs, dd in variants]))
    ref = res[(None, None)]
    print(f"# N = {N}, {seconds} s of the bench stream ({len(ref)} vision updates): the S-chain (Cholesky of S = C Sigma C^T + R in 64-wide block rows) with its")
    print("# trailing products U_ki^T U_kj and U_ki^T Y_k formed as the int8 kernel forms them; K from that factor, Sigma+ = Sigma - Y^T Y with that Y.")
    for key in variants[1:]:
        errs = [np.linalg.norm(a[0] - r[0]) / np.linalg.norm(r[0]) for a, r in zip(res[key], ref)]
        pose = max(np.abs(a[1] - r[1]).max() for a, r in zip(res[key], ref))
        print(f"  S-chain / E-chain products from {key[0]} slices (None = exact), downdate {'exact' if key[1] is None else 'from %d slices' % key[1]}: worst Sigma rel-Frobenius {max(errs):.3e} at frame {int(np.argmax(errs))}, last {errs[-1]:.3e}; worst pose component difference {pose:.2e}; frames over 1e-4: {sum(e > 1e-4 for e in errs)} / {len(errs)}")
