#!/usr/bin/env python
"""How many bits does the covariance downdate's product need for the filter to hold 1e-4 on Sigma?  (round 6; CPU only, numpy oracle.)

DESIGN.md section 2 showed that fp32 fails twice: 24 bits of STORAGE lose it and 24 bits of ACCUMULATION lose it.  The exact-integer route
(scripts/micro/i8_split_gemm.hip: every column of Y scaled by a power of two and cut into S signed 7-bit slices, slice pairs multiplied on the
int8 matrix pipe with exact int32 accumulation) has neither problem -- Sigma stays fp64 in memory, the accumulation is exact -- its only error is
the truncation of Y's entries to 6 + 7 (S - 1) bits relative to their column's largest entry.  This script puts exactly that error into the fp64
numpy restatement of the reference (oracle/eqf_numpy.py: test infrastructure, which is what a numerical study is) and runs the bench stream:

    Sigma+ = Sigma - Yq^T Yq,   Y = L^-1 (C Sigma),  S = L L^T  (VIOFilter.cpp:276-297 in Cholesky form),  Yq = Y with every column truncated

for S = 4 .. 7 slices (27 / 34 / 41 / 48 bits) and, as the calibration point, 24 bits -- and then with the KERNEL's own arithmetic (the slices,
and only the slice pairs ta + tb < S: the dropped pairs are of the truncation's size but add up coherently over correlated columns); against
the unmodified restatement, worst relative Frobenius error of Sigma over the run.      python scripts/slice_precision_study.py [N=200] [seconds=10] > profiles/r06_slice_precision_study.txt
"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

os.environ.setdefault("OPENBLAS_NUM_THREADS", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.linalg as sla  # noqa: E402

from eqf_vio_amd import synth  # noqa: E402
from oracle import eqf_numpy as O  # noqa: E402


def quantise_columns(Y, bits):
    """Every column scaled by the power of two that bounds it, entries rounded to `bits` fractional bits (what S slices keep), scaled back."""
    mx = np.abs(Y).max(axis=0)
    e = np.where(mx > 0, np.frexp(np.maximum(mx, 1e-300))[1], 0)
    sc = np.ldexp(1.0, bits - e)
    return np.rint(Y * sc) / sc


def sliced_gram(Y, S, extra=0):
    """Y^T Y exactly as k_i8_gemm forms it: the columns' slices (first 6 bits, then 7 each, round to nearest), the slice pairs (ta, tb) with
    ta + tb < S + extra only, every pair's product exact.  extra = 0 is the kernel; the pairs it drops are products of LOWER slices, of the size
    of the truncation itself -- and unlike the truncation they do not average out when the columns are correlated."""
    mx = np.abs(Y).max(axis=0)
    e = np.where(mx > 0, np.frexp(np.maximum(mx, 1e-300))[1], 0)
    r = Y * np.ldexp(1.0, -e)
    sl, w = [], 64.0
    for _ in range(S):
        q = np.rint(r * w)
        sl.append(q / w)
        r = r - q / w
        w *= 128.0
    G = np.zeros((Y.shape[1], Y.shape[1]))
    for ta in range(S):
        for tb in range(S):
            if ta + tb < S + extra:
                G += sl[ta].T @ sl[tb]
    sc = np.ldexp(1.0, e)
    return G * np.outer(sc, sc)


def run(args):
    N, seconds, bits = args
    st = synth.make_stream(N, seed=1234, duration=seconds)
    d = synth.template_settings_dict()
    cx, cq = d.pop("cameraOffset_x"), d.pop("cameraOffset_q")
    s = O.Settings(**d)
    s.cameraOffset = O.SE3(cq, cx)
    f = O.VIOFilter(s)
    fro = []
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]
            f.processIMUData(O.IMUVelocity(r[0], r[1:4], r[4:7]))
        else:
            f.processVisionData(st.vision_stamps[k], st.ids, st.bearings[k])
            if f.last and bits is not None:
                # redo the covariance part of the update from the operands the call kept: C Sigma = (K S)^T, in Cholesky form, truncated
                Sp, S_, K = f.last["Sigma_prior"], f.last["S"], f.last["K"]
                if Sp.shape == f.Sigma.shape:
                    L = np.linalg.cholesky(S_)
                    Y = sla.solve_triangular(L, (K @ S_).T, lower=True)
                    if isinstance(bits, tuple):  # ("pairs", S, extra): the kernel's own arithmetic
                        f.Sigma = Sp - sliced_gram(Y, bits[1], bits[2])
                    else:
                        Yq = quantise_columns(Y, bits)
                        f.Sigma = Sp - Yq.T @ Yq
            fro.append(f.Sigma.copy() if bits is None else f.Sigma)
    return bits, fro


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    variants = [None, 24, 27, 34, 41, 48, ("pairs", 5, 0), ("pairs", 5, 1), ("pairs", 6, 0), ("pairs", 7, 0)]
    with ProcessPoolExecutor(max_workers=3) as ex:
        out = dict(ex.map(run, [(N, seconds, b) for b in variants]))
    ref = out[None]
    print(f"# N = {N}, {seconds} s of the bench stream ({len(ref)} vision updates), template settings; oracle/eqf_numpy.py (dense fp64, the reference's")
    print("# operation order) with the covariance downdate redone as Sigma - Yq^T Yq, Y's columns truncated to `bits` below their largest entry.")
    print("# worst relative Frobenius error of Sigma against the unmodified restatement (north_star tolerance 1e-4):")
    for b in variants[1:]:
        errs = [np.linalg.norm(a - r) / np.linalg.norm(r) for a, r in zip(out[b], ref)]
        worst = int(np.argmax(errs))
        if isinstance(b, tuple):
            S_, ex = b[1], b[2]
            npairs = sum(1 for ta in range(S_) for tb in range(S_) if ta + tb < S_ + ex)
            what = f"the kernel: S = {S_} slices, pairs ta + tb < {S_ + ex}: {npairs} products"
            print(f"  {what:62s}  worst {max(errs):.3e} at frame {worst:3d}   last frame {errs[-1]:.3e}   frames over 1e-4: {sum(e > 1e-4 for e in errs)} / {len(errs)}")
            continue
        what = {24: "an fp32 significand (calibration)", 27: "S = 4 slices", 34: "S = 5 slices", 41: "S = 6 slices", 48: "S = 7 slices"}[b]
        what = "truncation only, " + what
        print(f"  {b:2d} bits ({what:55s})  worst {max(errs):.3e} at frame {worst:3d}   last frame {errs[-1]:.3e}   frames over 1e-4: {sum(e > 1e-4 for e in errs)} / {len(errs)}")
    # (bits = "all": the Cholesky form itself against the gain form)
