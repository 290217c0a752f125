"""The tile kernels of cfg 5 (csrc/eqf_tile.hpp: eqf_tile_propagate, eqf_tile_downdate, eqf_tile_potrf, eqf_tile_trsm) on the MI355X, through the C ABI on torch
tensors: against dense formulas on random tiles, and end to end -- the tiled Sigma of eqf_vio_amd/tiled.py on the GPU with
these kernels, driven by the oracle's linearisation blocks, against the oracle's Sigma after every call (the multi-rank exchange
schedule itself is validated on CPU with gloo, tests/test_tiled.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tile_propagate_and_downdate_against_dense_formulas():
    import torch

    from eqf_vio_amd import tiled

    dev = torch.device("cuda", 0)
    k = tiled.TileKernels(0)
    rng = np.random.default_rng(3)
    N = 45  # rows 0..19 against columns 20..44: a rectangular tile with ragged 16-landmark workgroups
    n = 11 + 3 * N
    M = rng.standard_normal((n, n))
    S = M @ M.T + n * np.eye(n)
    F = np.eye(n)
    F[:11, :11] += 0.01 * rng.standard_normal((11, 11))
    F[11:, :11] = 0.02 * rng.standard_normal((3 * N, 11))
    D = np.stack([np.eye(3) + 0.01 * rng.standard_normal((3, 3)) for _ in range(N)])
    for i in range(N):
        F[11 + 3 * i:14 + 3 * i, 11 + 3 * i:14 + 3 * i] = D[i]
    Bn = np.zeros((n, 6))
    Bn[6:] = rng.standard_normal((n - 6, 6))
    R6 = np.array([1e-4, 1e-4, 1e-4, 2e-4, 2e-4, 2e-4])
    T, pv = 0.005, 0.001
    P = np.concatenate([np.full(11, 0.01), np.full(3 * N, pv)])
    ref = F @ S @ F.T + T * (np.diag(P) + (Bn * R6) @ Bn.T)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Sd, Dd, Ld, Bd = t(S), t(D), t(F[11:, :11]), t(Bn)
    Sbb, Sb = Sd[:11, :11].contiguous(), Sd[:11, 11:].contiguous()
    for (i0, ni, j0, nj) in ((0, 20, 20, 25), (20, 25, 20, 25), (25, 20, 0, 25)):  # off-diagonal, diagonal, below the diagonal
        tile = Sd[11 + 3 * i0:11 + 3 * (i0 + ni), 11 + 3 * j0:11 + 3 * (j0 + nj)].contiguous()
        out = k.propagate(tile, ni, nj, Dd[i0:], Ld[3 * i0:], Dd[j0:], Ld[3 * j0:], Sbb, Sb[:, 3 * i0:], Sb.stride(0), Sb[:, 3 * j0:], Sb.stride(0),
                          Bd[11 + 3 * i0:], Bd[11 + 3 * j0:], R6, T, T * pv, i0 == j0 and ni == nj)
        want = ref[11 + 3 * i0:11 + 3 * (i0 + ni), 11 + 3 * j0:11 + 3 * (j0 + nj)]
        assert np.abs(out.cpu().numpy() - want).max() <= 1e-12 * np.abs(want).max(), (i0, j0)
    # downdate: sizes that are not multiples of the 64 x 64 x 32 tiling
    for (m, n2, kk) in ((64, 64, 32), (75, 130, 50), (12, 12, 7), (200, 96, 448)):
        A, B, C = rng.standard_normal((kk, m)), rng.standard_normal((kk, n2)), rng.standard_normal((m, n2))
        Cd = t(C)
        k.downdate(Cd, t(A), t(B))
        want = C - A.T @ B
        assert np.abs(Cd.cpu().numpy() - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), (m, n2, kk)


@pytest.mark.parametrize("n", [16, 64, 96, 200, 384])
def test_tile_potrf_and_trsm_against_lapack(n):
    """eqf_tile_potrf / eqf_tile_trsm (the panel operations of the distributed factorisation, csrc/eqf_tile.hpp) against numpy's
    Cholesky and triangular solves: block sizes below, at and between multiples of the 64-wide block column."""
    import torch

    from eqf_vio_amd import tiled

    dev = torch.device("cuda", 0)
    k = tiled.TileKernels(0)
    rng = np.random.default_rng(100 + n)
    M = rng.standard_normal((n, n))
    A = M @ M.T + n * np.eye(n)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    L, drec = k.potrf(t(A))
    torch.cuda.synchronize()
    assert int(k._info.item()) == 0
    Lref = np.linalg.cholesky(A)
    Lg = L.cpu().numpy()
    assert np.abs(np.triu(Lg, 1)).max() == 0.0
    assert np.abs(Lg - Lref).max() <= 1e-12 * np.abs(Lref).max()
    for m in (1, 64, 70, 150):
        B = rng.standard_normal((m, n))
        X = k.trsm(L, drec, t(B), True).cpu().numpy()          # B L^-T
        want = np.linalg.solve(Lref, B.T).T
        assert np.abs(X - want).max() <= 1e-11 * max(1.0, np.abs(want).max()), (n, m, "right")
        B2 = rng.standard_normal((n, m))
        Y = k.trsm(L, drec, t(B2), False).cpu().numpy()        # L^-1 B
        want = np.linalg.solve(Lref, B2)
        assert np.abs(Y - want).max() <= 1e-11 * max(1.0, np.abs(want).max()), (n, m, "left")
    # a matrix that is not positive definite raises the flag instead of producing NaNs silently
    Abad = A.copy()
    Abad[n // 2, n // 2] = -1.0
    k.potrf(t(Abad))
    torch.cuda.synchronize()
    assert int(k._info.item()) == 1


def test_tiled_sigma_on_the_gpu_with_the_tile_kernels(oracle_lib):
    """One rank (1 x 1 grid) on the GPU: Riccati steps, downdates and the panel operations (diagonal-block Cholesky, triangular solves) through eqf_tile_*; Sigma
    against the oracle after every IMU / vision call of a short stream, open loop as in tests/test_tiled.py."""
    import torch

    from eqf_vio_amd import synth, tiled

    ob = oracle_lib
    dev = torch.device("cuda", 0)
    N, bl = 48, 16
    grid = tiled.ProcessGrid(None, 1, 1, device=dev, kernels=tiled.TileKernels(0))
    st = synth.make_stream(N, duration=0.21)
    d = synth.template_settings_dict()
    fo = ob.OracleFilter(d)
    n = 11 + 3 * N
    Rdiag = torch.tensor([d["velOmegaVariance"]] * 3 + [d["velAccelVariance"]] * 3, dtype=torch.float64, device=dev)
    Pb = torch.tensor([d["biasOmegaProcessVariance"]] * 3 + [d["biasAccelProcessVariance"]] * 3 + [d["gravityProcessVariance"]] * 2
                      + [d["velocityProcessVariance"]] * 3, dtype=torch.float64, device=dev)

    def inputs(stamp, omega):
        T = stamp - fo.getTime()
        g_, x_ = fo.group(), fo.xi0()
        A0, Bm, C0 = ob.matrices(ob.pack_group(g_["Aq"], g_["Ax"], g_["w"], g_["Qq"], g_["Qa"]), ob.pack_state(x_["q"], x_["x"], x_["v"], x_["p"]),
                                 d["cameraOffset_q"], d["cameraOffset_x"], omega)
        Ab = np.zeros((n, n))
        Ab[6:, 6:] = A0
        Ab[6:, :6] = -Bm
        F = torch.from_numpy(np.eye(n) + T * Ab).to(dev)
        Bn = torch.zeros((n, 6), dtype=torch.float64, device=dev)
        Bn[6:] = torch.from_numpy(Bm).to(dev)
        Dblk = torch.stack([F[11 + 3 * i:14 + 3 * i, 11 + 3 * i:14 + 3 * i] for i in range(N)]).contiguous()
        Qbb = T * (torch.diag(Pb) + (Bn[:11] * Rdiag) @ Bn[:11].T)
        Cblk = torch.stack([torch.from_numpy(C0[2 * i:2 * i + 2, 5 + 3 * i:8 + 3 * i].copy()) for i in range(N)]).to(dev)
        return T, F[:11, :11].contiguous(), F[11:, :11].contiguous(), Dblk, Qbb, Bn, Cblk

    rel = lambda A, B: float(np.linalg.norm(A - B) / np.linalg.norm(B))
    ts, cur_w, worst, n_upd = None, np.zeros(3), 0.0, 0
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]
            if ts is not None:
                T, Fbb, L, Dblk, Qbb, Bn, _ = inputs(r[0], cur_w)
            bias = fo.bias()
            fo.processIMUData(r[0], r[1:4], r[4:7])
            cur_w = r[1:4] - bias[:3]
            if ts is not None:
                tiled.propagate(ts, Fbb, L, Dblk, Qbb, Bn, Rdiag, T, d["pointProcessVariance"])
                worst = max(worst, rel(ts.to_dense().cpu().numpy(), fo.stateCovariance()))
        else:
            stamp = st.vision_stamps[k]
            if ts is not None:
                T, Fbb, L, Dblk, Qbb, Bn, Cblk = inputs(stamp, cur_w)
            fo.processVisionData(stamp, st.ids, st.bearings[k])
            if ts is None:
                ts = tiled.TiledSigma.from_dense(grid, fo.stateCovariance(), bl)
                continue
            lu = fo.last_update()
            tiled.propagate(ts, Fbb, L, Dblk, Qbb, Bn, Rdiag, T, d["pointProcessVariance"])
            gamma = tiled.update(ts, Cblk, lu["delta"], d["measurementVariance"]).cpu().numpy()
            assert np.abs(gamma - lu["gamma"]).max() < 1e-8 * max(1.0, np.abs(lu["gamma"]).max())
            worst = max(worst, rel(ts.to_dense().cpu().numpy(), fo.stateCovariance()))
            n_upd += 1
    assert n_upd >= 3 and worst < 1e-9, worst
