"""BASELINE configs[4] / SURVEY.md 8(e) row 2 on CPU: the exchange schedule of eqf_vio_amd/tiled.py -- block-cyclic local matrices, the
two distributed factorisations with their row / column-restricted broadcasts, the downdate and the reductions -- CLOSED LOOP over gloo on
1 x 1, 1 x 2, 2 x 2 and 2 x 4 process grids (the grid of one 8-GPU node), whole and ragged landmark blocks.  Every rank runs
TiledFilter.processIMUData / processVisionData; gamma, the innovation lift and the state come out of the distributed quantities (nothing
is taken from a reference filter inside the loop).  The per-rank mathematics is the CPU test double of the HIP backend
(tests/tiled_double.py); the checker is the single-process dense fp64 oracle on the same stream."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, Pr, Pc, N, bl, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from eqf_vio_amd import synth, tiled
    from oracle import binding as ob
    from tiled_double import NumpyBackend

    torch.set_num_threads(1)
    dist_ = None
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist_ = dist
    grid = tiled.ProcessGrid(dist_, Pr, Pc)
    st = synth.make_stream(N, duration=0.26)
    d = synth.template_settings_dict()
    tf = tiled.TiledFilter(grid, NumpyBackend(d, N), bl)
    fo = ob.OracleFilter(d)  # the checker: dense fp64, reference op order
    worst = dict(S=0.0, pose=0.0, gamma=0.0, Gamma=0.0, delta=0.0)
    n_upd = 0
    rel = lambda A, B: float(np.linalg.norm(A - B) / np.linalg.norm(B))
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]
            fo.processIMUData(r[0], r[1:4], r[4:7])
            tf.processIMUData(r[0], r[1:4], r[4:7])
        else:
            stamp = st.vision_stamps[k]
            fo.processVisionData(stamp, st.ids, st.bearings[k])
            assert tf.processVisionData(stamp, st.ids, st.bearings[k]) == 0
            n_upd += 1
            worst["S"] = max(worst["S"], rel(tf.stateCovariance(), fo.stateCovariance()))
            eo, et = fo.stateEstimate(), tf.stateEstimate()
            worst["pose"] = max(worst["pose"], float(np.abs(eo["x"] - et["x"]).max()), float(np.abs(eo["q"] - et["q"]).max()),
                                float(np.abs(eo["p"] - et["p"]).max()), float(np.abs(fo.bias() - tf.be.bias()).max()))
            lo, lt = fo.last_update(), tf.be.last_update()
            for key in ("delta", "gamma", "Gamma"):
                worst[key] = max(worst[key], float(np.abs(lo[key] - lt[key]).max() / max(1.0, np.abs(lo[key]).max())))
    owned = tf.geo.nlr * tf.geo.nlc
    np.save(os.path.join(out_dir, f"worst_{rank}.npy"), np.array([worst["S"], worst["pose"], worst["gamma"], worst["Gamma"], worst["delta"], n_upd, owned]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("Pr,Pc,N,bl", [(1, 1, 26, 4), (1, 2, 24, 4), (2, 2, 26, 4), (2, 4, 34, 4)])
def test_tiled_filter_closed_loop_over_a_process_grid_matches_the_single_process_oracle(tmp_path, Pr, Pc, N, bl):
    world = Pr * Pc
    if world == 1:
        _worker(0, 1, 0, Pr, Pc, N, bl, str(tmp_path))
    else:
        mp.spawn(_worker, args=(world, _free_port(), Pr, Pc, N, bl, str(tmp_path)), nprocs=world, join=True)
    owned = 0
    for r in range(world):
        S, pose, gamma, Gamma, delta, n_upd, own = np.load(tmp_path / f"worst_{r}.npy")
        assert n_upd >= 5
        # closed loop over 5 updates + 50 Riccati steps: rounding amplified by cond(Sigma) ~ 1e7, as between any two fp64 evaluations
        assert S < 1e-9 and pose < 1e-9, (r, S, pose)
        assert delta < 1e-11 and gamma < 1e-8 and Gamma < 1e-8, (r, delta, gamma, Gamma)
        owned += own
    assert owned == N * N  # every landmark pair has exactly one owner


def test_block_cyclic_geometry():
    from eqf_vio_amd.tiled import BlockCyclic

    N, bl, Pr, Pc = 34, 4, 2, 4
    seen = np.zeros((N, N), dtype=int)
    for pr in range(Pr):
        for pc in range(Pc):
            g = BlockCyclic(N, bl, Pr, Pc, pr, pc)
            seen[np.ix_(g.rowMap, g.colMap)] += 1
            assert g.nlc == g.ncols_of(pc)
            # only the globally last block is ragged, and it is the last local block of whoever owns it
            assert all(g.block_size(b) == bl for b in g.row_blocks[:-1]) and all(g.block_size(b) == bl for b in g.col_blocks[:-1])
    assert (seen == 1).all()
    assert [BlockCyclic.blocks_upto(k, 1, 4) for k in range(10)] == [0, 1, 1, 1, 1, 2, 2, 2, 2, 3]


def test_tiled_filter_evaluates_the_outlier_gate_and_refuses_a_frame_that_trips_it():
    """removeOutliers (VIOFilter.cpp:429-443) at the reference's default threshold 0.01: on a clean stream the gate is evaluated every
    frame and changes nothing (the oracle removes nothing either); a corrupted bearing is refused loudly -- the fixed-landmark-set filter
    must not fuse what the reference would have dropped."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from eqf_vio_amd import synth, tiled
    from oracle import binding as ob
    from tiled_double import NumpyBackend

    N = 14
    st = synth.make_stream(N, duration=0.26)
    d = synth.template_settings_dict()
    d["outlierThreshold"] = 0.01  # include/eqf_vio/VIOFilterSettings.h default
    tf = tiled.TiledFilter(tiled.ProcessGrid(None, 1, 1), NumpyBackend(d, N), 4)
    fo = ob.OracleFilter(d)
    frames = 0
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]
            fo.processIMUData(r[0], r[1:4], r[4:7])
            tf.processIMUData(r[0], r[1:4], r[4:7])
        else:
            y = st.bearings[k].copy()
            if frames == 3:
                bad = y.copy()
                bad[5] = np.array([np.sin(0.3), 0.0, np.cos(0.3)])  # 0.3 rad off: far beyond the gate
                with pytest.raises(tiled.TiledOutlierError):
                    tf.processVisionData(st.vision_stamps[k], st.ids, bad)
                break
            fo.processVisionData(st.vision_stamps[k], st.ids, y)
            assert tf.processVisionData(st.vision_stamps[k], st.ids, y) == 0
            assert fo.N == N  # the oracle's gate removed nothing
            assert np.linalg.norm(tf.stateCovariance() - fo.stateCovariance()) / np.linalg.norm(fo.stateCovariance()) < 1e-9
            frames += 1
    assert frames == 3
