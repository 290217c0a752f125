"""BASELINE configs[4] / SURVEY.md 8(e) row 2 on CPU: the exchange schedule of the partitioned filter as tests/tiled_reference.py spells it out (the product's loop is its C++ twin, csrc/eqf_tiledf.hip, compared with it bit for bit on the GPU) -- block-cyclic local matrices, the
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

    import tiled_reference as tiled
    from eqf_vio_amd import synth
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
    from tiled_reference import BlockCyclic

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


def _churn_worker(rank, world, port, Pr, Pc, N, bl, cap, out_dir):
    """closed loop with landmarks entering, leaving and failing the outlier gate at the reference's default threshold"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import tiled_reference as tiled
    from eqf_vio_amd import synth
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
    st = synth.make_stream(N, duration=0.66)
    meas = synth.churn_measurements(st, seed=11, outlier_frames=(4, 7, 8), outlier_angle=0.05)
    d = synth.template_settings_dict()
    d["outlierThreshold"] = 0.01  # include/eqf_vio/VIOFilterSettings.h default
    tf = tiled.TiledFilter(grid, NumpyBackend(d, cap), bl, capacity=cap)
    fo = ob.OracleFilter(d)
    worst = dict(S=0.0, pose=0.0, gamma=0.0, Gamma=0.0, delta=0.0, hole=0.0)
    n_upd, sizes = 0, set()
    rel = lambda A, B: float(np.linalg.norm(A - B) / np.linalg.norm(B))
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]
            fo.processIMUData(r[0], r[1:4], r[4:7])
            tf.processIMUData(r[0], r[1:4], r[4:7])
            continue
        ids, y = meas[k]
        fo.processVisionData(st.vision_stamps[k], ids, y)
        assert tf.processVisionData(st.vision_stamps[k], ids, y) == 0
        n_upd += 1
        assert np.array_equal(tf.ids, fo.ids()), (k, tf.ids, fo.ids())  # same landmarks, same ORDER as the reference's X.id
        sizes.add(len(tf.ids))
        worst["S"] = max(worst["S"], rel(tf.stateCovariance(), fo.stateCovariance()))
        eo, et = fo.stateEstimate(), tf.stateEstimate()
        worst["pose"] = max(worst["pose"], float(np.abs(eo["x"] - et["x"]).max()), float(np.abs(eo["q"] - et["q"]).max()),
                            float(np.abs(eo["p"] - et["p"]).max()), float(np.abs(fo.bias() - tf.bias()).max()))
        lo, lt = fo.last_update(), tf.lastUpdate()
        for key in ("delta", "gamma", "Gamma"):
            worst[key] = max(worst[key], float(np.abs(lo[key] - lt[key]).max() / max(1.0, np.abs(lo[key]).max())))
        # the holes: exact zeros off their diagonal block, which stays a multiple of the identity
        Sfull = tf.slotCovariance()
        for s_ in np.nonzero(~tf.taken[: tf.nslots])[0]:
            blk = Sfull[11 + 3 * s_: 14 + 3 * s_].copy()
            dg = blk[:, 11 + 3 * s_: 14 + 3 * s_].copy()
            blk[:, 11 + 3 * s_: 14 + 3 * s_] = 0.0
            worst["hole"] = max(worst["hole"], float(np.abs(blk).max()), float(np.abs(dg - dg[0, 0] * np.eye(3)).max()))
            assert dg[0, 0] >= 1.0
    cs = tf.churn_stats
    np.save(os.path.join(out_dir, f"churn_{rank}.npy"), np.array([worst["S"], worst["pose"], worst["gamma"], worst["Gamma"], worst["delta"], worst["hole"],
                                                                  n_upd, cs["removed_old"], cs["removed_outliers"], cs["added"], len(sizes), tf.nslots]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("Pr,Pc,N,bl,cap", [(1, 1, 30, 4, 30), (2, 2, 30, 4, 32), (1, 2, 22, 3, 24)])
def test_tiled_filter_landmark_churn_and_outlier_gate_match_the_oracle(tmp_path, Pr, Pc, N, bl, cap):
    """removeOldLandmarks / removeOutliers / addNewLandmarks (VIOFilter.cpp:345-443) in the partitioned filter: landmarks enter and leave
    the field of view, three frames carry a bearing 0.05 rad off (gate at the reference default 0.01).  After every frame: the same ids in
    the same order as the oracle's state, covariance / state / bias / innovation against the oracle, and the inactive slots decoupled."""
    world = Pr * Pc
    if world == 1:
        _churn_worker(0, 1, 0, Pr, Pc, N, bl, cap, str(tmp_path))
    else:
        mp.spawn(_churn_worker, args=(world, _free_port(), Pr, Pc, N, bl, cap, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        S, pose, gamma, Gamma, delta, hole, n_upd, rem_old, rem_out, added, nsizes, nslots = np.load(tmp_path / f"churn_{r}.npy")
        assert n_upd >= 12
        assert rem_old >= 3 and rem_out >= 2 and added > N // 3 and nsizes >= 3, (rem_old, rem_out, added, nsizes)
        assert nslots <= cap
        assert S < 1e-9 and pose < 1e-9, (r, S, pose)
        assert delta < 1e-11 and gamma < 1e-8 and Gamma < 1e-8, (r, delta, gamma, Gamma)
        assert hole == 0.0


def test_tiled_filter_churn_argument_and_capacity_errors():
    """more landmarks in view than slots: refused loudly before any effect; a frame without bearings removes every landmark and
    reports EQF_SKIPPED_NO_BEARINGS like the reference (VIOFilter.cpp:242, :258-259)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tiled_reference as tiled
    from eqf_vio_amd import synth
    from oracle import binding as ob
    from tiled_double import NumpyBackend

    N = 10
    st = synth.make_stream(N, duration=0.16)
    d = synth.template_settings_dict()
    tf = tiled.TiledFilter(tiled.ProcessGrid(None, 1, 1), NumpyBackend(d, 8), 4, capacity=8)
    fo = ob.OracleFilter(d)
    frames = 0
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]
            fo.processIMUData(r[0], r[1:4], r[4:7])
            tf.processIMUData(r[0], r[1:4], r[4:7])
            continue
        if frames == 0:
            with pytest.raises(RuntimeError):
                tf.processVisionData(st.vision_stamps[k], st.ids, st.bearings[k])  # 10 landmarks, 8 slots
            assert tf.ids is None
        elif frames == 1:
            fo.processVisionData(st.vision_stamps[k], st.ids[:8], st.bearings[k][:8])
            assert tf.processVisionData(st.vision_stamps[k], st.ids[:8], st.bearings[k][:8]) == 0
            # (the first update after initialPointVariance = 5000 on 8 landmarks: the worst-conditioned step of a run)
            assert np.linalg.norm(tf.stateCovariance() - fo.stateCovariance()) / np.linalg.norm(fo.stateCovariance()) < 1e-7
        elif frames == 2:
            fo.processVisionData(st.vision_stamps[k], st.ids[:0], st.bearings[k][:0])
            assert tf.processVisionData(st.vision_stamps[k], st.ids[:0], st.bearings[k][:0]) == 4
            assert len(tf.ids) == 0 and fo.N == 0 and tf.nslots == 1 and not tf.taken.any()
        frames += 1
    assert frames == 3
