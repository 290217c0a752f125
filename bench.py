#!/usr/bin/env python
"""bench.py -- EqF propagate+update steps/s on MI355X (BASELINE.json metric).

A "step" is one filter API call (processIMUData = Riccati propagate; processVisionData = propagate + update)
on the synthetic stream of SURVEY.md section 8d (IMU 200 Hz, bearings 20 Hz, N landmarks, template settings,
fastRiccati = false).  The whole input stream is resident in HBM before the timed region
(eqf_stream_upload); the timed region replays events by index through the C ABI.

    python bench.py                       # 1 GPU, 1 filter, N = 200  (BASELINE configs[1])
    python bench.py --filters-per-gpu 64  # 64 independent filters batched on one GPU
    python bench.py --gpus 8              # starts 8 ranks ITSELF (one per GPU, RCCL rendezvous on 127.0.0.1), one JSON line from rank 0;
                                          # BASELINE configs[3] (64 filters over the 8 GPUs) is its `batch64_strong` leg
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8   # same job

Multi-GPU: independent filters are sharded over ranks (no data-path collective); RCCL is used once to
scatter the pre-generated input streams from rank 0 and once to gather the results.

The JSON line's `value` is the WEAK-scaling figure (--filters-per-gpu filters on every GPU, default 1 = BASELINE configs[1]
at N = 1).  BASELINE configs[3] / north_star's "64-instance batch" is a STRONG-scaling workload (64 filters in total, 64/G per
GPU); it is measured in the same run as a second leg and reported under "batch64_strong" (--no-batch64 skips it).
"""
import argparse
import json
import os
import sys
import time

# The GPU hosts run this under a cgroup CPU quota (16 CPUs on a 256-core machine).  numpy's OpenBLAS starts 64 threads that keep
# spinning after every call; that burns the quota and the kernel throttles the whole process for tens of milliseconds -- seen as
# random 45-65 ms stalls in the timed region (late launches, a late return from the final wait).  Nothing here needs threaded BLAS.
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TF = {"f64": 78.6, "f32": 157.3}  # f32: MI355X_MICROARCH.md; f64: half the f32 rate (AMD datasheet)
VALU_F64_PEAK_TF = 78.6  # vector fp64 FMA rate of the part (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz): what the structured Riccati step runs on


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2200)  # 10 s of stream: 2000 IMU + 200 vision
    ap.add_argument("--warmup", type=int, default=220)  # 1 s
    ap.add_argument("--landmarks", type=int, default=200)
    ap.add_argument("--filters-per-gpu", type=int, default=1)
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--dense-propagate", action="store_true", help="Riccati step as dense F Sigma F^T on MFMA (BASELINE cfg 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch64", action="store_true", help="skip the strong-scaling leg (64 filters in total over the GPUs)")
    ap.add_argument("--batch64-steps", type=int, default=440)
    ap.add_argument("--no-steady-state", action="store_true", help="skip the 2200-step leg that accompanies a short --steps run")
    ap.add_argument("--no-tiled", action="store_true", help="skip the cfg 5 leg: one N = --tiled-landmarks filter with Sigma 2-D block-partitioned "
                    "over the ranks of the job (1 x 1 grid on one GPU, 2 x 4 on eight), closed loop through the C++ host loop eqf_tf_* (csrc/eqf_tiledf.hip)")
    ap.add_argument("--tiled", action="store_true", help="(the cfg 5 leg is on by default; kept for explicitness)")
    ap.add_argument("--tiled-landmarks", type=int, default=4000)
    ap.add_argument("--tiled-block", type=int, default=0, help="landmarks per block of the 2-D partition (0: 320 on one GPU -- measured best there, "
                    "profiles/r06_tiled_block_sweep.txt -- and 250 on a grid, where N = 4000 then is 16 blocks: an even deal over 2 x 4)")
    ap.add_argument("--tiled-timeout", type=int, default=240, help="several GPUs: seconds after which the cfg 5 leg is given up")
    ap.add_argument("--tiled-frames", type=int, default=3, help="timed frames (a frame = 10 IMU calls + 1 vision call) after one warm-up frame")
    ap.add_argument("--no-i8-downdate", action="store_true", help="skip the cfg 5 sub-leg with the covariance downdate on the integer matrix pipe")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle-checked prefix (parity block of the JSON line)")
    ap.add_argument("--no-n1000", action="store_true", help="skip the BASELINE cfg 3 leg (one filter of N = 1000, 220 steps, with its own roofline)")
    ap.add_argument("--no-batch8", action="store_true", help="skip the 8-filters-per-GPU leg (one GPU's share of the 64-filter batch on an 8-GPU node)")
    ap.add_argument("--no-churn", action="store_true", help="skip the landmark-churn + outlier-gate leg (per-call API, N ~ 200)")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the untimed throw-away run that precedes the measured job")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--debug-option", action="append", default=[], metavar="NAME=VALUE", help="developer toggle of the library (eqf_debug_option), main job only")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--spawn-selftest", action="store_true", help="launcher check without a GPU: the ranks rendezvous over gloo, scatter the "
                    "input streams and gather a result; no filter runs and the line says so (tests/test_bench_spawn.py)")
    return ap.parse_args()


def run_events(fb, events):
    for kind, k in events:
        if kind == "imu":
            fb.stream_imu(k)
        else:
            fb.stream_vision(k)


def cpu_baseline(N, structured=False, budget_s=15.0):
    """The fp64 C++ oracle on ONE host core (pinned to the first allowed core, as `taskset -c` would), bounded sample.
    structured=False: the builder's restatement of the reference's DENSE operation sequence with a hand-written AVX2 GEMM
    standing in for Eigen's (oracle/eqf_oracle.cpp) -- the "cpu_baseline".  structured=True: the same equations without
    the structural-zero work and with Cholesky-form S^-1 / Sigma_e^-1 ("cpu_structured"), so that the algorithmic and the
    hardware parts of the GPU speed-up can be told apart."""
    from eqf_vio_amd import synth
    from oracle import binding as ob

    allowed = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, {allowed[0]})
    try:
        st = synth.make_stream(N, seed=1234, duration=3.0)
        fo = ob.OracleFilter(synth.template_settings_dict(), structured=structured)
        n = n_imu = n_vis = 0
        t_first = None  # the first events run on an 11 x 11 Sigma (no landmarks yet): not representative, not counted
        for kind, k in st.events():
            if kind == "imu":
                r = st.imu[k]
                fo.processIMUData(r[0], r[1:4], r[4:7])
            else:
                fo.processVisionData(st.vision_stamps[k], st.ids, st.bearings[k])
            if t_first is None:
                if kind == "vision":
                    t_first = time.perf_counter()
                continue
            n += 1
            n_imu += kind == "imu"
            n_vis += kind == "vision"
            if kind == "vision" and n_vis >= 2 and time.perf_counter() - t_first > budget_s:
                break
        dt = time.perf_counter() - t_first
    finally:
        os.sched_setaffinity(0, set(allowed))
    what = ("structured fp64 (sparse F and C, Cholesky-form update)" if structured
            else "dense fp64, reference op order, builder's port with a hand-written AVX2 GEMM")
    return {
        "value": n / dt,
        "unit": "steps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n} events ({n_imu} IMU + {n_vis} vision = whole frames, after the landmarks exist) of the same N={N} stream, "
        f"oracle/eqf_oracle.cpp ({what}), {dt:.1f} s pinned to core {allowed[0]} of {os.cpu_count()}",
    }


def parity_prefix(N, precision, frames=10):
    """Short oracle-checked prefix of the bench workload (one filter, per-call API): worst Sigma rel-Frobenius error, the
    frame where it occurs, worst position / attitude difference -- against the structured fp64 oracle."""
    from eqf_vio_amd import binding, synth
    from oracle import binding as ob

    st = synth.make_stream(N, seed=1234, duration=frames / 20.0 + 0.01)
    d = synth.template_settings_dict()
    fo = ob.OracleFilter(d, structured=True)
    fg = binding.FilterBatch(d, capacity=N, batch=1, precision=binding.PRECISION_F64 if precision == "f64" else binding.PRECISION_F32)
    worst = {"max_relS": 0.0, "frame": -1, "pos": 0.0, "att": 0.0}
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]
            fo.processIMUData(r[0], r[1:4], r[4:7])
            fg.process_imu([r[0]], r[1:4], r[4:7])
        else:
            fo.processVisionData(st.vision_stamps[k], st.ids, st.bearings[k])
            fg.process_vision([st.vision_stamps[k]], st.ids, st.bearings[k])
            So = fo.stateCovariance()
            rel = float(np.linalg.norm(fg.sigma() - So) / np.linalg.norm(So))
            eo, eg = fo.stateEstimate(), fg.state_estimate()
            if rel > worst["max_relS"]:
                worst["max_relS"], worst["frame"] = rel, int(k)
            worst["pos"] = max(worst["pos"], float(np.abs(eo["x"] - eg["x"]).max()))
            # attitude difference as the rotation angle between the two unit quaternions (rad): 2 |q1 - q2| for small angles
            sgn = 1.0 if float(np.dot(eo["q"], eg["q"])) >= 0 else -1.0
            worst["att"] = max(worst["att"], 2.0 * float(np.linalg.norm(eo["q"] - sgn * eg["q"])))
    worst["frames"] = frames
    worst["against"] = "oracle/eqf_oracle.cpp structured fp64 (pinned to the dense form by tests/test_oracle_structured.py)"
    worst["device_error_flag"] = fg.device_error()
    return worst


def roofline(fb, events, N, B, precision):
    """Per-kernel-class HIP-event timing over the timed region (second, profiled pass) -> dominant kernel."""
    fb.profile_enable(True)
    t0 = time.perf_counter()
    run_events(fb, events)
    fb.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    prof = fb.profile()
    fb.profile_enable(False)
    # integrateUpToTime steps per burst (IMU calls + the vision call's own step leave together, csrc/eqf_burst.hpp)
    n_bursts = max(prof.get("k_imu_burst", (0, 0.0))[0], 1)
    steps_per_burst = len(events) / n_bursts
    n = 11 + 3 * N
    m = 2 * N
    ne = 5 + 3 * N
    esz = 8 if precision == "f64" else 4
    # Algorithmic work PER CALL of the path, divided below by the launches the class needed for it.
    # vision updates in the timed region (the prep work is a launch of its own, or -- one small filter -- roles of k_chol_resident's launch)
    n_upd = max(prof.get("k_update_prep", (0, 0.0))[0], prof.get("k_chol_resident", (0, 0.0))[0], 1)
    # update flops = m^3/3 + 2 m^2 n + 2 n^2 m + n_e^3/3 (Cholesky of S, the solves for K on n right-hand
    # sides, the downdate Sigma -= K (C Sigma), Cholesky of Sigma_e); 5.89e8 at N = 200
    update_flops = m**3 / 3.0 + 2.0 * m * m * n + 2.0 * n * n * m + ne**3 / 3.0
    downdate_flops = 2.0 * n * n * m
    chain_flops = update_flops - downdate_flops
    # what the kernels execute: one forward solve instead of two (Sigma - Y^T Y form), half the downdate (symmetry)
    executed_flops = m**3 / 3.0 + m * m * (n + 7.0) + ne**3 / 3.0 + ne * ne * 11.0 + n * n * m
    # 64-wide path: the covariance downdate (and the reductions / innovation lift) ride along in ONE of the chain launches of an
    # update, bracketed as its own class k_chol_step_dd; both classes are launches of the kernel k_chol_step64
    c_plain, ms_plain = prof.get("k_chol_step", (0, 0.0))
    c_dd, ms_dd = prof.get("k_chol_step_dd", (0, 0.0))
    embedded = c_dd > 0
    chol_launches_per_update = max(c_plain + c_dd, 1) / n_upd
    # bytes a burst really moves: Sigma in + out once, plus the per-step per-landmark records (63 values, written by the builder
    # and read by the block kernel) -- DESIGN.md 4.1b
    burst_moved = (2.0 * n * n + steps_per_burst * N * 63 * 2.0) * esz * B
    shape = fb.launch_shape()  # of the most recent burst (every burst of a fixed-set run has the same shape)
    if shape["cs_out"]:
        # the burst's block kernel also leaves the landmark columns of C Sigma (m x 3N) and S (m x m) for the update (cs_in_burst)
        burst_moved += (m * 3.0 * N + m * m) * 8.0 * B
    # what the block kernel executes per Riccati step: 162 FMAs per 3 x 3 block, blocks on and below the diagonal only
    burst_flops = steps_per_burst * (N * (N + 1) / 2.0) * 324.0 * B
    # the update's prep work: with the burst's C Sigma / S it touches 12 leading columns per landmark row and the O(N) vectors -- a latency
    # chain, no byte count to hold against HBM; without, it reads the landmark rows of Sigma (n^2) and writes Y, S and the copy of Sigma_e
    prep_bytes_cs = 8.0 * (3.0 * N * 12 + m * 19.0 + 15.0 * N + 18.0 * N) * B
    prep_bytes_full = (n * n * esz + 8.0 * (m * (n + 7) + m * m + ne * ne)) * B
    algo = {
        # propagate = read Sigma once + write Sigma once
        "k_propagate": ("hbm", 2.0 * n * n * esz * B),
        "k_imu_burst": ("hbm", burst_moved),
        # every chain launch does ~1/L of the two factorisations; the one that carries the downdate does that on top
        "k_chol_step": ("mfma", chain_flops * B / max(chol_launches_per_update, 1)),
        "k_chol_step_dd": ("mfma", (downdate_flops + chain_flops / max(chol_launches_per_update, 1)) * B),
        # the whole factorisation part of an update in ONE launch (csrc/eqf_resident.hpp)
        "k_chol_resident": ("mfma", update_flops * B),
        "k_downdate": ("mfma", downdate_flops * B),
        # dense backend (cfg 3): build F + two n^3 GEMMs = 4 n^3 flops per Riccati step (SURVEY.md 8d "mfma_dense_equiv")
        "k_dense_riccati": ("mfma", 4.0 * n**3 * B),
        "k_update_prep": ("latency", prep_bytes_cs) if shape["cs_out"] else ("hbm", prep_bytes_full),
        "k_update_reduce": ("hbm", 8.0 * (m * (n + 7) + ne * 32) * B),
    }
    if prof.get("k_dense_riccati", (0, 0.0))[0]:
        # dense backend: k_propagate only steps the group / scalar state there, Sigma is moved by the GEMMs
        del algo["k_propagate"]
    rows = []
    for name, (cnt, ms) in prof.items():
        if cnt == 0:
            continue
        avg_us = ms * 1e3 / cnt
        row = {"kernel": name, "launches": cnt, "total_ms": round(ms, 3), "avg_us": round(avg_us, 3)}
        if name in algo:
            bound, work = algo[name]
            if bound == "latency":
                row.update(bound="latency", bytes_touched=work, note="C Sigma and S come from the burst's block kernel (cs_in_burst): this launch "
                           "reads 12 columns per landmark row -- a dependent chain of O(N) work, no HBM or matrix-core figure applies")
            elif bound == "hbm":
                ach = work / (avg_us * 1e-6) / 1e9
                row.update(bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5))
                if name == "k_imu_burst":
                    row["steps_per_burst"] = round(steps_per_burst, 2)
                    # the other pipe the burst can be bound by: the structured step's fp64 FMAs on the vector ALUs
                    vt = burst_flops / (avg_us * 1e-6) / 1e12
                    row.update(valu_tflops=round(vt, 3), valu_peak=VALU_F64_PEAK_TF, valu_frac=round(vt / VALU_F64_PEAK_TF, 5),
                               launch_shape=shape,
                               binds=("fp64 VALU" if vt / VALU_F64_PEAK_TF > ach / HBM_PEAK_GBS else "HBM") + " is the nearer roof (neither is reached: "
                               "the bracket covers the builder launch, a serial O(N) recurrence, and the block kernel)")
                    # what a step-by-step implementation would have to move (2 n^2 bytes per STEP): NOT a roofline
                    # fraction -- Sigma stays in registers across the steps of a burst
                    row["effective_step_by_step_gbs"] = round(2.0 * n * n * esz * B * steps_per_burst / (avg_us * 1e-6) / 1e9, 2)
            else:
                ach = work / (avg_us * 1e-6) / 1e12
                pk = MFMA_PEAK_TF["f64"]  # factorisation is always fp64
                if name in ("k_downdate", "k_dense_riccati", "k_chol_step_dd"):
                    pk = MFMA_PEAK_TF[precision]
                row.update(bound="mfma", achieved=round(ach, 4), peak=pk, unit="TFLOP/s", frac=round(ach / pk, 5))
        rows.append(row)
    rows.sort(key=lambda r: -r["total_ms"])
    out = None
    chol_ms, chol_cnt = ms_plain + ms_dd, c_plain + c_dd
    top = next((r for r in rows if "bound" in r), None)
    if top is not None and top["kernel"] == "k_chol_resident":
        # dominant kernel = k_chol_resident: one launch per update carries SURVEY 8(d)'s update flops
        out = {k: top[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
        out.update(traffic=None, kernel="k_chol_resident", avg_us=top["avg_us"], launches_per_update=1.0,
                   algorithmic_flops_per_update=update_flops * B, executed_flops_per_update=executed_flops * B,
                   achieved_executed=round(executed_flops * B / (top["avg_us"] * 1e-6) / 1e12, 4), us_per_update=top["avg_us"])
        out["frac_executed"] = round(out["achieved_executed"] / out["peak"], 5)
    elif chol_cnt and top is not None and top["kernel"].startswith("k_chol_step"):
        # dominant kernel = k_chol_step64 (all its launches of an update, the downdate-carrying one included): SURVEY 8(d)'s
        # update flops per update / launches per update / average launch duration
        avg_us = chol_ms * 1e3 / chol_cnt
        ach = update_flops * B / chol_launches_per_update / (avg_us * 1e-6) / 1e12
        pk = MFMA_PEAK_TF["f64"]
        out = {"bound": "mfma", "achieved": round(ach, 4), "peak": pk, "unit": "TFLOP/s", "frac": round(ach / pk, 5), "traffic": None,
               "kernel": "k_chol_step64", "avg_us": round(avg_us, 3), "launches_per_update": round(chol_launches_per_update, 2),
               "algorithmic_flops_per_update": update_flops * B, "executed_flops_per_update": executed_flops * B,
               "achieved_executed": round(executed_flops * B / chol_launches_per_update / (avg_us * 1e-6) / 1e12, 4),
               "us_per_update": round(chol_ms * 1e3 / n_upd, 2)}
        out["frac_executed"] = round(out["achieved_executed"] / pk, 5)
    elif top is not None:
        out = {k: top[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
        out.update(traffic=None, kernel=top["kernel"], avg_us=top["avg_us"])
    # cross-check: the class totals must account for the profiled pass (the rest is dispatch gaps + host launch time)
    cover = sum(r["total_ms"] for r in rows) / max(wall_ms, 1e-9)
    return out, rows, {"profiled_pass_wall_ms": round(wall_ms, 3), "kernel_time_over_wall": round(cover, 3),
                       "note": "sum of the HIP-event class totals / wall time of the profiled pass; < 1 = dispatch gaps and host launch "
                               "overhead, > 1.02 = double counting"}


def pmc_traffic(args, kernel):
    """HBM bytes per launch of `kernel` from rocprofv3 PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
    WRITE_SIZE in SEPARATE passes (kernel-trace only), values in KiB.  On gfx950 FETCH_SIZE under-reports coalesced
    reads by 2x; the factor is calibrated in the same pass on k_sigma_export, which reads exactly n^2 doubles."""
    import csv
    import shutil
    import statistics
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    out = {}
    note = ""
    n = 11 + 3 * args.landmarks
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="eqf_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "b", "--", sys.executable,
               os.path.abspath(__file__), "--steps", "220", "--warmup", "110", "--landmarks", str(args.landmarks), "--filters-per-gpu",
               str(args.filters_per_gpu), "--precision", args.precision, "--pmc-child"]
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)  # the counter pass is a plain one-process run, whatever launched this one
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
            path = None
            for root, _, files in os.walk(d):
                for fn in files:
                    if fn.endswith("counter_collection.csv"):
                        path = os.path.join(root, fn)
            vals, calib = [], []
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"]
                    if kernel in name:
                        vals.append(float(row["Counter_Value"]))
                    if "k_sigma_export" in name:
                        calib.append(float(row["Counter_Value"]))
            out[counter] = (statistics.mean(vals) * 1024.0 if vals else None, calib)
        except Exception as e:  # profiling is best effort: never fail the bench because of it
            return None, f"{counter} pass failed: {type(e).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, calib = out["FETCH_SIZE"]
    write, _ = out["WRITE_SIZE"]
    if fetch is None or write is None:
        return None, "kernel not found in the counter trace"
    esz = 8 if args.precision == "f64" else 4
    factor = 2.0  # MI355X_MICROARCH.md, section HBM
    if calib:
        factor = (n * n * esz) / (statistics.mean(calib) * 1024.0)
        note = f"FETCH_SIZE x{factor:.2f} (calibrated on k_sigma_export's known {n}x{n} read)"
    return fetch * factor + write, note


def timed_job(args, dist, rank, world, device, N, B, steps, warmup, dense=False):
    """One job: `B` filters of N landmarks on this rank, streams scattered from rank 0 (RCCL), `warmup` untimed events, then
    exactly `steps` timed events bracketed by barrier + synchronize; the time is the maximum over the ranks."""
    import torch

    from eqf_vio_amd import binding, shard, synth

    nev = steps + warmup
    imu, vst, bear, events = shard.scatter_streams(dist, rank, world, N, B, nev, device="cuda" if dist is not None else None)
    ids = np.arange(N, dtype=np.int32)
    prec = binding.PRECISION_F64 if args.precision == "f64" else binding.PRECISION_F32
    fb = binding.FilterBatch(synth.template_settings_dict(), capacity=N, batch=B, device=device, precision=prec)
    if dense:
        fb.set_dense_propagate(True)
    for kv in args.debug_option:
        fb.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
    fb.stream_upload(imu, vst, ids, bear)
    warm, timed = events[:warmup], events[warmup:]
    run_events(fb, warm)
    fb.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_events(fb, timed)
    fb.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # results gathered to rank 0 (pose + |Sigma|_F of every filter)
    res = np.zeros((B, 8))
    for b in range(B):
        e = fb.state_estimate(b)
        res[b, :3] = e["x"]
        res[b, 3:7] = e["q"]
        res[b, 7] = np.linalg.norm(fb.sigma(b))
    res = shard.gather_results(dist, rank, world, res, device="cuda" if dist is not None else None)
    return fb, timed, dt, res


def churn_leg(device, seconds=3.0, pool=260, fixed=200):
    """SURVEY.md 8(d) "a separate churn run exercises a18": the landmark bookkeeping of VIOFilter.cpp:211-230, 345-443 at N ~ 200 with the
    outlier gate at the REFERENCE DEFAULT (include/eqf_vio/VIOFilterSettings.h: outlierThreshold = 0.01), through the per-call API (host
    buffers: the caller owns the ids).  A pool of `pool` landmarks, each visible on a window of frames, ~`fixed` in view, a few entering
    and leaving every frame.  Four modes on the same stream, best of three runs each: a fixed set / churn, gate off / on."""
    from eqf_vio_amd import binding, synth

    st = synth.make_stream(pool, seed=1234, duration=seconds)
    F = st.bearings.shape[0]
    rng = np.random.default_rng(3)
    start = rng.integers(-F, F, size=pool)
    length = rng.integers(F // 2, F, size=pool)
    start[:120], length[:120] = -1, 10 * F
    ev = list(st.events())
    d0 = synth.template_settings_dict()

    def run(churn, gate):
        d = dict(d0)
        if gate:
            d["outlierThreshold"] = 0.01
        fb = binding.FilterBatch(d, capacity=pool, batch=1, device=device)
        prev, nch, nvis = None, 0, 0
        fb.synchronize()
        t0 = time.perf_counter()
        for kind, k in ev:
            if kind == "imu":
                r = st.imu[k]
                fb.process_imu([r[0]], r[1:4], r[4:7])
            else:
                vis = np.where((start <= k) & (k < start + length))[0] if churn else np.arange(fixed)
                if prev is not None:
                    nch += len(set(vis) ^ set(prev))
                prev = vis
                fb.process_vision([st.vision_stamps[k]], st.ids[vis].astype(np.int32), st.bearings[k, vis].copy())
                nvis += 1
        fb.synchronize()
        dt = time.perf_counter() - t0
        out = (dt, fb.num_landmarks(), nch / max(nvis - 1, 1), fb.device_error())
        del fb
        return out

    run(False, False)  # (first launches of the churn kernels)
    res = {}
    for name, churn, gate in (("fixed_set", False, False), ("fixed_set_gate", False, True), ("churn", True, False), ("churn_gate", True, True)):
        dt, nEnd, chg, err = min(run(churn, gate) for _ in range(3))
        res[name] = {"value": round(len(ev) / dt, 1), "landmarks_at_end": nEnd, "landmark_changes_per_frame": round(chg, 2), "device_error_flag": err}
    return {
        "metric": "EqF propagate+update steps/sec, per-call API, pool of %d landmarks, ~%d in view, outlier gate at the reference default 0.01" % (pool, fixed),
        "unit": "steps/s", "steps": len(ev), "modes": res,
        "churn_gate_over_fixed_set": round(res["churn_gate"]["value"] / res["fixed_set"]["value"], 3),
        "note": "per-call API with host buffers (PCIe and the host-side id matching included), best of 3; `value` of the main line replays a "
                "resident stream with a fixed set and the gate off",
    }


GRIDS = {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (2, 4)}
KERNEL_KEYS = ("kernel", "launches", "avg_us", "bound", "achieved", "unit", "frac", "valu_tflops", "valu_frac", "binds", "bytes_touched", "launch_shape")


def tiled_leg(args, dist, rank, world, device):
    """BASELINE configs[4]: ONE filter of N landmarks, Sigma 2-D block-partitioned over the ranks of the job (eqf_tf_*: csrc/eqf_tiledf.hip), closed
    loop, the same stream on every rank.  One warm-up frame (the first frame: landmarks appended + update), then `frames` timed frames of
    10 IMU calls + 1 vision call, bracketed by barrier + synchronize, maximum over the ranks.  On one GPU the monolithic single-GPU path
    runs the same events right after, for the number next to it."""
    import torch

    from eqf_vio_amd import binding, synth, tiled

    N, bl, frames = args.tiled_landmarks, args.tiled_block or (320 if world == 1 else 250), args.tiled_frames
    Pr, Pc = GRIDS[world]
    st = synth.make_stream(N, seed=1234, duration=(frames + 2) / 20.0 + 0.011)
    ev = list(st.events())
    first_vis = next(i for i, (kind, _) in enumerate(ev) if kind == "vision")
    warm, timed = ev[: first_vis + 1], ev[first_vis + 1: first_vis + 1 + 11 * frames]
    d = synth.template_settings_dict()
    be = tiled.HipBackend(d, capacity=N, device_index=device)
    tf = tiled.TiledFilter(tiled.ProcessGrid(dist if world > 1 else None, Pr, Pc, device=be.device), be, bl)

    def run(f_imu, f_vis, events):
        for kind, k in events:
            if kind == "imu":
                r = st.imu[k]
                f_imu(r[0], r[1:4], r[4:7])
            else:
                f_vis(st.vision_stamps[k], st.ids, st.bearings[k])

    def timed_run(f_imu, f_vis, sync):
        run(f_imu, f_vis, warm)
        sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(f_imu, f_vis, timed)
        sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    tf.check_every = 0  # (the pivot flag is looked at once, after the timed region)
    run(tf.processIMUData, tf.processVisionData, [])
    tf.phase_ms = None
    dt = timed_run(tf.processIMUData, tf.processVisionData, lambda: torch.cuda.synchronize())
    tf.check()
    sll_ref = tf.Sll.clone() if (world == 1 and not args.no_i8_downdate) else None  # (what the integer-pipe run below is held against)
    # per-phase GPU time: a second pass over the next frames with event brackets (kept out of the timed region)
    tf.phase_ms = {}
    more = ev[first_vis + 1 + 11 * frames: first_vis + 1 + 11 * (frames + 1)]
    run(tf.processIMUData, tf.processVisionData, more)
    tf.collect_phases()
    n_vis_ph = max(sum(1 for kind, _ in more if kind == "vision"), 1)
    n = 11 + 3 * N
    m, ne = 2 * N, 5 + 3 * N
    n_vis = sum(1 for kind, _ in timed if kind == "vision")
    update_flops = m**3 / 3.0 + 2.0 * m * m * n + 2.0 * n * n * m + ne**3 / 3.0
    out = {
        "metric": "EqF propagate+update steps/sec, ONE filter of N=%d landmarks, Sigma 2-D block-partitioned over %d GPU(s)" % (N, world),
        "value": len(timed) / dt, "unit": "steps/s", "n_gpus": world, "grid": "%d x %d" % (Pr, Pc), "block_landmarks": bl, "steps": len(timed),
        "ms_per_step": dt * 1e3 / len(timed), "ms_per_frame": dt * 1e3 / max(n_vis, 1), "scaling": "strong", "dtype": "f64",
        "device_error_flag": be.device_error(),
        "sigma_fro_local": float(torch.linalg.norm(tf.Sll).item()),
        "phases_ms_per_frame": {k: round(v / n_vis_ph, 3) for k, v in tf.phase_ms.items()},
        # chain_E runs on its own stream pair NEXT TO chain_S + downdate: the phases are stream-busy times and do not add up to the frame
        "phases_concurrent": [["chain_S", "downdate"], ["chain_E"]],
        "phases_note": "GPU time per phase on the phase's own stream; chain_E overlaps chain_S + downdate (two stream pairs on disjoint CU "
                       "sets), so ms_per_frame ~ propagate + prep + max(chain_S + downdate, chain_E) + finish, not the sum",
        # the rate of an update: SURVEY 8(d)'s flops over the WALL time of an update (frame minus its Riccati steps), not over the sum of phases
        "update_algorithmic_tflops": round(update_flops / max(dt * 1e3 / max(n_vis, 1) - tf.phase_ms.get("propagate", 0.0) / n_vis_ph, 1e-9) / 1e9, 2)
        if tf.phase_ms else None,
        "note": "closed loop through the C ABI (eqf_tiled_* / eqf_tile_*), torch.distributed only moves solved block rows; "
                + ("one rank: no exchange" if world == 1 else "RCCL broadcasts along process rows / columns"),
    }
    if sll_ref is not None:
        out["_sll_ref"] = sll_ref
    if world > 1:
        out["note"] += "; UNMEASURED on more than one GPU until a node is available to the builder -- this line is then the first measurement"
    tf.close()  # (explicitly: a handle left to the garbage collector keeps its CU-masked streams -- hardware queues -- while the next legs run)
    be.close()
    del tf, be
    # Landmark churn in the partitioned filter (VIOFilter.cpp:345-443 on slots, eqf_tiled_edit_landmarks) with the outlier gate at the
    # reference default: the same frames, but every frame `turn` landmarks (1 %) are out of view and the ones hidden a frame earlier come
    # back as new landmarks.  A fresh filter (the gate is a setting of the handle); one warm-up frame, then the timed frames.
    try:
        dc = dict(d)
        dc["outlierThreshold"] = 0.01
        be = tiled.HipBackend(dc, capacity=N, device_index=device)
        tf = tiled.TiledFilter(tiled.ProcessGrid(dist if world > 1 else None, Pr, Pc, device=be.device), be, bl)
        tf.check_every = 0
        turn = max(N // 100, 1)
        frame_no = [0]

        def vis_churn(stamp, ids, y):
            f = frame_no[0]
            frame_no[0] += 1
            hidden = (np.arange(turn) + f * turn) % N
            vis = np.ones(N, dtype=bool)
            vis[hidden] = False
            return tf.processVisionData(stamp, ids[vis], y[vis])

        dtc = timed_run(tf.processIMUData, vis_churn, lambda: torch.cuda.synchronize())
        tf.check()
        cs = tf.churn_stats
        out["churn"] = {"value": len(timed) / dtc, "unit": "steps/s", "ms_per_frame": dtc * 1e3 / max(n_vis, 1), "vs_fixed_set": dt / dtc,
                        "outlierThreshold": 0.01, "landmarks_in_view": N - turn, "slots_in_use": int(tf.nslots),
                        "removed_old": cs["removed_old"], "removed_outliers": cs["removed_outliers"], "added": cs["added"] - N + turn,
                        "device_error_flag": be.device_error(),
                        "note": "per frame %d landmarks leave and %d enter (slots: a removed landmark leaves a decoupled hole, the next new one "
                                "refills it; nothing moves between ranks), gate evaluated on the replicated state every frame" % (turn, turn)}
        tf.close()
        be.close()
        del tf, be
    except Exception as e:  # the fixed-set figure above stands on its own
        out["churn"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world == 1 and rank == 0:
        fb = binding.FilterBatch(d, capacity=N, batch=1, device=device)
        # (before the integer-pipe sub-legs: run after them, this leg came out at 134 - 142 steps/s in two runs of three of scripts/tiled_bench.py and at
        # 169 - 170 in the third, with the stream pool or without, with the handles closed explicitly or not; cause not found -- NOTES R6.7)
        dtm = timed_run(lambda s_, w_, a_: fb.process_imu([s_], w_, a_), lambda s_, i_, y_: fb.process_vision([s_], i_, y_), fb.synchronize)
        out["monolithic_single_gpu"] = {"value": len(timed) / dtm, "unit": "steps/s", "ms_per_frame": dtm * 1e3 / max(n_vis, 1),
                                        "note": "the single-GPU product path (eqf_process_*) on the same events"}
        del fb
    if world == 1 and rank == 0 and not args.no_i8_downdate:
        # Round 6: the same frames with the covariance downdate on the INTEGER matrix pipe (eqf_tf_set_option "downdate_slices" = 6: Y's columns
        # as six 7-bit slices, int8 MFMA with exact accumulation, fp64 recombination; csrc/eqf_tile.hpp) -- north_star's "low-precision MFMA for the
        # dense Sigma contractions, Sigma within 1e-4", for the one product it was built for.  Opt-in; the line's other figures are fp64.
        # "i8_chains": the trailing products of the two factorisations from FIVE slices as well ("chain_slices" = 5: they forgive more than the
        # downdate, scripts/slice_precision_study_chain.py) -- every large product of the update on the integer pipe, the solves and factors fp64.
        sll_ref = out.pop("_sll_ref")
        # "i8_chains_fp64_downdate": the factorisations' products alone -- Sigma stays within ~1e-7 of the fp64 run (three orders inside the tolerance).
        for key, dd, chain in (("i8_downdate", 6, 0), ("i8_chains", 6, 5), ("i8_chains_fp64_downdate", 0, 5)):
            try:
                be = tiled.HipBackend(d, capacity=N, device_index=device)
                tf = tiled.TiledFilter(tiled.ProcessGrid(None, Pr, Pc, device=be.device), be, bl)
                tf.check_every = 0
                if dd:
                    tf.downdate_slices = dd
                if chain:
                    tf.chain_slices = chain
                tf.phase_ms = None
                dti = timed_run(tf.processIMUData, tf.processVisionData, lambda: torch.cuda.synchronize())
                tf.check()
                diff = float((torch.linalg.norm(tf.Sll - sll_ref) / torch.linalg.norm(sll_ref)).item())
                tf.phase_ms = {}
                run(tf.processIMUData, tf.processVisionData, more)
                tf.collect_phases()
                out[key] = {"value": len(timed) / dti, "unit": "steps/s", "ms_per_frame": dti * 1e3 / max(n_vis, 1), "downdate_slices": dd,
                            "chain_slices": chain, "integer_products_per_product": {"downdate": dd * (dd + 1) // 2, "chains": chain * (chain + 1) // 2},
                            "vs_fp64": dt / dti, "device_error_flag": be.device_error(),
                            "sigma_rel_frobenius_difference_to_the_fp64_run": diff, "tolerance": 1e-4,
                            "phases_ms_per_frame": {k: round(v / n_vis_ph, 3) for k, v in tf.phase_ms.items()},
                            "note": "local blocks of Sigma after the same %d timed frames, against the fp64 run of this leg; opt-in "
                                    "(eqf_tf_set_option \"downdate_slices\" / \"chain_slices\"), every other figure of the line is the fp64 path" % frames}
                tf.close()
                be.close()
                del tf, be
            except Exception as e:
                out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        del sll_ref
    out.pop("_sll_ref", None)
    return out


def self_spawn(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks of the job ourselves -- this script once per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in each child's environment, exactly what torch.distributed.run would
    set -- and wait for them.  Rank 0 owns stdout (the ONE JSON line); the other ranks' stdout goes to stderr.  The first rank that fails
    takes the job down: the others are stopped (own PIDs only) and the exit code is that rank's."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n = args.gpus
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), EQF_BENCH_SELF_SPAWNED="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = set(range(n))
    rank0_done_at = None
    while alive:
        if rank0_done_at is not None and time.time() - rank0_done_at > 60.0:
            # rank 0 has printed the line and left cleanly a minute ago: a straggler (a rank stuck in a collective nobody else will join any
            # more) must not hold the job -- stop what is left (own PIDs only) and report rank 0's success
            for q in sorted(alive):
                if procs[q].poll() is None:
                    print("bench.py: rank %d still running 60 s after rank 0 finished; stopping it" % q, file=sys.stderr, flush=True)
                    procs[q].terminate()
            break
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if r == 0 and code == 0:
                rank0_done_at = time.time()
            if code != 0 and rc == 0:
                rc = code
                print("bench.py: rank %d of %d exited with code %d; stopping the job" % (r, n, code), file=sys.stderr, flush=True)
                deadline = time.time() + 10.0  # (the others get to print their own error first: a missing GPU is missing for all of them)
                while time.time() < deadline and any(procs[q].poll() is None for q in alive):
                    time.sleep(0.05)
                for q in alive:
                    if procs[q].poll() is None:
                        procs[q].terminate()
        time.sleep(0.05)
    sys.exit(rc)


def require_devices(world, rank, need_gpu=True):
    """Fail loudly, on EVERY rank, before any rendezvous: an N-rank job needs N visible GPUs (one process per GPU)."""
    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if need_gpu and have < world:
        print("bench.py: rank %d of %d: not enough devices -- --gpus %d needs %d visible GPUs (one rank per GPU), this process sees %d"
              % (rank, world, world, world, have), file=sys.stderr, flush=True)
        sys.exit(3)


def spawn_selftest(args, world, rank):
    """No GPU, no filter: the launcher's ranks meet over gloo, rank 0's streams reach every rank, a result comes back (shard.py's two
    collectives, the same calls the real job makes over RCCL).  Test infrastructure for `--gpus N` on a GPU-less box."""
    import torch.distributed as dist

    from eqf_vio_amd import shard

    dist.init_process_group("gloo")
    imu, vst, bear, events = shard.scatter_streams(dist, rank, world, 6, 2, 33, device=None)
    res = np.full((2, 3), float(rank))
    res[:, 1] = imu[0, :, 0]  # (the first stamp of each of my two filters: proves the slices arrived)
    res[:, 2] = len(events)
    allres = shard.gather_results(dist, rank, world, res, device=None)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest": "launcher only: gloo rendezvous + scatter + gather, no filter ran", "n_gpus": world,
                          "self_spawned": os.environ.get("EQF_BENCH_SELF_SPAWNED") == "1",
                          "ranks_seen": sorted(set(int(v) for v in allres[:, 0])), "filters": int(allres.shape[0]),
                          "events": int(allres[0, 2])}))


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_spawn(args)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print("bench.py: --gpus %d but the launcher's WORLD_SIZE is %d: refusing to print a line with the wrong n_gpus" % (args.gpus, world),
              file=sys.stderr, flush=True)
        sys.exit(2)
    if args.spawn_selftest:
        spawn_selftest(args, world, rank)
        return
    require_devices(world, rank)
    import torch

    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: RCCL path even for a single rank
        import torch.distributed as dist_

        dist = dist_
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = local_rank

    N, B = args.landmarks, args.filters_per_gpu
    if not args.no_prewarm and not args.pmc_child:
        # Untimed, on a THROW-AWAY handle with its own stream of the same shape: every kernel variant of the path has run once,
        # the code object is resident and the clocks are up before the measured handle exists.  (The driver times 20 steps
        # after 5 warm-up steps -- half a millisecond -- where a first-use hiccup of one kernel variant is 20 % of the figure.)
        pw, _, _, _ = timed_job(args, dist, rank, world, device, N, B, 44, 22, dense=args.dense_propagate)
        del pw
    fb, timed, dt, res = timed_job(args, dist, rank, world, device, N, B, args.steps, args.warmup, dense=args.dense_propagate)
    err = fb.device_error()
    if args.pmc_child:
        fb.sigma(0)  # one k_sigma_export launch: known byte count, calibrates FETCH_SIZE
        return

    n_timed = len(timed)
    total_steps = n_timed * B * world
    line = {
        # BASELINE.json's metric string, verbatim for the configuration it is quoted on (N = 200); other N only change the number
        "metric": "EqF propagate+update steps/sec at N=%d landmarks (fp32); 1-GPU + 8-GPU batch" % N,
        "baseline_metric": "EqF propagate+update steps/sec at N=200 landmarks (fp32); 1-GPU + 8-GPU batch",
        "metric_as_run": "EqF propagate+update steps/sec at N=%d landmarks, %s arithmetic, %d GPU(s) x %d filter(s)" % (
            N, "fp64" if args.precision == "f64" else "fp32", world, B),
        "metric_note": "arithmetic is %s, not the fp32 the metric string names: fp64 is the reference's own type and the only one that "
                       "meets the 1e-4 Sigma tolerance on the reference's settings (measured on the device: DESIGN.md section 2, "
                       "profiles/r02_fp32_study.txt); `dtype` says what was computed" % ("fp64" if args.precision == "f64" else "fp32 (EQF_PRECISION_F32, not parity grade)"),
        "value": total_steps / dt,
        "unit": "steps/s",
        "n_gpus": world,
        "launcher": ("bench.py --gpus %d started the ranks itself" % world) if os.environ.get("EQF_BENCH_SELF_SPAWNED") == "1"
        else ("torch.distributed.run" if "RANK" in os.environ else "single process"),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / n_timed,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # the arithmetic type of the path.  BASELINE.json's metric string says "(fp32)": NOT what is reported here -- fp64 is the
        # reference's own arithmetic and the only one that meets the 1e-4 Sigma tolerance (measured)
        "dtype": args.precision,
        "data": "synthetic",
        "config": {
            "workload": "Synthetic IMU@200Hz + bearings@20Hz, N=%d landmarks, %d filter(s) per GPU, fastRiccati=false" % (N, B),
            "filters_per_gpu": B,
            "filters_total": B * world,
            "landmarks": N,
            "events": {"imu": sum(1 for k, _ in timed if k == "imu"), "vision": sum(1 for k, _ in timed if k == "vision")},
            "parallelism": "independent filters sharded over %d GPU(s), RCCL scatter/gather only" % world,
            "imu_burst": "IMU calls between two vision frames + the vision call's integrateUpToTime leave as one burst of <= 16 steps "
                         "(one launch for a single small fp64 filter -- builder and block workgroups together, k_burst_fused -- else two; Sigma read/written "
                         "once; every step is the reference's step); EQF_IMU_BURST=0: one launch per call",
        },
        "device_error_flag": err,
        "sigma_fro_filter0": float(res[0, 7]) if res is not None else None,
        "prewarm": None if args.no_prewarm else "66 untimed events on a separate throw-away handle before the measured one is created",
    }
    from eqf_vio_amd import binding as _binding

    line["build"] = _binding.build_info()  # which .so ran, and that it was built from the sources beside it
    if rank == 0 and not args.no_roofline:
        rl, rows, cover = roofline(fb, timed, N, B, args.precision)
        line["roofline"] = rl
        line["kernels"] = rows
        line["profile_coverage"] = cover
        # propagate-only and update-only time per call (kernel time from the HIP events), frames/s
        t = {r["kernel"]: (r["total_ms"], r["launches"]) for r in rows}
        n_imu_vis = len(timed)
        n_upd = max(t.get("k_update_prep", (0, 0))[1], t.get("k_chol_resident", (0, 0))[1], 1)
        prop_ms = t.get("k_propagate", (0, 0))[0] + t.get("k_dense_riccati", (0, 0))[0] + t.get("k_imu_burst", (0, 0))[0]
        upd_ms = sum(t.get(k, (0, 0))[0] for k in ("k_update_prep", "k_chol_step", "k_chol_step_dd", "k_chol_resident", "k_update_reduce",
                                                   "k_update_finish", "k_downdate"))
        line["per_call"] = {
            "propagate_us": round(prop_ms * 1e3 / max(n_imu_vis, 1), 3),
            "update_us": round(upd_ms * 1e3 / n_upd, 3),
            "frames_per_s": round(line["value"] / B / world * n_upd / max(n_imu_vis, 1), 1),
            "note": "kernel time of the profiled pass (dispatch gaps excluded); a frame = 10 IMU calls + 1 vision call",
        }
        if isinstance(rl, dict) and "k_update_prep" not in t and "k_chol_resident" in t:
            rl["note"] = ("this launch of k_chol_resident CONTAINS the prep work (residuals, C Sigma, S, first diagonal blocks: prep roles, round 4) that "
                          "rounds 1-3 ran as a launch of its own in front of it (k_update_prep64, 12 us at N = 200): compare per_call.update_us "
                          "across rounds, not this kernel's duration alone; EQF_RES_FOLD_PREP=0 gives the two-launch shape")
    del fb  # free the GPU for the next legs

    # ---- steady-state leg: the driver's default run times 20 steps (two frames, half a millisecond); the same workload over 10 s of
    # stream (2200 steps after 220 of warm-up) is what DESIGN / README quote, so it is measured in the same run
    if not args.no_steady_state and args.steps < 2200 and not args.dense_propagate and N <= 400:
        fb3, timed3, dt3, _ = timed_job(args, dist, rank, world, device, N, B, 2200, 220)
        err3 = fb3.device_error()
        del fb3
        line["steady_state"] = {
            "metric": line["metric"], "value": len(timed3) * B * world / dt3, "unit": "steps/s", "steps": len(timed3), "warmup": 220,
            "ms_per_step": dt3 * 1e3 / len(timed3), "n_gpus": world, "filters_per_gpu": B, "device_error_flag": err3,
            "note": "same workload and code path as `value`, 10 s of stream instead of %d steps" % args.steps,
        }

    # ---- strong-scaling leg: BASELINE configs[3], 64 filters IN TOTAL over the GPUs of the job (64 / world per GPU)
    if not args.no_batch64 and 64 % world == 0 and not args.dense_propagate and N == 200:
        Bs = 64 // world
        fb2, timed2, dt2, res2 = timed_job(args, dist, rank, world, device, 200, Bs, args.batch64_steps, 110)
        err2 = fb2.device_error()
        rl2 = rows2 = None
        if rank == 0 and world == 1 and not args.no_roofline:
            rl2, rows2, _ = roofline(fb2, timed2, 200, Bs, args.precision)
        del fb2
        line["batch64_strong"] = {
            "metric": "EqF propagate+update steps/sec, 64 filters of N=200 in total",
            "value": len(timed2) * 64 / dt2,
            "unit": "steps/s",
            "scaling": "strong",
            "n_gpus": world,
            "filters_total": 64,
            "filters_per_gpu": Bs,
            "steps": len(timed2),
            "warmup": 110,
            "ms_per_step": dt2 * 1e3 / len(timed2),
            "device_error_flag": err2,
            "sigma_fro_min_max": [float(res2[:, 7].min()), float(res2[:, 7].max())] if res2 is not None else None,
        }
        if rows2 is not None:
            line["batch64_strong"]["roofline"] = rl2
            line["batch64_strong"]["kernels"] = [{k: r[k] for k in KERNEL_KEYS if k in r} for r in rows2]

    # ---- 8 filters per GPU: ONE GPU's share of BASELINE configs[3] on an 8-GPU node -- the figure the 0.9x strong-scaling target hangs on
    # (8 GPUs x this against 8 x the one-GPU batch of 64 above).  One GPU only: on several GPUs batch64_strong already runs 64 / world.
    if not args.no_batch8 and world == 1 and not args.dense_propagate and N == 200 and B == 1 and args.precision == "f64":
        fb8, timed8, dt8, _ = timed_job(args, dist, rank, world, device, 200, 8, 880, 110)
        err8 = fb8.device_error()
        leg8 = {"metric": "EqF propagate+update steps/sec, 8 filters of N=200 on one GPU (one GPU's share of the 64-filter batch on 8 GPUs)",
                "value": len(timed8) * 8 / dt8, "unit": "steps/s", "filters_per_gpu": 8, "steps": len(timed8), "warmup": 110,
                "ms_per_step": dt8 * 1e3 / len(timed8), "us_per_frame": dt8 * 1e6 / max(sum(1 for k, _ in timed8 if k == "vision"), 1),
                "device_error_flag": err8}
        if not args.no_roofline:
            rl8, rows8, cover8 = roofline(fb8, timed8, 200, 8, args.precision)
            leg8["roofline"] = rl8
            leg8["kernels"] = [{k: r[k] for k in KERNEL_KEYS if k in r} for r in rows8]
            leg8["profile_coverage"] = cover8["kernel_time_over_wall"]
        if "batch64_strong" in line:
            leg8["strong_scaling_projection_8gpu"] = round(8.0 * leg8["value"] / (8.0 * line["batch64_strong"]["value"]), 3)
            leg8["projection_note"] = ("8 GPUs x 8 filters over 8 x (one GPU x 64 filters): what `batch64_strong` at --gpus 8 would show relative to "
                                       "linear if the ranks do not disturb each other (no data-path collective); a projection from ONE GPU, not a measurement")
        del fb8
        line["batch8"] = leg8

    # ---- BASELINE configs[2]: one filter of N = 1000 (Sigma 3011 x 3011), the structured product path, with its own roofline
    if not args.no_n1000 and world == 1 and not args.dense_propagate and N == 200 and B == 1 and args.precision == "f64":
        fbk, timedk, dtk, _ = timed_job(args, dist, rank, world, device, 1000, 1, 220, 55)
        errk = fbk.device_error()
        legk = {"metric": "EqF propagate+update steps/sec at N=1000 landmarks (BASELINE configs[2]), structured product path, 1 GPU",
                "value": len(timedk) / dtk, "unit": "steps/s", "steps": len(timedk), "warmup": 55, "ms_per_step": dtk * 1e3 / len(timedk),
                "device_error_flag": errk, "dtype": args.precision}
        if not args.no_roofline:
            rlk, rowsk, coverk = roofline(fbk, timedk, 1000, 1, args.precision)
            legk["roofline"] = rlk
            legk["kernels"] = [{k: r[k] for k in KERNEL_KEYS if k in r} for r in rowsk]
            legk["profile_coverage"] = coverk["kernel_time_over_wall"]
            legk["roofline_note"] = ("`frac` prices SURVEY 8(d)'s 2 n^2 m for the downdate, the kernel executes n^2 m (symmetry): read `frac_executed` "
                                     "beside it; the dense MFMA Riccati backend of this config is `bench.py --landmarks 1000 --dense-propagate`")
        del fbk
        line["n1000"] = legk

    # ---- BASELINE configs[4]: one N = 4000 filter partitioned over the ranks of the job
    if not args.no_tiled and world in GRIDS and not args.dense_propagate and not args.pmc_child:
        # (a failure of this leg must not take the headline line with it: exceptions are reported in its place, and on several GPUs --
        # where a lost peer shows as a collective that never returns -- a watchdog prints the line without the leg and ends the rank)
        watchdog = None
        if world > 1:
            import threading

            def give_up():
                if rank == 0:
                    line["tiled_cfg5"] = {"error": "no result after %d s (watchdog)" % args.tiled_timeout}
                    print(json.dumps(line), flush=True)
                os._exit(0)

            watchdog = threading.Timer(args.tiled_timeout, give_up)
            watchdog.daemon = True
            watchdog.start()
        try:
            line["tiled_cfg5"] = tiled_leg(args, dist, rank, world, device)
        except Exception as e:
            line["tiled_cfg5"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if watchdog is not None:
            watchdog.cancel()
            if "error" in line["tiled_cfg5"]:  # the other ranks may be stuck in a collective of the leg: no barrier with them
                if rank == 0:
                    print(json.dumps(line), flush=True)
                os._exit(0)

    if rank == 0 and not args.no_churn and N == 200 and B == 1 and not args.dense_propagate and args.precision == "f64":
        try:
            line["churn"] = churn_leg(device)
        except Exception as e:
            line["churn"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0 and world == 1:
        rl = line.get("roofline")
        if rl is not None and not args.no_traffic:
            traffic, note = pmc_traffic(args, rl["kernel"])
            rl["traffic"] = traffic
            if note:
                rl["traffic_note"] = note
            if traffic and rl.get("launches_per_update"):
                rl["traffic_per_update"] = traffic * rl["launches_per_update"]
        if not args.no_parity:
            line["parity"] = parity_prefix(N, args.precision)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, structured=False)
            line["cpu_structured"] = cpu_baseline(N, structured=True, budget_s=8.0)
            line["speedup"] = {
                "gpu_over_cpu_dense": round(line["value"] / line["cpu_baseline"]["value"], 1),
                "algorithmic (cpu_structured / cpu_dense)": round(line["cpu_structured"]["value"] / line["cpu_baseline"]["value"], 2),
                "hardware (gpu / cpu_structured)": round(line["value"] / line["cpu_structured"]["value"], 1),
            }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
