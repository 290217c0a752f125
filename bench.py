#!/usr/bin/env python
"""bench.py -- EqF propagate+update steps/s on MI355X (BASELINE.json metric).

A "step" is one filter API call (processIMUData = Riccati propagate; processVisionData = propagate + update)
on the synthetic stream of SURVEY.md section 8d (IMU 200 Hz, bearings 20 Hz, N landmarks, template settings,
fastRiccati = false).  The whole input stream is resident in HBM before the timed region
(eqf_stream_upload); the timed region replays events by index through the C ABI.

    python bench.py                       # 1 GPU, 1 filter, N = 200  (BASELINE configs[1])
    python bench.py --filters-per-gpu 64  # 64 independent filters batched on one GPU
    torchrun ... bench.py --gpus 8 --filters-per-gpu 8   # BASELINE configs[3]: 64 filters over 8 GPUs

Multi-GPU: independent filters are sharded over ranks (no data-path collective); RCCL is used once to
scatter the pre-generated input streams from rank 0 and once to gather the results.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TF = {"f64": 78.6, "f32": 157.3}  # f32: MI355X_MICROARCH.md; f64: half the f32 rate (AMD datasheet)


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
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def run_events(fb, events):
    for kind, k in events:
        if kind == "imu":
            fb.stream_imu(k)
        else:
            fb.stream_vision(k)


def cpu_baseline(N, budget_s=20.0):
    """The fp64 C++ oracle (dense reference operation order) on one host core, bounded sample."""
    from eqf_vio_amd import synth
    from oracle import binding as ob

    st = synth.make_stream(N, seed=1234, duration=2.0)
    fo = ob.OracleFilter(synth.template_settings_dict())
    n = 0
    t0 = time.perf_counter()
    n_imu = n_vis = 0
    for kind, k in st.events():
        if kind == "imu":
            r = st.imu[k]
            fo.processIMUData(r[0], r[1:4], r[4:7])
            n_imu += 1
        else:
            fo.processVisionData(st.vision_stamps[k], st.ids, st.bearings[k])
            n_vis += 1
        n += 1
        if time.perf_counter() - t0 > budget_s and n_vis >= 2:
            break
    dt = time.perf_counter() - t0
    return {
        "value": n / dt,
        "unit": "steps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {n} events ({n_imu} IMU + {n_vis} vision) of the same N={N} stream, oracle/eqf_oracle.cpp "
        f"(dense fp64, reference op order), {dt:.1f} s on 1 of {os.cpu_count()} host cores",
    }


def roofline(fb, events, N, B, precision):
    """Per-kernel-class HIP-event timing over the timed region (second, profiled pass) -> dominant kernel."""
    fb.profile_enable(True)
    run_events(fb, events)
    prof = fb.profile()
    fb.profile_enable(False)
    # integrateUpToTime steps per burst (IMU calls + the vision call's own step leave together, csrc/eqf_burst.hpp)
    n_bursts = max(prof.get("k_imu_burst", (0, 0.0))[0], 1)
    steps_per_burst = len(events) / n_bursts
    n = 11 + 3 * N
    m = 2 * N
    ne = 6 + 3 * N
    esz = 8 if precision == "f64" else 4
    # Algorithmic work PER CALL of the path, divided below by the launches the class needed for it.
    n_upd = max(prof.get("k_update_prep", (0, 0.0))[0], 1)   # vision updates in the timed region
    chol_launches_per_update = max(prof.get("k_chol_step", (0, 0.0))[0], 1) / n_upd
    chain_flops = m**3 / 3.0 + m * m * (n + 7.0) + ne**3 / 3.0 + ne * ne * 11.0
    downdate_flops = 2.0 * n * n * m
    # 64-wide path: the covariance downdate (and the reductions / innovation lift) ride along in the chain launches
    embedded = prof.get("k_downdate", (0, 0.0))[0] == 0
    algo = {
        # propagate = read Sigma once + write Sigma once
        "k_propagate": ("hbm", 2.0 * n * n * esz * B),
        # a burst of K steps: SURVEY.md 8(d)'s per-step figure (read + write Sigma once per STEP) x K.  The burst kernels
        # keep Sigma in registers across the steps, so this "effective" rate may exceed what HBM could deliver step by step.
        "k_imu_burst": ("hbm", 2.0 * n * n * esz * B * steps_per_burst),
        # Cholesky of S + forward solves of n+7 rhs + Cholesky of Sigma_e + 11 rhs (+ Sigma - Y^T Y when it rides along),
        # spread over the chain launches of one update
        "k_chol_step": ("mfma", (chain_flops + (downdate_flops if embedded else 0.0)) * B / chol_launches_per_update),
        "k_downdate": ("mfma", downdate_flops * B),
        # dense backend (cfg 3): build F + two n^3 GEMMs = 4 n^3 flops per Riccati step (SURVEY.md 8d "mfma_dense_equiv")
        "k_dense_riccati": ("mfma", 4.0 * n**3 * B),
        "k_update_prep": ("hbm", (n * n * esz + 8.0 * (m * (n + 7) + m * m + ne * ne)) * B),
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
            if bound == "hbm":
                ach = work / (avg_us * 1e-6) / 1e9
                row.update(bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5))
            else:
                ach = work / (avg_us * 1e-6) / 1e12
                pk = MFMA_PEAK_TF["f64"]  # factorisation is always fp64
                if name in ("k_downdate", "k_dense_riccati"):
                    pk = MFMA_PEAK_TF[precision]
                if name == "k_imu_burst":
                    row["steps_per_burst"] = round(steps_per_burst, 2)
                if name == "k_chol_step":
                    row["launches_per_update"] = round(chol_launches_per_update, 2)
                    row["downdate_embedded"] = embedded
                row.update(bound="mfma", achieved=round(ach, 4), peak=pk, unit="TFLOP/s", frac=round(ach / pk, 5))
        rows.append(row)
    rows.sort(key=lambda r: -r["total_ms"])
    dom = next((r for r in rows if "bound" in r), None)
    out = None
    if dom:
        out = {k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
        out["traffic"] = None
        out["kernel"] = dom["kernel"]
        out["avg_us"] = dom["avg_us"]
    return out, rows


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
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=300, check=True)
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


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: RCCL path even for a single rank
        import torch.distributed as dist_

        dist = dist_
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = local_rank

    from eqf_vio_amd import binding, synth

    N, B = args.landmarks, args.filters_per_gpu
    nev = args.steps + args.warmup
    # ---- inputs: rank 0 generates every filter's stream, RCCL scatters them (one collective, before timing)
    from eqf_vio_amd import shard

    imu, vst, bear, events = shard.scatter_streams(dist, rank, world, N, B, nev, device="cuda" if dist is not None else None)
    ids = np.arange(N, dtype=np.int32)

    prec = binding.PRECISION_F64 if args.precision == "f64" else binding.PRECISION_F32
    fb = binding.FilterBatch(synth.template_settings_dict(), capacity=N, batch=B, device=device, precision=prec)
    if args.dense_propagate:
        fb.set_dense_propagate(True)
    fb.stream_upload(imu, vst, ids, bear)

    warm, timed = events[: args.warmup], events[args.warmup:]
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
    err = fb.device_error()

    # ---- results gathered to rank 0 (pose of filter 0 of each rank + Sigma Frobenius norms)
    res = np.zeros((B, 8))
    for b in range(B):
        e = fb.state_estimate(b)
        res[b, :3] = e["x"]
        res[b, 3:7] = e["q"]
        res[b, 7] = np.linalg.norm(fb.sigma(b))
    res = shard.gather_results(dist, rank, world, res, device="cuda" if dist is not None else None)

    n_timed = len(timed)
    total_steps = n_timed * B * world
    line = {
        "metric": "EqF propagate+update steps/sec at N=%d landmarks" % N,
        "value": total_steps / dt,
        "unit": "steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / n_timed,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
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
                         "(2 launches, Sigma read/written once; every step is the reference's step); EQF_IMU_BURST=0: one launch per call",
        },
        "device_error_flag": err,
        "sigma_fro_filter0": float(res[0, 7]) if res is not None else None,
    }
    if args.pmc_child:
        fb.sigma(0)  # one k_sigma_export launch: known byte count, calibrates FETCH_SIZE
        return
    if rank == 0 and not args.no_roofline:
        rl, rows = roofline(fb, timed, N, B, args.precision)
        line["roofline"] = rl
        line["kernels"] = rows
        # propagate-only and update-only time per call (kernel time from the HIP events), frames/s
        t = {r["kernel"]: (r["total_ms"], r["launches"]) for r in rows}
        n_imu_vis = len(timed)
        n_upd = max(t.get("k_update_prep", (0, 1))[1], 1)
        prop_ms = t.get("k_propagate", (0, 0))[0] + t.get("k_dense_riccati", (0, 0))[0] + t.get("k_imu_burst", (0, 0))[0]
        upd_ms = sum(t.get(k, (0, 0))[0] for k in ("k_update_prep", "k_chol_step", "k_update_reduce", "k_update_finish", "k_downdate"))
        line["per_call"] = {
            "propagate_us": round(prop_ms * 1e3 / max(n_imu_vis, 1), 3),
            "update_us": round(upd_ms * 1e3 / n_upd, 3),
            "frames_per_s": round(line["value"] / B / world * n_upd / max(n_imu_vis, 1), 1),
            "note": "kernel time of the profiled pass (dispatch gaps excluded); a frame = 10 IMU calls + 1 vision call",
        }
        if rl is not None and world == 1 and not args.no_traffic:
            del fb  # free the GPU for the profiled child runs
            traffic, note = pmc_traffic(args, rl["kernel"])
            rl["traffic"] = traffic
            if note:
                rl["traffic_note"] = note
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(N)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
