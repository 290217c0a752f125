"""Offline CSV replay: the event loop and on-disk formats of the reference's runner (eqf_vio/src/main.cpp:111-203).

  IMU file     header line, then   t, wx, wy, wz, ax, ay, az                         (main.cpp:184-190)
  vision file  header line, then   t, N, (id, x, y, z) x N                           (main.cpp:192-203)
  output file  time, tx, ty, tz, qw, qx, qy, qz, vx, vy, vz, N, (id, x, y, z) x N     (main.cpp:96-97, VIOState.cpp:72-84)
  config       YAML with the `eqf:` block of eqf_vio/EQVIO_config_template.yaml and `main: {startTime: ...}`

Events are interleaved exactly like the reference: IMU while imu.stamp < meas.stamp, otherwise vision; samples with
stamp <= startTime are consumed but not processed (main.cpp:113-118, 126-131).  The filter object only needs the
reference's interface (processIMUData / processVisionData / stateEstimate / getTime), so the same driver runs the
MI355X filter (eqf_vio_amd.filter.VIOFilter) and, in the CPU tests, an adapter around the fp64 test checker.

    python -m eqf_vio_amd.replay imu.csv meas.csv [config.yaml] [-o out.csv]
"""
import argparse
import csv
import sys

import numpy as np

from .filter import IMUVelocity, VisionMeasurement


def read_imu_csv(path):
    rows = []
    with open(path) as f:
        rd = csv.reader(f)
        next(rd, None)  # header (main.cpp:61)
        for row in rd:
            if row:
                rows.append([float(x) for x in row[:7]])
    return np.array(rows).reshape(-1, 7)


def read_vision_csv(path):
    """-> list of (stamp, ids int32 (n,), bearings (n,3))"""
    out = []
    with open(path) as f:
        rd = csv.reader(f)
        next(rd, None)  # header (main.cpp:66)
        for row in rd:
            if not row:
                continue
            n = int(row[1])
            v = np.array([float(x) for x in row[2:2 + 4 * n]]).reshape(n, 4)
            out.append((float(row[0]), v[:, 0].astype(np.int32), v[:, 1:4].copy()))
    return out


def write_imu_csv(path, imu):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["t", "wx", "wy", "wz", "ax", "ay", "az"])
        for r in imu:
            w.writerow([repr(float(x)) for x in r])


def write_vision_csv(path, frames):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["t", "N", "id, x, y, z, ..."])
        for stamp, ids, y in frames:
            row = [repr(float(stamp)), len(ids)]
            for i, p in zip(ids, y):
                row += [int(i)] + [repr(float(x)) for x in p]
            w.writerow(row)


def settings_from_yaml(path):
    """The `eqf:` block (VIOFilterSettings.h:56-109) -> settings dict, plus main.startTime."""
    import yaml

    with open(path) as f:
        cfg = yaml.safe_load(f)
    e = dict(cfg.get("eqf", {}))
    d = {}
    for k, v in e.items():
        if k == "cameraOffset":
            assert v[0] == "xw"  # VIOFilterSettings.h:94
            d["cameraOffset_x"] = np.array(v[1:4], dtype=float)
            d["cameraOffset_q"] = np.array(v[4:8], dtype=float)
        else:
            d[k] = v
    start = float(cfg.get("main", {}).get("startTime", 0.0))
    return d, start


def format_state(t, est):
    """One output row (main.cpp:135-137 with operator<<(VIOState), VIOState.cpp:72-84)."""
    vals = [repr(float(t))] + [f"{x:.17g}" for x in (*est.pose_x, *est.pose_q, *est.velocity)] + [str(len(est.ids))]
    for i, p in zip(est.ids, est.bodyLandmarks):
        vals += [str(int(i))] + [f"{x:.17g}" for x in p]
    return ", ".join(vals)


def replay(filt, imu, frames, start_time=0.0, out=None):
    """Drives `filt` through the two streams; returns (#imu processed, #vision processed, list of (t, state))."""
    k, f = 0, 0
    n_imu = n_vis = 0
    states = []
    if out is not None:
        out.write("time, tx, ty, tz, qw, qx, qy, qz, vx, vy, vz, N, p1id, p1x, p1y, p1z, ..., ..., ..., ..., pNid, pNx, pNy, pNz\n")
    while k < len(imu) and f < len(frames):
        if imu[k, 0] < frames[f][0]:
            if imu[k, 0] > start_time:
                filt.processIMUData(IMUVelocity(imu[k, 0], imu[k, 1:4], imu[k, 4:7]))
                n_imu += 1
            k += 1
        else:
            stamp, ids, y = frames[f]
            if stamp > start_time:
                filt.processVisionData(VisionMeasurement(stamp, ids, y))
                n_vis += 1
            est = filt.stateEstimate()
            states.append((filt.getTime(), est))
            if out is not None:
                out.write(format_state(filt.getTime(), est) + "\n")
            f += 1
    return n_imu, n_vis, states


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("imu_file")
    ap.add_argument("meas_file")
    ap.add_argument("config_file", nargs="?")
    ap.add_argument("-o", "--output")
    ap.add_argument("--capacity", type=int, default=256)
    args = ap.parse_args(argv)
    from .filter import VIOFilter

    settings, start = ({}, 0.0) if not args.config_file else settings_from_yaml(args.config_file)
    filt = VIOFilter(settings, capacity=args.capacity)
    imu = read_imu_csv(args.imu_file)
    frames = read_vision_csv(args.meas_file)
    out = open(args.output, "w") if args.output else None
    n_imu, n_vis, _ = replay(filt, imu, frames, start, out)
    if out:
        out.close()
    print(f"Processed {n_imu} IMU and {n_vis} vision measurements.")  # main.cpp:172-173
    return 0


if __name__ == "__main__":
    sys.exit(main())
