"""Sharding of a batch of independent filters over the GPUs of one node (BASELINE cfg 4).

The filter recursion is sequential in time, so the only thing that shards is the batch dimension: filter
`i` of the job runs on rank `i // filters_per_rank`.  There is no data-path collective; torch.distributed
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used exactly twice:
  * scatter_streams: rank 0 generates every filter's input stream and scatters the per-rank slices,
  * gather_results:  per-filter results (pose, |Sigma|_F) are gathered to rank 0.
"""
import numpy as np


def build_streams(N, nfilt, first_index, nevents, base_seed=1234):
    """One stream per filter (seed = base_seed + global filter index); all filters share the
    event schedule of the reference's runner (eqf_vio/src/main.cpp:111-170)."""
    from . import synth

    duration = max(1.0, (nevents + 40) / 220.0 + 0.1)
    streams = [synth.make_stream(N, seed=base_seed + first_index + i, duration=duration) for i in range(nfilt)]
    events = list(streams[0].events())[:nevents]
    return streams, events


def pack(streams):
    """-> imu (K, B, 7), vision stamps (F, B), bearings (F, B, N, 3): the layouts of eqf_stream_upload."""
    imu = np.stack([s.imu for s in streams], axis=1)
    vst = np.stack([s.vision_stamps for s in streams], axis=1)
    bear = np.stack([s.bearings for s in streams], axis=1)
    return np.ascontiguousarray(imu), np.ascontiguousarray(vst), np.ascontiguousarray(bear)


def scatter_streams(dist, rank, world, N, filters_per_rank, nevents, device=None):
    """Returns (imu, vst, bear, events) of THIS rank's filters as numpy arrays.  dist=None: single process."""
    if dist is None:
        streams, events = build_streams(N, filters_per_rank, 0, nevents)
        return (*pack(streams), events)
    import torch

    B = filters_per_rank
    if rank == 0:
        streams, events = build_streams(N, B * world, 0, nevents)
        full = pack(streams)
        chunks = [[torch.from_numpy(np.ascontiguousarray(a[:, r * B:(r + 1) * B])) for r in range(world)] for a in full]
        if device is not None:
            chunks = [[c.to(device) for c in cs] for cs in chunks]
        shapes = [tuple(cs[0].shape) for cs in chunks]
    else:
        chunks, shapes, events = [None] * 3, None, None
    meta = [shapes, events]
    dist.broadcast_object_list(meta, src=0)
    shapes, events = meta
    mine = []
    for i in range(3):
        buf = torch.empty(shapes[i], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.scatter(buf, chunks[i] if rank == 0 else None, src=0)
        mine.append(buf.cpu().numpy())
    return mine[0], mine[1], mine[2], events


def gather_results(dist, rank, world, res, device=None):
    """res: (B, k) float64 per-filter results of this rank -> (B*world, k) on rank 0 (None elsewhere)."""
    if dist is None:
        return res
    import torch

    rt = torch.from_numpy(np.ascontiguousarray(res))
    if device is not None:
        rt = rt.to(device)
    gl = [torch.empty_like(rt) for _ in range(world)] if rank == 0 else None
    dist.gather(rt, gl, dst=0)
    if rank == 0:
        return torch.cat(gl).cpu().numpy()
    return None
