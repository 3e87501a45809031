"""Frame-parallel sharding of a video stream over the GPUs of one node.

Frames are the independent units of the path (main.cpp:326-331: each frame's embed/detect depends only on that
frame and the constant W), so a stream shards with NO data-path collective: frame i goes to rank i mod G
(round-robin keeps the output re-sequencing latency minimal), every rank owns its own engine and W copy, and the
only exchange is the gather of the per-frame detector scores (4 B/frame) -- an all-gather over RCCL on GPUs
(`backend="nccl"`), over gloo in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world):
    """indices of the frames rank `rank` processes: i mod world == rank"""
    return list(range(rank, n_frames, world))


def gather_scores(local_scores, n_frames, rank, world, device=None, async_op=False, force_collective=False):
    """all-gather of per-frame scores and re-sequencing into stream order.

    local_scores: 1-D float32 tensor, the scores of shard_frames(n_frames, rank, world) in that order.
    Returns (scores[n_frames] float32 in frame order, work handle or None).  Ranks whose shard is one frame
    shorter pad with NaN (frames shard unevenly when world does not divide n_frames).
    force_collective: run the all-gather even with one rank (an initialised process group is then required): the
    single-GPU rehearsal of the RCCL path (tests/test_gpu_rccl.py, bench.py with WM_BENCH_FORCE_DIST=1)."""
    per = (n_frames + world - 1) // world
    dev = device if device is not None else local_scores.device
    send = torch.full((per,), float("nan"), dtype=torch.float32, device=dev)
    send[:local_scores.numel()] = local_scores.to(dev)
    if world == 1 and not force_collective:
        return send[:n_frames].clone(), None
    recv = torch.empty(per * world, dtype=torch.float32, device=dev)
    work = dist.all_gather_into_tensor(recv, send, async_op=async_op)

    def finish():
        # recv[r*per + k] is frame k*world + r
        out = recv.view(world, per).t().reshape(-1)[:n_frames]
        return out

    if async_op:
        return (recv, finish), work
    return finish(), None
