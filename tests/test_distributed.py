"""world_size-2 gloo tests (CPU) of the multi-GPU path: frame-parallel sharding and the score gather.
The GPU kernels are not involved here: each rank scores its frames with the CPU oracle, exactly as each GPU rank
scores its frames with the engine; the test checks that sharding + gather + re-sequencing reproduce the
single-process result in stream order."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, n_frames, port, ret):
    for p in (HERE, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = importlib.import_module("watermarking-gpu_amd.frames")
    synth = importlib.import_module("watermarking-gpu_amd.synth")
    import oracle_lib as O
    R, C = 64, 96
    W = synth.synth_watermark(R, C)
    mine = frames.shard_frames(n_frames, rank, world)
    scores = []
    for f in mine:
        x = synth.synth_frame(R, C, frame=f, dtype=np.uint8)
        st, y, a = O.embed_u8(x, W)
        st, corr = O.detect_u8(y, W)
        scores.append(corr)
    out, _ = frames.gather_scores(torch.tensor(scores, dtype=torch.float32), n_frames, rank, world)
    # async form
    (recv, finish), work = frames.gather_scores(torch.tensor(scores, dtype=torch.float32), n_frames, rank, world, async_op=True)
    work.wait()
    assert torch.equal(finish(), out)
    if rank == 0:
        ret["scores"] = out.numpy().copy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [6, 7])
def test_frame_sharding_and_gather_gloo(n_frames):
    for p in (HERE, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    frames = importlib.import_module("watermarking-gpu_amd.frames")
    synth = importlib.import_module("watermarking-gpu_amd.synth")
    import oracle_lib as O
    O.lib()  # build before forking
    world = 2
    assert frames.shard_frames(7, 0, 2) == [0, 2, 4, 6] and frames.shard_frames(7, 1, 2) == [1, 3, 5]
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, n_frames, port, ret), nprocs=world, join=True)
    R, C = 64, 96
    W = synth.synth_watermark(R, C)
    ref = []
    for f in range(n_frames):
        x = synth.synth_frame(R, C, frame=f, dtype=np.uint8)
        st, y, a = O.embed_u8(x, W)
        ref.append(O.detect_u8(y, W)[1])
    np.testing.assert_array_equal(ret["scores"], np.array(ref, np.float32))


def test_gather_world1():
    frames = importlib.import_module("watermarking-gpu_amd.frames")
    s = torch.tensor([0.1, 0.2, 0.3])
    out, w = frames.gather_scores(s, 3, 0, 1)
    assert w is None and torch.equal(out, s)


def test_band_partition_covers_the_image():
    """bands.band_rows / band_with_halo (intra-frame sharding): contiguous, disjoint, complete, halos inside the image"""
    bands = importlib.import_module("watermarking-gpu_amd.bands")
    for rows in (4, 7, 64, 1080, 2160, 4321):
        for world in (1, 2, 3, 8):
            if rows < world * 3:
                continue
            prev = 0
            for r in range(world):
                r0, r1 = bands.band_rows(rows, r, world)
                assert r0 == prev and r1 > r0
                prev = r1
                g0, g1, lo, hi = bands.band_with_halo(rows, r, world)
                assert 0 <= g0 <= r0 and r1 <= g1 <= rows
                assert (g0 + lo, g0 + hi) == (r0, r1)
                assert lo in (0, bands.HALO) and (g1 - g0) - hi in (0, bands.HALO)
                assert (lo == 0) == (r == 0) and ((g1 - g0) == hi) == (r == world - 1)
            assert prev == rows
