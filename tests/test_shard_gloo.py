"""N > 1 path on CPU: frames shard over ranks with a one-frame halo, one
all-gather of per-frame records (gloo here, RCCL on GPUs), and the result does
not depend on the partition.  The per-rank engine in this CPU test is the
oracle; on the GPU box the same driver runs the HIP batch pipeline
(tests/test_gpu_shard.py, same driver, engine = shard.gpu_engine)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _oracle_engine(kp, desc, n, stereo, temporal, param, seed, first_frame):
    from oracle import pyoracle
    out = pyoracle.sequence(kp, desc, n, stereo, temporal, param, seed=seed, first_frame=first_frame)
    return out["tr"], out["ok"], out["n_inl"]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from libviso_amd import shard, synth
    from libviso_amd.abi import MatchParams
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seq = synth.make_sequence(21, 8, n_kp=250, width=400, height=200)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    tr, ok, n_inl = shard.run_sharded(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], 9,
                                      _oracle_engine, rank, world, dist=dist)
    q.put((rank, tr, ok, n_inl))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_covers_every_pair_once():
    from libviso_amd import shard
    for n_frames in (1, 2, 3, 8, 4541):
        for world in (1, 2, 3, 8):
            r = shard.partition(n_frames, world)
            assert r[0][0] == 0 and r[-1][1] == max(0, n_frames - 1)
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))          # halo: ranges share one frame
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == max(0, n_frames - 1)


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_rank():
    from libviso_amd import hostmath, shard, synth
    from libviso_amd.abi import MatchParams
    seq = synth.make_sequence(21, 8, n_kp=250, width=400, height=200)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    tr1, ok1, ni1 = shard.run_sharded(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], 9,
                                      _oracle_engine, 0, 1)
    assert ok1[1:].all()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tr, ok, ni in res:
        # same frames, same global RANSAC stream keys => identical records on every rank
        assert np.array_equal(tr, tr1) and np.array_equal(ok, ok1) and np.array_equal(ni, ni1)
    poses, valid = hostmath.chain_poses(tr1, ok1)
    assert len(poses) == 8 and valid == list(range(1, 8))
    # the chained trajectory follows the ground-truth chain
    gt, _ = hostmath.chain_poses(seq["tr_gt"], np.r_[0, np.ones(7, int)])
    assert np.abs(poses[-1] - gt[-1]).max() < 0.2
