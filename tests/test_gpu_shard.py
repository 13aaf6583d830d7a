"""The HIP pipeline under the shard driver (reference src/viso.cpp:1208-1222, 1313-1321):
contiguous frame ranges with a one-frame halo, RANSAC streams keyed on the GLOBAL frame index.  One process,
one device: the ranks of a W-way partition run one after the other through shard.gpu_engine and their records
are stitched exactly like the all-gather of run_sharded does.  Any partition must give the records of the
single range bit for bit, and those must agree with the oracle."""
import numpy as np
import pytest

import libviso_amd
from libviso_amd import shard, synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu

N_FRAMES = 41      # 40 pairs: W = 3 and W = 8 cut mid-sequence at uneven places


@pytest.fixture(scope="module")
def seq():
    return synth.make_sequence(77, N_FRAMES, n_kp=2000)


def _run(seq, world, seed=5):
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    eng = shard.gpu_engine(0)
    parts = [shard.local_records(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], seed, eng, r, world)
             for r in range(world)]
    return shard.stitch(parts, N_FRAMES)


def test_partitions_are_bit_identical_and_match_oracle(viso, oracle, seq):
    tr1, ok1, ni1 = _run(seq, 1)
    assert ok1[1:].all() and ok1[0] == 0
    for world in (2, 3, 8):
        tr, ok, ni = _run(seq, world)
        assert np.array_equal(tr, tr1) and np.array_equal(ok, ok1) and np.array_equal(ni, ni1), world
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    want = oracle.sequence(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], seed=5)
    assert np.array_equal(ok1, want["ok"]) and np.array_equal(ni1, want["n_inl"])
    for t in range(1, N_FRAMES):
        a, r = libviso_amd.tr2mat(tr1[t]), oracle.tr2mat(want["tr"][t])
        assert np.linalg.norm(a - r) / np.linalg.norm(r) < 1e-5, t
    # and the chained trajectory (host prefix product, :1319) is the oracle's
    from libviso_amd import hostmath
    pa, va = hostmath.chain_poses(tr1, ok1)
    pb, vb = hostmath.chain_poses(want["tr"], want["ok"])
    assert va == vb and np.linalg.norm(pa[-1] - pb[-1]) / np.linalg.norm(pb[-1]) < 1e-5


def test_run_sharded_single_rank_uses_the_hip_engine(viso, seq):
    """run_sharded itself (world 1: no process group) on the HIP engine, ragged range sizes excluded."""
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    k = 9
    tr, ok, ni = shard.run_sharded(seq["kp"][:k], seq["desc"][:k], seq["n"][:k], st, tm, seq["param"], 5,
                                   shard.gpu_engine(0), 0, 1)
    tr1, ok1, ni1 = _run(seq, 1)
    # the first k frames of the long sequence: same global frame keys -> same records
    assert np.array_equal(tr, tr1[:k]) and np.array_equal(ok, ok1[:k]) and np.array_equal(ni, ni1[:k])
