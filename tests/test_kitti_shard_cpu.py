"""Host side of the sharded KITTI runner (BASELINE configs[3]) without a GPU: ranges, rank files, the gather, the
pose chain and the pose file.  The per-range engine is injected (records that are a fixed function of the absolute
frame index), so what is tested here is exactly what differs between W = 1 and W > 1; the HIP engine under the same
code runs in tests/test_gpu_kitti_shard.py."""
import os
import socket
import struct
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "libviso_amd", "viso_kitti")
P1 = "7.188560000000e+02 0 6.071928000000e+02 0 0 7.188560000000e+02 1.852157000000e+02 0 0 0 1 0"
P2 = "7.188560000000e+02 0 6.071928000000e+02 -3.861448000000e+02 0 7.188560000000e+02 1.852157000000e+02 0 0 0 1 0"


def _tree(home, seq, n_frames, begin=0):
    """File names only: the host code under test counts frames by opening them, it does not decode here."""
    base = os.path.join(home, "sequences", seq)
    for side in (0, 1):
        os.makedirs(os.path.join(base, f"image_{side}"), exist_ok=True)
        for t in range(begin, begin + n_frames):
            open(os.path.join(base, f"image_{side}", "%06d.png" % t), "wb").close()
    with open(os.path.join(base, "calib.txt"), "w") as f:
        f.write(f"P0: {P1}\nP1: {P2}\n")
    return base


def _fake_record(frame):
    """Record of the pair ending at absolute frame `frame`: depends on nothing else (like the HIP engine's, whose
    RANSAC stream is keyed on the absolute frame index)."""
    rng = np.random.default_rng(1000 + frame)
    tr = np.r_[rng.normal(0, 0.02, 3), rng.normal(0, 0.1, 2), rng.uniform(0.5, 1.5)]
    ok = 0.0 if frame % 7 == 3 else 1.0          # some frames fail: nothing is pushed for them (src/viso.cpp:1287,1323)
    return np.r_[tr, ok, float(50 + frame % 11)]


def _fake_engine(short_at=None):
    def run(seq_base, begin, first, last):
        rows = []
        for t in range(first + 1, last + 1):
            if short_at is not None and begin + t >= short_at:     # an undecodable image: the generator stops (src/viso.h:94-96)
                break
            rows.append(_fake_record(begin + t))
        return np.array(rows).reshape(-1, 8)
    return run


def _write_rank_file(path, first, last, rec):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", 0x56534B52, first, last, len(rec)))
        for i, r in enumerate(rec):
            f.write(struct.pack("<6d4i", *r[:6], int(r[6]), int(r[7]), 0, 0))


def _numpy_pose_lines(rec):
    from libviso_amd import hostmath
    poses, _ = hostmath.chain_poses(rec[:, :6], rec[:, 6])
    return np.array([p[:3].reshape(-1) for p in poses])


@pytest.fixture(scope="module")
def host():
    from libviso_amd import kitti_shard
    if not os.path.exists(kitti_shard.HOST_SO) or not os.path.exists(EXE):
        pytest.fail("libviso_host.so / viso_kitti missing: run __graft_entry__.build()")
    return kitti_shard.load_host()


def test_partition_is_the_same_in_cpp_and_python(host, tmp_path):
    """viso::partition is reached through `viso_kitti --gather`: rank files cut by the Python rule must be accepted."""
    from libviso_amd import kitti_shard
    home = str(tmp_path)
    for n_frames, world in ((11, 3), (2, 2), (9, 8), (30, 4)):
        seq = f"s{n_frames}w{world}"
        _tree(home, seq, n_frames)
        for r, (a, b) in enumerate(kitti_shard.partition(n_frames, world)):
            rec = [_fake_record(t) for t in range(a + 1, b + 1)]
            _write_rank_file(os.path.join(home, "results", seq, "x", "shards", f"{seq}.{r}of{world}.rec"), a, b, rec)
        r = subprocess.run([EXE, "x", seq, "--gather", str(world)], capture_output=True, text=True,
                           env=dict(os.environ, KITTI_HOME=home), timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        got = np.loadtxt(os.path.join(home, "results", seq, "x", "data", seq + ".txt")).reshape(-1, 12)
        want = _numpy_pose_lines(np.array([_fake_record(t) for t in range(1, n_frames)]).reshape(-1, 8))
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 2e-6 + 1e-9 * np.abs(want).max()          # "%lf": six decimals
    # a rank file of another cut is refused, not silently stitched
    _write_rank_file(os.path.join(home, "results", "s11w3", "x", "shards", "s11w3.1of3.rec"), 3, 7, [_fake_record(t) for t in range(4, 8)])
    r = subprocess.run([EXE, "x", "s11w3", "--gather", "3"], capture_output=True, text=True, env=dict(os.environ, KITTI_HOME=home), timeout=60)
    assert r.returncode != 0 and "expected" in r.stderr


def test_pose_file_format_and_chain(host, tmp_path):
    """viso_kitti_write_poses = chain (src/viso.cpp:1315-1321) + savePoses (src/kitti.cpp:49-64): first line identity,
    12 x %lf per line, failed frames push nothing."""
    from libviso_amd import kitti_shard
    rec = np.array([_fake_record(t) for t in range(1, 40)])
    out = str(tmp_path / "a" / "b" / "poses.txt")
    n = kitti_shard.write_poses(host, out, rec)
    lines = open(out).read().splitlines()
    assert n == len(lines) == 1 + int(rec[:, 6].sum())
    assert lines[0] == "1.000000 0.000000 0.000000 0.000000 0.000000 1.000000 0.000000 0.000000 0.000000 0.000000 1.000000 0.000000"
    got = np.array([[float(v) for v in ln.split()] for ln in lines])
    assert np.abs(got - _numpy_pose_lines(rec)).max() < 2e-6


def _worker(rank, world, port, home, sha, short_at, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from libviso_amd import kitti_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = kitti_shard.load_host()
    n_frames, full, out = kitti_shard.run_rank(home, sha, "09", 4, 2**31 - 1, rank, world, L, _fake_engine(short_at), dist)
    q.put((rank, n_frames, full, out))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.timeout(300)
@pytest.mark.parametrize("short_at", [None, 17])
def test_two_and_three_gloo_ranks_write_the_file_of_one(host, tmp_path, short_at):
    from libviso_amd import kitti_shard
    home = str(tmp_path)
    _tree(home, "09", 23, begin=4)                       # frames 000004 .. 000026
    n1, full1, out1 = kitti_shard.run_rank(home, "one", "09", 4, 2**31 - 1, 0, 1, host, _fake_engine(short_at))
    assert n1 == 23 and len(full1) == (22 if short_at is None else short_at - 5)
    one = open(out1, "rb").read()
    for world in (2, 3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, home, f"w{world}", short_at, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=240) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for rank, n_frames, full, out in res:
            assert n_frames == 23 and np.array_equal(full, full1)          # every rank holds the whole record list
            assert (out is not None) == (rank == 0)
        assert open(os.path.join(home, "results", "09", f"w{world}", "data", "09.txt"), "rb").read() == one


def test_launcher_refuses_a_world_that_is_not_gpus(tmp_path):
    env = dict(os.environ, KITTI_HOME=str(tmp_path), WORLD_SIZE="1", RANK="0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "libviso_amd.kitti_shard", "x", "00", "--gpus", "2", "--backend", "gloo"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=120)
    assert r.returncode == 7 and "WORLD_SIZE" in r.stderr


def test_broken_rank_files_are_refused(host, tmp_path):
    """A rank file with an impossible record count, a truncated one, a missing one: `--gather` reports and writes nothing
    (no pose file assembled from half a sequence).  The same inputs run clean under -fsanitize=address,undefined."""
    from libviso_amd import kitti_shard
    home = str(tmp_path)
    _tree(home, "05", 11)
    shards = os.path.join(home, "results", "05", "x", "shards")
    ranges = kitti_shard.partition(11, 3)
    for r, (a, b) in enumerate(ranges):
        _write_rank_file(os.path.join(shards, f"05.{r}of3.rec"), a, b, [_fake_record(t) for t in range(a + 1, b + 1)])
    env = dict(os.environ, KITTI_HOME=home)
    out = os.path.join(home, "results", "05", "x", "data", "05.txt")
    bad = os.path.join(shards, "05.1of3.rec")
    a, b = ranges[1]
    for blob in (struct.pack("<4i", 0x56534B52, a, b, 1000000),                 # more records than the range holds
                 struct.pack("<4i", 0x56534B52, a, b, b - a) + b"\0" * 10,      # truncated
                 struct.pack("<4i", 0x12345678, a, b, 0)):                      # not a rank file
        open(bad, "wb").write(blob)
        r = subprocess.run([EXE, "x", "05", "--gather", "3"], capture_output=True, text=True, env=env, timeout=60)
        assert r.returncode == 3 and "cannot read" in r.stderr and not os.path.exists(out)
    os.remove(bad)
    r = subprocess.run([EXE, "x", "05", "--gather", "3"], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 3 and not os.path.exists(out)


def _raising_engine(bad_rank, rank):
    def run(seq_base, begin, first, last):
        if rank == bad_rank:
            raise RuntimeError("viso_kitti_run_range failed with -2: no HIP device (injected)")
        return _fake_engine()(seq_base, begin, first, last)
    return run


def _failing_worker(rank, world, port, home, bad_rank, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from libviso_amd import kitti_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = kitti_shard.load_host()
    try:
        kitti_shard.run_rank(home, "bad", "09", 4, 2**31 - 1, rank, world, L, _raising_engine(bad_rank, rank), dist)
        q.put((rank, "no error"))
        code = 0
    except kitti_shard.RankFailed as e:
        q.put((rank, str(e)))
        code = 8
    dist.destroy_process_group()          # every rank is behind the same all-gather: a clean teardown is possible
    sys.exit(code)


@pytest.mark.timeout(300)
def test_a_rank_whose_range_fails_takes_every_rank_down_without_a_hang(host, tmp_path):
    """ADVICE r3 / VERDICT r3 1(b): a rank that fails before its all_gather used to wait in a barrier its peers never
    reach.  Now the failure travels in the rank's block of the one all-gather: every rank raises RankFailed behind the
    collective, no pose file is written, the processes end (exit code 8) instead of hanging."""
    from libviso_amd import kitti_shard
    home = str(tmp_path)
    _tree(home, "09", 23, begin=4)
    with pytest.raises(kitti_shard.RankFailed):                     # one rank, no process group
        kitti_shard.run_rank(home, "bad", "09", 4, 2**31 - 1, 0, 1, host, _raising_engine(0, 0))
    for world, bad_rank in ((2, 1), (3, 0)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_failing_worker, args=(r, world, port, home, bad_rank, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 8, p.exitcode
        assert all(f"[{bad_rank}]" in msg for msg in res.values()), res
    assert not os.path.exists(os.path.join(home, "results", "09", "bad", "data", "09.txt"))


def test_cli_error_paths_exit_nonzero_and_never_wait_for_a_collective(tmp_path):
    """main(): (a) every rank's range fails (no HIP device in this container, or undecodable images on a GPU box would
    come back short instead) -> exit 8 through the collective; (b) a failure before the block shape is known
    (begin > end is refused by viso_kitti_count_frames) -> the rank leaves at once with code 9, joining nothing."""
    import pngutil
    home = str(tmp_path)
    base = os.path.join(home, "sequences", "00")
    rng = np.random.default_rng(0)
    for side in (0, 1):
        os.makedirs(os.path.join(base, f"image_{side}"))
        for t in range(3):
            pngutil.write_gray_png(os.path.join(base, f"image_{side}", "%06d.png" % t), rng.integers(0, 255, (40, 64)).astype(np.uint8))
    with open(os.path.join(base, "calib.txt"), "w") as f:
        f.write(f"P0: {P1}\nP1: {P2}\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(KITTI_HOME=home, PYTHONPATH=ROOT)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, "-m", "libviso_amd.kitti_shard", "x", "00", "--gpus", "2", "--backend", "gloo", "--same-device"],
                           capture_output=True, text=True, env=env, cwd=ROOT, timeout=240)
        assert r.returncode != 0 and "failed" in r.stderr, r.stdout + r.stderr
        assert not os.path.exists(os.path.join(home, "results", "00", "x", "data", "00.txt"))
    r = subprocess.run([sys.executable, "-m", "libviso_amd.kitti_shard", "x", "00", "5", "3", "--gpus", "1", "--backend", "gloo", "--force-collective"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=240)
    assert r.returncode == 9 and "bad argument" in r.stderr, r.stdout + r.stderr


def test_a_group_of_one_goes_through_the_collective(host, tmp_path):
    """--force-collective at W = 1 (gloo here, RCCL in tests/test_gpu_kitti_shard.py): the records come back from a real
    all_gather and are what the no-group path returns."""
    import torch.distributed as dist
    from libviso_amd import kitti_shard
    home = str(tmp_path)
    _tree(home, "09", 9, begin=4)
    n0, full0, out0 = kitti_shard.run_rank(home, "plain", "09", 4, 2**31 - 1, 0, 1, host, _fake_engine())
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        n1, full1, out1 = kitti_shard.run_rank(home, "group", "09", 4, 2**31 - 1, 0, 1, host, _fake_engine(), dist)
    finally:
        dist.destroy_process_group()
    assert n0 == n1 and np.array_equal(full0, full1) and open(out0, "rb").read() == open(out1, "rb").read()


def test_reference_pose_list(host, tmp_path):
    """`Mat pose = poses.back(); pose = pose*tr_mat.inv(); poses.push_back(pose.clone());`
    (reference src/viso.cpp:1317-1321) overwrites poses.back() before pushing, so the reference's file is
    [P1, ..., Pn, Pn]; the default here stays [I, P1, ..., Pn].  Both against hostmath.chain_poses, through the Python
    writer and through `viso_kitti --gather --reference-pose-list`."""
    from libviso_amd import hostmath, kitti_shard
    rec = np.array([_fake_record(t) for t in range(1, 30)])
    for quirk in (False, True):
        out = str(tmp_path / f"p{int(quirk)}.txt")
        n = kitti_shard.write_poses(host, out, rec, reference_pose_list=quirk)
        poses, _ = hostmath.chain_poses(rec[:, :6], rec[:, 6], aliasing_quirk=quirk)
        got = np.loadtxt(out).reshape(-1, 12)
        assert n == len(poses) == len(got)
        assert np.abs(got - np.array([p[:3].reshape(-1) for p in poses])).max() < 2e-6
    a, b = np.loadtxt(str(tmp_path / "p0.txt")), np.loadtxt(str(tmp_path / "p1.txt"))
    assert np.array_equal(a[1:], b[:-1]) and np.array_equal(b[-1], b[-2]) and not np.array_equal(a[0], b[0])
    home = str(tmp_path)
    _tree(home, "07", 30)
    for r, (f0, f1) in enumerate(kitti_shard.partition(30, 2)):
        _write_rank_file(os.path.join(home, "results", "07", "x", "shards", f"07.{r}of2.rec"), f0, f1, [_fake_record(t) for t in range(f0 + 1, f1 + 1)])
    r = subprocess.run([EXE, "x", "07", "--gather", "2", "--reference-pose-list"], capture_output=True, text=True,
                       env=dict(os.environ, KITTI_HOME=home), timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(np.loadtxt(os.path.join(home, "results", "07", "x", "data", "07.txt")), b)
