"""BASELINE configs[3] (one KITTI sequence sharded over the GPUs of a node, trajectory gathered once) on a
synthetic KITTI tree: any number of ranks writes the byte-identical pose file.

  viso_kitti                                   one process (the reference's flow, src/kitti.cpp:79-118)
  viso_kitti --gpus W --same-device            W forked rank processes, rank files, gather on the host
  python -m libviso_amd.kitti_shard --gpus W   W torch.distributed ranks, records all-gathered (gloo here: the box
                                               has one GPU; --backend nccl is the same code path with RCCL)
Frames shard with a one-frame halo (src/viso.cpp:1208-1222), RANSAC streams are keyed on the absolute frame index,
the pose chain (src/viso.cpp:1315-1321) runs over the gathered records."""
import os
import subprocess
import sys

import numpy as np
import pytest

import kitti_tree
import libviso_amd
from libviso_amd import hostmath, synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(os.path.dirname(libviso_amd.SO_PATH), "viso_kitti")
N_FRAMES = 11          # 10 pairs: W = 3 cuts them 4 + 3 + 3
FIRST = 5              # the files are 000005.png ...: `begin` is not 0


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    home = str(tmp_path_factory.mktemp("kitti"))
    seq = synth.make_image_sequence(41, N_FRAMES, n_kp=1500, width=720, height=240)
    kitti_tree.write_tree(home, "03", seq, first_index=FIRST)
    return home, seq


def _pose_file(home, sha):
    return os.path.join(home, "results", "03", sha, "data", "03.txt")


def _run(cmd, home):
    env = dict(os.environ, KITTI_HOME=home, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_any_number_of_ranks_writes_the_same_pose_file(oracle, tree):
    home, seq = tree
    if not os.path.exists(EXE):
        pytest.fail("libviso_amd/viso_kitti is missing: run __graft_entry__.build()")
    out = _run([EXE, "w1", "03", str(FIRST)], home)
    one = open(_pose_file(home, "w1"), "rb").read()
    assert f"frames {N_FRAMES} " in out
    assert len(one.splitlines()) == N_FRAMES        # identity + one pose per solved pair
    # forked rank processes + rank files + host gather (C++ only)
    for w in (2, 3):
        out = _run([EXE, f"fork{w}", "03", str(FIRST), "--gpus", str(w), "--same-device"], home)
        assert f"ranks {w}" in out
        assert open(_pose_file(home, f"fork{w}"), "rb").read() == one, w
        for r in range(w):
            assert os.path.exists(os.path.join(home, "results", "03", f"fork{w}", "shards", f"03.{r}of{w}.rec"))
    # torch.distributed ranks + one all-gather of the records
    for w in (1, 2, 3):
        out = _run([sys.executable, "-m", "libviso_amd.kitti_shard", f"dist{w}", "03", str(FIRST), "--gpus", str(w),
                    "--backend", "gloo", "--same-device"], home)
        assert f"ranks {w}" in out
        assert open(_pose_file(home, f"dist{w}"), "rb").read() == one, w
    # chunking inside a rank does not matter either
    _run([EXE, "chunk3", "03", str(FIRST), "--gpus", "2", "--same-device", "--chunk", "3"], home)
    assert open(_pose_file(home, "chunk3"), "rb").read() == one

    # and the file is the oracle's: detector + extractor + loop body on the CPU, RANSAC keys = absolute frame index
    cap = 1200
    kp = np.zeros((N_FRAMES, 2, cap, 2), np.float32); n = np.zeros((N_FRAMES, 2), np.int32)
    desc = np.zeros((N_FRAMES, 2, cap, 121), np.float32)
    for t in range(N_FRAMES):
        for side in range(2):
            k, _ = oracle.detect_harris_binned(seq["images"][t, side])
            n[t, side] = len(k); kp[t, side, :len(k)] = k
            desc[t, side, :len(k)] = oracle.extract_descriptors(seq["images"][t, side], k)
    st, tm = MatchParams.stereo(oracle.F_from_P(seq["P1"], seq["P2"])), MatchParams.temporal()
    want = oracle.sequence(kp, desc, n, st, tm, seq["param"], seed=0, first_frame=FIRST)
    poses, _ = hostmath.chain_poses(want["tr"], want["ok"])
    got = np.loadtxt(_pose_file(home, "w1")).reshape(-1, 12)
    assert got.shape[0] == len(poses)
    for g, p in zip(got, poses):
        assert np.abs(g - p[:3].reshape(-1)).max() < 2e-6 + 1e-5 * np.abs(p).max()


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_rccl_process_group_of_one(tree):
    """the torch.distributed runner THROUGH RCCL on the one-GPU box.  `--force-collective` builds the
    nccl (= RCCL) process group at world size 1 -- init_process_group("nccl", device_id=...), the float64 CUDA record
    tensor through all_gather, the stats all_gather, all_reduce(MAX) of the wall time, destroy_process_group -- under the
    driver's launcher (`python -m torch.distributed.run --nproc-per-node 1`) and on its own.  Same pose file as the
    one-process C++ runner and the gloo ranks."""
    home, _ = tree
    _run([EXE, "ref1", "03", str(FIRST)], home)
    one = open(_pose_file(home, "ref1"), "rb").read()
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env = dict(env_clean, KITTI_HOME=home, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for sha, launcher in (("rccl_torchrun", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                                             "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "-m"]),
                          ("rccl_alone", [sys.executable, "-m"])):
        r = subprocess.run(launcher + ["libviso_amd.kitti_shard", sha, "03", str(FIRST), "--gpus", "1", "--backend", "nccl",
                                       "--force-collective"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        assert "backend rccl" in r.stdout and "'ranks': 1" in r.stdout and "cuda" in r.stdout, r.stdout
        assert open(_pose_file(home, sha), "rb").read() == one, sha


def test_reference_pose_list_switch(tree):
    """`--reference-pose-list`: [P1, ..., Pn, Pn], the list reference src/viso.cpp:1317-1321 actually
    writes, from the same records: line k of it is line k + 1 of the default file, the last line twice."""
    home, _ = tree
    _run([EXE, "dflt", "03", str(FIRST)], home)
    _run([EXE, "quirk", "03", str(FIRST), "--reference-pose-list"], home)
    _run([EXE, "quirk2", "03", str(FIRST), "--reference-pose-list", "--gpus", "2", "--same-device"], home)
    a = open(_pose_file(home, "dflt")).read().splitlines()
    b = open(_pose_file(home, "quirk")).read().splitlines()
    assert len(a) == len(b) and a[1:] == b[:-1] and b[-1] == b[-2]
    assert open(_pose_file(home, "quirk2")).read().splitlines() == b


def test_runner_reports_where_the_time_went(tree):
    home, _ = tree
    out = _run([EXE, "stats", "03", str(FIRST), "--decode-threads", "3"], home)
    assert "decode: 3 threads" in out and "GPU stamps: upload" in out and f"{N_FRAMES} frames in" in out


def test_decode_threads_and_chunking_do_not_change_the_poses(tree):
    """The images of a chunk are decoded by worker threads straight into the pinned upload buffer of the chunk's slot while
    the GPU works on the previous chunk; the halo frame is copied from the other slot.  Any number of threads, any chunk
    size: the same pose file.  And the sequence still ends at the first frame that cannot be decoded (src/viso.h:94-96)."""
    import shutil
    home, _ = tree
    _run([EXE, "t1", "03", str(FIRST), "--decode-threads", "1", "--chunk", "64"], home)
    one = open(_pose_file(home, "t1"), "rb").read()
    for sha, threads, chunk in (("t7c2", "7", "2"), ("t2c1", "2", "1"), ("t16c4", "16", "4")):
        _run([EXE, sha, "03", str(FIRST), "--decode-threads", threads, "--chunk", chunk], home)
        assert open(_pose_file(home, sha), "rb").read() == one, sha
    # a damaged frame in the middle: the run ends there (frames FIRST .. FIRST+5 remain), for every thread count
    seq_dir = os.path.join(home, "sequences", "03")
    bad = os.path.join(seq_dir, "image_1", "%06d.png" % (FIRST + 6))
    keep = bad + ".keep"
    shutil.copy(bad, keep)
    try:
        with open(bad, "r+b") as f:
            f.seek(60); f.write(b"\xff" * 40)
        for sha, threads in (("cut1", "1"), ("cut5", "5")):
            out = _run([EXE, sha, "03", str(FIRST), "--decode-threads", threads, "--chunk", "4"], home)
            assert "frames 6 " in out, out
            assert open(_pose_file(home, sha), "rb").read() == b"".join(one.splitlines(keepends=True)[:6]), sha
    finally:
        shutil.move(keep, bad)


def test_sub_range_with_begin_and_end(tree):
    """begin/end (src/kitti.cpp:86-94) under sharding: frames FIRST+2 .. FIRST+8, W = 1 and 2."""
    home, _ = tree
    _run([EXE, "sub1", "03", str(FIRST + 2), str(FIRST + 8)], home)
    _run([EXE, "sub2", "03", str(FIRST + 2), str(FIRST + 8), "--gpus", "2", "--same-device"], home)
    a, b = open(_pose_file(home, "sub1"), "rb").read(), open(_pose_file(home, "sub2"), "rb").read()
    assert a == b and len(a.splitlines()) == 7


def test_bench_gpus_2_starts_its_own_ranks_and_reports_them():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the bench launches two ranks itself (here both on
    the one device of the box, gloo for the record gather) and rank 0 prints a line for N = 2 whose `collective` object
    says how many ranks took part in the all-gather."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(VISO_BENCH_SAME_DEVICE="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--frames", "24",
                        "--kp", "600", "--steps", "3", "--warmup", "1", "--min-region-seconds", "0", "--no-cpu",
                        "--no-streaming", "--no-images"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["collective"]["ranks"] == 2 and d["collective"]["gathered_records"] == 2 * 24
    assert d["end_to_end"]["poses_ok"] > 0
    assert "rank 0/2 started" in r.stderr and "rank 1/2 started" in r.stderr


def test_bench_force_collective_goes_through_rccl():
    """VERDICT r3 1(a), bench side: `--force-collective` at N = 1 runs barrier / all_reduce(MAX) / the record all_gather on
    an nccl (= RCCL) process group of one rank, with CUDA tensors."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-collective", "--frames", "24", "--kp", "600", "--steps", "3",
                        "--warmup", "1", "--min-region-seconds", "0", "--no-cpu", "--no-streaming", "--no-images"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][0])
    assert d["n_gpus"] == 1 and d["collective"]["backend"] == "rccl" and d["collective"]["ranks"] == 1
    assert d["collective"]["gathered_records"] == 24 and d["end_to_end"]["poses_ok"] > 0
