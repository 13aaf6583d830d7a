"""The C++ host mirror (libviso_amd/host: viso.hpp, kitti_io.hpp) driven end to
end through its demo binary: KITTI calib.txt in, feature frames in, KITTI pose
file out — compared with the oracle's per-frame transforms chained on the host.
Chunked processing (one-frame halo between GPU batches) must not change the
result."""
import os
import struct
import subprocess

import numpy as np
import pytest

import libviso_amd
import kitti_tree
from libviso_amd import hostmath, synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu
DEMO = os.path.join(os.path.dirname(libviso_amd.SO_PATH), "viso_demo")


def _write_inputs(tmp, seq):
    nf, _, cap, _ = seq["kp"].shape
    with open(os.path.join(tmp, "features.bin"), "wb") as f:
        f.write(struct.pack("<4i", 0x5653464D, nf, cap, 121))
        f.write(seq["n"].astype("<i4").tobytes())
        f.write(seq["kp"].astype("<f4").tobytes())
        f.write(seq["desc"].astype("<f4").tobytes())
    with open(os.path.join(tmp, "calib.txt"), "w") as f:      # reference src/kitti.cpp:23-46
        for name, P in (("P0", seq["P1"]), ("P1", seq["P2"])):
            f.write(name + ": " + " ".join("%.12e" % v for v in P.reshape(-1)) + "\n")


def test_demo_pose_file_matches_oracle_chain(oracle, tmp_path):
    if not os.path.exists(DEMO):
        pytest.fail("libviso_amd/viso_demo is missing: run __graft_entry__.build()")
    seq = synth.make_sequence(31, 9, n_kp=400, width=620, height=188)
    tmp = str(tmp_path)
    _write_inputs(tmp, seq)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    want = oracle.sequence(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], seed=17)
    poses, valid = hostmath.chain_poses(want["tr"], want["ok"])
    results = []
    for chunk in (64, 3, 1):
        out = os.path.join(tmp, f"poses_{chunk}.txt")
        r = subprocess.run([DEMO, os.path.join(tmp, "features.bin"), os.path.join(tmp, "calib.txt"), out,
                            str(chunk), "17"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        got = np.loadtxt(out).reshape(-1, 12)
        assert got.shape[0] == len(poses) == 9              # identity + one pose per solved frame
        assert np.allclose(got[0], np.eye(4)[:3].reshape(-1))
        for g, p in zip(got, poses):
            assert np.abs(g - p[:3].reshape(-1)).max() < 2e-6 + 1e-5 * np.abs(p).max()   # "%lf" = 6 decimals
        results.append(got)
    assert np.array_equal(results[0], results[1]) and np.array_equal(results[0], results[2])


def test_kitti_driver_on_images(oracle, tmp_path):
    """viso_kitti = the reference's `kitti` executable (src/kitti.cpp:79-118) on the GPU pipeline:
    $KITTI_HOME/sequences/<seq>/{calib.txt,image_0/%06d.png,image_1/%06d.png} in, results/<seq>/<sha>/data/<seq>.txt out.
    Expected poses: oracle detector + extractor + loop body, chained on the host."""
    exe = os.path.join(os.path.dirname(libviso_amd.SO_PATH), "viso_kitti")
    if not os.path.exists(exe):
        pytest.fail("libviso_amd/viso_kitti is missing: run __graft_entry__.build()")
    seq = synth.make_image_sequence(12, 6, n_kp=1500, width=720, height=240)
    home = str(tmp_path)
    kitti_tree.write_tree(home, "07", seq)
    nf, cap = 6, 1200
    kp = np.zeros((nf, 2, cap, 2), np.float32); n = np.zeros((nf, 2), np.int32); desc = np.zeros((nf, 2, cap, 121), np.float32)
    for t in range(nf):
        for side in range(2):
            k, _ = oracle.detect_harris_binned(seq["images"][t, side])
            n[t, side] = len(k); kp[t, side, :len(k)] = k
            desc[t, side, :len(k)] = oracle.extract_descriptors(seq["images"][t, side], k)
    st, tm = MatchParams.stereo(oracle.F_from_P(seq["P1"], seq["P2"])), MatchParams.temporal()
    want = oracle.sequence(kp, desc, n, st, tm, seq["param"], seed=0)
    poses, valid = hostmath.chain_poses(want["tr"], want["ok"])
    env = dict(os.environ, KITTI_HOME=home)
    r = subprocess.run([exe, "abc123", "07"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    got = np.loadtxt(os.path.join(home, "results", "07", "abc123", "data", "07.txt")).reshape(-1, 12)
    assert got.shape[0] == len(poses) == 6
    for g, p in zip(got, poses):
        assert np.abs(g - p[:3].reshape(-1)).max() < 2e-6 + 1e-5 * np.abs(p).max()
    # begin/end arguments (src/kitti.cpp:86-94): frames 2..4 only -> identity + 2 poses
    r = subprocess.run([exe, "sub", "07", "2", "4"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert np.loadtxt(os.path.join(home, "results", "07", "sub", "data", "07.txt")).reshape(-1, 12).shape[0] == 3


def test_procrustes_start_for_the_gn_solve():
    """solveRigidMotion (reference src/estimation.cpp:29-51, dead on the reference's stereo path) wired as the optional
    closed-form start of minimize_reproj (viso.hpp: procrustes_tr / minimize_reproj_from_procrustes): close to the true
    motion, converges near the true pose, and to the pose the reference's zero start reaches whenever that start moves at
    all (Q7, src/viso.cpp:1610: a first step with only negative components ends the reference's solve at zero)."""
    exe = os.path.join(os.path.dirname(libviso_amd.SO_PATH), "viso_host_gputest")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "gputest ok" in r.stdout, r.stdout + r.stderr
