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
