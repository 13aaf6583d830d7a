"""Seeded soak of the full pipeline against the oracle: small sequences with varying keypoint counts, clustering,
outlier shares, RANSAC seeds and global frame offsets.  A handful of cases by default; VISO_SOAK_CASES=500 for a long
run on the GPU box (VISO_SOAK_SEED: another stream of cases; VISO_SOAK_VARIANT: one matcher kernel only)."""
import os

import numpy as np
import pytest

import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu


def test_pipeline_soak(viso, oracle):
    rng = np.random.default_rng(int(os.environ.get("VISO_SOAK_SEED", "4242")))
    only = os.environ.get("VISO_SOAK_VARIANT")
    for c in range(int(os.environ.get("VISO_SOAK_CASES", "5"))):
        nf = int(rng.integers(3, 8))
        nkp = int(rng.choice([300, 800, 1500, 2000]))
        cf = float(rng.choice([0.0, 0.0, 0.4, 0.7]))
        of = float(rng.choice([0.1, 0.2, 0.5]))
        seq = synth.make_sequence(int(rng.integers(0, 1 << 30)), nf, n_kp=nkp, cluster_frac=cf, outlier_frac=of)
        st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
        seed, ff = int(rng.integers(0, 1 << 40)), int(rng.integers(0, 1 << 20))
        want = oracle.sequence(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], seed=seed, first_frame=ff)
        ctx = libviso_amd.Context(0)
        variants = (int(only),) if only else libviso_amd.MATCHER_VARIANTS
        libviso_amd.set_matcher_variant(variants[c % len(variants)], ctx)      # every matcher kernel of the build takes its turn
        b = libviso_amd.Batch(ctx, nf, seq["kp"].shape[2])
        b.upload(seq["kp"], seq["desc"], seq["n"])
        b.set_params(st, tm, seq["param"], seed=seed, first_frame=ff)
        b.run()
        tr, ok, n_inl = b.poses()
        what = (c, nf, nkp, cf, of, seed, ff)
        assert np.array_equal(ok, want["ok"]) and np.array_equal(n_inl, want["n_inl"]), what
        for t in range(1, nf):
            if ok[t]:
                A, B = libviso_amd.tr2mat(tr[t]), oracle.tr2mat(want["tr"][t])
                assert np.linalg.norm(A - B) / np.linalg.norm(B) < 1e-5, what
        b.close(); ctx.close()
        if c % 20 == 19:
            print(f"soak: {c + 1} cases clean", flush=True)
