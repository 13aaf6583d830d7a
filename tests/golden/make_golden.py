"""Generates tests/golden/*.npz: small seeded inputs and the CPU oracle's
outputs for them.  The reference holds no golden vectors for this path and
cannot be built here, so these fixtures pin the oracle's
behaviour at the time it was validated against the known answers in
tests/test_oracle.py; both the oracle (CPU suite) and the HIP path (GPU suite)
must keep reproducing them.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from libviso_amd import synth                      # noqa: E402
from libviso_amd.abi import MatchParams            # noqa: E402
from oracle import pyoracle                        # noqa: E402


def main():
    # 1. matcher, both parameter sets, with duplicated patches (ties) and ragged counts
    seq = synth.make_sequence(77, 2, n_kp=160, width=300, height=140, dup_frac=0.1, ragged=True)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    n = seq["n"]
    kL1, dL1 = seq["kp"][1, 0, :n[1, 0]], seq["desc"][1, 0, :n[1, 0]]
    kR1, dR1 = seq["kp"][1, 1, :n[1, 1]], seq["desc"][1, 1, :n[1, 1]]
    kL0, dL0 = seq["kp"][0, 0, :n[0, 0]], seq["desc"][0, 0, :n[0, 0]]
    m_st, sc_st = pyoracle.match_desc(kL1, kR1, dL1, dR1, st, return_scored=True)
    m_tm, sc_tm = pyoracle.match_desc(kL1, kL0, dL1, dL0, tm, return_scored=True)
    np.savez_compressed(os.path.join(HERE, "matcher.npz"), F=seq["F"],
                        kL1=kL1, kR1=kR1, kL0=kL0, dL1=dL1.astype(np.int16), dR1=dR1.astype(np.int16),
                        dL0=dL0.astype(np.int16), m_stereo=m_st, m_temporal=m_tm,
                        scored=np.array([sc_st, sc_tm]))
    # 2. solver
    X, obs, tr_gt, param = synth.make_solver_case(5, m=120, outlier_frac=0.25, noise=0.3)
    samples = pyoracle.ransac_samples(3, 8, 50, 120)
    ok, tr, inl = pyoracle.ransac_minimize_reproj(X, obs, param, samples=samples)
    ok_gn, tr_gn, it_gn = pyoracle.minimize_reproj(X, obs, np.zeros(6), param, np.arange(0, 120, 3))
    np.savez_compressed(os.path.join(HERE, "solver.npz"), X=X, obs=obs, tr_gt=tr_gt, samples=samples,
                        ok=ok, tr=tr, inl=inl, ok_gn=ok_gn, tr_gn=tr_gn, it_gn=it_gn,
                        calib=np.array([param.base, param.f, param.cu, param.cv]))
    # 3. whole loop body over a short sequence
    seq = synth.make_sequence(78, 4, n_kp=150, width=320, height=160)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    out = pyoracle.sequence(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], seed=4, first_frame=10)
    np.savez_compressed(os.path.join(HERE, "sequence.npz"), F=seq["F"], kp=seq["kp"],
                        desc=seq["desc"].astype(np.int16), n=seq["n"], tr=out["tr"], ok=out["ok"],
                        n_inl=out["n_inl"], scored=out["scored"], m_out=out["m_out"],
                        calib=np.array([seq["param"].base, seq["param"].f, seq["param"].cu, seq["param"].cv]))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
