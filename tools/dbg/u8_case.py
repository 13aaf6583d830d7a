import numpy as np, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import libviso_amd
import test_gpu_union8 as T
from oracle import pyoracle as oracle
vrange = int(sys.argv[1]) if len(sys.argv) > 1 else 1020
rng = np.random.default_rng(8800 + vrange)
libviso_amd.set_matcher_variant(6)
for it in range(40):
    second = it % 4 != 3
    ratio = [0.9, 0.8, 1.0, 0.5, 0.9, 1.5][it % 6]
    kp1, kp2, d1, d2, mp = T._graded_case(rng, int(rng.integers(8, 70)), int(rng.integers(4, 240)), vrange, second, ratio)
    want = oracle.match_desc(kp1, kp2, d1, d2, mp)
    got = libviso_amd.match_desc(kp1, kp2, d1, d2, mp)
    got2 = libviso_amd.match_desc(kp1, kp2, d1, d2, mp)
    if not np.array_equal(got, got2): print("NONDETERMINISTIC", it)
    if np.array_equal(got, want):
        continue
    wd = {int(r[0]): (int(r[1]), int(r[2])) for r in want}
    gd = {int(r[0]): (int(r[1]), int(r[2])) for r in got}
    print("case", it, len(kp1), len(kp2), second, ratio, "want", len(want), "got", len(got))
    a = d1.astype(np.int64); b = d2.astype(np.int64)
    ha = np.clip((a + 1024) >> 3, 0, 255); hb = np.clip((b + 1024) >> 3, 0, 255)
    for i in sorted(set(wd) | set(gd)):
        if wd.get(i) != gd.get(i):
            dist = np.abs(kp1[i, 0] - kp2[:, 0]) + np.abs(kp1[i, 1] - kp2[:, 1])
            mem = dist <= mp.radius
            if dist[0] <= mp.radius: mem &= dist < dist[0]
            idx = np.nonzero(mem)[0]
            sad = np.abs(a[i][None] - b[idx]).sum(1); s8 = np.abs(ha[i][None] - hb[idx]).sum(1)
            o = np.argsort(s8, kind="stable")[:4]
            print("  query", i, "kp", kp1[i], "want", wd.get(i), "got", gd.get(i), "members", len(idx),
                  "top by s8:", [(int(idx[j]), int(s8[j]), int(sad[j])) for j in o], "min sad", int(sad.min()), int(idx[sad.argmin()]))
