"""The arithmetic match_union8_kernel (matcher variant 6) rests on, checked without a GPU:

* the rows' 8-bit planes bound the SAD from below for ANY int16 descriptors: 8 * SAD8 - 896 <= SAD (libviso_amd/csrc/common.h
  VISO_ROW8_SLACK, match_dev.h row8_of), also with the rescue's coarser byte (SAD8 in units of 128);
* the kernel's verdict — score the two best candidates by SAD8 exactly, let the third key's bound decide, otherwise score
  the candidates whose bound fails the same test and decide again (once more if a reject turned into an accept) —
  reproduces match_desc (src/viso.cpp:703-716) as the oracle computes it, on data that sit on the bound's edges.

This is a numpy transcription of the kernel's decision logic (csrc/match_union8.hip, "verdict" and "rescue"), not of its
lane arithmetic: the GPU tests (tests/test_gpu_union8.py) compare the kernel itself with the oracle."""
import numpy as np
import pytest

from libviso_amd.abi import MatchParams

SLACK = 7 * 128


def plane(v, s=3):
    return np.clip((v.astype(np.int64) + (128 << s)) >> s, 0, 255)


def test_plane_bound_holds_for_any_int16_rows():
    rng = np.random.default_rng(61)
    for scale in (3, 40, 300, 1020, 5000, 32767):
        a = rng.integers(-scale, scale + 1, (400, 121))
        b = rng.integers(-scale, scale + 1, (400, 121))
        b[:100] = a[:100] + rng.integers(-9, 10, (100, 121))          # near-equal rows: the bound is tightest there
        a[200:230] = rng.choice([-32768, 32767, -1025, -1024, 1023, 1024], (30, 121))
        a, b = np.clip(a, -32768, 32767), np.clip(b, -32768, 32767)
        sad = np.abs(a - b).sum(1)
        for s in range(4):                                              # every shift the kernel may run with
            sad8 = np.abs(plane(a, s) - plane(b, s)).sum(1)
            slack = ((1 << s) - 1) * 128
            assert np.all((sad8 << s) - slack <= sad)
            assert np.all(((sad8 >> 7) << (7 + s)) - slack <= sad)      # the byte the rescue reads: floor(SAD8 / 128)
    # per element: 2^s |h(a) - h(b)| - (2^s - 1) <= |a - b| over the whole int16 range of one operand
    a = np.arange(-32768, 32768)
    for s in range(4):
        for bv in (-32768, -1500, -1024, -129, -128, -7, 0, 5, 127, 128, 1016, 1023, 1024, 32767):
            assert np.all((np.abs(plane(a, s) - plane(np.full_like(a, bv), s)) << s) - ((1 << s) - 1) <= np.abs(a - bv))


def _emulate(kp1, kp2, d1, d2, mp):
    """match_desc through the kernel's decision logic; returns {query: (target, sad)} of the accepted queries and how many
    queries needed the rescue / were left to the overflow kernel (exact ties)."""
    a, b = d1.astype(np.int64), d2.astype(np.int64)
    ha, hb = plane(a), plane(b)
    second, ratio = bool(mp.enforce_2nd_best), float(mp.ratio_2nd_best)
    out, n_rescue, n_tie = {}, 0, 0
    for i in range(len(kp1)):
        dist = np.abs(kp1[i, 0] - kp2[:, 0]) + np.abs(kp1[i, 1] - kp2[:, 1])
        mem = dist <= mp.radius
        if dist[0] <= mp.radius:                       # Q1 (src/viso.cpp:693): target 0 in radius cuts the list
            mem &= dist < dist[0]
        idx = np.nonzero(mem)[0]
        if len(idx) == 0:
            continue
        sad = np.abs(a[i][None] - b[idx]).sum(1)
        s8 = np.abs(ha[i][None] - hb[idx]).sum(1)
        order = np.argsort(s8 * 512 + np.arange(len(idx)), kind="stable")
        scored = {int(order[0]): int(sad[order[0]])}
        if len(idx) > 1:
            scored[int(order[1])] = int(sad[order[1]])

        def verdict():
            v = sorted(scored.values())
            if len(v) == 1:
                return v[0], None, (not second) or float(v[0]) < 1.7976931348623157e308 * ratio
            return v[0], v[1], (not second) or float(v[0]) < float(v[1]) * ratio

        e1, e2, acc = verdict()
        if e2 is not None and e1 == e2:                 # exact tie of the two: overflow kernel (not emulated)
            n_tie += 1
            out[i] = None
            continue
        if len(idx) > 2:
            L3 = 8 * int(s8[order[2]]) - SLACK

            def clear(L, e1, e2, acc):
                if not second:
                    return L > e1
                return (L > e1 and float(e1) < float(L) * ratio) if acc else (float(L) >= float(e2) * ratio)

            if not clear(L3, e1, e2, acc):
                n_rescue += 1
                for _ in range(2):                      # the second time only after a reject turned into an accept
                    was = acc
                    for j in range(len(idx)):
                        Lq = ((int(s8[j]) >> 7) << 10) - SLACK
                        if j not in scored and not clear(Lq, e1, e2, acc):
                            scored[j] = int(sad[j])
                    e1, e2, acc = verdict()
                    if not (acc and not was):
                        break
                if e1 == e2:
                    n_tie += 1
                    out[i] = None
                    continue
        if acc:
            win = min(scored, key=lambda j: scored[j])
            out[i] = (int(idx[win]), e1)
    return out, n_rescue, n_tie


def _spread(rng, total, dlen=121):
    v = np.zeros(dlen, np.int64)
    left = int(total)
    while left > 0:
        s = min(left, int(rng.integers(1, 41)))
        v[rng.integers(0, dlen)] += s * (1 if rng.random() < 0.5 else -1)
        left -= s
    return v


@pytest.mark.parametrize("vrange", [1020, 40])
def test_verdict_logic_equals_match_desc(oracle, vrange):
    rng = np.random.default_rng(9100 + vrange)
    gaps = np.array([0, 1, 7, 60, 300, 700, 890, 896, 897, 1000, 1790, 1800, 2500, 6000])
    tot_rescue = tot_acc = 0
    for it in range(12):
        n1, n2 = int(rng.integers(8, 50)), int(rng.integers(4, 160))
        kp1 = rng.integers(0, 30, (n1, 2)).astype(np.float32)
        kp2 = rng.integers(0, 30, (n2, 2)).astype(np.float32)
        d1 = rng.integers(-vrange, vrange + 1, (n1, 121)).astype(np.int64)
        d2 = rng.integers(-vrange, vrange + 1, (n2, 121)).astype(np.int64)
        free = list(rng.permutation(np.arange(1, n2)))
        for i in range(n1):
            k = int(rng.integers(0, 6))
            if len(free) < k:
                break
            d = int(rng.integers(0, 5000))
            for _ in range(k):
                d2[free.pop()] = d1[i] + _spread(rng, d)
                d += int(rng.choice(gaps))
        mp = MatchParams.temporal()
        mp.enforce_2nd_best = int(it % 4 != 3)
        mp.ratio_2nd_best = [0.9, 0.8, 1.0, 0.5][it % 4]
        mp.radius = 200.0
        f1, f2 = d1.astype(np.float32), d2.astype(np.float32)
        want = {int(r[0]): (int(r[1]), int(r[2])) for r in oracle.match_desc(kp1, kp2, f1, f2, mp)}
        got, n_rescue, _ = _emulate(kp1, kp2, d1, d2, mp)
        for i in range(n1):
            if i in got and got[i] is None:             # exact tie: the overflow kernel's business
                continue
            assert got.get(i) == want.get(i), (vrange, it, i, got.get(i), want.get(i))
        tot_rescue += n_rescue
        tot_acc += len(want)
    assert tot_acc > 50 and tot_rescue > 10             # both the settled and the rescued verdicts were exercised
