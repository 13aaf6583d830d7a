"""Robustness of the batch family on the device: process teardown with live handles, the per-image
general-path flag, asynchronous (pinned, stream-ordered) uploads, uploads while a run is in flight."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exit_with_live_handles_keeps_the_python_exit_code():
    """A Batch and a Context leaked by an exception: the process must end with the traceback's exit code (1), not
    with an abort from inside the HIP runtime (round 1: exit 134, std::bad_variant_access)."""
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import libviso_amd
        from libviso_amd import synth
        from libviso_amd.abi import MatchParams
        seq = synth.make_sequence(1, 3, n_kp=300, width=400, height=200)
        ctx = libviso_amd.Context(0)
        b = libviso_amd.Batch(ctx, 3, 300)
        b.upload(seq["kp"], seq["desc"], seq["n"])
        b.set_params(MatchParams.stereo(seq["F"]), MatchParams.temporal(), seq["param"], seed=1)
        b.run()
        keep = [b, ctx]                      # still alive at interpreter exit
        raise RuntimeError("leak on purpose")
    """ % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stderr[-2000:])
    assert "leak on purpose" in r.stderr and "terminate called" not in r.stderr


@pytest.mark.parametrize("order", ["alive_at_exit", "stream_first", "context_first"])
def test_context_on_a_callers_stream(order):
    """viso_ctx_create(device, stream) BORROWS the caller's stream.  A context (and a batch) on such a stream must survive
    every teardown order a caller can produce: everything still alive at interpreter exit; the stream destroyed before
    the context (viso_ctx_destroy then REPORTS the dead handle, it does not crash); the context first.  Round 4 left a
    probe that exited with a live context on a CU-masked stream of its own and died in __cxa_finalize under rocprofv3
    (gpurun_out/cum.txt): the probe is gone, the teardown orders are pinned here."""
    code = textwrap.dedent("""
        import ctypes as C, sys
        sys.path.insert(0, %r)
        import torch
        import libviso_amd
        from libviso_amd import synth
        from libviso_amd.abi import MatchParams
        order = %r
        import libviso_amd as _l; _l.load()
        path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
        hip = C.CDLL(path)                  # the HIP runtime this process already holds (torch's): the same handle, no second copy
        stream = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(stream)) == 0
        seq = synth.make_sequence(1, 3, n_kp=300, width=400, height=200)
        ctx = libviso_amd.Context(0, stream=stream)
        b = libviso_amd.Batch(ctx, 3, 300)
        b.upload(seq["kp"], seq["desc"], seq["n"])
        b.set_params(MatchParams.stereo(seq["F"]), MatchParams.temporal(), seq["param"], seed=1)
        b.run()
        tr, ok, n_inl = b.poses()
        assert ok[1:].all()
        L = libviso_amd.load()
        if order == "stream_first":
            b.close()
            assert hip.hipStreamDestroy(stream) == 0
            r = L.viso_ctx_destroy(C.c_void_p(ctx.h))     # a dead stream: reported (VISO_ERR_HIP) or tolerated, never a crash
            assert r in (1, -2), r
            ctx.h = None
        elif order == "context_first":
            b.close(); ctx.close()
            assert hip.hipStreamDestroy(stream) == 0
        else:
            keep = [b, ctx, stream]          # all alive at interpreter exit
        print("done", order)
    """ % (ROOT, order))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and ("done " + order) in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "terminate called" not in r.stderr and "Segmentation" not in r.stderr


def test_destroy_reports_instead_of_swallowing(viso):
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, 2, 64)
    b.close(); b.close()                     # idempotent
    ctx.close(); ctx.close()
    assert viso.viso_batch_destroy(None) == 1 and viso.viso_ctx_destroy(None) == 1


def _per_call(oracle, seq, which, t, st, tm):
    n = seq["n"]
    q = (0, t) if which < 2 else (1, t)
    tg = (1, t) if which == 0 else ((0, t - 1) if which == 1 else (1, t - 1))
    nq, nt = n[q[1], q[0]], n[tg[1], tg[0]]
    return oracle.match_desc(seq["kp"][q[1], q[0], :nq], seq["kp"][tg[1], tg[0], :nt],
                             seq["desc"][q[1], q[0], :nq], seq["desc"][tg[1], tg[0], :nt],
                             st if which == 0 else tm, return_scored=True)


def test_one_fractional_descriptor_flags_one_image_only(viso, oracle):
    """16 frames, ONE non-integer descriptor value in image (t=7, right): only that image is flagged, so only the
    match_desc calls that read it (stereo 7, temporal-right 7 and 8) take the general kernel; everything stays
    bit-exact and the SAD counters equal the oracle's."""
    seq = synth.make_sequence(41, 16, n_kp=500, width=640, height=240)
    seq["desc"][7, 1, 123, 17] += 0.5
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, 16, 500)
    b.upload(seq["kp"], seq["desc"], seq["n"])
    b.set_params(st, tm, seq["param"], seed=2)
    b.run()
    flags = b.general_path_flags()
    want_flags = np.zeros((16, 2), np.int32); want_flags[7, 1] = 1
    assert np.array_equal(flags, want_flags)
    sc, _ = b.counters()
    for t in range(16):
        for which in range(3 if t else 1):
            want, wsc = _per_call(oracle, seq, which, t, st, tm)
            assert np.array_equal(b.matches(which, t), want) and sc[which, t] == wsc, (which, t)
    want = oracle.sequence(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], seed=2)
    tr, ok, n_inl = b.poses()
    assert np.array_equal(ok, want["ok"]) and np.array_equal(n_inl, want["n_inl"])
    # the flag is per run: integer data again -> no image flagged
    seq["desc"][7, 1, 123, 17] -= 0.5
    b.upload(seq["kp"], seq["desc"], seq["n"])
    b.run()
    assert not b.general_path_flags().any()
    b.close(); ctx.close()


def test_async_upload_streams_fresh_frames(viso):
    """The streaming mode: two different sequences alternate through ONE batch by pinned asynchronous uploads,
    three lanes deep, nothing synchronised between enqueue and run; each result equals the synchronous path's."""
    nf, kp = 9, 400
    seqs = [synth.make_sequence(500 + i, nf, n_kp=kp, width=640, height=240) for i in range(2)]
    st, tm = MatchParams.stereo(seqs[0]["F"]), MatchParams.temporal()
    want = []
    for s in seqs:
        ctx = libviso_amd.Context(0)
        b = libviso_amd.Batch(ctx, nf, kp)
        b.upload(s["kp"], s["desc"], s["n"])
        b.set_params(st, tm, s["param"], seed=3)
        b.run()
        want.append(b.poses() + (b.matches(1, 4),))
        b.close(); ctx.close()
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, nf, kp)
    b.set_params(st, tm, seqs[0]["param"], seed=3)
    pins = []
    for s in seqs:
        pk = libviso_amd.PinnedArray(s["kp"].shape, np.float32); pk.a[...] = s["kp"]
        pd = libviso_amd.PinnedArray(s["desc"].shape, np.float32); pd.a[...] = s["desc"]
        pins.append((pk, pd))
    for step in range(6):
        i = step % 2
        b.upload_async(pins[i][0].a, pins[i][1].a, seqs[i]["n"])     # enqueued behind the previous run
        b.run()
        if step >= 4:                                                 # the last two steps: check both sequences
            tr, ok, ninl = b.poses()
            assert np.array_equal(tr, want[i][0]) and np.array_equal(ok, want[i][1]) and np.array_equal(ninl, want[i][2])
            assert np.array_equal(b.matches(1, 4), want[i][3])
    b.close(); ctx.close()
    for pk, pd in pins:
        pk.close(); pd.close()


def test_synchronous_upload_waits_for_a_run_in_flight(viso):
    """viso_batch_upload / set_params called right behind an asynchronous run must not corrupt that run."""
    nf, kp = 33, 1000
    a = synth.make_sequence(600, nf, n_kp=kp)
    c = synth.make_sequence(601, nf, n_kp=kp)
    st, tm = MatchParams.stereo(a["F"]), MatchParams.temporal()
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, nf, kp)
    b.upload(a["kp"], a["desc"], a["n"])
    b.set_params(st, tm, a["param"], seed=4)
    b.run()
    ref = b.poses()
    for _ in range(3):
        b.run()                                  # asynchronous
        b.upload(c["kp"], c["desc"], c["n"])     # must wait for it
        b.set_params(st, tm, c["param"], seed=4)
        b.upload(a["kp"], a["desc"], a["n"])
        b.set_params(st, tm, a["param"], seed=4)
        b.run()
        got = b.poses()
        assert all(np.array_equal(x, y) for x, y in zip(got, ref))
    b.close(); ctx.close()


def test_a_context_closed_first_takes_its_batches_along(viso):
    """The order an exception or the garbage collector can produce: the context goes first.  Its batches are closed with it
    (viso_batch_destroy on a destroyed context aborted the process from inside the HIP runtime)."""
    ctx = libviso_amd.Context(0)
    b1, b2 = libviso_amd.Batch(ctx, 3, 64), libviso_amd.Batch(ctx, 2, 32)
    ctx.close()
    assert b1.h is None and b2.h is None
    b1.close(); b2.close()      # no-ops
    del b1, b2, ctx


def test_raw_handles_destroyed_in_the_wrong_order_get_return_codes(viso, oracle):
    """The C-ABI itself (no Context / Batch wrapper): a batch that outlives its context.  include/viso_hip.h promises return
    codes, never an abort: viso_ctx_destroy takes the context's live batches along, the caller's late viso_batch_destroy is
    a no-op, every other call on a dead handle is VISO_ERR_ARG (-1).  Run once, in this process."""
    import ctypes as C
    L = viso
    seq = synth.make_sequence(2, 3, n_kp=200, width=400, height=200)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    c = L.viso_ctx_create(0, None)
    assert c
    b1, b2 = L.viso_batch_create(c, 3, 200, 121), L.viso_batch_create(c, 2, 32, 121)
    assert b1 and b2
    from libviso_amd.abi import ptr
    kp, desc, n = (np.ascontiguousarray(seq[k]) for k in ("kp", "desc", "n"))
    assert L.viso_batch_upload(b1, 0, 3, ptr(kp, C.c_float), ptr(desc, C.c_float), ptr(n, C.c_int32)) == 1
    assert L.viso_batch_set_params(b1, C.byref(st), C.byref(tm), C.byref(seq["param"]), 1, 0) == 1
    assert L.viso_batch_run(b1) == 1                       # work in flight when the context goes
    assert L.viso_batch_destroy(b2) == 1
    assert L.viso_batch_destroy(b2) == -1                  # twice
    assert L.viso_ctx_destroy(c) == 1                      # b1 still alive: freed with its context
    tr, ok, ni = np.zeros((3, 6)), np.zeros(3, np.int32), np.zeros(3, np.int32)
    assert L.viso_batch_get_poses(b1, ptr(tr, C.c_double), ptr(ok, C.c_int32), ptr(ni, C.c_int32)) == -1
    assert L.viso_batch_run(b1) == -1
    assert L.viso_batch_destroy(b1) == 1                   # the late destroy: a no-op ...
    assert L.viso_batch_destroy(b1) == -1                  # ... once
    assert L.viso_ctx_destroy(c) == -1
    assert L.viso_ctx_synchronize(c) == -1
    assert not L.viso_batch_create(c, 2, 32, 121)
    # the library is still usable, and still right
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, 3, 200)
    b.upload(seq["kp"], seq["desc"], seq["n"]); b.set_params(st, tm, seq["param"], seed=1); b.run()
    want = oracle.sequence(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], seed=1)
    _, ok2, n2 = b.poses()
    assert np.array_equal(ok2, want["ok"]) and np.array_equal(n2, want["n_inl"])
    b.close(); ctx.close()


def test_handles_from_several_threads_in_any_order(viso):
    """The handle registry under concurrency: four threads create contexts and batches, run a matcher step, and destroy
    every handle exactly ONCE in a random order -- the context often before its batches.  Every destroy returns VISO_OK
    (a batch its context took along: the no-op), a getter on a batch whose context is gone returns VISO_ERR_ARG, nothing
    crashes.  (Destroying a handle TWICE is an error the registry catches only until the allocator hands the address out
    again -- to another thread here: that case is the single-threaded test above.)"""
    import ctypes as C
    import threading
    from libviso_amd.abi import ptr
    L = viso
    errors = []
    seq = synth.make_sequence(3, 3, n_kp=150, width=300, height=150)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    kp, desc, n = (np.ascontiguousarray(seq[k]) for k in ("kp", "desc", "n"))

    def worker(seed):
        rng = np.random.default_rng(seed)
        try:
            for _ in range(6):
                c = L.viso_ctx_create(0, None)
                assert c
                bs = [L.viso_batch_create(c, 3, 150, 121) for _ in range(int(rng.integers(1, 4)))]
                assert all(bs)
                assert L.viso_batch_upload(bs[0], 0, 3, ptr(kp, C.c_float), ptr(desc, C.c_float), ptr(n, C.c_int32)) == 1
                assert L.viso_batch_set_params(bs[0], C.byref(st), C.byref(tm), C.byref(seq["param"]), 1, 0) == 1
                assert L.viso_batch_run(bs[0]) == 1            # in flight while handles go
                order = list(bs) + [("ctx", c)]
                rng.shuffle(order)
                ctx_dead = False
                for h in order:
                    if isinstance(h, tuple):
                        assert L.viso_ctx_destroy(h[1]) == 1
                        ctx_dead = True
                    else:
                        if ctx_dead:
                            cnt = C.c_int32(0)
                            assert L.viso_batch_get_overflow_count(h, C.byref(cnt)) == -1
                        assert L.viso_batch_destroy(h) == 1
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, 2, 64)
    b.close(); ctx.close()


def test_hypotheses_getter_with_a_wider_capacity(viso):
    """viso_batch_get_hypotheses2 with arrays of [n_frames][capacity > ransac_iter]: every frame's row at the caller's stride
    (the first version forwarded to the tight getter and put frame 1's row where frame 0's padding belonged)."""
    import ctypes as C
    from libviso_amd.abi import f64p, i32p, ptr
    seq = synth.make_sequence(4, 4, n_kp=300, width=400, height=200)
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, 4, 300)
    b.upload(seq["kp"], seq["desc"], seq["n"])
    b.set_params(MatchParams.stereo(seq["F"]), MatchParams.temporal(), seq["param"], seed=3)
    b.run()
    tr_h, ok_h, cnt_h, nu = b.hypotheses()
    iters, cap = tr_h.shape[1], tr_h.shape[1] + 14
    viso.viso_batch_get_hypotheses2.argtypes = [C.c_void_p, C.c_int, f64p, i32p, i32p, i32p]
    tr2 = np.full((4, cap, 6), -7.0); ok2 = np.full((4, cap), -7, np.int32); cnt2 = np.full((4, cap), -7, np.int32)
    nu2 = np.zeros(1, np.int32)
    assert viso.viso_batch_get_hypotheses2(b.h, cap, ptr(tr2, C.c_double), ptr(ok2, C.c_int32), ptr(cnt2, C.c_int32), ptr(nu2, C.c_int32)) == 1
    assert np.array_equal(tr2[:, :iters], tr_h) and np.array_equal(ok2[:, :iters], ok_h) and np.array_equal(cnt2[:, :iters], cnt_h)
    assert (tr2[:, iters:] == -7).all() and (ok2[:, iters:] == -7).all() and (cnt2[:, iters:] == -7).all() and nu2[0] == nu
    assert viso.viso_batch_get_hypotheses2(b.h, iters - 1, ptr(tr2, C.c_double), ptr(ok2, C.c_int32), ptr(cnt2, C.c_int32), None) == -1
    b.close(); ctx.close()
