import os, sys, time, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # the tree this file lives in, not a fixed path
import libviso_amd
from libviso_amd import synth, drop_in
from libviso_amd.abi import MatchParams
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 101
kp = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
import os
if os.environ.get("GN_SPLIT"): libviso_amd.set_gn_split(int(os.environ["GN_SPLIT"]))
seq = synth.make_sequence(1000, nf, n_kp=kp)
if os.environ.get("DIRECT") == "1": drop_in.plain_speculate(False)   # every call does its own work
# warm
drop_in.run(seq["kp"][:4], seq["desc"][:4], seq["n"][:4], seq["F"], seq["param"], seed=1)
if os.environ.get("VISO_PLAIN_TRACE") == "1":   # the first calls create streams and allocate: their own lines, not in the loop's averages
    print("(warm-up calls:)", file=sys.stderr); libviso_amd.load().viso_plain_trace_dump(); print("(the loop:)", file=sys.stderr)
o = drop_in.run(seq["kp"], seq["desc"], seq["n"], seq["F"], seq["param"], seed=1)
import ctypes
try: libviso_amd.load().viso_plain_trace_dump()
except Exception as e: print(e)
print("libraries loaded:", sorted({l.split()[-1] for l in open("/proc/self/maps") if "libviso_" in l}))   # an A/B that swaps .so files shows WHICH one it timed
print("frames", o["frames"], "loop_s", o["loop_s"], "fps", (o["frames"]-1)/o["loop_s"], "carry_s", o["carry_s"])
for k,(c,us) in o["calls"].items(): print(f"  {k:28s} calls {c:5d}  {us/c:9.1f} us/call  {us/(o['frames']-1):9.1f} us/frame")
st=(ctypes.c_int64*8)(); libviso_amd.load().viso_plain_speculate_stats(st); print("speculation served (temporal, collect, tri/circle, ransac):", list(st)[:4], "wasted:", list(st)[4:])
drop_in.plain_profile(True)
o2 = drop_in.run(seq["kp"], seq["desc"], seq["n"], seq["F"], seq["param"], seed=1)
drop_in.plain_profile(False)
print("profiled fps", (o2["frames"]-1)/o2["loop_s"])
for k,v in drop_in.plain_profile_rows().items():
    c=v["calls"]; print(f"  {k:28s} calls {c:5d} host {v['host_us']/c:8.1f} h2d {v['h2d_us']/c:8.1f} kern {v['kernel_us']/c:8.1f} d2h {v['d2h_us']/c:8.1f} wait {v['wait_us']/c:8.1f}")
# compare with batch
ctx = libviso_amd.Context(0); b = libviso_amd.Batch(ctx, nf, kp)
st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
b.upload(seq["kp"], seq["desc"], seq["n"]); b.set_params(st, tm, seq["param"], seed=1); b.run()
tr, ok, n_inl = b.poses()
print("ok equal", np.array_equal(ok, o["ok"]), "n_inl equal", np.array_equal(n_inl[ok==1], o["n_inl"][ok==1]), "tr maxdiff", np.abs(tr[ok==1]-o["tr"][ok==1]).max(), "ok sum", ok.sum())
print("n_inl where not ok: batch", n_inl[ok==0][:8], "dropin", o["n_inl"][ok==0][:8])
