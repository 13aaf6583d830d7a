#!/usr/bin/env python3
"""Full-length rehearsal of BASELINE configs[3] on the one-GPU box.

Real KITTI data does not exist in this environment, so this builds a synthetic KITTI tree of the size of sequence 00
(4541 stereo frames, 1241 x 376, 8-bit grayscale PNG; reference layout src/kitti.cpp:96-110) under a scratch directory
and runs it through every runner:

    viso_kitti                                  one process, one GPU              (the reference's flow, src/kitti.cpp:79-118)
    viso_kitti --gpus W --same-device           W forked ranks on the one device  (W = 6: the box allows 6 GPU processes)
    python -m libviso_amd.kitti_shard --gpus 1 --backend nccl --force-collective   the torch.distributed runner through RCCL

and checks that the pose files are byte-identical -- and that they are RIGHT: every 50th frame pair and every pair the
runner reports as unsolved goes through the CPU oracle from the PNG files (tests/rehearsal_check.py), the unsolved ones
tabulated by the exit of the reference's loop body they take.  Every rank prints where its wall time went (PNG decode on the worker
threads / upload / GPU); the report goes to stdout and to --out.

The frames come from independent synthetic blocks of 71 frames (generated in parallel): consecutive frames inside a
block are a real camera motion, the seams between blocks are not (those pairs fail or solve to nonsense — irrelevant
for what is measured here: decode, upload and GPU seconds, and that every partition writes the same file).  This script
never touches the GPU itself.
"""
import argparse
import multiprocessing as mp
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
BLOCK = 71


def _block(job):
    base, b, first, count, width, height = job
    import pngutil
    from libviso_amd import synth
    seq = synth.make_image_sequence(9000 + b, count, n_kp=1200, width=width, height=height)
    size = 0
    for t in range(count):
        for side in (0, 1):
            path = os.path.join(base, f"image_{side}", "%06d.png" % (first + t))
            pngutil.write_gray_png(path, seq["images"][t, side])
            size += os.path.getsize(path)
    return size


def build_tree(home, seq_name, n_frames, width, height, procs):
    from libviso_amd import synth
    base = os.path.join(home, "sequences", seq_name)
    for side in (0, 1):
        os.makedirs(os.path.join(base, f"image_{side}"), exist_ok=True)
    with open(os.path.join(base, "calib.txt"), "w") as f:
        for name, P in (("P0", synth.KITTI_P1), ("P1", synth.KITTI_P2)):
            f.write(name + ": " + " ".join("%.12e" % v for v in P.reshape(-1)) + "\n")
    jobs = [(base, b, first, min(BLOCK, n_frames - first), width, height)
            for b, first in enumerate(range(0, n_frames, BLOCK))]
    with mp.get_context("spawn").Pool(procs) as pool:
        return sum(pool.map(_block, jobs))


def run(cmd, env, log):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    dt = time.perf_counter() - t0
    log(f"$ {' '.join(cmd)}\n[exit {r.returncode}, {dt:.2f} s wall including process start]\n{r.stdout}{r.stderr[-3000:] if r.returncode else ''}")
    if r.returncode:
        raise SystemExit(f"{cmd[0]} failed")
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4541, help="KITTI 00 has 4541")
    ap.add_argument("--width", type=int, default=1241)
    ap.add_argument("--height", type=int, default=376)
    ap.add_argument("--ranks", type=int, default=6, help="forked ranks on the one device (the box allows 6 GPU processes)")
    ap.add_argument("--cpus", type=int, default=min(16, len(os.sched_getaffinity(0))), help="host threads to use in total")
    ap.add_argument("--home", default=os.path.join(os.environ.get("TMPDIR", "/tmp"), "viso_kitti_rehearsal"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_kitti_rehearsal.txt"))
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--skip-rccl", action="store_true")
    ap.add_argument("--check-every", type=int, default=50, help="every n-th pair (and every unsolved one) goes through the CPU oracle")
    args = ap.parse_args()
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    exe = os.path.join(ROOT, "libviso_amd", "viso_kitti")
    shutil.rmtree(args.home, ignore_errors=True)
    t0 = time.perf_counter()
    size = build_tree(args.home, "00", args.frames, args.width, args.height, args.cpus)
    log(f"synthetic KITTI tree: {args.frames} stereo frames {args.width}x{args.height}, {2 * args.frames} PNGs, {size / 1e6:.0f} MB "
        f"({size / (2 * args.frames) / 1e3:.0f} kB per image; raw {args.width * args.height / 1e3:.0f} kB), built in {time.perf_counter() - t0:.1f} s "
        f"with {args.cpus} processes")
    env = dict(os.environ, KITTI_HOME=args.home, HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    pose = lambda sha: os.path.join(args.home, "results", "00", sha, "data", "00.txt")   # noqa: E731
    # page cache warm-up of nothing: the files were just written, every run below reads them from memory
    run([exe, "one", "00", "--decode-threads", str(args.cpus)], env, log)
    # the runner's own default (round 6: min(64, hardware threads / ranks), 16 before) and the counts around it: decoding is
    # what bounds the run, the box decides how many threads it honours (nproc says 256, one GPU's share of the host is 16)
    try:
        cg = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        cg = "(no /sys/fs/cgroup/cpu.max)"
    from libviso_amd.kitti_shard import cpu_budget
    log("hardware threads: os.cpu_count() %s, affinity %d, cgroup cpu.max '%s' -> the runners' budget %d" % (os.cpu_count(), len(os.sched_getaffinity(0)), cg, cpu_budget()))
    run([exe, "one_default", "00"], env, log)
    for nt in (32, 64):
        run([exe, f"one_{nt}threads", "00", "--decode-threads", str(nt)], env, log)
    for sha in ("one_default", "one_32threads", "one_64threads"):
        log(f"pose file of {sha} byte-identical to the 16-thread run's: {open(pose('one'), 'rb').read() == open(pose(sha), 'rb').read()}")
    run([exe, "one_1thread", "00", "0", "567", "--decode-threads", "1"], env, log)      # one rank's share of 8, one decode thread
    per_rank = max(1, args.cpus // args.ranks)
    run([exe, f"fork{args.ranks}", "00", "--gpus", str(args.ranks), "--same-device", "--decode-threads", str(per_rank)], env, log)
    same = open(pose("one"), "rb").read() == open(pose(f"fork{args.ranks}"), "rb").read()
    log(f"pose files of 1 process and of {args.ranks} forked ranks byte-identical: {same} ({len(open(pose('one')).read().splitlines())} lines)")
    ok = same
    # the oracle on a sample of the pairs and on every pair the runner reports as unsolved (tests/rehearsal_check.py: the
    # part of the rehearsal that touches oracle/ is test infrastructure and lives under tests/); CPU only
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rehearsal_check.py"), args.home, "00", f"fork{args.ranks}", str(args.ranks),
                        "--every", str(args.check_every), "--procs", str(args.cpus)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    log(f"$ python tests/rehearsal_check.py ... fork{args.ranks} {args.ranks} --every {args.check_every}\n[exit {r.returncode}]\n{r.stdout}{r.stderr[-3000:] if r.returncode else ''}")
    ok = ok and r.returncode == 0
    if not args.skip_rccl:
        run([sys.executable, "-m", "libviso_amd.kitti_shard", "rccl1", "00", "--gpus", "1", "--backend", "nccl", "--force-collective",
             "--decode-threads", str(args.cpus)], env, log)
        same2 = open(pose("one"), "rb").read() == open(pose("rccl1"), "rb").read()
        log(f"pose file of the torch.distributed runner (RCCL process group of one, all_gather on the device) byte-identical: {same2}")
        ok = ok and same2
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    if not args.keep:
        shutil.rmtree(args.home, ignore_errors=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
