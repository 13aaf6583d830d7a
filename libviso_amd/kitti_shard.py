"""One KITTI sequence sharded over the GPUs of a node (BASELINE configs[3]).

    KITTI_HOME=... python -m libviso_amd.kitti_shard result_sha seq_name [begin [end]] --gpus W

is the reference's `kitti` driver (src/kitti.cpp:79-118: calib.txt + image_0/%06d.png + image_1/%06d.png in,
results/<seq>/<sha>/data/<seq>.txt out) with the frame loop of sequence_odometry (src/viso.cpp:1205-1327) cut into
W contiguous ranges with a one-frame halo, one rank per GPU:

  * every rank runs its range through the C++ host mirror (libviso_host.so: image decoding, chunked device batches;
    `viso_kitti_run_range`, libviso_amd/host/kitti_shard.hpp) on its own device; RANSAC streams are keyed on the
    absolute frame index, so a frame's record is the same in every partition;
  * ONE collective: an all-gather of the fixed-size per-pair records {tr[6], ok, n_inl} (64 B per frame pair, about
    0.3 MB for KITTI 00) -- RCCL over xGMI (`--backend nccl`), gloo for rehearsals on CPU tensors;
  * rank 0 chains the records (pose <- pose * inv(tr2mat(tr)), src/viso.cpp:1315-1321) and writes the pose file
    through the same C++ `savePoses` as the one-process driver (`viso_kitti`): any W gives the byte-identical file.

With --gpus W > 1 and no WORLD_SIZE in the environment this module starts its own ranks
(`python -m torch.distributed.run --nproc-per-node W`, as a child process, before anything touches the GPU).
`--force-collective` builds the process group and runs the all-gather even at W = 1 (the one-GPU box's way through
RCCL: init_process_group("nccl", device_id=...), a float64 CUDA tensor through all_gather, an all_reduce(MAX) of the
wall time, destroy_process_group).

Failure handling (no collective is ever entered on one side only): a rank whose range fails still contributes its
block to the ONE all-gather, with a status word set; every rank sees it and all of them exit non-zero after the
collective.  A rank that fails before it knows the block shape (no calib, no frames) exits at once without joining
anything; torchrun then tears the others down.
This file is host plumbing: no arithmetic of the hot path happens in Python.
"""
import argparse
import ctypes as C
import os
import socket
import subprocess
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_SO = os.path.join(_HERE, "libviso_host.so")
REC = 8   # doubles per record: tr[6], ok, n_inl


def partition(n_frames, world):
    """Same rule as viso::partition (host/kitti_shard.cpp) and shard.partition."""
    from .shard import partition as p
    return p(n_frames, world)


def load_host():
    """ctypes handle of libviso_host.so (C++ host mirror; links libviso_hip.so).  No fallback."""
    if not os.path.exists(HOST_SO):
        raise RuntimeError(f"{HOST_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        import torch  # noqa: F401  one HIP runtime per process: torch's first (see libviso_amd.load)
    except Exception:  # pragma: no cover
        pass
    L = C.CDLL(HOST_SO)
    L.viso_kitti_count_frames.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.viso_kitti_run_range.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                       C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.viso_kitti_write_poses.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    L.viso_kitti_write_poses2.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.viso_kitti_last_stats.argtypes = [C.POINTER(C.c_double)]
    L.viso_kitti_last_stats.restype = None
    L.viso_kitti_set_decode_threads.argtypes = [C.c_int]
    L.viso_kitti_set_decode_threads.restype = None
    L.viso_host_last_error.restype = C.c_char_p
    return L


def cpu_budget():
    """Host threads this process may really use: hardware threads cut down to the affinity mask and the cgroup's CPU quota
    (the C++ runner's viso::cpu_budget, host/viso_host.cpp)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


STAT_NAMES = ("frames", "decode_threads", "wall_s", "decode_wait_s", "decode_cpu_s", "issue_s", "drain_wait_s", "upload_ms", "gpu_ms")


def last_stats(L):
    """OdometryStats of this thread's last viso_kitti_run_range (host/viso.hpp): where the range's time went."""
    v = (C.c_double * 9)()
    L.viso_kitti_last_stats(v)
    return dict(zip(STAT_NAMES, [float(x) for x in v]))


def hip_engine(L, device, chunk=64, seed=0):
    """engine(seq_base, begin, first, last) -> rec [(n_done), 8]: the range on the HIP pipeline."""
    def run(seq_base, begin, first, last):
        rec = np.zeros((max(last - first, 1), REC), np.float64)
        n_done = C.c_int(0)
        r = L.viso_kitti_run_range(seq_base.encode(), begin, first, last, device, chunk, seed,
                                   rec.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n_done))
        if r != 1:
            raise RuntimeError(f"viso_kitti_run_range failed with {r}: {L.viso_host_last_error().decode()}")
        return rec[:n_done.value]
    return run


def count_frames(L, seq_base, begin, end):
    n = L.viso_kitti_count_frames(seq_base.encode(), begin, end)
    if n < 0:
        raise RuntimeError(L.viso_host_last_error().decode())
    return n


def gather_records(rec, n_pairs, first, rank, world, dist=None, device="cpu", failed=False):
    """The one exchange step.  Every rank contributes a fixed-size block [n_pairs + 1, 8]: its records at the rows of
    its pairs, and in the last row how many it solved (a rank comes back short when an image of its range cannot be
    decoded) and a status word (1 = this rank's range failed: the block carries no records).  Returns the blocks of
    all ranks, rank order.  With dist=None and world == 1 nothing is exchanged."""
    block = np.zeros((n_pairs + 1, REC), np.float64)
    if not failed:
        block[first:first + len(rec)] = rec
        block[n_pairs, 0] = len(rec)
    block[n_pairs, 1] = 1.0 if failed else 0.0
    if dist is None:
        if world != 1:
            raise RuntimeError("gather_records: world > 1 needs a process group")
        return [block]
    import torch
    t = torch.from_numpy(block).to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)                          # fixed-size records, one collective
    return [o.cpu().numpy() for o in out]


class RankFailed(RuntimeError):
    """Raised on EVERY rank, after the all-gather, when some rank's range failed."""


def stitch(blocks, n_frames):
    """Blocks of all ranks -> the records of the sequence in frame order, cut where a rank came back short
    (viso::stitch_records)."""
    world = len(blocks)
    n_pairs = max(0, n_frames - 1)
    rows = []
    for r, (a, b) in enumerate(partition(n_frames, world)):
        done = int(blocks[r][n_pairs, 0])
        rows.append(blocks[r][a:a + done])
        if done < b - a:
            break
    return np.concatenate(rows, 0) if rows else np.zeros((0, REC))


def write_poses(L, file_name, rec, reference_pose_list=False):
    """reference_pose_list: the list the reference's code actually writes, [P1, ..., Pn, Pn] (src/viso.cpp:1317-1321,
    see host/kitti_shard.hpp), instead of [I, P1, ..., Pn]."""
    rec = np.ascontiguousarray(rec, np.float64)
    n_poses = C.c_int(0)
    r = L.viso_kitti_write_poses2(file_name.encode(), rec.ctypes.data_as(C.POINTER(C.c_double)), len(rec),
                                  1 if reference_pose_list else 0, C.byref(n_poses))
    if r != 1:
        raise RuntimeError(f"viso_kitti_write_poses failed with {r}: {L.viso_host_last_error().decode()}")
    return n_poses.value


def run_rank(home, result_sha, seq_name, begin, end, rank, world, L, engine, dist=None, coll_device="cpu",
             reference_pose_list=False):
    """One rank's whole job; returns (n_frames, records of the sequence, pose file or None).  `engine` is what
    turns a frame range into records (hip_engine here; the CPU tests inject the oracle).  An engine that raises does
    not keep this rank out of the collective: its block carries the failure, and every rank raises RankFailed behind
    the all-gather."""
    seq_base = os.path.join(home, "sequences", seq_name)
    n_frames = count_frames(L, seq_base, begin, end)      # may raise: BEFORE any collective, see main()
    n_pairs = max(0, n_frames - 1)
    first, last = partition(n_frames, world)[rank]
    rec, err = np.zeros((0, REC)), None
    try:
        if last > first:
            rec = engine(seq_base, begin, first, last)
    except Exception as e:                                # noqa: BLE001  reported through the collective
        err = e
        print(f"kitti_shard: rank {rank} failed on frames {begin + first}..{begin + last}: {e}", file=sys.stderr, flush=True)
    blocks = gather_records(rec, n_pairs, first, rank, world, dist, coll_device, failed=err is not None)
    bad = [r for r, b in enumerate(blocks) if b[n_pairs, 1] != 0]
    if bad:
        raise RankFailed(f"rank(s) {bad} failed; no pose file written") from err
    full = stitch(blocks, n_frames)
    out = None
    if rank == 0:
        out = os.path.join(home, "results", seq_name, result_sha, "data", seq_name + ".txt")   # src/kitti.cpp:100,112-114
        write_poses(L, out, full, reference_pose_list)
    return n_frames, full, out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m libviso_amd.kitti_shard")
    ap.add_argument("result_sha")
    ap.add_argument("seq_name")
    ap.add_argument("begin", nargs="?", type=int, default=0)            # src/kitti.cpp:86-94
    ap.add_argument("end", nargs="?", type=int, default=2**31 - 1)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--backend", default="nccl", help="nccl = RCCL; gloo for rehearsals")
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--same-device", action="store_true", help="every rank on device 0 (rehearsal on a one-GPU box; use --backend gloo)")
    ap.add_argument("--force-collective", action="store_true",
                    help="build the process group and run the all-gather even with one rank (RCCL on a one-GPU box)")
    ap.add_argument("--decode-threads", type=int, default=0, help="PNG decoding threads per rank (0: min(64, cpus / ranks))")
    ap.add_argument("--reference-pose-list", action="store_true",
                    help="write the list the reference's code actually produces, [P1..Pn, Pn] (src/viso.cpp:1317-1321)")
    args = ap.parse_args(argv)
    home = os.environ.get("KITTI_HOME")
    if not home:
        print("KITTI_HOME is not set", file=sys.stderr)
        return 2
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the ranks as a child process tree; this process has not imported torch and never touches the GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               "-m", "libviso_amd.kitti_shard"] + (list(argv) if argv is not None else sys.argv[1:])
        env = dict(os.environ, PYTHONPATH=os.path.dirname(_HERE) + os.pathsep + os.environ.get("PYTHONPATH", ""))
        return subprocess.call(cmd, env=env)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"kitti_shard: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 7
    device = 0 if args.same_device else local_rank
    dist = None
    coll_device = "cpu"
    t_start = time.perf_counter()
    if world > 1 or args.force_collective:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:               # --force-collective outside torchrun: a group of one
            os.environ["MASTER_PORT"] = str(_free_port())
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
            coll_device = torch.device("cuda", device)
        else:
            dist.init_process_group(args.backend)
    L = load_host()
    threads = args.decode_threads or int(os.environ.get("VISO_DECODE_THREADS", "0")) or max(1, min(64, cpu_budget() // world))
    L.viso_kitti_set_decode_threads(threads)

    def die(code, what):
        # a failure that the peers cannot learn about through the collective: leave WITHOUT joining one (no barrier,
        # no destroy_process_group: the peers may be inside the all-gather); torchrun sees the exit code and stops them
        print(f"kitti_shard: rank {rank}: {what}", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(code)

    try:
        n_frames, full, out = run_rank(home, args.result_sha, args.seq_name, args.begin, args.end, rank, world, L,
                                       hip_engine(L, device, args.chunk, args.seed), dist, coll_device,
                                       args.reference_pose_list)
    except RankFailed as e:                                # every rank is here, behind the same all-gather
        print(f"kitti_shard: rank {rank}: {e}", file=sys.stderr, flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return 8
    except Exception as e:                                 # noqa: BLE001  before / outside the collective
        if dist is not None:
            die(9, f"{type(e).__name__}: {e}")
        raise
    st = last_stats(L)
    wall = time.perf_counter() - t_start
    coll = None
    if dist is not None:
        import torch
        # per-rank stats to rank 0's report: one all_gather of 9 doubles; the job's wall time is the slowest rank's
        t = torch.tensor([st[k] for k in STAT_NAMES], dtype=torch.float64, device=coll_device)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        w = torch.tensor([wall], dtype=torch.float64, device=coll_device)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
        per_rank = [dict(zip(STAT_NAMES, o.cpu().tolist())) for o in outs]
        coll = {"backend": "rccl" if args.backend == "nccl" else args.backend, "ranks": dist.get_world_size(),
                "device": str(coll_device)}
    else:
        per_rank = [st]
    if rank == 0:
        n_all = len(full) + (n_frames > 0)
        print(f"frames {n_all} solved {int(full[:, 6].sum()) if len(full) else 0} ranks {world} "
              f"backend {coll['backend'] if coll else 'none'} -> {out}", flush=True)
        for r, s in enumerate(per_rank):
            print(f"rank {r}: {int(s['frames'])} frames in {s['wall_s']:.3f} s | decode: {int(s['decode_threads'])} threads, "
                  f"{s['decode_cpu_s']:.3f} s of thread time, runner waited {s['decode_wait_s']:.3f} s | GPU stamps: upload "
                  f"{s['upload_ms'] * 1e-3:.3f} s, kernels {s['gpu_ms'] * 1e-3:.3f} s | host: issue {s['issue_s']:.3f} s, waiting "
                  f"for results {s['drain_wait_s']:.3f} s", flush=True)
        print(f"runner: {n_all} frames in {wall:.3f} s = {n_all / wall:.0f} frames/s (process group + decode + GPU + gather + pose file"
              f"{'; collective ' + str(coll) if coll else ''})", flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
