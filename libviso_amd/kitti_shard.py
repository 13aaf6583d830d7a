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
This file is host plumbing: no arithmetic of the hot path happens in Python.
"""
import argparse
import ctypes as C
import os
import socket
import subprocess
import sys

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
    L.viso_host_last_error.restype = C.c_char_p
    return L


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


def gather_records(rec, n_pairs, first, rank, world, dist=None, device="cpu"):
    """The one exchange step.  Every rank contributes a fixed-size block [n_pairs + 1, 8]: its records at the rows of
    its pairs, and in the last row how many it solved (a rank comes back short when an image of its range cannot be
    decoded).  Returns the blocks of all ranks, rank order."""
    block = np.zeros((n_pairs + 1, REC), np.float64)
    block[first:first + len(rec)] = rec
    block[n_pairs, 0] = len(rec)
    if world == 1:
        return [block]
    import torch
    t = torch.from_numpy(block).to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)                          # fixed-size records, one collective
    return [o.cpu().numpy() for o in out]


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


def write_poses(L, file_name, rec):
    rec = np.ascontiguousarray(rec, np.float64)
    n_poses = C.c_int(0)
    r = L.viso_kitti_write_poses(file_name.encode(), rec.ctypes.data_as(C.POINTER(C.c_double)), len(rec), C.byref(n_poses))
    if r != 1:
        raise RuntimeError(f"viso_kitti_write_poses failed with {r}: {L.viso_host_last_error().decode()}")
    return n_poses.value


def run_rank(home, result_sha, seq_name, begin, end, rank, world, L, engine, dist=None, coll_device="cpu"):
    """One rank's whole job; returns (n_frames, records of the sequence, pose file or None).  `engine` is what
    turns a frame range into records (hip_engine here; the CPU tests inject the oracle)."""
    seq_base = os.path.join(home, "sequences", seq_name)
    n_frames = count_frames(L, seq_base, begin, end)
    n_pairs = max(0, n_frames - 1)
    first, last = partition(n_frames, world)[rank]
    rec = engine(seq_base, begin, first, last) if last > first else np.zeros((0, REC))
    blocks = gather_records(rec, n_pairs, first, rank, world, dist, coll_device)
    full = stitch(blocks, n_frames)
    out = None
    if rank == 0:
        out = os.path.join(home, "results", seq_name, result_sha, "data", seq_name + ".txt")   # src/kitti.cpp:100,112-114
        write_poses(L, out, full)
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
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
            coll_device = torch.device("cuda", device)
        else:
            dist.init_process_group(args.backend)
    L = load_host()
    try:
        n_frames, full, out = run_rank(home, args.result_sha, args.seq_name, args.begin, args.end, rank, world, L,
                                       hip_engine(L, device, args.chunk, args.seed), dist, coll_device)
        if rank == 0:
            print(f"frames {len(full) + (n_frames > 0)} solved {int(full[:, 6].sum()) if len(full) else 0} ranks {world} "
                  f"backend {args.backend if world > 1 else 'none'} -> {out}", flush=True)
    finally:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
