#!/usr/bin/env python3
"""bench.py — stereo frames/s of the libviso hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic frames that is
already resident in HBM: by default BASELINE.json configs[1] (1241x376,
~2k features/frame, SAD matcher only = pack + 3 match_desc per frame incl. the
final sort).  The same JSON line also carries

  resident_i16 the same matcher step with the descriptors resident as int16 rows
               (the step without the f32 -> u16 + u8 repack of the CV_32F boundary)
  end_to_end   configs[2]: matcher + circle join + RANSAC/Gauss-Newton
  drop_in_per_call  the LITERAL drop-in path: the reference's sequence_odometry
               loop calling the plain C-ABI one function per call, one frame
               at a time, host pointers in and out (C++ loop, libviso_host.so)
  streaming    every step consumes FRESH host frames through pinned asynchronous
               uploads (feature-in and image-in): the PCIe-inclusive rate
  roofline     the dominant kernel against the ceilings that can bound it, each
               a fraction <= 1: HBM (PMC bytes), L2 (PMC requests), VALU issue
               (PMC instructions), v_sad_u8 / v_sad_u16 issue (scored pairs); kernel time
               from a single-stream pass measured live with HIP events
  cpu_baseline the CPU oracle on the host cores: per stage, matcher-only and
               end to end on one thread, and frames-parallel on all cores

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B] [--kp N]

N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
one rank per GPU; frames shard across ranks (each rank owns its own
subsequence: weak scaling), no data-path collective; RCCL only gathers the
final trajectory in the end-to-end leg.  `python bench.py --gpus N` on its own
starts those N ranks itself (a child process tree, before this process touches
the GPU); a world size that is not --gpus is an error, never an N = 1 line.

The timed region of K steps is repeated (at least 5 regions, at least ~0.6 s of
GPU work in total): `value` is the median region, min / max are beside it.
"""
import argparse
import concurrent.futures
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # HBM3E spec
L2_GATHER_PEAK_GBS = 18800.0   # "Indexed rows: gather": rows shared by every workgroup, served by the XCD's L2, chip-wide: the top of
                               # the guide's 16.8-18.8 TB/s range, for another row shape
L2_GATHER_MEASURED_GBS = (15000.0, 17000.0)   # tools/gather_ceiling.hip with THIS kernel's block shape (profiles/r01_gather_ceiling.txt);
                                              # 24-25 TB/s with longer-lived blocks
L2_STREAM_PEAK_GBS = 34500.0   # "L2 (per XCD)": aggregate
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
CLK_GHZ = 2.4                  # max clock
# VALU issue on gfx950 (tools/valu_rate.hip, 8 waves per SIMD of independent chains; profiles/r02_valu_rate.txt):
# plain 32-bit add / sub / logic / mov / v_fma_f32 issue at the FULL rate, one wave64 instruction per 2 cycles per SIMD
# (the guide's figure, the 157.3 TFLOP/s FP32 spec; measured 2.4-2.7); every other opcode the matcher's loop is made
# of is a HALF-rate opcode, 4 cycles (measured: v_sad_u16 4.56, DPP adds 4.47, v_med3_u32 4.43, v_min_u32 4.27,
# v_cndmask_b32 4.26, v_lshl_or_b32 4.42, v_cmp 4.58).  The VALU-issue ceiling below is the nominal full rate; the
# v_sad_u16 ceiling is that opcode's own (half) rate.
VALU_CYCLES_FULL = 2
VALU_CYCLES_SAD = 4
L2_REQ_BYTES = 128             # TCP_TCC_READ_REQ: one 128-B line per request (r01: 35.3 M requests for 4.6 GB of row gathers)
PROFILE_ROUND = "r06"
GATHER_KERNELS = ("match_union_kernel", "match_union8_kernel")   # rows gathered from the XCD's L2 by index


def b_alg_bytes(n, scored, m_out, dlen=121):
    """Algorithmic bytes of one batch: per match_desc call
    8(N1+N2) + 4 D N1 + 4 D C + 12 M_out, C = scored (query,candidate) pairs.
    Returns (stereo problems, temporal problems)."""
    nf = n.shape[0]
    stereo = temporal = 0
    for t in range(nf):
        nL, nR = int(n[t, 0]), int(n[t, 1])
        stereo += 8 * (nL + nR) + 4 * dlen * nL + 4 * dlen * int(scored[0, t]) + 12 * int(m_out[0, t])
        if t == 0:
            continue
        pL, pR = int(n[t - 1, 0]), int(n[t - 1, 1])
        temporal += 8 * (nL + pL) + 4 * dlen * nL + 4 * dlen * int(scored[1, t]) + 12 * int(m_out[1, t])
        temporal += 8 * (nR + pR) + 4 * dlen * nR + 4 * dlen * int(scored[2, t]) + 12 * int(m_out[2, t])
    return stereo, temporal


def b_min_bytes(n, m_out, dlen=121, elem=4):
    """Compulsory bytes: (8 + elem D)(N1 + N2) + 12 M_out per call; elem = 4 for the reference's
    f32 boundary layout, 2 for the packed u16 rows the matcher kernels actually read (256-B rows).
    Returns (stereo, temporal)."""
    nf = n.shape[0]
    row = elem * dlen if elem == 4 else 256
    stereo = temporal = 0
    for t in range(nf):
        nL, nR = int(n[t, 0]), int(n[t, 1])
        stereo += (8 + row) * (nL + nR) + 12 * int(m_out[0, t])
        if t == 0:
            continue
        pL, pR = int(n[t - 1, 0]), int(n[t - 1, 1])
        temporal += (8 + row) * (nL + pL) + 12 * int(m_out[1, t]) + (8 + row) * (nR + pR) + 12 * int(m_out[2, t])
    return stereo, temporal


def b_min_bytes_union8(n, m_out):
    """What match_union8_kernel (matcher variant 6) MUST read per temporal call: keypoints and 8-bit planes of both images
    ((8 + 128) B per keypoint), every query's own u16 row (256 B) and the u16 rows of the two candidates per query that are
    scored exactly -- at most every target row once from HBM (256 B x min(2 N1, N2)) -- plus the 12-B result rows.
    Sum over the temporal calls of the batch."""
    nf = n.shape[0]
    tot = 0
    for t in range(1, nf):
        for side, w in ((0, 1), (1, 2)):
            n1, n2 = int(n[t, side]), int(n[t - 1, side])
            tot += (8 + 128) * (n1 + n2) + 256 * n1 + 256 * min(2 * n1, n2) + 12 * int(m_out[w, t])
    return tot


def workload_name(args):
    geo = f"synthetic {args.width}x{args.height} stereo pairs, {args.kp} features/image"
    if args.clustered > 0:
        geo += f" ({args.clustered:.0%} of them clustered in blobs)"
    if (args.width, args.height, args.kp) == (1241, 376, 2000):
        tag = "configs[1]" if args.clustered == 0 else "configs[1] geometry, clustered features"
    elif (args.width, args.height, args.kp) == (2048, 1024, 8000):
        tag = "configs[4] geometry, matcher only, this rank's share"
    else:
        tag = "custom geometry"
    return f"{tag}: {geo}, SAD matcher only (pack + 3 match_desc/frame + sort)"


def kernel_source_sha():
    """sha256 over the sources libviso_hip.so is built from (libviso_amd/csrc: *.hip, *.h, *.cpp, Makefile), in name
    order.  tools/pmc_to_json.py and tools/pmc_summary.py store it in the counter files they write; load_pmc refuses
    counters taken on other sources."""
    d = os.path.join(ROOT, "libviso_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            h.update(name.encode() + b"\0")
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def load_pmc(kname, default_workload):
    """Counter figures of the committed rocprofv3 --pmc passes (profiles/, tools/profile_round.sh + tools/pmc_batch.sh,
    `--streams 1`, this command's defaults).  PMC counters cannot be read from inside this process; they are only
    used when the workload is the one those passes ran AND the files carry the sha256 of the kernel sources this
    tree holds: counters of another build are dropped, loudly."""
    out = {"hbm_bytes": None, "sq": None, "src": [], "dropped": [], "hbm_all": None}
    if not default_workload:
        out["dropped"].append("not the workload of the committed counter passes")
        return out
    sha = kernel_source_sha()

    def usable(d, name):
        got = d.get("kernel_source_sha256")
        if got == sha:
            return True
        out["dropped"].append(f"profiles/{name}: taken on sources {str(got)[:12]}, this tree is {sha[:12]} -> counters NOT used")
        print(f"bench.py: PMC file profiles/{name} does not belong to these kernel sources; its ceilings are null "
              f"(re-run tools/profile_round.sh + tools/pmc_batch.sh)", file=sys.stderr)
        return False
    name = f"{PROFILE_ROUND}_pmc_hbm.json"
    p = os.path.join(ROOT, "profiles", name)
    if os.path.exists(p):
        d = json.load(open(p))
        if kname in d.get("kernel", "") and usable(d, name):
            out["hbm_bytes"] = d["hbm_bytes_per_launch_corrected"]
            out["hbm_all"] = {d["kernel"]: d["hbm_bytes_per_launch_corrected"]}
            out["hbm_all"].update({k: v["hbm_bytes_per_launch_corrected"] for k, v in d.get("other_kernels", {}).items()})
            out["src"].append(f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                              "one stream, gfx950 x2 fetch correction)")
    else:
        out["dropped"].append(f"profiles/{name} does not exist")
    name = f"{PROFILE_ROUND}_pmc_sq.json"
    p = os.path.join(ROOT, "profiles", name)
    if os.path.exists(p):
        d = json.load(open(p))
        if usable(d, name):
            for k, v in d.get("kernels", {}).items():
                if kname in k:
                    out["sq"] = v
                    out["src"].append(f"profiles/{name} (rocprofv3 --pmc SQ_* / TCP_* passes, one stream)")
                    break
    else:
        out["dropped"].append(f"profiles/{name} does not exist")
    return out


def cpu_baseline(seq, st, tm, budget_s):
    """The oracle (C restatement of the reference's CPU path, -O2) timed on the host cores.  Never on the measured
    GPU path; outside every GPU timed region."""
    from oracle import pyoracle   # checker/baseline only
    nf = seq["kp"].shape[0]

    def run(lo, hi, matcher_only):
        t0 = time.perf_counter()
        o = pyoracle.sequence(seq["kp"][lo:hi], seq["desc"][lo:hi], seq["n"][lo:hi], st, tm, seq["param"],
                              seed=1, first_frame=lo, matcher_only=matcher_only)
        return time.perf_counter() - t0, o

    # calibrate on 2 pairs, then size every sample to ~budget/4
    t_cal, _ = run(0, 3, True)
    per_pair = t_cal / 2
    k = int(max(2, min(nf - 1, (budget_s / 4) / per_pair)))
    t_m, o_m = run(0, k + 1, True)
    t_e, o_e = run(0, k + 1, False)
    stg = o_e["stage_s"]
    cores = min(16, len(os.sched_getaffinity(0)))   # the GPU box gives one GPU's CPU share: 16
    kc = int(max(2, min(nf - 1, k // 2)))          # per thread; ranges start at staggered frames and may overlap
    step = max(1, (nf - 1 - kc) // max(1, cores - 1))
    chunks = [(min(i * step, nf - 1 - kc), min(i * step, nf - 1 - kc) + kc + 1) for i in range(cores)]
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(cores) as ex:   # ctypes releases the GIL during the call
        list(ex.map(lambda c: run(c[0], c[1], False), chunks))
    t_all = time.perf_counter() - t0
    return {"value": k / t_m, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"oracle (C restatement of src/viso.cpp, -O2, 1 thread) matcher-only (configs[1]) on the first {k} "
                      f"frame pairs of the same batch, {t_m:.1f} s; the reference itself cannot be built here (OpenCV/Boost/Eigen absent)",
            "end_to_end": {"value": k / t_e, "unit": "frames/s", "cores": 1,
                           "sample": f"configs[2] (matcher + circle + RANSAC/GN) on the same {k} pairs, {t_e:.1f} s"},
            "stage_seconds_per_frame": {"neighbour_search": stg[4] / k, "gate_and_sad_walk": (stg[0] - stg[4]) / k,
                                        "match_circle": stg[1] / k, "collect_triangulate_gather": stg[2] / k,
                                        "ransac_gauss_newton": stg[3] / k},
            "all_cores": {"value": kc * cores / t_all, "unit": "frames/s", "cores": cores,
                          "sample": f"configs[2], {cores} threads x {kc} frame pairs each (frames-parallel; the reference is "
                                    f"single threaded), {t_all:.1f} s", "nproc": len(os.sched_getaffinity(0))}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--frames", type=int, default=512, help="frame pairs per batch (per GPU)")
    ap.add_argument("--kp", type=int, default=2000, help="keypoints per image")
    ap.add_argument("--width", type=int, default=1241)
    ap.add_argument("--height", type=int, default=376)
    ap.add_argument("--clustered", type=float, default=0.0,
                    help="share of the synthetic features drawn around a few blobs instead of uniformly (0.7: real-image-like clustering; exercises the K cap and the overflow kernel)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="budget of the CPU baseline samples (all of them)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-streaming", action="store_true")
    ap.add_argument("--matcher", type=int, default=None, help="kernel for the temporal calls (viso_ctx_set_matcher); default: the build's")
    ap.add_argument("--gn-split", type=int, default=0, help="viso_ctx_set_gn_split (0 = the build's default)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--force-collective", action="store_true",
                    help="build the process group and run barrier / all_reduce(MAX) / the record all_gather even with ONE rank "
                         "(the one-GPU box's way through RCCL; the N = 1 default touches no process group)")
    ap.add_argument("--images", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-images", action="store_true",
                    help="skip the resident image-in legs (device-side descriptor extraction / Harris on synthetic images)")
    ap.add_argument("--min-region-seconds", type=float, default=0.6,
                    help="the K-step timed region is repeated until this much GPU work has been timed (at least 5 regions)")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent batches in flight per GPU, one HIP stream each (steps go round robin over them)")
    ap.add_argument("--image-frames", type=int, default=512,
                    help="frame pairs per batch of the image-in legs (128 until late in round 4: the synthetic images take 35 ms "
                         "per frame to paint; 512 like the other legs: +16 %% / +6 %%)")
    ap.add_argument("--e2e-streams", type=int, default=5,
                    help="batches in flight for the end-to-end leg (each adds a RANSAC stream of its own: 5 measured +1.3 %% over 3; 0 = --streams)")
    ap.add_argument("--no-i16", action="store_true", help="skip the resident int16-rows matcher leg")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the per-call drop-in leg (the plain C-ABI, one reference function per call)")
    ap.add_argument("--drop-in-frames", type=int, default=256, help="frame pairs of the per-call drop-in leg")
    ap.add_argument("--ab-variants", default="", help="matcher variants timed by --ab (default: all of the build)")
    ap.add_argument("--ab", action="store_true", help="also time the other matcher variants, interleaved, same process")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as a child process tree.  Nothing in this process
        # has touched the GPU (torch is not imported yet), and it is a child, not an exec.
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # never an honest-looking N = 1 line for a run that was asked for N GPUs (or the other way round)
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to run", file=sys.stderr)
        raise SystemExit(2)
    if world > 1:
        print(f"bench.py: rank {rank}/{world} started (pid {os.getpid()})", file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    # rehearsal on a 1-GPU box: VISO_BENCH_SAME_DEVICE=1 puts every rank on device 0 (use --backend gloo)
    same_device = os.environ.get("VISO_BENCH_SAME_DEVICE") == "1"
    if not same_device and torch.cuda.device_count() < (local_rank + 1):
        raise SystemExit(f"bench.py: rank {rank} needs device {local_rank}, the node has {torch.cuda.device_count()} "
                         "(VISO_BENCH_SAME_DEVICE=1 --backend gloo rehearses N ranks on one device)")
    dev_index = 0 if same_device else local_rank
    torch.cuda.set_device(dev_index)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"
    use_dist = world > 1 or args.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:      # --force-collective outside torchrun: a group of one
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group of {dist.get_world_size()} ranks for --gpus {args.gpus}")

    import libviso_amd
    from libviso_amd import synth
    from libviso_amd.abi import MatchParams

    variant = args.matcher if args.matcher is not None else libviso_amd.DEFAULT_MATCHER
    nf = args.frames + 1                      # B pairs need B+1 frames (one-frame halo)
    t_syn = time.perf_counter()
    seq = synth.make_sequence(1000 + rank, nf, n_kp=args.kp, width=args.width, height=args.height, cluster_frac=args.clustered)
    # a first multi-GPU run that looks hung is usually here: every rank paints its own synthetic data on the CPU first
    print(f"bench.py: rank {rank}/{world}: {nf} synthetic feature frames generated in {time.perf_counter() - t_syn:.1f} s", file=sys.stderr, flush=True)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    # S independent batches per GPU, each on its own context / HIP stream: consecutive steps go to
    # different streams, so one batch's latency-bound stages (sorts, RANSAC) overlap the next one's matcher
    n_streams = max(1, args.streams)
    lanes = []

    def add_lane():
        c = libviso_amd.Context(dev_index)
        libviso_amd.set_matcher_variant(variant, c)
        if args.gn_split:
            libviso_amd.set_gn_split(args.gn_split, c)
        b = libviso_amd.Batch(c, nf, args.kp)
        b.upload(seq["kp"], seq["desc"], seq["n"])
        b.set_params(st, tm, seq["param"], seed=1, first_frame=rank * args.frames)
        lanes.append((c, b))
        return c, b

    for _ in range(n_streams):
        add_lane()
    ctx, batch = lanes[0]
    n_pipe = args.e2e_streams if args.e2e_streams > 0 else n_streams

    def pipeline_lanes():
        """The lanes of the legs that run the whole pipeline (matcher + join + RANSAC, with or without the image front
        end): every batch in flight brings a RANSAC stream of its own, and five of them overlap the latency-bound solver
        chains with the other steps' matcher work better than three (the matcher-only optimum, --streams): +1.3 % end to
        end, +2.7 % / +6 % on the image legs.  The extra lanes are created when the first such leg runs."""
        while len(lanes) < n_pipe:
            add_lane()
        return lanes[:n_pipe]

    def sync_all():
        for c, _ in lanes:
            c.synchronize()
        torch.cuda.synchronize()

    def barrier():
        sync_all()
        if use_dist:
            dist.barrier()

    def timed(fn, steps, warmup, objs=None):
        """fn(obj) is called once per step, round robin over the per-stream objects (default: the resident batches).
        One region: exactly `steps` steps between barrier + synchronize on both sides; MAX over ranks."""
        objs = objs or [b for _, b in lanes[:n_streams]]
        for i in range(warmup):
            fn(objs[i % len(objs)])
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(objs[i % len(objs)])
        sync_all()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed_regions(fn, steps, warmup, objs=None, max_regions=64):
        """The K-step region repeated: warm-up once, then R >= 5 regions with R * region >= --min-region-seconds
        (R derived from the first region's max-over-ranks time, so every rank agrees).  Returns the list of region
        times; the reported figure is the median."""
        dts = [timed(fn, steps, warmup, objs)]
        r = int(min(max_regions, max(5, math.ceil(args.min_region_seconds / max(dts[0], 1e-6)))))
        for _ in range(r - 1):
            dts.append(timed(fn, steps, 0, objs))
        return dts

    def spread(dts, units):
        """units per region / region time: median, min, max over the regions."""
        v = sorted(units / d for d in dts)
        return {"median": float(np.median(v)), "min": v[0], "max": v[-1], "regions": len(v),
                "timed_seconds_total": float(sum(dts))}

    # ---- the dominant kernel alone: a single-stream pass, HIP events on its stream ------------
    # (with several streams the events of the timed region also see the other streams' kernels sharing the CUs,
    # so that figure can exceed the step time; the roofline uses this pass).  It runs FIRST, under the conditions of the
    # committed `--streams 1` profile (profiles/*_kernel_stats_1stream.csv), behind a warm-up long enough for the clocks
    # to leave their idle state (the first ~30 launches of a fresh process read ~7 % longer); after the repeated
    # multi-stream regions below the card sits at its sustained clock and the same kernel reads 5-10 % longer again.
    n_single = max(8, min(40, args.steps))
    for _ in range(max(40, args.warmup)):       # ~50 ms of work first: a fresh process starts at idle clocks
        batch.run_matcher()
    ctx.synchronize()
    batch.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(n_single):
        batch.run_matcher()
    ctx.synchronize()
    step_ms_single = (time.perf_counter() - t0) * 1e3 / n_single
    kern_ms, kern_n1 = batch.kernel_ms()
    batch.kernel_timing(False)

    # ---- configs[1]: matcher only ------------------------------------------
    for _, b in lanes:
        b.kernel_timing(False)
    for i in range(max(args.warmup, n_streams)):
        lanes[i % n_streams][1].run_matcher()
    sync_all()
    for _, b in lanes:
        b.kernel_timing(True)
    dts_m = timed_regions(lambda b: b.run_matcher(), args.steps, 0)
    dt = float(np.median(dts_m))
    kern_ms_sum, kern_n = 0.0, 0
    for _, b in lanes:
        ms, n = b.kernel_ms()
        kern_ms_sum += ms * n
        kern_n += n
        b.kernel_timing(False)
    kern_ms_region = kern_ms_sum / max(kern_n, 1)
    frames_total = args.frames * args.steps * world
    fps = frames_total / dt
    fps_spread = spread(dts_m, frames_total)

    # ---- the same step with the descriptors resident as int16 rows (viso_batch_upload_i16: the lossless boundary format) --
    # the f32 headline pays 0.3 ms of every 1.0 ms step for the f32 -> u16 + u8 repack of the reference's CV_32F rows: the
    # boundary's cost, not the matcher's; this is the matcher step without it
    res_i16 = None
    if not args.no_i16:
        d16 = np.ascontiguousarray(seq["desc"].astype(np.int16))
        b16 = []
        for c, _ in lanes[:n_streams]:
            b = libviso_amd.Batch(c, nf, args.kp)
            b.upload_i16(seq["kp"], d16, seq["n"])
            b.set_params(st, tm, seq["param"], seed=1, first_frame=rank * args.frames)
            b16.append(b)
        dts16r = timed_regions(lambda b: b.run_matcher(), args.steps, n_streams, objs=b16)
        sc16, mo16 = b16[0].counters()
        res_i16 = {"fps": frames_total / float(np.median(dts16r)), "ms_per_step": float(np.median(dts16r)) / args.steps * 1e3,
                   "fps_spread": spread(dts16r, frames_total),
                   "workload": "the configs[1] step with the descriptors uploaded once as N x 121 int16 (viso_batch_upload_i16) instead of "
                               "CV_32F: pack_desc_i16_kernel reads half the bytes; same matches",
                   "same_counters_as_f32": bool(np.array_equal(sc16, batch.counters()[0]) and np.array_equal(mo16, batch.counters()[1]))}
        for b in b16:
            b.close()
        del d16

    scored, m_out = batch.counters()
    n_overflow = batch.overflow_count()
    balg_stereo, balg_temporal = b_alg_bytes(seq["n"], scored, m_out)
    bmin32_s, bmin32_t = b_min_bytes(seq["n"], m_out, elem=4)
    bmin16_s, bmin16_t = b_min_bytes(seq["n"], m_out, elem=2)
    kname = libviso_amd.matcher_kernel_name(ctx)
    pairs = int(scored[1:].sum())              # the timed kernel takes the temporal problems (2 of 3 calls, ~97 % of the pairs)
    t_k = kern_ms * 1e-3
    default_workload = (args.frames, args.kp, args.width, args.height, args.clustered) == (512, 2000, 1241, 376, 0.0)
    pmc = load_pmc(kname, default_workload)
    ceilings = {}
    # (iv) the arithmetic this path exists for.  u16 kernels: one v_sad_u16 wave-instruction (64 lanes x 2 elements) scores
    # 128 elements = one (query, candidate) pair.  match_union8_kernel ranks every pair on the rows' 8-bit planes: one
    # v_sad_u8 / v_sad_hi_u8 wave-instruction (64 lanes x 4 elements) scores TWO pairs, and only the two best candidates
    # of a query are scored again on the u16 rows.  All of them are half-rate opcodes (one per 4 cycles per SIMD)
    u8_kernel = kname == "match_union8_kernel"
    bmin_k = b_min_bytes_union8(seq["n"], m_out) if u8_kernel else bmin16_t
    sad_peak = N_SIMD * CLK_GHZ * 1e9 / VALU_CYCLES_SAD * (2 if u8_kernel else 1)
    valu_peak = N_SIMD * CLK_GHZ * 1e9 / VALU_CYCLES_FULL
    ceilings["sad_valu"] = {"achieved": pairs / t_k, "peak": sad_peak, "unit": "scored pairs/s",
                            "frac": pairs / t_k / sad_peak,
                            "what": ("useful v_sad_u8 issue: scored pairs (device counter) x HALF a wave-instruction each (a pair is 128 byte "
                                     "elements, an instruction does 256); the two exact v_sad_u16 scorings per query are not counted as useful"
                                     if u8_kernel else
                                     "useful v_sad_u16 issue: scored pairs (device counter) x 1 wave-instruction each")
                                    + "; a half-rate opcode: 4 cycles per wave-instruction per SIMD (tools/valu_rate.hip measures 4.56), 1024 SIMDs at 2.4 GHz"}
    valu_busy = None
    if pmc["hbm_bytes"] is not None:
        a = pmc["hbm_bytes"] / t_k / 1e9
        ceilings["hbm"] = {"achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                           "what": "PMC bytes per launch (FETCH_SIZE x2 + WRITE_SIZE) / live single-stream kernel time"}
    if pmc["sq"] is not None:
        sq = pmc["sq"]
        if "SQ_INSTS_VALU" in sq:
            a = sq["SQ_INSTS_VALU"] / t_k
            ceilings["valu_issue"] = {"achieved": a, "peak": valu_peak, "unit": "VALU wave-instructions/s", "frac": a / valu_peak,
                                      "what": "SQ_INSTS_VALU / kernel time against the NOMINAL issue rate (every opcode at the full rate: "
                                              "one wave64 instruction per 2 cycles per SIMD, 1024 SIMDs, 2.4 GHz). ~90 % of the kernel's "
                                              "instructions are half-rate opcodes (v_sad_u8 / v_sad_u16, DPP adds, selects, v_med3/v_min): see valu_busy"}
            if "SQ_ACTIVE_INST_VALU" in sq and "GRBM_GUI_ACTIVE" in sq:
                cyc = sq["GRBM_GUI_ACTIVE"] / 8.0            # rocprofv3 sums the 8 XCDs: shader cycles of one launch
                valu_busy = {"rocprof_VALUBusy": sq["SQ_ACTIVE_INST_VALU"] / 256.0 / cyc,
                             "cycles_per_valu_instruction_per_simd": cyc * N_SIMD / sq["SQ_INSTS_VALU"],
                             "note": "rocprofv3's VALUBusy formula (SQ_ACTIVE_INST_VALU / CU_NUM / GRBM_GUI_ACTIVE per XCD; the counter books "
                                     "one quad-cycle per instruction): the share of the launch's cycles in which the vector ALUs execute. "
                                     "cycles_per_valu_instruction_per_simd is the issue interval this launch sustained: 4 is what a loop of "
                                     "half-rate opcodes can reach (tools/valu_rate.hip measures 4.2-4.7 for them, 2.4-2.7 for the full-rate "
                                     "ones): what is left to gain is fewer or cheaper instructions, not a higher issue rate"}
        if "TCP_TCC_READ_REQ_sum" in sq:
            l2b = sq["TCP_TCC_READ_REQ_sum"] * L2_REQ_BYTES
            peak = L2_GATHER_PEAK_GBS if kname in GATHER_KERNELS else L2_STREAM_PEAK_GBS
            a = l2b / t_k / 1e9
            ceilings["l2"] = {"achieved": a, "peak": peak, "unit": "GB/s", "frac": a / peak,
                              "what": "TCP_TCC_READ_REQ_sum x 128 B (row gathers served by the XCD L2) against the guide's "
                                      + ("indexed-row gather rate" if kname in GATHER_KERNELS else "aggregate L2 rate")}
    # The unit that is FULL names the bound: rocprofv3's VALUBusy >= 0.90 means the vector ALUs hardly ever idle, whatever the
    # nominal-peak fractions say (they differ by a few hundredths and flip with the choice of a peak: 18.8 TB/s is the
    # guide's best case for another row shape, this kernel's own gather pattern measured 15-17 TB/s).  Otherwise the
    # highest fraction.
    if valu_busy is not None and valu_busy["rocprof_VALUBusy"] >= 0.90 and "valu_issue" in ceilings:
        bound = "valu_issue"
    else:
        bound = max(ceilings, key=lambda k: ceilings[k]["frac"])
    top = ceilings[bound]
    if "l2" in ceilings and kname in GATHER_KERNELS:
        a = ceilings["l2"]["achieved"]
        ceilings["l2"]["peak_range_measured_for_this_block_shape"] = list(L2_GATHER_MEASURED_GBS)
        ceilings["l2"]["frac_range_against_measured_peak"] = [a / L2_GATHER_MEASURED_GBS[1], a / L2_GATHER_MEASURED_GBS[0]]
    # what the whole matcher STEP does to HBM: PMC bytes of every kernel of the step / ms_per_step / 8 TB/s, and the two
    # kernels that are HBM kernels (pack: a format conversion stream; stereo: few pairs per row) on their own
    step_hbm = None
    if pmc["hbm_all"]:
        tot = float(sum(pmc["hbm_all"].values()))
        step_s = dt / args.steps
        step_hbm = {"bytes_per_step": tot, "GB/s": tot / step_s / 1e9, "frac_of_8TBs": tot / step_s / 1e9 / HBM_PEAK_GBS,
                    "ms_per_step": step_s * 1e3,
                    "note": "sum over the step's kernels of PMC (FETCH_SIZE x2 + WRITE_SIZE) bytes per launch, one-stream counter passes, "
                            "divided by the measured step time of this run (3 batches in flight)"}
        kms = {}
        p1 = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_kernel_stats_1stream.csv")
        if os.path.exists(p1):
            import csv
            for r in csv.DictReader(open(p1)):
                kms[r["Name"]] = float(r["AverageNs"]) * 1e-6
        per = {}
        for kn, b in pmc["hbm_all"].items():
            ms = next((v for k, v in kms.items() if k.split("(")[0] == kn.split("(")[0]), None)
            if ms and b > 50e6:
                per[kn.split("(")[0]] = {"ms_alone": ms, "hbm_bytes": b, "GB/s": b / ms / 1e6, "frac_of_8TBs": b / ms / 1e6 / HBM_PEAK_GBS}
        step_hbm["kernels"] = per
        step_hbm["kernels_note"] = f"ms_alone: profiles/{PROFILE_ROUND}_kernel_stats_1stream.csv (one batch in flight)"
    roofline = {
        "bound": bound, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
        "traffic": pmc["hbm_bytes"], "traffic_source": "; ".join(pmc["src"]) or None,
        "pmc_dropped": pmc["dropped"] or None, "kernel_source_sha256": kernel_source_sha(),
        "kernel": kname, "kernel_ms": kern_ms, "kernel_launches": kern_n1,
        "kernel_ms_source": f"HIP events on the kernel's stream, {kern_n1} launches, ONE batch in flight (step alone: {step_ms_single:.3f} ms)",
        "kernel_ms_in_timed_region_overlapped": kern_ms_region,
        "kernel_ms_in_timed_region_note": f"{n_streams} batches in flight share the CUs: not a per-step cost, may exceed ms_per_step",
        "ceilings": ceilings,
        "valu_busy": valu_busy,
        "bound_rule": "valu_issue when rocprofv3's VALUBusy >= 0.90 (the unit that is full), else the highest nominal-peak fraction",
        "step_hbm": step_hbm,
        "scored_pairs_per_launch": pairs,
        "overflow_queries_per_step": n_overflow,
        "overflow_note": "queries of one step (all three calls) handed to match_overflow_kernel: K cap, exact SAD tie, LDS list overflow",
        "rows8_note": ("the pack kernels also write the rows' 8-bit planes (128 B per keypoint) for this kernel: inside the timed step" if u8_kernel else None),
        "effective_bandwidth": {"algorithmic_bytes_per_launch": balg_temporal, "GB/s": balg_temporal / t_k / 1e9,
                                "note": "SURVEY 8(d) B_alg (every scored pair counted as a fresh 484-B f32 row) / kernel time: an "
                                        "effective figure served by L2/LDS, not comparable with the HBM peak"},
        "compulsory_bytes": {"per_launch_f32_boundary": bmin32_t, "per_launch_u16_rows": bmin16_t,
                             "per_launch_this_kernel": bmin_k,
                             "per_launch_this_kernel_what": ("match_union8_kernel: planes + keypoints of both images, every query's u16 row, the u16 rows of the two "
                                                             "exactly scored candidates per query (at most every target row once), result rows" if u8_kernel
                                                             else "the u16 rows + keypoints of both images, result rows"),
                             "traffic_over_compulsory": (pmc["hbm_bytes"] / bmin_k) if pmc["hbm_bytes"] else None,
                             "per_step_all_calls_f32_boundary": bmin32_s + bmin32_t,
                             "hbm_frac_if_only_compulsory": bmin_k / t_k / 1e9 / HBM_PEAK_GBS},
        "algorithmic_bytes_per_step_all_calls": balg_stereo + balg_temporal,
    }

    ab = None
    if args.ab:   # interleaved rounds in ONE process (cdna guide rule 24)
        vs = [int(v) for v in args.ab_variants.split(",")] if args.ab_variants else list(libviso_amd.MATCHER_VARIANTS)
        rounds = {v: [] for v in vs}
        walls = {v: [] for v in vs}
        names = {}
        for _ in range(5):
            for v in vs:
                libviso_amd.set_matcher_variant(v, ctx)
                names[v] = libviso_amd.matcher_kernel_name(ctx)
                batch.kernel_timing(True)
                for _ in range(4):
                    batch.run_matcher()
                rounds[v].append(batch.kernel_ms()[0])
                batch.kernel_timing(False)
                t0 = time.perf_counter()
                for _ in range(4):
                    batch.run_matcher()
                ctx.synchronize()
                walls[v].append((time.perf_counter() - t0) * 1e3 / 4)
        libviso_amd.set_matcher_variant(variant, ctx)
        ab = {"timed_kernel_ms_median": {names[v]: float(np.median(rounds[v])) for v in vs},
              "run_matcher_ms_median": {names[v]: float(np.median(walls[v])) for v in vs},
              "note": "timed kernel = the kernel that takes the temporal problems; one stream"}

    # ---- configs[2]: end to end (matcher + circle + RANSAC/GN) ---------------
    e2e = None
    collective = None
    if not args.no_e2e:
        # the end-to-end leg has its own number of batches in flight: every batch brings a RANSAC stream, and the chain of
        # latency-bound solver kernels of one step overlaps more of the other steps' matcher work with five than with three
        # (three is the matcher-only optimum); the extra batches exist for this leg only
        n_e2e = len(pipeline_lanes())
        dts2 = timed_regions(lambda b: b.run(), args.steps, n_e2e, objs=[b for _, b in pipeline_lanes()])
        dt2 = float(np.median(dts2))
        tr, ok, n_inl = batch.poses()
        if use_dist:   # the one exchange step: gather per-frame records {tr[6], ok, n_inl} (RCCL over xGMI)
            rec = torch.tensor(np.concatenate([tr, ok[:, None].astype(np.float64), n_inl[:, None].astype(np.float64)], 1),
                               device=coll_dev)
            out = [torch.empty_like(rec) for _ in range(world)]
            t0 = time.perf_counter()
            dist.all_gather(out, rec)
            gathered = torch.stack(out).cpu().numpy()
            collective = {"backend": "rccl" if args.backend == "nccl" else args.backend, "ranks": dist.get_world_size(),
                          "op": "all_gather of per-frame records {tr[6], ok, n_inl}", "gathered_records": int(gathered.shape[0] * (gathered.shape[1] - 1)),
                          "bytes_per_rank": int(rec.numel() * 8), "seconds": time.perf_counter() - t0,
                          "poses_ok_all_ranks": int(gathered[:, 1:, 6].sum())}
        err = float(np.abs(tr[1:][ok[1:] == 1] - seq["tr_gt"][1:][ok[1:] == 1]).max()) if ok[1:].any() else None
        e2e = {"fps": args.frames * args.steps * world / dt2, "ms_per_step": dt2 / args.steps * 1e3,
               "fps_spread": spread(dts2, args.frames * args.steps * world),
               "workload": "configs[2]: matcher + circle join + RANSAC/Gauss-Newton",
               "batches_in_flight": int(n_e2e),
               "poses_ok": int(ok[1:].sum()), "frames": int(args.frames),
               "max_abs_tr_err_vs_ground_truth": err}
        # the per-frame use of the reference's loop (one new frame pair at a time, pose needed before the next frame):
        # a 1-pair batch, host buffers in, pose out, synchronous -- latency, not throughput
        lb = libviso_amd.Batch(ctx, 2, args.kp)
        lb.set_params(st, tm, seq["param"], seed=1, first_frame=rank * args.frames)
        lat = []
        for i in range(24):
            t0 = time.perf_counter()
            lb.upload(seq["kp"][i:i + 2], seq["desc"][i:i + 2], seq["n"][i:i + 2])
            lb.run()
            lb.poses()
            lat.append(time.perf_counter() - t0)
        lb.close()
        e2e["latency_one_pair"] = {"ms_median": float(np.median(lat[4:]) * 1e3), "ms_min": float(np.min(lat[4:]) * 1e3),
                                   "what": "viso_batch_upload (pageable host memory, 2 frames) + viso_batch_run + viso_batch_get_poses of a "
                                           "1-pair batch, wall time per call sequence; the oracle needs ~35 ms for the same frame on one core"}

    # ---- the LITERAL drop-in path: the reference's loop calling the plain C-ABI one function per call ----------
    # (viso::sequence_odometry_per_call, libviso_amd/host: src/viso.cpp:1205-1327 with match_desc x3, collect_matches,
    # triangulate_rectified, match_circle, ransac_minimize_reproj per frame, host pointers in, host results out, the
    # copyTo carry-over of :1208-1222 -- what an unchanged kitti.cpp gets from adapters/libviso_hip.patch).  One frame at a
    # time, synchronous: a latency figure per GPU, not a throughput one.
    drop = None
    if not args.no_e2e and not args.no_drop_in:
        from libviso_amd import drop_in
        nd = min(nf, args.drop_in_frames + 1)
        dk, dd, dn = seq["kp"][:nd], seq["desc"][:nd], seq["n"][:nd]
        sync_all()

        def loop(cache=True, speculate=True, profile=False, want_matches=False):
            drop_in.plain_cache(cache)
            drop_in.plain_speculate(speculate)
            drop_in.run(dk[:12], dd[:12], dn[:12], seq["F"], seq["param"], seed=1, first_frame=rank * args.frames)   # warm-up, pattern learnt afresh below
            if profile:
                drop_in.plain_profile(True)
            o = drop_in.run(dk, dd, dn, seq["F"], seq["param"], seed=1, first_frame=rank * args.frames, want_matches=want_matches)
            rows = None
            if profile:
                drop_in.plain_profile(False)
                rows = drop_in.plain_profile_rows()
            return o, rows

        def per_call(o):
            return {k: {"calls": c, "us_per_call": us / c, "us_per_frame": us / (o["frames"] - 1)} for k, (c, us) in o["calls"].items()}
        o_best = None
        fps_runs = []
        for _ in range(3):
            o, _r = loop()
            fps_runs.append((o["frames"] - 1) / o["loop_s"])
            if o_best is None or o["loop_s"] < o_best["loop_s"]:
                o_best = o
        o_dir, _r = loop(speculate=False)
        _o, rows_dir = loop(speculate=False, profile=True)
        o_nc, _r = loop(cache=False, speculate=False)
        o_chk, _r = loop(want_matches=True)
        drop_in.plain_cache(True)
        drop_in.plain_speculate(True)
        st_plain = drop_in.plain_stats()
        tr_b, ok_b, inl_b = batch.poses()
        same = bool(np.array_equal(o_chk["ok"], ok_b[:nd]) and np.array_equal(o_chk["n_inl"][ok_b[:nd] == 1], inl_b[:nd][ok_b[:nd] == 1])
                    and all(np.array_equal(o_chk["matches"][w][t], batch.matches(w, t)) for w in range(3) for t in range(1 if w else 0, min(nd, 24))))
        tr_err = float(np.abs(o_chk["tr"][ok_b[:nd] == 1] - tr_b[:nd][ok_b[:nd] == 1]).max()) if ok_b[:nd].any() else None
        drop = {"fps": float(np.median(fps_runs)), "fps_runs": fps_runs, "frames": int(o_best["frames"]), "unit": "frames/s (one process, one GPU, one frame at a time)",
                "ms_per_frame": o_best["loop_s"] / (o_best["frames"] - 1) * 1e3,
                "workload": "configs[2] frames through viso::sequence_odometry_per_call (C++): per frame match_desc x3, collect_matches, "
                            "triangulate_rectified, match_circle, ransac_minimize_reproj on the plain (host-pointer) C-ABI, in the order of "
                            "src/viso.cpp:1240-1313, with the copyTo carry-over of :1208-1222; pageable host buffers in, host results out",
                "per_call_wall": per_call(o_best),
                "carry_over_us_per_frame": o_best["carry_s"] / (o_best["frames"] - 1) * 1e6,
                "every_call_direct": {"fps": (o_dir["frames"] - 1) / o_dir["loop_s"], "per_call_wall": per_call(o_dir),
                                      "per_call_gpu_us": {k: {kk: (vv / v["calls"] if kk != "calls" else vv) for kk, vv in v.items()} for k, v in rows_dir.items()},
                                      "note": "viso_plain_speculate(0): every call does its own work (image cache on); per_call_gpu_us = hipEvent "
                                              "brackets of the phases (viso_plain_profile): h2d = inputs, kernel, d2h = results, wait = host blocked, host = the call"},
                "every_call_direct_no_image_cache": {"fps": (o_nc["frames"] - 1) / o_nc["loop_s"], "per_call_wall": per_call(o_nc)},
                "equals_batch_family": same, "max_abs_tr_diff_vs_batch_family": tr_err,
                "plain_family_stats": st_plain,
                "bound": "one frame = one unavoidable round trip (the stereo call: 2 x 0.98 MB of descriptors from pageable memory into a pinned "
                         "shadow, then pulled over PCIe by the pack kernel) + a dependent chain of 13 small kernels (sort_kp, 2 x pack, head copy, one kernel for "
                         "the 3 match_desc problems, overflow, sort, join, 5 x RANSAC/GN) of ~210 us on an otherwise idle GPU; the later calls of the frame compare their arguments with what "
                         "the stereo call assumed (memcmp) and return"}

    # ---- streaming: every step consumes fresh host frames (pinned, asynchronous, stream ordered) -------------
    # sequence_odometry reads new images every frame (src/viso.cpp:1205-1231); the resident figures above never
    # touch PCIe.  Two different host sequences (the batch's and its time reversal: other pairs, other poses)
    # alternate, so no step re-reads what the device already holds; copies on one lane's stream overlap the other
    # lanes' kernels.
    streaming = None
    iseq = None
    nfi = min(nf, max(2, args.image_frames + 1))
    if not args.no_streaming:
        hosts = []
        for rev in (False, True):
            pk = libviso_amd.PinnedArray(seq["kp"].shape, np.float32)
            pd = libviso_amd.PinnedArray(seq["desc"].shape, np.float32)
            pk.a[...] = seq["kp"][::-1] if rev else seq["kp"]
            pd.a[...] = seq["desc"][::-1] if rev else seq["desc"]
            hosts.append((pk, pd, np.ascontiguousarray(seq["n"][::-1] if rev else seq["n"])))
        cnt = {"i": 0}

        def stream_step(full):
            def f(b):
                pk, pd, nn = hosts[cnt["i"] % 2]
                cnt["i"] += 1
                b.upload_async(pk.a, pd.a, nn)
                (b.run if full else b.run_matcher)()
            return f
        s_steps = max(2 * n_streams, args.steps // 4)
        dts = float(np.median(timed_regions(stream_step(False), s_steps, n_streams)))
        dte = float(np.median(timed_regions(stream_step(True), s_steps, n_streams))) if not args.no_e2e else None
        bytes_step = seq["kp"].nbytes + seq["desc"].nbytes + seq["n"].nbytes
        streaming = {"feature_in": {"fps_matcher": args.frames * s_steps * world / dts,
                                    "fps_end_to_end": args.frames * s_steps * world / dte if dte else None,
                                    "host_bytes_per_step_per_gpu": bytes_step, "pcie_GBps_per_gpu": bytes_step * s_steps / dts / 1e9,
                                    "workload": "every step: viso_batch_upload_async of N x 121 f32 descriptors + keypoints from pinned "
                                                "host memory, then the resident pipeline; alternating between two different sequences"}}
        # the same with int16 descriptor rows (viso_batch_upload_i16_async: the lossless encoding, half the bytes)
        h16 = []
        for rev in (False, True):
            pd16 = libviso_amd.PinnedArray(seq["desc"].shape, np.int16)
            pd16.a[...] = (seq["desc"][::-1] if rev else seq["desc"]).astype(np.int16)
            h16.append(pd16)

        def stream_step_i16(full):
            def f(b):
                i = cnt["i"] % 2
                cnt["i"] += 1
                b.upload_i16(hosts[i][0].a, h16[i].a, hosts[i][2], asynchronous=True)
                (b.run if full else b.run_matcher)()
            return f
        dts16 = float(np.median(timed_regions(stream_step_i16(False), s_steps, n_streams)))
        dte16 = float(np.median(timed_regions(stream_step_i16(True), s_steps, n_streams))) if not args.no_e2e else None
        bytes16 = seq["kp"].nbytes + seq["desc"].nbytes // 2 + seq["n"].nbytes
        streaming["feature_in_i16"] = {"fps_matcher": args.frames * s_steps * world / dts16,
                                       "fps_end_to_end": args.frames * s_steps * world / dte16 if dte16 else None,
                                       "host_bytes_per_step_per_gpu": bytes16, "pcie_GBps_per_gpu": bytes16 * s_steps / dts16 / 1e9,
                                       "workload": "as feature_in, descriptors as N x 121 int16 (viso_batch_upload_i16_async): same results, "
                                                   "half the descriptor bytes over PCIe"}
        for pd16 in h16:
            pd16.close()
        for pk, pd, _ in hosts:
            pk.close(); pd.close()
        for _, b in lanes:   # the resident legs below expect the original sequence
            b.upload(seq["kp"], seq["desc"], seq["n"])
        # image-in: uint8 images + keypoints cross PCIe, descriptors are extracted on the device
        t_syn = time.perf_counter()
        iseq = synth.make_image_sequence(2000 + rank, nfi, n_kp=args.kp, width=args.width, height=args.height)
        print(f"bench.py: rank {rank}/{world}: {nfi} synthetic stereo images painted in {time.perf_counter() - t_syn:.1f} s (CPU, ~35 ms per frame)", file=sys.stderr, flush=True)
        ibs, ihosts = [], []
        for c, _ in lanes[:n_streams]:
            ib = libviso_amd.Batch(c, nfi, args.kp)
            ib.upload_images(iseq["images"], iseq["kp"], iseq["n"])
            ib.set_params(st, tm, iseq["param"], seed=1, first_frame=rank * (nfi - 1))
            ibs.append(ib)
        for rev in (False, True):
            pi = libviso_amd.PinnedArray(iseq["images"].shape, np.uint8)
            pk = libviso_amd.PinnedArray(iseq["kp"].shape, np.float32)
            pi.a[...] = iseq["images"][::-1] if rev else iseq["images"]
            pk.a[...] = iseq["kp"][::-1] if rev else iseq["kp"]
            ihosts.append((pi, pk, np.ascontiguousarray(iseq["n"][::-1] if rev else iseq["n"])))

        def img_step(b):
            pi, pk, nn = ihosts[cnt["i"] % 2]
            cnt["i"] += 1
            b.upload_images_async(pi.a, pk.a, nn)
            b.run_images(False)
        i_steps = max(2 * n_streams, args.steps // 2)
        dti = float(np.median(timed_regions(img_step, i_steps, n_streams, objs=ibs)))
        ibytes = iseq["images"].nbytes + iseq["kp"].nbytes + iseq["n"].nbytes
        streaming["image_in"] = {"fps_end_to_end": (nfi - 1) * i_steps * world / dti, "frames_per_step": nfi - 1,
                                 "host_bytes_per_step_per_gpu": ibytes, "pcie_GBps_per_gpu": ibytes * i_steps / dti / 1e9,
                                 "workload": "every step: uint8 stereo images + keypoints uploaded asynchronously from pinned memory -> "
                                             "Sobel descriptor windows on device -> matcher + circle + RANSAC/GN"}
        for pi, pk, _ in ihosts:
            pi.close(); pk.close()
        for ib in ibs:
            ib.close()

    # ---- resident image-in pipeline: uint8 images + keypoints already in HBM ---------
    e2e_img = None
    if not args.no_images:
        if iseq is None:
            t_syn = time.perf_counter()
            iseq = synth.make_image_sequence(2000 + rank, nfi, n_kp=args.kp, width=args.width, height=args.height)
            print(f"bench.py: rank {rank}/{world}: {nfi} synthetic stereo images painted in {time.perf_counter() - t_syn:.1f} s (CPU, ~35 ms per frame)", file=sys.stderr, flush=True)
        isteps = max(n_pipe, args.steps // 2)
        ibs = []
        for c, _ in pipeline_lanes():   # one image batch per stream, same synthetic frames in each
            ib = libviso_amd.Batch(c, nfi, args.kp)
            ib.upload_images(iseq["images"], iseq["kp"], iseq["n"])
            ib.set_params(st, tm, iseq["param"], seed=1, first_frame=rank * (nfi - 1))
            ibs.append(ib)
        dts3 = timed_regions(lambda b: b.run_images(False), isteps, n_pipe, objs=ibs)
        dt3 = float(np.median(dts3))
        tri, oki, _ = ibs[0].poses()
        e2e_img = {"fps": (nfi - 1) * isteps * world / dt3, "frames": nfi - 1,
                   "fps_spread": spread(dts3, (nfi - 1) * isteps * world),
                   "workload": "resident uint8 images + keypoints -> Sobel descriptor windows on device -> matcher + circle + RANSAC/GN",
                   "batches_in_flight": int(n_pipe),
                   "poses_ok": int(oki[1:].sum()),
                   "max_abs_tr_err_vs_ground_truth": float(np.abs(tri[1:][oki[1:] == 1] - iseq["tr_gt"][1:][oki[1:] == 1]).max()) if oki[1:].any() else None}
        for ib in ibs:
            ib.close()
        # complete front-end on device too: binned Harris -> descriptors -> matcher -> solver
        dbs = []
        for c, _ in pipeline_lanes():
            db = libviso_amd.Batch(c, nfi, 1200)
            db.upload_images_only(iseq["images"])
            db.set_params(st, tm, iseq["param"], seed=1, first_frame=rank * (nfi - 1))
            dbs.append(db)

        def detect_and_run(b):
            b.detect()
            b.run_images(False)
        dts4 = timed_regions(detect_and_run, isteps, n_pipe, objs=dbs)
        dt4 = float(np.median(dts4))
        trd, okd, _ = dbs[0].poses()
        e2e_img["with_harris_detection"] = {
            "fps": (nfi - 1) * isteps * world / dt4,
            "fps_spread": spread(dts4, (nfi - 1) * isteps * world),
            "workload": "resident uint8 images only -> binned Harris (1200 corners/image, 24x5 bins) -> descriptors -> matcher + circle + RANSAC/GN",
            "poses_ok": int(okd[1:].sum()),
            "max_abs_tr_err_vs_ground_truth": float(np.abs(trd[1:][okd[1:] == 1] - iseq["tr_gt"][1:][okd[1:] == 1]).max()) if okd[1:].any() else None}
        for db in dbs:
            db.close()

    # ---- CPU baseline: the oracle on bounded samples of the same workload (rank 0, N = 1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(seq, st, tm, args.cpu_seconds)

    if rank == 0:
        line = {
            "metric": "stereo_frames_per_sec_1241x376_matcher",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "value_spread": fps_spread,
            "timing": f"the {args.steps}-step region (barrier + synchronize on both sides, max over ranks) repeated "
                      f"{fps_spread['regions']} times after the warm-up; value / ms_per_step = the median region",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": workload_name(args),
                       "frames_per_step_per_gpu": args.frames,
                       "residency": "inputs resident in HBM when the timed region starts (the PCIe-inclusive rate is under \"streaming\")",
                       "parallelism": f"frames sharded over {world} rank(s), no collective; {n_streams} batches in flight per GPU (one HIP stream each); "
                                      f"the legs that run the whole pipeline (end_to_end, end_to_end_from_images): {n_pipe}"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "resident_i16": res_i16,
            "end_to_end": e2e,
            "drop_in_per_call": drop,
            "collective": collective,
            "streaming": streaming,
            "end_to_end_from_images": e2e_img,
            "matcher_ab": ab,
        }
        print(json.dumps(line), flush=True)
    for c, b in lanes:
        b.close()
    for c, b in lanes:
        c.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
