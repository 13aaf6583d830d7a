"""Frame sharding across ranks and the one exchange step.

The only cross-frame dependency of sequence_odometry is the *_prev state
(reference src/viso.cpp:1208-1222): the relative motion of frame t needs frames
t-1 and t, and the trajectory is the prefix product pose_t = pose_{t-1} *
inv(Tr_t) (:1319).  So ranks take contiguous frame ranges with a one-frame halo
(each rank re-does the stereo match of its first frame), RANSAC streams are
keyed on the GLOBAL frame index (partition invariant), and the per-frame
records {tr[6], ok, n_inl} are all-gathered once: RCCL (backend "nccl") over
xGMI on GPUs, gloo on CPU.  64 B per frame; latency, not bandwidth.
"""
import numpy as np


def partition(n_frames, world):
    """[(first_frame, last_frame_inclusive)] per rank.  Rank r solves the
    pairs (t-1, t) for t in (first, last]; frames first..last are resident."""
    n_pairs = max(0, n_frames - 1)
    base, rem = divmod(n_pairs, world)
    out, t = [], 0
    for r in range(world):
        k = base + (1 if r < rem else 0)
        out.append((t, t + k))
        t += k
    return out


def gpu_engine(device=0):
    """engine(kp, desc, n, stereo, temporal, param, seed, first_frame) -> (tr, ok, n_inl)
    running the HIP batch pipeline."""
    import libviso_amd

    def run(kp, desc, n, stereo, temporal, param, seed, first_frame):
        ctx = libviso_amd.Context(device)
        b = libviso_amd.Batch(ctx, kp.shape[0], kp.shape[2], desc.shape[-1])
        b.upload(kp, desc, n)
        b.set_params(stereo, temporal, param, seed=seed, first_frame=first_frame)
        b.run()
        res = b.poses()
        b.close(); ctx.close()
        return res
    return run


def local_records(seq_kp, seq_desc, seq_n, stereo, temporal, param, seed, engine, rank, world):
    """This rank's share of the work: rec [n_frames, 8] = tr[6], ok, n_inl with only the rows of the
    pairs (t-1, t), t in (first, last], filled in."""
    n_frames = seq_kp.shape[0]
    first, last = partition(n_frames, world)[rank]
    rec = np.zeros((n_frames, 8), np.float64)
    if last > first:
        tr, ok, n_inl = engine(seq_kp[first:last + 1], seq_desc[first:last + 1], seq_n[first:last + 1],
                               stereo, temporal, param, seed, first)
        # local frame j is global frame first + j; local frame 0 is the halo (no pose)
        rec[first + 1:last + 1, :6] = tr[1:]
        rec[first + 1:last + 1, 6] = ok[1:]
        rec[first + 1:last + 1, 7] = n_inl[1:]
    return rec


def stitch(parts, n_frames):
    """parts[r] = rank r's rec (what the all-gather delivers) -> (tr, ok, n_inl) of the whole sequence."""
    full = np.zeros((n_frames, 8), np.float64)
    for r, (a, b) in enumerate(partition(n_frames, len(parts))):
        full[a + 1:b + 1] = np.asarray(parts[r])[a + 1:b + 1]
    return full[:, :6].copy(), full[:, 6].astype(np.int32), full[:, 7].astype(np.int32)


def run_sharded(seq_kp, seq_desc, seq_n, stereo, temporal, param, seed, engine, rank, world,
                dist=None, device="cpu"):
    """Process this rank's frame range with `engine`, all-gather the records and
    return (tr [n_frames,6], ok [n_frames], n_inl [n_frames]) for the whole
    sequence on every rank."""
    n_frames = seq_kp.shape[0]
    rec = local_records(seq_kp, seq_desc, seq_n, stereo, temporal, param, seed, engine, rank, world)
    parts = [rec]
    if world > 1:
        import torch
        t = torch.from_numpy(rec).to(device)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)                         # fixed-size records, one collective
        parts = [p.cpu().numpy() for p in out]
    return stitch(parts, n_frames)
