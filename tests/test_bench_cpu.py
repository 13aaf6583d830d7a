"""bench.py's contract pieces that do not need a GPU: --gpus N is honoured or refused (never an N = 1 line for an
N-GPU request), and counter files that do not belong to the kernel sources in the tree are dropped."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)


def _json_lines(out):
    return [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]


def test_world_size_that_is_not_gpus_is_refused():
    r = _run(["--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 2 and "refusing" in r.stderr and not _json_lines(r.stdout)
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode == 2 and not _json_lines(r.stdout)


def test_gpus_2_starts_two_ranks_and_never_prints_an_n1_line():
    """Without WORLD_SIZE, `bench.py --gpus 2` launches its own 2 ranks (torch.distributed.run as a child).  Here there
    is no HIP device, so the ranks stop with the no-fallback error: what matters is that the parent reports failure and
    that no rank printed a result line."""
    r = _run(["--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--no-cpu"], {})
    assert r.returncode != 0 and not _json_lines(r.stdout)
    assert "needs a HIP device" in r.stderr
    assert "rank 0/2 started" in r.stderr and "rank 1/2 started" in r.stderr      # both ranks were started


def test_counters_of_other_sources_are_dropped(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.kernel_source_sha()
    assert len(sha) == 64 and sha == bench.kernel_source_sha()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: sha)
    k = "match_union_kernel(BatchMatchArgs)"
    for good in (True, False):
        s = sha if good else "0" * 64
        (prof / f"{bench.PROFILE_ROUND}_pmc_hbm.json").write_text(json.dumps(
            {"kernel": k, "hbm_bytes_per_launch_corrected": 1.0e9, "kernel_source_sha256": s}))
        (prof / f"{bench.PROFILE_ROUND}_pmc_sq.json").write_text(json.dumps(
            {"kernel_source_sha256": s, "kernels": {k: {"SQ_INSTS_VALU": 4.0e8}}}))
        out = bench.load_pmc("match_union_kernel", True)
        if good:
            assert out["hbm_bytes"] == 1.0e9 and out["sq"]["SQ_INSTS_VALU"] == 4.0e8 and not out["dropped"]
        else:
            assert out["hbm_bytes"] is None and out["sq"] is None and len(out["dropped"]) == 2
    assert bench.load_pmc("match_union_kernel", False)["hbm_bytes"] is None
