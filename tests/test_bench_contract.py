"""CPU checks of bench.py's plumbing: flags, constants, and that the profile summaries it
quotes (PMC traffic, matrix-pipe utilisation) are committed with the keys it reads."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_flags_and_defaults(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0
    assert a.batch == 4096 and a.sims == 64 and a.env == "CartPole-v1"   # the headline configuration
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "3"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 3)


def test_algorithmic_work_constants():
    # 2 * sum(in * out) over the layers of policy (4x32) and value (5x256) networks
    def flop(obs, act):
        pol = [obs, 32, 32, 32, 32, act]
        val = [obs, 256, 256, 256, 256, 256, 1]
        return 2 * sum(a * b for a, b in zip(pol, pol[1:])) + 2 * sum(a * b for a, b in zip(val, val[1:]))
    assert bench.FLOP_PER_EXPANSION["CartPole-v1"] == flop(4, 2) == 533376
    assert bench.FLOP_PER_EXPANSION["Acrobot-v1"] == flop(6, 3)
    assert bench.FLOP_PER_EXPANSION["MountainCar-v0"] == flop(2, 3)
    assert bench.PEAK_HBM_GBS == 8000.0 and abs(bench.PEAK_F32_MFMA_TFLOPS - 157.3) < 0.5


def test_committed_profile_summaries_are_readable():
    t = bench.pmc_traffic("k_search_fused_wave")
    assert t is not None and 1e8 < t < 1e10
    assert bench.pmc_traffic("k_search_wave_aos21") is not None
    u = bench.pmc_mfma_util("k_search_fused_wave")
    assert u is not None and 0.05 < u < 1.0
    stats = open(os.path.join(ROOT, "profiles", "r03_kernel_stats.csv")).read()
    assert "k_search_fused_wave" in stats and "k_search_wave_aos21" in stats
    line = open(os.path.join(ROOT, "profiles", "r03_bench_under_rocprof.log")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert "workload" in d["config"] and d["vs_baseline"] is None


def test_self_launch_runs_two_ranks_over_gloo():
    """`python bench.py --gpus 2` without a launcher starts its own ranks (as a child process,
    before anything touches a GPU) and rank 0 prints the one JSON line; --dry-run keeps the GPU
    work out so that the plumbing runs here."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                        "--steps", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] > 0


def test_launcher_and_flag_must_agree(monkeypatch):
    import pytest
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    with pytest.raises(SystemExit):
        bench.main()
