"""GPU: bench.py's contract line, and its process-group path (RCCL init, barrier, max over ranks) at the
one world size a single-GPU box allows."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_line_small_workload(gpu):
    r = subprocess.run([sys.executable, "bench.py", "--workload", "cfg1", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert KEYS <= set(line)
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["dtype"] == "u8" and line["value"] > 0
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12


def test_bench_under_torchrun_with_rccl_group(gpu):
    env = dict(os.environ, P2P_BENCH_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "bench.py", "--gpus", "1", "--workload", "cfg1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] == "weak"


def test_two_ranks_on_one_gpu_dry_run(gpu):
    """The N > 1 path with real kernels: two ranks share the one GPU of the box (gloo for the barrier and the max
    over ranks, because RCCL refuses two ranks on one device).  Whole-job value = both ranks' pixels over the
    slower rank's time."""
    env = dict(os.environ, P2P_BENCH_BACKEND="gloo", P2P_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "bench.py", "--gpus", "2", "--workload", "cfg1", "--steps", "20", "--warmup", "5"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["cpu_baseline"] is None
    assert line["config"]["views_per_gpu"] == 1 and line["value"] > 0


def test_multi_rank_defaults_config3_share_and_view_sharding(gpu):
    """--gpus N > 1 defaults to the metric's own configuration on every rank (one panorama x 36 views each, "weak");
    --workload cfg3 deals config 3's 64 panoramas (64 / N resident per GPU, "strong"); --scaling strong on config 2 deals
    its 36 views (pitch-major runs; on two ranks every other yaw of all pitch views), one job per rank.  Two gloo ranks on the one GPU."""
    env = dict(os.environ, P2P_BENCH_BACKEND="gloo", P2P_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
            "--master-addr", "127.0.0.1"]
    r = subprocess.run(base + ["--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "10", "--warmup", "2",
                               "--kind", "N", "--preroll-s", "0.1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = _last_json(r.stdout)
    assert line["scaling"] == "weak" and line["config"]["panos_per_gpu"] == 1 and line["config"]["views_per_gpu"] == 36
    assert line["metric"].startswith("Mpix/s remapped, 8K equirect->1080p x36 views") and line["value"] > 0
    # two ranks x 36 views x 1920 x 1080 per step
    assert abs(line["value"] * line["ms_per_step"] * 1e3 - 2 * 36 * 1920 * 1080) / (2 * 36 * 1920 * 1080) < 1e-6
    # ... and, next to the weak-scaling headline, one number about the sharding code: config 3's 64 panoramas dealt to the ranks
    s3 = line["secondary"]["cfg3_strong"]
    assert "error" not in s3 and s3["scaling"] == "strong" and s3["panos_per_gpu"] == 32 and s3["value_Mpix_s"] > 0
    assert abs(s3["value_Mpix_s"] * s3["ms_per_step"] * 1e3 - 64 * 36 * 1920 * 1080) / (64 * 36 * 1920 * 1080) < 1e-6
    r = subprocess.run(base + ["--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--workload", "cfg3", "--steps", "3", "--warmup", "1",
                               "--kind", "N", "--preroll-s", "0.1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = _last_json(r.stdout)
    assert line["scaling"] == "strong" and line["config"]["panos_per_gpu"] == 32 and line["config"]["views_per_gpu"] == 32 * 36
    assert "64 panos" in line["config"]["workload"] and line["value"] > 0
    for shard in ("rows", "views"):  # a band of rows of every view per rank (the default) | whole views per rank
        r = subprocess.run(base + ["--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "10", "--warmup", "2",
                                   "--workload", "cfg2", "--scaling", "strong", "--shard", shard, "--kind", "N", "--preroll-s", "0.1"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        line = _last_json(r.stdout)
        assert line["scaling"] == "strong" and line["config"]["launches_per_step"] == 1  # one job per rank
        if shard == "views":
            assert line["config"]["views_per_gpu"] == 18  # one masked job
        else:
            assert abs(line["config"]["views_per_gpu"] - 36 * 544 / 1080.0) < 1e-9  # rows 0..544 of all 36 views on rank 0
        # 36 views x 1920 x 1080 per step, whatever the number of ranks
        assert abs(line["value"] * line["ms_per_step"] * 1e3 - 36 * 1920 * 1080) / (36 * 1920 * 1080) < 1e-6
