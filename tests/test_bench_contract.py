"""bench.py prints ONE JSON line carrying the keys its callers rely on, for the headline workload and the two
secondary ones (small batches here; the numbers themselves are not checked)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}
ROOF = {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"] + list(args),
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_headline_line():
    d = run_bench("--packets", "4096", "--cpu-sample", "256")
    assert KEYS <= set(d) and ROOF <= set(d["roofline"])
    assert d["ranks"] == {"world_size": 1, "backend": None, "gpus_requested": 1, "started_by": "single process",
                          "kernel_ms_per_rank": d["ranks"]["kernel_ms_per_rank"]} and len(d["ranks"]["kernel_ms_per_rank"]) == 1
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "Msamples/s" and d["value"] > 0 and d["roofline"]["bound"] == "hbm"
    assert d["config"]["samples_per_packet"] == 6000 and "workload" in d["config"]
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0
    assert cb["cores"] >= 1 and cb["host_cpu_count"] >= cb["cores"] and cb["config1_py_refshaped"]["value"] > 0
    assert d["roofline"]["traffic_source"].startswith("profiles/traffic.json")
    # the map + IFFT + CP stage alone, priced at message bytes + symbol rows
    assert d["map_ifft_cp_only"]["bytes_per_packet"] == 2058 + 58 * 640 and d["map_ifft_cp_only"]["ms"] > 0
    # the fused kernel and the two-kernel path were both run
    assert d["kernel_ms"]["k_tx"] > 0 and d["two_kernel_path"]["k_tx_msg_ms"] > 0
    # how the output buffer was placed is part of the line, with the ordinary allocation's time beside it
    pl = d["iq_placement"]
    assert pl["mode"] in ("spread", "probe") and (pl["mode"] != "spread" or pl["first_allocation_ms"] > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("--workload", "rawofdm", "--raw-symbols", "8192"),
                                  ("--workload", "rawofdm", "--raw-n", "1024", "--raw-cp", "128", "--raw-symbols", "4096"),
                                  ("--workload", "ragged", "--ragged-packets", "3000")], ids=["raw256", "raw1024", "ragged"])
def test_secondary_lines(args):
    d = run_bench("--placement", "probe", *args)
    assert KEYS <= set(d) and ROOF <= set(d["roofline"]) and d["iq_placement"]["mode"] == "probe"
    assert d["value"] > 0 and 0 < d["roofline"]["frac"] < 1 and "workload" in d["config"]


def test_parent_of_a_multi_rank_run_stays_off_the_gpu_and_reports_failure():
    """`python bench.py --gpus 2` without a launcher: the parent imports neither torch nor the library before
    it starts its ranks, and a rank that fails makes the whole run fail (here: ranks that need GPUs 0 and 1
    over RCCL on a machine that has at most one)."""
    code = ("import sys; sys.argv = ['bench.py', '--gpus', '2']; import bench; a = bench.parse_args(); "
            "assert a.gpus == 2 and 'torch' not in sys.modules and 'ofdm_amd' not in sys.modules "
            "and 'numpy' not in sys.modules; print('ok')")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs here: the failing case needs fewer GPUs than ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--packets", "64"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "a rank ended with status" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def plain_multi_rank(*args):
    """`python bench.py --gpus 2 ...` started plainly (no launcher, no WORLD_SIZE): bench.py starts the ranks.
    Rehearsed on one GPU with the ranks sharing it over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
                         + list(args), cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_plain_command_starts_two_ranks():
    d = plain_multi_rank("--packets", "4096")
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0 and "cpu_baseline" not in d
    r = d["ranks"]
    assert r["world_size"] == 2 and r["backend"] == "gloo" and r["gpus_requested"] == 2
    assert r["started_by"].startswith("bench.py") and len(r["kernel_ms_per_rank"]) == 2 and min(r["kernel_ms_per_rank"]) > 0
    assert d["config"]["packets_per_gpu"] == 4096 and "2 GPU(s)" in d["config"]["parallelism"]


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("--workload", "rawofdm", "--raw-symbols", "8192"),
                                  ("--workload", "ragged", "--ragged-packets", "3000")], ids=["raw256", "ragged"])
def test_plain_command_starts_two_ranks_secondary(args):
    d = plain_multi_rank(*args)
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and len(d["ranks"]["kernel_ms_per_rank"]) == 2


@pytest.mark.gpu
def test_plain_command_gather_line():
    d = plain_multi_rank("--packets", "4096", "--gather", "1", "--gather-chunks", "4")
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and "gathered on rank 0" in d["config"]["parallelism"]
    g = d["gather"]
    assert g["chunks"] == 4 and g["packets_per_chunk"] == 1024 and g["root_inbound_GBps"] > 0
    assert g["root_inbound_bound_GBps"] == pytest.approx(153.6)


@pytest.mark.gpu
def test_two_rank_gather_line():
    """bench.py --gpus 2 --gather 1 the way the driver launches it, rehearsed on one GPU: two ranks over
    gloo share the device (BENCH_BACKEND=gloo); the line keeps the N = 1 shape."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--packets", "4096", "--gather", "1",
                          "--gather-chunks", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0
    assert "gathered on rank 0" in d["config"]["parallelism"]
    assert d["ranks"]["started_by"] == "external launcher" and d["ranks"]["world_size"] == 2
