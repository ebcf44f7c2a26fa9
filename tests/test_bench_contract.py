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
