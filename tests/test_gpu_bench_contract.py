"""The bench line the driver parses (task contract): one JSON object on stdout with the metric, the timing fields, `roofline` (the
dominant kernel priced against HBM, measured with hipEvents inside the timed regions) and, unless switched off, `cpu_baseline`.
Run here with a handful of steps: the figures are meaningless, the structure and the arithmetic between the fields are checked."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly one line on stdout
    return json.loads(lines[0])


def test_default_line_has_the_contract_fields_and_consistent_arithmetic():
    d = _bench("--gpus", "1", "--steps", "8", "--warmup", "2", "--repeats", "3", "--no-cpu-baseline")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]  # grad-steps/s of one GPU = 1 / step time
    n_reg = d["timing"]["regions"]  # short regions are repeated until ~1500 steps have been timed; the median region is reported
    assert n_reg >= 3 and len(d["timing"]["ms_per_step_all"]) == n_reg and n_reg * 8 >= 600
    assert n_reg % 2 == 1 and abs(d["ms_per_step"] - sorted(d["timing"]["ms_per_step_all"])[n_reg // 2]) <= 1e-9
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["algorithmic_bytes"] == 5 * (6 * 7744 * 512 * 4 + 2 * 7744 * 32 * 4 + 512 * 32 * 4)
    assert r["launches_timed"] == n_reg * 2 and r["timed_every"] == 4  # steps 0 and 4 of each 8-step region carry the event bracket
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    assert 0.2 < r["frac"] < 1.0  # an HBM-bound kernel on an MI355X, whatever the box
    assert r["launch_ms"] < d["ms_per_step"]
    # PMC traffic comes from separate passes: quoted only for the kernel sources it was collected on, else null with the reason
    assert (r["traffic"] is None and r["traffic_source"].startswith("null:")) or (r["traffic"] > 0 and "csrc digest" in r["traffic_source"])
    assert {"gather_B32", "sumtree_B32", "prioritized_protocol"} <= set(d["sampling"])
    names = [k["launch"] for k in d["kernels"]]
    assert any("dense0 wgrad" in n for n in names) and not any("finalize" in n for n in names)  # the pair kernel finishes dL/da3 itself
    hf = d["heads_fit"]  # the K-independent part of the step as a tracked number
    assert hf["heads"] == [1, 2, 3, 5] and len(hf["us_per_step"]) == 4 and hf["per_head_us"] > 0 and hf["fixed_us"] > 0
    assert abs(hf["fixed_us"] + 5 * hf["per_head_us"] - hf["us_per_step"][3]) < 0.15 * hf["us_per_step"][3]
    assert "traffic" in d["step_roofline"]


@pytest.mark.parametrize("flags,K,B", [(("--batch", "256"), 5, 256), (("--heads", "64"), 64, 32)])
def test_config_4_and_5_single_device_lines(flags, K, B):
    """BASELINE configs 4 (B = 256 per step) and 5 (K = 64) as kept single-device bench lines with their floors."""
    d = _bench(*flags, "--steps", "6", "--warmup", "2", "--repeats", "1", "--no-cpu-baseline")
    assert d["config"]["heads"] == K and d["config"]["batch_per_gpu"] == B and f"K={K} batch={B}" in d["metric"]
    nb = B // 32
    # (from three sample blocks on the data gradient is its own launch: the update kernel reads a3 and dh of every block, writes no dL/da3)
    assert d["roofline"]["algorithmic_bytes"] == K * (6 * 7744 * 512 * 4 + nb * ((2 if nb < 3 else 1) * 7744 * 32 * 4 + 512 * 32 * 4))
    sr = d["step_roofline"]
    assert sr["flops"] > 0 and sr["floor_us_mfma_f32"] > 0 and 0 < sr["frac_mfma"] < 1.0
    assert "sampling" not in d and "heads_fit" not in d  # side legs ride on the headline line only


def test_cpu_baseline_object():
    d = _bench("--steps", "4", "--warmup", "1", "--repeats", "1")
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert d["gpu_over_cpu"] > 10


def test_data_parallel_line_separates_rank_steps_from_global_steps():
    """The N > 1 line (rehearsed with one rank over RCCL): `value` counts rank-steps, so the line also carries the global step rate,
    the per-rank step time and t(1) / t(N) measured in the same run -- and RCCL's own log excerpt."""
    d = _bench("--gpus", "1", "--force-dp", "--steps", "8", "--warmup", "2", "--repeats", "3")
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp1"
    p = d["data_parallel"]
    assert abs(p["per_rank_ms_per_step"] - d["ms_per_step"]) <= 1e-9
    assert abs(p["global_steps_per_s"] - 1e3 / d["ms_per_step"]) <= 1e-6 * p["global_steps_per_s"]
    assert abs(p["rank_steps_per_s"] - d["value"]) <= 1e-9 and abs(p["rank_steps_per_s"] - d["n_gpus"] * p["global_steps_per_s"]) <= 1e-6 * d["value"]
    assert abs(p["samples_per_s"] - d["config"]["global_batch"] * p["global_steps_per_s"]) <= 1e-6 * p["samples_per_s"]
    assert p["single_gpu_ms_per_step"] > 0 and abs(p["scaling_efficiency"] - p["single_gpu_ms_per_step"] / d["ms_per_step"]) <= 1e-9
    assert 0.5 < p["scaling_efficiency"] <= 1.05  # one rank: the data-parallel schedule costs its factor exchange, nothing else
    assert "debug_excerpt" in d["rccl"]
    assert "cpu_baseline" not in d  # rank 0 of a single-GPU run only
