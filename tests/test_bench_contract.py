"""bench.py's one-line JSON contract (what the round driver parses), on a small workload."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--config", "1", "--events", "24", "--minutes", "3", "--window", "10", "--cpu-sample", "8"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                        # exactly one line on stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "events/s" and d["higher_is_better"] is True and d["scaling"] == "strong"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 24 * 1000.0 / d["ms_per_step"]) <= 0.02 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    # parity of the whole job against the planted offset and the oracle sample is part of the line
    assert d["parity"]["max_shift_err_samples_vs_planted"] <= 1.0
    assert d["parity"]["max_idx_err_vs_oracle_sample"] == 0 and d["parity"]["oracle_sample_searches"] >= 8
    assert d["config"]["global_events"] == 24 and d["config"]["events_per_gpu"] == [24]
    # `value` is a resident-state rate; the line also says what a one-shot job gets (set-up + one step)
    assert 0 < d["one_shot_events_per_s"] < d["value"] and sum(d["setup_ms"].values()) > 0
    assert "custom" in d["config"]["workload"]


def test_bench_default_workload_is_the_north_star_configuration():
    """No flags = BASELINE configs[2]: 3000 events, 2-h 12 kHz streams, +-120 s window (the target's configuration)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    c = bench.CONFIGS[2]
    assert (c["events"], c["minutes"], c["window"], c["rate"]) == (3000, 120.0, 120.0, 12000)
    import inspect
    assert 'default=2' in inspect.getsource(bench.main)


def test_pmc_traffic_of_every_bench_workload_is_current():
    """profiles/pmc_traffic.json (rocprofv3 FETCH_SIZE / WRITE_SIZE passes, tools/gpu_final_r5.sh) holds an entry for the default
    workload -- and for the other lines DESIGN.md quotes -- taken on THESE kernel sources: bench.py reports roofline.traffic
    only while the digest matches, so a kernel edit without new counter passes shows up here, not as a silent null."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        entries = json.load(f)
    digest = bench.kernel_source_digest()
    for key in ("config2/fft/float32/3000/w120/m120/n1", "config2/fft/uint8/3000/w120/m120/n1",
                "config2/fft/float32/3000/w120/m120/n1/hard0.05/off7.25", "config2/fft/float32/3000/w120/m120/n1/ccoeff_normed",
                "config1/fft/float32/1000/w60/m45/n1", "config4/fft/float32/5000/w120/m240/n1"):
        e = entries[key]
        assert e["kernel_source_digest"] == digest, (key, e["kernel_source_digest"], digest)
        k = e["kernels"]["mac_kernel"] if "mac_kernel" in e["kernels"] else e["kernels"]["mac_long_kernel"]
        assert k["fetch_bytes"] > 0 and k["write_bytes"] > 0 and k["write_source"] == "WRITE_SIZE"
        # (every one of these workloads takes the band-split form of the exclusion: its bound pass is bound_low_kernel)
        assert e["kernels"]["bound_low_kernel"]["fetch_bytes"] > 0 and e["kernels"]["mac_list_kernel"]["fetch_bytes"] > 0


def test_design_table_is_the_committed_evidence():
    """DESIGN.md's round-6 numbers table is generated from profiles/r06/final/bench_*_n1.json (tools/design_table.py): a table that
    differs from what the committed bench lines say fails here."""
    import subprocess
    import sys
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "design_table.py"), "--check"]) == 0
