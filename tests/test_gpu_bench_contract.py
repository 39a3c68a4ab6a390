"""bench.py prints ONE JSON line with the driver's contract keys plus `roofline` and `cpu_baseline`; checked on the
quick workloads (the default, config 5, takes a minute and is run by the driver itself), and under
torch.distributed.run with one rank (the launch line the driver uses for N > 1)."""
import json
import os
import subprocess
import sys

import pytest

from plo_testlib import ROOT

pytestmark = pytest.mark.gpu
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"}
ROOF = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
CPUB = {"value", "unit", "cores", "kind", "sample"}


def one_line(cmd, env=None):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["winograd", "tril"])
def test_bench_line_has_the_contract_keys(hip, workload):
    d = one_line([sys.executable, "bench.py", "--workload", workload, "--steps", "2", "--warmup", "1"])
    assert KEYS <= set(d) and ROOF <= set(d["roofline"]) and CPUB <= set(d["cpu_baseline"])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and d["roofline"]["frac"] == pytest.approx(d["roofline"]["achieved"] / d["roofline"]["peak"])
    assert "workload" in d["config"] and d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1


def test_bench_under_torch_distributed_run_one_rank(hip):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    d = one_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                  "--master-port", "29541", "bench.py", "--gpus", "1", "--workload", "4x4x4_L", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["scaling"] == "weak"


@pytest.mark.parametrize("workload", ["tril", "cob", "kmethod"])
def test_tril_and_cob_run_under_torch_distributed_run(hip, workload):
    """the N > 1 launch line of the driver with one rank: seed shards / enumeration shards + the 8-byte all-reduce"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    d = one_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                  "--master-port", "29543", "bench.py", "--gpus", "1", "--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["scaling"] == "weak" and ROOF <= set(d["roofline"])
    assert d["roofline"]["bound"] == "issue" and 0 < d["roofline"]["frac"] <= 1 and ("roofline_hbm" in d or workload == "kmethod")


def test_lds_resident_cse_workload_reports_issue_roofline(hip):
    """4x4x4_L: state in LDS, the HBM figure says nothing -- the line carries the issue-rate roofline from the committed counters and
    keeps the HBM one beside it"""
    d = one_line([sys.executable, "bench.py", "--workload", "4x4x4_L", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert d["roofline"]["bound"] == "issue" and d["roofline"]["frac"] == pytest.approx(d["roofline"]["achieved"] / d["roofline"]["peak"])
    assert d["roofline_hbm"]["bound"] == "hbm" and d["value"] > 1e7
