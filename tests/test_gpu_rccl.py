"""-m gpu: RCCL itself under test on a 1-GPU box.

The multi-rank tests of test_gpu_multirank.py share ONE GPU between two processes and therefore run over gloo: everything but
the collective library.  Here a ONE-rank process group is initialised with backend "nccl" (== RCCL on ROCm) and
view_parallel.force_collectives() switches the size-1 short-circuits off, so every collective the 8-GPU run issues -- the
in-place `all_reduce(ReduceOp.AVG)` of allreduce_mean_inplace (view_parallel.py), the uint8 `ReduceOp.MAX` and the compacted
all-reduce / `all_gather_into_tensor` of exchange_gradients(sparse=True), the asynchronous all-gather started inside the backward
(overlap_factor_exchange), the asynchronous all-reduce + statistic reductions of distributed_step, barrier, broadcast -- goes
through the library that will run it there, on device buffers, on this stream model.  With one rank a mean is the identity, so
the scripts' own comparisons (factor exchange == plain all-reduce; distributed_step == the reference's batch loop,
scene/saro_gaussian.py:226-276) are exact checks of the plumbing."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, GSRAST_FORCE_COLLECTIVES="1", GSRAST_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "GSRAST_SINGLE_DEVICE"):
        env.pop(k, None)
    return env


def _run(args, timeout=900):
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=timeout)


def test_exchange_through_rccl_one_rank():
    """all-reduce(AVG) of 59 floats, factor exchange (dense, overlapped, sparse), RAW leaves: tests/mr_exchange_check.py over nccl."""
    out = _run([os.path.join("tests", "mr_exchange_check.py")])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    m = re.search(r"EXCHANGE_CHECK rank 0 worst (\S+) same_on_all_ranks (True|False) backend (\S+) world (\d+) avg_ok (\S+)", out.stdout)
    assert m, out.stdout[-2000:]
    assert m.group(2) == "True" and m.group(3) == "nccl" and m.group(4) == "1", m.groups()
    assert m.group(5) == "True", "RCCL rejected ReduceOp.AVG: allreduce_mean_inplace fell back to SUM + scale"


def test_distributed_step_through_rccl_one_rank():
    """distributed_step's asynchronous all-reduce(SUM) of the gradient cache beside the three statistic reductions, over nccl."""
    out = _run([os.path.join("tests", "mr_step_check.py")])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    reports = re.findall(r"STEP_CHECK rank (\d+) worst \S+ stats_ok (True|False)", out.stdout)
    assert reports == [("0", "True")], out.stdout[-2000:]


@pytest.mark.parametrize("exchange", ["gather", "sparse", "factors", "allreduce"])
def test_bench_one_gpu_through_the_nccl_code_path(exchange):
    out = _run(["bench.py", "--gpus", "1", "--force-collectives", "--steps", "3", "--warmup", "1", "--gaussians", "100000", "--exchange", exchange])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    d, _ = json.JSONDecoder().raw_decode(out.stdout[out.stdout.rfind('{"metric"'):])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["exchange"] == exchange and d["config"]["exchange_backend"] == "nccl"
    assert d["config"]["exchange_bytes_per_rank_and_step"]["rows"] > 0
