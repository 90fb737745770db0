"""-m gpu: the N > 1 code path on a 1-GPU box -- two ranks (torch.distributed.run, gloo) share cuda:0.
RCCL itself needs one GPU per rank, so the collective library differs from the 8-GPU run; everything else (arena
layout, factor exchange, the HIP combine kernel, bench.py's multi-rank flow) is the code the driver launches."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A port nobody listens on right now (a fixed one may still be in TIME_WAIT when the suite is run twice in a row)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _torchrun(script_args, port=None):
    port = port or _free_port()
    env = dict(os.environ, GSRAST_DIST_BACKEND="gloo", GSRAST_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def test_factor_exchange_equals_plain_allreduce():
    out = _torchrun([os.path.join("tests", "mr_exchange_check.py")])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    # the two ranks write to the same pipe: their lines can end up on one line, so match reports, not lines
    import re
    reports = re.findall(r"EXCHANGE_CHECK rank (\d+) worst \S+ same_on_all_ranks (True|False)", out.stdout)
    assert sorted(r for r, _ in reports) == ["0", "1"] and all(ok == "True" for _, ok in reports), out.stdout[-2000:]


def test_distributed_step_equals_reference_batch_loop_on_the_real_chain():
    """view_parallel.distributed_step (4 views on 2 ranks) over epilogue -> rasterizer -> loss against the sequential batch loop."""
    out = _torchrun([os.path.join("tests", "mr_step_check.py")])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    import re
    reports = re.findall(r"STEP_CHECK rank (\d+) worst \S+ stats_ok (True|False)", out.stdout)
    assert sorted(r for r, _ in reports) == ["0", "1"] and all(ok == "True" for _, ok in reports), out.stdout[-2000:]


@pytest.mark.parametrize("ranks", [1, 2])
def test_distributed_step_with_two_views_in_flight_on_the_real_chain(ranks):
    """The same check with each rank's views alternating between two streams (distributed_step(views_in_flight=2)): one process
    rendering all four views, and two ranks rendering two each."""
    env = dict(os.environ, GSRAST_TEST_IN_FLIGHT="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if ranks == 1:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, os.path.join("tests", "mr_step_check.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    else:
        os.environ["GSRAST_TEST_IN_FLIGHT"] = "2"
        try:
            out = _torchrun([os.path.join("tests", "mr_step_check.py")])
        finally:
            del os.environ["GSRAST_TEST_IN_FLIGHT"]
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    import re
    reports = re.findall(r"STEP_CHECK rank (\d+) worst \S+ stats_ok (True|False)", out.stdout)
    assert len(reports) == ranks and all(ok == "True" for _, ok in reports), out.stdout[-2000:]


@pytest.mark.parametrize("exchange", ["gather", "sparse", "factors", "allreduce"])
def test_bench_two_ranks(exchange):
    out = _torchrun(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--gaussians", "100000", "--exchange", exchange,
                     "--sweep", "", "--no-cpu-baseline"])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d, _ = json.JSONDecoder().raw_decode(out.stdout[out.stdout.rfind('{"metric"'):])
    assert d["n_gpus"] == 2 and d["config"]["views_per_step"] == 2 and d["value"] > 0 and d["scaling"] == "weak"


def test_bench_launches_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` with no torchrun environment -- the command the driver's scaling step runs -- starts two ranks
    itself and prints ONE JSON line with n_gpus = 2 (both ranks on cuda:0 over gloo here: this box has one GPU)."""
    env = dict(os.environ, GSRAST_DIST_BACKEND="gloo", GSRAST_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--gaussians", "100000"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["views_per_step"] == 2 and d["value"] > 0 and d["scaling"] == "weak" and d["steps"] == 3


def test_bench_refuses_more_ranks_than_devices():
    """Without GSRAST_SINGLE_DEVICE a 1-GPU box cannot run --gpus 2: non-zero exit code, no JSON line (never n_gpus != --gpus)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GSRAST_SINGLE_DEVICE", "GSRAST_DIST_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--gaussians", "50000"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert '{"metric"' not in out.stdout
