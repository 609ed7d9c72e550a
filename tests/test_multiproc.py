"""The N > 1 path: one process per rank, point-sharded MSMs, all-gather of the
partial sums, local EC reduction.  World size 2 with gloo on the CPU (checker = the oracle), and
the real HIP path with two ranks sharing the one GPU of the test box."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(mode, world, n):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_mp_shard_worker.py"), mode, str(n)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "MP-OK %s %d %d" % (mode, world, n) in res.stdout


@pytest.mark.parametrize("world,n", [(2, 16), (3, 10), (4, 12), (8, 12)])      # proof bytes independent of the number of ranks, up to a node's eight
def test_point_sharded_sum_is_exact_gloo_cpu(world, n):
    _run("cpu", world, n)


@pytest.mark.gpu
@pytest.mark.parametrize("world,n", [(2, 256), (3, 1000), (4, 512), (2, 1 << 17)])      # the last: a realistic slice (2^16 per rank, LDS sort path, c = 16)
def test_point_sharded_prove_gpu(world, n):
    _run("gpu", world, n)


def _bench(args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_bench_gpus_2_without_a_gpu_refuses_instead_of_printing_a_one_gpu_line():
    """`python bench.py --gpus N` is its own launcher; where it cannot start N ranks on real devices it must fail, never fall through to N = 1."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the launcher would start the ranks (covered by the -m gpu test)")
    res = _bench(["--gpus", "2", "--log-n", "12", "--steps", "2"], timeout=300)
    assert res.returncode != 0
    assert '"metric"' not in res.stdout and '"n_gpus"' not in res.stdout, res.stdout[-500:]


def test_bench_rank_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode != 0 and '"metric"' not in res.stdout


@pytest.mark.gpu
def test_bench_gpus_2_starts_its_own_two_ranks():
    """The N-rank line as the driver asks for it: `python bench.py --gpus 2` with no launcher around it.  On the one-GPU test box the two ranks share
    the card and exchange over gloo (rehearsal); the line must say n_gpus 2, carry the parity gate's verdict and the proof of groth16.ml:123-161
    must not depend on N (the gate compares with the oracle's trapdoor evaluation, which knows nothing of ranks)."""
    import json
    res = _bench(["--gpus", "2", "--log-n", "12", "--steps", "2", "--warmup", "1", "--settle", "0"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    import torch
    assert d["n_gpus"] == 2 and d["config"]["constraints"] == 4096 and d["scaling"] == "strong"
    assert d["config"]["rehearsal_ranks_share_gpus"] == (torch.cuda.device_count() < 2)
    assert d["parity"] and "oracle" in d["parity"]["checked"]
