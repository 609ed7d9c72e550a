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
