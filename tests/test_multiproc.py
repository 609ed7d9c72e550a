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
@pytest.mark.parametrize("world,n", [(2, 256), (3, 1000), (4, 512), (2, 1 << 16)])      # the last: a realistic slice (1.5 * 2^16 G1 points per rank: the two-level LDS sort, c = 16; 2^17 ran here until round 5: 31 s)
def test_point_sharded_prove_gpu(world, n):
    _run("gpu", world, n)


@pytest.mark.gpu
def test_rccl_branches_of_the_collectives_on_a_real_communicator():
    """Every `dist.get_backend() == "nccl"` branch of zukelang_amd/groth16.py (device-resident partial sums, all_gather_into_tensor,
    all_to_all_single with per-destination splits, zk_groth16_combine_device) on a real RCCL communicator.  One rank per device is all
    RCCL allows: the world is the number of cards the box has (up to four -- the box's process guard allows six on the GPU), so the one-GPU
    box runs a world of one (self-exchanges through the code N ranks run) and a multi-GPU node runs RCCL between devices with no edit.  The
    proofs must be the trapdoor oracle's."""
    import torch
    _run("gpu-rccl", max(1, min(4, torch.cuda.device_count())), 1000)


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
    res = _bench(["--gpus", "2", "--log-n", "12", "--steps", "2", "--warmup", "1", "--settle", "0", "--config4-world", "2", "--config4-log-n", "10"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    import torch
    assert d["n_gpus"] == 2 and d["config"]["constraints"] == 4096 and d["scaling"] == "strong"
    assert d["config"]["fr_stage"].startswith("distributed Fr stage (rehearsed")                 # the rehearsal ran and passed: no fallback
    c4 = d["other_workloads"][0]                                                                  # config 4's leg at the world size the flag names
    assert "BASELINE config 4" in c4["workload"] and c4["parity"] is True and c4["value"] > 0
    assert d["config"]["rehearsal_ranks_share_gpus"] == (torch.cuda.device_count() < 2)
    assert d["parity"].startswith("passed") and "oracle" in d["parity"]
    assert len(lines[0]) < 4096                        # the driver parses this line: compact, the detail is in bench_detail.json / stderr
    assert "BENCH_DETAIL {" in res.stderr


@pytest.mark.gpu
def test_bench_line_of_a_real_one_gpu_run_is_compact_and_complete():
    """The N = 1 line as the driver reads it (round-3 verdict: `parsed` was null for a 24 KB line): one real run at 2^12 with every leg the default run has
    (tau-power pass, derivation, derived pass, a second workload, Pinocchio off, the cpu_baseline ladder cut short) -- the LAST stdout line is < 4 KB of
    strict JSON carrying `roofline` and `cpu_baseline`, and bench_detail.json holds the long form."""
    import json
    res = _bench(["--log-n", "12", "--steps", "2", "--warmup", "1", "--settle", "0", "--sizes", "10", "--dense-rows", "10", "--no-pinocchio", "--cpu-baseline-budget", "1"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    last = [ln for ln in res.stdout.splitlines() if ln.strip()][-1]
    assert len(last) < 4096
    d = json.loads(last)
    assert d["n_gpus"] == 1 and d["unit"] == "constraints/s" and d["value"] > 0 and d["vs_baseline"] is None and d["dtype"] == "u32"
    assert d["config"]["constraints"] == 4096 and d["config"]["key_form"] == "tau_powers_uploaded_lagrange_derived_on_device" and d["config"]["tau_power_value"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["kernel"].startswith("msm_accumulate") and r["avg_launch_ms"] > 0 and 0 < r["frac"] < 1 and r["alu_frac"] < 1
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert d["parity"].startswith("passed") and [o["parity"] for o in d["other_workloads"]] == [True, True] and "dense_rows" in d["other_workloads"][1]["workload"]
    assert 0 < r["alu_frac_hw"] < r["alu_frac"] * 1.2 and "v_mad_u64_u32" in r["alu_peak_hw"]
    detail = json.load(open(os.path.join(ROOT, "bench_detail.json")))
    assert detail["roofline_g1"] and detail["roofline_g2"] and detail["cpu_baseline"]["ladder"] and detail["proof_compressed_hex"]


@pytest.mark.gpu
def test_bench_device_list_runs_the_one_process_multi_device_path():
    """`python bench.py --device-list 0,0`: ONE process, one key handle sharded over two entries of the device list (the path an OCaml host takes:
    zk_set_device_list, csrc/groth16_multi.hip) -- both entries are the one card of the test box, so the line must call itself a rehearsal; the parity
    gate compares the last timed proof with the oracle's trapdoor evaluation (groth16.ml:123-161: the bytes do not depend on how the sums are cut)."""
    import json
    res = _bench(["--device-list", "0,0", "--log-n", "12", "--steps", "2", "--warmup", "1", "--settle", "0", "--headline-only", "--no-cpu-baseline"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 1 and d["config"]["device_list"] == [0, 0] and d["config"]["rehearsal_ranks_share_gpus"] is True
    assert d["config"]["key_form"] == "tau_powers_uploaded_lagrange_derived_on_device" and d["config"]["tau_power_value"] > 0          # the multi-device derivation ran too
    assert d["parity"].startswith("passed") and "device list [0, 0]" in d["config"]["sharding"]
