#!/usr/bin/env python3
"""Groth16 constraints/sec on BLS12-381 (BASELINE.json metric) on N MI355X.

Workload = Groth16 prove of the synthetic iterated-cubic R1CS with the proving key, the circuit and
the witness already resident in HBM.  A STEP = `--proofs-per-step` (16) consecutive proofs of the pipelined prover
(`--inflight` proofs in flight, one HIP stream each; the pipeline is NOT drained between steps).
N = 1: n = 2^20 constraints (BASELINE.json configs[2]: the largest configuration the metric names for ONE GPU) is the headline
`value`; the same run then times the other single-GPU configurations (2^16 = configs[1], 2^18, 2^22 = config 4's size on one
GPU, and Pinocchio 2^18 = config 5) and reports them under `other_workloads`, each with its own parity check.
N > 1 (`python bench.py --gpus N` starts its own N ranks; under torch.distributed.run it is one of them): the SAME proof of
2^log_n constraints, its three multi-scalar products sharded by base points over the ranks (one process per GPU) -- total work
fixed => "strong" scaling; `--weak` makes it 2^log_n constraints PER GPU instead (BASELINE config 4 = --gpus 8 --log-n 19 --weak).
Proofs go in rounds: rank j runs the Fr stage of the proofs it owns, one all-to-all per scalar vector over RCCL hands every rank
its slice of every proof's scalars, the 768-byte partial sums of a round travel in one all-gather (EC addition is not an RCCL
reduction operator) and are added on the GPU.

PARITY GATE: after every timed region the proof of the LAST timed (r, s) is compared with the oracle's trapdoor
evaluation (exact at any n; CPU, outside the timed region).  A mismatch aborts the run: no throughput is printed for
wrong proofs.

Prints ONE JSON line on rank 0 (contract: see the round brief): at most 4 KB (`compact_line`, LINE_BUDGET) -- the contract's fields, the dominant kernel's
`roofline`, `cpu_baseline`, the parity verdict, one short record per other workload.  The long form (both accumulate rooflines, ladders, per-kernel
milliseconds, proof bytes) goes to bench_detail.json beside this file and to stderr (`BENCH_DETAIL {...}`).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")   # before torch / HIP initialise (see zk_api.hip)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
T_START = time.perf_counter()
PROFILE_ROUND = "r05"


MAX_RANKS_PER_DEVICE = 6      # a GPU box admits at most 6 processes on its card at once


def visible_devices():
    """Number of HIP devices the ranks will see, counted in a CHILD process: the launcher itself must stay free of any GPU state."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else 0
    except Exception:
        return 0


def launch_ranks(n, argv):
    """Starts `n` ranks of this script (one process per GPU, torch.distributed.run on 127.0.0.1), relays rank 0's JSON line and returns the
    children's exit code.  Fewer visible devices than ranks: the ranks share devices and exchange over gloo (a REHEARSAL of the code path --
    the line says so in config.rehearsal_ranks_share_gpus); no device at all, or more than 6 ranks per device: refuse.  Never falls through to one rank."""
    import socket
    import subprocess
    ndev = visible_devices()
    if ndev == 0:
        print("bench.py: --gpus %d needs at least one MI355X (none visible); refusing to print a line for fewer ranks" % n, file=sys.stderr)
        return 2
    if n > ndev * MAX_RANKS_PER_DEVICE:
        print("bench.py: --gpus %d on %d visible device(s) would put more than %d ranks on one card; refusing" % (n, ndev, MAX_RANKS_PER_DEVICE), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)       # stderr passes through
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 3
    if line is not None and rc == 0:
        print(line)
    return rc


def seeded(seed):
    from zukelang_amd.r1cs import fr_stream
    st = fr_stream(seed)
    return lambda: next(st)


def oracle():
    """The CPU checker (oracle/): imported only by the parity gate and the cpu_baseline leg, never inside a timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    return O


def host_threads():
    """Threads for the CPU context leg: this process's CPU share -- the cgroup quota when there is one, else its affinity mask, at most 64
    (ZK_BENCH_CPU_THREADS overrides)."""
    if os.environ.get("ZK_BENCH_CPU_THREADS"):
        return max(1, int(os.environ["ZK_BENCH_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 64))


def pmc_traffic():
    """HBM bytes per launch per kernel family from the committed PMC collection (profiles/r02_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md says).
    Counters cannot be read from inside this process; the figures apply to the named workload only."""
    for rnd in (PROFILE_ROUND, "r03"):          # this round's collection when it exists, else the last one (the figures name their source file)
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", rnd + "_pmc_traffic.json")))
            t["file"] = "profiles/%s_pmc_traffic.json" % rnd
            return t
        except Exception:
            continue
    return None


def cpu_baseline(budget_s=20.0):
    """The oracle's LITERAL restatement of groth16.ml:116-161 + QAP.ml:120-135 (per-variable apply_powers = m*n single double-and-add
    scalar multiplications, schoolbook mul / div_rem) on ONE host core, at the ladder of sizes BASELINE.md 3.2 names -- n = 16, 64, 256 (and
    1024 when the fitted model says it fits the budget) -- with the fitted cost model t(n) = a * m * n + b * n^2 and what that model says about
    the benchmark sizes ("infeasible": it is never extrapolated into a throughput).  `value` is the LARGEST size measured.
    Also a checker: the GPU proof of every ladder sample must be byte-identical."""
    O = oracle()
    from zukelang_amd import r1cs as RC
    from zukelang_amd.groth16 import Groth16, PKey
    frs = lambda xs: bytes(RC.fr_bytes(xs))
    ladder, spent = [], 0.0
    sizes = [16, 64, 128, 256, 1024]          # the default budget stops after n = 128 (~0.16 + 2.5 + 10 s of one core): "about 10-30 s of CPU work"
    for n in sizes:
        if ladder:          # predicted from the last point with the O(m n) law (m = n + 2): skip what does not fit
            n0, t0 = ladder[-1]["n"], ladder[-1]["seconds"]
            pred = t0 * (n * (n + 2.0)) / (n0 * (n0 + 2.0))
            if spent + pred > budget_s:
                break
        cs, w = RC.iterated_cubic(n, next(RC.fr_stream(0x5EED0001)))
        rng = seeded(0x5EED0002)
        toxic = [rng() for _ in range(5)]
        r, s = rng(), rng()
        q = O.QAP(cs.n, cs.m, *[O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)])
        pk1, pk2, _, _ = q.groth16_setup(frs(toxic), cs.mid)
        t0 = time.perf_counter()
        rc, a, b, c = q.groth16_prove(pk1, pk2, cs.mid, frs(w), frs([r]), frs([s]), 1)
        dt = time.perf_counter() - t0
        assert rc == 0
        prover = Groth16(cs, PKey(np.frombuffer(pk1, dtype=np.uint8), np.frombuffer(pk2, dtype=np.uint8)))
        proof = prover.prove_rs(w, r, s)
        prover.close()
        if (proof.a, proof.b, proof.c) != (a, b, c):
            raise SystemExit("PARITY FAILURE: GPU proof differs from the oracle on the cpu_baseline sample n = %d" % n)
        spent += dt
        ladder.append({"n": n, "m": cs.m, "seconds": round(dt, 4), "constraints_per_s": n / dt, "gpu_proof_identical": True})
    # least squares for t = a * (m n) + b * n^2 over the ladder (two unknowns; the m n term -- 3 m n scalar multiplications -- dominates)
    A = np.array([[p["m"] * p["n"], p["n"] ** 2] for p in ladder], dtype=np.float64)
    y = np.array([p["seconds"] for p in ladder], dtype=np.float64)
    if len(ladder) >= 2:
        coef, *_ = np.linalg.lstsq(A, y, rcond=None)
        a_c, b_c = float(coef[0]), float(coef[1])
        if b_c < 0 or a_c <= 0:          # degenerate fit on a noisy box: fall back to the pure m n law through the largest point
            a_c, b_c = float(y[-1] / A[-1, 0]), 0.0
    else:
        a_c, b_c = float(y[-1] / A[-1, 0]), 0.0
    model = lambda n: a_c * (n + 2.0) * n + b_c * n * n
    year = 365.25 * 86400
    top = ladder[-1]
    return {"value": top["constraints_per_s"], "unit": "constraints/s", "cores": 1, "kind": "port",
            "sample": "literal groth16.ml:116-161 + QAP.ml:120-135 (3 m n single scalar-muls, schoolbook polynomials) on the iterated-cubic R1CS at n = %s "
                      "(m = n + 2), %.1f s in all; value = the largest size measured (n = %d, %.2f s); GPU proof of every sample byte-identical"
                      % (", ".join(str(p["n"]) for p in ladder), spent, top["n"], top["seconds"]),
            "ladder": ladder,
            "cost_model": {"form": "seconds(n) = a * m * n + b * n^2, m = n + 2 (least squares over the ladder)", "a": a_c, "b": b_c,
                           "at_benchmark_sizes": {"2^%d" % k: "infeasible (model: %.2g core-years per proof)" % (model(float(1 << k)) / year) for k in (16, 20, 22)}}}


def collect_families(L, _lib, nproofs):
    out = {}
    buf = C.create_string_buffer(8192)
    _lib.check(L.zk_profile_names(buf, 8192))
    for name in buf.value.decode().split(","):
        if not name:
            continue
        ms, cnt = C.c_double(), C.c_uint64()
        _lib.check(L.zk_profile_get(name.encode(), C.byref(ms), C.byref(cnt)))
        out[name] = {"ms_total": ms.value, "launches": cnt.value, "ms_per_proof": ms.value / nproofs}
    return out


COUNTERS = ("entries", "copies", "second_steps", "full_additions")


def collect_counters(L, _lib, nproofs):
    """Work counters of the accumulate launches (zk_profile_counter; gathered by the library during the one-proof-at-a-time pass), per proof."""
    out = {}
    for fam in ("msm_accumulate_g1", "msm_accumulate_g2"):
        d = {}
        for c in COUNTERS:
            v = C.c_uint64()
            _lib.check(L.zk_profile_counter(("%s:%s" % (fam, c)).encode(), C.byref(v)))
            d[c] = v.value / nproofs
        if d["entries"]:
            out[fam] = d
    return out


# ALGORITHMIC bytes per (scalar, point) pair of a multi-scalar product: 32 B scalar + the affine point.
PAIR_BYTES = {"g1": 128.0, "g2": 224.0}
# Field products per group addition in the accumulate kernels: the multiplier-bound model, in units of one 14 x 14-limb Montgomery
# product (392 multiply-adds).  A full mixed addition XYZZ += affine: G1 6 products + 2 squares (301) + 1 fused double product (588) = 3542 = 9.04;
# G2 on a lane pair 8 fused double products + 2 squares per lane = 2 x 5488 = 28.0.  The SECOND step of a chunk adds two affine points
# (mmadd-2008-s): G1 2 + 2 squares + 1 fused = 1974 = 5.04; G2 2 x 3136 = 16.0.  The FIRST entry of a chunk or of a run is a copy: no product.
MADD_PRODUCTS = {"g1": 9.04, "g2": 28.0}
# The multiplier ceiling from the HARDWARE, not from a kernel of this library: one 14 x 14-limb Montgomery product is 392 v_mad_u64_u32, which issues at
# ~5 cycles per wave64 and SIMD (scripts/proto/valu_rate.hip; DESIGN.md 4 -- 4.4-5 measured), 1024 SIMDs, 2.4 GHz maximum clock (MI355X_MICROARCH.md;
# under this load the chip holds ~2.2 GHz at its power cap, which the measured peak beside it includes and this figure does not).
ALU_PEAK_HW = 1024 * 2.4e9 / 5.0 * 64 / 392 / 1e9          # = 80.2 G Fp products/s
MMADD_PRODUCTS = {"g1": 5.04, "g2": 16.0}


def roofline_objects(fam_timed, n_timed, fam_alone, n_alone, pairs, world, windows, peak_products, traffic, workload_key, counters=None):
    """One roofline object per MSM bucket-accumulation family (`msm_accumulate_g1` covers the two G1 products A and C of
    a proof in ONE launch, `msm_accumulate_g2` the G2 product B).  achieved = algorithmic bytes per launch / average launch
    duration.  `achieved` / `frac` are the UN-OVERLAPPED launch (one proof in flight: the duration rocprofv3 lists for the kernel under
    profiles/); `timed_region` keeps the same launch as it ran inside the timed region, stretched by the other proofs sharing the SIMDs.
    `alu`: field products the launch really executed (the library's own counts of full additions / second steps / copies when `counters`
    has them -- identity bases, zero scalars and zero digits never enter a bucket -- else pairs x windows) against the measured multiplier peak."""
    objs = {}
    for key, fam in (("g1", "msm_accumulate_g1"), ("g2", "msm_accumulate_g2")):
        alg_per_proof = PAIR_BYTES[key] * pairs[key] / world
        o = {"bound": "hbm", "kernel": fam, "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_proof": alg_per_proof}
        cnt = (counters or {}).get(fam)
        if cnt:
            products = cnt["full_additions"] * MADD_PRODUCTS[key] + cnt["second_steps"] * MMADD_PRODUCTS[key]
            o["additions_per_proof"] = {k: cnt[k] for k in COUNTERS}
            o["additions_per_proof"]["source"] = "counted by the library (zk_profile_counter), one proof at a time"
        else:
            products = alg_per_proof / PAIR_BYTES[key] * windows * MADD_PRODUCTS[key]          # upper bound: one full addition per (point, window) digit
            o["additions_per_proof"] = {"entries": alg_per_proof / PAIR_BYTES[key] * windows, "source": "pairs x windows (upper bound: no counters in this pass)"}
        for tag, fams, nproofs in (("timed_region", fam_timed, n_timed), ("one_proof_in_flight", fam_alone, n_alone)):
            members = [k for k in fams if k == fam or k.startswith(fam + ":")]          # sub-steps of a family are "family:step"
            if not members or not fams[fam if fam in fams else members[0]]["launches"]:
                continue
            ms_per_proof = sum(fams[k]["ms_total"] for k in members) / nproofs
            launches_per_proof = max(fams[k]["launches"] for k in members) / nproofs
            avg_ms = ms_per_proof / launches_per_proof
            bytes_per_launch = alg_per_proof / launches_per_proof
            ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            mul_equiv = products / (ms_per_proof * 1e-3) / 1e9
            o[tag] = {"avg_launch_ms": avg_ms, "launches_per_proof": launches_per_proof, "algorithmic_bytes_per_launch": bytes_per_launch,
                      "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                      "alu": {"unit": "G Fp products/s", "achieved": mul_equiv, "peak_measured": peak_products,
                              "frac": mul_equiv / peak_products if peak_products else None,
                              "peak_hw": ALU_PEAK_HW, "frac_hw": mul_equiv / ALU_PEAK_HW,
                              "peak_hw_is": "1024 SIMDs x 2.4 GHz / 5 cycles per v_mad_u64_u32 (wave64) x 64 lanes / 392 multiply-adds per Fp product"}}
        t = None
        if traffic and workload_key in traffic.get("workloads", {}):
            t = traffic["workloads"][workload_key].get(fam, {}).get("hbm_bytes_per_launch")
        o["traffic"] = t
        if "one_proof_in_flight" in o:
            o["achieved"], o["frac"], o["frac_is"] = o["one_proof_in_flight"]["achieved"], o["one_proof_in_flight"]["frac"], "un-overlapped launch (one proof in flight)"
        elif "timed_region" in o:
            o["achieved"], o["frac"], o["frac_is"] = o["timed_region"]["achieved"], o["timed_region"]["frac"], "launch inside the timed region (no un-overlapped pass at N > 1)"
        # the rate the launch REALLY moves HBM bytes at: the PMC bytes per launch (profiles/) over this run's live launch time -- north_star's "rocprof-reported
        # achieved HBM GB/s ... against gfx950 peak" (the algorithmic figure above counts what the problem needs, this one what the kernel fetched and wrote)
        live = o.get("one_proof_in_flight") or o.get("timed_region")
        if t and live and live.get("avg_launch_ms"):
            o["traffic_gbs"] = t / (live["avg_launch_ms"] * live.get("launches_per_proof", 1.0) * 1e-3) / 1e9
            o["traffic_frac"] = o["traffic_gbs"] / HBM_PEAK_GBS
        objs[key] = o
    return objs


def bench_groth16(args, L, _lib, log_n, steps, warmup, inflight, settle, rank, world, dist, lagrange=False, replicated_fr=False, events=True, derive_upto=None,
                  family="iterated_cubic"):
    """Times `steps` steps of `--proofs-per-step` proofs at n = 2^log_n * world; returns the result dict (rank 0 checks parity).
    family: "iterated_cubic" (the metric's circuit, one entry per l / r row) or "dense_rows" (zukelang_amd/r1cs.py random_r1cs: 8 entries in
    every row of all three matrices, full-width coefficients mixed in, an eighth of the variables in no gate -- never the headline)."""
    from zukelang_amd import r1cs as RC
    from zukelang_amd.groth16 import Groth16
    t_setup = time.perf_counter()
    n = (1 << log_n) * world if args.weak else (1 << log_n)          # strong scaling by default: the same proof on every N
    if family == "dense_rows":
        cs, w = RC.random_r1cs(n, n + 2, 0xD0D0, nnz=(8, 8))
    else:
        cs, w = RC.iterated_cubic(n, next(RC.fr_stream(0x5EED0001)))
    rng = seeded(0x5EED0002)
    toxic = [rng() for _ in range(5)]
    it = iter(toxic)
    pk, _vk = Groth16.keygen(lambda: next(it), cs, lagrange=lagrange)
    # N > 1 with the derivation enabled for this total size: every rank uploads the key WHOLE, the ranks SHARE the derivation of its Lagrange form
    # (one of the three independent sets per rank, broadcast) and each keeps its shard (cut for equal work, zk_groth16_shard_range).  Otherwise (--derive-lagrange-upto -1, or a total size
    # beyond it) the key is sharded at upload and stays in tau-power form.
    derive_s = None
    total_log = (n - 1).bit_length()
    sharded_derived = world > 1 and derive_upto is not None and total_log <= derive_upto and not lagrange and not args.tau_power_key
    if sharded_derived:
        prover = Groth16(cs, pk)
        t_d = time.perf_counter()
        prover.derive_lagrange_shared(rank, world)      # one derived set per rank, broadcast over RCCL, this rank's shard installed
        derive_s = time.perf_counter() - t_d
        # default: the Fr stage of a proof runs on ONE rank (its owner) and the scalar slices travel (GroupProver: one all-to-all per
        # vector) -- replicated, every rank would run a (2^log_n x world)-constraint Fr stage per proof beside an MSM slice that does
        # not grow with the world size; --replicated-fr keeps the simpler scheme
    else:
        prover = Groth16(cs, pk, rank, world, lagrange=lagrange)
    prover.set_witness(w)
    w_host = RC.fr_bytes(w) if args.host_witness else None          # --host-witness: every proof gets the witness as a HOST buffer (one PCIe copy per proof)
    pps = args.proofs_per_step
    nproofs = steps * pps
    nwarm = warmup * pps
    rs = [(rng(), rng()) for _ in range(min(nproofs + nwarm, 64) + 8)]       # recycled: the parity gate checks the last one used

    def sync():
        _lib.check(L.zk_sync())
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    depth = max(1, min(inflight, 15))      # the chip runs 16 hardware queues side by side: 15 slots + the context stream
    if world > 1:
        depth = min(depth, 8)      # sharded proofs: the host also runs an all-gather + combine per proof
    prover.reserve_slots(depth if world == 1 or replicated_fr else 1)          # setup, not warm-up: slots are otherwise created at first use
    group = None
    if world > 1 and not replicated_fr:
        from zukelang_amd.groth16 import GroupProver
        group = GroupProver(prover)
    setup_s = time.perf_counter() - t_setup

    def run(first, count):
        """count proofs; proof i uses rs[(first + i) % len(rs)].  One GPU: `depth` of them in flight, proof i on slot i % depth.
        N > 1: rounds of `group.batch` (distributed Fr stage), each rank finishing the proofs it owns.
        Returns (last proof this rank holds, index into rs of that proof)."""
        if group is not None:
            got = group.prove_many([rs[(first + i) % len(rs)] for i in range(count)], combine_all=False)
            mine = [(g, (first + i) % len(rs)) for i, g in enumerate(got) if g is not None]
            return mine[-1] if mine else (None, None)
        last = None
        for i in range(count):
            if depth == 1:
                last = prover.prove_rs(w_host, *rs[(first + i) % len(rs)])
                continue
            if i >= depth:
                last = prover.prove_wait(i % depth)
            prover.prove_async(w_host, *rs[(first + i) % len(rs)], i % depth)
        if depth > 1:
            for i in range(max(0, count - depth), count):
                last = prover.prove_wait(i % depth)
        return last, (first + count - 1) % len(rs)

    last_idx = [None]
    counters = [None]

    def measure(with_families):
        """warm-up + settle, the timed region, single-proof latency (and, with_families, the un-overlapped per-family pass), the parity gate"""
        run(0, max(nwarm, 1))
        sync()
        # A box that has been idle starts in a low-power state and needs seconds of load before its clocks
        # settle (first bench of a fresh box: 8.4 ms/proof against 2.6 ms once warm).  Untimed: keep proving
        # until three consecutive batches are within 5 % of the best one, at most --settle seconds.
        t_settle = time.perf_counter()
        best, stable, batches = None, 0, 0
        while settle > 0 and (batches < 6 if dist is not None else (time.perf_counter() - t_settle < settle and stable < 3)):
            batches += 1          # N > 1: a fixed count, every rank must run the same number of (collective) proofs
            t0 = time.perf_counter()
            run(0, depth)
            sync()
            bt = time.perf_counter() - t0
            stable = stable + 1 if best is not None and bt <= 1.05 * best else 0
            best = bt if best is None else min(best, bt)
        # the MSM accumulate kernels are bracketed by HIP events on their own streams DURING the timed region
        # (level 1: two recycled event records per launch; nothing synchronises)
        _lib.check(L.zk_profile_reset())
        _lib.check(L.zk_profile_enable(1 if events else 0))
        t0 = time.perf_counter()
        proof, proof_idx = run(nwarm, nproofs)
        last_idx[0] = proof_idx
        sync()
        dt = time.perf_counter() - t0
        _lib.check(L.zk_profile_enable(0))
        if dist is not None:
            import torch
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        fam_timed = collect_families(L, _lib, nproofs)           # accumulate kernels inside the timed region (empty with --no-live-events)

        # ---- single-proof latency and every kernel family un-overlapped: one proof at a time, untimed w.r.t. `value`
        n_alone = 4 if log_n <= 20 else 2
        lat = None
        fam_alone = {}
        if world == 1:
            _lib.check(L.zk_profile_reset())
            prover.prove_rs(None, *rs[0])
            t1 = time.perf_counter()
            for i in range(n_alone):
                prover.prove_rs(None, *rs[1 + i])
            lat = (time.perf_counter() - t1) / n_alone
            if with_families:
                _lib.check(L.zk_profile_enable(2))
                for i in range(n_alone):
                    prover.prove_rs(None, *rs[1 + i])
                fam_alone = collect_families(L, _lib, n_alone)
                counters[0] = collect_counters(L, _lib, n_alone)
                _lib.check(L.zk_profile_enable(0))
            _lib.check(L.zk_profile_reset())

        # ---- PARITY GATE (CPU oracle, outside every timed region): the last timed proof this rank holds
        parity = None
        if proof is not None and not args.no_parity_gate:
            O = oracle()
            frs = lambda xs: bytes(RC.fr_bytes(xs))
            csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
            r_, s_ = rs[proof_idx]
            t2 = time.perf_counter()
            exp = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(toxic), frs([r_]), frs([s_]))
            if (proof.a, proof.b, proof.c) != exp:
                raise SystemExit("PARITY FAILURE: the last timed proof at n = 2^%d x %d differs from the oracle's trapdoor evaluation -- no throughput reported" % (log_n, world))
            parity = {"checked": "last timed proof == oracle groth16_prove_trapdoor (exact at any n), bytes of a | b | c", "oracle_s": round(time.perf_counter() - t2, 2)}
        return dt, fam_timed, proof, lat, fam_alone, n_alone, parity

    # A key in the reference's format (tau powers) can be turned into its Lagrange form ON THE DEVICE, once per key, without tau
    # (zk_groth16_pk_derive_lagrange): the per-proof basis conversion disappears, the proofs do not change.  When asked to, the run measures
    # the key as uploaded first, then derives (untimed, reported as derive_lagrange_s) and measures again: `value` is the second figure,
    # `tau_power_form` keeps the first.
    derive = derive_upto is not None and log_n <= derive_upto and world == 1 and not lagrange
    if derive and not args.derived_only and args.time_budget > 0:
        est = 31.0 * n / (1 << 20) * (1.08 if log_n > 20 else 1.0) + 12.0 * n / (1 << 22)          # derivation + the second measurement and its parity check
        spent = time.perf_counter() - T_START
        if spent + est > args.time_budget:
            print("bench.py: %.0f s spent, the 2^%d derivation (~%.0f s) does not fit --time-budget %.0f: this workload reports the key as uploaded only" % (spent, log_n, est, args.time_budget), file=sys.stderr)
            derive = False
    power_form = None
    if derive and args.derived_only:
        # profiling runs: only the path the headline runs (no tau-power pass whose kernels would mix into the kernel statistics)
        t_d = time.perf_counter()
        prover.derive_lagrange()
        derive_s = time.perf_counter() - t_d
        prover.reserve_slots(depth)
    elif derive:
        dt0, _ft, _pr, lat0, _fa, _na, par0 = measure(False)
        power_form = {"value": n * nproofs / dt0, "ms_per_proof": dt0 / nproofs * 1e3, "single_proof_latency_ms": None if lat0 is None else lat0 * 1e3, "parity": par0 is not None}
        t_d = time.perf_counter()
        prover.derive_lagrange()
        derive_s = time.perf_counter() - t_d
        prover.reserve_slots(depth)
    dt, fam_timed, proof, lat, fam_alone, n_alone, parity = measure(True)
    if power_form is not None:
        # the derivation is per-key preprocessing OUTSIDE the timed region: say what it costs and when it has paid for itself
        saved = power_form["ms_per_proof"] - dt / nproofs * 1e3
        power_form["derive_lagrange_s"] = round(derive_s, 2)
        power_form["break_even_proofs"] = int(derive_s * 1e3 / saved) + 1 if saved > 0 else None
        power_form["note"] = ("value = this key exactly as an OCaml keygen emits it (tau powers); the headline proves from the SAME key after a one-time, untimed "
                              "on-device derivation of its Lagrange form -- amortised after break_even_proofs proofs (derive_lagrange_s / per-proof saving)")

    # ---- context: the same job on the HOST cores with the same algorithmic freedom (oracle/fast_cpu.c: Pippenger + NTT
    # convolutions over the Lagrange-form pools this key now holds), all threads of this process's CPU share, one proof, untimed w.r.t. `value`;
    # its bytes must equal the GPU's last timed proof
    cpu_fast = None
    if world == 1 and (derive or lagrange) and log_n <= args.cpu_fast_upto and proof is not None and not args.no_cpu_baseline:
        O = oracle()
        frs = lambda xs: bytes(RC.fr_bytes(xs))
        csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
        threads = host_threads()
        t_k = time.perf_counter()
        fp = O.FastGroth16(cs.n, cs.m, *csr, cs.mid, prover.pool_points(1), prover.pool_points(2), threads)
        key_s = time.perf_counter() - t_k
        wb = frs(w)
        r_, s_ = rs[last_idx[0]]
        t_c = time.perf_counter()
        rc, ca, cb, cc = fp.prove(wb, frs([r_]), frs([s_]))
        cdt = time.perf_counter() - t_c
        fp.close()
        if rc != 0 or (ca, cb, cc) != (proof.a, proof.b, proof.c):
            raise SystemExit("PARITY FAILURE: the multi-threaded CPU prover and the GPU disagree at n = 2^%d" % log_n)
        cpu_fast = {"value": n / cdt, "unit": "constraints/s", "cores": threads, "kind": "port-fast", "s_per_proof": round(cdt, 3), "key_parse_s": round(key_s, 2),
                    "sample": "ONE proof of this workload (n = 2^%d) by oracle/fast_cpu.c: Pippenger bucket sums (XYZZ, signed digits, a slice of the points per thread) + "
                              "NTT convolutions (Newton-basis extrapolation) over the Lagrange-form pools read back from the device, %d threads; NOT the reference's "
                              "algorithm (that is cpu_baseline); portable C, about 1.5-2x blst's cost per field product; bytes equal to the GPU's last timed proof" % (log_n, threads)}

    p1 = 3 + (n + 2) + (n - 1) + cs.n_mid
    p2 = 2 + (n + 2)
    pairs = {"g1": (n + 2) + p1, "g2": p2}          # scalar-point pairs actually multiplied per proof: A (n+2) + C (the whole pool) | B
    res = {"log_n": log_n, "constraints": n, "variables": cs.m, "value": n * nproofs / dt, "unit": "constraints/s", "ms_per_proof": dt / nproofs * 1e3,
           "ms_per_step": dt / steps * 1e3, "timed_s": dt, "timed_proofs": nproofs, "proofs_in_flight": 2 * group.batch if group is not None else depth,
           "single_proof_latency_ms": None if lat is None else lat * 1e3, "single_proof_value": None if lat is None else n / lat,
           "setup_s": round(setup_s, 1), "derive_lagrange_s": None if derive_s is None else round(derive_s, 2), "tau_power_form": power_form, "parity": parity, "cpu_fast_context": cpu_fast, "pairs": pairs, "p1": p1, "p2": p2,
           "fam_timed": fam_timed, "fam_alone": fam_alone, "n_alone": n_alone, "nproofs": nproofs, "counters": counters[0],
           "kernel_ms_per_proof": {k: round(v["ms_per_proof"], 4) for k, v in sorted(fam_alone.items())},
           "proof_compressed_hex": proof.to_compressed().hex() if proof is not None else None,
           "group_batch": group.batch if group is not None else None}
    if group is not None:
        # exchange volume per proof and rank (device to device over xGMI): three scalar vectors, 32 B per element
        res["exchange"] = {"all_to_all_bytes_per_proof_sent_by_owner": 32 * ((min(group.nzA, group.p1) if group.clipA else group.p1) + group.p1 + group.p2),      # A: clipped to its non-zero prefix only when no rank's slice gets empty
                           "a_vector_clipped": group.clipA,
                           "all_gather_bytes_per_proof_per_rank": 768,
                           "slice_points_g1": [hi - lo for lo, hi in group.bounds1],
                           "slice_points_g2": [hi - lo for lo, hi in group.bounds2]}
    prover.close()
    del prover
    return res


def bench_pinocchio(args, L, _lib, log_n, nproofs, inflight, peak_products=None):
    """BASELINE config 5: Pinocchio Protocol-2 ZK prove (pinocchio.ml:427-514), witness resident, proofs pipelined over slots."""
    from zukelang_amd import r1cs as RC, pinocchio as PIN
    n = 1 << log_n
    cs, w = RC.iterated_cubic(n, next(RC.fr_stream(0x5EED0001)))
    st = RC.fr_stream(0x5EED0003)
    tox = [next(st) for _ in range(8)]
    it = iter(tox)
    pk, _vk = PIN.ZK.keygen(lambda: next(it), cs)
    prover = PIN.ZK(cs, pk)
    wb = RC.fr_bytes(w)
    prover.set_witness(wb)
    depth = inflight
    prover.reserve_slots(depth)
    ds = [[next(st) for _ in range(3)] for _ in range(16)]

    def run(count):
        last = None
        for i in range(count):
            if i >= depth:
                last = prover.prove_wait(i % depth)
            prover.prove_async(*ds[i % len(ds)], i % depth)
        for i in range(max(0, count - depth), count):
            last = prover.prove_wait(i % depth)
        return last, (count - 1) % len(ds)
    fams = {}

    def measure():
        run(2 * depth)
        _lib.check(L.zk_sync())
        # the accumulate launches are bracketed by HIP events on their own streams inside the timed region (level 1), as for Groth16
        _lib.check(L.zk_profile_reset())
        _lib.check(L.zk_profile_enable(0 if args.no_live_events else 1))
        t0 = time.perf_counter()
        proof, idx = run(nproofs)
        _lib.check(L.zk_sync())
        dt = time.perf_counter() - t0
        _lib.check(L.zk_profile_enable(0))
        fams["timed"] = collect_families(L, _lib, nproofs)
        t1 = time.perf_counter()
        for i in range(2):
            prover.prove_with(wb, *ds[i])      # the witness as 32-byte values: no Python conversion inside the latency
        lat = (time.perf_counter() - t1) / 2
        _lib.check(L.zk_profile_reset())
        _lib.check(L.zk_profile_enable(2))     # every kernel family un-overlapped: one proof at a time on one stream
        for i in range(2):
            prover.prove_with(wb, *ds[i])
        fams["alone"] = collect_families(L, _lib, 2)
        fams["counters"] = collect_counters(L, _lib, 2)
        _lib.check(L.zk_profile_enable(0))
        _lib.check(L.zk_profile_reset())
        parity = None
        if not args.no_parity_gate:
            O = oracle()
            frs = lambda xs: bytes(RC.fr_bytes(xs))
            csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
            exp = O.pinocchio_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(tox), *(frs([x]) for x in ds[idx]))
            if proof.to_bytes() != exp:
                raise SystemExit("PARITY FAILURE: the last timed Pinocchio proof differs from the oracle's trapdoor evaluation")
            parity = {"checked": "last timed proof == oracle pinocchio_prove_trapdoor, 960 bytes"}
        return dt, lat, parity
    as_uploaded, derive_s = None, None
    if args.derive_lagrange_upto >= log_n:
        # the key as uploaded first, then with its h pool derived for the values of h (zk_pinocchio_pk_derive_lagrange, once per key, untimed)
        dt0, lat0, par0 = measure()
        as_uploaded = {"value": n * nproofs / dt0, "ms_per_proof": dt0 / nproofs * 1e3, "single_proof_latency_ms": lat0 * 1e3, "parity": par0 is not None}
        t_d = time.perf_counter()
        prover.derive_lagrange()
        derive_s = time.perf_counter() - t_d
    dt, lat, parity = measure()
    h_points = prover.pool_size(5)
    if as_uploaded is not None:
        saved = as_uploaded["ms_per_proof"] - dt / nproofs * 1e3
        as_uploaded["break_even_proofs"] = int(derive_s * 1e3 / saved) + 1 if saved > 0 else None
    prover.close()
    # ---- CPU baseline of THIS protocol: the oracle's restatement of ZKCompute.f (pinocchio.ml:427-514: one scalar multiplication per key
    # element touched, QAP.eval with schoolbook polynomials) on one host core at n = 1024; the GPU proof of the sample must be byte-identical
    cpu = None
    if not args.no_cpu_baseline:
        O = oracle()
        frs = lambda xs: bytes(RC.fr_bytes(xs))
        ns = 1024
        cs2, w2 = RC.iterated_cubic(ns, next(RC.fr_stream(0x5EED0001)))
        it2 = iter(tox)
        pk2, _ = PIN.ZK.keygen(lambda: next(it2), cs2)
        csr2 = [O.CSR(M.ptr, M.col, M.val) for M in (cs2.L, cs2.R, cs2.O)]
        q = O.QAP(cs2.n, cs2.m, *csr2)
        dvs = [frs([x]) for x in ds[0]]
        t_c = time.perf_counter()
        rc, ref = O.pinocchio_prove(q, bytes(pk2.g1), bytes(pk2.g2), cs2.mid, frs(w2), *dvs)
        cdt = time.perf_counter() - t_c
        p2 = PIN.ZK(cs2, pk2)
        got = p2.prove_with(w2, *ds[0]).to_bytes()
        p2.close()
        if rc != 0 or got != ref:
            raise SystemExit("PARITY FAILURE: GPU Pinocchio proof differs from the oracle on the cpu_baseline sample")
        cpu = {"value": ns / cdt, "unit": "constraints/s", "cores": 1, "kind": "port",
               "sample": "oracle restatement of pinocchio.ml:427-514 (ZKCompute.f: single scalar multiplications, schoolbook QAP.eval) on the iterated-cubic "
                         "R1CS at n=%d; %.2f s; GPU proof of the sample byte-identical" % (ns, cdt)}
    m_mid = cs.n_mid
    alg = 5 * 128 * m_mid + 2 * 128 * cs.m + 2 * 128 * n + 2 * 224 * m_mid + 192 * n       # 1792 n B at m_mid = m = n
    # scalar-point pairs per proof: five pools over I_mid (+ their appended single points), the h pool (n + 1 + 2m), two G2 pools over I_mid
    # -- the h pool as the prover holds it: n + 2 points (compact, derived), n + 1 (compact, as uploaded) or the reference's n + 1 + 2m (csrc/pinocchio.hip);
    # the kernel rooflines count the pairs the launches really carry, the whole-prove figure keeps the reference's 1792 B per constraint
    pairs = {"g1": 4 * (m_mid + 1) + (m_mid + 3) + h_points, "g2": 2 * (m_mid + 1)}
    roofs = roofline_objects(fams.get("timed", {}), nproofs, fams.get("alone", {}), 2, pairs, 1, 16, peak_products, None, "", fams.get("counters"))
    return {"workload": "pinocchio_zk_prove (BASELINE config 5), iterated-cubic R1CS, key+circuit+witness resident in HBM", "log_n": log_n, "constraints": n,
            "value": n * nproofs / dt, "unit": "constraints/s", "ms_per_proof": dt / nproofs * 1e3, "timed_s": dt, "timed_proofs": nproofs,
            "proofs_in_flight": depth, "single_proof_latency_ms": lat * 1e3, "single_proof_note": "one at a time, witness handed over as a host buffer",
            "h_pool_points": h_points, "h_pool_form": "compact: dw [v(s)] + dv [w(s)] ride on the bases of h (v_all | w_all checked against si at upload)" if h_points <= n + 2 else "si | v_all | w_all",
            "prove_algorithmic_bytes_per_constraint": alg / n, "prove_hbm_frac": alg / (dt / nproofs) / 1e9 / HBM_PEAK_GBS, "parity": parity,
            "derive_lagrange_s": None if derive_s is None else round(derive_s, 2), "as_uploaded": as_uploaded, "cpu_baseline": cpu,
            "roofline_g1": roofs.get("g1"), "roofline_g2": roofs.get("g2"),      # the six G1 / two G2 accumulate launches of a proof, as for Groth16
            "kernel_ms_per_proof": {k: round(v["ms_per_proof"], 4) for k, v in sorted(fams.get("alone", {}).items())}}


def summarize(res, world, peak_products, traffic, lagrange):
    """Public form of one Groth16 workload result: throughput, latency, both accumulate rooflines, whole-prove HBM fraction."""
    # the key's window width (groth16.hip upload): 20 bits from 2^21 points in the rank's G1 pool, 16 below (from 2^16 constraints up)
    cbits = int(os.environ.get("ZK_MSM_WINDOW", "20" if res["p1"] // world >= 1 << 21 else "16"))
    windows = 255 // cbits + 1
    key = "groth16_2^%d" % res["log_n"] + ("_derived" if res.get("derive_lagrange_s") is not None else ("_lagrange" if lagrange else ""))
    roofs = roofline_objects(res["fam_timed"], res["nproofs"], res["fam_alone"], res["n_alone"], res["pairs"], world, windows, peak_products, traffic, key, res.get("counters"))
    out = {k: res[k] for k in ("log_n", "constraints", "variables", "value", "unit", "ms_per_proof", "timed_s", "timed_proofs", "proofs_in_flight",
                               "single_proof_latency_ms", "single_proof_value", "setup_s", "derive_lagrange_s", "tau_power_form", "parity", "cpu_fast_context", "kernel_ms_per_proof")}
    out["prove_algorithmic_bytes_per_constraint"] = 928
    out["prove_hbm_frac"] = 928.0 * res["constraints"] / (res["ms_per_proof"] * 1e-3) / 1e9 / HBM_PEAK_GBS / world
    out["roofline_g1"], out["roofline_g2"] = roofs.get("g1"), roofs.get("g2")
    out["msm_window_bits"] = cbits
    return out, roofs


LINE_BUDGET = 4096          # bytes of the FINAL stdout line: the driver parses that line (a 24 KB one came back parsed = null in round 3)
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")


def _sig(x, digits=5):
    """floats to `digits` significant figures (the line is for reading and for the driver's parser, the detail file keeps everything)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(full):
    """The ONE line the driver parses, built from the full result dict (which goes to bench_detail.json and to stderr): the contract's fields, the dominant
    kernel's roofline, the CPU baseline, the parity verdict and one short record per other workload -- never more than LINE_BUDGET bytes.  Every derived-key
    figure keeps the tau-power figure of the SAME key beside it (`tau_power_value`)."""
    cfg = full.get("config") or {}
    tpf = cfg.get("tau_power_form") or {}
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    key_form = "lagrange_extension_uploaded" if "LAGRANGE-FORM KEY EXTENSION" in (cfg.get("workload") or "") else (
        "tau_powers_uploaded_lagrange_derived_on_device" if cfg.get("derive_lagrange_s") is not None else "tau_powers_as_uploaded")
    log_n = (int(cfg.get("constraints") or 1) - 1).bit_length()
    out["config"] = {"workload": "groth16_prove 2^%d, iterated-cubic R1CS, BLS12-381, %s" % (log_n, "key+circuit resident, witness handed over per proof (PCIe-inclusive)"
                                 if "PCIe-inclusive" in (cfg.get("workload") or "") else "key+circuit+witness resident in HBM"),
                     "constraints": cfg.get("constraints"), "key_form": key_form, "tau_power_value": tpf.get("value"),
                     "derive_lagrange_s": cfg.get("derive_lagrange_s"), "break_even_proofs": tpf.get("break_even_proofs"),
                     "proofs_in_flight": cfg.get("proofs_in_flight"), "proofs_per_step": cfg.get("proofs_per_step"),
                     "sharding": (cfg.get("sharding") or "")[:160], "fr_stage": (cfg.get("fr_stage") or None) and cfg["fr_stage"][:120], "rehearsal_ranks_share_gpus": cfg.get("rehearsal_ranks_share_gpus"), "device_list": cfg.get("device_list"),
                     "prove_algorithmic_bytes_per_constraint": cfg.get("prove_algorithmic_bytes_per_constraint"), "prove_hbm_frac": cfg.get("prove_hbm_frac")}
    out["ms_per_proof"] = full.get("ms_per_proof")
    out["single_proof_latency_ms"] = full.get("single_proof_latency_ms")
    roof = full.get("roofline")
    if roof:
        t = roof.get("one_proof_in_flight") or roof.get("timed_region") or {}
        out["roofline"] = {"kernel": roof.get("kernel"), "bound": roof.get("bound"), "peak": roof.get("peak"), "unit": roof.get("unit"),
                           "algorithmic_bytes_per_launch": t.get("algorithmic_bytes_per_launch"), "avg_launch_ms": t.get("avg_launch_ms"),
                           "achieved": roof.get("achieved"), "frac": roof.get("frac"), "traffic": roof.get("traffic"), "traffic_gbs": roof.get("traffic_gbs"), "traffic_frac": roof.get("traffic_frac"),
                           "alu_frac": (t.get("alu") or {}).get("frac"), "alu_frac_hw": (t.get("alu") or {}).get("frac_hw"),
                           "alu_peak_hw": "%.1f G Fp products/s = 1024 SIMDs x 2.4 GHz / 5 cyc per v_mad_u64_u32 x 64 lanes / 392 mads" % ALU_PEAK_HW,
                           "measured": roof.get("frac_is")}
    else:
        out["roofline"] = None
    cpu = full.get("cpu_baseline")
    if cpu:
        top = (cpu.get("ladder") or [{}])[-1]
        model = ((cpu.get("cost_model") or {}).get("at_benchmark_sizes") or {})
        out["cpu_baseline"] = {"value": cpu.get("value"), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"), "n": top.get("n"),
                               "seconds": top.get("seconds"), "model_2^20": model.get("2^20"),
                               "sample": "literal groth16.ml:116-161 + QAP.ml:120-135 on one host core, ladder n = %s; GPU proof of every sample byte-identical"
                                         % ",".join(str(p_.get("n")) for p_ in (cpu.get("ladder") or []))}
    else:
        out["cpu_baseline"] = None
    fast = full.get("cpu_fast_context")
    if fast:
        out["cpu_fast_context"] = {"value": fast.get("value"), "cores": fast.get("cores"), "kind": fast.get("kind"), "s_per_proof": fast.get("s_per_proof")}
    par = full.get("parity")
    out["parity"] = "passed: last timed proof == oracle trapdoor evaluation, bytes of a | b | c" if par else ("skipped" if par is None else par)
    out["other_workloads"] = [{"workload": (o.get("workload") or "")[:64], "value": o.get("value"),
                               "tau_power_value": ((o.get("tau_power_form") or o.get("as_uploaded") or {}).get("value")),
                               "derive_lagrange_s": o.get("derive_lagrange_s"), "parity": bool(o.get("parity"))} for o in (full.get("other_workloads") or [])]
    out["detail"] = "bench_detail.json (beside bench.py) and stderr: ladders, both accumulate rooflines, per-kernel ms, proof bytes"
    out = _sig(out)
    # hard cap: shed the optional blocks, least important first, until the line fits
    for drop in (None, "cpu_fast_context", "detail", "other_workloads", "single_proof_latency_ms"):
        if drop is not None:
            out.pop(drop, None)
        line = json.dumps(out, separators=(",", ":"))
        if len(line) <= LINE_BUDGET:
            return line
    raise SystemExit("bench.py: the result line does not fit %d bytes even without its optional blocks" % LINE_BUDGET)


def emit(full):
    """detail -> bench_detail.json + stderr; the compact line -> stdout (the LAST line of stdout)"""
    detail = json.dumps(full)
    try:
        with open(DETAIL_FILE, "w") as f:
            f.write(detail + "\n")
    except OSError as e:
        print("bench.py: could not write %s: %s" % (DETAIL_FILE, e), file=sys.stderr)
    print("BENCH_DETAIL " + detail, file=sys.stderr)
    sys.stderr.flush()
    line = compact_line(full)
    assert len(line) <= LINE_BUDGET and json.loads(line)["metric"] == full["metric"]
    print(line)
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps; a step = --proofs-per-step consecutive proofs of the pipelined prover")
    ap.add_argument("--warmup", type=int, default=2, help="untimed warm-up steps")
    ap.add_argument("--proofs-per-step", type=int, default=16)
    ap.add_argument("--log-n", type=int, default=20, help="log2 constraints of the headline workload (the WHOLE proof, whatever N: strong scaling; with --weak: per GPU)")
    ap.add_argument("--weak", action="store_true", help="N > 1: 2^log_n constraints PER GPU (n = 2^log_n x N, \"scaling\": \"weak\") instead of the same 2^log_n-constraint proof on every N; "
                    "BASELINE config 4 is --gpus 8 --log-n 19 --weak")
    ap.add_argument("--sizes", default="16,18,22", help="one GPU: further log2 sizes timed in the same run and reported under other_workloads ('' = none)")
    ap.add_argument("--config4-world", type=int, default=8, help="N at which the run ALSO measures BASELINE config 4 (2^config4-log-n constraints per rank, weak) into other_workloads")
    ap.add_argument("--config4-log-n", type=int, default=19)
    ap.add_argument("--dense-rows", type=int, default=20, help="one GPU: log2 size of the dense_rows workload (random R1CS, 8 entries per row; other_workloads only), -1 = skip")
    ap.add_argument("--no-pinocchio", action="store_true", help="skip the Pinocchio 2^18 workload (config 5) of the default run")
    ap.add_argument("--headline-only", action="store_true", help="only the headline workload (profiling runs): same as --sizes '' --no-pinocchio")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-budget", type=float, default=20.0, help="seconds of one host core the literal-algorithm ladder of cpu_baseline may take (n = 16, 64, 128, 256, 1024 while the fitted O(m n) law says the next size fits; "
                    "45 reaches n = 256 as rounds 2-3 did)")
    ap.add_argument("--cpu-fast-upto", type=int, default=20, help="workloads of at most 2^K constraints that hold Lagrange-form pools also run ONE proof on the multi-threaded CPU "
                    "prover of oracle/fast_cpu.c (context figure `cpu_fast_context`, BASELINE.md 3.3; ~0.5 s at 2^16, ~3 s at 2^18, ~12 s at 2^20 on 16 threads); -1 = never")
    ap.add_argument("--no-parity-gate", action="store_true", help="skip the oracle comparison of the last timed proof (profiling runs only)")
    ap.add_argument("--no-live-events", action="store_true", help="do not bracket the accumulate kernels with HIP events inside the timed region")
    ap.add_argument("--lagrange-key", action="store_true", help="one GPU: prove from the Lagrange-form EXTENSION of the key (scope row f4; "
                    "not the reference's key format -- the default and the headline use the tau-power key)")
    ap.add_argument("--derive-lagrange-upto", type=int, default=22, help="one GPU: for workloads of at most 2^K constraints, measure the reference-format key as uploaded, "
                    "then derive its Lagrange form on the device (zk_groth16_pk_derive_lagrange: once per key, untimed -- 2.7 s at 2^16, 6.7 s at 2^18, 31 s at 2^20, 147 s at 2^22) and "
                    "measure again: `value` is the derived key's figure, `tau_power_form` the other one.  -1 = never derive")
    ap.add_argument("--time-budget", type=float, default=330.0, help="seconds of wall clock the whole run aims to stay within: a derivation whose estimate (31 s x n / 2^20, 8 %% more above 2^20) does not fit what is left "
                    "is skipped and that workload reports its tau-power figure only (the default run is ~5 min with the 147 s derivation of the 2^22 key)")
    ap.add_argument("--host-witness", action="store_true", help="one GPU: hand the witness over as a host buffer with every proof (the PCIe-inclusive rate of DESIGN.md 8) instead of "
                    "proving from the copy made resident by zk_groth16_set_witness; never the headline")
    ap.add_argument("--derived-only", action="store_true", help="profiling runs: derive the key's Lagrange form right after the upload and measure only that path "
                    "(the default measures the key as uploaded first: its kernels would mix into rocprofv3's per-kernel statistics)")
    ap.add_argument("--tau-power-key", action="store_true", help="N > 1: keep the key in tau-power form (sharded at upload) instead of deriving the Lagrange form on every rank")
    ap.add_argument("--replicated-fr", action="store_true", help="N > 1: every rank runs the Fr stage of every proof (one collective per proof: the all-gather of the 768-byte "
                    "partial sums) instead of the default, the owner's Fr stage + an all-to-all of scalar slices (GroupProver)")
    ap.add_argument("--device-list", default="", help="ONE process, N devices behind one key handle (the path an OCaml host takes: zk_set_device_list, csrc/groth16_multi.hip): "
                    "comma-separated HIP device indices, e.g. 0,1,2,3,4,5,6,7; an index may repeat (several shards on one card: a rehearsal, never faster).  "
                    "Not the driver's --gpus N contract (that is one process per GPU over RCCL); n_gpus in the line = distinct devices of the list")
    ap.add_argument("--settle", type=float, default=4.0, help="max seconds of untimed load before timing so the clocks leave the idle state (0 = off)")
    ap.add_argument("--inflight", type=int, default=14, help="proofs kept in flight on one GPU, one stream each (1 = strictly serial)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver calls the N = 1 line: this process becomes the LAUNCHER.  It never touches HIP (no
        # torch.cuda call, no zk_init): the N ranks are fresh child processes under torch.distributed.run, rank 0's JSON line is relayed.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus (%d) must equal WORLD_SIZE (%d): a line for N GPUs is never measured on another number of ranks" % (args.gpus, world))

    dist = None
    rehearsal = False
    if world > 1:
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        if ndev >= world:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))      # RCCL over xGMI
        else:
            # fewer GPUs than ranks (a 1-GPU development box): ranks share devices and exchange over
            # gloo -- exercises the same code path, the numbers are not a scaling measurement
            rehearsal = True
            local_rank = local_rank % max(ndev, 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
    if args.lagrange_key and world > 1:
        raise SystemExit("--lagrange-key is a single-GPU option")

    from zukelang_amd import _lib
    L = _lib.lib()
    dev_list = [int(t) for t in args.device_list.split(",") if t.strip()]
    if dev_list:
        if world > 1:
            raise SystemExit("--device-list is the ONE-process multi-device path; --gpus N starts one process per GPU")
        _lib.set_device_list(dev_list)          # every key uploaded from here on is sharded over the list behind one handle
    else:
        _lib.check(L.zk_init(local_rank))

    # N > 1: a REHEARSAL first -- one tiny parity-gated proof through the distributed Fr stage (all-to-all of scalar slices + all-gather of partial sums
    # over RCCL).  If that path raises on this node (on every rank together: a collective the backend refuses), the SAME processes go on with the
    # replicated Fr stage (one all-gather per proof) instead of dying without a line; the line says which one ran.  A hang cannot be caught: none is known.
    fr_note = None
    replicated = args.replicated_fr
    if world > 1 and not replicated:
        import torch
        tiny = argparse.Namespace(**vars(args))
        tiny.weak = False
        err = ""
        try:
            bench_groth16(tiny, L, _lib, min(args.log_n, 10), 1, 0, 2, 0.0, rank, world, dist, derive_upto=None, events=False)
            ok = 1
        except BaseException as e:          # SystemExit of a parity failure included: wrong bytes are a reason to fall back AND to say so
            ok, err = 0, "%s: %s" % (type(e).__name__, str(e)[:200])
        flag = torch.tensor([ok], dtype=torch.int32)
        if dist.get_backend() == "nccl":
            flag = flag.cuda()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            replicated = True
            fr_note = "REPLICATED Fr stage: the distributed-Fr rehearsal (2^%d) failed on at least one rank (this rank: %s)" % (min(args.log_n, 10), err or "ok")
            print("bench.py rank %d: %s" % (rank, fr_note), file=sys.stderr)
        else:
            fr_note = "distributed Fr stage (rehearsed at 2^%d against the oracle before the timed run)" % min(args.log_n, 10)
    head = bench_groth16(args, L, _lib, args.log_n, args.steps, args.warmup, args.inflight, args.settle, rank, world, dist,
                         lagrange=args.lagrange_key, replicated_fr=replicated, events=not args.no_live_events,
                         derive_upto=args.derive_lagrange_upto if args.derive_lagrange_upto >= 0 else None)
    # BASELINE config 4 (2^22 constraints, MSM point-sharded over 8 ranks) beside the N = 8 line of the scaling curve: the driver's SCALE command
    # measures the SAME 2^log_n proof on every N (strong scaling); config 4 is `--gpus 8 --weak --log-n 19`, so the 8-rank run adds it itself
    # (reference-format key sharded at upload, no derivation: a few proofs, parity-gated)
    config4 = None
    if world > 1 and world == args.config4_world and not args.weak and not args.headline_only:
        a4 = argparse.Namespace(**vars(args))
        a4.weak = True
        # never at the price of the headline: the N = 8 line above is already measured; if this extra leg raises (memory, a parity failure, a collective the
        # node refuses) the run reports that instead of dying without its line
        try:
            r4 = bench_groth16(a4, L, _lib, args.config4_log_n, 1, 0, args.inflight, 0.0, rank, world, dist, replicated_fr=replicated, events=False, derive_upto=None)
        except BaseException as e:
            r4 = None
            print("bench.py rank %d: the config 4 leg failed and is left out: %s: %s" % (rank, type(e).__name__, str(e)[:300]), file=sys.stderr)
        if rank == 0 and r4 is not None:
            config4 = {"workload": "groth16_prove 2^%d point-sharded over %d ranks (BASELINE config 4: --weak --log-n %d), reference-format key as uploaded"
                                   % ((r4["constraints"] - 1).bit_length(), world, args.config4_log_n),
                       "value": r4["value"], "constraints": r4["constraints"], "ms_per_proof": r4["ms_per_proof"], "parity": r4["parity"], "tau_power_form": {"value": r4["value"]},
                       "derive_lagrange_s": None}
    peak = C.c_double()
    _lib.check(L.zk_bench_field_mul(1 | 8, 2000, C.byref(peak)))   # the library's dependent-chain product benchmark on this chip, in this process: best of 2 / 4 / 6 / 8 waves per SIMD
    traffic = pmc_traffic()
    head_pub, roofs = summarize(head, world, peak.value, traffic, args.lagrange_key)

    others = [config4] if config4 else []

    def other_size(ln, family="iterated_cubic"):
        per = {16: 320, 18: 80, 20: 24, 22: 8}.get(ln, max(8, int(0.6 / (2e-3 * (1 << max(0, ln - 16))))))       # proofs for >= 0.5 s at last round's rates
        infl = args.inflight if ln <= 20 else 4
        steps = max(1, (per + args.proofs_per_step - 1) // args.proofs_per_step)
        dense = family == "dense_rows"          # proved from the Lagrange-form EXTENSION the keygen emits: the derived key's prove path without paying its derivation again
        r = bench_groth16(args, L, _lib, ln, steps, 1 if ln <= 18 else 0, infl, args.settle if ln <= 18 else min(args.settle, 2.0), 0, 1, None,
                          lagrange=args.lagrange_key or dense, events=not args.no_live_events,
                          derive_upto=args.derive_lagrange_upto if args.derive_lagrange_upto >= 0 else None, family=family)
        pub, _ = summarize(r, 1, peak.value, traffic, args.lagrange_key or dense)
        pub["workload"] = ("groth16_prove 2^%d dense_rows (random R1CS, 8 entries per row in L, R and O; Lagrange-form key extension)" % ln if dense else
                           "groth16_prove 2^%d (BASELINE config %s), same run" % (ln, {16: "2", 20: "3's size", 22: "4's size on ONE GPU"}.get(ln, "-")))
        others.append(pub)

    # order: the short workloads, Pinocchio, the CPU ladder -- and the sizes above 2^20 LAST, so that the wall-clock guard of their (minutes-long)
    # derivation sees everything else already spent (--time-budget)
    sizes = [int(t) for t in args.sizes.split(",") if t.strip() and int(t) != args.log_n] if world == 1 and rank == 0 and not args.headline_only else []
    for ln in [x for x in sizes if x <= 20]:
        other_size(ln)
    if world == 1 and rank == 0 and not args.headline_only and args.dense_rows >= 0:
        if time.perf_counter() - T_START + 60 < args.time_budget:          # ~25 s of host-side generation + keygen at 2^20
            other_size(args.dense_rows, "dense_rows")
        else:
            print("bench.py: the dense_rows workload does not fit --time-budget: skipped", file=sys.stderr)
    if world == 1 and rank == 0 and not args.headline_only and not args.no_pinocchio:
        others.append(bench_pinocchio(args, L, _lib, 18, 48, 8, peak.value))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_baseline_budget)
    for ln in [x for x in sizes if x > 20]:
        other_size(ln)

    if rank == 0:
        # the dominant kernel = the accumulate family with the larger UN-OVERLAPPED time per proof (one proof in flight); N > 1: the
        # timed region's own events (no un-overlapped pass there)
        def weight(k):
            o = roofs[k]
            t = o.get("one_proof_in_flight") or o.get("timed_region")
            return t["avg_launch_ms"] * t["launches_per_proof"] if t else 0.0
        cands = [k for k in roofs if roofs[k].get("one_proof_in_flight") or roofs[k].get("timed_region")]
        roof = None
        if cands:
            dom = max(cands, key=weight)
            roof = dict(roofs[dom])
            roof["dominant_by"] = "un-overlapped time per proof (one proof in flight)" if roofs[dom].get("one_proof_in_flight") else "timed-region events"
            roof["note"] = ("algorithmic bytes = 128 B (G1) / 224 B (G2) per scalar-point pair; achieved / frac = the UN-OVERLAPPED launch (HIP events, one proof in flight: "
                            "the duration a reader recomputes from the rocprofv3 kernel stats under profiles/); `timed_region` = the same launch inside the timed region, stretched "
                            "by the other proofs sharing the SIMDs; the kernel is bound by the integer multiplier, not by HBM: see `alu` (products really executed, counted by the "
                            "library); traffic (PMC, profiles/%s_pmc_traffic.json) > algorithmic because the resident key stores one precomputed point per (point, window), see DESIGN.md" % PROFILE_ROUND)
        out = {
            "metric": "Groth16 constraints/sec on BLS12-381 at 1/2/4/8 MI355X; proof bit-exact",
            "value": head["value"],
            "unit": "constraints/s",
            "n_gpus": len(set(dev_list)) if dev_list else world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong",      # default: the SAME 2^log_n-constraint proof on every N (total work fixed)
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "groth16_prove, iterated-cubic R1CS (u -> u^3+u+3), BLS12-381, " + ("key+circuit resident in HBM, witness handed over as a HOST buffer with every proof (PCIe-inclusive)" if args.host_witness else "key+circuit+witness resident in HBM")
                                   + (", LAGRANGE-FORM KEY EXTENSION (not the reference key format)" if args.lagrange_key else "")
                                   + (", reference-format key (tau powers) uploaded, its Lagrange form DERIVED on the device once (zk_groth16_pk_derive_lagrange); tau_power_form = the same key before the derivation" if head.get("derive_lagrange_s") is not None else ""),
                       "step": "%d consecutive proofs of the pipelined prover (pipeline not drained between steps)" % args.proofs_per_step,
                       "proofs_per_step": args.proofs_per_step, "timed_proofs": head["timed_proofs"], "timed_s": head["timed_s"],
                       "constraints": head["constraints"], "variables": head["variables"], "proofs_in_flight": head["proofs_in_flight"], "constraints_per_gpu": head["constraints"] // world,
                       "sharding": ("MSM base points over ranks (slices cut for equal work: the A prefix counts twice); "
                                    + ("the ranks shared the derivation of the key's Lagrange form (one set per rank, broadcast: derive_lagrange_s) and kept their shards; " if head.get("derive_lagrange_s") is not None else "")
                                    + "Fr stage of a proof on its owner rank + all-to-all of scalar slices; all-gather of 768 B partial sums + local EC reduce"
                                    if head["group_batch"] is not None else
                                    ("MSM base points over ranks; every rank derived the key's Lagrange form (derive_lagrange_s) and runs the three-convolution Fr stage replicated; "
                                     "all-gather of 768 B partial sums + local EC reduce" if head.get("derive_lagrange_s") is not None else
                                     "MSM base points over ranks, Fr stage replicated; all-gather of 768 B partial sums + local EC reduce")) if world > 1 else
                                    ("ONE process, one key handle sharded over device list %s (zk_set_device_list): Fr stage on the slot's owner device, peer copies of the scalar slices, 768 B partial sums added on the first device" % dev_list
                                     if len(dev_list) > 1 else "single GPU"),
                       "exchange": head.get("exchange"),
                       "fr_stage": fr_note,
                       "derive_lagrange_s": head.get("derive_lagrange_s"),      # one-time, per key, outside the timed region
                       "tau_power_form": head.get("tau_power_form"),
                       "rehearsal_ranks_share_gpus": rehearsal or (len(dev_list) > len(set(dev_list))),
                       "device_list": dev_list or None,          # one process, one key handle over these devices (zk_set_device_list)
                       "prove_algorithmic_bytes_per_constraint": 928,
                       "prove_hbm_frac": head_pub["prove_hbm_frac"]},
            "ms_per_proof": head["ms_per_proof"],
            "single_proof_latency_ms": head["single_proof_latency_ms"],
            "single_proof_value": head["single_proof_value"],
            "parity": head["parity"],
            "roofline": roof,
            "roofline_g1": roofs.get("g1"),
            "roofline_g2": roofs.get("g2"),
            "cpu_baseline": cpu,
            "cpu_fast_context": head.get("cpu_fast_context"),        # same workload, host cores, Pippenger + NTT (not the reference's algorithm): context only
            "kernel_ms_per_proof": head["kernel_ms_per_proof"],
            "other_workloads": others,
            "proof_compressed_hex": head["proof_compressed_hex"],     # N > 1: rank 0 prints a proof it combined itself
        }
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
