"""Worker of tests/test_multiproc.py (one process per rank, started by torch.distributed.run).

mode "cpu": world_size ranks on the CPU with gloo.  The per-rank partial products are computed by
            the ORACLE over the rank's slice of the key pools (no GPU here), exchanged with the
            product's all_gather_bytes, summed, and compared with the single-rank oracle proof:
            covers the slicing rule, the scalar-vector layout, the exchange and its ordering.
mode "gpu-rccl": mode "gpu" over backend nccl (RCCL), one rank per device -- a world of one on the one-GPU box.
mode "gpu": the real path: every rank uploads its slice to the (shared) GPU, proves its partial
            sums with the HIP kernels, all-gathers 768 B per rank, combines on the GPU.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import oracle_lib as O  # noqa: E402
from oracle import pyref as P  # noqa: E402
from zukelang_amd import r1cs as RC  # noqa: E402
from zukelang_amd.groth16 import all_gather_bytes, msm_scalar_vectors, shard_bounds  # noqa: E402


def frs(xs):
    return b"".join(P.fr_to_bytes(x) for x in xs)


def main():
    mode = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    if mode == "gpu-rccl":
        # the RCCL branches of the collectives (device tensors, all_gather_into_tensor, all_to_all_single with splits, broadcast) on a REAL
        # communicator: one rank per device, so on the one-GPU box that is a world of one -- self-collectives through the code N ranks run
        # (tests/test_multiproc.py starts as many ranks as the box has cards, up to four: on an 8-GPU node RCCL runs BETWEEN devices with no edit)
        import torch
        dev = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cs, w = RC.iterated_cubic(n, 0xFEED)
    csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
    st = P.fr_stream(0x5EED0002)
    toxic = [next(st) for _ in range(5)]
    r, s = next(st), next(st)
    expect = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(toxic), P.fr_to_bytes(r), P.fr_to_bytes(s))
    if mode == "cpu":
        q = O.QAP(cs.n, cs.m, *csr)
        pk1, pk2, _, _ = q.groth16_setup(frs(toxic), cs.mid)
        v, ww, _ = q.eval_vwy(frs(w))
        rc, _, h = q.eval(frs(w))
        assert rc == 0
        sa, sc, sb = msm_scalar_vectors(n, RC.fr_ints(v), RC.fr_ints(ww), RC.fr_ints(h), w, cs.mid, r, s)
        p1, p2 = len(pk1) // 96, len(pk2) // 192
        assert len(sa) == len(sc) == p1 and len(sb) == p2
        lo1, hi1 = shard_bounds(p1, rank, world)
        lo2, hi2 = shard_bounds(p2, rank, world)
        _, A = O.g1_msm_naive(pk1[96 * lo1:96 * hi1], frs(sa[lo1:hi1]))
        _, Cc = O.g1_msm_naive(pk1[96 * lo1:96 * hi1], frs(sc[lo1:hi1]))
        _, B = O.g2_msm_naive(pk2[192 * lo2:192 * hi2], frs(sb[lo2:hi2]))
        gathered = bytes(all_gather_bytes(np.frombuffer(A + Cc + B, dtype=np.uint8), world))
        blk = 96 + 96 + 192
        inf1, inf2 = bytes([0x40]) + bytes(95), bytes([0x40]) + bytes(191)
        a, c, b = inf1, inf1, inf2
        for j in range(world):
            part = gathered[blk * j:blk * (j + 1)]
            a = O.g1_add(a, part[:96]); c = O.g1_add(c, part[96:192]); b = O.g2_add(b, part[192:])
        assert (a, b, c) == expect, "rank %d: sharded sum differs from the single-rank proof" % rank
        # distributed Fr stage (GroupProver's data flow with the oracle as the engine): rank j builds the scalar
        # vectors of proof j only, exchange_slices hands every rank its slice of every proof's vectors,
        # partial sums of all `world` proofs travel in one all-gather
        from zukelang_amd.groth16 import exchange_slices
        rs = [(next(st), next(st)) for _ in range(world)]
        rj, sj = rs[rank]
        va, vc, vb = msm_scalar_vectors(n, RC.fr_ints(v), RC.fr_ints(ww), RC.fr_ints(h), w, cs.mid, rj, sj)
        b1 = [shard_bounds(p1, g, world) for g in range(world)]
        b2 = [shard_bounds(p2, g, world) for g in range(world)]
        mine = []
        for vec, bounds in ((va, b1), (vc, b1), (vb, b2)):
            got = exchange_slices(np.frombuffer(frs(vec), dtype=np.uint8), bounds, rank, world)
            mine.append(bytes(got.numpy()))
        l1, l2 = 32 * (hi1 - lo1), 32 * (hi2 - lo2)
        parts = b""
        for j in range(world):
            _, A = O.g1_msm_naive(pk1[96 * lo1:96 * hi1], mine[0][l1 * j:l1 * (j + 1)])
            _, Cc = O.g1_msm_naive(pk1[96 * lo1:96 * hi1], mine[1][l1 * j:l1 * (j + 1)])
            _, B = O.g2_msm_naive(pk2[192 * lo2:192 * hi2], mine[2][l2 * j:l2 * (j + 1)])
            parts += A + Cc + B
        gathered = bytes(all_gather_bytes(np.frombuffer(parts, dtype=np.uint8), world))
        for j in range(world):
            a, c, b = inf1, inf1, inf2
            for g in range(world):
                part = gathered[blk * (world * g + j):blk * (world * g + j + 1)]
                a = O.g1_add(a, part[:96]); c = O.g1_add(c, part[96:192]); b = O.g2_add(b, part[192:])
            ej = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(toxic), P.fr_to_bytes(rs[j][0]), P.fr_to_bytes(rs[j][1]))
            assert (a, b, c) == ej, "rank %d: distributed-Fr group proof %d differs" % (rank, j)
        # the RCCL branch of exchange_slices (all_to_all_single with per-destination splits) has never run on hardware: drive its
        # split arithmetic here with an emulation of the collective on top of gloo, full and clipped (non-zero prefix) bounds alike
        import torch
        from zukelang_amd.groth16 import clip_bounds
        real_backend, real_a2a = dist.get_backend, getattr(dist, "all_to_all_single")

        def fake_all_to_all_single(out, inp, out_splits, in_splits):
            blobs = [None] * world
            dist.all_gather_object(blobs, (bytes(inp.numpy().tobytes()), list(in_splits)))
            pos = 0
            for j, (data, splits) in enumerate(blobs):
                start = sum(splits[:rank])
                assert splits[rank] == out_splits[j], "split mismatch between sender %d and receiver %d" % (j, rank)
                assert sum(splits) == len(data), "sender %d: splits do not cover its input" % j
                if out_splits[j]:
                    out[pos:pos + out_splits[j]] = torch.frombuffer(bytearray(data[start:start + splits[rank]]), dtype=torch.uint8)
                pos += out_splits[j]
            assert pos == out.numel()
        try:
            dist.get_backend = lambda *a, **k: "nccl"
            dist.all_to_all_single = fake_all_to_all_single
            for vec, bounds in ((va, b1), (vb, b2), (va, clip_bounds(b1, 3 + n + 2)), (va, clip_bounds(b1, 1)), (vc, clip_bounds(b1, p1 - 1))):
                mine_b = bounds[rank]
                got = exchange_slices(torch.from_numpy(np.frombuffer(frs(vec), dtype=np.uint8).copy()), bounds, rank, world)
                # the same exchange through the gloo branch is the expectation: rank j's vector is msm_scalar_vectors(..., rs[j])
                exp = b""
                for j in range(world):
                    vj = msm_scalar_vectors(n, RC.fr_ints(v), RC.fr_ints(ww), RC.fr_ints(h), w, cs.mid, *rs[j])
                    src = vj[0] if vec is va else (vj[1] if vec is vc else vj[2])
                    exp += frs(src[mine_b[0]:mine_b[1]])
                assert bytes(got.numpy().tobytes()) == exp, "rank %d: all_to_all split arithmetic" % rank
            # the A vector really is zero beyond its prefix: clipping loses nothing
            assert not any(va[3 + n + 2:])
        finally:
            dist.get_backend, dist.all_to_all_single = real_backend, real_a2a
        # collective error agreement: a failure only ONE rank sees reaches every rank before the next collective
        from zukelang_amd.groth16 import agree_on_status
        assert agree_on_status(0) == 0
        assert agree_on_status(-4 if rank == world - 1 else 0) == -4
        assert agree_on_status(-5 if rank == 0 else (-4 if rank == 1 else 0)) == -5
    else:
        from zukelang_amd import _lib
        from zukelang_amd.groth16 import Groth16
        # one rank per card where the box has several (the ranks of the gloo rehearsal share the one card otherwise)
        _lib.check(_lib.lib().zk_init(int(os.environ.get("LOCAL_RANK", "0")) % max(1, _lib.lib().zk_device_count())))
        it = iter(toxic)
        pk, _ = Groth16.keygen(lambda: next(it), cs)
        prover = Groth16(cs, pk, rank, world)
        proof = prover.prove_rs(w, r, s)
        assert (proof.a, proof.b, proof.c) == expect, "rank %d: sharded GPU proof differs" % rank
        # pipelined: three sharded proofs in flight, collected in order (every rank runs the same
        # sequence of all-gathers)
        prover.set_witness(w)
        for slot in range(3):
            prover.prove_async(None, r, s, slot)
        for slot in range(3):
            p2 = prover.prove_wait(slot)
            assert (p2.a, p2.b, p2.c) == expect, "rank %d slot %d: pipelined sharded proof differs" % (rank, slot)
        # distributed Fr stage: proofs in groups of `world`, rank j runs the Fr stage of the j-th proof only
        from zukelang_amd.groth16 import GroupProver
        rs = [(r, s)] + [(next(st), next(st)) for _ in range(2 * world)]          # 2 full groups + 1 proof
        gp = GroupProver(prover, batch=3 if world == 2 else 4)      # rounds that are NOT a multiple of the world size: ownership rotates
        got = gp.prove_many(rs)
        assert len(got) == len(rs)
        assert (got[0].a, got[0].b, got[0].c) == expect, "rank %d: group proof 0 differs" % rank
        for (rr, ss), pr in zip(rs[1:], got[1:]):
            e2 = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(toxic), P.fr_to_bytes(rr), P.fr_to_bytes(ss))
            assert (pr.a, pr.b, pr.c) == e2, "rank %d: a group proof differs" % rank
        start = gp.count                                              # proof i of the job belongs to rank (start + i) % world
        own = gp.prove_many(rs[:world + 1], combine_all=False)
        mine = [i for i in range(world + 1) if (start + i) % world == rank]
        assert [i for i, x in enumerate(own) if x is not None] == mine
        for i in mine:
            assert (own[i].a, own[i].b, own[i].c) == tuple(getattr(got[i], f) for f in "abc")
        # round 2: the derived Lagrange form on N ranks -- every rank uploads the key WHOLE, derives, keeps its shard (zk_groth16_pk_shard); the
        # Fr stage (three convolutions) is then cheap enough to run replicated, and the only collective left is the all-gather of 768 bytes
        dp = Groth16(cs, pk)
        dp.derive_lagrange()
        dp.shard(rank, world)
        pd = dp.prove_rs(w, r, s)
        assert (pd.a, pd.b, pd.c) == expect, "rank %d: proof from the derived + sharded key differs" % rank
        dp.set_witness(w)
        for slot, (rr, ss) in enumerate(rs[1:4]):
            dp.prove_async(None, rr, ss, slot)
        for slot, (rr, ss) in enumerate(rs[1:4]):
            p3 = dp.prove_wait(slot)
            e3 = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(toxic), P.fr_to_bytes(rr), P.fr_to_bytes(ss))
            assert (p3.a, p3.b, p3.c) == e3, "rank %d slot %d: pipelined proof from the derived + sharded key differs" % (rank, slot)
        # ... and with the DISTRIBUTED Fr stage on top of it (bench.py's default at N > 1): the owner of a proof runs the three-convolution
        # Fr stage, the scalar slices travel (cut for equal work: the A prefix counts twice), every rank multiplies its slice
        gd = GroupProver(dp, batch=3 if world == 2 else 4)
        assert gd.bounds1[rank] == (gd.lo1, gd.hi1)
        gdp = gd.prove_many(rs[:2 * world + 1])
        for (rr, ss), pr in zip(rs[:2 * world + 1], gdp):
            e4 = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(toxic), P.fr_to_bytes(rr), P.fr_to_bytes(ss))
            assert (pr.a, pr.b, pr.c) == e4, "rank %d: a group proof from the derived + sharded key differs" % rank
        # round 3: the derivation SHARED by the ranks (one set per rank, broadcast, install of the rank's shard): the pools a rank ends up with
        # must be the ones the redundant derivation + zk_groth16_pk_shard left it with, byte for byte, and so must the proofs
        ds = Groth16(cs, pk)
        ds.derive_lagrange_shared(rank, world)
        assert bytes(ds.pool_points(1)) == bytes(dp.pool_points(1)) and bytes(ds.pool_points(2)) == bytes(dp.pool_points(2)), "rank %d: shared derivation gives other pools" % rank
        ps = ds.prove_rs(w, r, s)
        assert (ps.a, ps.b, ps.c) == expect, "rank %d: proof from the shared derivation differs" % rank
        ds.close()
        dp.close()
        # an unsatisfied witness: only the OWNER of a proof sees ZK_ERR_REMAINDER (QAP.ml:134); every rank must raise
        # before the round's all-to-all instead of hanging in it, and the prover must stay usable afterwards
        w_bad = list(w)
        w_bad[3] = (w_bad[3] + 1) % RC.FR_MODULUS
        prover.set_witness(w_bad)
        try:
            gp.prove_many(rs[:1])            # ONE proof: a single rank owns it
            raise SystemExit("rank %d: bad witness was not reported" % rank)
        except AssertionError:
            pass
        prover.set_witness(w)
        again = gp.prove_many(rs[:world])
        for (rr, ss), pr in zip(rs[:world], again):
            e2 = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(toxic), P.fr_to_bytes(rr), P.fr_to_bytes(ss))
            assert (pr.a, pr.b, pr.c) == e2, "rank %d: proof after the failed round differs" % rank
        prover.close()
        if mode == "gpu-rccl":
            import torch
            from zukelang_amd.groth16 import agree_on_status, exchange_slices
            assert dist.get_backend() == "nccl"
            blk = np.arange(768, dtype=np.uint8)
            assert bytes(all_gather_bytes(blk, world)) == bytes(blk) * world
            vec = torch.from_numpy(np.frombuffer(frs(list(range(1, 41))), dtype=np.uint8).copy()).cuda()
            got = exchange_slices(vec, [(5, 29)], rank, world)
            assert got.is_cuda and bytes(got.cpu().numpy()) == frs(list(range(6, 30)))
            assert agree_on_status(-4) == -4
    dist.barrier()
    if rank == 0:
        print("MP-OK", mode, world, n)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
