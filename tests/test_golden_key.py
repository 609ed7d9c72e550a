"""tests/golden/readme_groth16_key.json -- the README circuit's Groth16 proving key, its Lagrange form and a proof, computed from first
principles with Python integers (tests/golden/make_readme_keys.py) -- against the three implementations that must reproduce it:
the C oracle (setup, literal prove, trapdoor prove), the multi-threaded CPU context prover, and on the GPU the keygen, the on-device
derivation of the Lagrange form (which never sees tau) and the prover on every key form."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from oracle import pyref as P
from zukelang_amd import r1cs as RC

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "readme_groth16_key.json")))
cat = lambda xs: b"".join(bytes.fromhex(x) for x in xs)
PK1, PK2, LG1, LG2 = cat(FIX["pk_g1"]), cat(FIX["pk_g2"]), cat(FIX["lagrange_g1"]), cat(FIX["lagrange_g2"])
PROOF = tuple(bytes.fromhex(FIX["proof"][k]) for k in "abc")
TOX = [int(FIX["toxic"][k], 16) for k in ("alpha", "beta", "gamma", "delta", "tau")]
R_, S_ = int(FIX["r"], 16), int(FIX["s"], 16)
W = [int(x, 16) for x in FIX["witness"]]
frb = P.fr_to_bytes
frs = lambda xs: b"".join(frb(x) for x in xs)


def _circuit():
    cs, w = RC.readme_circuit(3)
    assert w == W and list(cs.mid) == FIX["mid"]
    return cs, [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]


def test_oracle_reproduces_the_fixture():
    cs, csr = _circuit()
    q = O.QAP(cs.n, cs.m, *csr)
    pk1, pk2, _, _ = q.groth16_setup(frs(TOX), cs.mid)
    assert pk1 == PK1 and pk2 == PK2
    rc, a, b, c = q.groth16_prove(PK1, PK2, cs.mid, frs(W), frb(R_), frb(S_), 1)          # literal groth16.ml:116-161
    assert rc == 0 and (a, b, c) == PROOF
    assert O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(W), frs(TOX), frb(R_), frb(S_)) == PROOF
    fp = O.FastGroth16(cs.n, cs.m, *csr, cs.mid, LG1, LG2, 2)                              # Pippenger + NTT over the Lagrange form
    rc, a, b, c = fp.prove(frs(W), frb(R_), frb(S_))
    fp.close()
    assert rc == 0 and (a, b, c) == PROOF


@pytest.mark.gpu
def test_gpu_reproduces_the_fixture():
    from zukelang_amd.groth16 import Groth16, PKey
    cs, _ = _circuit()
    it = iter(TOX)
    pk, _vk = Groth16.keygen(lambda: next(it), cs, lagrange=True)
    assert bytes(pk.g1) == PK1 and bytes(pk.g2) == PK2 and bytes(pk.lag_g1) == LG1 and bytes(pk.lag_g2) == LG2
    # from the FIXTURE's bytes (not this library's keygen): as uploaded, derived on the device, uploaded in Lagrange form
    key = PKey(np.frombuffer(PK1, dtype=np.uint8), np.frombuffer(PK2, dtype=np.uint8), np.frombuffer(LG1, dtype=np.uint8), np.frombuffer(LG2, dtype=np.uint8))
    pr = Groth16(cs, key)
    p = pr.prove_rs(W, R_, S_)
    assert (p.a, p.b, p.c) == PROOF
    pr.derive_lagrange()
    assert bytes(pr.pool_points(1)) == LG1 and bytes(pr.pool_points(2)) == LG2
    p = pr.prove_rs(W, R_, S_)
    assert (p.a, p.b, p.c) == PROOF
    pr.close()
    pl = Groth16(cs, key, lagrange=True)
    p = pl.prove_rs(W, R_, S_)
    assert (p.a, p.b, p.c) == PROOF
    pl.close()


# ---------------------------------------------------------------- Pinocchio Protocol 2 (tests/golden/make_readme_pinocchio.py)
PFIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "readme_pinocchio_key.json")))
PPK1, PPK2, PDER = cat(PFIX["pk_g1"]), cat(PFIX["pk_g2"]), cat(PFIX["derived_h_pool_g1"])
PPROOF = bytes.fromhex(PFIX["proof"])
PTOX = [int(x, 16) for x in PFIX["toxic"]]
PDEL = [int(x, 16) for x in PFIX["deltas"]]


def test_oracle_reproduces_the_pinocchio_fixture():
    cs, csr = _circuit()
    assert [int(x, 16) for x in PFIX["witness"]] == W
    ex = O.pinocchio_keygen_exponents(None, cs.n, cs.m, *csr, cs.mid, frs(PTOX), False)
    assert O.points_of_exponents_g1(ex[0]) == PPK1 and O.points_of_exponents_g2(ex[1]) == PPK2
    q = O.QAP(cs.n, cs.m, *csr)
    rc, proof = O.pinocchio_prove(q, PPK1, PPK2, cs.mid, frs(W), *(frb(d) for d in PDEL))      # literal ZKCompute.f
    assert rc == 0 and proof == PPROOF
    assert O.pinocchio_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(W), frs(PTOX), *(frb(d) for d in PDEL)) == PPROOF


@pytest.mark.gpu
def test_gpu_reproduces_the_pinocchio_fixture():
    from zukelang_amd import pinocchio as PIN
    cs, _ = _circuit()
    it = iter(PTOX)
    pk, _vk = PIN.ZK.keygen(lambda: next(it), cs)
    assert bytes(pk.g1) == PPK1 and bytes(pk.g2) == PPK2
    key = PIN.PKey(np.frombuffer(PPK1, dtype=np.uint8), np.frombuffer(PPK2, dtype=np.uint8))       # the FIXTURE's bytes
    pp = PIN.ZK(cs, key)
    assert pp.prove_with(W, *PDEL).to_bytes() == PPROOF
    n, m = cs.n, cs.m
    assert bytes(pp.pool_points(5)) == PPK1[96 * 15:96 * (15 + n + 1)]                              # the compact h pool: si alone (v_all | w_all passed the consistency check)
    pp.derive_lagrange()
    assert bytes(pp.pool_points(5)) == PDER                                                        # [lambda_t(s)] | [Z(s)] | [1] | [s^(n-1)], derived without s
    assert pp.prove_with(W, *PDEL).to_bytes() == PPROOF
    pp.close()
    # the full pool (ZK_PIN_COMPACT_H=0): si | v_all | w_all as uploaded, [lambda_t(s)] | [Z(s)] | [1] | v_all | w_all derived
    from zukelang_amd import _lib
    _lib.check(_lib.lib().zk_set_option(b"ZK_PIN_COMPACT_H", b"0"))
    try:
        pp = PIN.ZK(cs, key)
    finally:
        _lib.check(_lib.lib().zk_set_option(b"ZK_PIN_COMPACT_H", None))
    assert bytes(pp.pool_points(5)) == PPK1[96 * 15:96 * (15 + n + 1 + 2 * m)]
    assert pp.prove_with(W, *PDEL).to_bytes() == PPROOF
    pp.derive_lagrange()
    assert bytes(pp.pool_points(5)) == cat(PFIX["derived_h_pool_g1_full"])
    assert pp.prove_with(W, *PDEL).to_bytes() == PPROOF
    pp.close()


@pytest.mark.gpu
def test_a_plain_c_host_proves_the_fixture(tmp_path):
    """examples/c_prove.c: upload, prove, derive, read back, prove -- all through the C-ABI from C99, bytes against the fixture (no ctypes,
    no Python in the loop: the process only starts the binary)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_prove")
    libdir = os.path.join(root, "zukelang_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "examples"),
                           os.path.join(root, "examples", "c_prove.c"), "-o", exe, "-L" + libdir, "-lzkmi355x", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith(b"c-prove ok (1 device entry)"), (out.returncode, out.stdout, out.stderr)
    # the same binary with a device list: a multi-device key behind the one handle (zk_set_device_list; the one card of the test box listed twice and
    # three times -- with several MI355X, `c_prove 0 1 2`), same calls, same first-principles bytes
    for devs in (["0", "0"], ["0", "0", "0"]):
        out = subprocess.run([exe] + devs, capture_output=True, timeout=300)
        assert out.returncode == 0 and out.stdout.startswith(b"c-prove ok (%d device entries)" % len(devs)), (devs, out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
def test_a_plain_c_host_runs_pinocchio_like_the_ocaml_shim(tmp_path):
    """examples/c_pinocchio.c: the call sequence of ocaml/pinocchio_mi355x.ml (keygen through zk_g1/g2_of_fr, upload, ZK and NonZK prove, verify,
    derive, pipelined prove) from C99, every byte against the first-principles fixture tests/golden/readme_pinocchio_key.json."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_pinocchio")
    libdir = os.path.join(root, "zukelang_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "examples"),
                           os.path.join(root, "examples", "c_pinocchio.c"), "-o", exe, "-L" + libdir, "-lzkmi355x", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith(b"c-pinocchio ok (1 device entry)"), (out.returncode, out.stdout, out.stderr)
    # the same binary with a device list (round 5: Pinocchio keys shard over it too): the one card of the test box listed twice and three times
    for devs in (["0", "0"], ["0", "0", "0"]):
        out = subprocess.run([exe] + devs, capture_output=True, timeout=300)
        assert out.returncode == 0 and out.stdout.startswith(b"c-pinocchio ok (%d device entries)" % len(devs)), (devs, out.returncode, out.stdout, out.stderr)


def test_the_pinocchio_fixture_header_is_the_json():
    """examples/readme_pinocchio_fixture.h is generated from the JSON fixture (tests/golden/make_readme_c_header.py): same bytes."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = open(os.path.join(root, "examples", "readme_pinocchio_fixture.h")).read()
    arrays = {m.group(1): bytes(int(x) for x in m.group(2).split(",")) for m in re.finditer(r"static const uint8_t (\w+)\[\d+\] = \{([^}]*)\};", h)}
    assert arrays["PFIX_PK_G1"] == PPK1 and arrays["PFIX_PK_G2"] == PPK2 and arrays["PFIX_PROOF"] == PPROOF and arrays["PFIX_DERIVED_H_POOL"] == PDER
    assert arrays["PFIX_VK_G1"] == cat(PFIX["vk_g1"]) and arrays["PFIX_VK_G2"] == cat(PFIX["vk_g2"]) and arrays["PFIX_PROOF_NONZK"] == bytes.fromhex(PFIX["proof_nonzk"])
    # and the new parts of the JSON against the oracle: verification key and NonZK proof
    cs, csr = _circuit()
    ex = O.pinocchio_keygen_exponents(None, cs.n, cs.m, *csr, cs.mid, frs(PTOX), False)
    assert O.points_of_exponents_g1(ex[2]) == cat(PFIX["vk_g1"]) and O.points_of_exponents_g2(ex[3]) == cat(PFIX["vk_g2"])
    z = frb(0)
    assert O.pinocchio_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(W), frs(PTOX), z, z, z) == bytes.fromhex(PFIX["proof_nonzk"])
    assert O.pinocchio_verify(cat(PFIX["vk_g1"]), cat(PFIX["vk_g2"]), [int(x, 16) for x in PFIX["io"]], PPROOF)
